#!/usr/bin/env python3
"""In-kernel timeline of bucket_sort_kernel (diagnostic build -DGSR_SORT_TIMELINE=1): per workgroup
s_memrealtime at start / offsets + prefix done / 8-bin count done / keys placed in LDS / bitonic done / lists written."""
import ctypes as C
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
lib = pkg._lib.load()
fn = getattr(lib, "gsr_debug_set_sort_timeline", None)
if fn is None:
    sys.exit("this libgsrast.so was not built with -DGSR_SORT_TIMELINE=1")
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 1_000_000
sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
buf = torch.zeros(1100, 8, dtype=torch.int64, device=dev)
assert fn(buf.data_ptr()) == 0
for cam in (0, 0, 25):
    vm, K = scenes.cameras([cam], width=W, height=H)
    buf.zero_()
    with torch.no_grad():
        R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]),
                        vm.to(dev), K.to(dev), W, H, sh_degree=3, packed=False, _tight_tiles=True)
    torch.cuda.synchronize()
b = buf.double().cpu() * 0.01                      # us
b = b[b[:, 5] > 0]
names = ["keys loaded, prefix of real counts, depth range", "(tile, depth-bin) histogram", "scan + keys placed in LDS by bin",
         "in-bin ranks + permutation (or networks)", "lists written"]
d = {n: round(float((b[:, i + 1] - b[:, i]).mean()), 3) for i, n in enumerate(names)}
d["whole workgroup mean / max"] = [round(float((b[:, 5] - b[:, 0]).mean()), 3), round(float((b[:, 5] - b[:, 0]).max()), 3)]
d["first start -> last end"] = round(float(b[:, 5].max() - b[:, 0].min()), 3)
d["workgroups"] = int(len(b))
print(json.dumps({"unit": "us (s_memrealtime)", "phases": d}, indent=1))
