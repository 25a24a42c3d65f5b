#!/usr/bin/env python3
"""Phase timeline of bucket_emit_kernel (diagnostic build -DGSR_EMIT_TIMELINE=1): s_memtime at
the phase boundaries of every workgroup -> mean cycles per phase. Shares only."""
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
fn = lib.gsr_debug_set_emit_timeline
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 1_000_000
sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
buf = torch.zeros(264, 8, dtype=torch.int64, device=dev)
assert fn(buf.data_ptr()) == 0
for cam in (0, 0, 25):
    vm, K = scenes.cameras([cam], width=W, height=H)
    buf.zero_()
    with torch.no_grad():
        R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]),
                        vm.to(dev), K.to(dev), W, H, sh_degree=3, packed=False, _tight_tiles=True)
    torch.cuda.synchronize()
b = buf.double().cpu()
b = b[(b[:, 5] > 0) & (b[:, 2] > 0)]          # workgroups that ran all phases (the publisher leaves early)
names = ["load counts + order (wg 0) + scan", "pass 1: rectangle histogram", "reservation (global atomics)",
         "pass 2: masks + scatter", "sentinels + real counts"]
d = {n: float((b[:, i + 1] - b[:, i]).mean()) for i, n in enumerate(names)}
d["whole workgroup"] = float((b[:, 5] - b[:, 0]).mean())
d["first start -> last end"] = float(b[:, 5].max() - b[:, 0].min())
d["workgroup 0 whole"] = float(b[0, 5] - b[0, 0])
print(json.dumps({"unit": "shader cycles (s_memtime)", "phases": d}, indent=1))
