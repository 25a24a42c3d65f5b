#!/usr/bin/env python3
"""Time bucket_count + bucket_emit ALONE on the c4 projection (nothing downstream consumes the keys,
so knock-out builds of the emit kernel -- -DGSR_EMIT_KO=n, tools/build_variants.sh -- are safe here).

  GSRAST_LIB=.../lib/variants/libgsrast_ko1.so python tools/emit_knockout.py
"""
import importlib
import json
import os
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
from importlib import import_module  # noqa: E402
L = import_module("3dgs_monocular_depth_init_amd._lib")
call, ptr = L.call, L.ptr
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 1_000_000
CACHE = ROOT / "gpurun_out" / "proj_c4.pt"
if not CACHE.exists():            # projection by the in-tree library only (run this once without GSRAST_LIB)
    assert "GSRAST_LIB" not in os.environ, "make the projection cache with the in-tree library first"
    sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
    vm, K = scenes.cameras([0], width=W, height=H)
    with torch.no_grad():
        _, _, meta = R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]),
                                     vm.to(dev), K.to(dev), W, H, sh_degree=3, packed=False, _tight_tiles=True)
    torch.save({k: meta[k].contiguous().cpu() for k in ("means2d", "radii", "depths", "conics")} |
               {"opac": torch.sigmoid(sc["opacities"]).cpu()}, CACHE)
d = torch.load(CACHE, weights_only=True)
means2d, radii, depths, conics, opac = (d[k].to(dev).contiguous() for k in ("means2d", "radii", "depths", "conics", "opac"))
tile_w, tile_h = (W + 15) // 16, (H + 15) // 16
bw = (tile_w + 7) // 8
nb = tile_h * bw
scratch = torch.zeros(3, nb, dtype=torch.int32, device=dev)
counts, cursor, real = scratch[0], scratch[1], scratch[2]
offsets = torch.empty(nb + 1, dtype=torch.int32, device=dev)
order = torch.empty(nb, dtype=torch.int32, device=dev)
cap = 4_000_000
keys = torch.empty(cap, dtype=torch.int64, device=dev)
st = torch.cuda.current_stream().cuda_stream
wg = torch.empty(256, nb, dtype=torch.int32, device=dev)
WG = os.environ.get("NO_WG_HIST") is None          # per-workgroup counts handed from count to emit
res = {}
for tight in (1, 0):
    ts = []
    for it in range(25):
        scratch.zero_()
        e0, e1, e2 = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        e0.record()
        call("gsr_bucket_count", 1, N, ptr(means2d), ptr(radii), tile_w, tile_h, ptr(counts), ptr(cursor), ptr(real), 1, ptr(wg) if WG else None, st)
        e1.record()
        call("gsr_bucket_emit", 1, N, ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opac), 0, tile_w, tile_h,
             tight, ptr(counts), ptr(cursor), ptr(real), ptr(offsets), ptr(order), None, None, ptr(keys), cap, ptr(wg) if WG else None, st)
        e2.record()
        torch.cuda.synchronize()
        if it >= 5:
            ts.append((e0.elapsed_time(e1), e1.elapsed_time(e2)))
    t = torch.tensor(ts)
    res["tight" if tight else "rect"] = {"count_ms": round(float(t[:, 0].median()), 4), "emit_ms": round(float(t[:, 1].median()), 4),
                                         "slots": int(offsets[nb].item()), "listed": int(real.sum().item())}
print(os.environ.get("GSRAST_LIB", "head").split("_")[-1], json.dumps(res))
