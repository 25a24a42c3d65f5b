#!/usr/bin/env python3
"""In-kernel timelines of the compositing forward and backward (diagnostic build -DGSR_RASTER_TIMELINE=1):

  bash tools/build_variants.sh rtl "-DGSR_RASTER_TIMELINE=1"
  GSRAST_LIB=.../lib/variants/libgsrast_rtl.so python tools/raster_timeline.py > gpurun_out/raster_timeline.json

Per tile (= wave) the instrumented kernels add up s_memtime segments: waiting for the staged batch (vmcnt),
issuing the next batch's LDS-DMAs (backward: + flushing the previous batch's gradient rows by atomics), the
compositing loop, the rest. The stamps serialise what the product kernels overlap: read the SHARES."""
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
fns = []
for name in ("gsr_debug_set_fwd_timeline", "gsr_debug_set_bwd_timeline"):
    fn = getattr(lib, name, None)
    if fn is None:
        sys.exit("this libgsrast.so was not built with -DGSR_RASTER_TIMELINE=1")
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    fns.append(fn)
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 1_000_000
sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
bufs = [torch.zeros(n_tiles, 8, dtype=torch.int64, device=dev) for _ in range(2)]
for fn, b in zip(fns, bufs):
    assert fn(b.data_ptr()) == 0
target = torch.rand(1, H, W, 3, device=dev)
out = {}
for cam in (0, 25):
    vm, K = scenes.cameras([cam], width=W, height=H)
    p = {k: sc[k].clone().requires_grad_(True) for k in ("means", "quats", "scales", "opacities", "sh0", "shN")}
    for b in bufs:
        b.zero_()
    img, _, _ = R.rasterization(p["means"], p["quats"], p["scales"], p["opacities"], (p["sh0"], p["shN"]),
                                vm.to(dev), K.to(dev), W, H, sh_degree=3, packed=False, _tight_tiles=True)
    (img - target).abs().mean().backward()
    torch.cuda.synchronize()
    rec = {}
    for name, b in zip(("forward", "backward"), bufs):
        t = b.double().cpu()
        live = t[:, 4] > 0
        tot = t[live, 4].sum()
        seg = {"wait_for_staged_batch": t[live, 0].sum() / tot,
               ("flush_rows_and_issue_dma" if name == "backward" else "issue_next_dma"): t[live, 1].sum() / tot,
               "compositing_loop": t[live, 2].sum() / tot, "rest": t[live, 3].sum() / tot}
        nb, npair = t[live, 5].sum(), t[live, 6].sum()
        rec[name] = {"tiles": int(live.sum()), "pairs": int(npair), "batches": int(nb),
                     "share_of_wave_time": {k: round(float(v), 4) for k, v in seg.items()},
                     "loop_cycles_per_pair": round(float(t[live, 2].sum() / npair), 1),
                     "wave_cycles_per_pair": round(float(tot / npair), 1),
                     "wait_cycles_per_batch": round(float(t[live, 0].sum() / nb), 1),
                     "wave_lifetime_cycles_mean_p50_p99": [round(float(t[live, 4].mean())), round(float(t[live, 4].median())),
                                                           round(float(torch.quantile(t[live, 4], 0.99)))]}
    out[f"camera {cam}"] = rec
print(json.dumps({"note": "diagnostic build; shares only", "scene": "c4, tight lists", "runs": out}, indent=1))
