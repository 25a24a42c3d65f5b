#!/usr/bin/env python3
"""In-kernel timeline of the compositing forward (diagnostic build -DGSR_FWD_TIMELINE=1).

  bash tools/build_variants.sh fwdtl "-DGSR_FWD_TIMELINE=1"
  GSRAST_LIB=.../lib/variants/libgsrast_fwdtl.so python tools/fwd_timeline.py > gpurun_out/fwd_timeline.json

Reads the per-tile segment sums the instrumented raster_fwd_kernel leaves in a side buffer
(s_memtime stamps at: batch top / gathered records landed / LDS image written / next loads
issued / compositing loop done) and prints their SHARES. The instrumented build's run time
is not a measurement of the product kernel (its fences forbid overlaps the real kernel has).
"""
from __future__ import annotations

import ctypes as C
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

import torch  # noqa: E402

from tests import scenes  # noqa: E402


def main():
    pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
    R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
    lib = pkg._lib.load()
    fn = getattr(lib, "gsr_debug_set_fwd_timeline", None)
    if fn is None:
        sys.exit("this libgsrast.so was not built with -DGSR_FWD_TIMELINE=1")
    fn.argtypes = [C.c_void_p]
    fn.restype = C.c_int
    dev = torch.device("cuda", 0)
    W, H = 1920, 1080
    N = 1_000_000
    sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
    n_tiles = ((W + 15) // 16) * ((H + 15) // 16)
    buf = torch.zeros(n_tiles, 8, dtype=torch.int64, device=dev)
    assert fn(buf.data_ptr()) == 0
    out = []
    for cam in (0, 0, 25):           # first one warms up
        vm, K = scenes.cameras([cam], width=W, height=H)
        buf.zero_()
        with torch.no_grad():
            R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"],
                            (sc["sh0"], sc["shN"]), vm.to(dev), K.to(dev), W, H, sh_degree=3, packed=False)
        torch.cuda.synchronize()
        b = buf.double().cpu()
        wait, rec, issue, loop, total, batches, pairs, t0 = (b[:, i] for i in range(8))
        live = total > 0
        tot = total[live].sum()
        seg = {"wait_for_gather": wait[live].sum() / tot, "make_rec_lds_barrier": rec[live].sum() / tot,
               "issue_next_loads": issue[live].sum() / tot, "compositing_loop": loop[live].sum() / tot}
        seg["outside_batches"] = 1.0 - sum(seg.values())
        nb = batches[live].sum()
        npair = pairs[live].sum()
        t_start = t0[live]
        rec_ = {
            "camera": cam, "tiles": int(live.sum()), "pairs": int(npair), "batches": int(nb),
            "share_of_wave_time": {k: round(float(v), 4) for k, v in seg.items()},
            "cycles_per_batch": {"wait_for_gather": float(wait[live].sum() / nb),
                                 "make_rec_lds_barrier": float(rec[live].sum() / nb),
                                 "issue_next_loads": float(issue[live].sum() / nb),
                                 "compositing_loop": float(loop[live].sum() / nb)},
            "loop_cycles_per_pair": float(loop[live].sum() / npair),
            "wave_cycles_per_pair": float(tot / npair),
            "wave_lifetime_cycles_mean_p50_p99": [float(total[live].mean()), float(total[live].median()),
                                                  float(torch.quantile(total[live], 0.99))],
            "dispatch_spread_cycles": float(t_start.max() - t_start.min()),
        }
        out.append(rec_)
    print(json.dumps({"note": "diagnostic build; shares only", "runs": out[1:]}, indent=1))


if __name__ == "__main__":
    main()
