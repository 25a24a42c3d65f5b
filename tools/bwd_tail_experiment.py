#!/usr/bin/env python3
"""How much of the compositing backward's launch is the under-occupied tail of its schedule? 8 160 tiles (one wave each,
longest first) run on 6 144 wave slots (6 waves/SIMD): the first 6 144 tiles start together, the 2 016 shortest fill in as
slots free up. Needs a library built with -DGSR_EXPERIMENT_KNOBS=1 (GSR_BWD_TILE_LIMIT = n: only the n longest tiles are
launched; -n: the n longest are skipped; timing only, the gradients are wrong). Prints, per limit, the backward's time and
the share of the (tile, Gaussian) pairs the launched tiles hold.
    GSRAST_LIB=.../libgsrast_knobs.so python tools/bwd_tail_experiment.py"""
import importlib
import json
import os
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))

if len(sys.argv) > 1 and sys.argv[1] == "--worker":
    import torch
    from tests import scenes
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
    N = 1_000_000
    sc = scenes.make_scene(N, 0)
    splats, _ = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                     opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    vm, K = scenes.cameras([0])
    c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
    target = torch.rand(1, 1080, 1920, 3, device="cuda")

    def step():
        _, info = runner.train_step(splats, None, c2w, K, target, step=10_000)
        for p in splats.values():
            p.grad = None
        return info

    for _ in range(3):
        info = step()
    torch.cuda.synchronize()
    L.TIMERS = {}
    L.TIMER_ONLY = {"gsr_rasterize_bwd"}
    for _ in range(10):
        step()
    torch.cuda.synchronize()
    t = L.kernel_times_ms()["gsr_rasterize_bwd"][1]
    off = info["isect_offsets"].reshape(-1).long()
    lens = torch.diff(torch.cat([off, torch.tensor([info["flatten_ids"].numel()], device=off.device)]))
    srt = torch.sort(lens, descending=True).values.cumsum(0).float() / float(lens.sum())
    lim = int(os.environ.get("GSR_BWD_TILE_LIMIT", "0"))
    n = lens.numel()
    share = 1.0 if lim == 0 else (float(srt[lim - 1]) if lim > 0 else 1.0 - float(srt[-lim - 1]))
    print(json.dumps({"limit": lim, "tiles_launched": n if lim == 0 else (lim if lim > 0 else n + lim), "share_of_pairs": round(share, 4),
                      "raster_bwd_ms": round(t, 4), "tile_len_min_median_max": [int(lens.min()), int(lens.median()), int(lens.max())]}))
else:
    for lim in (0, 6144, 7168, 4096, 2048, -6144, 0):
        env = dict(os.environ, GSR_BWD_TILE_LIMIT=str(lim))
        out = subprocess.run([sys.executable, __file__, "--worker"], env=env, capture_output=True, text=True)
        print(out.stdout.strip().splitlines()[-1] if out.stdout.strip() else out.stderr[-400:], flush=True)
