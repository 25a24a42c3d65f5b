#!/usr/bin/env python3
"""BASELINE config c2 as a step time: 100 k Gaussians, FOUR views per step at 1080p (one 4-camera `runner.train_step`,
L1 loss, Adam fused into the backward), and the same with 1 M Gaussians; next to it the host's cost of issuing a 4-camera
step (a 2 000-Gaussian scene at 96 x 64).   python tools/bench_c2.py [--steps 60]"""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=60)
ap.add_argument("--views", type=int, default=4)
args = ap.parse_args()
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
V = args.views


def case(name, N, W, H, **scene_kw):
    sc = scenes.make_scene(N, 3, **scene_kw)
    splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                        opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    fused = D.fuse_optimizers(splats, opts)
    fused.fuse_into_backward(True)
    kw = dict(width=W, height=H) if W == 1920 else dict(width=W, height=H, f=90.0, dist=2.5)
    vms, Ks = scenes.cameras(range(100), **kw)
    c2ws, Ks = torch.linalg.inv(vms).contiguous().cuda(), Ks.cuda()
    target = torch.rand(V, H, W, 3, device="cuda")
    cfg = runner.RasterConfig()

    def run(n, k0):
        for k in range(n):
            a = (V * (k0 + k)) % (100 - V)
            runner.train_step(splats, fused, c2ws[a:a + V], Ks[a:a + V], target, step=10_000 + k0 + k, cfg=cfg)

    run(10, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, 10)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    print(json.dumps({"case": name, "gaussians": N, "views_per_step": V, "ms_per_step": round(ms, 4), "views_per_s": round(V * 1e3 / ms, 1)}), flush=True)
    fused.fuse_into_backward(False)
    del splats, opts, fused


case("host only (2 000 Gaussians, 96 x 64)", 2000, 96, 64, box=(1.0, 0.7, 0.4), scale_mean=0.03)
case("c2: 100 k Gaussians", 100_000, 1920, 1080)
case("c2 shape, 1 M Gaussians", 1_000_000, 1920, 1080)
