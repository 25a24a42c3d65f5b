#!/usr/bin/env python3
"""Host cost of one training step: the same `runner.train_step` (fused Adam in the backward) on a scene so small
that the GPU work is negligible (2 000 Gaussians, 96 x 64), so the wall time per step IS the Python / ctypes /
autograd / allocator time of issuing the step's launches. Compare with the c4 step's GPU time: the step is
host-bound when they meet.   python tools/host_step_cost.py [--steps 300] [--ssim]"""
import argparse
import cProfile
import importlib
import json
import pstats
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=300)
ap.add_argument("--ssim", action="store_true")
ap.add_argument("--profile", action="store_true")
args = ap.parse_args()
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
N, W, H = 2000, 96, 64
sc = scenes.make_scene(N, 3, box=(1.0, 0.7, 0.4), scale_mean=0.03)
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts)
fused.fuse_into_backward(True)
vms, Ks = scenes.cameras(range(100), width=W, height=H, f=90.0, dist=2.5)
c2ws, Ks = torch.linalg.inv(vms).contiguous().cuda(), Ks.cuda()
target = torch.rand(1, H, W, 3, device="cuda")
cfg = runner.RasterConfig()
lam = 0.2 if args.ssim else 0.0


def run(n, k0):
    for k in range(n):
        runner.train_step(splats, fused, c2ws[k % 100:k % 100 + 1], Ks[k % 100:k % 100 + 1], target, step=10_000 + k0 + k,
                          cfg=cfg, ssim_lambda=lam)


run(30, 0)
torch.cuda.synchronize()
t0 = time.perf_counter()
run(args.steps, 30)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
print(json.dumps({"host_ms_per_step": 1e3 * dt / args.steps, "steps": args.steps, "ssim": args.ssim}))
if args.profile:
    pr = cProfile.Profile()
    pr.enable()
    run(200, 1000)
    pr.disable()
    pstats.Stats(pr).sort_stats("cumulative").print_stats(35)
