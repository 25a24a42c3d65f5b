#!/usr/bin/env python3
"""Soak run of the training loop at a realistic size: 300 k -> densified Gaussians, 1080p, 60 cameras,
DefaultStrategy (refine every 100 steps, one opacity reset), fused Adam in the backward on the steps
without refinement, tight tile lists, deferred tile-list sizing (overflow rebuilds included).
Checks: finite losses, falling loss, Gaussian count moves, no GPU fault.   python tools/soak_train.py [--steps 1500]"""
import argparse
import importlib
import json
import math
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--steps", type=int, default=1500)
ap.add_argument("--gaussians", type=int, default=300_000)
ap.add_argument("--grow-grad2d", type=float, default=0.0002)
ap.add_argument("--mcmc", action="store_true", help="MCMCStrategy (relocation + growth to cap_max, noise, both regularisers)")
args = ap.parse_args()
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
S = importlib.import_module("3dgs_monocular_depth_init_amd.strategy")
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
torch.manual_seed(0)
W, H, N = 1920, 1080, args.gaussians
gt = scenes.make_scene(N, 3)
cams = list(range(0, 100, 100 // 60 or 1))[:60]
vms, Ks = scenes.cameras(cams, width=W, height=H)
c2ws, Ks = torch.linalg.inv(vms).cuda(), Ks.cuda()
gt_splats, _ = runner.create_splats_with_optimizers(gt["means"], torch.rand(N, 3), torch.log(gt["scales"]), quats=gt["quats"],
                                                    opacities_logit=torch.logit(gt["opacities"]), shN=gt["shN"])
with torch.no_grad():
    gt_splats["sh0"].copy_(gt["sh0"].cuda())
    targets = [runner.rasterize_splats(gt_splats, c2ws[i:i + 1], Ks[i:i + 1], W, H, sh_degree=3)[0].clamp(0, 1).detach()
               for i in range(len(cams))]
del gt_splats
n0 = N // 2
pts = gt["means"][:n0] + 0.01 * torch.randn(n0, 3)
knn = importlib.import_module("3dgs_monocular_depth_init_amd.knn")
splats, opts = runner.create_splats_with_optimizers(pts, torch.rand(n0, 3), knn.initial_log_scales(pts.cuda()).cpu(), init_opacity=0.3)
fused = D.fuse_optimizers(splats, opts)
fused.fuse_into_backward(True)
if args.mcmc:
    strat = S.MCMCStrategy(cap_max=int(1.5 * n0), refine_start_iter=100, refine_every=100, refine_stop_iter=args.steps - 100)
    strat.check_sanity(splats, fused)
    st = strat.initialize_state()
    reg = dict(opacity_reg=0.01, scale_reg=0.01)           # the "mcmc" preset (trainer.py:83-92)
else:
    strat = S.DefaultStrategy(refine_start_iter=100, refine_every=100, reset_every=700, refine_stop_iter=args.steps - 100, grow_grad2d=args.grow_grad2d)
    strat.check_sanity(splats, fused)
    st = strat.initialize_state(scene_scale=1.0)
    reg = {}
losses, counts = [], []
t0 = time.perf_counter()
try:
    for step in range(args.steps):
        i = (step * 7) % len(cams)
        loss, info = runner.train_step(splats, fused, c2ws[i:i + 1], Ks[i:i + 1], targets[i], step=step, strategy=strat,
                                       strategy_state=st, **reg)
        if step % 50 == 0 or step == args.steps - 1:
            losses.append(float(loss))
            counts.append(len(splats["means"]))
finally:
    R.set_backward_optimizer(None)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
ok = all(math.isfinite(x) for x in losses) and losses[-1] < 0.8 * losses[0] and len(set(counts)) > 1
print(json.dumps({"steps": args.steps, "seconds": round(dt, 1), "ms_per_step": round(dt / args.steps * 1e3, 3), "loss_first_last": [losses[0], losses[-1]],
                  "gaussians_first_max_last": [counts[0], max(counts), counts[-1]], "ok": ok}))
sys.exit(0 if ok else 1)
