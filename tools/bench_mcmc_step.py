#!/usr/bin/env python3
"""The "mcmc" preset's step at c4 size (1 M Gaussians, 1080p, L1 loss): MCMCStrategy's position noise on every step,
opacity_reg = scale_reg = 0.01 (trainer.py:83-92), between refine steps.   python tools/bench_mcmc_step.py [--steps 40]"""
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
ap.add_argument("--steps", type=int, default=40)
args = ap.parse_args()
P = "3dgs_monocular_depth_init_amd."
runner = importlib.import_module(P + "runner")
D = importlib.import_module(P + "distributed")
S = importlib.import_module(P + "strategy")
N, W, H = 1_000_000, 1920, 1080
sc = scenes.make_scene(N, 0)
vms, Ks = scenes.cameras(range(100))
c2ws, Ks = torch.linalg.inv(vms).contiguous().cuda(), Ks.cuda()
target = torch.rand(1, H, W, 3, device="cuda")
cfg = runner.RasterConfig()
for name in ("default strategy (statistics only), no regularisers", "mcmc preset"):
    splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                        opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    fused = D.fuse_optimizers(splats, opts)
    fused.fuse_into_backward(True)
    if name == "mcmc preset":
        strat, kw = S.MCMCStrategy(cap_max=N), dict(opacity_reg=0.01, scale_reg=0.01)
    else:
        strat, kw = S.DefaultStrategy(), {}
    state = strat.initialize_state()

    def run(n, k0):
        for k in range(n):
            step = 10_001 + (k0 + k) % 98
            runner.train_step(splats, fused, c2ws[k % 100:k % 100 + 1], Ks[k % 100:k % 100 + 1], target, step=step, cfg=cfg,
                              strategy=strat, strategy_state=state, **kw)

    run(5, 0)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    run(args.steps, 5)
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / args.steps
    print(json.dumps({"case": name, "ms_per_step": round(ms, 4), "iters_per_s": round(1e3 / ms, 1)}), flush=True)
    fused.fuse_into_backward(False)
    del splats, opts, fused
