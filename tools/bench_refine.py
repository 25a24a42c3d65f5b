#!/usr/bin/env python3
"""What a densification (refine) step of DefaultStrategy costs at the c4 scene size: the c4 training
step with the strategy attached, `refine_every` shortened so that refine steps occur inside the
timed window; prints the mean time of a plain step, of a refine step, and the growth of N.

    python tools/bench_refine.py [--gaussians 1000000] [--steps 24] [--refine-every 8]
"""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--steps", type=int, default=24)
    ap.add_argument("--refine-every", type=int, default=8)
    ap.add_argument("--grow-grad2d", type=float, default=2e-6,
                    help="threshold on the mean pixel-space gradient (the scene is synthetic: chosen so that "
                         "a few per cent of the Gaussians are duplicated / split per refine step)")
    ap.add_argument("--tensor-ops", action="store_true",
                    help="the torch formulation (duplicate / split / remove as tensor operations) instead of the one-pass kernels")
    args = ap.parse_args()
    import torch

    from tests import scenes
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    distributed = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
    S = importlib.import_module("3dgs_monocular_depth_init_amd.strategy")
    dev = torch.device("cuda", 0)
    N = args.gaussians
    sc = scenes.make_scene(N, 0)
    splats, optimizers = runner.create_splats_with_optimizers(
        sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), sh_degree=3, batch_size=1, device=str(dev),
        world_size=1, quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
    optimizers = distributed.fuse_optimizers(splats, optimizers)
    optimizers.fuse_into_backward(True)
    vms, Ks = scenes.cameras(range(100), width=1920, height=1080)
    c2ws = torch.linalg.inv(vms).contiguous().to(dev)
    Ks = Ks.to(dev)
    gen = torch.Generator().manual_seed(2)
    targets = [torch.rand(1, 1080, 1920, 3, generator=gen).to(dev) for _ in range(4)]
    cfg = runner.RasterConfig(sh_degree=3)
    strat = S.DefaultStrategy(refine_start_iter=0, refine_every=args.refine_every, reset_every=10 ** 9,
                              grow_grad2d=args.grow_grad2d, verbose=False, one_pass=not args.tensor_ops)
    state = strat.initialize_state(scene_scale=1.0)
    strat.check_sanity(splats, optimizers)
    plain, refine, sizes, post = [], [], [], []
    orig_post = strat.step_post_backward

    def timed_post(*a, **k):                      # the strategy's own share of a refine step
        torch.cuda.synchronize()
        t = time.perf_counter()
        orig_post(*a, **k)
        torch.cuda.synchronize()
        post.append((time.perf_counter() - t, strat.mutates_params(a[3])))

    object.__setattr__(strat, "step_post_backward", timed_post)
    for k in range(1, args.steps + 1):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        runner.train_step(splats, optimizers, c2ws[k % 100:k % 100 + 1], Ks[k % 100:k % 100 + 1], targets[k % 4], k, cfg,
                          strategy=strat, strategy_state=state)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        if k > args.refine_every:                       # the first window warms everything up
            (refine if strat.mutates_params(k) else plain).append(dt)
        sizes.append(len(splats["means"]))
    print(json.dumps({"metric": "DefaultStrategy refine step @ c4 scene", "gaussians_start": N, "gaussians_end": sizes[-1],
                      "plain_step_ms": 1e3 * sum(plain) / max(len(plain), 1),
                      "refine_step_ms": 1e3 * sum(refine) / max(len(refine), 1), "refine_steps": len(refine),
                      "refine_every_used": args.refine_every, "formulation": "tensor ops" if args.tensor_ops else "one pass",
                      "step_post_backward_ms_on_refine_steps": 1e3 * sum(t for t, r in post[args.refine_every:] if r)
                      / max(sum(1 for t, r in post[args.refine_every:] if r), 1),
                      "step_post_backward_ms_on_plain_steps": 1e3 * sum(t for t, r in post[args.refine_every:] if not r)
                      / max(sum(1 for t, r in post[args.refine_every:] if not r), 1),
                      "amortised_over_100_steps_ms": (1e3 * sum(refine) / max(len(refine), 1)
                                                      - 1e3 * sum(plain) / max(len(plain), 1)) / 100.0,
                      "N_after_each_step": sizes[::args.refine_every]}))


if __name__ == "__main__":
    main()
