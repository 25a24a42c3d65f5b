#!/usr/bin/env python3
"""Is the c4 training step host-bound? Times the Python loop that ISSUES n steps (no synchronisation
inside) and the wall time until the GPU has finished them: if issuing takes as long as the whole run,
the GPU waits for the host between kernels.

    python tools/host_vs_gpu.py [--steps 100]
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
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    args = ap.parse_args()
    import torch
    from tests import scenes
    runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
    distributed = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
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

    def step(k):
        runner.train_step(splats, optimizers, c2ws[k % 100:k % 100 + 1], Ks[k % 100:k % 100 + 1], targets[k % 4],
                          step=10_000 + k, cfg=cfg)

    for k in range(10):
        step(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for k in range(args.steps):
        step(10 + k)
    t_issue = time.perf_counter() - t0
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    print(json.dumps({"steps": args.steps, "issue_ms_per_step": 1e3 * t_issue / args.steps,
                      "wall_ms_per_step": 1e3 * t_all / args.steps,
                      "host_is_ahead_by_ms_at_the_end": 1e3 * (t_all - t_issue)}))


if __name__ == "__main__":
    main()
