#!/usr/bin/env python3
"""Race screen of the compositing forward (LDS-DMA staged records and pair words): it has no atomics, so every
launch at the c4 scene -- alone and with a memory-hungry kernel on a second stream -- must reproduce the first
image bit for bit."""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import scenes  # noqa: E402

R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
dev = torch.device("cuda", 0)
W, H, N = 1920, 1080, 1_000_000
sc = {k: v.to(dev) for k, v in scenes.make_scene(N, 0).items()}
side = torch.cuda.Stream()
noise = torch.randn(64 << 20, device=dev)
bad = 0
for cam in (0, 25, 60):
    vm, K = scenes.cameras([cam], width=W, height=H)
    vm, K = vm.to(dev), K.to(dev)

    def run():
        with torch.no_grad():
            img, alpha, _ = R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]),
                                            vm, K, W, H, sh_degree=3, packed=False, _tight_tiles=True)
        return img, alpha
    first, first_a = run()
    diffs = 0
    for rep in range(60):
        if rep % 3 == 0:
            with torch.cuda.stream(side):
                noise.mul_(1.0000001)
        img, alpha = run()
        if not (torch.equal(img, first) and torch.equal(alpha, first_a)):
            diffs += 1
    torch.cuda.synchronize()
    print(f"camera {cam}: renders differing from the first: {diffs} / 60", flush=True)
    bad += diffs
sys.exit(1 if bad else 0)
