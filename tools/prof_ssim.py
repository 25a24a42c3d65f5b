#!/usr/bin/env python3
"""Twenty fused L1 + SSIM forward / backward passes at 1080p (for rocprofv3 --pmc / --kernel-trace runs)."""
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
a = torch.rand(1, 1080, 1920, 3, device="cuda", requires_grad=True)
b = torch.rand(1, 1080, 1920, 3, device="cuda")
for _ in range(20):
    loss = L.l1_ssim_loss(a, b, 0.2)
    loss.backward()
    a.grad = None
torch.cuda.synchronize()
