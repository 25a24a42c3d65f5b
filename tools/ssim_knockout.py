#!/usr/bin/env python3
"""SSIM forward with and without its three derivative-map stores (train=True / False), 1080p, events around 50 calls."""
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
a = torch.rand(1, 1080, 1920, 3, device="cuda", requires_grad=True).permute(0, 3, 1, 2)
b = torch.rand(1, 1080, 1920, 3, device="cuda").permute(0, 3, 1, 2)
ap = torch.rand(1, 3, 1080, 1920, device="cuda", requires_grad=True)
bp = torch.rand(1, 3, 1080, 1920, device="cuda")
for name, x, y in (("NHWC memory (the rasterizer's output)", a, b), ("planar NCHW memory", ap, bp)):
    for train in (True, False):
        for _ in range(5):
            L.fused_ssim(x, y, padding="valid", train=train)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            L.fused_ssim(x, y, padding="valid", train=train)
        e1.record()
        torch.cuda.synchronize()
        print(name, "maps written" if train else "no maps", "%.1f us per call (forward + finalize)" % (e0.elapsed_time(e1) * 1e3 / 50))

# forward / backward launches of the fused L1 + SSIM loss on channel-interleaved vs planar memory, events around the C calls
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
for name, x, y in (("NHWC memory", a, b), ("planar NCHW memory", ap, bp)):
    x = x.detach().requires_grad_(True)
    lib.TIMERS, lib.TIMER_ONLY = {}, {"gsr_ssim_l1_fwd", "gsr_ssim_l1_bwd"}
    for it in range(30):
        _, _, loss = L._SsimL1.apply(x, y.detach(), True, True, 0.2)
        loss.backward()
        x.grad = None
    torch.cuda.synchronize()
    t = lib.kernel_times_ms()
    lib.TIMERS, lib.TIMER_ONLY = None, None
    print(name, {k: round(v[1] * 1e3, 1) for k, v in t.items()}, "us per call")
