"""GPU parity of the fused L1 + SSIM loss against oracle/loss_oracle.py
(tolerance: loss 1e-4 relative to the fp64 oracle, gradient 1e-3 L2-relative and
1e-3*max element-wise)."""
import importlib

import pytest
import torch

from oracle import loss_oracle as LO

pytestmark = pytest.mark.gpu


def _imgs(N, H, W, seed):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    base = torch.stack([xx, yy, 0.5 * (xx + yy)], -1)[None].repeat(N, 1, 1, 1)
    a = (base + 0.2 * torch.rand(N, H, W, 3, generator=g)).clamp(0, 1.3)
    b = (base + 0.2 * torch.rand(N, H, W, 3, generator=g)).clamp(0, 1)
    return a, b


@pytest.mark.parametrize("shape", [(1, 37, 53), (2, 64, 96), (1, 270, 480)])
@pytest.mark.parametrize("padding", ["same", "valid"])
def test_fused_ssim_matches_oracle(shape, padding):
    L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    N, H, W = shape
    a, b = _imgs(N, H, W, 1)
    # fp64 oracle as the yardstick: sigma = E[x^2] - mu^2 cancels catastrophically in
    # fp32, so two fp32 implementations (2-D conv vs separable) differ by ~2e-5 relative
    a_c = a.permute(0, 3, 1, 2).contiguous().double().requires_grad_(True)
    ref = LO.fused_ssim(a_c, b.permute(0, 3, 1, 2).contiguous().double(), padding)
    ref.backward()
    a_g = a.permute(0, 3, 1, 2).contiguous().cuda().requires_grad_(True)
    got = L.fused_ssim(a_g, b.permute(0, 3, 1, 2).contiguous().cuda(), padding=padding)
    got.backward()
    assert float(got) == pytest.approx(float(ref), rel=1e-4, abs=1e-6)
    g, r = a_g.grad.cpu().double(), a_c.grad
    assert float((g - r).norm() / r.norm()) < 1e-3
    assert float((g - r).abs().max()) <= 1e-3 * float(r.abs().max())


def test_l1_ssim_loss_nhwc_in_place_and_scaled_upstream():
    L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    a, b = _imgs(1, 128, 160, 2)
    a_c = a.clone().requires_grad_(True)
    (LO.l1_ssim_loss(a_c, b, 0.2) * 3.0).backward()
    a_g = a.clone().cuda().requires_grad_(True)
    loss = L.l1_ssim_loss(a_g, b.cuda(), 0.2)
    (loss * 3.0).backward()
    assert float(loss) == pytest.approx(float(LO.l1_ssim_loss(a, b, 0.2)), rel=1e-5)
    g, r = a_g.grad.cpu(), a_c.grad
    assert g.shape == r.shape == (1, 128, 160, 3)
    assert float((g - r).norm() / r.norm()) < 1e-3
    # identical images: SSIM = 1, L1 = 0
    same = L.l1_ssim_loss(b.cuda().requires_grad_(True), b.cuda(), 0.2)
    assert float(same) == pytest.approx(0.0, abs=1e-6)


def test_c4_sized_image_properties():
    """1080p: SSIM of an image with itself is 1; symmetric in its arguments."""
    L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    g = torch.Generator().manual_seed(0)
    a = torch.rand(1, 3, 1080, 1920, generator=g).cuda()
    b = torch.rand(1, 3, 1080, 1920, generator=g).cuda()
    assert float(L.fused_ssim(a, a, padding="valid", train=False)) == pytest.approx(1.0, abs=1e-5)
    s_ab = float(L.fused_ssim(a, b, padding="valid", train=False))
    s_ba = float(L.fused_ssim(b, a, padding="valid", train=False))
    assert s_ab == pytest.approx(s_ba, rel=1e-5) and 0.0 < s_ab < 0.2


@pytest.mark.parametrize("n", [(1, 7, 5, 3), (1, 128, 160, 3)])
def test_l1_loss_matches_torch(n):
    L = importlib.import_module("3dgs_monocular_depth_init_amd.losses")
    g = torch.Generator().manual_seed(0)
    a, b = torch.rand(n, generator=g), torch.rand(n, generator=g)
    a_c = a.clone().requires_grad_(True)
    (torch.nn.functional.l1_loss(a_c, b) * 2.5).backward()
    a_g = a.clone().cuda().requires_grad_(True)
    loss = L.l1_loss(a_g, b.cuda())
    (loss * 2.5).backward()
    assert float(loss) == pytest.approx(float(torch.nn.functional.l1_loss(a, b)), rel=1e-6)
    assert torch.allclose(a_g.grad.cpu(), a_c.grad, rtol=1e-6, atol=1e-9)
