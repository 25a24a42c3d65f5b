"""Seeded synthetic depth/SfM scene shared by the golden generators (no reference code)."""
import torch


def depth_scene(H, W, seed, n_sfm, outlier_frac=0.2, noise=0.02):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    t = torch.clamp((xx + 0.5 * yy) / 1.5, 0, 1)
    depth = 2.0 + 6.0 * (t * t * (3 - 2 * t)) + 0.05 * torch.randn(H, W, generator=g)
    mask = torch.rand(H, W, generator=g) > 0.05
    xs = torch.randint(0, W, (n_sfm,), generator=g)
    ys = torch.randint(0, H, (n_sfm,), generator=g)
    coords = torch.stack([xs, ys]).long()
    gt = 1.7 * depth[ys, xs] + 0.4 + noise * torch.randn(n_sfm, generator=g)
    outl = torch.rand(n_sfm, generator=g) < outlier_frac
    gt = torch.where(outl, gt * (0.3 + 2.7 * torch.rand(n_sfm, generator=g)), gt)
    return depth.float(), mask, coords, gt.float()
