"""Seeded synthetic camera + depth + SfM scene for make_points_golden.py (no reference code).
The generated inputs are STORED in the fixture (depth, mask, SfM points, P, K, c2w): torch's
randn and matmul are not bit-reproducible across CPU models, the tests must not depend on that.

SfM points are placed so that they reproject to (integer pixel + U(-0.4, 0.4)): the rounded
pixel does not depend on the last bits of the projection arithmetic, so integer outputs of
the reference (CPU torch) and of the HIP kernels can be compared bit for bit."""
import math

import torch


def scene_rgb(H, W):
    y, x, c = torch.meshgrid(torch.arange(H), torch.arange(W), torch.arange(3), indexing="ij")
    return ((x * 7 + y * 13 + c * 29) % 256).float() / 255.0


def camera_scene(H, W, M, seed, frac_outside=0.1, outlier_frac=0.2, noise=0.02):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    t = torch.clamp((xx + 0.5 * yy) / 1.5, 0, 1)
    depth = (2.0 + 6.0 * (t * t * (3 - 2 * t)) + 0.05 * torch.randn(H, W, generator=g)).float()
    # a few depth edges so that the gradient mask has something to reject
    depth[H // 3: H // 2, W // 4: W // 2] += 1.5
    mask = torch.rand(H, W, generator=g) > 0.05
    # colours by exact integer arithmetic (no RNG, identical on every CPU): tests rebuild them
    # with scene_rgb(H, W) instead of storing 3 floats per pixel
    rgb = scene_rgb(H, W)
    f = 0.9 * W
    K = torch.tensor([[f, 0.0, W / 2 + 0.3], [0.0, 1.05 * f, H / 2 - 0.2], [0.0, 0.0, 1.0]])
    ax, ay, az = (0.2 * torch.rand(3, generator=g) - 0.1).tolist()
    Rx = torch.tensor([[1, 0, 0], [0, math.cos(ax), -math.sin(ax)], [0, math.sin(ax), math.cos(ax)]])
    Ry = torch.tensor([[math.cos(ay), 0, math.sin(ay)], [0, 1, 0], [-math.sin(ay), 0, math.cos(ay)]])
    Rz = torch.tensor([[math.cos(az), -math.sin(az), 0], [math.sin(az), math.cos(az), 0], [0, 0, 1]])
    c2w = torch.eye(4)
    c2w[:3, :3] = (Rz @ Ry @ Rx).float()
    c2w[:3, 3] = torch.tensor([0.3, -0.2, 0.5]) + 0.1 * torch.randn(3, generator=g)
    xs = torch.randint(0, W, (M,), generator=g)
    ys = torch.randint(0, H, (M,), generator=g)
    gt = 1.7 * depth[ys, xs] + 0.4 + noise * torch.randn(M, generator=g)
    outl = torch.rand(M, generator=g) < outlier_frac
    gt = torch.where(outl, gt * (0.3 + 2.7 * torch.rand(M, generator=g)), gt)
    u = xs.float() + (0.8 * torch.rand(M, generator=g) - 0.4)
    v = ys.float() + (0.8 * torch.rand(M, generator=g) - 0.4)
    outside = torch.rand(M, generator=g) < frac_outside
    kind = torch.randint(0, 3, (M,), generator=g)
    u = torch.where(outside & (kind == 0), u + W + 3.0, u)          # right of the image
    v = torch.where(outside & (kind == 1), -v - 3.0, v)             # above the image
    gt = torch.where(outside & (kind == 2), -gt, gt)                # behind the camera
    cam = torch.linalg.inv(K) @ torch.stack([u * gt, v * gt, gt])
    world = (c2w[:3, :3] @ cam + c2w[:3, 3:4]).T.contiguous()
    R = c2w[:3, :3].T
    C = c2w[:3, 3]
    P = K @ R @ torch.hstack([torch.eye(3), -C[:, None]])
    return {"depth": depth, "mask": mask, "rgb": rgb, "K": K, "c2w": c2w, "sfm": world.float(),
            "P": P.float()}
