"""Seeded synthetic scenes shared by tests, smoke() and bench.py
(definitions: SURVEY.md section 8d / BASELINE.md section 2)."""
from __future__ import annotations

import math

import torch

SH_C0 = 0.28209479177387814


def make_scene(N: int, seed: int = 0, box=(3.2, 1.8, 1.0), scale_mean=0.005, scale_std=0.3,
               sh_K: int = 16):
    """Scene S(N, seed): returns dict of CPU fp32 tensors (activated params)."""
    g = torch.Generator().manual_seed(seed)
    means = (torch.rand(N, 3, generator=g) * 2 - 1) * torch.tensor(box)
    scales = torch.exp(torch.randn(N, 3, generator=g) * scale_std + math.log(scale_mean))
    quats = torch.rand(N, 4, generator=g)
    u = torch.rand(N, generator=g) * 0.9 + 0.05
    opacities = u
    rgb = torch.rand(N, 3, generator=g)
    sh0 = ((rgb - 0.5) / SH_C0)[:, None, :]
    shN = torch.randn(N, sh_K - 1, 3, generator=g) * 0.05
    return dict(means=means, quats=quats, scales=scales, opacities=opacities, sh0=sh0, shN=shN)


def look_at_camera(i: int, n: int = 100, width=1920, height=1080, f=1200.0, dist=4.0,
                   rx=0.5, ry=0.3):
    """Cam(i): centre (rx sin t, ry cos t, -dist), looking at the origin, +y down."""
    t = 2 * math.pi * i / n
    eye = torch.tensor([rx * math.sin(t), ry * math.cos(t), -dist])
    fwd = -eye / eye.norm()
    up = torch.tensor([0.0, -1.0, 0.0])           # world up = -y  (camera +y is down)
    right = torch.linalg.cross(fwd, up)
    right = right / right.norm()
    down = torch.linalg.cross(fwd, right)
    R = torch.stack([right, down, fwd], dim=0)    # world->camera rows
    viewmat = torch.eye(4)
    viewmat[:3, :3] = R
    viewmat[:3, 3] = -R @ eye
    K = torch.tensor([[f, 0, width / 2], [0, f, height / 2], [0, 0, 1.0]])
    return viewmat, K


def cameras(ids, **kw):
    vms, Ks = zip(*(look_at_camera(i, **kw) for i in ids))
    return torch.stack(vms), torch.stack(Ks)


def config_c1(N=10_000, seed=0):
    """c1: 10k Gaussians in [-1,1]^2 x [-0.5,0.5], camera at z=-3, 256x256, f=300."""
    sc = make_scene(N, seed, box=(1.0, 1.0, 0.5), scale_mean=0.02)
    vm = torch.eye(4)[None].clone()
    vm[0, 2, 3] = 3.0
    K = torch.tensor([[[300.0, 0, 128], [0, 300.0, 128], [0, 0, 1]]])
    return sc, vm, K, 256, 256
