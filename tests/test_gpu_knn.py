"""GPU parity of the kNN distances (A9/F3) with the reference's sklearn-based
`knn` (golden vector from the reference module) and the brute-force oracle."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import init_oracle as IO

pytestmark = pytest.mark.gpu
G = np.load(Path(__file__).resolve().parent / "golden" / "init_golden.npz")


def test_knn_matches_reference_golden():
    K = importlib.import_module("3dgs_monocular_depth_init_amd.knn")
    pts = torch.from_numpy(G["knn_pts"])
    d = K.knn(pts.cuda(), 4).cpu()
    assert torch.allclose(d, torch.from_numpy(G["knn_d4"]), rtol=1e-5, atol=1e-6)
    assert (d[:, 0] == 0).all()


@pytest.mark.parametrize("kind", ["uniform", "surface", "clustered"])
def test_grid_knn_is_exact(kind):
    K = importlib.import_module("3dgs_monocular_depth_init_amd.knn")
    g = torch.Generator().manual_seed(3)
    N = 20000
    if kind == "uniform":
        pts = torch.rand(N, 3, generator=g)
    elif kind == "surface":                      # a thin sheet + far outliers
        pts = torch.rand(N, 3, generator=g)
        pts[:, 2] = 0.01 * torch.sin(6 * pts[:, 0]) + 1e-4 * torch.randn(N, generator=g)
        pts[:20] = torch.rand(20, 3, generator=g) * 50 + 10
    else:
        centers = torch.rand(30, 3, generator=g) * 10
        pts = centers[torch.randint(0, 30, (N,), generator=g)] + 0.01 * torch.randn(N, 3, generator=g)
    got = K.knn(pts.cuda(), 4).cpu()
    idx = torch.randperm(N, generator=g)[:600]
    ref = torch.cdist(pts[idx].double(), pts.double()).topk(4, largest=False).values.float()
    assert torch.allclose(got[idx], ref, rtol=1e-4, atol=1e-6)
    s = K.initial_log_scales(pts.cuda())
    assert s.shape == (N, 3) and torch.isfinite(s).all()
    d3 = IO.knn_dists(pts[idx[:50]], 4) if False else None   # (oracle knn is O(N^2); golden covers it)
