"""LOF outlier removal of the initial cloud (SURVEY.md F4 tail) against the output of the
reference's own `lof_outlier_removal` (scikit-learn's LocalOutlierFactor; tests/golden/make_lof_golden.py).
CPU: the oracle restatement (oracle/lof_oracle.py) reproduces the recorded masks and scores.
GPU: `knn.knn_neighbors` (exact neighbours with indices), `knn.local_outlier_factor` and the
`postprocess_point_cloud` hook. scikit-learn keeps float32 inputs in float32 for its scores; the
kernels and the oracle work in float64, so scores are compared at 2e-5 and the masks exactly (the
recorded clouds have no score closer than 8e-5 to the threshold)."""
import importlib
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import lof_oracle

G = np.load(Path(__file__).resolve().parent / "golden" / "lof_golden.npz")
P_ = "3dgs_monocular_depth_init_amd."


def _case(i):
    pts = G[f"c{i}_pts"]
    n = pts.shape[0]
    return pts, int(G[f"c{i}_k"]), np.unpackbits(G[f"c{i}_outlier"])[:n].astype(bool), G[f"c{i}_nof"]


@pytest.mark.parametrize("i", [1, 2, 3])          # (the 30 000- and 12 000-point clouds are the GPU's)
def test_oracle_matches_reference(i):
    pts, k, ref_mask, ref_nof = _case(i)
    mask, nof = lof_oracle.lof(pts, k)
    assert np.array_equal(mask, ref_mask)
    assert np.allclose(nof, ref_nof, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("i", range(int(G["n"])))
def test_gpu_lof_matches_reference(i):
    K = importlib.import_module(P_ + "knn")
    pts, k, ref_mask, ref_nof = _case(i)
    mask, nof = K.local_outlier_factor(torch.from_numpy(pts).cuda(), k)
    assert mask.dtype == torch.bool and np.array_equal(mask.cpu().numpy(), ref_mask)
    assert np.allclose(nof.cpu().numpy(), ref_nof, rtol=2e-5, atol=2e-5)


@pytest.mark.gpu
def test_gpu_neighbours_exact_vs_brute_force():
    """Distances and indices of the grid search (grid passes, coarser retries, the all-in-one-cell
    finish) against a float64 brute force; K = 40 on a cloud with dense and sparse parts."""
    K = importlib.import_module(P_ + "knn")
    pts, _, _, _ = _case(0)
    pts = pts[:20000]
    dist, idx = K.knn_neighbors(torch.from_numpy(pts).cuda(), 40)
    x = torch.from_numpy(pts).double().cuda()
    d = torch.cdist(x, x)
    d.fill_diagonal_(float("inf"))
    ref = torch.topk(d, 40, dim=1, largest=False).values
    assert torch.allclose(dist, ref, rtol=1e-6, atol=1e-9)     # (cdist goes through a matrix product)
    assert bool((idx >= 0).all()) and bool((idx != torch.arange(len(pts), device="cuda")[:, None]).all())
    back = (x[idx.long()] - x[:, None, :]).norm(dim=-1)
    assert torch.allclose(back, dist, rtol=1e-12, atol=1e-12)  # the indices are the points at those distances
    assert bool((dist[:, 1:] >= dist[:, :-1]).all())


@pytest.mark.gpu
def test_gpu_postprocess_point_cloud_with_lof():
    PP = importlib.import_module(P_ + "point_cloud_postprocess.postprocess")
    PC = importlib.import_module(P_ + "point_cloud_postprocess.config")
    pts, k, ref_mask, _ = _case(1)
    cfg = PC.PointCloudPostprocessConfig(outlier_removal=PC.OutlierRemovalMethod.lof, lof_num_neighbors=k)
    rgbs = torch.rand(len(pts), 3)
    p, c = PP.postprocess_point_cloud(torch.from_numpy(pts), rgbs, [], [], np.zeros((0, 2)), cfg, "cuda")
    assert torch.equal(p.cpu(), torch.from_numpy(pts)[~torch.from_numpy(ref_mask)])
    assert torch.equal(c.cpu(), rgbs[~torch.from_numpy(ref_mask)])


@pytest.mark.gpu
def test_gpu_lof_200k_against_scikit_learn_live():
    """At the size of a real initial cloud: scikit-learn (the library the reference calls) run on
    the test box's host cores against the kernels, same 200 000 points."""
    sk = pytest.importorskip("sklearn.neighbors")
    K = importlib.import_module(P_ + "knn")
    g = torch.Generator().manual_seed(11)
    pts = torch.cat([torch.randn(150000, 3, generator=g) * torch.tensor([1.0, 0.6, 0.05]),
                     torch.randn(40000, 3, generator=g) * 0.08 + torch.tensor([0.5, 0.2, 0.4]),
                     (torch.rand(10000, 3, generator=g) - 0.5) * 8.0]).float()
    clf = sk.LocalOutlierFactor(n_neighbors=40, n_jobs=-1)
    ref = clf.fit_predict(pts.numpy()) == -1
    mask, nof = K.local_outlier_factor(pts.cuda(), 40)
    ref_nof = clf.negative_outlier_factor_.astype(np.float64)
    assert np.allclose(nof.cpu().numpy(), ref_nof, rtol=2e-5, atol=2e-5)
    # scores within float32 rounding of the threshold may fall on either side
    sure = np.abs(ref_nof + 1.5) > 1e-5
    assert np.array_equal(mask.cpu().numpy()[sure], ref[sure]) and int((~sure).sum()) < 20
