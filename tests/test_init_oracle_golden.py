"""Pin oracle/init_oracle.py against golden vectors produced by the reference's
own modules (tests/golden/make_init_golden.py; runs without /root/reference)."""
from pathlib import Path

import numpy as np
import pytest
import torch

from oracle import init_oracle as IO

G = np.load(Path(__file__).resolve().parent / "golden" / "init_golden.npz")


def _t(name):
    return torch.from_numpy(G[name])


def _bits(name, n):
    return torch.from_numpy(np.unpackbits(G[name])[:n].astype(bool))


@pytest.mark.parametrize("i", [0, 1, 2])
def test_lstsq(i):
    depth, coords, gt = _t(f"lsq{i}_depth"), _t(f"lsq{i}_coords"), _t(f"lsq{i}_gt")
    s, t, aligned = IO.lstsq_align(depth, coords, gt)
    ref = G[f"lsq{i}_scale_shift"]
    assert float(s) == pytest.approx(ref[0], rel=1e-6) and float(t) == pytest.approx(ref[1], rel=1e-6, abs=1e-7)
    assert torch.equal(aligned, _t(f"lsq{i}_aligned"))


@pytest.mark.parametrize("i", [0, 1])
def test_masks(i):
    depth, mask, coords = _t(f"msk{i}_depth"), _t(f"msk{i}_mask"), _t(f"msk{i}_coords")
    H, W = depth.shape
    for k in (3, 10):
        assert torch.equal(IO.static_mask((H, W), k, mask), _bits(f"msk{i}_static{k}", H * W))
    assert torch.equal(IO.adaptive_mask((H, W, 3), depth.clone(), mask), _bits(f"msk{i}_adaptive", H * W))
    lo, hi = IO.iqr_outlier_bounds(depth[mask])
    assert [float(lo), float(hi)] == pytest.approx(list(G[f"msk{i}_iqr"]), rel=1e-7)
    assert torch.equal(IO.get_depth_multiplier_map(depth.clone(), mask), _t(f"msk{i}_multiplier"))
    nps, thr = (int(v) for v in G[f"msk{i}_sfmcfg"])
    m = IO.num_sfm_points_mask(coords, (H, W), nps, thr)
    assert torch.equal(m.reshape(-1), _bits(f"msk{i}_sfmmask", H * W))
    assert m.any() and ((~m).any() or i == 1)


def test_patch_sizes():
    for j in range(5):
        h, w, ph, pw, gh, gw = (int(v) for v in G[f"patch{j}"])
        assert IO.calculate_patch_sizes((h, w), 20) == ((ph, pw), (gh, gw))
    assert IO.calculate_patch_sizes((1080, 1920), 20) == ((54, 53), (20, 36))


def test_knn_and_rgb_to_sh():
    assert torch.allclose(IO.knn_dists(_t("knn_pts"), 4), _t("knn_d4"), rtol=1e-5, atol=1e-6)
    assert torch.equal(IO.rgb_to_sh(_t("sh_rgb")), _t("sh_out"))


# ---- B3: the reference's own RANSAC / MSAC run (tests/golden/make_ransac_golden.py) ----
RG = np.load(Path(__file__).resolve().parent / "golden" / "ransac_golden.npz")


def ransac_case(i):
    t = lambda n: torch.from_numpy(RG[f"r{i}_{n}"])
    thr, max_iters, conf, ssz, min_iters = RG[f"r{i}_cfg"]
    cfg = dict(inlier_threshold=float(thr), max_iters=int(max_iters), confidence=float(conf),
               sample_size=int(ssz), min_iters=int(min_iters))
    return dict(depth=t("depth"), mask=t("mask"), coords=t("coords"), gt=t("gt"),
                seed=int(RG[f"r{i}_rng_seed"]), loss=str(RG[f"r{i}_loss"]), cfg=cfg,
                iterations=int(RG[f"r{i}_iterations"]), inliers=int(RG[f"r{i}_inliers"]),
                scale_shift=RG[f"r{i}_scale_shift"], aligned=t("aligned"))


@pytest.mark.parametrize("i", range(int(RG["n_cases"])))
def test_ransac_oracle_matches_reference_run(i):
    c = ransac_case(i)
    torch.manual_seed(c["seed"])
    s, t, aligned, it, inl = IO.ransac_align(c["depth"], c["coords"], c["gt"], c["loss"],
                                             IO.RansacConfig(**c["cfg"]))
    assert it == c["iterations"] and inl == c["inliers"]
    assert float(s) == pytest.approx(c["scale_shift"][0], rel=1e-7)
    assert float(t) == pytest.approx(c["scale_shift"][1], rel=1e-7)
    assert torch.equal(aligned, c["aligned"])
