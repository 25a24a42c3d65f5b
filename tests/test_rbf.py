"""interp.method = "rbf" (reference interp.py:30-72).

CPU: oracle/rbf_oracle.py (numpy restatement of scipy's published RBFInterpolator algorithm, which the
reference's torchrbf dependency ports) against scipy.interpolate.RBFInterpolator itself. torchrbf is absent
here: parity against it is UNPINNED, and says so in the oracle's header.
GPU: csrc/rbf.hip (dense float64 LU + grid evaluation + bilinear upsampling) against the oracle."""
import importlib
from types import SimpleNamespace

import numpy as np
import pytest
import torch

from oracle import rbf_oracle as O


def _problem(P, W, H, seed):
    rng = np.random.default_rng(seed)
    xy = np.stack([rng.integers(0, W, P), rng.integers(0, H, P)])               # integer pixels, may repeat
    vals = 1.0 + 0.3 * np.sin(xy[0] / W * 5.0) * np.cos(xy[1] / H * 3.0) + 0.02 * rng.standard_normal(P)
    return xy, vals.astype(np.float32)


@pytest.mark.parametrize("kernel", ["thin_plate_spline", "linear", "cubic", "quintic"])
def test_oracle_equals_scipy_rbf_interpolator(kernel):
    from scipy.interpolate import RBFInterpolator
    rng = np.random.default_rng(0)
    y, d, x = rng.random((500, 2)), rng.random(500) * 3 + 1, rng.random((2000, 2))
    c, sh, sc = O.fit(y, d, 0.001, kernel)
    ref = RBFInterpolator(y, d, smoothing=0.001, kernel=kernel)(x)
    assert np.abs(O.evaluate(x, y, c, sh, sc, kernel) - ref).max() <= 1e-9 * np.abs(ref).max()


def test_oracle_bilinear_equals_torch_align_corners():
    a = np.random.default_rng(1).random((9, 6))
    t = torch.nn.functional.interpolate(torch.from_numpy(a)[None, None], size=(31, 17), mode="bilinear",
                                        align_corners=True)[0, 0].numpy()
    assert np.abs(O.bilinear_align_corners(a, 31, 17) - t).max() <= 1e-12


@pytest.mark.gpu
@pytest.mark.parametrize("kernel,P,W,H", [("thin_plate_spline", 700, 640, 360), ("thin_plate_spline", 3, 100, 80),
                                          ("linear", 300, 300, 200), ("cubic", 257, 512, 384),
                                          ("quintic", 400, 480, 320)])
def test_hip_rbf_interpolation_equals_oracle(kernel, P, W, H):
    I = importlib.import_module("3dgs_monocular_depth_init_amd.depth_alignment.alignment.interp")
    xy, vals = _problem(P, W, H, P)
    if P == 3:
        xy = np.array([[3, 90, 40], [5, 10, 70]])                                  # not collinear
    cfg = SimpleNamespace(method="rbf", kernel=kernel, smoothing=0.001)
    got = I.interpolate_scale(torch.from_numpy(xy).cuda(), torch.from_numpy(vals).cuda(), cfg, "cuda", W, H)
    ref = O.rbf_interpolation(xy, vals, W, H, 0.001, kernel)
    assert got.shape == (H, W) and got.dtype == torch.float32
    err = np.abs(got.cpu().numpy().astype(np.float64) - ref).max()
    assert err <= 2e-5 * np.abs(ref).max(), err        # float32 output of a float64 solve (measured ~1e-6)


@pytest.mark.gpu
def test_hip_rbf_full_size_system_and_point_subset():
    """max_rbf_points = 5000 sites (the config's cap) at 1080p: the 5003-unknown system is solved and the
    map reproduces the data at the sites to within the smoothing; align_depth_interpolate picks the subset."""
    I = importlib.import_module("3dgs_monocular_depth_init_amd.depth_alignment.alignment.interp")
    W, H, P = 1920, 1080, 5000
    rng = np.random.default_rng(7)
    lin = rng.choice(W * H, P, replace=False)                                       # distinct pixels
    xy = np.stack([lin % W, lin // W])
    vals = (1.0 + 0.3 * np.sin(xy[0] / W * 5.0) * np.cos(xy[1] / H * 3.0)).astype(np.float32)
    cfg = SimpleNamespace(method="rbf", kernel="thin_plate_spline", smoothing=0.001)
    got = I.interpolate_scale(torch.from_numpy(xy).cuda(), torch.from_numpy(vals).cuda(), cfg, "cuda", W, H)
    assert torch.isfinite(got).all()
    at = got[torch.from_numpy(xy[1]).cuda(), torch.from_numpy(xy[0]).cuda()].cpu().numpy()
    assert np.abs(at - vals).max() < 0.05 and np.abs(at - vals).mean() < 5e-3      # smooth field, grid 256 wide


@pytest.mark.gpu
def test_kernels_that_need_epsilon_take_the_median_fallback():
    """torchrbf (like scipy) refuses the shape-parameter kernels without `epsilon`, and the reference's call never
    passes one (interp.py:44-50): its `except Exception` then uses the median scale (interp.py:345-359). Same here."""
    I = importlib.import_module("3dgs_monocular_depth_init_amd.depth_alignment.alignment.interp")
    cfg = SimpleNamespace(method="rbf", kernel="gaussian", smoothing=0.001)
    xy, vals = _problem(50, 64, 48, 3)
    with pytest.raises(ValueError, match="epsilon"):
        I.interpolate_scale(torch.from_numpy(xy).cuda(), torch.from_numpy(vals).cuda(), cfg, "cuda", 64, 48)
