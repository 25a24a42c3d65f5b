"""TEST INFRASTRUCTURE (oracle): CPU restatement of the reference's `rbf_interpolation`
(/root/reference/gs_init_compare/depth_alignment/alignment/interp.py:30-72). Only tests/, smoke() and
bench.py's cpu_baseline may import this.

The reference builds a `torchrbf.RBFInterpolator` (third-party, NOT vendored under /root/reference and not
installable here; ArmanMaesumi/torchrbf, a PyTorch port of `scipy.interpolate.RBFInterpolator` without the
`neighbors` option) over the SfM pixels normalised to [0,1]^2, evaluates it on a grid 256 pixels wide and
upsamples bilinearly (align_corners) to the image. This file restates the published algorithm of
scipy/interpolate/_rbfinterp.py (`_build_and_solve_system`, `_build_evaluation_coefficients`), which torchrbf
ports line by line:

    shift = (max + min) / 2, scale = (max - min) / 2 over the data sites (scale 1 where it is 0)
    lhs = [[ K(y eps, y eps) + smoothing I , P((y - shift)/scale) ], [ P^T , 0 ]],  rhs = [ d ; 0 ]
    f(x) = K(x eps, y eps) @ coeffs[:P] + P((x - shift)/scale) @ coeffs[P:]

with the kernel functions (as scipy's `_rbfinterp_pythran.py`) linear -r, thin_plate_spline r^2 log r
(0 at r = 0), cubic r^3, quintic -r^5; degree = the kernel's minimum degree (0, 1, 1, 2); epsilon = 1
(these are the scale-invariant kernels, the only ones usable without `epsilon`, which the reference's call
never passes); monomials in scipy's order [1, x, y, x^2, x y, y^2] (`_monomial_powers`).

PARITY: unpinned against torchrbf (absent). Pinned against scipy.interpolate.RBFInterpolator itself -- the
library torchrbf ports -- in tests/test_rbf.py (float64). The reference runs torchrbf in float32; this
restatement and the HIP path solve in float64.
"""
from __future__ import annotations

import numpy as np

KERNELS = {"linear": 0, "thin_plate_spline": 1, "cubic": 2, "quintic": 3}
MIN_DEGREE = {"linear": 0, "thin_plate_spline": 1, "cubic": 1, "quintic": 2}


def _phi(r: np.ndarray, kernel: str) -> np.ndarray:
    if kernel == "linear":
        return -r
    if kernel == "cubic":
        return r ** 3
    if kernel == "quintic":
        return -r ** 5
    if kernel == "thin_plate_spline":
        with np.errstate(divide="ignore", invalid="ignore"):
            return np.where(r == 0.0, 0.0, r * r * np.log(r))
    raise ValueError(f"kernel {kernel!r}: built are {sorted(KERNELS)}")


def _poly(x: np.ndarray, degree: int) -> np.ndarray:
    cols = [np.ones(len(x))]
    if degree >= 1:
        cols += [x[:, 0], x[:, 1]]
    if degree >= 2:
        cols += [x[:, 0] ** 2, x[:, 0] * x[:, 1], x[:, 1] ** 2]
    return np.stack(cols, 1)


def fit(y: np.ndarray, d: np.ndarray, smoothing: float, kernel: str):
    """y [P,2], d [P] -> (coeffs [P+R], shift [2], scale [2]) as scipy's _build_and_solve_system."""
    y = np.asarray(y, np.float64)
    d = np.asarray(d, np.float64)
    P = len(y)
    mins, maxs = y.min(0), y.max(0)
    shift, scale = (maxs + mins) / 2, (maxs - mins) / 2
    scale[scale == 0.0] = 1.0
    Pm = _poly((y - shift) / scale, MIN_DEGREE[kernel])
    R = Pm.shape[1]
    r = np.linalg.norm(y[:, None, :] - y[None, :, :], axis=-1)
    lhs = np.zeros((P + R, P + R))
    lhs[:P, :P] = _phi(r, kernel) + smoothing * np.eye(P)
    lhs[:P, P:] = Pm
    lhs[P:, :P] = Pm.T
    rhs = np.concatenate([d, np.zeros(R)])
    return np.linalg.solve(lhs, rhs), shift, scale


def evaluate(x: np.ndarray, y: np.ndarray, coeffs: np.ndarray, shift, scale, kernel: str) -> np.ndarray:
    x = np.asarray(x, np.float64)
    P = len(y)
    r = np.linalg.norm(x[:, None, :] - np.asarray(y, np.float64)[None, :, :], axis=-1)
    return _phi(r, kernel) @ coeffs[:P] + _poly((x - shift) / scale, MIN_DEGREE[kernel]) @ coeffs[P:]


def bilinear_align_corners(src: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """torch.nn.functional.interpolate(mode="bilinear", align_corners=True) on a 2-D array."""
    h, w = src.shape
    ys = np.arange(out_h) * ((h - 1) / (out_h - 1) if out_h > 1 else 0.0)
    xs = np.arange(out_w) * ((w - 1) / (out_w - 1) if out_w > 1 else 0.0)
    y0, x0 = np.minimum(ys.astype(int), h - 1), np.minimum(xs.astype(int), w - 1)
    y1, x1 = np.minimum(y0 + 1, h - 1), np.minimum(x0 + 1, w - 1)
    fy, fx = (ys - y0)[:, None], (xs - x0)[None, :]
    top = src[y0][:, x0] * (1 - fx) + src[y0][:, x1] * fx
    bot = src[y1][:, x0] * (1 - fx) + src[y1][:, x1] * fx
    return top * (1 - fy) + bot * fy


def rbf_interpolation(coords: np.ndarray, values: np.ndarray, W: int, H: int, smoothing: float = 0.001,
                      kernel: str = "thin_plate_spline") -> np.ndarray:
    """interp.py:30-72. coords [2,P] pixel (x, y), values [P] -> [H,W]."""
    y = np.stack([coords[0].astype(np.float32) / np.float32(W - 1.0), coords[1].astype(np.float32) / np.float32(H - 1.0)], 1)
    coeffs, shift, scale = fit(y, values, smoothing, kernel)
    factor = max(W / 256, 1)
    qw, qh = int(W / factor), int(H / factor)
    gx, gy = np.linspace(0, 1, qw, dtype=np.float32), np.linspace(0, 1, qh, dtype=np.float32)
    grid = np.stack(np.meshgrid(gx, gy, indexing="ij"), -1).reshape(-1, 2)          # x-major, as the reference
    f = evaluate(grid, y, coeffs, shift, scale, kernel).reshape(qw, qh)
    return bilinear_align_corners(f, W, H).T                                          # [W,H] -> [H,W]
