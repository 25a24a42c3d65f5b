"""Properties of the point-cloud subsampler oracle (oracle/pointcloud_oracle.py, parity unpinned:
the reference's C++/Eigen module cannot be built here) -- the invariants the algorithm of
native_modules/subsampling/src/impl.cpp:313-426 guarantees."""
import numpy as np

from oracle import pointcloud_oracle as PO


def _cloud(N=2500, seed=0):
    g = np.random.default_rng(seed)
    pts = (g.random((N, 3)) * [4, 3, 2] + [-2, -1.5, 3]).astype(np.float32)
    pts[: N // 3] = pts[N // 3: 2 * (N // 3)] + g.normal(0, 0.002, (N // 3, 3)).astype(np.float32)   # tight pairs
    rgb = g.random((N, 3)).astype(np.float32)
    K = np.array([[500, 0, 320], [0, 480, 240], [0, 0, 1]], np.float32)
    Ps = [(K @ np.hstack([np.eye(3), np.array([[dx], [0.0], [0.0]])])).astype(np.float32) for dx in (0.0, 0.5)]
    return pts, rgb, [K, K], Ps, [(640, 480), (640, 480)]


def test_extents_formula_and_visibility():
    pts, _, Ks, Ps, sizes = _cloud()
    ext = PO.min_gaussian_extents(pts, Ks, Ps, sizes)
    seen = ext > 0
    assert seen.any() and (~seen).any() and np.all(ext[~seen] == -1.0)
    # one camera at the origin looking down +z: extent = 2 z / min(fx, fy) where only it sees the point
    i = int(np.argmax(seen))
    assert ext[i] <= 2.0 * pts[i, 2] / 480.0 * (1 + 1e-6)


def test_merging_preserves_mass_and_is_monotone_in_the_multiplier():
    pts, rgb, Ks, Ps, sizes = _cloud()
    ext = PO.min_gaussian_extents(pts, Ks, Ps, sizes)
    p0, c0 = PO.subsample(pts, rgb, ext, 1.1, 0.0)
    assert len(p0) == len(pts)                              # multiplier 0: nothing may merge
    assert np.allclose(np.sort(p0, 0), np.sort(pts, 0))
    n_prev = len(pts)
    for mult in (1.0, 4.0, 16.0):
        p, c = PO.subsample(pts, rgb, ext, 1.1, mult)
        assert len(p) <= n_prev and np.isfinite(p).all() and np.isfinite(c).all()
        assert p.min(0).min() >= pts.min() - 1e-5 and p.max() <= pts.max() + 1e-5    # means stay inside
        assert c.min() >= 0 and c.max() <= 1
        n_prev = len(p)
    assert n_prev < len(pts)
