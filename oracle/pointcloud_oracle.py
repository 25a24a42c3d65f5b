"""CPU restatement of the reference's native point-cloud subsampler -- TEST INFRASTRUCTURE ONLY
(imported by tests/ only; the product path is csrc/pointcloud.hip).

PARITY UNPINNED: the reference module is C++20 + Eigen 3.4 + pybind11 built through Conan /
scikit-build (native_modules/CMakeLists.txt, conanfile.py:20); Eigen is absent here, so it cannot
be compiled and the reference holds no fixtures for it. This file restates, in float32 numpy
with the reference's own evaluation order:
  compute_minimal_gaussian_extents   native_modules/subsampling/src/impl.cpp:17-35, 70-126
  subsample_pointcloud_impl          impl.cpp:313-426 (spatial_median split 217-228,
                                     BoundingBox geometry.h:27-86)
"""
import numpy as np

F = np.float32


def min_gaussian_extents(points, Ks, Ps, image_sizes):
    out = np.full(points.shape[0], np.finfo(F).max, F)
    for i, pt in enumerate(points.astype(F)):
        for K, P, (w, h) in zip(Ks, Ps, image_sizes):
            P = P.astype(F)
            proj = np.array([F(P[r, 0] * pt[0] + P[r, 1] * pt[1] + P[r, 2] * pt[2] + P[r, 3]) for r in range(3)], F)
            d = proj[2]
            if d <= 0:
                continue
            u, v = F(proj[0] / d), F(proj[1] / d)
            if u < 0 or u >= w or v < 0 or v >= h:
                continue
            f = min(F(K[0, 0]), F(K[1, 1]))
            out[i] = min(out[i], F(2.0) * F(d / f))
    out[out == np.finfo(F).max] = F(-1.0)
    return out


def subsample(points, rgbs, extents, max_aspect=1.1, min_mult=1.0):
    pos = points.astype(F)
    rgb = rgbs.astype(F)
    mn, mx = pos.min(0), pos.max(0)
    half = F((mx - mn).max() / F(2.0))
    centre = (mn + mx) / F(2.0)
    stack = [(np.arange(len(pos)), centre - half, centre + half, 0)]       # indices, box min, box max, prev axis X
    out_p, out_c = [], []
    while stack:
        idx, bmin, bmax, prev = stack.pop()
        if len(idx) == 0:
            continue
        if len(idx) == 1:
            out_p.append(pos[idx[0]]); out_c.append(rgb[idx[0]])
            continue
        avg = F(0)
        for ix in idx:
            avg = F(avg + extents[ix])
        avg = F(avg / F(len(idx)))
        tmin, tmax = pos[idx].min(0), pos[idx].max(0)
        od, td = bmax - bmin, tmax - tmin
        with np.errstate(divide="ignore", invalid="ignore"):
            orig_ar = F(od.max() / od.min())
            tight_ar = F(td.max() / td.min())
        ar = tight_ar if tight_ar < orig_ar else orig_ar               # std::min(orig, tight)
        if ar <= F(max_aspect) and td.max() <= F(F(min_mult) * avg):
            sp, sc = np.zeros(3, F), np.zeros(3, F)
            for ix in idx:
                sp = (sp + pos[ix]).astype(F); sc = (sc + rgb[ix]).astype(F)
            out_p.append(sp / F(len(idx))); out_c.append(sc / F(len(idx)))
            continue
        if len(idx) <= 2:
            for ix in idx:
                out_p.append(pos[ix]); out_c.append(rgb[ix])
            continue
        axis = (prev + 1) % 3
        split = F((bmin[axis] + bmax[axis]) / F(2.0))
        left = idx[pos[idx, axis] < split]
        right = idx[~(pos[idx, axis] < split)]
        lmax, rmin = bmax.copy(), bmin.copy()
        lmax[axis] = split
        rmin[axis] = split
        stack.append((left, bmin, lmax, axis))
        stack.append((right, rmin, bmax, axis))
    return np.array(out_p, F).reshape(-1, 3), np.array(out_c, F).reshape(-1, 3)
