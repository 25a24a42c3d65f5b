"""Generate tests/golden/init_golden.npz by running the REFERENCE's own modules.

Run only in the build container (needs /root/reference, which never travels
to the GPU box): PYTHONPATH=/root/reference python tests/golden/make_init_golden.py
Only reference modules that import cleanly here are used:
lstsqrs, static_subsampler, adaptive_subsampling, num_sfm_points_mask,
runner_utils (knn, rgb_to_sh). The output is data only: seeded inputs and the
reference's outputs for them.
"""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, "/root/reference")
from gs_init_compare.depth_alignment.alignment.lstsqrs import (  # noqa: E402
    DepthAlignmentLstSqrs, align_depth_least_squares)
from gs_init_compare.depth_prediction.predictors.depth_predictor_interface import PredictedDepth  # noqa: E402
from gs_init_compare.depth_subsampling.adaptive_subsampling import (  # noqa: E402
    AdaptiveDepthSubsampler, get_depth_multipler_map, iqr_outlier_bounds)
from gs_init_compare.depth_subsampling.config import AdaptiveSubsamplingConfig, NumSfMPointsMaskConfig  # noqa: E402
from gs_init_compare.depth_subsampling.num_sfm_points_mask import calculate_patch_sizes, num_sfm_points_mask  # noqa: E402
from gs_init_compare.depth_subsampling.static_subsampler import StaticDepthSubsampler  # noqa: E402
from gs_init_compare.utils.runner_utils import knn, rgb_to_sh  # noqa: E402

out = {}


def depth_scene(H, W, seed, n_sfm):
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    t = torch.clamp((xx + 0.5 * yy) / 1.5, 0, 1)
    depth = 2.0 + 6.0 * (t * t * (3 - 2 * t)) + 0.05 * torch.randn(H, W, generator=g)
    mask = torch.rand(H, W, generator=g) > 0.05
    xs = torch.randint(0, W, (n_sfm,), generator=g)
    ys = torch.randint(0, H, (n_sfm,), generator=g)
    coords = torch.stack([xs, ys]).long()
    gt = 1.7 * depth[ys, xs] + 0.4 + 0.02 * torch.randn(n_sfm, generator=g)
    outl = torch.rand(n_sfm, generator=g) < 0.2
    gt = torch.where(outl, gt * (0.3 + 2.7 * torch.rand(n_sfm, generator=g)), gt)
    return depth.float(), mask, coords, gt.float()


# --- B2 least squares ---------------------------------------------------------
for i, (H, W, n) in enumerate([(64, 96, 200), (270, 480, 1500), (48, 40, 9)]):
    depth, mask, coords, gt = depth_scene(H, W, 10 + i, n)
    res = DepthAlignmentLstSqrs.align(PredictedDepth(depth=depth.clone(), mask=mask), coords, gt)
    d = torch.vstack([depth[coords[1], coords[0]].flatten(), torch.ones(n)])
    s, t = align_depth_least_squares(d, gt)
    out[f"lsq{i}_depth"] = depth.numpy(); out[f"lsq{i}_mask"] = mask.numpy()
    out[f"lsq{i}_coords"] = coords.numpy(); out[f"lsq{i}_gt"] = gt.numpy()
    out[f"lsq{i}_scale_shift"] = np.array([float(s), float(t)], np.float64)
    out[f"lsq{i}_aligned"] = res.aligned_depth.numpy()

# --- B5 / B6 / B7 masks ---------------------------------------------------------
for i, (H, W, n) in enumerate([(64, 96, 300), (270, 480, 4000)]):
    depth, mask, coords, gt = depth_scene(H, W, 20 + i, n)
    rgb = torch.rand(H, W, 3, generator=torch.Generator().manual_seed(5))
    out[f"msk{i}_depth"] = depth.numpy(); out[f"msk{i}_mask"] = mask.numpy()
    out[f"msk{i}_coords"] = coords.numpy()
    for k in (3, 10):
        m = StaticDepthSubsampler(k).get_mask(rgb, depth, mask)
        out[f"msk{i}_static{k}"] = np.packbits(m.numpy())
    m = AdaptiveDepthSubsampler(AdaptiveSubsamplingConfig()).get_mask(rgb, depth.clone(), mask)
    out[f"msk{i}_adaptive"] = np.packbits(m.numpy())
    lo, hi = iqr_outlier_bounds(depth[mask])
    out[f"msk{i}_iqr"] = np.array([float(lo), float(hi)], np.float64)
    out[f"msk{i}_multiplier"] = get_depth_multipler_map(depth.clone(), mask).numpy()
    cfg = NumSfMPointsMaskConfig(num_patches_small_axis=8 if i == 0 else 20, threshold=3 if i == 0 else 15)
    m = num_sfm_points_mask(coords, (H, W), cfg)
    out[f"msk{i}_sfmmask"] = np.packbits(m.numpy().reshape(-1))
    out[f"msk{i}_sfmcfg"] = np.array([cfg.num_patches_small_axis, cfg.threshold])
for j, shape in enumerate([(1080, 1920), (1920, 1080), (270, 480), (64, 96), (100, 100)]):
    ps, pg = calculate_patch_sizes(shape, 20)
    out[f"patch{j}"] = np.array([*shape, *ps, *pg])

# --- A9 helpers ----------------------------------------------------------------
pts = torch.rand(500, 3, generator=torch.Generator().manual_seed(7))
out["knn_pts"] = pts.numpy()
out["knn_d4"] = knn(pts, 4).numpy()
rgb = torch.rand(32, 3, generator=torch.Generator().manual_seed(8))
out["sh_rgb"] = rgb.numpy()
out["sh_out"] = rgb_to_sh(rgb).numpy()

path = Path(__file__).resolve().parent / "init_golden.npz"
np.savez_compressed(path, **out)
print(path, path.stat().st_size, "bytes", len(out), "arrays")
