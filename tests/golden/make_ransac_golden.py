"""Generate tests/golden/ransac_golden.npz by running the REFERENCE's own RANSAC/MSAC
(gs_init_compare/depth_alignment/alignment/ransacs.py:100-189).

Run only in the build container: python tests/golden/make_ransac_golden.py

ransacs.py imports gs_init_compare.config, whose module level does
`from gsplat.strategy import DefaultStrategy, MCMCStrategy` (config.py:5, used only
as dataclass field types/defaults) and, through point_cloud_postprocess/config.py:5,
`from pointcloud_subsampling import PointCloudSubsamplingParams` (the un-built native
module). Neither package exists here. As SURVEY.md section 8c proposes, this script --
and only this script, which never travels or ships -- registers INERT placeholders for
those two names (empty dataclasses) so that the reference's pure-torch RANSAC can be
imported and run for real. No gsplat functionality is emulated and nothing the
placeholders return takes part in the recorded numbers. The output is data only:
seeded inputs, the iteration/inlier counts the reference prints, its scale/shift and
its aligned depth map.
"""
import contextlib
import io
import re
import sys
import types
from dataclasses import dataclass
from pathlib import Path

import numpy as np
import torch


def _placeholder(name, **attrs):
    m = types.ModuleType(name)
    m.__dict__.update(attrs)
    sys.modules[name] = m
    return m


@dataclass
class DefaultStrategy:      # placeholder: type name only (config.py:141-143)
    verbose: bool = False


@dataclass
class MCMCStrategy:         # placeholder
    verbose: bool = False


@dataclass
class PointCloudSubsamplingParams:   # placeholder (point_cloud_postprocess/config.py:5)
    pass


g = _placeholder("gsplat")
g.__path__ = []
g.strategy = _placeholder("gsplat.strategy", DefaultStrategy=DefaultStrategy, MCMCStrategy=MCMCStrategy)
ps = _placeholder("pointcloud_subsampling")
ps.__path__ = []
ps.subsampling_params = _placeholder("pointcloud_subsampling.subsampling_params",
                                     PointCloudSubsamplingParams=PointCloudSubsamplingParams)

sys.path.insert(0, "/root/reference")
from gs_init_compare.depth_alignment.alignment import ransacs  # noqa: E402
from gs_init_compare.depth_alignment.config import RansacConfig  # noqa: E402
from gs_init_compare.depth_prediction.predictors.depth_predictor_interface import PredictedDepth  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_init_golden_scene import depth_scene  # noqa: E402

out = {}
LINE = re.compile(r"Iterations: (\d+), Inliers: (\d+)/(\d+), best scale: ([-\d.e+]+), best shift: ([-\d.e+]+)")
cases = [
    # (H, W, n_sfm, outlier_frac, noise, seed, rng_seed, loss, cfg overrides)
    (64, 96, 300, 0.2, 0.02, 30, 42, "ransac", {}),
    (64, 96, 300, 0.2, 0.02, 30, 42, "msac", {}),
    (270, 480, 2000, 0.3, 0.02, 31, 43, "ransac", {}),
    (270, 480, 2000, 0.3, 0.02, 31, 43, "msac", {}),
    (64, 96, 400, 0.1, 0.002, 32, 44, "ransac", {"inlier_threshold": 0.05}),   # high inlier ratio: adaptive stop after a few draws
    (64, 96, 400, 0.1, 0.002, 32, 44, "msac", {"inlier_threshold": 0.05}),
    (64, 96, 150, 0.5, 0.05, 33, 45, "ransac", {"max_iters": 60}),             # runs into max_iters
    (48, 40, 12, 0.0, 0.0, 34, 46, "msac", {"min_iters": 5}),                  # tiny, exact model
]
for i, (H, W, n, frac, noise, seed, rng_seed, loss, over) in enumerate(cases):
    depth, mask, coords, gt = depth_scene(H, W, seed, n, outlier_frac=frac, noise=noise)
    cfg = RansacConfig(**over)
    torch.manual_seed(rng_seed)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        res = ransacs._align_depth_ransac_generic(
            PredictedDepth(depth=depth.clone(), mask=mask), coords, gt,
            ransacs._ransac_loss if loss == "ransac" else ransacs._msac_loss, cfg)
    m = LINE.search(buf.getvalue())
    assert m, buf.getvalue()
    out[f"r{i}_depth"] = depth.numpy(); out[f"r{i}_mask"] = mask.numpy()
    out[f"r{i}_coords"] = coords.numpy(); out[f"r{i}_gt"] = gt.numpy()
    out[f"r{i}_rng_seed"] = np.int64(rng_seed)
    out[f"r{i}_loss"] = np.array(loss)
    out[f"r{i}_cfg"] = np.array([cfg.inlier_threshold, cfg.max_iters, cfg.confidence, cfg.sample_size, cfg.min_iters], np.float64)
    out[f"r{i}_iterations"] = np.int64(m.group(1))
    out[f"r{i}_inliers"] = np.int64(m.group(2))
    out[f"r{i}_scale_shift"] = np.array([float(m.group(4)), float(m.group(5))], np.float64)
    out[f"r{i}_aligned"] = res.aligned_depth.numpy()
    print(i, loss, "iterations", m.group(1), "inliers", m.group(2), "/", n, "scale", m.group(4), "shift", m.group(5))
out["n_cases"] = np.int64(len(cases))
np.savez_compressed(Path(__file__).resolve().parent / "ransac_golden.npz", **out)
print("wrote ransac_golden.npz")
