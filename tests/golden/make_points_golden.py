"""Generate tests/golden/points_golden.npz by running the REFERENCE's own
gs_init_compare/depth_prediction/points_from_depth.py and depth_alignment/pipeline.py:

  B1  project_and_filter_sfm_pts / get_valid_sfm_pts     points_from_depth.py:111-180
      (incl. the "< 1/4 of the points land in the image" error branch, :124-129)
  B4  DepthAlignmentPipeline.align, no-segmentation       pipeline.py:248-288
  B8  depth_gradient_mask                                 points_from_depth.py:192-212
  B9  get_pts_from_depth (masks, compaction, unprojection) points_from_depth.py:215-329

Run only in the build container: python tests/golden/make_points_golden.py

Both modules are pure torch, but their import chain names nine third-party packages
that do not exist here (gsplat, pointcloud_subsampling, pycolmap, cv2, imageio,
segment_anything, skimage, nerfbaselines, open3d) -- as type names, dataset loaders,
segmenters and debug exporters, none of which the recorded functions execute (the
parser is duck-typed: `.points`, `.point_indices[name]`; segmenter=None; no debug dir).
The seeded inputs (our own generator, make_points_golden_scene.py) are stored next to the
outputs -- depth, packed mask, SfM points, P, K, c2w; the colours are exact integer arithmetic
(scene_rgb) and are rebuilt by the tests.
As SURVEY.md section 8c proposes, this script -- and only this script, which never
travels or ships -- registers INERT placeholder modules for those nine names so that
the import succeeds. No functionality of those packages is emulated and nothing a
placeholder returns takes part in the recorded numbers. The output is data only.
"""
import contextlib
import importlib.abc
import importlib.machinery
import io
import sys
import types
from pathlib import Path

import numpy as np
import torch

ABSENT = ("gsplat", "pointcloud_subsampling", "pycolmap", "cv2", "imageio", "segment_anything",
          "skimage", "nerfbaselines", "open3d")


class _InertModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        cls = type(name, (object,), {"__init__": lambda self, *a, **k: None})
        setattr(self, name, cls)
        return cls


class _InertFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name.split(".")[0] in ABSENT:
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        m = _InertModule(spec.name)
        m.__path__ = []
        return m

    def exec_module(self, module):
        pass


sys.meta_path.insert(0, _InertFinder())
sys.path.insert(0, "/root/reference")
from gs_init_compare.config import Config  # noqa: E402
from gs_init_compare.depth_alignment.config import DepthAlignmentStrategyEnum  # noqa: E402
from gs_init_compare.depth_alignment.exceptions import LowDepthAlignmentConfidenceError  # noqa: E402
from gs_init_compare.depth_alignment.pipeline import DepthAlignmentPipeline  # noqa: E402
from gs_init_compare.depth_prediction import points_from_depth as pfd  # noqa: E402
from gs_init_compare.depth_prediction.predictors.depth_predictor_interface import PredictedDepth  # noqa: E402
from gs_init_compare.types import InputImage  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_points_golden_scene import camera_scene  # noqa: E402


class _Parser:
    """What get_pts_from_depth reads from the dataset parser (points_from_depth.py:233-237)."""

    def __init__(self, name, pts):
        self.points = pts.numpy()
        self.point_indices = {name: np.arange(pts.shape[0])}


SUB = (5, 7)


def _store(key, sc):
    out[f"{key}_depth"] = sc["depth"].numpy()
    out[f"{key}_mask"] = np.packbits(sc["mask"].numpy())
    for k in ("sfm", "P", "K", "c2w"):
        out[f"{key}_{k}"] = sc[k].numpy()


out = {}
quiet = contextlib.redirect_stdout(io.StringIO())

# ---- B1: reprojection + validity filter ------------------------------------------------------
b1_cases = [(64, 96, 400, 0.15, 40), (135, 240, 1500, 0.3, 41), (48, 40, 50, 0.0, 42)]
for i, (H, W, M, frac_out, seed) in enumerate(b1_cases):
    sc = camera_scene(H, W, M, seed, frac_outside=frac_out)
    pd = PredictedDepth(depth=sc["depth"].clone(), mask=sc["mask"].clone())
    with quiet:
        coords, depths = pfd.project_and_filter_sfm_pts(sc["rgb"], sc["sfm"].clone(), sc["P"], (W, H), pd, None)
    _store(f"b1_{i}", sc)
    out[f"b1_{i}_coords"] = coords.numpy()
    out[f"b1_{i}_depths"] = depths.numpy()
# the error branch: most points behind / outside the camera
sc = camera_scene(64, 96, 200, 43, frac_outside=0.9)
pd = PredictedDepth(depth=sc["depth"].clone(), mask=sc["mask"].clone())
try:
    with quiet:
        pfd.project_and_filter_sfm_pts(sc["rgb"], sc["sfm"].clone(), sc["P"], (96, 64), pd, None)
    raised = False
except LowDepthAlignmentConfidenceError:
    raised = True
assert raised
_store("b1_err", sc)
out["b1_n"] = np.int64(len(b1_cases))

# ---- B8: depth-gradient mask -----------------------------------------------------------------
b8_cases = [(64, 96, 50, 0.05), (135, 240, 51, 0.02), (33, 17, 52, 0.2)]
for i, (H, W, seed, thr) in enumerate(b8_cases):
    d = camera_scene(H, W, 10, seed)["depth"]
    out[f"b8_{i}_depth"] = d.numpy()
    out[f"b8_{i}_thr"] = np.float64(thr)
    out[f"b8_{i}_gradmask"] = pfd.depth_gradient_mask(d.clone(), thr).numpy()
out["b8_n"] = np.int64(len(b8_cases))

# ---- B4 + B9: no-segmentation alignment and the full get_pts_from_depth chain ------------------
stored = {}
b9_cases = [
    # (H, W, M, seed, rng_seed, aligner, subsample_factor, grad_thresh, use_num_sfm_mask)
    (64, 96, 400, 60, 70, "lstsqrs", 4, None, False),
    (64, 96, 400, 60, 70, "ransac", 4, None, True),
    (270, 480, 3000, 61, 71, "msac", 10, None, True),
    (135, 240, 1500, 64, 74, "ransac", "adaptive", None, False),
    (135, 240, 1500, 62, 72, "lstsqrs", 10, 0.05, True),
    (135, 240, 1500, 63, 73, "ransac", "adaptive", 0.1, True),
]
for i, (H, W, M, seed, rng_seed, aligner, factor, grad_thr, nsfm) in enumerate(b9_cases):
    sc = camera_scene(H, W, M, seed, frac_outside=0.1)
    cfg = Config()
    cfg.mdi.alignment.aligner = DepthAlignmentStrategyEnum[aligner]
    cfg.mdi.alignment.segmenter = None
    cfg.mdi.subsample_factor = factor
    cfg.mdi.depth_grad_mask_thresh = grad_thr
    cfg.mdi.use_num_sfm_points_mask = nsfm
    image = InputImage(name="img0", cam2world=sc["c2w"], K=sc["K"], data=sc["rgb"])
    # B4 on its own
    pd = PredictedDepth(depth=sc["depth"].clone(), mask=sc["mask"].clone())
    with quiet:
        coords, depths = pfd.project_and_filter_sfm_pts(sc["rgb"], sc["sfm"].clone(), sc["P"], (W, H), pd, None)
        torch.manual_seed(rng_seed)
        res = DepthAlignmentPipeline.from_config(cfg).align(image, pd, coords, depths, cfg, None)
    # the full map: it is the INPUT of the mask / unprojection kernels in the "identical inputs"
    # test (an LSQ restated on another CPU differs from it in the last bit)
    out[f"b9_{i}_aligned"] = res.aligned_depth.numpy()
    out[f"b9_{i}_align_mask"] = np.packbits(res.mask.numpy())
    # the whole chain
    pd = PredictedDepth(depth=sc["depth"].clone(), mask=sc["mask"].clone())
    with quiet:
        torch.manual_seed(rng_seed)
        pts, mask, P = pfd.get_pts_from_depth(pd, image, _Parser("img0", sc["sfm"]), cfg, "cpu", None)
    if (H, W, M, seed) not in stored:
        stored[(H, W, M, seed)] = i
        _store(f"b9_{i}", sc)
    out[f"b9_{i}_scene_of"] = np.int64(stored[(H, W, M, seed)])     # cases sharing one stored scene
    out[f"b9_{i}_cfg"] = np.array([aligner, str(factor), str(grad_thr), str(int(nsfm))])
    out[f"b9_{i}_rng_seed"] = np.int64(rng_seed)
    out[f"b9_{i}_pts"] = pts.numpy()
    out[f"b9_{i}_final_mask"] = np.packbits(mask.numpy())
    out[f"b9_{i}_P"] = P.numpy()
    print(i, aligner, factor, grad_thr, nsfm, "->", pts.shape[0], "points")
out["b9_n"] = np.int64(len(b9_cases))

np.savez_compressed(Path(__file__).resolve().parent / "points_golden.npz", **out)
print("wrote points_golden.npz")
