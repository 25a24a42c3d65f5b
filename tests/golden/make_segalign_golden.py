"""Generate tests/golden/segalign_golden.npz by running the REFERENCE's own
gs_init_compare/depth_alignment/segmentation/region_margin.py (`calculate_region_margin_mask`,
:21-35) and the segmentation branch of depth_alignment/pipeline.py (`DepthAlignmentPipeline.align`,
:193-288: region ids over the valid pixels, per-region SfM subsets inside the margin mask,
per-region alignment written through the region masks, optional mask propagation).

Run only in the build container: python tests/golden/make_segalign_golden.py

What the reference's branch needs that is absent here: scikit-image (SLIC, and the region
adjacency graph / morphology of region_merging.py) and segment_anything. The recorded runs
therefore use (a) a label map of our own as the segmenter's output (any callable with the
`DepthSegmentationFn` signature is a segmenter, interface.py:44-46) and (b) the identity in
place of `merge_segmentation_regions` -- the label array turned into a tensor, which is what the
merge returns when nothing needs merging. Both are inputs of the recorded code, not part of it:
everything that is recorded -- the margin mask, the region bookkeeping, the aligners, the
composition of the output -- is the reference's. As in make_points_golden.py, INERT placeholder
modules are registered for the absent packages so that the imports succeed; nothing a placeholder
returns takes part in the recorded numbers. The output is data only (inputs are stored too:
depth, mask, SfM pixel coordinates and depths, label maps).
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
from gs_init_compare.depth_alignment import pipeline as ref_pipeline  # noqa: E402
from gs_init_compare.depth_alignment.config import DepthAlignmentStrategyEnum  # noqa: E402
from gs_init_compare.depth_alignment.segmentation.region_margin import (  # noqa: E402
    calculate_region_margin_mask, get_actual_margin_size)
from gs_init_compare.depth_prediction import points_from_depth as pfd  # noqa: E402
from gs_init_compare.depth_prediction.predictors.depth_predictor_interface import PredictedDepth  # noqa: E402
from gs_init_compare.types import InputImage  # noqa: E402

sys.path.insert(0, str(Path(__file__).resolve().parent))
from make_points_golden_scene import camera_scene  # noqa: E402

quiet = contextlib.redirect_stdout(io.StringIO())
# (b) above: nothing to merge -> the label array as a tensor on the depth's device
ref_pipeline.merge_segmentation_regions = lambda pd, coords, seg, cfg: torch.from_numpy(np.asarray(seg)).to(pd.depth.device)


def label_map(H, W, ny, nx, seed, values=None):
    """ny x nx blocks with wavy borders (integer arithmetic only), labels 0..ny*nx-1 (or `values`)."""
    y, x = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    wob_x = ((y * 7 + seed * 13) % 11) - 5
    wob_y = ((x * 5 + seed * 17) % 9) - 4
    bx = np.clip((x + wob_x) * nx // W, 0, nx - 1)
    by = np.clip((y + wob_y) * ny // H, 0, ny - 1)
    lab = (by * nx + bx).astype(np.int64)
    if values is not None:
        lab = np.asarray(values, dtype=np.int64)[lab]
    return lab


out = {}
# ---- region margin mask alone ---------------------------------------------------------------------
mm_cases = [
    # (H, W, ny, nx, seed, region_margin, label values)
    (135, 240, 3, 4, 1, 10, None),          # half width 1
    (135, 240, 3, 4, 2, 40, None),          # 7
    (96, 64, 4, 2, 3, 40, None),            # 2 (portrait)
    (270, 480, 2, 3, 4, 0, None),           # margin 0: all true
    (270, 480, 5, 6, 5, 25, None),          # 9
    (135, 240, 2, 2, 6, 40, (0, 700, 1400, 2100)),   # labels large enough for isclose to snap 1/k^2 steps
    (64, 96, 1, 1, 7, 30, None),            # a single region
]
for i, (H, W, ny, nx, seed, margin, values) in enumerate(mm_cases):
    lab = label_map(H, W, ny, nx, seed, values)
    m = calculate_region_margin_mask(torch.from_numpy(lab), margin)
    out[f"mm_{i}_labels"] = lab.astype(np.int32)
    out[f"mm_{i}_margin"] = np.int64(margin)
    out[f"mm_{i}_half"] = np.int64(get_actual_margin_size(lab.shape, margin))
    out[f"mm_{i}_mask"] = np.packbits(m.numpy())
    print("margin", i, (H, W), "half width", int(out[f"mm_{i}_half"]), "interior", float(m.float().mean()))
out["mm_n"] = np.int64(len(mm_cases))

# ---- the segmentation branch of the pipeline -------------------------------------------------------
sa_cases = [
    # (H, W, M, scene seed, rng seed, aligner, ny, nx, label seed, region_margin, propagate_mask)
    (135, 240, 1500, 80, 90, "lstsqrs", 3, 4, 11, 10, False),
    (135, 240, 1500, 80, 90, "lstsqrs", 3, 4, 11, 40, True),
    (135, 240, 1500, 81, 91, "ransac", 2, 3, 12, 25, False),
    (96, 64, 300, 82, 92, "lstsqrs", 6, 4, 13, 60, False),     # small regions: some lose all their points
    (135, 240, 1500, 83, 93, "msac", 2, 2, 14, 40, True),
]
for i, (H, W, M, seed, rng_seed, aligner, ny, nx, lseed, margin, propagate) in enumerate(sa_cases):
    sc = camera_scene(H, W, M, seed, frac_outside=0.1)
    lab = label_map(H, W, ny, nx, lseed)
    cfg = Config()
    cfg.mdi.alignment.aligner = DepthAlignmentStrategyEnum[aligner]
    cfg.mdi.alignment.segmentation.region_margin = margin
    cfg.mdi.alignment.segmentation.propagate_mask = propagate
    image = InputImage(name="img0", cam2world=sc["c2w"], K=sc["K"], data=sc["rgb"])
    pd = PredictedDepth(depth=sc["depth"].clone(), mask=sc["mask"].clone())
    with quiet:
        coords, depths = pfd.project_and_filter_sfm_pts(sc["rgb"], sc["sfm"].clone(), sc["P"], (W, H), pd, None)
    pipe = ref_pipeline.DepthAlignmentPipeline(cfg, lambda p, ckpt, scfg, lab=lab: lab.copy(),
                                               cfg.mdi.alignment.aligner.get_implementation())
    with quiet:
        torch.manual_seed(rng_seed)
        res = pipe.align(image, pd, coords, depths, cfg, None)
    out[f"sa_{i}_depth"] = sc["depth"].numpy()
    out[f"sa_{i}_mask"] = np.packbits(sc["mask"].numpy())
    out[f"sa_{i}_coords"] = coords.numpy()
    out[f"sa_{i}_gt"] = depths.numpy()
    out[f"sa_{i}_labels"] = lab.astype(np.int32)
    out[f"sa_{i}_cfg"] = np.array([aligner, str(margin), str(int(propagate)), str(rng_seed)])
    out[f"sa_{i}_aligned"] = res.aligned_depth.numpy()
    out[f"sa_{i}_out_mask"] = np.packbits(res.mask.numpy())
    out[f"sa_{i}_pd_mask_after"] = np.packbits(pd.mask.numpy())      # propagate_mask edits the input in place
    dropped = int((res.aligned_depth == -42.0).sum())
    print("pipeline", i, aligner, (H, W), "margin", margin, "propagate", propagate, "invalid pixels", dropped,
          "valid", int(res.mask.sum()))
out["sa_n"] = np.int64(len(sa_cases))

np.savez_compressed(Path(__file__).resolve().parent / "segalign_golden.npz", **out)
print("wrote segalign_golden.npz")
