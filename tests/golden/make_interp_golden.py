"""Generate tests/golden/interp_golden.npz by running the REFERENCE's own scale-map interpolation
aligner (gs_init_compare/depth_alignment/alignment/interp.py: scale_factor_outlier_removal
161-201, linear_interpolation 77-110, align_depth_interpolate 281-361).

Run only in the build container: python tests/golden/make_interp_golden.py

interp.py is torch + scipy + scikit-learn (all present), but its import chain names packages
that are absent here (torchrbf -- only used by method="rbf" --, gsplat, pointcloud_subsampling,
segment_anything, skimage, cv2, ...). As in make_points_golden.py, this script -- and only it --
registers INERT placeholder modules for those names so that the import succeeds; nothing a
placeholder returns takes part in the recorded numbers (method="linear", no segmentation, no
debug export). The output is data only: seeded inputs and the reference's outputs.
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

ABSENT = ("gsplat", "pointcloud_subsampling", "pycolmap", "cv2", "imageio", "segment_anything", "skimage",
          "nerfbaselines", "open3d", "torchrbf")


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
from gs_init_compare.depth_alignment.alignment import interp as RI  # noqa: E402
from gs_init_compare.depth_alignment.config import DepthAlignmentStrategyEnum  # noqa: E402
from gs_init_compare.depth_prediction.predictors.depth_predictor_interface import PredictedDepth  # noqa: E402


def scene(H, W, M, seed):
    """Depth with a scale error that varies smoothly over the image (what the interpolation is for)."""
    g = torch.Generator().manual_seed(seed)
    yy, xx = torch.meshgrid(torch.linspace(0, 1, H), torch.linspace(0, 1, W), indexing="ij")
    t = torch.clamp((xx + 0.5 * yy) / 1.5, 0, 1)
    depth = (2.0 + 6.0 * (t * t * (3 - 2 * t)) + 0.05 * torch.randn(H, W, generator=g)).float()
    mask = torch.rand(H, W, generator=g) > 0.05
    xs = torch.randint(0, W, (M,), generator=g)
    ys = torch.randint(0, H, (M,), generator=g)
    field = 1.5 + 0.5 * xx - 0.3 * yy
    gt = (field * depth)[ys, xs] + 0.3 + 0.01 * torch.randn(M, generator=g)
    outl = torch.rand(M, generator=g) < 0.05
    gt = torch.where(outl, gt * (0.5 + 1.5 * torch.rand(M, generator=g)), gt)
    return depth, mask, torch.stack([xs, ys]).long(), gt.float()


out = {}
quiet = contextlib.redirect_stdout(io.StringIO())
cases = [(64, 96, 300, 80, 90, "lstsqrs", True), (64, 96, 300, 80, 90, None, False),
         (135, 240, 1500, 81, 91, "ransac", True), (135, 240, 1500, 82, 92, "lstsqrs", False)]
for i, (H, W, M, seed, rng_seed, init, removal) in enumerate(cases):
    depth, mask, coords, gt = scene(H, W, M, seed)
    cfg = Config()
    cfg.mdi.alignment.aligner = DepthAlignmentStrategyEnum.interp
    cfg.mdi.alignment.interp.method = "linear"
    cfg.mdi.alignment.interp.init = init
    cfg.mdi.alignment.interp.scale_outlier_removal = removal
    pd = PredictedDepth(depth=depth.clone(), mask=mask.clone())
    with quiet:
        torch.manual_seed(rng_seed)
        pre = RI.initial_alignment(pd, coords, gt, cfg, None)
        sf = gt / pre.aligned_depth[coords[1], coords[0]]
        oc = RI.scale_factor_outlier_removal(coords.T, sf, None)
        keep = ~oc.scale_only_outliers if removal else torch.ones(M, dtype=torch.bool)
        smap = RI.linear_interpolation(coords[:, keep], sf[keep], cfg.mdi.alignment.interp, "cpu", W, H)
        torch.manual_seed(rng_seed)
        res = RI.align_depth_interpolate(PredictedDepth(depth=depth.clone(), mask=mask.clone()), coords, gt, cfg, None)
    out[f"i{i}_depth"] = depth.numpy(); out[f"i{i}_mask"] = np.packbits(mask.numpy())
    out[f"i{i}_coords"] = coords.numpy(); out[f"i{i}_gt"] = gt.numpy()
    out[f"i{i}_cfg"] = np.array([str(init), str(int(removal)), str(rng_seed)])
    out[f"i{i}_prealigned"] = pre.aligned_depth.numpy()
    out[f"i{i}_scale_factors"] = sf.numpy()
    out[f"i{i}_scale_only_outliers"] = oc.scale_only_outliers.numpy()
    out[f"i{i}_position_only_outliers"] = oc.position_only_outliers.numpy()
    out[f"i{i}_scale_map"] = smap.numpy()
    out[f"i{i}_aligned"] = res.aligned_depth.numpy()
    print(i, init, removal, "outliers", int(oc.scale_only_outliers.sum()), "scale map range", float(smap.min()), float(smap.max()))
out["n"] = np.int64(len(cases))
np.savez_compressed(Path(__file__).resolve().parent / "interp_golden.npz", **out)
print("wrote interp_golden.npz")
