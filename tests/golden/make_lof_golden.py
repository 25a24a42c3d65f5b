"""Generate tests/golden/lof_golden.npz by running the REFERENCE's own
gs_init_compare/point_cloud_postprocess/postprocess.py:16-22 (`lof_outlier_removal`: scikit-learn's
LocalOutlierFactor(n_neighbors=config.lof_num_neighbors, n_jobs=-1).fit_predict == -1) on seeded
synthetic clouds, and scikit-learn's `negative_outlier_factor_` for the same fit beside it (the
library call the reference makes; its scores let the tests compare numbers, not only the mask).

Run only in the build container: python tests/golden/make_lof_golden.py
The reference module imports `pointcloud_subsampling` (its own native module, not built here) and
`gs_init_compare.utils.point_cloud_export` (open3d): INERT placeholders are registered for the
absent packages, as in make_points_golden.py; the recorded function executes none of them.
The clouds are stored (float32), the output is data only.
"""
import importlib.abc
import importlib.machinery
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
from gs_init_compare.point_cloud_postprocess.config import PointCloudPostprocessConfig  # noqa: E402
from gs_init_compare.point_cloud_postprocess.postprocess import lof_outlier_removal  # noqa: E402
from sklearn.neighbors import LocalOutlierFactor  # noqa: E402


def cloud(n, seed):
    """Three blobs of different density, a thin slab and 3 % uniform background."""
    g = torch.Generator().manual_seed(seed)
    n_bg = max(n * 3 // 100, 1)
    n_slab = n // 4
    n_blob = (n - n_bg - n_slab) // 3
    parts = [torch.randn(n_blob, 3, generator=g) * s + torch.tensor(c)
             for s, c in ((0.05, (0.0, 0.0, 0.0)), (0.15, (1.0, 0.5, 0.2)), (0.4, (-1.0, 0.3, 1.5)))]
    slab = torch.rand(n_slab, 3, generator=g) * torch.tensor([2.0, 2.0, 0.01]) + torch.tensor([-1.0, -1.0, -0.8])
    rest = n - 3 * n_blob - n_slab
    bg = (torch.rand(rest, 3, generator=g) - 0.5) * 6.0
    pts = torch.cat(parts + [slab, bg])
    return pts[torch.randperm(n, generator=g)].float().contiguous()


out = {}
cases = [(30000, 40, 1), (5000, 40, 2), (30, 40, 3), (2000, 10, 4), (12000, 64, 5)]
for i, (n, k, seed) in enumerate(cases):
    pts = cloud(n, seed)
    cfg = PointCloudPostprocessConfig()
    cfg.lof_num_neighbors = k
    mask = lof_outlier_removal(pts, cfg)
    clf = LocalOutlierFactor(n_neighbors=k, n_jobs=-1)
    pred = clf.fit_predict(pts.numpy())
    assert np.array_equal(mask, pred == -1)
    out[f"c{i}_pts"] = pts.numpy()
    out[f"c{i}_k"] = np.int64(k)
    out[f"c{i}_outlier"] = np.packbits(mask)
    out[f"c{i}_nof"] = clf.negative_outlier_factor_.astype(np.float64)
    print(i, n, k, "outliers", int(mask.sum()), "min |nof + 1.5|", float(np.abs(clf.negative_outlier_factor_ + 1.5).min()))
out["n"] = np.int64(len(cases))
np.savez_compressed(Path(__file__).resolve().parent / "lof_golden.npz", **out)
print("wrote lof_golden.npz")
