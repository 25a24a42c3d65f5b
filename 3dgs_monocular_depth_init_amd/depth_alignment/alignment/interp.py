"""Scale-map interpolation alignment (tail).

Mirror of /root/reference/gs_init_compare/depth_alignment/alignment/interp.py:
  linear_interpolation            77-110
  scale_factor_outlier_removal    161-201
  initial_alignment               204-235
  align_depth_interpolate         281-361, DepthAlignmentInterpolate 364-380
The per-image pre-alignment is this build's RANSAC / LSQ (HIP). What the reference does on a few
thousand SfM points with CPU libraries stays exactly that -- scikit-learn's LocalOutlierFactor /
NearestNeighbors and scipy's Delaunay on the host (their neighbour tie-breaking and
triangulation are part of the reference's results) -- while the expensive part, evaluating the
piecewise-linear scale map at every pixel (LinearNDInterpolator on ~2 M queries on the CPU in
the reference), runs as one HIP launch per image (`gsr_tri_interp`). method="rbf" needs torchrbf
(absent) and is not built.
"""
from __future__ import annotations

import logging
from pathlib import Path
from typing import NamedTuple, Optional

import numpy as np
import torch

from ..._lib import call, ptr
from ..interface import DepthAlignmentResult, DepthAlignmentStrategy
from .lstsqrs import DepthAlignmentLstSqrs
from .ransacs import DepthAlignmentRansac

LOGGER = logging.getLogger(__name__)


class OutlierClassification(NamedTuple):
    scale_only_outliers: torch.Tensor
    both_outliers: torch.Tensor
    position_only_outliers: torch.Tensor
    regular: torch.Tensor


def scale_factor_outlier_removal(coords: torch.Tensor, scales: torch.Tensor, debug_export_dir=None):
    """interp.py:161-201. coords [M,2] pixel coordinates, scales [M] (any device)."""
    from sklearn.neighbors import LocalOutlierFactor, NearestNeighbors
    K_lof, K_scale_knn = 10, 5
    num_pts = coords.shape[0]
    if num_pts < min(K_lof + 1, K_scale_knn + 1):
        z = torch.zeros(num_pts, dtype=torch.bool)
        return OutlierClassification(z, z.clone(), z.clone(), torch.ones(num_pts, dtype=torch.bool))
    coords_np = coords.cpu().numpy()
    position_outliers_np = LocalOutlierFactor(n_neighbors=K_lof, n_jobs=-1).fit_predict(coords_np) == -1
    model = NearestNeighbors(n_neighbors=K_scale_knn + 1, metric="euclidean").fit(coords_np)
    _, knn_indices = model.kneighbors(coords_np)
    knn_indices = torch.from_numpy(knn_indices[:, 1:]).to(scales.device)
    knn_median_scale = torch.median(scales[knn_indices], dim=1).values
    scale_diff = torch.abs(scales - knn_median_scale)
    scale_outliers = scale_diff > torch.quantile(scale_diff, 0.99)
    position_outliers = torch.from_numpy(position_outliers_np).to(scale_outliers.device)
    return OutlierClassification(
        scale_only_outliers=scale_outliers & ~position_outliers,
        both_outliers=scale_outliers & position_outliers,
        position_only_outliers=position_outliers & ~scale_outliers,
        regular=~(scale_outliers | position_outliers))


def linear_interpolation(coords: torch.Tensor, values: torch.Tensor, config, device, W: int, H: int) -> torch.Tensor:
    """interp.py:77-110: Delaunay over the SfM pixels + the four image corners (corner values =
    inverse-distance mean of their non-corner neighbours), piecewise-linear on every pixel.
    Returns [H,W] on the device of `values`."""
    from scipy.spatial import Delaunay
    coords_np = coords.T.cpu().numpy()
    values_np = values.cpu().numpy()
    corner_coords = np.array([[0, 0], [0, H - 1], [W - 1, 0], [W - 1, H - 1]])
    corner_indices = np.arange(coords_np.shape[0], coords_np.shape[0] + 4)
    coords_np = np.vstack((coords_np, corner_coords))
    values_np = np.hstack((values_np, np.empty(4, dtype=values_np.dtype)))
    dt = Delaunay(coords_np)
    for corner_ix in corner_indices:
        indptr, indices = dt.vertex_neighbor_vertices
        neighbors = np.setdiff1d(indices[indptr[corner_ix]: indptr[corner_ix + 1]], corner_indices)
        distances = np.linalg.norm(coords_np[neighbors] - coords_np[corner_ix], axis=1)
        weights = 1.0 / (distances + 1e-8)
        weights /= np.sum(weights)
        corner_value = np.sum(values_np[neighbors] * weights)
        if np.isnan(corner_value):
            corner_value = np.median(values_np[neighbors])
        values_np[corner_ix] = corner_value
    dev = values.device if values.is_cuda else torch.device(device)
    out = torch.full((H, W), float(np.median(values_np)), dtype=torch.float32, device=dev)     # fill_value
    xy = torch.from_numpy(np.ascontiguousarray(dt.points, dtype=np.float64)).to(dev)
    tris = torch.from_numpy(np.ascontiguousarray(dt.simplices, dtype=np.int32)).to(dev)
    vals = torch.from_numpy(values_np.astype(np.float64)).to(dev)
    call("gsr_tri_interp", H, W, tris.shape[0], ptr(xy), ptr(tris), ptr(vals), ptr(out),
         torch.cuda.current_stream().cuda_stream)
    return out.to(values.dtype)


def interpolate_scale(coords, values, config, device, W, H):
    if config.method == "linear":
        return linear_interpolation(coords, values, config, device, W, H)
    if config.method == "rbf":
        raise NotImplementedError("interp.method='rbf' needs torchrbf, which is not available; use 'linear'")
    raise ValueError(f"Unknown interpolation method: {config.method}")


def initial_alignment(predicted_depth, sfm_points_camera_coords, gt_depth, config, debug_export_dir=None):
    """interp.py:204-235."""
    init = config.mdi.alignment.interp.init
    if init is None:
        return DepthAlignmentResult(predicted_depth.depth, predicted_depth.mask)
    if init == "lstsqrs":
        return DepthAlignmentLstSqrs.align(predicted_depth, sfm_points_camera_coords, gt_depth, config, debug_export_dir)
    if init == "ransac":
        return DepthAlignmentRansac.align(predicted_depth, sfm_points_camera_coords, gt_depth, config, debug_export_dir)
    raise ValueError(f"Unknown interp alignment init method: {init}")


def align_depth_interpolate(predicted_depth, sfm_points_camera_coords, gt_depth, config,
                            debug_export_dir: Optional[Path] = None, return_parts: bool = False):
    """interp.py:281-361."""
    H, W = predicted_depth.depth.shape
    num_sfm_pts = sfm_points_camera_coords.shape[1]
    device = predicted_depth.depth.device
    interp_config = config.mdi.alignment.interp
    prealigned = initial_alignment(predicted_depth, sfm_points_camera_coords, gt_depth, config, debug_export_dir)
    scale_factors = gt_depth / prealigned.aligned_depth[sfm_points_camera_coords[1], sfm_points_camera_coords[0]]
    outlier_mask = None
    if interp_config.scale_outlier_removal:
        outlier_mask = scale_factor_outlier_removal(sfm_points_camera_coords.T, scale_factors,
                                                    debug_export_dir).scale_only_outliers.to(device)
        if outlier_mask.sum() > 0:
            LOGGER.info("Removed %d/%d scale outlier points.", outlier_mask.sum().item(), num_sfm_pts)
        scale_factors = scale_factors[~outlier_mask]
        sfm_points_camera_coords = sfm_points_camera_coords[:, ~outlier_mask]
    try:
        scale_map = interpolate_scale(sfm_points_camera_coords, scale_factors, interp_config, device, W, H)
    except NotImplementedError:
        raise
    except Exception as e:  # noqa: BLE001  (reference: any failure -> median scale, interp.py:351-359)
        LOGGER.warning("Scale factor interpolation failed; using median scale instead of interpolation. %s", e)
        scale_map = scale_factors.median()
    res = DepthAlignmentResult(scale_map * prealigned.aligned_depth, prealigned.mask)
    return (res, scale_map, outlier_mask) if return_parts else res


class DepthAlignmentInterpolate(DepthAlignmentStrategy):
    @classmethod
    def align(cls, predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
              debug_export_dir: Optional[Path] = None) -> DepthAlignmentResult:
        return align_depth_interpolate(predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
                                       debug_export_dir)
