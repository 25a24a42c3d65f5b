"""Scale-map interpolation alignment (tail).

Counterpart of /root/reference/gs_init_compare/depth_alignment/alignment/interp.py:
  linear_interpolation            77-110
  scale_factor_outlier_removal    161-201
  initial_alignment               204-235
  align_depth_interpolate         281-361, DepthAlignmentInterpolate 364-380
Everything per point runs on the device with this repository's own kernels: the position-outlier
test is `knn.local_outlier_factor` (gsr_knn_grid_idx + gsr_lof, the kernels pinned against
scikit-learn by tests/test_lof_golden.py), the scale-outlier test takes the 5 nearest other points
from `knn.knn_neighbors`, the per-image pre-alignment is the RANSAC / LSQ of this build, and the
piecewise-linear scale map is evaluated at every pixel by one launch (`gsr_tri_interp`; the
reference runs scipy's LinearNDInterpolator over ~2 M pixels on the CPU). The one host library
left is scipy's Delaunay triangulation of the few thousand SfM pixels (the triangulation IS the
reference's result; Qhull is what both use). scikit-learn is not imported here.
Neighbour ties: SfM pixels are integers, so several points can sit at exactly the distance of the
K-th neighbour; which of them scikit-learn's KD-tree reports is an accident of its traversal. The
kernels here break ties by point index. A classification can differ from the reference's only for
a point with such a tie (tests/test_interp_golden.py checks exactly that).
method="rbf" (interp.py:30-72): the reference fits a torchrbf.RBFInterpolator -- a port of
scipy.interpolate.RBFInterpolator; neither is used here -- and this file runs the same published algorithm on
the device in float64 (csrc/rbf.hip: dense LU of the (P + 3)-square system, grid evaluation, bilinear
upsampling), restated and pinned against scipy in oracle/rbf_oracle.py / tests/test_rbf.py; parity against
torchrbf itself is unpinned (the package is absent; it runs in float32).
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


N_POSITION_NEIGHBOURS = 10     # local outlier factor of the pixel positions (interp.py:170)
N_SCALE_NEIGHBOURS = 5         # a point's scale is compared with the median over this many neighbours


def _on_device(t: torch.Tensor) -> torch.Tensor:
    return t if t.is_cuda else t.cuda()


def scale_factor_outlier_removal(coords: torch.Tensor, scales: torch.Tensor, debug_export_dir=None):
    """interp.py:161-201. coords [M,2] pixel coordinates, scales [M]; masks come back on the
    device of `scales`. Position outliers: local outlier factor (10 neighbours, threshold -1.5);
    scale outliers: |scale - median scale of the 5 nearest other points| above its own 0.99
    quantile. Fewer than 6 points: nothing is an outlier."""
    from ... import knn
    M = coords.shape[0]
    out_dev = scales.device
    if M < min(N_POSITION_NEIGHBOURS, N_SCALE_NEIGHBOURS) + 1:
        none = torch.zeros(M, dtype=torch.bool, device=out_dev)
        return OutlierClassification(none, none.clone(), none.clone(), ~none)
    xy = _on_device(coords).float()
    pts = torch.cat([xy, torch.zeros(M, 1, device=xy.device)], dim=1).contiguous()   # the kNN kernels are 3-D
    sc = _on_device(scales)
    position_out, _ = knn.local_outlier_factor(pts, N_POSITION_NEIGHBOURS)
    _, nb = knn.knn_neighbors(pts, N_SCALE_NEIGHBOURS)
    deviation = (sc - sc[nb.long()].median(dim=1).values).abs()
    scale_out = deviation > torch.quantile(deviation, 0.99)
    return OutlierClassification(
        scale_only_outliers=(scale_out & ~position_out).to(out_dev),
        both_outliers=(scale_out & position_out).to(out_dev),
        position_only_outliers=(position_out & ~scale_out).to(out_dev),
        regular=(~(scale_out | position_out)).to(out_dev))


def _corner_values(simplices: np.ndarray, xy: np.ndarray, values: np.ndarray, n_pts: int) -> np.ndarray:
    """Values of the four appended image corners (vertex ids n_pts .. n_pts+3): the mean of the
    corner's triangulation neighbours among the real points, weighted by inverse distance
    (interp.py:92-103). A corner's neighbours are the other vertices of the triangles it belongs to."""
    out = np.empty(4, dtype=values.dtype)
    for k in range(4):
        c = n_pts + k
        ring = np.unique(simplices[(simplices == c).any(axis=1)])
        ring = ring[ring < n_pts]                                   # real points only, ascending
        w = 1.0 / (np.hypot(*(xy[ring] - xy[c]).T) + 1e-8)
        v = np.sum(values[ring] * (w / np.sum(w)))
        out[k] = v if not np.isnan(v) else np.median(values[ring])
    return out


def linear_interpolation(coords: torch.Tensor, values: torch.Tensor, config, device, W: int, H: int) -> torch.Tensor:
    """interp.py:77-110: Delaunay over the SfM pixels + the four image corners, piecewise-linear
    on every pixel (pixels outside the hull: the median value). Returns [H,W] on the device of
    `values` (or `device` for CPU input)."""
    from scipy.spatial import Delaunay
    n_pts = coords.shape[1]
    corners = np.array([[0, 0], [0, H - 1], [W - 1, 0], [W - 1, H - 1]], dtype=np.float64)
    xy = np.concatenate([coords.T.cpu().numpy().astype(np.float64), corners])
    vals = values.cpu().numpy()
    tri = Delaunay(xy)
    vals = np.concatenate([vals, _corner_values(tri.simplices, xy, vals, n_pts)])
    dev = values.device if values.is_cuda else torch.device(device)
    out = torch.full((H, W), float(np.median(vals)), dtype=torch.float32, device=dev)
    xy_d = torch.from_numpy(np.ascontiguousarray(tri.points, dtype=np.float64)).to(dev)
    tri_d = torch.from_numpy(np.ascontiguousarray(tri.simplices, dtype=np.int32)).to(dev)
    val_d = torch.from_numpy(vals.astype(np.float64)).to(dev)
    call("gsr_tri_interp", H, W, tri_d.shape[0], ptr(xy_d), ptr(tri_d), ptr(val_d), ptr(out),
         torch.cuda.current_stream().cuda_stream)
    return out.to(values.dtype)


# The scale-invariant kernels of torchrbf / scipy: the ones the reference's call can run at all -- it never passes
# `epsilon`, and the library refuses every other kernel without it (gaussian, multiquadric, inverse_multiquadric,
# inverse_quadratic), upon which the caller's `except Exception` falls back to the median scale (interp.py:345-359).
RBF_KERNELS = {"linear": 0, "thin_plate_spline": 1, "cubic": 2, "quintic": 3}


def rbf_interpolation(coords: torch.Tensor, values: torch.Tensor, config, device, W: int, H: int) -> torch.Tensor:
    """interp.py:30-72: RBF interpolant (kernel / smoothing of the config, polynomial of the kernel's minimum
    degree) over the pixels normalised by (W - 1, H - 1), evaluated on a grid 256 pixels wide, upsampled
    bilinearly (align_corners) to [H, W]. `gsr_rbf_fit` + `gsr_rbf_eval_grid` + `gsr_bilinear_ac_t`."""
    if config.kernel not in RBF_KERNELS:        # torchrbf's own refusal (-> the caller's median-scale fallback)
        raise ValueError(f"`epsilon` must be specified if `kernel` is not one of {sorted(RBF_KERNELS)}; got {config.kernel!r}")
    from ..._lib import load
    lib = load()
    dev = values.device if values.is_cuda else torch.device(device)
    st = torch.cuda.current_stream(dev).cuda_stream
    P = coords.shape[1]
    c = coords.to(dev).float()
    sites = torch.stack([c[0] / (W - 1.0), c[1] / (H - 1.0)], dim=1).contiguous()            # coords_norm
    vals = values.to(dev).float().contiguous()
    nbytes = int(lib.gsr_rbf_workspace_bytes(P))
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    coeffs = torch.empty(P + 6, dtype=torch.float64, device=dev)
    ss = torch.empty(4, dtype=torch.float64, device=dev)
    k = RBF_KERNELS[config.kernel]
    call("gsr_rbf_fit", P, ptr(sites), ptr(vals), float(config.smoothing), k, ptr(ws), nbytes, ptr(coeffs), ptr(ss), st)
    factor = max(W / 256, 1)
    qw, qh = int(W / factor), int(H / factor)
    grid = torch.empty(qw, qh, dtype=torch.float32, device=dev)
    call("gsr_rbf_eval_grid", P, ptr(sites), ptr(coeffs), ptr(ss), k, qw, qh, ptr(grid), st)
    out = torch.empty(H, W, dtype=torch.float32, device=dev)
    call("gsr_bilinear_ac_t", qw, qh, ptr(grid), W, H, ptr(out), st)
    return out.to(values.dtype)


def interpolate_scale(coords, values, config, device, W, H):
    if config.method == "linear":
        return linear_interpolation(coords, values, config, device, W, H)
    if config.method == "rbf":
        return rbf_interpolation(coords, values, config, device, W, H)
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
    # limit the number of points of the dense RBF system (interp.py:308-324, pick_rbf_point_subset :129-143)
    if interp_config.method == "rbf" and interp_config.max_rbf_points != -1 and num_sfm_pts > interp_config.max_rbf_points:
        indices = torch.randperm(num_sfm_pts, device=device)[:interp_config.max_rbf_points]
        sfm_points_camera_coords, gt_depth = sfm_points_camera_coords[:, indices], gt_depth[indices]
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
