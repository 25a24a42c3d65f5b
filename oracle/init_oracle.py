"""CPU oracle for the monocular-depth initialisation path.  TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg
may import this module.

Plain-torch restatement (CPU) of the reference's own Python for rows B1-B9 of
SURVEY.md section 8a; every function cites the lines it follows under
/root/reference/gs_init_compare/.

Pinning status:
  * PINNED by golden vectors produced by importing the reference modules that
    import cleanly in the build container (tests/golden/make_init_golden.py):
    align_depth_least_squares / DepthAlignmentLstSqrs (B2), StaticDepthSubsampler
    (B5), AdaptiveDepthSubsampler + helpers (B6), calculate_patch_sizes /
    num_sfm_points_mask (B7), rgb_to_sh, knn (A9).
  * PINNED by a recorded run of the reference's own ransacs.py (8 seeded cases,
    tests/golden/make_ransac_golden.py -> ransac_golden.npz; that script registers
    inert placeholders for the two absent third-party NAMES config.py imports for
    its dataclass fields, nothing else): the RANSAC/MSAC loop (B3) -- iteration
    count, inlier count, scale/shift and aligned map all reproduce exactly.
  * PARITY UNPINNED (restated from the cited lines only):
    get_valid_sfm_pts / project_and_filter_sfm_pts (B1), the pipeline's
    no-segmentation branch (B4), depth_gradient_mask (B8) and the unprojection
    (B9): points_from_depth.py / pipeline.py import dataset, viewer and
    segmentation packages that are absent (ordinary ImportError), and the
    reference holds no tests or fixtures for them.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np
import torch
from torch import Tensor

INVALID_DEPTH_VAL = -42.0          # depth_alignment/pipeline.py:253


class LowDepthAlignmentConfidenceError(Exception):
    """depth_alignment/exceptions.py"""


# ---- B2: depth_alignment/alignment/lstsqrs.py:9-54 --------------------------
def align_depth_least_squares(depth: Tensor, gt_depth: Tensor) -> Tuple[Tensor, Tensor]:
    """depth [2,N] (row 1 = ones), gt [N] -> (scale, shift). lstsqrs.py:9-26."""
    outer_product = torch.einsum("ib,jb->bij", depth, depth)
    h = torch.linalg.pinv(torch.sum(outer_product, axis=0)) @ torch.sum(depth * gt_depth, axis=1)
    return h[0], h[1]


def lstsq_align(depth_map: Tensor, coords: Tensor, gt: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """DepthAlignmentLstSqrs.align, lstsqrs.py:29-54. coords [2,M] row 0 = x."""
    d = depth_map[coords[1], coords[0]].flatten()
    scale, shift = align_depth_least_squares(torch.vstack([d, torch.ones(gt.numel())]), gt)
    return scale, shift, depth_map * scale + shift


# ---- B3: depth_alignment/alignment/ransacs.py:60-189 ------------------------
@dataclass
class RansacConfig:                 # depth_alignment/config.py:104-110
    inlier_threshold: float = 0.01
    max_iters: int = 2500
    confidence: float = 0.999
    sample_size: int = 4
    min_iters: int = 0


def ransac_loss(dists: Tensor, thr: float):          # ransacs.py:60-61
    return torch.sum(dists >= thr)


def msac_loss(dists: Tensor, thr: float):            # ransacs.py:64-65
    return torch.sum(torch.minimum(dists, torch.full_like(dists, thr)))


def required_samples(inlier_count, total, min_sample_size, confidence):   # ransacs.py:79-91
    inlier_ratio = inlier_count / total
    try:
        return math.log(1 - confidence) / math.log(1 - inlier_ratio ** min_sample_size)
    except (ZeroDivisionError, ValueError):
        return 0


def ransac_align(depth_map: Tensor, coords: Tensor, gt_depth: Tensor, loss: str = "ransac",
                 cfg: RansacConfig = RansacConfig()):
    """_align_depth_ransac_generic, ransacs.py:100-189. Consumes the global torch
    RNG exactly like the reference (one randperm per iteration, line 131).
    Returns (scale, shift, aligned, iterations, num_inliers)."""
    loss_func = ransac_loss if loss == "ransac" else msac_loss
    depth = depth_map[coords[1], coords[0]].flatten()
    num_samples = depth.shape[0]
    depth = torch.vstack([depth.reshape(-1), torch.ones(num_samples)])
    h_best_lo = None
    num_inliers_best_lo = 0
    loss_best_lo = float("inf")
    loss_best_sample = float("inf")
    iteration = -1
    for iteration in range(cfg.max_iters):
        sample_indices = torch.randperm(num_samples)[: cfg.sample_size]
        h_sample = align_depth_least_squares(depth[:, sample_indices], gt_depth[sample_indices])
        dists_sample = (h_sample[0] * depth[0] + h_sample[1] - gt_depth) ** 2
        inlier_indices_sample = dists_sample < cfg.inlier_threshold
        loss_sample = loss_func(dists_sample, cfg.inlier_threshold)
        if loss_sample < loss_best_sample:
            h_lo = align_depth_least_squares(depth[:, inlier_indices_sample],
                                             gt_depth[inlier_indices_sample])
            dists_lo = (h_lo[0] * depth[0] + h_lo[1] - gt_depth) ** 2
            loss_lo = loss_func(dists_lo, cfg.inlier_threshold)
            if loss_lo < loss_best_lo:
                h_best_lo = h_lo
                loss_best_lo = loss_lo
                loss_best_sample = loss_sample
                num_inliers_best_lo = torch.sum(dists_lo < cfg.inlier_threshold)
        if (required_samples(num_inliers_best_lo, num_samples, cfg.sample_size, cfg.confidence)
                <= iteration and h_best_lo is not None and iteration >= cfg.min_iters):
            break
    aligned = depth_map * h_best_lo[0] + h_best_lo[1]
    return h_best_lo[0], h_best_lo[1], aligned, iteration, int(num_inliers_best_lo)


# ---- B4: depth_alignment/pipeline.py:248-288 (no segmentation) ---------------
def pipeline_align_noseg(aligned_depth: Tensor, predicted_mask: Tensor, aligner_mask: Tensor):
    """One region covering the image: out_depth = aligned; mask =
    (out_depth != -42) & predicted mask  (pipeline.py:253-288)."""
    out_depth = torch.full_like(aligned_depth, INVALID_DEPTH_VAL)
    region_mask = torch.ones_like(aligned_depth, dtype=torch.bool)
    out_depth[region_mask] = aligned_depth[region_mask]
    return out_depth, (out_depth != INVALID_DEPTH_VAL) & predicted_mask


# ---- F4 tail: segmentation/region_margin.py:6-35 + pipeline.py:193-288 --------
def get_actual_margin_size(image_shape, region_margin) -> int:       # region_margin.py:16-18
    return int(region_margin * max(image_shape) / 1297)


def region_margin_mask(region_map: Tensor, region_margin: int) -> Tensor:
    """calculate_region_margin_mask, region_margin.py:21-35: fp32 box blur of the label map
    (utils/image_filtering.py:88-103: replicate padding, k x k kernel of 1/k^2), values
    `isclose` to an integer snapped to it, compared with the labels."""
    if region_margin == 0:
        return torch.ones_like(region_map, dtype=torch.bool)
    m = get_actual_margin_size(region_map.shape, region_margin)
    k = 2 * m + 1
    x = region_map[None, None].float()
    padded = torch.nn.functional.pad(x, (m, m, m, m), mode="replicate")
    blurred = torch.nn.functional.conv2d(padded, (torch.ones((k, k)) / (k * k))[None, None])[0, 0]
    nearest = blurred.round()
    blurred = torch.where(torch.isclose(blurred, nearest), nearest, blurred)
    return blurred == region_map


def pipeline_align_seg(depth_map: Tensor, mask: Tensor, coords: Tensor, gt: Tensor, segmentation: Tensor,
                       region_margin: int, propagate_mask: bool, align_fn):
    """The segmentation branch of DepthAlignmentPipeline.align (pipeline.py:193-288) after
    region merging; align_fn(depth_map, coords, gt) -> aligned map (called region by region, in
    ascending id order, so an RNG-consuming aligner draws as in the reference).
    Returns (out_depth, out_mask, predicted mask after the optional propagation)."""
    deadzone = region_margin_mask(segmentation, region_margin)
    region_ids = torch.unique(segmentation[mask])
    if propagate_mask:
        mask = mask & deadzone
    pts_regions = segmentation[coords[1], coords[0]]
    pts_ok = deadzone[coords[1], coords[0]]
    region_points = [torch.where((pts_regions == r) & pts_ok)[0] for r in region_ids]
    out_depth = torch.full_like(depth_map, INVALID_DEPTH_VAL)
    for region in region_ids:
        idx = region_points[region.item()]            # indexed with the region ID, as the reference does
        if idx.numel() == 0:
            continue
        aligned = align_fn(depth_map, coords[:, idx], gt[idx])
        region_mask = segmentation == region
        out_depth[region_mask] = aligned[region_mask]
    return out_depth, (out_depth != INVALID_DEPTH_VAL) & mask, mask


# ---- B5: depth_subsampling/static_subsampler.py:8-22 ------------------------
def static_mask(depth_shape, k: int, mask: Tensor) -> Tensor:
    pixel_coords = torch.cartesian_prod(torch.arange(depth_shape[0]), torch.arange(depth_shape[1]))
    return torch.logical_and(
        torch.logical_and((pixel_coords[:, 0] % k) == 0, (pixel_coords[:, 1] % k) == 0),
        mask.view(-1))


# ---- B6: depth_subsampling/adaptive_subsampling.py:12-17, 48-122 -------------
def _map_to_range(tensor, output_range=(0.0, 1.0), input_range=None):
    if input_range is None:
        input_range = (tensor.min(), tensor.max())
    tensor = tensor - input_range[0]
    tensor /= input_range[1] - input_range[0]
    return (output_range[1] - output_range[0]) * tensor + output_range[0]


def iqr_outlier_bounds(data: Tensor):                 # adaptive_subsampling.py:82-86
    q1 = torch.quantile(data, 0.25)
    q3 = torch.quantile(data, 0.75)
    iqr = q3 - q1
    return q1 - 1.5 * iqr, q3 + 1.5 * iqr


def get_depth_multiplier_map(depth: Tensor, mask: Tensor):   # adaptive_subsampling.py:89-98
    masked_depth = depth[mask]
    outlier_bounds = iqr_outlier_bounds(masked_depth)
    input_range = (max(masked_depth.min(), outlier_bounds[0]),
                   min(masked_depth.max(), outlier_bounds[1]))
    multiplier_map = torch.clamp(_map_to_range(depth, input_range=input_range), 0, 1)
    multiplier_map[~mask] = 0.5
    return 1.0 - multiplier_map


def get_sample_mask(downsample_factor_map: Tensor, image_size) -> Tensor:   # :48-79
    per_pixel_df = (torch.nn.functional.interpolate(
        downsample_factor_map[None, None].to(float), size=image_size, mode="nearest")
        .squeeze().to(int))
    pixel_coords = torch.cartesian_prod(torch.arange(per_pixel_df.shape[0]),
                                        torch.arange(per_pixel_df.shape[1]))
    per_pixel_df[per_pixel_df == 0] = 1
    return torch.logical_and((pixel_coords[:, 0] % per_pixel_df.view(-1)) == 0,
                             (pixel_coords[:, 1] % per_pixel_df.view(-1)) == 0)


def adaptive_mask(rgb_shape, depth: Tensor, mask: Tensor, fmin: int = 5, fmax: int = 15):
    """AdaptiveDepthSubsampler.get_mask, adaptive_subsampling.py:102-122."""
    multiplier_map = get_depth_multiplier_map(depth, mask)
    factor_map = torch.clamp(
        _map_to_range(multiplier_map, output_range=(fmin, fmax), input_range=(0.0, 1.0)), fmin, fmax)
    return torch.logical_and(get_sample_mask(factor_map.to(int), rgb_shape[:2]), mask.view(-1))


# ---- B7: depth_subsampling/num_sfm_points_mask.py:7-64 ----------------------
def calculate_patch_sizes(image_shape, num_patches_small_axis):
    small_axis = int(np.argmin([image_shape[0], image_shape[1]]))
    large_axis = 1 - small_axis
    patch_size_small_axis = int(image_shape[small_axis] // num_patches_small_axis)
    num_patches_large_axis = int(np.ceil(image_shape[large_axis] / patch_size_small_axis))
    patch_size_large_axis = int(image_shape[large_axis] // num_patches_large_axis)
    if small_axis == 0:
        patch_grid = (num_patches_small_axis, int(num_patches_large_axis))
        patch_size = (patch_size_small_axis, patch_size_large_axis)
    else:
        patch_grid = (int(num_patches_large_axis), num_patches_small_axis)
        patch_size = (patch_size_large_axis, patch_size_small_axis)
    return patch_size, patch_grid


def num_sfm_points_mask(sfm_points_camera: Tensor, imsize, num_patches_small_axis=20, threshold=15):
    mask = torch.ones(imsize, dtype=bool)
    patch_size, patch_grid = calculate_patch_sizes(imsize, num_patches_small_axis)
    for i in range(patch_grid[0]):
        for j in range(patch_grid[1]):
            y_start = i * patch_size[0]
            y_end = min((i + 1) * patch_size[0], imsize[0])
            x_start = j * patch_size[1]
            x_end = min((j + 1) * patch_size[1], imsize[1])
            points_in_patch = ((sfm_points_camera[0, :] >= x_start) & (sfm_points_camera[0, :] < x_end)
                               & (sfm_points_camera[1, :] >= y_start) & (sfm_points_camera[1, :] < y_end))
            if points_in_patch.sum().item() > threshold:
                mask[y_start:y_end, x_start:x_end] = False
    return mask


# ---- B8: depth_prediction/points_from_depth.py:192-212 ----------------------
def depth_gradient_mask(depth: Tensor, gradient_threshold: float) -> Tensor:
    depth_dx = torch.abs(depth[:, 1:] - depth[:, :-1])
    depth_dy = torch.abs(depth[1:, :] - depth[:-1, :])
    depth_grad_both = torch.zeros_like(depth, dtype=depth.dtype)
    depth_grad_both[:, 1:] += depth_dx
    depth_grad_both[1:, :] += depth_dy
    depth_grad_both = depth_grad_both - depth_grad_both.min()
    depth_grad_both = depth_grad_both / (depth_grad_both.max() + 1e-8)
    return depth_grad_both <= gradient_threshold


# ---- B1: depth_prediction/points_from_depth.py:111-180 ----------------------
def project_and_filter_sfm_pts(sfm_points: Tensor, P: Tensor, imsize, pred_mask: Tensor):
    """imsize = (W, H) as in the reference (`predicted_depth.depth.T.shape`).
    Returns coords int64 [2,M'] (row 0 = x), depth [M']."""
    cam = P @ torch.vstack([sfm_points.T, torch.ones(sfm_points.shape[0])])
    sfm_points_depth = cam[2]
    cam = cam[:2] / cam[2]
    cam = torch.round(cam).to(int)
    valid = torch.logical_and(torch.logical_and(cam[0] >= 0, cam[0] < imsize[0]),
                              torch.logical_and(cam[1] >= 0, cam[1] < imsize[1]))
    valid = torch.logical_and(valid, sfm_points_depth >= 0)
    if torch.sum(valid) < cam.shape[1] / 4:
        raise LowDepthAlignmentConfidenceError("Less than 1/4 of SFM points reprojected into image bounds.")
    cam[:, ~valid] = torch.zeros_like(cam[:, ~valid])
    valid = torch.logical_and(valid, pred_mask[cam[1], cam[0]])
    return cam[:, valid], sfm_points_depth[valid]


# ---- B9: depth_prediction/points_from_depth.py:270-312 ----------------------
def assemble_mask_and_unproject(aligned_depth: Tensor, mask: Tensor, subsampling_mask: Tensor,
                                K: Tensor, cam2world: Tensor, sfm_coords: Optional[Tensor] = None,
                                depth_grad_mask_thresh: Optional[float] = None,
                                use_num_sfm_points_mask: bool = True,
                                num_patches_small_axis: int = 20, threshold: int = 15):
    """Mask assembly (270-290) + unprojection (292-312). Returns (pts_world [n,3], mask [H*W])."""
    H, W = aligned_depth.shape
    imsize = (W, H)
    mask = (mask & (aligned_depth >= 0)).flatten()
    if depth_grad_mask_thresh is not None:
        mask &= depth_gradient_mask(aligned_depth, depth_grad_mask_thresh).flatten()
    if use_num_sfm_points_mask:
        mask &= num_sfm_points_mask(sfm_coords, (imsize[1], imsize[0]), num_patches_small_axis,
                                    threshold).flatten()
    mask = mask & subsampling_mask
    pts_camera = torch.dstack([
        torch.from_numpy(np.mgrid[0:imsize[0], 0:imsize[1]].T), aligned_depth]).reshape(-1, 3)[mask]
    pts_camera[:, 0] = (pts_camera[:, 0] + 0.5) * pts_camera[:, 2]
    pts_camera[:, 1] = (pts_camera[:, 1] + 0.5) * pts_camera[:, 2]
    dense_world = torch.linalg.inv(K) @ pts_camera.reshape((-1, 3)).T
    dense_world = (cam2world @ torch.vstack([dense_world, torch.ones(dense_world.shape[1])]))[:3].T
    return dense_world.reshape([-1, 3]).float(), mask


# ---- A9 helpers: utils/runner_utils.py:142-151 -------------------------------
def rgb_to_sh(rgb: Tensor) -> Tensor:
    return (rgb - 0.5) / 0.28209479177387814


def knn_dists(x: Tensor, K: int = 4) -> Tensor:
    """Brute-force equivalent of sklearn NearestNeighbors(K).kneighbors distances
    (self included at distance 0), utils/runner_utils.py:142-146."""
    d = torch.cdist(x.double(), x.double())
    return torch.sort(d, dim=1).values[:, :K].to(x.dtype)


# ---- B10: depth_prediction/predictors/metric3d.py:42-83, 96-131 --------------
# PARITY UNPINNED: cv2 is absent, so the keep-ratio resize (cv2.INTER_LINEAR on
# uint8, metric3d.py:50-52) is restated as bilinear interpolation at half-pixel
# centres rounded to uint8; OpenCV's fixed-point weights can differ by 1 LSB.
def metric3d_preprocess(img: Tensor, input_size=(616, 1064)):
    """img float [H,W,3] in [0,1] -> (net input [1,3,616,1064], pad_info, scale)."""
    import torch.nn.functional as F
    rgb_origin = (img * 255.0).numpy().astype(np.uint8)[:, :, ::-1]            # :44
    h, w = rgb_origin.shape[:2]
    scale = min(input_size[0] / h, input_size[1] / w)                           # :49
    rh, rw = int(h * scale), int(w * scale)
    src = torch.from_numpy(rgb_origin.copy()).float().permute(2, 0, 1)[None]
    rgb = F.interpolate(src, size=(rh, rw), mode="bilinear", align_corners=False)[0]
    rgb = torch.clamp(torch.round(rgb), 0, 255)                                  # uint8 result
    pad_h, pad_w = input_size[0] - rh, input_size[1] - rw
    pad_info = [pad_h // 2, pad_h - pad_h // 2, pad_w // 2, pad_w - pad_w // 2]
    border = torch.tensor([124.0, 116.0, 104.0])                                 # saturate_cast<uchar>
    out = border[:, None, None].repeat(1, *input_size)
    out[:, pad_info[0]:pad_info[0] + rh, pad_info[2]:pad_info[2] + rw] = rgb
    mean = torch.tensor([123.675, 116.28, 103.53])[:, None, None]               # :79-83
    std = torch.tensor([58.395, 57.12, 57.375])[:, None, None]
    return ((out - mean) / std)[None], pad_info, scale


def metric3d_to_og_size(t: Tensor, pad_info, size):
    """metric3d.py:96-118 for a [h,w] map."""
    import torch.nn.functional as F
    t = t[pad_info[0]: t.shape[0] - pad_info[1], pad_info[2]: t.shape[1] - pad_info[3]]
    return F.interpolate(t[None, None], size, mode="bilinear").squeeze()


def metric3d_postprocess_depth(pred_depth: Tensor, pad_info, size, fx: float, scale: float):
    d = metric3d_to_og_size(pred_depth, pad_info, size)
    return torch.clamp(d * (fx * scale / 1000.0), 0, 300)                        # :127-131
