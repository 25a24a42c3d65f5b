"""Seed points from an aligned monocular depth map.

Mirror of /root/reference/gs_init_compare/depth_prediction/points_from_depth.py:
  get_valid_sfm_pts / project_and_filter_sfm_pts   111-180
  get_subsampler                                   183-189
  depth_gradient_mask                              192-212
  get_pts_from_depth                               215-329
The reference assembles masks with boolean indexing and two small matmuls; the
mask assembly, ordered stream compaction and unprojection are fused here into
a count kernel, a scan and an emit kernel (gsr_unproject_*).
"""
from pathlib import Path
from typing import Optional

import numpy as np
import torch

from .._lib import call, load, ptr
from ..depth_alignment.exceptions import LowDepthAlignmentConfidenceError
from ..depth_alignment.pipeline import DepthAlignmentPipeline
from ..depth_subsampling.adaptive_subsampling import AdaptiveDepthSubsampler
from ..depth_subsampling.num_sfm_points_mask import num_sfm_points_mask
from ..depth_subsampling.static_subsampler import StaticDepthSubsampler
from .predictors.depth_predictor_interface import PredictedDepth


def _st():
    return torch.cuda.current_stream().cuda_stream


def project_and_filter_sfm_pts(image, sfm_points: torch.Tensor, P: torch.Tensor, imsize,
                               predicted_depth: PredictedDepth, debug_export_dir=None):
    """points_from_depth.py:142-180 (+ get_valid_sfm_pts 111-139).
    imsize = (W, H). Returns (coords int64 [2,M'], depths [M']) on the device."""
    W, H = int(imsize[0]), int(imsize[1])
    dev = sfm_points.device
    M = sfm_points.shape[0]
    pts = sfm_points.contiguous().float()
    P = P.contiguous().float()
    mask = predicted_depth.mask.contiguous()
    coords = torch.empty(2, M, dtype=torch.int64, device=dev)
    depth = torch.empty(M, dtype=torch.float32, device=dev)
    inb = torch.empty(M, dtype=torch.bool, device=dev)
    valid = torch.empty(M, dtype=torch.bool, device=dev)
    call("gsr_project_sfm", M, ptr(pts), ptr(P), W, H, ptr(mask), ptr(coords), ptr(depth), ptr(inb),
         ptr(valid), _st())
    n_inb = int(inb.sum().item())
    print(f"Num invalid reprojected SfM points: {M - n_inb} out of {M}")
    if n_inb < M / 4:                                      # points_from_depth.py:124-129
        raise LowDepthAlignmentConfidenceError(
            "Less than 1/4 of SFM points", f" ({n_inb} / {M})", " reprojected into image bounds.")
    return coords[:, valid], depth[valid]


def get_subsampler(cfg):
    """points_from_depth.py:183-189."""
    if cfg.mdi.subsample_factor == "adaptive":
        return AdaptiveDepthSubsampler(cfg.mdi.adaptive_subsampling)
    elif isinstance(cfg.mdi.subsample_factor, int):
        return StaticDepthSubsampler(cfg.mdi.subsample_factor)
    raise ValueError(f"Unsupported subsampling factor: {cfg.mdi.subsample_factor}")


def depth_gradient_mask(depth: torch.Tensor, gradient_threshold: float) -> torch.Tensor:
    """points_from_depth.py:192-212: |dx|+|dy| (kernel), min-max normalised, <= threshold."""
    H, W = depth.shape
    depth = depth.contiguous().float()
    grad = torch.empty_like(depth)
    call("gsr_depth_grad", H, W, ptr(depth), ptr(grad), _st())
    grad = grad - grad.min()
    grad = grad / (grad.max() + 1e-8)
    return grad <= gradient_threshold


def unproject_masked(aligned_depth: torch.Tensor, valid: torch.Tensor, subsample: torch.Tensor,
                     extra: Optional[torch.Tensor], rgb: Optional[torch.Tensor], K: torch.Tensor,
                     cam2world: torch.Tensor):
    """Fused points_from_depth.py:270 (`depth >= 0`), 290 (mask & subsampling) and
    292-312 (unprojection). Returns (pts_world [n,3], rgbs [n,3] or None, mask [H*W])."""
    H, W = aligned_depth.shape
    dev = aligned_depth.device
    lib = load()
    nb = lib.gsr_unproject_num_blocks(H, W)
    depth = aligned_depth.contiguous().float()
    valid = valid.contiguous()
    subsample = subsample.contiguous()
    extra = extra.contiguous() if extra is not None else None
    counts = torch.empty(nb, dtype=torch.int32, device=dev)
    offsets = torch.empty(nb + 1, dtype=torch.int32, device=dev)
    call("gsr_unproject_count", H, W, ptr(depth), ptr(valid), ptr(subsample), ptr(extra),
         ptr(counts), _st())
    call("gsr_isect_scan", nb, ptr(counts), ptr(offsets), None, _st())
    n = int(offsets[-1].item())
    pts = torch.empty(max(n, 1), 3, dtype=torch.float32, device=dev)
    rgbs = torch.empty(max(n, 1), 3, dtype=torch.float32, device=dev) if rgb is not None else None
    final_mask = torch.empty(H * W, dtype=torch.bool, device=dev)
    Kinv = torch.linalg.inv(K.to(dev).float()).contiguous()
    c2w = cam2world.to(dev).float().contiguous()
    rgb_c = rgb.contiguous().float() if rgb is not None else None
    call("gsr_unproject_emit", H, W, ptr(depth), ptr(valid), ptr(subsample), ptr(extra),
         ptr(rgb_c), ptr(Kinv), ptr(c2w), ptr(offsets), ptr(pts), ptr(rgbs), ptr(final_mask),
         _st())
    return pts[:n], (rgbs[:n] if rgbs is not None else None), final_mask


def _sfm_points_of(parser, image_name: str, device) -> torch.Tensor:
    """points_from_depth.py:233-237: `parser.points[parser.point_indices[image.name]]` for any
    object with those two attributes (the reference's colmap Parser or its nerfbaselines
    gs_Parser); a bare [M,3] tensor / array is taken as that image's points already."""
    if hasattr(parser, "points") and hasattr(parser, "point_indices"):
        pts = parser.points[parser.point_indices[image_name]]
    else:
        pts = parser
    if not isinstance(pts, torch.Tensor):
        pts = torch.from_numpy(np.ascontiguousarray(pts))
    return pts.to(device).float()


def get_pts_from_depth(predicted_depth: PredictedDepth, image, parser, config,
                       device: str, debug_export_dir: Optional[Path] = None, return_rgb: bool = False):
    """points_from_depth.py:215-329, same positional arguments: `parser` is the dataset parser
    (`.points` [P,3], `.point_indices[image.name]`), or -- an extra -- the image's SfM points
    [M,3] themselves. Returns (pts_world [n,3] on device, mask [H*W] on CPU, P [3,4])
    (+ rgbs [n,3] when return_rgb). The debug exports of the reference (matplotlib / PLY
    files) are not produced; `debug_export_dir` is accepted and ignored."""
    imsize = predicted_depth.depth.T.shape                       # (W, H)
    R = image.cam2world[:3, :3].T
    C = image.cam2world[:3, 3]
    P = image.K @ R @ torch.hstack([torch.eye(3), -C[:, None].cpu()]).to(image.K.device)
    sfm_points = _sfm_points_of(parser, image.name, device)
    cam2world = image.cam2world.to(device).float()
    P = P.to(device).float()
    K = image.K.to(device).float()

    sfm_points_camera, sfm_points_depth = project_and_filter_sfm_pts(
        image.data, sfm_points, P, imsize, predicted_depth, debug_export_dir)
    aligned_depth, mask = DepthAlignmentPipeline.from_config(config).align(
        image, predicted_depth, sfm_points_camera, sfm_points_depth, config, debug_export_dir)
    subsampling_mask = get_subsampler(config).get_mask(image.data, aligned_depth, mask)

    extra = None
    if config.mdi.depth_grad_mask_thresh is not None:
        extra = depth_gradient_mask(aligned_depth, config.mdi.depth_grad_mask_thresh).flatten()
    if config.mdi.use_num_sfm_points_mask:
        m = num_sfm_points_mask(sfm_points_camera, (imsize[1], imsize[0]),
                                config.mdi.num_sfm_points_mask).flatten()
        extra = m if extra is None else (extra & m)
    rgb = image.data.to(device) if return_rgb else None
    pts_world, rgbs, final_mask = unproject_masked(aligned_depth, mask, subsampling_mask, extra,
                                                   rgb, K, cam2world)
    out = (pts_world.reshape([-1, 3]).float(), final_mask.cpu(), P)
    return out + (rgbs,) if return_rgb else out
