"""Closed-form scale/shift least squares on the device.

Mirror of /root/reference/gs_init_compare/depth_alignment/alignment/lstsqrs.py:9-54.
The reference materialises [M,2,2] outer products and calls torch.linalg.pinv
(several tiny kernels, launch-latency bound); here one reduction kernel
accumulates the five normal-equation sums in fp64 (gsr_lsq_sums), one solves
the 2x2 system with pinv semantics (gsr_solve_scale_shift) and one applies the
affine map (gsr_affine_depth); scale/shift never visit the host.
"""
import torch

from ..._lib import call, ptr
from ..interface import DepthAlignmentResult, DepthAlignmentStrategy


def _st():
    return torch.cuda.current_stream().cuda_stream


def gather_depth(depth: torch.Tensor, coords: torch.Tensor) -> torch.Tensor:
    """depth[coords[1], coords[0]] -> [M]."""
    M = coords.shape[1]
    out = torch.empty(M, dtype=torch.float32, device=depth.device)
    call("gsr_gather_depth", M, ptr(depth), depth.shape[1], ptr(coords), ptr(out), _st())
    return out


def align_depth_least_squares(depth: torch.Tensor, gt_depth: torch.Tensor):
    """depth: [2,N] (row 0 = predicted depth, row 1 = ones, as in the reference)
    or [N]; gt_depth [N]. Returns (scale, shift) as 0-dim device tensors."""
    d = (depth[0] if depth.dim() == 2 else depth).contiguous().float()
    g = gt_depth.contiguous().float()
    sums = torch.empty(5, dtype=torch.float64, device=d.device)
    h = torch.empty(2, dtype=torch.float32, device=d.device)
    call("gsr_lsq_sums", 1, d.numel(), 0, ptr(d), ptr(g), None, 0, None, 0.0, ptr(sums), _st())
    call("gsr_solve_scale_shift", 1, ptr(sums), ptr(h), _st())
    return h[0], h[1]


def apply_scale_shift(depth: torch.Tensor, h: torch.Tensor) -> torch.Tensor:
    out = torch.empty_like(depth)
    call("gsr_affine_depth", depth.numel(), ptr(depth), ptr(h), ptr(out), _st())
    return out


class DepthAlignmentLstSqrs(DepthAlignmentStrategy):
    @classmethod
    def align(cls, predicted_depth, sfm_points_camera_coords, sfm_points_depth, *args, **kwargs):
        depth = predicted_depth.depth.contiguous().float()
        coords = sfm_points_camera_coords.contiguous().long()
        d = gather_depth(depth, coords)
        scale, shift = align_depth_least_squares(d, sfm_points_depth)
        h = torch.stack([scale, shift])
        return DepthAlignmentResult(aligned_depth=apply_scale_shift(depth, h),
                                    mask=predicted_depth.mask)
