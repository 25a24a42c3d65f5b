"""Patch-density mask (/root/reference/gs_init_compare/depth_subsampling/
num_sfm_points_mask.py:7-64): patches that already hold more than `threshold`
SfM points get no seeds. The reference loops over 720 patches in Python; here
it is a 2-D histogram kernel plus a per-pixel lookup."""
import numpy as np
import torch

from .._lib import call, ptr
from .config import NumSfMPointsMaskConfig


def calculate_patch_sizes(image_shape, num_patches_small_axis):
    """num_sfm_points_mask.py:7-35 (pure integer arithmetic, kept on the host)."""
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


def num_sfm_points_mask(sfm_points_camera: torch.Tensor, imsize, sfm_pts_mask_config:
                        NumSfMPointsMaskConfig) -> torch.Tensor:
    """sfm_points_camera int64 [2,M] (row 0 = x) on the device; imsize = (H, W).
    Returns bool [H,W] on the same device."""
    H, W = int(imsize[0]), int(imsize[1])
    (ph, pw), (gh, gw) = calculate_patch_sizes((H, W), sfm_pts_mask_config.num_patches_small_axis)
    dev = sfm_points_camera.device
    coords = sfm_points_camera.contiguous().long()
    counts = torch.empty(gh * gw, dtype=torch.int32, device=dev)
    mask = torch.empty(H, W, dtype=torch.bool, device=dev)
    call("gsr_sfm_patch_mask", H, W, coords.shape[1], ptr(coords), ph, pw, gh, gw,
         int(sfm_pts_mask_config.threshold), ptr(counts), ptr(mask),
         torch.cuda.current_stream().cuda_stream)
    return mask
