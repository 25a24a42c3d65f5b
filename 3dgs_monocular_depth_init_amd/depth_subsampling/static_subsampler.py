"""StaticDepthSubsampler (/root/reference/gs_init_compare/depth_subsampling/
static_subsampler.py:8-22): keep every k-th pixel in x and y. The reference
builds a 33 MB int64 cartesian_prod on the CPU per 1080p image (0.6 s); here
it is index arithmetic in one kernel."""
from dataclasses import dataclass

import torch

from .._lib import call, ptr
from .interface import DepthSubsampler


@dataclass
class StaticDepthSubsampler(DepthSubsampler):
    subsample_factor: int

    def get_mask(self, rgb, depth, mask):
        H, W = depth.shape
        mask = mask.contiguous()
        keep = torch.empty(H * W, dtype=torch.bool, device=depth.device)
        call("gsr_subsample_mask", H, W, 0, int(self.subsample_factor), None, ptr(mask), None, 0, 0,
             ptr(keep), torch.cuda.current_stream().cuda_stream)
        return keep
