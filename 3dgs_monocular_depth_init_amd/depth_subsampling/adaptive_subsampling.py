"""AdaptiveDepthSubsampler (/root/reference/gs_init_compare/depth_subsampling/
adaptive_subsampling.py:82-122): depth-dependent stride, near pixels get the
larger stride. The IQR-clipped range (two quantiles of the valid depths) is
computed with torch-ROCm ops and stays on the device; the per-pixel factor and
the modulo test run in one kernel with the reference's fp32 operation order."""
from dataclasses import dataclass, field

import torch

from .._lib import call, ptr
from .config import AdaptiveSubsamplingConfig
from .interface import DepthSubsampler


def iqr_outlier_bounds(data: torch.Tensor):          # adaptive_subsampling.py:82-86
    q = _quantiles(data, (0.25, 0.75))
    iqr = q[1] - q[0]
    return q[0] - 1.5 * iqr, q[1] + 1.5 * iqr


def _quantiles(data: torch.Tensor, qs):
    """torch.quantile (linear interpolation) without its 16M-element input limit."""
    srt = torch.sort(data.reshape(-1)).values
    n = srt.numel()
    out = []
    for q in qs:
        pos = q * (n - 1)
        lo = int(pos // 1)
        hi = min(lo + 1, n - 1)
        w = torch.tensor(pos - lo, dtype=srt.dtype, device=srt.device)
        out.append(torch.lerp(srt[lo], srt[hi], w))
    return out


def depth_range(depth: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
    """input_range of get_depth_multipler_map (adaptive_subsampling.py:89-95) -> device [2]."""
    masked = depth[mask]
    lo_b, hi_b = iqr_outlier_bounds(masked)
    return torch.stack([torch.maximum(masked.min(), lo_b), torch.minimum(masked.max(), hi_b)]).float()


@dataclass
class AdaptiveDepthSubsampler(DepthSubsampler):
    config: AdaptiveSubsamplingConfig = field(default_factory=AdaptiveSubsamplingConfig)

    def get_mask(self, rgb, depth, mask):
        H, W = depth.shape
        depth = depth.contiguous().float()
        mask = mask.contiguous()
        rng = depth_range(depth, mask).contiguous()
        keep = torch.empty(H * W, dtype=torch.bool, device=depth.device)
        call("gsr_subsample_mask", H, W, 1, 0, ptr(depth), ptr(mask), ptr(rng),
             int(self.config.factor_range_min), int(self.config.factor_range_max), ptr(keep),
             torch.cuda.current_stream().cuda_stream)
        return keep
