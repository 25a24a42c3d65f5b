"""depth_subsampling/interface.py:6-18 of the reference."""
import abc

import torch


class DepthSubsampler(abc.ABC):
    def get_mask(self, rgb: torch.Tensor, depth: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        """rgb [H,W,3], depth [H,W], mask bool [H,W] -> bool [H*W] sampling mask
        (False wherever `mask` is False)."""
