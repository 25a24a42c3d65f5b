"""depth_alignment/interface.py:14-39 of the reference."""
import abc
from pathlib import Path
from typing import NamedTuple, Optional

import torch


class DepthAlignmentResult(NamedTuple):
    aligned_depth: torch.Tensor
    mask: torch.Tensor


class DepthAlignmentStrategy(abc.ABC):
    @classmethod
    @abc.abstractmethod
    def align(cls, predicted_depth, sfm_points_camera_coords: torch.Tensor,
              sfm_points_depth: torch.Tensor, config, debug_export_dir: Optional[Path] = None,
              ) -> DepthAlignmentResult:
        """predicted_depth: PredictedDepth (depth [H,W], mask [H,W]);
        sfm_points_camera_coords: int64 [2,M], row 0 = x, row 1 = y (every
        indexing site of the reference uses [coords[1], coords[0]] = [y, x]);
        sfm_points_depth: [M]."""
