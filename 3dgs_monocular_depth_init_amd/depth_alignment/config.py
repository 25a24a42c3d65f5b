"""depth_alignment/config.py:6-142 of the reference, restricted to the built strategies
(segmentation-based alignment -- SLIC / SAM -- is out of scope)."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Literal, Optional


class DepthAlignmentStrategyEnum(str, Enum):
    lstsqrs = "lstsqrs"
    ransac = "ransac"
    msac = "msac"
    interp = "interp"

    def get_implementation(self):
        from .alignment.lstsqrs import DepthAlignmentLstSqrs
        from .alignment.ransacs import DepthAlignmentMsac, DepthAlignmentRansac
        if self.value == "interp":
            from .alignment.interp import DepthAlignmentInterpolate
            return DepthAlignmentInterpolate
        return {"lstsqrs": DepthAlignmentLstSqrs, "ransac": DepthAlignmentRansac,
                "msac": DepthAlignmentMsac}[self.value]


@dataclass
class RansacConfig:                 # config.py:104-110
    inlier_threshold: float = 0.01
    max_iters: int = 2500
    confidence: float = 0.999
    sample_size: int = 4
    min_iters: int = 0


@dataclass
class InterpConfig:                 # config.py:113-130
    method: Literal["rbf", "linear"] = "linear"
    init: Optional[Literal["lstsqrs", "ransac"]] = "ransac"
    scale_outlier_removal: bool = True
    smoothing: float = 0.001
    kernel: str = "thin_plate_spline"
    max_rbf_points: int = 5000


@dataclass
class DepthAlignmentConfig:         # config.py:133-142
    segmenter: Optional[str] = None          # segmentation is out of scope: must stay None
    aligner: DepthAlignmentStrategyEnum = DepthAlignmentStrategyEnum.ransac
    ransac: RansacConfig = field(default_factory=RansacConfig)
    interp: InterpConfig = field(default_factory=InterpConfig)
