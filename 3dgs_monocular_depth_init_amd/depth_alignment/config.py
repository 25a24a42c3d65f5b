"""depth_alignment/config.py:6-142 of the reference, restricted to the built strategies."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Optional


class DepthAlignmentStrategyEnum(str, Enum):
    lstsqrs = "lstsqrs"
    ransac = "ransac"
    msac = "msac"

    def get_implementation(self):
        from .alignment.lstsqrs import DepthAlignmentLstSqrs
        from .alignment.ransacs import DepthAlignmentMsac, DepthAlignmentRansac
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
class DepthAlignmentConfig:         # config.py:133-142
    segmenter: Optional[str] = None          # segmentation is out of scope: must stay None
    aligner: DepthAlignmentStrategyEnum = DepthAlignmentStrategyEnum.ransac
    ransac: RansacConfig = field(default_factory=RansacConfig)
