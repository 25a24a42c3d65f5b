"""depth_subsampling/config.py:5-26 of the reference."""
from dataclasses import dataclass


@dataclass
class AdaptiveSubsamplingConfig:
    factor_range_min: int = 5
    factor_range_max: int = 15


@dataclass
class NumSfMPointsMaskConfig:
    num_patches_small_axis: int = 20
    threshold: int = 15
