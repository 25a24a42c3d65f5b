"""Depth subsampling: mirror of
/root/reference/gs_init_compare/depth_subsampling/."""
from .adaptive_subsampling import AdaptiveDepthSubsampler  # noqa: F401
from .config import AdaptiveSubsamplingConfig, NumSfMPointsMaskConfig  # noqa: F401
from .interface import DepthSubsampler  # noqa: F401
from .num_sfm_points_mask import calculate_patch_sizes, num_sfm_points_mask  # noqa: F401
from .static_subsampler import StaticDepthSubsampler  # noqa: F401
