"""Depth alignment: mirror of
/root/reference/gs_init_compare/depth_alignment/ for the strategies on the hot
path (lstsqrs, ransac, msac) and the pipeline's no-segmentation branch. The
interp / SLIC / SAM / region-merging parts are out of scope."""
from .config import DepthAlignmentConfig, DepthAlignmentStrategyEnum, RansacConfig  # noqa: F401
from .exceptions import LowDepthAlignmentConfidenceError  # noqa: F401
from .interface import DepthAlignmentResult, DepthAlignmentStrategy  # noqa: F401
from .pipeline import DepthAlignmentPipeline  # noqa: F401
