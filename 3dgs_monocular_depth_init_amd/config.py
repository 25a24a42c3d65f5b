"""The configuration fields the hot path reads, mirror of
/root/reference/gs_init_compare/config.py:20-66 (MonocularDepthInitConfig) and
the rasterisation-related fields of Config (config.py:103-149)."""
from dataclasses import dataclass, field
from enum import Enum
from typing import Literal, Optional, Union

from .depth_alignment.config import DepthAlignmentConfig
from .depth_subsampling.config import AdaptiveSubsamplingConfig, NumSfMPointsMaskConfig
from .point_cloud_postprocess.config import PointCloudPostprocessConfig


class Metric3dBackbone(str, Enum):          # depth_prediction/configs.py:33-36
    vits = "vits"
    vitl = "vitl"
    vitg = "vitg"


@dataclass
class Metric3dV2Config:                     # depth_prediction/configs.py:39-46
    backbone: Metric3dBackbone = Metric3dBackbone.vitl


@dataclass
class MonocularDepthInitConfig:
    predictor: Optional[str] = "metric3d"
    alignment: DepthAlignmentConfig = field(default_factory=DepthAlignmentConfig)
    depth_grad_mask_thresh: Optional[float] = None
    include_sfm_points: bool = True
    subsample_factor: Union[int, Literal["adaptive"]] = 10
    adaptive_subsampling: AdaptiveSubsamplingConfig = field(default_factory=AdaptiveSubsamplingConfig)
    use_num_sfm_points_mask: bool = True
    num_sfm_points_mask: NumSfMPointsMaskConfig = field(default_factory=NumSfMPointsMaskConfig)
    postprocess: PointCloudPostprocessConfig = field(default_factory=PointCloudPostprocessConfig)
    limit_init_scale: bool = False
    init_scale_clamp_quantile: float = 0.75
    noise_std_scene_frac: Optional[float] = None
    metric3d: Metric3dV2Config = field(default_factory=Metric3dV2Config)
    ignore_cache: bool = False
    cache_dir: Optional[str] = "__mono_depth_cache__"     # None (an extra): no depth cache


@dataclass
class Config:
    mdi: MonocularDepthInitConfig = field(default_factory=MonocularDepthInitConfig)
    init_scale: float = 1.0
    init_opa: float = 0.1
    sh_degree: int = 3
    near_plane: float = 0.01
    far_plane: float = 1e10
    packed: bool = False
    antialiased: bool = False
    batch_size: int = 1
    ssim_lambda: float = 0.2
