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
    """The fields of the reference's Config (config.py:69-202) that the path and the training loop
    around it read, with the reference's defaults, and `adjust_steps` (config.py:204-221)."""
    mdi: MonocularDepthInitConfig = field(default_factory=MonocularDepthInitConfig)
    batch_size: int = 1                    # config.py:103
    steps_scaler: float = 1.0              # :105
    max_steps: int = 30_000                # :108
    eval_steps: list = field(default_factory=lambda: [7_000, 30_000])    # :110
    save_steps: list = field(default_factory=lambda: [7_000, 30_000])    # :112
    save_final_ply: bool = True            # :113
    init_type: str = "sfm"                 # :116  sfm | random | monocular_depth
    init_num_pts: int = 100_000            # :121
    init_extent: float = 3.0               # :123
    sh_degree: int = 3                     # :125
    sh_degree_interval: int = 1000         # :127
    init_opa: float = 0.1                  # :129
    init_scale: float = 1.0                # :131
    ssim_lambda: float = 0.2               # :133
    near_plane: float = 0.01               # :136
    far_plane: float = 1e10                # :138
    strategy: object = None                # :141-143 DefaultStrategy | MCMCStrategy (None -> DefaultStrategy())
    packed: bool = False                   # :145
    sparse_grad: bool = False              # :147
    antialiased: bool = False              # :149
    random_background: bool = False        # :152
    opacity_reg: float = 0.0               # :155
    scale_reg: float = 0.0                 # :157
    depth_loss: bool = False               # :183
    depth_lambda: float = 1e-2             # :185
    absgrad: bool = False                  # (strategy.absgrad, runner.py:352-356)
    camera_model: str = "pinhole"          # :97
    # not in the reference: list a (tile, Gaussian) pair only when its alpha >= 1/255 ellipse reaches the
    # tile -- same image and gradients (runner.RasterConfig.tight_tiles)
    tight_tiles: bool = True

    def __post_init__(self):
        if self.strategy is None:
            from .strategy import DefaultStrategy
            self.strategy = DefaultStrategy()

    def adjust_steps(self, factor: float):
        """config.py:204-221, verbatim semantics (int() truncation, which fields scale)."""
        from .strategy import DefaultStrategy, MCMCStrategy
        self.eval_steps = [int(i * factor) for i in self.eval_steps]
        self.save_steps = [int(i * factor) for i in self.save_steps]
        self.max_steps = int(self.max_steps * factor)
        self.sh_degree_interval = int(self.sh_degree_interval * factor)
        strategy = self.strategy
        if isinstance(strategy, DefaultStrategy):
            strategy.refine_start_iter = int(strategy.refine_start_iter * factor)
            strategy.refine_stop_iter = int(strategy.refine_stop_iter * factor)
            strategy.reset_every = int(strategy.reset_every * factor)
            strategy.refine_every = int(strategy.refine_every * factor)
        elif isinstance(strategy, MCMCStrategy):
            strategy.refine_start_iter = int(strategy.refine_start_iter * factor)
            strategy.refine_stop_iter = int(strategy.refine_stop_iter * factor)
            strategy.refine_every = int(strategy.refine_every * factor)
        else:
            raise TypeError(f"unknown strategy {type(strategy).__name__}")
