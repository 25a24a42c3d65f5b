"""depth_alignment/config.py:6-142 of the reference. The two segmenters it names (SLIC: scikit-image,
SAM: segment_anything + a ViT-H checkpoint) are third-party packages absent from this build;
`DepthAlignmentPipeline` takes any callable with their signature instead."""
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


_SEGMENTERS = {}


def register_segmenter(name: str, fn) -> None:
    """Supply the implementation behind `segmenter="slic"` / `"sam"`: a callable
    fn(predicted_depth, checkpoint_dir, segmentation_config) -> integer label map [H, W]
    (the reference's DepthSegmentationFn, depth_alignment/interface.py:44-46)."""
    _SEGMENTERS[DepthSegmentationStrategyEnum(name).value] = fn


class DepthSegmentationStrategyEnum(str, Enum):      # config.py:36-51
    slic = "slic"
    sam = "sam"

    def get_implementation(self):
        if self.value in _SEGMENTERS:
            return _SEGMENTERS[self.value]
        raise NotImplementedError(
            f"segmenter {self.value!r} is a third-party algorithm (scikit-image SLIC / segment_anything) that "
            "this build does not contain: supply one with depth_alignment.config.register_segmenter(name, fn), "
            "fn(predicted_depth, checkpoint_dir, segmentation_config) -> label map")


@dataclass
class SAMSegmentationconfig:        # config.py:54-75
    use_normals: bool = True
    degenerate_mask_thresh: float = 0.9
    expansion_radius: int = 4
    tiny_region_area_fraction: float = 1e-4


@dataclass
class SLICSegmentationConfig:       # config.py:78-81
    compactness = 0.01
    num_regions = 40


@dataclass
class DepthSegmentationConfig:      # config.py:84-101
    region_margin: int = 10
    propagate_mask: bool = False
    min_border_grad_threshold: float = 0.0005
    min_sfm_pts_in_region: int = 5
    sam: SAMSegmentationconfig = field(default_factory=SAMSegmentationconfig)
    slic: SLICSegmentationConfig = field(default_factory=SLICSegmentationConfig)


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
    kernel: str = "thin_plate_spline"     # linear | thin_plate_spline | cubic | quintic (the scale-invariant kernels: the call passes
    # no `epsilon`, without which torchrbf refuses the others and the aligner falls back to the median scale)
    max_rbf_points: int = 5000


@dataclass
class DepthAlignmentConfig:         # config.py:133-142
    segmenter: Optional[DepthSegmentationStrategyEnum] = None
    aligner: DepthAlignmentStrategyEnum = DepthAlignmentStrategyEnum.ransac
    segmentation: DepthSegmentationConfig = field(default_factory=DepthSegmentationConfig)
    ransac: RansacConfig = field(default_factory=RansacConfig)
    interp: InterpConfig = field(default_factory=InterpConfig)
