"""Mirror of /root/reference/gs_init_compare/point_cloud_postprocess/config.py:8-21 and
native_modules/subsampling/src/pointcloud_subsampling/subsampling_params.py:4-17."""
from dataclasses import dataclass, field
from enum import Enum


@dataclass
class PointCloudSubsamplingParams:
    max_bbox_aspect_ratio: float = 1.1     # longest / shortest box side up to which a node may merge
    min_extent_multiplier: float = 1.0     # merge when the tight box fits this many mean extents


class OutlierRemovalMethod(str, Enum):
    off = "none"
    lof = "lof"


@dataclass
class PointCloudPostprocessConfig:
    outlier_removal: OutlierRemovalMethod = OutlierRemovalMethod.off
    lof_num_neighbors: int = 40
    subsample: bool = False
    subsample_params: PointCloudSubsamplingParams = field(default_factory=PointCloudSubsamplingParams)
