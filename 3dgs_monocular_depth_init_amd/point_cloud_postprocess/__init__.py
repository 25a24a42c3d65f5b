from .config import OutlierRemovalMethod, PointCloudPostprocessConfig, PointCloudSubsamplingParams  # noqa: F401
from .postprocess import postprocess_point_cloud, subsample_pointcloud, subsample_pointcloud_device  # noqa: F401
