"""Top level of the monocular-depth initialisation.

Mirror of /root/reference/gs_init_compare/monocular_depth_init.py:95-224
(`pts_and_rgb_from_monocular_depth`) around the device kernels. The depth
network itself (B10: Metric3D via torch.hub, remote weights) is pluggable: any
object with `predict_depth(img [H,W,3], CameraIntrinsics) -> PredictedDepth`;
dataset parsing is replaced by an iterable of frames.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from typing import Iterable, List, Optional

import torch

from .depth_alignment.exceptions import LowDepthAlignmentConfidenceError
from .depth_prediction.points_from_depth import get_pts_from_depth
from .depth_prediction.predictors.depth_predictor_interface import CameraIntrinsics
from .types import InputImage

_LOGGER = logging.getLogger(__name__)


@dataclass
class Frame:
    """What the reference reads per training image (datasets/colmap.py:381-412 +
    parser.points[parser.point_indices[name]])."""
    image: torch.Tensor            # [H,W,3] 0-255
    image_name: str
    camtoworld: torch.Tensor       # [4,4]
    K: torch.Tensor                # [3,3]
    sfm_points: torch.Tensor       # [M,3] SfM points visible in this image


def add_noise_to_point_cloud(pts: torch.Tensor, noise_std: float):      # monocular_depth_init.py:90-92
    return pts + torch.randn_like(pts) * noise_std


@torch.no_grad()
def pts_and_rgb_from_monocular_depth(config, frames: Iterable[Frame], model, device: str = "cuda",
                                     sfm_points: Optional[torch.Tensor] = None,
                                     sfm_points_rgb: Optional[torch.Tensor] = None,
                                     scene_scale: float = 1.0):
    """monocular_depth_init.py:95-224. Returns (pts [n,3], rgbs [n,3], scales or None)."""
    points_list: List[torch.Tensor] = []
    rgbs_list: List[torch.Tensor] = []
    for data in frames:
        assert data.image.max() > 1                                     # :122 images are 0-255
        image = InputImage(name=data.image_name, cam2world=data.camtoworld, K=data.K,
                           data=data.image / 255.0)
        predicted_depth = model.predict_depth(image.data, CameraIntrinsics(image.K))
        assert predicted_depth.depth.device == torch.device(device)     # :140
        try:
            points, subsampling_mask, P, rgbs = get_pts_from_depth(
                predicted_depth, image, data.sfm_points, config, device, None, return_rgb=True)
        except LowDepthAlignmentConfidenceError as e:                   # :157-161
            _LOGGER.warning("Low depth alignment confidence for image %s: {%s}", image.name, e)
            continue
        if config.mdi.noise_std_scene_frac is not None:                 # :163-166
            points = add_noise_to_point_cloud(points, scene_scale * config.mdi.noise_std_scene_frac)
        points_list.append(points)
        rgbs_list.append(rgbs.float())
    if config.mdi.include_sfm_points and sfm_points is not None:        # :179-181
        points_list.append(sfm_points.float().to(device))
        rgbs_list.append((sfm_points_rgb / 255.0).float().to(device))
    pts = torch.cat(points_list, dim=0).float()
    rgbs = torch.cat(rgbs_list, dim=0).float()
    scales = None
    if config.mdi.limit_init_scale:                                     # :215-223
        from .knn import knn
        dist2_avg = (knn(pts, 4)[:, 1:] ** 2).mean(dim=-1)
        dist_avg = torch.sqrt(dist2_avg)
        quantile = torch.quantile(dist_avg, config.mdi.init_scale_clamp_quantile)
        dist_avg = torch.clamp(dist_avg, max=quantile)
        scales = torch.log(dist_avg * config.init_scale).unsqueeze(-1).repeat(1, 3)
    return pts, rgbs, scales
