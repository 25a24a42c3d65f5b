"""Top level of the monocular-depth initialisation.

Mirror of /root/reference/gs_init_compare/monocular_depth_init.py:
  pick_model                              32-57
  predict_depth_or_get_cached_depth       60-87
  add_noise_to_point_cloud                90-92
  pts_and_rgb_from_monocular_depth        95-224   (config, parser, device)
around the device kernels. `parser` is duck-typed exactly as the reference uses it:
`.dataset_name`, `.points`, `.points_rgb`, `.point_indices`, `.scene_scale` and
`type(parser).DatasetCls(parser, split="train")` yielding dicts with `image` (0-255),
`image_name`, `camtoworld`, `K` (datasets/colmap.py:381-412). Dataset parsing itself,
the PLY debug exports and the (default-off) post-processing are out of scope.
"""
from __future__ import annotations

import logging
from dataclasses import dataclass
from pathlib import Path
from typing import Callable, Iterable, List, Optional

import numpy as np
import torch

from . import io as gs_io
from .depth_alignment.exceptions import LowDepthAlignmentConfidenceError
from .depth_prediction.points_from_depth import get_pts_from_depth
from .depth_prediction.predictors.depth_predictor_interface import CameraIntrinsics, DepthPredictor
from .types import InputImage

_LOGGER = logging.getLogger(__name__)

# predictor name -> factory(config, device) -> DepthPredictor. The reference hard-codes five
# classes (monocular_depth_init.py:36-57); only Metric3D is on the path, the
# others can be registered by the embedding application.
_PREDICTORS = {}


def register_predictor(name: str, factory: Callable) -> None:
    _PREDICTORS[name] = factory


def pick_model(config) -> Callable:
    """monocular_depth_init.py:32-57."""
    if config.mdi.predictor is None:
        raise ValueError("No depth predictor model specified in config.")
    if config.mdi.predictor in _PREDICTORS:
        return _PREDICTORS[config.mdi.predictor]
    if config.mdi.predictor == "metric3d":
        from .depth_prediction.predictors.metric3d import Metric3d
        return Metric3d
    raise ValueError(f"Unsupported monodepth model: {config.mdi.predictor}")


def predict_depth_or_get_cached_depth(model: DepthPredictor, image: torch.Tensor,
                                      intrinsics: CameraIntrinsics, image_name: str, config,
                                      dataset_name: str, device=None):
    """monocular_depth_init.py:60-87; same directory layout and file name
    (`cache_dir/model.name/dataset/{image_name}.pth`). The payload is a weights-only-safe
    dict of the PredictedDepth fields (see io.save_predicted_depth), not a pickled object.

    A cache hit is loaded onto `device` (an extra argument; default: the predictor's own
    device, else the image's). The reference's `torch.load` restores the tensors to the cuda
    device they were predicted on, whereas the dataset hands over CPU images
    (datasets/colmap.py:384): loading onto `image.device` would return a CPU depth on every
    second run and trip the caller's device assertion (monocular_depth_init.py:140)."""
    if device is None:
        device = getattr(model, "device", None) or image.device
    cache_path = gs_io.depth_cache_path(config.mdi.cache_dir, model.name, dataset_name, image_name)
    cache_path.parent.mkdir(exist_ok=True, parents=True)
    depth = None
    if not config.mdi.ignore_cache and cache_path.exists():
        try:
            depth = gs_io.load_predicted_depth(cache_path, device=device)
        except Exception as e:  # noqa: BLE001  (reference: any failure -> recompute)
            _LOGGER.warning("Failed to load cached depth for image %s: %s", image_name, e)
    if depth is None:
        depth = model.predict_depth(image, intrinsics)
        try:
            gs_io.save_predicted_depth(depth, cache_path)
        except KeyboardInterrupt:
            cache_path.unlink(missing_ok=True)
            raise
    return depth


def _with_predicted_depths(entries, model, device, get_image, get_K, cached=None):
    """Yield (entry, PredictedDepth) over `entries`, `model.batch_size` images per network call when the predictor
    has `predict_depths` (Metric3d: one batched encoder pass + parallel decoder branches), else one at a time --
    the reference's loop (monocular_depth_init.py:120-147) predicts every training image one by one, but nothing
    in it depends on that. `cached(entry)` may return a stored PredictedDepth (or None)."""
    bs = max(1, int(getattr(model, "batch_size", 1))) if hasattr(model, "predict_depths") else 1
    group = []

    def flush():
        todo = [(i, e) for i, (e, d) in enumerate(group) if d is None]
        if todo:
            if bs > 1:
                preds = model.predict_depths([get_image(e) for _, e in todo], [CameraIntrinsics(get_K(e)) for _, e in todo])
            else:
                preds = [model.predict_depth(get_image(e), CameraIntrinsics(get_K(e))) for _, e in todo]
            for (i, e), pd in zip(todo, preds):
                group[i] = (e, pd)
        out = list(group)
        group.clear()
        return out

    for e in entries:
        group.append((e, cached(e) if cached is not None else None))
        if sum(1 for _, d in group if d is None) >= bs:
            yield from flush()
    yield from flush()


def add_noise_to_point_cloud(pts: torch.Tensor, noise_std: float):      # monocular_depth_init.py:90-92
    return pts + torch.randn_like(pts) * noise_std


def _finish(config, points_list, rgbs_list, device, cams=None):
    pts = torch.cat(points_list, dim=0).float()
    rgbs = torch.cat(rgbs_list, dim=0).float()
    pp = getattr(config.mdi, "postprocess", None)
    if pp is not None and cams is not None and (pp.subsample or pp.outlier_removal.value != "none"):
        from .point_cloud_postprocess.postprocess import postprocess_point_cloud      # :187-196
        pts, rgbs = postprocess_point_cloud(pts, rgbs, cams[0], cams[1], cams[2], pp, device)
    scales = None
    if config.mdi.limit_init_scale:                                     # :215-223
        from .knn import knn
        dist2_avg = (knn(pts, 4)[:, 1:] ** 2).mean(dim=-1)
        dist_avg = torch.sqrt(dist2_avg)
        quantile = torch.quantile(dist_avg, config.mdi.init_scale_clamp_quantile)
        dist_avg = torch.clamp(dist_avg, max=quantile)
        scales = torch.log(dist_avg * config.init_scale).unsqueeze(-1).repeat(1, 3)
    return pts, rgbs, scales


@torch.no_grad()
def pts_and_rgb_from_monocular_depth(config, parser, device: str = "cuda", model=None):
    """monocular_depth_init.py:95-224, same positional arguments. `model=` (an extra) injects
    a ready DepthPredictor instead of `pick_model(config)(config, device)`.
    Returns (pts [n,3], rgbs [n,3], scales [n,3] or None)."""
    if model is None:
        model = pick_model(config)(config, device)
    _LOGGER.info("Using depth predictor model: %s", model.name)
    dataset_name = parser.dataset_name
    points_list: List[torch.Tensor] = []
    rgbs_list: List[torch.Tensor] = []
    dataset = type(parser).DatasetCls(parser, split="train")
    use_cache = getattr(config.mdi, "cache_dir", None) is not None
    intrinsic_matrices, proj_matrices, image_sizes = [], [], []
    cache_hits = set()

    def _cached(data):
        if not use_cache or config.mdi.ignore_cache:
            return None
        path = gs_io.depth_cache_path(config.mdi.cache_dir, model.name, dataset_name, data["image_name"])
        if not path.exists():
            return None
        try:
            depth = gs_io.load_predicted_depth(path, device=device)
            cache_hits.add(data["image_name"])
            return depth
        except Exception as e:  # noqa: BLE001  (reference: any failure -> recompute, :75-80)
            _LOGGER.warning("Failed to load cached depth for image %s: %s", data["image_name"], e)
            return None

    def _checked(dataset):
        for data in dataset:
            assert data["image"].max() > 1                              # :122 images are 0-255
            yield data

    for data, predicted_depth in _with_predicted_depths(_checked(dataset), model, device, lambda d: d["image"] / 255.0,
                                                        lambda d: d["K"], _cached):
        image = InputImage(name=data["image_name"], cam2world=data["camtoworld"], K=data["K"],
                           data=data["image"] / 255.0)
        if use_cache and image.name not in cache_hits:                  # :60-87: a fresh prediction is stored
            path = gs_io.depth_cache_path(config.mdi.cache_dir, model.name, dataset_name, image.name)
            path.parent.mkdir(exist_ok=True, parents=True)
            try:
                gs_io.save_predicted_depth(predicted_depth, path)
            except KeyboardInterrupt:
                path.unlink(missing_ok=True)
                raise
        assert predicted_depth.depth.device == torch.device(device)     # :140
        try:
            points, subsampling_mask, P, rgbs = get_pts_from_depth(
                predicted_depth, image, parser, config, device, None, return_rgb=True)
        except LowDepthAlignmentConfidenceError as e:                   # :157-161
            _LOGGER.warning("Low depth alignment confidence for image %s: {%s}", image.name, e)
            continue
        if config.mdi.noise_std_scene_frac is not None:                 # :163-166
            points = add_noise_to_point_cloud(
                points, parser.scene_scale * config.mdi.noise_std_scene_frac)
        points_list.append(points)
        rgbs_list.append(rgbs.float())
        intrinsic_matrices.append(image.K.cpu().numpy())                # :170-172
        proj_matrices.append(P.cpu().numpy())
        image_sizes.append(np.array(image.data.shape[:2][::-1], dtype=np.int32))
    if config.mdi.include_sfm_points:                                   # :179-181
        points_list.append(torch.from_numpy(np.asarray(parser.points)).float().to(device))
        rgbs_list.append(torch.from_numpy(np.asarray(parser.points_rgb) / 255.0).float().to(device))
    return _finish(config, points_list, rgbs_list, device, (intrinsic_matrices, proj_matrices, image_sizes))


# --------------------------------------------------------------------------------------------- #
# extra: the same loop over in-memory frames (synthetic scenes, benchmarks)
# --------------------------------------------------------------------------------------------- #
@dataclass
class Frame:
    """What the reference reads per training image (datasets/colmap.py:381-412 +
    parser.points[parser.point_indices[name]])."""
    image: torch.Tensor            # [H,W,3] 0-255
    image_name: str
    camtoworld: torch.Tensor       # [4,4]
    K: torch.Tensor                # [3,3]
    sfm_points: torch.Tensor       # [M,3] SfM points visible in this image


@torch.no_grad()
def pts_and_rgb_from_frames(config, frames: Iterable[Frame], model, device: str = "cuda",
                            sfm_points: Optional[torch.Tensor] = None,
                            sfm_points_rgb: Optional[torch.Tensor] = None,
                            scene_scale: float = 1.0):
    """The loop of pts_and_rgb_from_monocular_depth over in-memory frames, no depth cache."""
    points_list: List[torch.Tensor] = []
    rgbs_list: List[torch.Tensor] = []
    def _checked(frames):
        for data in frames:
            assert data.image.max() > 1
            yield data

    for data, predicted_depth in _with_predicted_depths(_checked(frames), model, device, lambda d: d.image / 255.0,
                                                        lambda d: d.K):
        image = InputImage(name=data.image_name, cam2world=data.camtoworld, K=data.K,
                           data=data.image / 255.0)
        assert predicted_depth.depth.device == torch.device(device)
        try:
            points, subsampling_mask, P, rgbs = get_pts_from_depth(
                predicted_depth, image, data.sfm_points, config, device, None, return_rgb=True)
        except LowDepthAlignmentConfidenceError as e:
            _LOGGER.warning("Low depth alignment confidence for image %s: {%s}", image.name, e)
            continue
        if config.mdi.noise_std_scene_frac is not None:
            points = add_noise_to_point_cloud(points, scene_scale * config.mdi.noise_std_scene_frac)
        points_list.append(points)
        rgbs_list.append(rgbs.float())
    if config.mdi.include_sfm_points and sfm_points is not None:
        points_list.append(sfm_points.float().to(device))
        rgbs_list.append((sfm_points_rgb / 255.0).float().to(device))
    return _finish(config, points_list, rgbs_list, device)
