"""Data contracts of the path, gathered in one module.

The field names and defaults are the interface the reference's callers rely on
(/root/reference/gs_init_compare/depth_prediction/predictors/depth_predictor_interface.py:9-39,
types.py:6-10, depth_subsampling/config.py:5-26); the modules of the same names in this
package re-export them so that the reference's import paths keep working.
"""
from __future__ import annotations

import abc
import dataclasses
import typing

import torch
from torch import Tensor


@dataclasses.dataclass
class PredictedDepth:
    """What a depth predictor hands to alignment: depth [H,W] float, mask [H,W] bool
    (valid pixels), and optional per-pixel extras some predictors produce."""
    depth: Tensor
    mask: Tensor
    depth_confidence: typing.Optional[Tensor] = None
    normal: typing.Optional[Tensor] = None
    normal_confidence: typing.Optional[Tensor] = None


def _k_entry(row: int, col: int):
    return property(lambda self: self.K[row, col].item())


class CameraIntrinsics(typing.NamedTuple):
    """Pinhole intrinsics as the 3x3 matrix K; fx / fy / cx / cy read its entries."""
    K: Tensor
    fx = _k_entry(0, 0)
    fy = _k_entry(1, 1)
    cx = _k_entry(0, 2)
    cy = _k_entry(1, 2)


class DepthPredictor(abc.ABC):
    """Predictor plug-in: constructed from (config, device), named, maps an image
    [H,W,3] in [0,1] and its intrinsics to a PredictedDepth."""

    @abc.abstractmethod
    def __init__(self, config, device):
        ...

    @property
    @abc.abstractmethod
    def name(self) -> str:
        ...

    def predict_depth(self, img: Tensor, intrinsics: CameraIntrinsics) -> PredictedDepth:
        raise NotImplementedError


class InputImage(typing.NamedTuple):
    """One training view: pixels [H,W,3] in [0,1], file name, camera-to-world [4,4], K [3,3]."""
    data: Tensor
    name: str
    cam2world: Tensor
    K: Tensor


@dataclasses.dataclass
class AdaptiveSubsamplingConfig:
    """Stride range of the depth-adaptive subsampler (near pixels get the larger stride)."""
    factor_range_min: int = 5
    factor_range_max: int = 15


@dataclasses.dataclass
class NumSfMPointsMaskConfig:
    """Patch grid (patches along the short image axis) and the SfM-point count above
    which a patch is left to the SfM points alone."""
    num_patches_small_axis: int = 20
    threshold: int = 15
