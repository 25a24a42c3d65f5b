"""Data contracts of the predictor boundary, mirror of
/root/reference/gs_init_compare/depth_prediction/predictors/depth_predictor_interface.py:9-71."""
from abc import ABCMeta, abstractmethod
from dataclasses import dataclass
from typing import NamedTuple, Optional

import torch


@dataclass
class PredictedDepth:
    depth: torch.Tensor                                  # float [H,W]
    mask: torch.Tensor                                   # bool  [H,W] valid pixels
    depth_confidence: Optional[torch.Tensor] = None
    normal: Optional[torch.Tensor] = None
    normal_confidence: Optional[torch.Tensor] = None


class CameraIntrinsics(NamedTuple):
    K: torch.Tensor

    @property
    def fx(self):
        return self.K[0, 0].item()

    @property
    def fy(self):
        return self.K[1, 1].item()

    @property
    def cx(self):
        return self.K[0, 2].item()

    @property
    def cy(self):
        return self.K[1, 2].item()


class DepthPredictor(metaclass=ABCMeta):
    @abstractmethod
    def __init__(self, config, device):
        pass

    @property
    @abstractmethod
    def name(self) -> str:
        ...

    def predict_depth(self, img: torch.Tensor, intrinsics: CameraIntrinsics) -> PredictedDepth:
        raise NotImplementedError
