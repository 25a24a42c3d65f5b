"""Metric3D v2 predictor: device pre/post-processing around
an injected depth network.

Mirror of /root/reference/gs_init_compare/depth_prediction/predictors/metric3d.py:18-139.
The reference loads the network with `torch.hub.load("yvanyin/metric3d", ...,
pretrain=True)` (lines 27-31): a remote fetch of code + weights that is not
available offline, so the network is a constructor argument here -- any object
with `inference({"input": x [1,3,616,1064]}) -> (depth [1,1,h,w], confidence,
{"prediction_normal": [1,4,h,w]})`. Everything around it -- uint8 conversion,
channel flip, keep-ratio resize (cv2.INTER_LINEAR on the CPU in the reference),
mean-colour border, normalisation, un-padding, bilinear upsampling,
de-canonicalisation (x fx/1000) and the [0,300] clamp -- runs in two fused
kernels on the device, without the reference's device->host->device round trip.
"""
from __future__ import annotations

import torch

from ..._lib import call, ptr
from .depth_predictor_interface import CameraIntrinsics, DepthPredictor, PredictedDepth

INPUT_SIZE = (616, 1064)          # metric3d.py:47 (ViT models)
CANONICAL_FOCAL = 1000.0          # metric3d.py:127-129


def _st():
    return torch.cuda.current_stream().cuda_stream


def preprocess(img: torch.Tensor):
    """img float [H,W,3] in [0,1] on the device -> (net input [1,3,616,1064], pad_info, scale)."""
    H, W = img.shape[:2]
    scale = min(INPUT_SIZE[0] / H, INPUT_SIZE[1] / W)                 # :49
    rh, rw = int(H * scale), int(W * scale)                           # :50-52
    pad_h, pad_w = INPUT_SIZE[0] - rh, INPUT_SIZE[1] - rw             # :62-65
    pad_info = [pad_h // 2, pad_h - pad_h // 2, pad_w // 2, pad_w - pad_w // 2]
    out = torch.empty(1, 3, *INPUT_SIZE, dtype=torch.float32, device=img.device)
    img = img.contiguous().float()
    call("gsr_m3d_preprocess", H, W, ptr(img), rh, rw, pad_info[0], pad_info[2], INPUT_SIZE[0],
         INPUT_SIZE[1], ptr(out), _st())
    return out, pad_info, scale


def to_og_size(t: torch.Tensor, pad_info, size, scale: float = 1.0, clamp=None) -> torch.Tensor:
    """metric3d.py:96-118 for one [h,w] map (+ optional scale / clamp, :127-131)."""
    t = t.contiguous().float()
    H, W = size
    out = torch.empty(H, W, dtype=torch.float32, device=t.device)
    lo, hi = clamp if clamp is not None else (0.0, 0.0)
    call("gsr_m3d_postprocess", t.shape[0], t.shape[1], ptr(t), pad_info[0], pad_info[1],
         pad_info[2], pad_info[3], H, W, float(scale), float(lo), float(hi), int(clamp is not None),
         ptr(out), _st())
    return out


class Metric3d(DepthPredictor):
    def __init__(self, config, device: str, model=None, backbone: str = None, weights=None):
        """`Metric3d(config, device)` as in the reference (metric3d.py:19-33). The network is
        `metric3d_net.Metric3DNet` (MFMA fp16) built from a local state dict: `weights=` (a path
        or a dict) or $METRIC3D_WEIGHTS -- the reference's torch.hub download is unavailable
        offline, and nothing is fetched silently. `model=` injects any object with `.inference`."""
        import os
        if backbone is None:
            b = getattr(getattr(getattr(config, "mdi", None), "metric3d", None), "backbone", "vits")
            backbone = getattr(b, "value", b)
        if model is None:
            weights = weights if weights is not None else os.environ.get("METRIC3D_WEIGHTS")
            if weights is None:
                raise RuntimeError(
                    "Metric3d needs network weights: the reference fetches them with torch.hub "
                    "(metric3d.py:27-31), which is unavailable offline; pass weights=<state dict or "
                    "path>, set METRIC3D_WEIGHTS, or pass model=<network>")
            from .metric3d_net import Metric3DNet
            if not isinstance(weights, dict):
                weights = torch.load(weights, map_location="cpu", weights_only=True)
            model = Metric3DNet(weights, backbone=backbone, device=device, input_size=INPUT_SIZE)
        self.__name = f"Metric3d_{backbone}"
        self.__model = model
        self.device = device

    @property
    def name(self) -> str:
        return self.__name

    batch_size = 4      # images per network call in predict_depths (the init loop predicts every training image)

    @torch.no_grad()
    def predict_depth(self, img: torch.Tensor, intrinsics: CameraIntrinsics) -> PredictedDepth:
        return self.predict_depths([img], [intrinsics])[0]

    @torch.no_grad()
    def predict_depths(self, imgs, intrinsics) -> list:
        """predict_depth (metric3d.py:38-139) for a list of images, `batch_size` of them per network call
        (metric3d_net: one batched encoder pass, the decoders as parallel branches of one HIP graph)."""
        out = []
        for a in range(0, len(imgs), self.batch_size):
            chunk = [im.to(self.device) for im in imgs[a:a + self.batch_size]]
            pre = [preprocess(im) for im in chunk]
            rgb = pre[0][0] if len(pre) == 1 else torch.cat([p[0] for p in pre], 0)
            pred_depth, confidence, output_dict = self.__model.inference({"input": rgb})
            normal = output_dict["prediction_normal"]
            for i, (im, (_, pad_info, scale)) in enumerate(zip(chunk, pre)):
                H, W = im.shape[:2]
                fx_scaled = intrinsics[a + i].fx * scale                            # :54-59
                depth = to_og_size(pred_depth[i, 0], pad_info, (H, W),
                                   scale=fx_scaled / CANONICAL_FOCAL, clamp=(0.0, 300.0))   # :127-131
                conf = to_og_size(confidence[i, 0], pad_info, (H, W))
                n = torch.stack([to_og_size(normal[i, k], pad_info, (H, W)) for k in range(3)], dim=-1)
                n_conf = to_og_size(normal[i, 3], pad_info, (H, W))
                out.append(PredictedDepth(depth=depth, mask=torch.ones_like(depth, dtype=torch.bool),
                                          depth_confidence=conf, normal=n, normal_confidence=n_conf))
        return out
