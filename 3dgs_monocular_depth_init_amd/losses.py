"""Fused training loss: L1 + SSIM in two HIP launches.

`fused_ssim(img1, img2, padding=..., train=...)` keeps the call signature of the
third-party op the reference imports (`from fused_ssim import fused_ssim`,
/root/reference/gs_init_compare/runner.py:17, used at runner.py:507: NCHW
inputs, 11x11 Gaussian window sigma 1.5, zero padding for the window,
padding="valid" = mean over the map cropped by 5 px). `l1_ssim_loss` fuses the
whole loss of runner.py:506-510 (forward: one launch, backward: one launch,
instead of ~12 elementwise/conv kernels) and consumes the rasterizer's NHWC
output in place.
"""
from __future__ import annotations

import ctypes as C

import torch
from torch import Tensor

from ._lib import current_stream as _raw_stream
from ._lib import call, ptr


def _strides(t: Tensor):
    return (C.c_int64 * 4)(*t.stride())


def _st():
    return _raw_stream()


_SSIM_WS = {}    # (device index, shape) -> per-workgroup partial sums of the forward


class _SsimL1(torch.autograd.Function):
    """Returns (mean SSIM over the counted region, mean |a-b|, (1-lam)*L1 + lam*(1-SSIM))
    for logical NCHW views: one launch forward, one backward, no elementwise launches."""

    @staticmethod
    def forward(ctx, img1: Tensor, img2: Tensor, valid: bool, train: bool, lam: float):
        N, CH, H, W = img1.shape
        dev = img1.device
        # (unused outputs must reach backward as None, not as zero tensors: materialised, they cost three fill launches and
        # send the backward down the several-outputs branch -- six more element-wise launches and 0.05 ms of host time per step)
        ctx.set_materialize_grads(False)
        ws = _SSIM_WS.get((dev.index, N, CH, H, W))
        if ws is None:
            n_ws = 2 * ((W + 31) // 32) * ((H + 31) // 32) * N * CH      # = gsr_ssim_workspace_doubles
            ws = _SSIM_WS[(dev.index, N, CH, H, W)] = torch.empty(n_ws, dtype=torch.float64, device=dev)
        out = torch.empty(3, dtype=torch.float32, device=dev)
        need = train and img1.requires_grad
        maps = torch.empty(3, N, CH, H, W, dtype=torch.float32, device=dev) if need else None
        call("gsr_ssim_l1_fwd", N, CH, H, W, ptr(img1), _strides(img1), ptr(img2), _strides(img2),
             int(valid), ptr(ws), ptr(out), float(lam), ptr(maps[0]) if need else None,
             ptr(maps[1]) if need else None, ptr(maps[2]) if need else None, _st())
        hh, ww = (H - 10, W - 10) if valid else (H, W)
        ctx.counts = (max(N * CH * max(hh, 0) * max(ww, 0), 1), N * CH * H * W)
        ctx.lam = float(lam)
        ctx.save_for_backward(img1, img2, maps)
        return out[0], out[1], out[2]

    @staticmethod
    def backward(ctx, v_ssim, v_l1, v_loss):
        img1, img2, maps = ctx.saved_tensors
        if maps is None:
            raise RuntimeError("fused_ssim was called with train=False; no backward available")
        if v_ssim is None and v_l1 is None and v_loss is None:
            return None, None, None, None, None
        N, CH, H, W = img1.shape
        n_ssim, n_l1 = ctx.counts
        lam = ctx.lam
        dev = img1.device
        grad = torch.empty_strided(img1.shape, img1.stride(), dtype=torch.float32, device=dev)
        live = [(v, k) for k, v in enumerate((v_ssim, v_l1, v_loss)) if v is not None]
        w = up = None
        s_ssim = s_l1 = 0.0
        if len(live) == 1:          # the usual cases: only the loss, or only the SSIM, is used
            up, k = live[0]
            up = up if (up.dtype == torch.float32 and up.is_cuda) else up.to(dev, torch.float32)
            s_ssim = (1.0 / n_ssim, 0.0, -lam / n_ssim)[k]
            s_l1 = (0.0, 1.0 / n_l1, (1.0 - lam) / n_l1)[k]
        else:                        # several outputs used at once: combine the weights on the device
            z = torch.zeros((), device=dev)
            vs, vl, vt = ((v if v is not None else z).float() for v in (v_ssim, v_l1, v_loss))
            w = torch.stack([(vs - lam * vt) / n_ssim, (vl + (1.0 - lam) * vt) / n_l1]).contiguous()
        call("gsr_ssim_l1_bwd", N, CH, H, W, ptr(img1), _strides(img1), ptr(img2), _strides(img2),
             ptr(maps[0]), ptr(maps[1]), ptr(maps[2]), ptr(w), ptr(up), s_ssim, s_l1, ptr(grad),
             _strides(grad), _st())
        return grad, None, None, None, None


_L1_WS = {}      # device index -> 2 zeroed doubles the forward kernel leaves zero again
_UNIT = {}       # device -> the scalar 1.0 handed to loss.backward() by runner.train_step


def unit_gradient(device) -> Tensor:
    """A cached scalar 1.0 to pass as `loss.backward(gradient=...)`: saves autograd's
    ones_like + fill per step, and lets `_L1.backward` recognise (by identity, without
    reading the device) that its precomputed gradient needs no scaling."""
    t = _UNIT.get(device)
    if t is None:
        t = _UNIT[device] = torch.ones((), dtype=torch.float32, device=device)
    return t


class _L1(torch.autograd.Function):
    """mean |a - b| over contiguous buffers: one streaming launch (per-workgroup partial sums, and
    sign(a-b)/n written on the way) plus a one-workgroup launch that adds the partials; the backward returns that as is when the
    upstream gradient is the cached unit scalar, and scales it (one launch) otherwise."""

    @staticmethod
    def forward(ctx, a: Tensor, b: Tensor):
        ws = _L1_WS.get(a.device.index)
        if ws is None:
            ws = _L1_WS[a.device.index] = torch.empty(512, dtype=torch.float64, device=a.device)   # GSR_L1_WS_DOUBLES
        out = torch.empty((), dtype=torch.float32, device=a.device)
        grad = torch.empty_like(a) if a.requires_grad else None
        call("gsr_l1_fwd", a.numel(), ptr(a), ptr(b), ptr(ws), ptr(out), ptr(grad), _st())
        ctx.unit_grad = grad
        return out

    @staticmethod
    def backward(ctx, v):
        grad = ctx.unit_grad
        if grad is None:
            return None, None
        unit = _UNIT.get(v.device)
        if unit is not None and v.data_ptr() == unit.data_ptr():
            return grad, None                      # upstream is exactly 1
        return grad * v.to(grad.dtype), None


def l1_loss(colors: Tensor, pixels: Tensor) -> Tensor:
    """F.l1_loss(colors, pixels) (runner.py:506) in two launches."""
    if not (colors.is_cuda and pixels.is_cuda):
        from ._lib import GsrastError
        raise GsrastError("l1_loss: tensors must be on a ROCm device; there is no CPU path")
    assert colors.shape == pixels.shape
    return _L1.apply(colors.contiguous().float(), pixels.detach().contiguous().float())


def _check(img1: Tensor, img2: Tensor):
    if not (img1.is_cuda and img2.is_cuda):
        from ._lib import GsrastError
        raise GsrastError("fused_ssim: tensors must be on a ROCm device; there is no CPU path")
    assert img1.shape == img2.shape and img1.dim() == 4, "expected [N,C,H,W]"
    return img1.float(), img2.float()


def fused_ssim(img1: Tensor, img2: Tensor, padding: str = "same", train: bool = True) -> Tensor:
    """Mean SSIM of NCHW images (drop-in for fused_ssim.fused_ssim, runner.py:507)."""
    assert padding in ("same", "valid")
    img1, img2 = _check(img1, img2)
    ssim, _, _ = _SsimL1.apply(img1, img2.detach(), padding == "valid", train, 0.0)
    return ssim


def l1_ssim_loss(colors: Tensor, pixels: Tensor, ssim_lambda: float = 0.2) -> Tensor:
    """runner.py:506-510 fused: colors / pixels are NHWC [C,H,W,3]."""
    c, p = _check(colors.permute(0, 3, 1, 2), pixels.permute(0, 3, 1, 2))
    _, _, loss = _SsimL1.apply(c, p.detach(), True, True, float(ssim_lambda))
    return loss
