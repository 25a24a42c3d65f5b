"""CPU oracle for the training loss.  TEST INFRASTRUCTURE ONLY.

PARITY UNPINNED: `fused_ssim` is the third-party CUDA op rahul-goel/fused-ssim
@30fb258 (/root/reference/setup.py:14), absent from /root/reference and not
installable offline; the reference holds no fixtures for it. This restates its
published definition (11x11 Gaussian window sigma 1.5, zero padding,
C1 = 0.01^2, C2 = 0.03^2, padding="valid" crops 5 px) with torch ops + autograd,
anchored on the call site gs_init_compare/runner.py:506-510.
"""
import math

import torch
import torch.nn.functional as F


def _window(channels: int, dtype):
    g = torch.tensor([math.exp(-((i - 5) ** 2) / (2 * 1.5 ** 2)) for i in range(11)], dtype=torch.float64)
    g = (g / g.sum()).to(dtype)
    w2 = (g[:, None] * g[None, :])[None, None]
    return w2.expand(channels, 1, 11, 11).contiguous()


def ssim_map(img1, img2):
    C = img1.shape[1]
    w = _window(C, img1.dtype)
    conv = lambda x: F.conv2d(x, w, padding=5, groups=C)
    mu1, mu2 = conv(img1), conv(img2)
    s1 = conv(img1 * img1) - mu1 * mu1
    s2 = conv(img2 * img2) - mu2 * mu2
    s12 = conv(img1 * img2) - mu1 * mu2
    C1, C2 = 0.01 ** 2, 0.03 ** 2
    return ((2 * mu1 * mu2 + C1) * (2 * s12 + C2)) / ((mu1 * mu1 + mu2 * mu2 + C1) * (s1 + s2 + C2))


def fused_ssim(img1, img2, padding="same"):
    m = ssim_map(img1, img2)
    if padding == "valid":
        m = m[:, :, 5:-5, 5:-5]
    return m.mean()


def l1_ssim_loss(colors, pixels, ssim_lambda=0.2):
    """runner.py:506-510 on NHWC inputs."""
    l1 = F.l1_loss(colors, pixels)
    ssim = fused_ssim(colors.permute(0, 3, 1, 2), pixels.permute(0, 3, 1, 2), padding="valid")
    return l1 * (1.0 - ssim_lambda) + (1.0 - ssim) * ssim_lambda
