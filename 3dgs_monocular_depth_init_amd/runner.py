"""Host-side mirror of the reference's training harness around the hot path.

Same names / argument meaning as /root/reference/gs_init_compare/runner.py:
  create_splats_with_optimizers   runner.py:53-138   (A9)
  Runner.rasterize_splats         runner.py:311-365  (A1)
  the step body of Runner.train   runner.py:464-547, 676-689 (A8: loss,
                                  backward, per-parameter Adam)
Everything numeric runs on the device through libgsrast.so or torch-ROCm ops;
dataset parsing, viewer, tensorboard, checkpoints etc. of the reference's
Runner are out of scope.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from .rendering import inverse4x4, rasterization

SH_C0 = 0.28209479177387814          # utils/runner_utils.py:150


def rgb_to_sh(rgb: Tensor) -> Tensor:
    """utils/runner_utils.py:149-151."""
    return (rgb - 0.5) / SH_C0


def set_random_seed(seed: int) -> None:
    """utils/runner_utils.py:154-157."""
    import random

    import numpy as np
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)


def create_splats_with_optimizers(
    points: Tensor,                     # [N,3] initial means (sfm | random | monocular_depth)
    rgbs: Tensor,                       # [N,3] in [0,1]
    scales: Tensor,                     # [N,3] LOG scales (runner.py:88-91 computes them from kNN)
    init_opacity: float = 0.1,
    scene_scale: float = 1.0,
    sh_degree: int = 3,
    batch_size: int = 1,
    device: str = "cuda",
    world_size: int = 1,
    quats: Optional[Tensor] = None,
    opacities_logit: Optional[Tensor] = None,
    shN: Optional[Tensor] = None,
    sparse_grad: bool = False,          # runner.py:63, 130: torch.optim.SparseAdam instead of Adam
) -> Tuple[torch.nn.ParameterDict, Dict[str, torch.optim.Optimizer]]:
    """runner.py:53-138 without the dataset parser: the caller supplies the
    point cloud. Unlike the reference (which shards Gaussians over ranks,
    runner.py:94-96) every rank keeps ALL Gaussians: the multi-GPU mode here is
    view-parallel replicas + gradient all-reduce (see distributed.py)."""
    N = points.shape[0]
    if quats is None:
        quats = torch.rand((N, 4))                                   # runner.py:99
    if opacities_logit is None:
        opacities_logit = torch.logit(torch.full((N,), init_opacity))   # runner.py:100
    K = (sh_degree + 1) ** 2
    sh0 = rgb_to_sh(rgbs)[:, None, :]
    if shN is None:
        shN = torch.zeros((N, K - 1, 3))
    params = [
        ("means", points, 1.6e-4 * scene_scale),
        ("scales", scales, 5e-3),
        ("quats", quats, 1e-3),
        ("opacities", opacities_logit, 5e-2),
        ("sh0", sh0, 2.5e-3),
        ("shN", shN, 2.5e-3 / 20),
    ]
    splats = torch.nn.ParameterDict(
        {n: torch.nn.Parameter(v.detach().clone().float().contiguous()) for n, v, _ in params}
    ).to(device)
    BS = batch_size * world_size                                         # runner.py:128-137
    optimizers = {
        name: (torch.optim.SparseAdam if sparse_grad else torch.optim.Adam)(
            [{"params": splats[name], "lr": lr * math.sqrt(BS), "name": name}],
            eps=1e-15 / math.sqrt(BS),
            betas=(1 - BS * (1 - 0.9), 1 - BS * (1 - 0.999)),
        )
        for name, _, lr in params
    }
    return splats, optimizers


@dataclass
class RasterConfig:
    """The subset of reference Config fields the path reads (config.py:125-149)."""
    sh_degree: int = 3
    near_plane: float = 0.01
    far_plane: float = 1e10
    packed: bool = False
    sparse_grad: bool = False           # config.py:147 (needs packed; SparseAdam on the rendered rows)
    antialiased: bool = False
    absgrad: bool = False
    camera_model: str = "pinhole"
    # list a (tile, Gaussian) pair only when the ellipse alpha >= 1/255 reaches the tile (exact
    # test) instead of gsplat's bounding-rectangle rule: same image and gradients, ~17 % fewer
    # pairs on the c4 scene (rendering.rasterization `_tight_tiles`)
    tight_tiles: bool = True


def rasterize_splats(
    splats,
    camtoworlds: Tensor,     # [C,4,4]
    Ks: Tensor,              # [C,3,3]
    width: int,
    height: int,
    cfg: RasterConfig = RasterConfig(),
    masks: Optional[Tensor] = None,
    **kwargs,
) -> Tuple[Tensor, Tensor, Dict]:
    """Runner.rasterize_splats (runner.py:311-365): activations + boundary call.
    sh0 / shN go to the kernels as two tensors: the reference's torch.cat at
    runner.py:338 (192 B/Gaussian read + write) is not materialised."""
    means = splats["means"]
    quats = splats["quats"]                         # normalised inside the kernel
    kwargs.pop("image_ids", None)
    colors = (splats["sh0"], splats["shN"])
    rasterize_mode = "antialiased" if cfg.antialiased else "classic"
    # runner.py:324-325 (exp / sigmoid) run inside the projection kernels, and
    # runner.py:347 `torch.linalg.inv(camtoworlds)` is one small launch
    viewmats, campos = inverse4x4(camtoworlds, translation_of="input")   # + camera positions, one launch
    render_colors, render_alphas, info = rasterization(
        means=means, quats=quats, scales=splats["scales"], opacities=splats["opacities"],
        colors=colors, viewmats=viewmats, Ks=Ks, width=width, height=height,
        packed=cfg.packed, absgrad=cfg.absgrad, sparse_grad=cfg.sparse_grad,
        rasterize_mode=rasterize_mode, distributed=False, camera_model=cfg.camera_model,
        _raw_activations=True, _campos=campos, _tight_tiles=cfg.tight_tiles, _isect_ids=False, **kwargs,
    )
    if masks is not None:
        render_colors[~masks] = 0
    return render_colors, render_alphas, info


def train_step(
    splats,
    optimizers: Optional[Dict[str, torch.optim.Optimizer]],
    camtoworlds: Tensor,
    Ks: Tensor,
    pixels: Tensor,          # [C,H,W,3] in [0,1]
    step: int,
    cfg: RasterConfig = RasterConfig(),
    ssim_lambda: float = 0.0,
    grad_sync=None,          # callable run between backward and optimizer.step (all-reduce)
    strategy=None,           # strategy.DefaultStrategy / MCMCStrategy (runner.py:497-503, 639-658)
    strategy_state=None,
    opacity_reg: float = 0.0,   # runner.py:535-539 (0.01 in the "mcmc" preset, trainer.py:83-92)
    scale_reg: float = 0.0,     # runner.py:540-545
) -> Tuple[Tensor, Dict]:
    """One iteration of Runner.train's body (runner.py:464-547, 676-689):
    SH-degree schedule, render, L1 (+ optional SSIM term), backward,
    optional gradient synchronisation, Adam step, zero_grad."""
    height, width = pixels.shape[1:3]
    sh_degree_to_use = min(step // 1000, cfg.sh_degree)                 # runner.py:464
    renders, alphas, info = rasterize_splats(
        splats, camtoworlds, Ks, width, height, cfg,
        sh_degree=sh_degree_to_use, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        render_mode="RGB")
    colors = renders if renders.shape[-1] == 3 else renders[..., :3]     # (no slice node in the RGB case)
    if strategy is not None:
        strategy.step_pre_backward(splats, optimizers, strategy_state, step, info)   # runner.py:497
    if ssim_lambda > 0.0:
        from .losses import l1_ssim_loss                                 # runner.py:506-510 fused
        loss = l1_ssim_loss(colors, pixels, ssim_lambda)
    else:
        from .losses import l1_loss
        loss = l1_loss(colors, pixels)                                  # runner.py:506
    if opacity_reg > 0.0:                                                # runner.py:535-539
        loss = loss + opacity_reg * torch.abs(torch.sigmoid(splats["opacities"])).mean()
    if scale_reg > 0.0:                                                  # runner.py:540-545
        loss = loss + scale_reg * torch.abs(torch.exp(splats["scales"])).mean()
    from . import rendering as _R
    from .losses import unit_gradient
    fused = _R._BACKWARD_OPTIMIZER
    # reference order on the steps where the strategy edits parameters: backward -> strategy ->
    # optimizer (runner.py:638-679); the fused update would land before the strategy
    ordered = fused is not None and strategy is not None and strategy.mutates_params(step)
    if fused is not None and not ordered and (opacity_reg > 0.0 or scale_reg > 0.0):
        # (on `ordered` steps -- every step of MCMCStrategy, whose preset uses both regularisers --
        # the fusion is suspended: the regularisers' gradients accumulate into .grad next to the
        # rasterizer's, identically on all ranks, and the optimizer steps once, afterwards)
        raise RuntimeError(
            "train_step: opacity_reg / scale_reg reach the parameters outside the rasterizer; "
            "optimizer-in-backward (FusedAdam.fuse_into_backward, GatherRowsSync) would apply them "
            "in a second Adam step. Disable the fusion for this preset.")
    if ordered:
        _R.set_backward_optimizer(None)
    try:
        one = unit_gradient(loss.device)
        loss.backward(one if one.dtype == loss.dtype else None)          # runner.py:547 (root
        # gradient handed over instead of a ones_like + fill launch per step)
    finally:
        if ordered:
            _R.set_backward_optimizer(fused)
    if grad_sync is not None:
        grad_sync()
    if strategy is not None:                                             # runner.py:639-658
        from .strategy import MCMCStrategy
        if isinstance(strategy, MCMCStrategy):
            means_lr = optimizers["means"].param_groups[0]["lr"]        # schedulers[0].get_last_lr()[0]
            strategy.step_post_backward(splats, optimizers, strategy_state, step, info, lr=means_lr)
        else:
            strategy.step_post_backward(splats, optimizers, strategy_state, step, info, packed=cfg.packed)
    if optimizers is not None:
        from .optim import FusedSparseAdam
        if isinstance(optimizers, FusedSparseAdam):                      # cfg.sparse_grad, one launch
            optimizers.step(info)
            optimizers.zero_grad(set_to_none=True)
        elif hasattr(optimizers, "step"):                                # FusedAdam: one launch
            optimizers.step()
            optimizers.zero_grad(set_to_none=True)
        else:
            if cfg.sparse_grad:                                          # runner.py:661-672, literally
                assert cfg.packed, "Sparse gradients only work with packed mode."
                gaussian_ids = info["gaussian_ids"]
                for k in splats.keys():
                    grad = splats[k].grad
                    if grad is None or grad.is_sparse:
                        continue
                    splats[k].grad = torch.sparse_coo_tensor(
                        indices=gaussian_ids[None], values=grad[gaussian_ids], size=splats[k].size(),
                        is_coalesced=len(Ks) == 1)
            for opt in optimizers.values():                              # runner.py:676-679
                opt.step()
                opt.zero_grad(set_to_none=True)
    if grad_sync is not None and hasattr(grad_sync, "finish"):
        grad_sync.finish()           # chunks no optimizer consumed (e.g. optimizers=None)
    return loss.detach(), info
