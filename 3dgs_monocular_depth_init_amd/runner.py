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
    sh_degree_interval: int = 1000      # config.py:127 (scaled by Config.adjust_steps)
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
    pre = kwargs.pop("_viewmats_campos", None)      # (runner.train: the frame's inverse, taken once when it became resident)
    viewmats, campos = pre if pre is not None else inverse4x4(camtoworlds, translation_of="input")   # + camera positions, one launch
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


# The plain L1 loss inside the compositing forward (gsr_rasterize_fwd_l1) when a step's configuration allows it
# (train_step); False keeps the separate loss launches (A/B, tests).
L1_IN_FORWARD = True
# With the L1 + SSIM loss the render (and the gradient the loss returns) is kept in planes (gsr_rasterize_fwd_planar /
# _bwd_planar); False keeps [C,H,W,3] memory (A/B, tests).
PLANAR_RENDER_FOR_SSIM = True


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
    random_background: bool = False,    # runner.py:493-495 (config.py:152)
    masks: Optional[Tensor] = None,     # runner.py:449, 363
    depth_points: Optional[Tensor] = None,   # runner.py:450-452, 511-529 (cfg.depth_loss): [C,M,2] pixel coordinates
    depth_gt: Optional[Tensor] = None,       # [C,M]
    depth_lambda: float = 1e-2,
    scene_scale: float = 1.0,
    before_update=None,         # callable(loss, info): runs after backward, before strategy / optimizer
                                # (the reference's checkpoint point, runner.py:592-637); suspends the fusion
    viewmats_campos=None,       # (viewmats [C,4,4], camera positions [C,3]) = inverse4x4(camtoworlds, "input") when the
                                # caller already holds them (fixed cameras): saves the step's 4x4-inverse launch
) -> Tuple[Tensor, Dict]:
    """One iteration of Runner.train's body (runner.py:464-547, 639-689):
    SH-degree schedule, render, L1 (+ optional SSIM term), backward,
    optional gradient synchronisation, strategy, Adam step, zero_grad."""
    height, width = pixels.shape[1:3]
    sh_degree_to_use = min(step // max(cfg.sh_degree_interval, 1), cfg.sh_degree)   # runner.py:464
    depth_loss = depth_points is not None
    # the plain L1 loss (runner.py:506 with ssim_lambda = 0) is taken inside the compositing forward when nothing else
    # reads the render: no 75 MB loss pass of its own (gsr_rasterize_fwd_l1)
    l1_in_forward = (L1_IN_FORWARD and ssim_lambda <= 0.0 and not depth_loss and not random_background and masks is None
                     and pixels.dtype == torch.float32 and pixels.shape[-1] == 3 and pixels.is_cuda)
    renders, alphas, info = rasterize_splats(
        splats, camtoworlds, Ks, width, height, cfg, masks=masks,
        sh_degree=sh_degree_to_use, near_plane=cfg.near_plane, far_plane=cfg.far_plane,
        render_mode="RGB+ED" if depth_loss else "RGB",                   # runner.py:476
        **({"_l1_target": pixels} if l1_in_forward else {}),
        # (the fused L1 + SSIM loss reads the render, and writes its gradient, a plane at a time)
        **({"_planar_render": True} if (PLANAR_RENDER_FOR_SSIM and ssim_lambda > 0.0 and not depth_loss and masks is None) else {}),
        **({"_viewmats_campos": viewmats_campos} if viewmats_campos is not None else {}))
    if renders is None:                                                  # (l1_in_forward)
        colors, depths = None, None
    elif renders.shape[-1] == 4:                                         # runner.py:479-482
        colors, depths = renders[..., 0:3], renders[..., 3:4]
    else:
        colors, depths = renders, None                                   # (no slice node in the RGB case)
    if random_background:                                                # runner.py:493-495
        background = torch.rand(1, 3, device=colors.device)
        colors = colors + background * (1.0 - alphas)
    if strategy is not None:
        strategy.step_pre_backward(splats, optimizers, strategy_state, step, info)   # runner.py:497
    if l1_in_forward:
        loss = info["l1_loss"]
    elif ssim_lambda > 0.0:
        from .losses import l1_ssim_loss                                 # runner.py:506-510 fused
        loss = l1_ssim_loss(colors, pixels, ssim_lambda)
    else:
        from .losses import l1_loss
        loss = l1_loss(colors, pixels)                                  # runner.py:506
    if depth_loss:                                                       # runner.py:511-529
        import torch.nn.functional as F
        pts = torch.stack([depth_points[:, :, 0] / (width - 1) * 2 - 1,
                           depth_points[:, :, 1] / (height - 1) * 2 - 1], dim=-1)    # normalize to [-1, 1]
        d = F.grid_sample(depths.permute(0, 3, 1, 2), pts.unsqueeze(2), align_corners=True)   # [C,1,M,1]
        d = d.squeeze(3).squeeze(1)
        disp = torch.where(d > 0.0, 1.0 / d, torch.zeros_like(d))       # loss in disparity space
        loss = loss + F.l1_loss(disp, 1.0 / depth_gt) * scene_scale * depth_lambda
    from . import rendering as _R
    from .losses import unit_gradient
    from .strategy import MCMCStrategy
    fused = _R._BACKWARD_OPTIMIZER
    regs = opacity_reg > 0.0 or scale_reg > 0.0
    # The fused backward can take the "mcmc" preset's extras along (optim.FusedAdam.set_step_extras ->
    # gsr_project_bwd_adam_ex): the regularisers' gradients, and MCMCStrategy's position noise on the steps between
    # refinements -- computed from the pre-update parameters and added before the Adam update, i.e. the reference's
    # strategy-then-optimizer order (runner.py:649-679). Single rank only.
    extras_ok = fused is not None and hasattr(fused, "set_step_extras") and grad_sync is None
    mcmc_noise = (extras_ok and isinstance(strategy, MCMCStrategy) and not strategy.refines(step)
                  and before_update is None)
    # reference order on the steps where the strategy edits parameters: backward -> strategy ->
    # optimizer (runner.py:638-679); the fused update would land before the strategy
    ordered = fused is not None and ((strategy is not None and strategy.mutates_params(step) and not mcmc_noise)
                                     or before_update is not None)
    # DefaultStrategy's per-step statistics travel the same way (the rows and radii are in the kernel's registers)
    from .strategy import DefaultStrategy
    stats = None
    if (extras_ok and not ordered and isinstance(strategy, DefaultStrategy) and step < strategy.refine_stop_iter
            and camtoworlds.shape[0] == 1 and not cfg.packed):
        stats = strategy.stats_for_fused_backward(splats, strategy_state, info)
    use_extras = extras_ok and not ordered and (regs or mcmc_noise or stats is not None)
    reg_value = None
    if regs and use_extras:             # their gradients come from the fused backward; the VALUE still belongs to the loss
        with torch.no_grad():
            reg_value = 0.0
            if opacity_reg > 0.0:
                reg_value = reg_value + opacity_reg * torch.sigmoid(splats["opacities"]).mean()
            if scale_reg > 0.0:
                reg_value = reg_value + scale_reg * torch.exp(splats["scales"]).mean()
    else:
        if opacity_reg > 0.0:                                                # runner.py:535-539
            loss = loss + opacity_reg * torch.abs(torch.sigmoid(splats["opacities"])).mean()
        if scale_reg > 0.0:                                                  # runner.py:540-545
            loss = loss + scale_reg * torch.abs(torch.exp(splats["scales"])).mean()
    if fused is not None and not ordered and regs and not use_extras:
        # (on `ordered` steps the fusion is suspended: the regularisers' gradients accumulate into .grad next
        # to the rasterizer's, identically on all ranks, and the optimizer steps once, afterwards)
        raise RuntimeError(
            "train_step: opacity_reg / scale_reg reach the parameters outside the rasterizer; "
            "optimizer-in-backward under a row exchange (GatherRowsSync) would apply them "
            "in a second Adam step. Disable the fusion for this preset.")
    if ordered:
        _R.set_backward_optimizer(None)
    if use_extras:
        noise, noise_scale = None, 0.0
        if mcmc_noise:
            noise = strategy.draw_noise(splats, strategy_state)
            noise_scale = optimizers["means"].param_groups[0]["lr"] * strategy.noise_lr
        fused.set_step_extras(noise, noise_scale, opacity_reg, scale_reg, stats)
    try:
        one = unit_gradient(loss.device)
        loss.backward(one if one.dtype == loss.dtype else None)          # runner.py:547 (root
        # gradient handed over instead of a ones_like + fill launch per step)
    finally:
        if ordered:
            _R.set_backward_optimizer(fused)
    if use_extras and fused.take_step_extras() is not None:
        raise RuntimeError("train_step: the projection backward did not take the fused update (and with it the step's "
                           "noise / regulariser gradients); disable fuse_into_backward for this configuration")
    if reg_value is not None:
        loss = loss.detach() + reg_value
    if grad_sync is not None:
        grad_sync()
    if before_update is not None:
        before_update(loss.detach(), info)
    if strategy is not None:                                             # runner.py:639-658
        if isinstance(strategy, MCMCStrategy):
            means_lr = optimizers["means"].param_groups[0]["lr"]        # schedulers[0].get_last_lr()[0]
            strategy.step_post_backward(splats, optimizers, strategy_state, step, info, lr=means_lr,
                                        noise_done=mcmc_noise)
        else:
            strategy.step_post_backward(splats, optimizers, strategy_state, step, info, packed=cfg.packed,
                                        **({"stats_done": True} if stats is not None else {}))
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


def raster_config_of(cfg) -> RasterConfig:
    """The RasterConfig a reference-style Config (config.Config) implies."""
    return RasterConfig(
        sh_degree=cfg.sh_degree, near_plane=cfg.near_plane, far_plane=cfg.far_plane, packed=cfg.packed,
        sparse_grad=cfg.sparse_grad, antialiased=cfg.antialiased,
        absgrad=bool(getattr(cfg.strategy, "absgrad", False) or getattr(cfg, "absgrad", False)),   # runner.py:352-356
        camera_model=cfg.camera_model, sh_degree_interval=cfg.sh_degree_interval,
        tight_tiles=getattr(cfg, "tight_tiles", True))


@torch.no_grad()
def evaluate(splats, valset, cfg, device: str = "cuda") -> Dict[str, float]:
    """Runner.eval (runner.py:711-789) without the image dumps, tensorboard and LPIPS (a downloaded
    network): per held-out view render at the full SH degree, clamp to [0,1], PSNR (data range 1) and
    mean SSIM (the 11x11 Gaussian-window kernel of losses.fused_ssim), averaged over the views;
    `ellipse_time` = seconds per image bracketed by synchronisations as in runner.py:731-744."""
    import time

    from .losses import fused_ssim
    rc = raster_config_of(cfg)
    psnr, ssim, t = [], [], 0.0
    for data in valset:
        c2w = data["camtoworld"].to(device).reshape(1, 4, 4)
        K = data["K"].to(device).reshape(1, 3, 3)
        pixels = data["image"].to(device).float().reshape(1, *data["image"].shape[-3:]) / 255.0
        masks = data["mask"].to(device)[None] if "mask" in data else None
        height, width = pixels.shape[1:3]
        torch.cuda.synchronize()
        tic = time.time()
        colors, _, _ = rasterize_splats(splats, c2w, K, width, height, rc, masks=masks, sh_degree=cfg.sh_degree,
                                        near_plane=cfg.near_plane, far_plane=cfg.far_plane)
        torch.cuda.synchronize()
        t += time.time() - tic
        colors = torch.clamp(colors, 0.0, 1.0)
        mse = torch.mean((colors - pixels) ** 2)
        psnr.append(-10.0 * torch.log10(mse.clamp_min(1e-20)))
        ssim.append(fused_ssim(colors.permute(0, 3, 1, 2), pixels.permute(0, 3, 1, 2), padding="same", train=False))
    n = max(len(psnr), 1)
    return {"psnr": float(torch.stack(psnr).mean()) if psnr else float("nan"),
            "ssim": float(torch.stack(ssim).mean()) if ssim else float("nan"),
            "ellipse_time": t / n, "num_GS": len(splats["means"])}


def train(
    splats,
    optimizers: Dict[str, torch.optim.Optimizer],
    trainset,                       # sequence of dicts as datasets/colmap.py:381-412 yields them:
                                    # "camtoworld" [4,4], "K" [3,3], "image" [H,W,3] 0-255 (+ "mask", "points", "depths")
    cfg,                            # config.Config (reference field names and defaults)
    strategy_state=None,
    *,
    valset=None,
    scene_scale: float = 1.0,
    result_dir=None,                # checkpoints / stats are written only when given (rank 0 writes stats)
    world_rank: int = 0,
    world_size: int = 1,
    grad_sync=None,                 # distributed.GradSync / GatherRowsSync for view-parallel replicas
    fuse_optimizer: bool = True,    # wrap the six Adams in optim.FusedAdam, update inside the backward where legal
    device: str = "cuda",
    seed: int = 42,
    progress=None,                  # callable(step, stats_dict) every `progress_every` steps (stands in for tqdm / tensorboard)
    progress_every: int = 100,
    device_cache_bytes: int = 64 << 30,   # frames kept resident on the device once used (0: none), see below
) -> Dict:
    """Runner.train (runner.py:367-709) around `train_step`, with the reference's schedule:

    * `ExponentialLR(optimizers["means"], gamma = 0.01 ** (1 / max_steps))`, stepped once per iteration
      after the optimizers (runner.py:381-386, 687-689);
    * SH degree `min(step // cfg.sh_degree_interval, cfg.sh_degree)` from step 0 (runner.py:464);
    * loss `(1 - ssim_lambda) * L1 + ssim_lambda * (1 - SSIM)` (runner.py:506-510), optional depth loss,
      opacity / scale regularisers, random background;
    * `cfg.strategy` pre / post backward in the reference's order (runner.py:497-503, 639-658);
    * the checkpoint `{"step", "splats"}` (+ PLY) at `step in [i - 1 for i in cfg.save_steps]` or the last
      step, written BEFORE that step's update (runner.py:592-637), and `evaluate` at `cfg.eval_steps`
      (runner.py:692-694);
    * shuffled epochs over `trainset` in batches of `cfg.batch_size` (runner.py:411-441): the permutation
      comes from a generator seeded with `seed` on every rank, rank r takes batch entries r, r + W, ...
      (view-parallel replicas; the reference's Gaussian-sharded ranks each draw their own).

    Out of scope, as in SURVEY.md section 2: viewer, tensorboard, pose / appearance optimisation, bilateral
    grid, compression. Returns the run's statistics (per-interval ms/step, Gaussian counts, losses, the
    learning rate at the end, evaluation results, checkpoint paths)."""
    import json
    import time
    import warnings
    from pathlib import Path

    from . import io as gs_io
    from . import rendering as _R
    from .optim import FusedAdam, FusedSparseAdam
    from .strategy import DefaultStrategy, MCMCStrategy

    max_steps = int(cfg.max_steps)
    strategy = cfg.strategy
    rc = raster_config_of(cfg)
    if strategy_state is None and strategy is not None:
        strategy.check_sanity(splats, optimizers)                                    # runner.py:208
        strategy_state = (strategy.initialize_state(scene_scale=scene_scale)         # runner.py:210-217
                          if isinstance(strategy, DefaultStrategy) else strategy.initialize_state())
    opt = optimizers
    regs = cfg.opacity_reg > 0.0 or cfg.scale_reg > 0.0
    fused_here = False
    if fuse_optimizer and isinstance(optimizers, dict):
        if cfg.sparse_grad:
            opt = FusedSparseAdam(optimizers)
        else:
            opt = FusedAdam(optimizers)
            if grad_sync is not None and hasattr(grad_sync, "attach"):
                grad_sync.attach(opt)
            elif grad_sync is None and world_size == 1 and _R._BACKWARD_OPTIMIZER is None:
                # (the regularisers' gradients and MCMCStrategy's noise travel with the fused backward: train_step)
                opt.fuse_into_backward(True)
                fused_here = True
    means_opt = opt["means"]
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        # (the fused launch never calls torch's Optimizer.step(), which is all the scheduler's
        # "lr_scheduler.step() before optimizer.step()" warning looks at)
        schedulers = [torch.optim.lr_scheduler.ExponentialLR(means_opt, gamma=0.01 ** (1.0 / max_steps))]
    means_opt._opt_called = True
    out_dir = Path(result_dir) if result_dir is not None else None
    if out_dir is not None:
        (out_dir / "ckpts").mkdir(parents=True, exist_ok=True)
        (out_dir / "stats").mkdir(parents=True, exist_ok=True)
    n_train = len(trainset)
    gen = torch.Generator().manual_seed(seed)
    B, W = int(cfg.batch_size), int(world_size)
    order: list = []

    # Frames stay on the device once they have been used, as the batch entries the step consumes: the image as
    # [1,H,W,3] float32 in [0,1] (the reference uploads and divides by 255 on every step, runner.py:446: 25 MB over
    # PCIe per 1080p frame -- longer than a step takes here -- and, for frames already on the device, one element-wise
    # launch per step), poses / intrinsics / masks / depth points as [1, ...]. 100 views at 1080p are 2.5 GB of the
    # 288; beyond `device_cache_bytes` frames are prepared per step as before.
    resident: dict = {}
    resident_bytes = 0

    def prepared(i):
        nonlocal resident_bytes
        hit = resident.get(i)
        if hit is not None:
            return hit
        d = trainset[i]
        # (contiguous once, here: a camtoworld that came out of torch.linalg.inv is column-major, and the 4x4 inverse
        # launch would re-pack it on every step)
        e = {k: d[k].to(device)[None].contiguous() for k in ("camtoworld", "K", "mask", "points", "depths") if k in d}
        e["pixels"] = (d["image"].to(device)[None].float() / 255.0).contiguous()
        if cfg.ssim_lambda > 0.0:        # planes in memory, [1,H,W,3] in shape: what the SSIM kernels read fastest
            e["pixels"] = e["pixels"].permute(0, 3, 1, 2).contiguous().permute(0, 2, 3, 1)
        if e["camtoworld"].is_cuda:      # fixed cameras (no pose optimisation here): the inverse once per frame, not per step
            e["viewmats_campos"] = _R.inverse4x4(e["camtoworld"], translation_of="input")
        size = sum(t.numel() * t.element_size() for t in e.values() if isinstance(t, Tensor))
        if resident_bytes + size <= device_cache_bytes:
            resident[i] = e
            resident_bytes += size
        return e

    def next_batch():
        nonlocal order
        need = B * W
        if len(order) < need:                       # DataLoader(shuffle=True): a fresh permutation per epoch
            order = order + torch.randperm(n_train, generator=gen).tolist()
        take, order = order[:need], order[need:]
        return [prepared(i) for i in take[world_rank::W]], take

    save_at = {i - 1 for i in cfg.save_steps} | {max_steps - 1}
    eval_at = {i - 1 for i in cfg.eval_steps}
    stats = {"steps": max_steps, "intervals": [], "evals": [], "checkpoints": [], "sh_degree_switches": [],
             "refine_steps": 0, "reset_steps": 0}
    last_deg = -1
    global_tic = time.time()
    torch.cuda.synchronize()
    tic, tic_step = time.perf_counter(), 0
    try:
        for step in range(max_steps):
            data, everyone = next_batch()
            if grad_sync is not None and hasattr(grad_sync, "set_views"):
                # distributed.GatherRowsSync: every rank runs the projection backward over the cameras of ALL
                # ranks (row r = the camera rank r renders in this step)
                if B != 1:
                    raise ValueError("the row exchange renders one view per rank per step (cfg.batch_size = 1)")
                grad_sync.set_views(torch.stack([trainset[i]["camtoworld"].to(device) for i in everyone]),
                                    torch.stack([trainset[i]["K"].to(device) for i in everyone]))
            def batch_of(key):       # the DataLoader's collation; one frame (cfg.batch_size 1) is used as it is
                return data[0][key] if len(data) == 1 else torch.cat([d[key] for d in data])

            c2w, Ks = batch_of("camtoworld"), batch_of("K")
            pixels = batch_of("pixels")
            masks = batch_of("mask") if "mask" in data[0] else None
            extra = {}
            if cfg.depth_loss:                                                         # runner.py:450-452
                extra = dict(depth_points=batch_of("points"), depth_gt=batch_of("depths"),
                             depth_lambda=cfg.depth_lambda, scene_scale=scene_scale)
            deg = min(step // max(cfg.sh_degree_interval, 1), cfg.sh_degree)
            if deg != last_deg:
                stats["sh_degree_switches"].append((step, deg))
                last_deg = deg
            hook = None
            if step in save_at and out_dir is not None:
                def hook(loss, info, step=step):                                       # runner.py:592-637
                    mem = torch.cuda.max_memory_allocated() / 1024 ** 3
                    st = {"mem": mem, "ellipse_time": time.time() - global_tic, "num_GS": len(splats["means"])}
                    (out_dir / "stats" / f"train_step{step:04d}_rank{world_rank}.json").write_text(json.dumps(st))
                    path = gs_io.save_checkpoint(splats, step, out_dir / "ckpts", world_rank)
                    stats["checkpoints"].append(str(path))
                    if cfg.save_final_ply:
                        gs_io.export_ply(splats, out_dir / "ckpts" / f"splats_{step}.ply")
            if isinstance(strategy, DefaultStrategy) and strategy.mutates_params(step):
                stats["reset_steps" if (step % strategy.reset_every == 0 and step > 0) else "refine_steps"] += 1
            loss, info = train_step(
                splats, opt, c2w, Ks, pixels, step, rc, ssim_lambda=cfg.ssim_lambda, grad_sync=grad_sync,
                strategy=strategy, strategy_state=strategy_state, opacity_reg=cfg.opacity_reg,
                scale_reg=cfg.scale_reg, random_background=cfg.random_background, masks=masks,
                before_update=hook, viewmats_campos=(data[0].get("viewmats_campos") if len(data) == 1 else None), **extra)
            for sch in schedulers:                                                     # runner.py:687-689
                sch.step()
            if step in eval_at and valset is not None:                                 # runner.py:692-694
                ev = evaluate(splats, valset, cfg, device)
                ev["step"] = step
                stats["evals"].append(ev)
                if out_dir is not None and world_rank == 0:
                    (out_dir / "stats" / f"val_step{step:04d}.json").write_text(json.dumps(ev))
            if (step + 1) % progress_every == 0 or step == max_steps - 1:
                lv = float(loss)                     # (the one host read per interval; also drains the queue)
                now = time.perf_counter()
                rec = {"step": step, "loss": lv, "num_GS": len(splats["means"]), "sh_degree": deg,
                       "ms_per_step": (now - tic) / max(step + 1 - tic_step, 1) * 1e3,
                       "lr_means": means_opt.param_groups[0]["lr"]}
                stats["intervals"].append(rec)
                tic, tic_step = now, step + 1
                if progress is not None:
                    progress(step, rec)
    finally:
        if fused_here:
            _R.set_backward_optimizer(None)
    torch.cuda.synchronize()
    stats["seconds"] = time.time() - global_tic
    stats["final_lr_means"] = means_opt.param_groups[0]["lr"]
    stats["num_GS"] = len(splats["means"])
    stats["peak_mem_gib"] = torch.cuda.max_memory_allocated() / 1024 ** 3
    return stats
