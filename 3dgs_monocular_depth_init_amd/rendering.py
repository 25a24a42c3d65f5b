"""`rasterization()` -- the operator boundary of the hot path.

Drop-in for the third-party call the reference makes at
/root/reference/gs_init_compare/runner.py:341-362
(`gsplat.rendering.rasterization`, gsplat==1.5.2 pinned at setup.py:15), with
the same keyword names, argument meaning, return triple
`(render_colors [C,H,W,D], render_alphas [C,H,W,1], meta)` and the `meta`
entries the reference's callers consume (runner.py:497-503, 639-647, 663:
`means2d` as an autograd intermediate that accepts `.retain_grad()`,
`radii`, `width`, `height`, `n_cameras`, `gaussian_ids`).

All arithmetic runs in hand-written HIP kernels (libgsrast.so, C ABI in
include/gsrast.h). PyTorch only owns memory, the stream and the autograd
graph. There is no CPU / eager fallback: tensors must live on a ROCm device
and the library must be built, otherwise the call raises.
"""
from __future__ import annotations

import math
import os
import time
from typing import Dict, Optional, Tuple

import torch
from torch import Tensor

from . import _lib
from ._lib import current_stream as _raw_stream
from ._lib import GsrastError, call, ptr

TILE = 16
GRAD_ROW = 16
PACKED_ROW = 9                                 # GSR_PACKED_ROW
ACT_EXP_SCALES, ACT_SIGMOID_OPAC = 1, 2      # GSR_ACT_* of include/gsrast.h
GR_MEAN2D, GR_CONIC, GR_OPAC, GR_COLOR, GR_ABS = 0, 2, 5, 6, 12


def _stream() -> int:
    return _raw_stream()


class GradArena:
    """One flat fp32 buffer for the gradients of all Gaussian parameters.

    When registered (`set_grad_arena`), the projection backward -- which, with
    fused activations, produces ALL six parameter gradients -- writes them into
    16-byte-aligned segments of this buffer instead of six separate tensors, so
    the multi-GPU path all-reduces ONE 59*N-float message (236 MB at 1M
    Gaussians) and autograd hands the parameters views of it without copies.
    Each segment is handed out at most once per generation (`reset()`), so a
    second backward in the same step falls back to fresh tensors."""

    ORDER = ("shN", "sh0", "means", "quats", "scales", "opacities")

    def __init__(self, shapes: Dict[str, Tuple[int, ...]], device):
        self.shapes = {k: tuple(v) for k, v in shapes.items()}
        self.offsets = {}
        off = 0
        for name in self.ORDER:
            if name not in self.shapes:
                continue
            self.offsets[name] = off
            n = 1
            for d in self.shapes[name]:
                n *= d
            off += (n + 3) // 4 * 4
        self.flat = torch.zeros(off, dtype=torch.float32, device=device)
        self._handed = set()

    def reset(self) -> None:
        self._handed.clear()

    def take(self, name: str, shape) -> Optional[Tensor]:
        if name in self._handed or self.shapes.get(name) != tuple(shape):
            return None
        self._handed.add(name)
        return self.view(name)

    def view(self, name: str) -> Tensor:
        n = 1
        for d in self.shapes[name]:
            n *= d
        o = self.offsets[name]
        return self.flat[o:o + n].view(self.shapes[name])

    def owns(self, name: str, t: Optional[Tensor]) -> bool:
        return (t is not None and name in self.offsets and t.is_contiguous()
                and t.data_ptr() == self.flat.data_ptr() + 4 * self.offsets[name]
                and tuple(t.shape) == self.shapes[name])


_GRAD_ARENA: Optional[GradArena] = None

# Optimizer in backward (optim.FusedAdam.fuse_into_backward): when set, the projection
# backward applies the Adam update itself (gsr_project_bwd_adam) and returns no parameter
# gradients. `claim(tensors)` returns the launch arguments for exactly these six parameter
# tensors, or None when the fused step does not apply to this call.
_BACKWARD_OPTIMIZER = None


def set_backward_optimizer(obj) -> None:
    global _BACKWARD_OPTIMIZER
    _BACKWARD_OPTIMIZER = obj


# Exchange of view-space gradient rows between view-parallel ranks
# (distributed.GatherRowsSync): when set, the projection backward hands its rows to
# `exchange(rows, radii)` and runs over the cameras of ALL ranks it gets back.
_ROW_EXCHANGE = None


def set_row_exchange(obj) -> None:
    global _ROW_EXCHANGE
    _ROW_EXCHANGE = obj


def set_grad_arena(arena: Optional[GradArena]) -> None:
    global _GRAD_ARENA
    _GRAD_ARENA = arena


def _grad_out(name: str, like: Tensor) -> Tensor:
    if _GRAD_ARENA is not None and _GRAD_ARENA.flat.device == like.device:
        t = _GRAD_ARENA.take(name, like.shape)
        if t is not None:
            return t
    return torch.empty_like(like)


def _check_cuda(*tensors: Optional[Tensor]) -> None:
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise _lib.GsrastError(
                "rasterization(): tensors must be on a ROCm device; this build has no CPU path"
            )


def _f32c(t: Tensor) -> Tensor:
    if t.dtype != torch.float32:
        t = t.float()
    return t.contiguous()


# --------------------------------------------------------------------------- #
# A3 + A4: projection fused with SH evaluation
# --------------------------------------------------------------------------- #
class _ProjectSH(torch.autograd.Function):
    """(means, quats, scales, sh) -> (radii, means2d, depths, conics, comps, colors).

    sh_a / sh_b: either (colors [N,K,3], None) -- gsplat's concatenated layout --
    or (sh0 [N,1,3], shN [N,K-1,3]) -- the reference's own parameter layout
    (runner.py:112-115), which avoids the torch.cat of runner.py:338.
    """

    @staticmethod
    def forward(ctx, means, quats, scales, opacities, sh_a, sh_b, viewmats, Ks, campos, cfg):
        (width, height, eps2d, near, far, radius_clip, calc_comp, sh_degree, color_stride,
         depth_channel, activations, tile_w, tile_h) = cfg
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        opac_act = (torch.empty(N, dtype=torch.float32, device=dev)
                    if activations & ACT_SIGMOID_OPAC else None)
        tile_counts = (torch.empty(C * tile_w * tile_h, dtype=torch.int32, device=dev)
                       if tile_w > 0 else None)
        records = (torch.empty(C * N, GRAD_ROW, dtype=torch.float32, device=dev)
                   if sh_degree >= 0 and opacities is not None else None)
        radii = torch.empty(C, N, 2, dtype=torch.int32, device=dev)
        means2d = torch.empty(C, N, 2, dtype=torch.float32, device=dev)
        depths = torch.empty(C, N, dtype=torch.float32, device=dev)
        conics = torch.empty(C, N, 3, dtype=torch.float32, device=dev)
        comps = torch.empty(C, N, dtype=torch.float32, device=dev) if calc_comp else None
        colors = None
        sh0_ptr = shN_ptr = None
        sh0_stride = shN_stride = 0
        if sh_degree >= 0:
            colors = torch.empty(C, N, color_stride, dtype=torch.float32, device=dev)
            if sh_b is None:
                K = sh_a.shape[1]
                sh0_ptr, shN_ptr = sh_a.data_ptr(), sh_a.data_ptr() + 12
                sh0_stride = shN_stride = 3 * K
            else:
                K = 1 + sh_b.shape[1]
                sh0_ptr, shN_ptr = sh_a.data_ptr(), sh_b.data_ptr()
                sh0_stride, shN_stride = 3, 3 * (K - 1)
            if (sh_degree + 1) ** 2 > K:
                raise ValueError(f"sh_degree {sh_degree} needs {(sh_degree + 1) ** 2} coefficients, got {K}")
        call("gsr_project_fwd", C, N, ptr(means), ptr(quats), ptr(scales), ptr(opacities),
             ptr(viewmats), ptr(Ks), ptr(campos), width, height, eps2d, near, far, radius_clip,
             int(calc_comp), sh_degree, sh0_ptr, sh0_stride, shN_ptr, shN_stride, ptr(radii),
             ptr(means2d), ptr(depths), ptr(conics), ptr(comps), ptr(colors), color_stride,
             depth_channel, activations, ptr(opac_act), tile_w, tile_h, ptr(tile_counts),
             ptr(records), _stream())
        ctx.cfg = cfg
        ctx.split = sh_b is not None
        ctx.raw_opacities = opacities      # identity only (optimizer in backward); not read
        ctx.save_for_backward(means, quats, scales, sh_a, sh_b, viewmats, Ks, campos, radii,
                              opac_act)
        if comps is None:
            comps = torch.empty(0, device=dev)
        if colors is None:
            colors = torch.empty(0, device=dev)
        if opac_act is None:
            opac_act = torch.empty(0, device=dev)
        if tile_counts is None:
            tile_counts = torch.empty(0, dtype=torch.int32, device=dev)
        if records is None:
            records = torch.empty(0, device=dev)
        ctx.mark_non_differentiable(radii, tile_counts, records)
        ctx.set_materialize_grads(False)
        return radii, means2d, depths, conics, comps, colors, opac_act, tile_counts, records

    @staticmethod
    def backward(ctx, _v_radii, v_means2d, v_depths, v_conics, v_comps, v_colors, v_opac_act,
                 _v_tc, _v_rec):
        (width, height, eps2d, near, far, radius_clip, calc_comp, sh_degree, color_stride,
         depth_channel, activations, tile_w, tile_h) = ctx.cfg
        (means, quats, scales, sh_a, sh_b, viewmats, Ks, campos, radii,
         opac_act) = ctx.saved_tensors
        C, N = viewmats.shape[0], means.shape[0]
        dev = means.device
        rows, fast = _rows_from_grads(C, N, v_means2d, v_conics,
                                      v_colors if sh_degree >= 0 else None, color_stride)
        v_opacities = None
        if activations & ACT_SIGMOID_OPAC:
            # the kernel sums rows[.][GR_OPAC] over cameras and applies o(1-o)
            if not fast and v_opac_act is not None:
                rows.view(C, N, GRAD_ROW)[0, :, GR_OPAC] = v_opac_act.reshape(N)
        if v_depths is not None:
            v_depths = _f32c(v_depths)
        if v_comps is not None and calc_comp:
            v_comps = _f32c(v_comps)
        else:
            v_comps = None
        ex = _ROW_EXCHANGE
        fusable = (ctx.split and sh_degree >= 0 and sh_b is not None and sh_b.shape[1] == 15
                   and activations == (ACT_EXP_SCALES | ACT_SIGMOID_OPAC)
                   and all(ctx.needs_input_grad[:6]))
        chunks = None
        if ex is not None:
            if (not fusable or C != 1 or v_depths is not None or v_comps is not None
                    or depth_channel >= 0):
                raise NotImplementedError(
                    "row exchange: one colour-only view per rank per step, on the six raw parameters")
            # every rank's 36-byte rows, and every rank's camera: the sum over the views is taken
            # inside the projection backward, identically on all ranks. The exchange is pipelined
            # over Gaussian ranges: all-gather(k+1) runs while the backward of range k does.
            chunks, vm_all, Ks_all, campos_all, W = ex.exchange(rows, radii, N)
        bo = _BACKWARD_OPTIMIZER
        if bo is not None and fusable:
            args = bo.claim((means, quats, scales, ctx.raw_opacities, sh_a, sh_b))
            if args is not None:
                P, M, V, ss, bc2, beta1, beta2, eps = args
                if chunks is not None:
                    if getattr(bo, "step_extras", None) is not None:
                        raise NotImplementedError("row exchange: the mcmc step extras are single-rank (runner.train_step keeps "
                                                  "the reference's order on view-parallel ranks)")
                    PA = type(P)
                    # bytes per Gaussian of means, quats, scales, opacities, sh0, shN
                    row_bytes = (12, 16, 12, 4, 12, 180)
                    for a0, n, rows_k, wait in chunks:
                        wait()
                        off = lambda arr: PA(*[(arr[t] or 0) + a0 * row_bytes[t] for t in range(6)])
                        call("gsr_project_bwd_adam", W, n, ptr(vm_all), ptr(Ks_all), ptr(campos_all),
                             width, height, eps2d, sh_degree, None, ptr(rows_k), getattr(ex, "row_stride", PACKED_ROW), None,
                             None, -1, activations, opac_act.data_ptr() + 4 * a0, off(P), off(M),
                             off(V), ss, bc2, beta1, beta2, eps, _stream())
                    return (None,) * 10
                extras = bo.take_step_extras() if hasattr(bo, "take_step_extras") else None
                if extras is not None:       # mcmc noise / regulariser gradients / strategy statistics in the same pass
                    import ctypes
                    st = extras.get("stats")
                    ex = _lib.StepExtras(ptr(extras.get("noise")), extras["noise_scale"], extras["opacity_reg"],
                                         extras["scale_reg"], ptr(st[0]) if st else None, ptr(st[1]) if st else None,
                                         ptr(st[2]) if st else None, st[3] if st else 0.0, st[4] if st else 0.0,
                                         st[5] if st else 0.0, int(bool(st[6])) if st else 0)
                    call("gsr_project_bwd_adam_ex", C, N, ptr(viewmats), ptr(Ks), ptr(campos), width,
                         height, eps2d, sh_degree, ptr(radii), ptr(rows), GRAD_ROW, ptr(v_depths),
                         ptr(v_comps), depth_channel, activations, ptr(opac_act), P, M, V, ss, bc2,
                         beta1, beta2, eps, ctypes.byref(ex), _stream())
                else:
                    call("gsr_project_bwd_adam", C, N, ptr(viewmats), ptr(Ks), ptr(campos), width,
                         height, eps2d, sh_degree, ptr(radii), ptr(rows), GRAD_ROW, ptr(v_depths),
                         ptr(v_comps), depth_channel, activations, ptr(opac_act), P, M, V, ss, bc2,
                         beta1, beta2, eps, _stream())
                return (None,) * 10
        if chunks is not None:
            # gathered rows, optimizer NOT fused (a step on which the strategy must see the
            # gradients before the update): the W-view gradients are written out, identical on
            # every rank, and the optimizer steps afterwards as in the reference's order
            v_opacities = _grad_out("opacities", opac_act)
            v_means, v_quats, v_scales = (_grad_out("means", means), _grad_out("quats", quats),
                                          _grad_out("scales", scales))
            v_sh_a, v_sh_b = _grad_out("sh0", sh_a), _grad_out("shN", sh_b)
            for a0, n, rows_k, wait in chunks:
                wait()
                call("gsr_project_bwd_rows", W, n, means.data_ptr() + 12 * a0, quats.data_ptr() + 16 * a0,
                     scales.data_ptr() + 12 * a0, ptr(vm_all), ptr(Ks_all), ptr(campos_all), width,
                     height, eps2d, sh_degree, sh_a.data_ptr() + 12 * a0, 3, sh_b.data_ptr() + 180 * a0,
                     45, None, ptr(rows_k), getattr(ex, "row_stride", PACKED_ROW), v_means.data_ptr() + 12 * a0,
                     v_quats.data_ptr() + 16 * a0, v_scales.data_ptr() + 12 * a0,
                     v_sh_a.data_ptr() + 12 * a0, 3, v_sh_b.data_ptr() + 180 * a0, 45, 16, activations,
                     opac_act.data_ptr() + 4 * a0, v_opacities.data_ptr() + 4 * a0, _stream())
            return v_means, v_quats, v_scales, v_opacities, v_sh_a, v_sh_b, None, None, None, None
        if activations & ACT_SIGMOID_OPAC:
            v_opacities = _grad_out("opacities", opac_act)
        v_means = _grad_out("means", means)
        v_quats = _grad_out("quats", quats)
        v_scales = _grad_out("scales", scales)
        v_sh_a = v_sh_b = None
        v_sh0_ptr = v_shN_ptr = None
        sh0_ptr = shN_ptr = None
        sh0_stride = shN_stride = v0_stride = vN_stride = 0
        K = 0
        if sh_degree >= 0:
            if not ctx.split:
                K = sh_a.shape[1]
                v_sh_a = torch.empty_like(sh_a)
                sh0_ptr, shN_ptr = sh_a.data_ptr(), sh_a.data_ptr() + 12
                v_sh0_ptr, v_shN_ptr = v_sh_a.data_ptr(), v_sh_a.data_ptr() + 12
                sh0_stride = shN_stride = v0_stride = vN_stride = 3 * K
            else:
                K = 1 + sh_b.shape[1]
                v_sh_a, v_sh_b = _grad_out("sh0", sh_a), _grad_out("shN", sh_b)
                sh0_ptr, shN_ptr = sh_a.data_ptr(), sh_b.data_ptr()
                v_sh0_ptr, v_shN_ptr = v_sh_a.data_ptr(), v_sh_b.data_ptr()
                sh0_stride, shN_stride = 3, 3 * (K - 1)
                v0_stride, vN_stride = 3, 3 * (K - 1)
        call("gsr_project_bwd", C, N, ptr(means), ptr(quats), ptr(scales), ptr(viewmats), ptr(Ks),
             ptr(campos), width, height, eps2d, sh_degree, sh0_ptr, sh0_stride, shN_ptr,
             shN_stride, ptr(radii), None, None, ptr(rows), ptr(v_depths), ptr(v_comps),
             depth_channel if sh_degree >= 0 else -1, ptr(v_means), ptr(v_quats), ptr(v_scales),
             v_sh0_ptr, v0_stride, v_shN_ptr, vN_stride, K, activations, ptr(opac_act),
             ptr(v_opacities), _stream())
        return v_means, v_quats, v_scales, v_opacities, v_sh_a, v_sh_b, None, None, None, None


def _rows_from_grads(C: int, N: int, v_means2d, v_conics, v_colors, color_stride: int) -> Tensor:
    """Return the [C*N,16] gradient-row buffer the projection backward reads.

    Fast path: the incoming grads are the strided views _Rasterize.backward
    returned over one row buffer -> reuse it, no copy. Otherwise (user-built
    grads, hooks) pack them into a fresh buffer with torch ops on the device.
    """
    base = None
    for v, off, width in ((v_means2d, GR_MEAN2D, 2), (v_conics, GR_CONIC, 3), (v_colors, GR_COLOR, None)):
        if v is None:
            continue
        b = getattr(v, "_base", None)
        ok = (
            b is not None and b.dim() == 2 and b.shape == (C * N, GRAD_ROW) and b.is_contiguous()
            and v.stride()[-1] == 1 and v.stride()[-2] == GRAD_ROW
            and v.storage_offset() == b.storage_offset() + off
            and (base is None or base is b)
        )
        if not ok:
            base = None
            break
        base = b
    else:
        if base is not None:
            return base, True
    any_v = next((v for v in (v_means2d, v_conics, v_colors) if v is not None), None)
    dev = any_v.device if any_v is not None else torch.device("cuda")
    rows = torch.zeros(C * N, GRAD_ROW, dtype=torch.float32, device=dev)
    if v_means2d is not None:
        rows[:, GR_MEAN2D:GR_MEAN2D + 2] = v_means2d.reshape(C * N, 2)
    if v_conics is not None:
        rows[:, GR_CONIC:GR_CONIC + 3] = v_conics.reshape(C * N, 3)
    if v_colors is not None:
        w = v_colors.shape[-1]
        rows[:, GR_COLOR:GR_COLOR + w] = v_colors.reshape(C * N, w)
    return rows, False


@torch.no_grad()
def inverse4x4(mats: Tensor, translation_of: str = "inverse") -> Tuple[Tensor, Tensor]:
    """Batched 4x4 inverse on the device in ONE launch. Returns (inverse [C,4,4],
    translation column [C,3] of the inverse or of the input): either way the camera
    positions, for world-to-camera or camera-to-world input."""
    mats = _f32c(mats)
    C = mats.shape[0]
    out = torch.empty_like(mats)
    tr = torch.empty(C, 3, dtype=torch.float32, device=mats.device)
    if translation_of == "input":
        call("gsr_inverse4x4", C, ptr(mats), ptr(out), ptr(tr), None, _stream())
    else:
        call("gsr_inverse4x4", C, ptr(mats), ptr(out), None, ptr(tr), _stream())
    return out, tr


# --------------------------------------------------------------------------- #
# A5: tile intersection lists (not differentiable)
# --------------------------------------------------------------------------- #
@torch.no_grad()
def isect_tiles_sorted(means2d: Tensor, radii: Tensor, depths: Tensor, tile_w: int, tile_h: int,
                       want_tiles_per_gauss: bool = False, tile_counts: Optional[Tensor] = None,
                       capacity: Optional[int] = None, conics: Optional[Tensor] = None,
                       opacities: Optional[Tensor] = None, tight: bool = False, want_keys: bool = True):
    """Returns (tile_offsets [n_tiles+1] int32, tile_order [n_tiles] int32 (longest list
    first), flatten_ids [I] int32, isect_keys [I] int64 (sorted per tile), tiles_per_gauss or
    None, pair_ids [I] int32 or None).

    With `conics` [C,N,3] and `opacities` ([N] or [C,N]) the pair words the compositing
    kernels read are produced as well (flatten id | 4-bit quadrant mask << 28); `tight=True`
    additionally drops the pairs whose ellipse alpha >= 1/255 misses the tile (bucketed
    builder only; gsplat's rectangle rule is the default)."""
    C, N = depths.shape
    dev = depths.device
    n_tiles = C * tile_w * tile_h
    with_pairs = conics is not None and opacities is not None
    if with_pairs and (tile_counts is None or tile_counts.numel() == 0) and not want_tiles_per_gauss \
            and bucket_layout_ok(C, N, tile_w, tile_h):
        return _isect_bucketed(means2d, radii, depths, tile_w, tile_h, capacity, conics, opacities, tight, want_keys)
    counted = tile_counts is not None and tile_counts.numel() == n_tiles and not want_tiles_per_gauss
    if not counted:
        tile_counts = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    tile_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
    tile_order = torch.empty(n_tiles, dtype=torch.int32, device=dev)
    tpg = torch.empty(C, N, dtype=torch.int32, device=dev) if want_tiles_per_gauss else None
    st = _stream()
    if not counted:      # otherwise the projection kernel already counted (fused A5 pass 1)
        call("gsr_isect_count", C, N, ptr(means2d), ptr(radii), tile_w, tile_h, ptr(tpg),
             ptr(tile_counts), st)
    call("gsr_isect_scan", n_tiles, ptr(tile_counts), ptr(tile_offsets), ptr(tile_order), st)
    n_isects = int(tile_offsets[-1].item())          # the one host sync of the step
    keys = torch.empty(max(n_isects, 1), dtype=torch.int64, device=dev)
    flatten_ids = torch.empty(max(n_isects, 1), dtype=torch.int32, device=dev)
    pair_ids = None
    if n_isects > 0:
        call("gsr_isect_emit", C, N, ptr(means2d), ptr(radii), ptr(depths), tile_w, tile_h,
             ptr(tile_offsets), ptr(tile_counts), ptr(keys), n_isects, st)
        big_list = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        call("gsr_tile_sort", n_tiles, ptr(tile_offsets), ptr(tile_order), ptr(keys),
             ptr(flatten_ids), ptr(big_list), st)
    if with_pairs:
        pair_ids = torch.empty(max(n_isects, 1), dtype=torch.int32, device=dev)
        if n_isects > 0:
            call("gsr_pair_masks", C, N, tile_w, tile_h, ptr(tile_offsets), ptr(flatten_ids),
                 ptr(means2d), ptr(conics), ptr(opacities), int(opacities.dim() == 2), ptr(pair_ids), st)
        pair_ids = pair_ids[:n_isects]
    return tile_offsets, tile_order, flatten_ids[:n_isects], keys[:n_isects], tpg, pair_ids


def bucket_layout_ok(C: int, N: int, tile_w: int, tile_h: int) -> bool:
    """The bucketed tile-list builder needs <= 8192 buckets (LDS histogram) and C*N < 2^25
    (25 bits of the composite key hold the pair id, 5 its quadrant mask and clamp flag)."""
    return C * tile_h * ((tile_w + 7) // 8) <= 8192 and C * N < (1 << 25)


# Compositing work order: True (default) = longest TILE first, one more (single-workgroup, 8 us)
# launch after the sort pass; False = longest BUCKET first, from the emit pass (no launch of its
# own). Measured on c4 inside one gpurun call (profiles/r03_ab_tile_order.log): the bucket-granular
# order costs the compositing kernels 45 us (forward 0.145 -> 0.160 ms, backward 0.405 -> 0.437 ms):
# neighbouring tiles have lists of similar length and land on the same CUs.
EXACT_TILE_ORDER = os.environ.get("GSR_EXACT_TILE_ORDER", "1") == "1"


# The backward's [C*N, 16] gradient rows are cleared by the compositing forward on the side (True)
# or by a fill launch in front of the backward (False).
CLEAR_ROWS_IN_FORWARD = os.environ.get("GSR_CLEAR_ROWS_IN_FORWARD", "1") == "1"


class _IsectState:
    """Per-device memory of the last frame's intersection count: lets the next frame
    size its buffers and launch emit / sort / compositing WITHOUT waiting for its own
    count to reach the host (the wait is deferred until after those launches, when the
    GPU has work queued; an overflowing frame is simply rebuilt with larger buffers)."""
    capacity: Dict[int, int] = {}
    pinned: Dict[int, Tensor] = {}       # one reusable pinned int32[8] per device
    pinned_np: Dict[int, object] = {}    # its numpy view (polled by _PendingIsect.resolve)
    seq: int = 0                         # sequence number of the last frame whose totals were asked for
    poll_fallbacks: int = 0              # frames whose totals only arrived with a stream synchronisation
    # (device, n_buckets) -> [3, n_buckets] int32: bucket counts | emit cursor | listed pairs. The
    # counts are zero between frames (the sort kernel clears them), the other two rows are cleared
    # by the count kernel: no memset launches
    scratch: Dict[Tuple[int, int], Tensor] = {}
    wg_hist: Dict[Tuple[int, int], Tensor] = {}   # [256, n_buckets]: per-workgroup counts, count pass -> emit pass


# How the host learns a frame's totals. True: the sort kernel stores them into pinned host memory followed by the
# frame's sequence number (system-scope release) and the host polls that word -- no event packet in the queue (an
# event record between the tile-list kernels and the compositing forward leaves the GPU idle for ~6 us per frame,
# `profiles/r04c_kernel_stats.csv` traces), and the totals arrive as soon as the last bucket's workgroup is through.
# False: an event recorded behind the sort, as in rounds 2-4. Polling falls back to a stream synchronisation if the
# word has not arrived after POLL_DEADLINE_S (pinned memory the device cannot write coherently).
POLL_TOTALS = os.environ.get("GSR_POLL_TOTALS", "1") == "1"
POLL_DEADLINE_S = 0.05


class _PendingIsect:
    def __init__(self, dev, n_host, event, capacity, seq=0):
        self.dev, self.n_host, self.event, self.capacity, self.seq = dev, n_host, event, capacity, seq

    def resolve(self) -> Tuple[int, bool]:
        """(n_isects, overflowed). Blocks until the tile-list kernels have delivered the frame's totals (the
        compositing forward has been queued behind them by then)."""
        if self.event is not None:
            self.event.synchronize()
            slots, n = int(self.n_host[0]), int(self.n_host[1])   # reserved slots >= listed pairs
        else:
            words = _IsectState.pinned_np[self.dev.index]          # numpy view of n_host: [.., .., slots, pairs, seq]
            seq, deadline = self.seq, None
            while int(words[4]) != seq:
                if deadline is None:
                    deadline = time.perf_counter() + POLL_DEADLINE_S
                elif time.perf_counter() > deadline:
                    torch.cuda.current_stream(self.dev).synchronize()
                    if int(words[4]) != seq:
                        raise GsrastError("tile lists: the sort pass never delivered this frame's totals")
                    _IsectState.poll_fallbacks += 1
                    break
            slots, n = int(words[2]), int(words[3])
        _IsectState.capacity[self.dev.index] = int(slots * 1.25) + 8192
        return n, slots > self.capacity


@torch.no_grad()
def _isect_bucketed(means2d, radii, depths, tile_w, tile_h, capacity: Optional[int], conics, opacities,
                    tight: bool, want_keys: bool = True):
    """isect_bucket.hip: buckets of 8 tiles, LDS histograms, one LDS sort per bucket.
    With `capacity` the call never blocks: returns full-capacity buffers and a
    _PendingIsect to resolve after the consumer kernels have been launched."""
    C, N = depths.shape
    dev = depths.device
    n_tiles = C * tile_w * tile_h
    n_buckets = C * tile_h * ((tile_w + 7) // 8)
    st = _stream()
    key = (dev.index, n_buckets)
    if capacity is None:                     # blocking path (first frame, overflow rebuild)
        _IsectState.scratch.pop(key, None)   # fresh buffers; a failed frame may have left them dirty
    sc = _IsectState.scratch.get(key)
    if sc is None:
        sc = _IsectState.scratch[key] = torch.zeros(3, n_buckets, dtype=torch.int32, device=dev)
    counts, cursor, real = sc[0], sc[1], sc[2]
    wg = _IsectState.wg_hist.get(key)
    if wg is None:
        wg = _IsectState.wg_hist[key] = torch.empty(256, n_buckets, dtype=torch.int32, device=dev)
    per_cam = int(opacities.dim() == 2)
    n_host = _IsectState.pinned.get(dev.index)
    if n_host is None:
        # [0] reserved slots (emit pass), [1] listed pairs (sort pass): read after the event; [2..4] = slots, pairs and
        # the frame's sequence number, stored by the sort pass for the polling host
        n_host = _IsectState.pinned[dev.index] = torch.zeros(8, dtype=torch.int32, pin_memory=True)
        _IsectState.pinned_np[dev.index] = n_host.numpy()
    poll = POLL_TOTALS and capacity is not None
    seq = 0
    if poll:         # (never reused while its predecessor may still sit in the pinned word: strictly increasing, wraps at 2^31)
        seq = _IsectState.seq = (_IsectState.seq % 0x7fffffff) + 1
    try:
        offsets = torch.empty(n_buckets + 1, dtype=torch.int32, device=dev)
        order = torch.empty(n_buckets, dtype=torch.int32, device=dev)
        call("gsr_bucket_count", C, N, ptr(means2d), ptr(radii), tile_w, tile_h, ptr(counts), ptr(cursor),
             ptr(real), 1, ptr(wg), st)
        if capacity is None:
            slots = int(counts.sum().item())             # blocking: first frame / explicit request
            cap = max(slots, 1)
            _IsectState.capacity[dev.index] = int(slots * 1.25) + 8192
        else:
            cap = max(int(capacity), 1)
        keys = torch.empty(cap, dtype=torch.int64, device=dev)
        # gsplat's `isect_ids` (the sorted keys) are meta data only: nothing downstream reads them, and the
        # training step does not ask for them (18 MB less to write per step at c4)
        keys_sorted = torch.empty(cap, dtype=torch.int64, device=dev) if want_keys else None
        flatten_ids = torch.empty(cap, dtype=torch.int32, device=dev)
        pair_ids = torch.empty(cap, dtype=torch.int32, device=dev)
        tile_offsets = torch.empty(n_tiles + 1, dtype=torch.int32, device=dev)
        tile_order = torch.empty(n_tiles, dtype=torch.int32, device=dev)
        # the two totals land in pinned host memory straight from the kernels: no copy launches
        call("gsr_bucket_emit", C, N, ptr(means2d), ptr(radii), ptr(depths), ptr(conics), ptr(opacities),
             per_cam, tile_w, tile_h, int(tight), ptr(counts), ptr(cursor), ptr(real), ptr(offsets),
             ptr(order), None if EXACT_TILE_ORDER else ptr(tile_order), n_host.data_ptr(), ptr(keys), cap, ptr(wg), st)
        call("gsr_bucket_sort", C, tile_w, tile_h, ptr(offsets), ptr(order), ptr(real), ptr(keys),
             ptr(keys_sorted), ptr(flatten_ids), ptr(pair_ids), ptr(tile_offsets),
             ptr(tile_order) if EXACT_TILE_ORDER else None, cap, ptr(counts), n_host.data_ptr() + 4,
             n_host.data_ptr() + 8 if poll else None, seq, st)
        ev = None
        if not poll:
            ev = torch.cuda.Event()
            ev.record()
        pending = _PendingIsect(dev, n_host, ev, cap, seq)
    except BaseException:
        _IsectState.scratch.pop(key, None)     # the kernels that clear it may not have run
        raise
    if capacity is None:
        n_isects, _ = pending.resolve()
        return (tile_offsets, tile_order, flatten_ids[:n_isects],
                keys_sorted[:n_isects] if want_keys else None, None, pair_ids[:n_isects])
    return tile_offsets, tile_order, flatten_ids, keys_sorted, pending, pair_ids


# --------------------------------------------------------------------------- #
# A6 / A7: compositing
# --------------------------------------------------------------------------- #
_L1_PARTIALS: Dict[Tuple[int, int], Tensor] = {}     # (device, tiles) -> per-tile partial sums of gsr_rasterize_fwd_l1


class _Rasterize(torch.autograd.Function):
    @staticmethod
    def forward(ctx, means2d, conics, colors, opacities, backgrounds, tile_offsets, tile_order,
                pair_ids, records, cfg, l1_target=None, planar=False):
        """l1_target ([C,H,W,3], the step's target image): the FIRST output is then the scalar mean |render - target|
        instead of the render (gsr_rasterize_fwd_l1: the L1 loss taken inside the compositing forward).
        planar: the render is stored in planes ([C,CH,H,W] memory) and returned as its [C,H,W,CH] view -- the layout the
        fused L1 + SSIM loss reads / writes a third of the lines in (gsr_rasterize_fwd_planar / _bwd_planar)."""
        width, height, tile_w, tile_h, CH, absgrad = cfg
        C, N = means2d.shape[0], means2d.shape[1]
        dev = means2d.device
        color_stride = colors.shape[-1]
        per_cam = int(opacities.dim() == 2)
        if records is None or records.numel() != C * N * GRAD_ROW:
            # caller-supplied colours / opacities: pack the compositing records here
            records = torch.empty(C * N, GRAD_ROW, dtype=torch.float32, device=dev)
            call("gsr_pack_records", C, N, CH, ptr(means2d), ptr(conics), ptr(colors), color_stride,
                 ptr(opacities), per_cam, ptr(records), _stream())
        planar = bool(planar) and l1_target is None
        render_colors = (torch.empty(C, CH, height, width, dtype=torch.float32, device=dev) if planar
                         else torch.empty(C, height, width, CH, dtype=torch.float32, device=dev))
        render_alphas = torch.empty(C, height, width, 1, dtype=torch.float32, device=dev)
        last_ids = torch.empty(C, height, width, dtype=torch.int32, device=dev)
        # the backward's gradient rows are cleared by the forward kernel itself, on the side (it is
        # not HBM-bound): no 64 MB fill launch in front of the backward
        rows = None
        if CLEAR_ROWS_IN_FORWARD and C * N > 0 and C * tile_w * tile_h > 0 and any(ctx.needs_input_grad[:4]):
            rows = torch.empty(C * N, GRAD_ROW, dtype=torch.float32, device=dev)
        ctx.l1_grad = None
        if l1_target is not None:
            assert CH == 3 and tuple(l1_target.shape) == (C, height, width, 3) and l1_target.is_contiguous()
            n_tiles = C * tile_w * tile_h
            part = _L1_PARTIALS.get((dev.index, n_tiles))
            if part is None:
                part = _L1_PARTIALS[(dev.index, n_tiles)] = torch.empty(max(n_tiles, 1), dtype=torch.float64, device=dev)
            loss = torch.empty((), dtype=torch.float32, device=dev)
            call("gsr_rasterize_fwd_l1", C, ptr(records), ptr(backgrounds), width, height, tile_w, tile_h,
                 ptr(tile_offsets), ptr(tile_order), ptr(pair_ids), ptr(l1_target), ptr(render_colors),
                 ptr(render_alphas), ptr(last_ids), ptr(rows), C * N, ptr(part), ptr(loss), _stream())
            ctx.l1_grad = render_colors          # (holds d loss / d render, not the render)
            render_colors = loss
        else:
            call("gsr_rasterize_fwd_planar" if planar else "gsr_rasterize_fwd", C, CH, ptr(records), ptr(backgrounds),
                 width, height, tile_w, tile_h, ptr(tile_offsets), ptr(tile_order), ptr(pair_ids), ptr(render_colors),
                 ptr(render_alphas), ptr(last_ids), ptr(rows), C * N, _stream())
            if planar:
                render_colors = render_colors.permute(0, 2, 3, 1)      # [C,H,W,CH] view of the planes
        ctx.rows = rows
        ctx.cfg = cfg
        ctx.shape = (C, N, color_stride, per_cam)
        ctx.save_for_backward(means2d, backgrounds, tile_offsets, tile_order, pair_ids,
                              render_alphas, last_ids, records)
        ctx.mark_non_differentiable(last_ids)
        # unused outputs (the alphas under a colour-only loss) arrive as None in backward
        # instead of freshly filled zero tensors: two fill launches less per step
        ctx.set_materialize_grads(False)
        return render_colors, render_alphas, last_ids

    @staticmethod
    def backward(ctx, v_render_colors, v_render_alphas, _v_last):
        width, height, tile_w, tile_h, CH, absgrad = ctx.cfg
        (means2d, backgrounds, tile_offsets, tile_order, pair_ids, render_alphas, last_ids,
         records) = ctx.saved_tensors
        C, N, color_stride, per_cam = ctx.shape
        dev = means2d.device
        if ctx.l1_grad is not None:          # first output was the L1 loss: its upstream gradient scales the stored image
            from .losses import _UNIT
            unit = _UNIT.get(dev)
            up = v_render_colors
            v_render_colors = ctx.l1_grad
            if up is None:
                v_render_colors = None
            elif not (unit is not None and up.data_ptr() == unit.data_ptr()):
                v_render_colors = v_render_colors * up.to(v_render_colors.dtype)
        if v_render_colors is None:
            v_render_colors = torch.zeros(C, height, width, CH, dtype=torch.float32, device=dev)
        # a gradient that arrives in planes (the fused SSIM backward writes it in the layout of the render it was
        # given) is read as it is; anything else as [C,H,W,CH]
        v_planar = (v_render_colors.dtype == torch.float32 and not v_render_colors.is_contiguous()
                    and v_render_colors.permute(0, 3, 1, 2).is_contiguous())
        if not v_planar:
            v_render_colors = _f32c(v_render_colors)
        if v_render_alphas is not None:          # None: the kernel takes a null pointer as zeros
            v_render_alphas = _f32c(v_render_alphas)
        rows, ctx.rows = ctx.rows, None      # cleared by the forward; a second backward gets fresh zeros
        if rows is None:
            rows = torch.zeros(C * N, GRAD_ROW, dtype=torch.float32, device=dev)
        call("gsr_rasterize_bwd_planar" if v_planar else "gsr_rasterize_bwd", C, CH, ptr(records), ptr(backgrounds), width, height, tile_w,
             tile_h, ptr(tile_offsets), ptr(tile_order), ptr(pair_ids), ptr(render_alphas),
             ptr(last_ids), ptr(v_render_colors), ptr(v_render_alphas), int(absgrad), ptr(rows),
             _stream())
        v_means2d = rows[:, GR_MEAN2D:GR_MEAN2D + 2].view(C, N, 2)
        v_conics = rows[:, GR_CONIC:GR_CONIC + 3].view(C, N, 3)
        v_colors = rows[:, GR_COLOR:GR_COLOR + color_stride].view(C, N, color_stride)
        v_opac = rows[:, GR_OPAC].view(C, N)
        if not per_cam:
            v_opac = v_opac.sum(0) if C > 1 else v_opac.reshape(N)
        if absgrad:
            means2d.absgrad = rows[:, GR_ABS:GR_ABS + 2].view(C, N, 2)
        if getattr(means2d, "_gsr_keep_grad", False):      # strategy.step_pre_backward asked for it (instead of retain_grad)
            means2d.grad = v_means2d
        v_bg = None
        if backgrounds is not None and ctx.needs_input_grad[4]:
            T_final = 1.0 - render_alphas
            v_bg = (v_render_colors * T_final).sum(dim=(1, 2))
        return v_means2d, v_conics, v_colors, v_opac, v_bg, None, None, None, None, None, None, None


# --------------------------------------------------------------------------- #
# the boundary
# --------------------------------------------------------------------------- #
def rasterization(
    means: Tensor,                 # [N,3]
    quats: Tensor,                 # [N,4] wxyz, un-normalised is fine
    scales: Tensor,                # [N,3] (already exp-activated)
    opacities: Tensor,             # [N]   (already sigmoid-activated)
    colors,                        # [N,K,3] SH | [N,D] | [C,N,D] | (sh0, shN) tuple
    viewmats: Tensor,              # [C,4,4] world->camera
    Ks: Tensor,                    # [C,3,3]
    width: int,
    height: int,
    near_plane: float = 0.01,
    far_plane: float = 1e10,
    radius_clip: float = 0.0,
    eps2d: float = 0.3,
    sh_degree: Optional[int] = None,
    packed: bool = True,
    tile_size: int = TILE,
    backgrounds: Optional[Tensor] = None,
    render_mode: str = "RGB",
    sparse_grad: bool = False,
    absgrad: bool = False,
    rasterize_mode: str = "classic",
    channel_chunk: int = 32,
    distributed: bool = False,
    camera_model: str = "pinhole",
    covars: Optional[Tensor] = None,
    _raw_activations: bool = False,
    _campos: Optional[Tensor] = None,
    _tight_tiles: bool = False,
    _isect_ids: bool = True,
    _l1_target: Optional[Tensor] = None,
    _planar_render: bool = False,
    **unsupported,
) -> Tuple[Tensor, Tensor, Dict]:
    """See module docstring. `packed` only changes gsplat's intermediate
    memory layout, never the rendered result: both values run the same dense
    [C,N] kernels here (and `meta` keeps the dense layout); `sparse_grad=True` (with
    packed=True) adds `meta["camera_ids"]` / `meta["gaussian_ids"]` of the rendered pairs and leaves
    the gradients dense (zero rows elsewhere) for `optim.FusedSparseAdam`.

    Private extensions used by `runner.rasterize_splats` (not part of gsplat's
    signature): `_raw_activations=True` means `scales` are log-scales and
    `opacities` logits, i.e. the exp/sigmoid of runner.py:324-325 (and their
    backward) run inside the projection kernels; `_campos` [C,3] supplies the
    camera centres so that the view matrices need not be inverted again;
    `_isect_ids=False` leaves `meta["isect_ids"]` (the sorted keys, meta data only) unwritten and None.
    `_tight_tiles=True` lists a (tile, Gaussian) pair only when some pixel of the tile can
    reach alpha >= 1/255 (exact ellipse test) instead of whenever gsplat's bounding rectangle
    touches the tile: identical image and gradients, `meta["flatten_ids"]` / `isect_offsets`
    are then a subset of gsplat's lists.

    Size limits (checked here): C*N < 2^27 -- a pair word holds the flat (camera, Gaussian) index in 27 bits
    beside the quadrant mask and the clamp flag; up to C*N < 2^25 (and <= 8192 buckets of 8 tiles) the bucketed
    tile-list builder of isect_bucket.hip runs, above that the global-sort builder of isect.hip, which honours
    `_tight_tiles` only through the masks (mask-0 pairs stay listed and are skipped by the compositing
    kernels)."""
    if unsupported:
        raise TypeError(f"rasterization(): unsupported arguments {sorted(unsupported)}")
    if camera_model != "pinhole":
        raise NotImplementedError(f"camera_model={camera_model!r}: only 'pinhole' is built")
    if covars is not None:
        raise NotImplementedError("covars=: pass quats + scales")
    if distributed:
        raise NotImplementedError(
            "distributed=True (gsplat's Gaussian-sharded all-to-all) is not built; use the "
            "view-parallel replicas (`distributed.GradSync` / `distributed.GatherRowsSync`) and "
            "call with distributed=False"
        )
    if sparse_grad and not packed:
        raise ValueError("sparse_grad=True only works with packed=True (as in gsplat)")
    if tile_size != TILE:
        raise NotImplementedError(f"tile_size={tile_size}: kernels are built for {TILE}")
    if render_mode not in ("RGB", "D", "ED", "RGB+D", "RGB+ED"):
        raise ValueError(f"render_mode={render_mode!r}")
    if rasterize_mode not in ("classic", "antialiased"):
        raise ValueError(f"rasterize_mode={rasterize_mode!r}")
    split_sh = isinstance(colors, (tuple, list))
    if split_sh and sh_degree is None:
        raise ValueError("colors=(sh0, shN) are spherical-harmonics coefficients: pass sh_degree")
    _check_cuda(means, quats, scales, opacities, viewmats, Ks, backgrounds,
                *(colors if split_sh else (colors,)))
    _lib.load()

    N, C = means.shape[0], viewmats.shape[0]
    if C * N >= (1 << 27):
        raise ValueError(f"rasterization(): C*N = {C * N} (camera, Gaussian) pairs; the pair words of the tile lists "
                         "hold 27 bits (< 134 217 728). Render the cameras in smaller batches.")
    assert means.shape == (N, 3) and quats.shape == (N, 4) and scales.shape == (N, 3), "shapes"
    assert opacities.shape == (N,), f"opacities {tuple(opacities.shape)}"
    assert viewmats.shape == (C, 4, 4) and Ks.shape == (C, 3, 3), "camera shapes"
    means, quats, scales, opacities = _f32c(means), _f32c(quats), _f32c(scales), _f32c(opacities)
    viewmats, Ks = _f32c(viewmats), _f32c(Ks)
    antialiased = rasterize_mode == "antialiased"
    with_depth = render_mode in ("RGB+D", "RGB+ED")
    depth_only = render_mode in ("D", "ED")

    use_sh = sh_degree is not None and not depth_only
    sh_a = sh_b = None
    if use_sh:
        if split_sh:
            sh_a, sh_b = _f32c(colors[0]), _f32c(colors[1])
            assert sh_a.shape == (N, 1, 3) and sh_b.shape[0] == N and sh_b.shape[2] == 3
        else:
            assert colors.dim() == 3 and colors.shape[0] == N and colors.shape[2] == 3, (
                "with sh_degree set, colors must be [N,K,3]")
            sh_a = _f32c(colors)
        color_stride = 4 if with_depth else 3
        depth_channel = 3 if with_depth else -1
        campos = _f32c(_campos) if _campos is not None else inverse4x4(viewmats)[1]
    else:
        color_stride, depth_channel, campos = 0, -1, None

    tile_w = math.ceil(width / TILE)
    tile_h = math.ceil(height / TILE)
    activations = 0
    if _raw_activations:
        # the sigmoid chain rule is fused only when the compositing gradient of the
        # opacity reaches the projection backward unchanged (classic mode)
        activations = ACT_EXP_SCALES | (0 if antialiased else ACT_SIGMOID_OPAC)
        if antialiased:
            opacities = torch.sigmoid(opacities)
    # the bucketed tile-list builder does its own (LDS) counting; otherwise the per-tile
    # count pass is fused into the projection kernel
    fuse_count = not bucket_layout_ok(C, N, tile_w, tile_h)
    cfg = (int(width), int(height), float(eps2d), float(near_plane), float(far_plane),
           float(radius_clip), bool(antialiased), int(sh_degree) if use_sh else -1, color_stride,
           depth_channel, activations, tile_w if fuse_count else 0, tile_h if fuse_count else 0)
    radii, means2d, depths, conics, comps, sh_colors, opac_act, tile_counts, records = _ProjectSH.apply(
        means, quats, scales, opacities, sh_a, sh_b, viewmats, Ks, campos, cfg)

    if activations & ACT_SIGMOID_OPAC:
        opac = opac_act                                # [N], sigmoid done in the kernel
    elif antialiased:
        opac = opacities[None, :] * comps              # [C,N]
    else:
        opac = opacities

    if use_sh:
        feats = sh_colors                                  # [C,N,3|4], depth already in ch 3
        CH = color_stride
    else:
        if depth_only:
            feats = depths[..., None]
        else:
            col = colors[0] if split_sh else colors
            col = _f32c(col)
            if col.dim() == 2:
                col = col[None].expand(C, -1, -1)
            assert col.shape[:2] == (C, N), f"colors {tuple(col.shape)}"
            feats = torch.cat([col, depths[..., None]], dim=-1) if with_depth else col
        feats = feats.contiguous()
        CH = feats.shape[-1]
        if CH > 5:
            raise NotImplementedError(f"{CH} colour channels: kernels are built for <= 5")
    if backgrounds is not None:
        backgrounds = _f32c(backgrounds)
        if with_depth:
            backgrounds = torch.cat([backgrounds, torch.zeros_like(backgrounds[:, :1])], dim=-1)
        elif depth_only:
            backgrounds = torch.zeros(C, 1, dtype=torch.float32, device=means.device)
        backgrounds = backgrounds.contiguous()

    rcfg = (int(width), int(height), tile_w, tile_h, CH, bool(absgrad))
    l1_target = None
    if _l1_target is not None:
        # (private, runner.train_step: the step's plain L1 loss inside the compositing forward; `render_colors` is then
        # returned as None and meta["l1_loss"] holds mean |render - target|)
        if CH != 3 or render_mode != "RGB":
            raise ValueError("_l1_target: three colour channels, render_mode RGB")
        l1_target = _f32c(_l1_target)
        if tuple(l1_target.shape) != (C, int(height), int(width), 3):
            raise ValueError(f"_l1_target: expected {(C, int(height), int(width), 3)}, got {tuple(l1_target.shape)}")
    # Tile lists + compositing. When the bucketed builder applies and a previous frame
    # told us how many intersections to expect, nothing here waits for the GPU until the
    # compositing forward has been queued (see _IsectState).
    capacity = _IsectState.capacity.get(means.device.index) if not fuse_count else None
    while True:
        tile_offsets, tile_order, flatten_ids, isect_keys, tpg, pair_ids = isect_tiles_sorted(
            means2d.detach(), radii, depths.detach(), tile_w, tile_h, want_tiles_per_gauss=False,
            tile_counts=tile_counts, capacity=capacity, conics=conics.detach(),
            opacities=opac.detach().contiguous(), tight=bool(_tight_tiles), want_keys=bool(_isect_ids))
        render_colors, render_alphas, _last = _Rasterize.apply(
            means2d, conics, feats, opac, backgrounds, tile_offsets, tile_order, pair_ids,
            records if use_sh else None, rcfg, l1_target, bool(_planar_render) and render_mode == "RGB")
        if not isinstance(tpg, _PendingIsect):
            break
        n_isects, overflowed = tpg.resolve()
        if not overflowed:
            flatten_ids, tpg = flatten_ids[:n_isects], None
            isect_keys = isect_keys[:n_isects] if isect_keys is not None else None
            pair_ids = pair_ids[:n_isects]
            break
        capacity = None        # rare: this frame outgrew the guess -> rebuild, blocking

    if means2d.requires_grad:
        # strategy.step_pre_backward may ask the compositing backward to leave `.grad` on this tensor itself (a view into
        # its gradient rows) instead of registering retain_grad's cloning hook
        means2d._gsr_grad_in_backward = True
    if render_mode in ("ED", "RGB+ED"):
        render_colors = torch.cat(
            [render_colors[..., :-1], render_colors[..., -1:] / render_alphas.clamp(min=1e-10)],
            dim=-1)

    camera_ids = gaussian_ids = None
    if sparse_grad:
        # what gsplat's packed mode reports and runner.py:661-672 builds its sparse gradients over: the
        # (camera, Gaussian) pairs that are rendered. (The gradients themselves stay dense here, exact
        # zeros on the other rows: `optim.FusedSparseAdam` / torch.sparse_coo_tensor select the rows.)
        vis = (radii > 0).all(-1).flatten().nonzero(as_tuple=True)[0]
        camera_ids, gaussian_ids = vis // N, vis % N
    meta = {
        "camera_ids": camera_ids, "gaussian_ids": gaussian_ids,
        "radii": radii, "means2d": means2d, "depths": depths, "conics": conics,
        "opacities": opac, "tile_width": tile_w, "tile_height": tile_h,
        "tiles_per_gauss": tpg, "isect_ids": isect_keys, "flatten_ids": flatten_ids,
        "pair_ids": pair_ids,
        "isect_offsets": tile_offsets[:-1].view(C, tile_h, tile_w),
        "width": width, "height": height, "tile_size": TILE, "n_cameras": C,
    }
    if l1_target is not None:
        meta["l1_loss"] = render_colors          # (the first output of _Rasterize in this mode)
        render_colors = None
    return render_colors, render_alphas, meta
