"""Fused multi-tensor Adam behind the reference's optimizer layout.

The reference keeps one torch.optim.Adam per parameter tensor
(/root/reference/gs_init_compare/runner.py:129-137) and steps them in a loop
(runner.py:676-679); gsplat's densification strategy edits their `state`
in place. FusedAdam keeps exactly those objects as the source of truth
(param_groups[0]["lr"] for schedulers, state[p]["exp_avg"/"exp_avg_sq"/"step"]
for the strategy) but performs all updates in ONE HIP launch
(gsr_adam_step) instead of ~8 foreach kernels per parameter.
"""
from __future__ import annotations

import ctypes as C
from typing import Dict

import torch

from ._lib import call
from ._lib import current_stream as _raw_stream


class FusedAdam:
    def __init__(self, optimizers: Dict[str, torch.optim.Optimizer]):
        for name, opt in optimizers.items():
            if not isinstance(opt, torch.optim.Adam):
                raise TypeError(f"optimizer {name!r} is {type(opt).__name__}, expected torch.optim.Adam")
            g = opt.param_groups[0]
            if len(opt.param_groups) != 1 or len(g["params"]) != 1:
                raise ValueError("FusedAdam expects one parameter per optimizer (runner.py:129-137)")
            if g.get("weight_decay", 0) != 0 or g.get("amsgrad", False) or g.get("maximize", False):
                raise NotImplementedError("weight_decay / amsgrad / maximize are not built")
        if len(optimizers) > 8:
            raise ValueError("at most 8 tensors per fused launch")
        self.optimizers = optimizers
        self.grad_sync = None           # distributed.GradSync.attach() sets this
        self._claimed = False           # the projection backward applied this step's update

    # dict-like access so existing code (`optimizers["means"]`, `.values()`) keeps working
    def __getitem__(self, k):
        return self.optimizers[k]

    def __contains__(self, k):
        return k in self.optimizers

    def keys(self):
        return self.optimizers.keys()

    def items(self):
        return self.optimizers.items()

    def _prepare(self):
        """Per-tensor launch parameters; creates missing state and advances `step`.
        Returns ([(name, p, g, exp_avg, exp_avg_sq, step_size, bc2_sqrt)], beta1, beta2, eps)."""
        items = []
        beta1 = beta2 = eps = None
        for name, opt in self.optimizers.items():
            grp = opt.param_groups[0]
            p = grp["params"][0]
            if p.grad is None:
                continue
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous()):
                raise ValueError("FusedAdam: parameters must be contiguous fp32 ROCm tensors")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            t = float(st["step"])
            b1, b2 = grp["betas"]
            if beta1 is None:
                beta1, beta2, eps = float(b1), float(b2), float(grp["eps"])
            elif (float(b1), float(b2), float(grp["eps"])) != (beta1, beta2, eps):
                raise ValueError("FusedAdam: all parameters must share betas and eps")
            items.append((name, p, g, st["exp_avg"], st["exp_avg_sq"],
                          float(grp["lr"]) / (1.0 - beta1 ** t), (1.0 - beta2 ** t) ** 0.5))
        return items, beta1, beta2, eps

    @staticmethod
    def _launch(pieces, beta1, beta2, eps) -> None:
        """pieces: [(p_ptr, g_ptr, m_ptr, v_ptr, numel, step_size, bc2_sqrt)], at most 8."""
        n = len(pieces)
        if n == 0:
            return
        PA = C.c_void_p * n
        call("gsr_adam_step", n, PA(*[x[0] for x in pieces]), PA(*[x[1] for x in pieces]),
             PA(*[x[2] for x in pieces]), PA(*[x[3] for x in pieces]),
             (C.c_int64 * n)(*[x[4] for x in pieces]), (C.c_float * n)(*[x[5] for x in pieces]),
             (C.c_float * n)(*[x[6] for x in pieces]), beta1, beta2, eps,
             _raw_stream())

    @staticmethod
    def pieces_for_range(a: int, b: int, segments):
        """Intersect the arena range [a, b) with parameter segments
        [(key, arena_offset, numel)] -> [(key, start_in_param, count)]."""
        out = []
        for key, off, n in segments:
            lo, hi = max(a, off), min(b, off + n)
            if hi > lo:
                out.append((key, lo - off, hi - lo))
        return out

    @torch.no_grad()
    def step(self) -> None:
        if self._claimed:
            # optimizer in backward: the update (and the step count) already happened inside
            # loss.backward(). A gradient found here reached the parameters OUTSIDE the
            # rasterizer (a regulariser, a second loss term): stepping on it would count the
            # step twice and decay the moments twice -- refuse instead of training wrongly.
            self._claimed = False
            stray = [n for n, o in self.optimizers.items()
                     if o.param_groups[0]["params"][0].grad is not None]
            if stray:
                raise RuntimeError(
                    f"FusedAdam: optimizer-in-backward already stepped, but {stray} carry gradients "
                    "from outside the rasterizer; disable fuse_into_backward for such losses")
            return
        items, beta1, beta2, eps = self._prepare()
        sync = self.grad_sync
        pending = sync.take_pending() if sync is not None else []
        if pending:
            arena = sync.arena
            by_name = {it[0]: it for it in items}
            ok = arena is not None and all(
                name in by_name and arena.owns(name, by_name[name][2]) for name in arena.offsets
            ) and len(by_name) == len(arena.offsets)
            if not ok:                       # grads are not the arena's views: plain path
                for work, a, b in pending:
                    work.wait()
                    if sync.average:
                        arena.flat[a:b].div_(sync.world)
                pending = []
            else:
                segs = [(name, arena.offsets[name], by_name[name][1].numel()) for name in arena.offsets]
                for work, a, b in pending:   # Adam on chunk k overlaps the all-reduce of k+1
                    work.wait()
                    if sync.average:
                        arena.flat[a:b].div_(sync.world)
                    pieces = []
                    for name, start, cnt in self.pieces_for_range(a, b, segs):
                        _, p, g, m, v, ss, bc2 = by_name[name]
                        o = 4 * start
                        pieces.append((p.data_ptr() + o, g.data_ptr() + o, m.data_ptr() + o,
                                       v.data_ptr() + o, cnt, ss, bc2))
                    self._launch(pieces, beta1, beta2, eps)
                return
        self._launch([(p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), ss, bc2)
                      for _, p, g, m, v, ss, bc2 in items], beta1, beta2, eps)

    # ---- optimizer in backward ------------------------------------------------------
    FUSED_ORDER = ("means", "quats", "scales", "opacities", "sh0", "shN")
    # What else ONE step does to every Gaussian, applied inside the fused backward (gsr_project_bwd_adam_ex,
    # gsr_step_extras): the "mcmc" preset's position noise and regulariser gradients, DefaultStrategy's statistics.
    # Set by runner.train_step before loss.backward(), consumed (and cleared) by the projection backward.
    step_extras = None      # dict: noise, noise_scale, opacity_reg, scale_reg, stats

    def set_step_extras(self, noise=None, noise_scale: float = 0.0, opacity_reg: float = 0.0, scale_reg: float = 0.0,
                        stats=None) -> None:
        """stats: (grad2d [N], count [N], radii_state [N] or None, sx, sy, inv_max_wh, use_absgrad)."""
        self.step_extras = dict(noise=noise, noise_scale=float(noise_scale), opacity_reg=float(opacity_reg),
                                scale_reg=float(scale_reg), stats=stats)

    def take_step_extras(self):
        ex, self.step_extras = self.step_extras, None
        return ex

    def fuse_into_backward(self, enable: bool = True) -> None:
        """Single-process training with a purely photometric loss: let the projection
        backward apply this optimizer's update itself (`gsr_project_bwd_adam`) -- the
        parameter gradients are then never written to memory nor read back, and
        `step()` finds nothing left to do. Not valid together with a gradient all-reduce,
        gradient clipping, more than one backward per step, or loss terms that reach the
        parameters outside the rasterizer: `step()` raises when it finds such a gradient,
        and `runner.train_step` raises for the reference's opacity / scale regularisers.
        ORDER: the update lands during `loss.backward()`, i.e. before
        `strategy.step_post_backward`, whereas the reference runs the strategy first
        (runner.py:638-679). `runner.train_step` therefore suspends the fusion on every step
        on which the strategy edits parameters or optimizer state (`strategy.mutates_params`),
        so those steps run in the reference's order."""
        from .rendering import set_backward_optimizer
        set_backward_optimizer(self if enable else None)

    @torch.no_grad()
    def claim(self, tensors):
        """Called by the projection backward with its six parameter inputs. Returns the
        launch arguments (and advances `step`) when they are exactly this optimizer's
        parameters, else None."""
        if any(n not in self.optimizers for n in self.FUSED_ORDER):
            return None
        ps, ms, vs, ss, bc2 = [], [], [], [], []
        beta1 = beta2 = eps = None
        states = []
        for name, t in zip(self.FUSED_ORDER, tensors):
            opt = self.optimizers[name]
            grp = opt.param_groups[0]
            p = grp["params"][0]
            if t is None or p.data_ptr() != t.data_ptr() or p.shape != t.shape or not p.is_contiguous() \
                    or p.dtype != torch.float32 or not p.is_cuda:
                return None
            b1, b2 = grp["betas"]
            if beta1 is None:
                beta1, beta2, eps = float(b1), float(b2), float(grp["eps"])
            elif (float(b1), float(b2), float(grp["eps"])) != (beta1, beta2, eps):
                return None
            states.append((opt, grp, p))
        for opt, grp, p in states:
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = torch.tensor(0.0, dtype=torch.float32)
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            t = float(st["step"])
            ps.append(p.data_ptr()); ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
            ss.append(float(grp["lr"]) / (1.0 - beta1 ** t))
            bc2.append((1.0 - beta2 ** t) ** 0.5)
        self._claimed = True
        PA = C.c_void_p * 6
        FA = C.c_float * 6
        return PA(*ps), PA(*ms), PA(*vs), FA(*ss), FA(*bc2), beta1, beta2, eps

    def zero_grad(self, set_to_none: bool = True) -> None:
        # (directly: torch.optim.Optimizer.zero_grad costs ~20 us of host time per optimizer --
        # profiler scopes, dynamo wrappers -- i.e. 0.1 ms per step for the six of them, a fifth of
        # the step's host time; with the optimizer in the backward the gradients are None anyway)
        for opt in self.optimizers.values():
            for grp in opt.param_groups:
                for p in grp["params"]:
                    if p.grad is None:
                        continue
                    if set_to_none:
                        p.grad = None
                    else:
                        p.grad.detach_()
                        p.grad.requires_grad_(False)
                        p.grad.zero_()

    def values(self):
        """Iterating `.values()` and calling step()/zero_grad() on each (the
        reference's loop) still works: the first step() performs the fused
        launch for all parameters, the others are no-ops for that iteration."""
        return [_Member(self, i) for i in range(len(self.optimizers))]


class FusedSparseAdam:
    """The reference's `sparse_grad` optimizers -- one torch.optim.SparseAdam per parameter tensor
    (runner.py:130), stepped on sparse gradients over `info["gaussian_ids"]` (runner.py:661-679) -- behind
    the same objects (param_groups for the schedulers, state[p]["step" / "exp_avg" / "exp_avg_sq"]), with
    all six updates in ONE launch (`gsr_sparse_adam_step`): rows of Gaussians the step rendered get
    SparseAdam's update, every other row keeps its parameter and its moments. `step(info)` takes the
    rasterizer's `info` (its `radii` decide the rows)."""

    def __init__(self, optimizers: Dict[str, torch.optim.Optimizer]):
        for name, opt in optimizers.items():
            if not isinstance(opt, torch.optim.SparseAdam):
                raise TypeError(f"optimizer {name!r} is {type(opt).__name__}, expected torch.optim.SparseAdam")
            if len(opt.param_groups) != 1 or len(opt.param_groups[0]["params"]) != 1:
                raise ValueError("FusedSparseAdam expects one parameter per optimizer (runner.py:129-137)")
            if opt.param_groups[0].get("maximize", False):
                raise NotImplementedError("maximize is not built")
        if len(optimizers) > 8:
            raise ValueError("at most 8 tensors per fused launch")
        self.optimizers = optimizers

    def __getitem__(self, k):
        return self.optimizers[k]

    def __contains__(self, k):
        return k in self.optimizers

    def keys(self):
        return self.optimizers.keys()

    def items(self):
        return self.optimizers.items()

    def values(self):
        return self.optimizers.values()

    @torch.no_grad()
    def step(self, info) -> None:
        radii = info["radii"]
        if radii.shape[0] > 255:
            raise ValueError("FusedSparseAdam: at most 255 cameras per step")
        # [N]: by how many cameras of the batch a row is rendered. The reference's sparse gradient has one entry
        # per rendered (camera, Gaussian) pair, each carrying the row's dense gradient, and SparseAdam coalesces
        # (sums) them (runner.py:661-672): a row seen by k cameras steps on k times its gradient.
        visible = (radii > 0).all(-1).sum(0).to(torch.uint8).contiguous()
        rows = visible.numel()
        P, G, M, V, L, S = [], [], [], [], [], []
        beta1 = beta2 = eps = None
        for name, opt in self.optimizers.items():
            grp = opt.param_groups[0]
            p = grp["params"][0]
            if p.grad is None:
                continue
            if p.grad.is_sparse:
                raise ValueError("FusedSparseAdam takes the dense gradients of the rasterizer (zero rows elsewhere)")
            if not (p.is_cuda and p.dtype == torch.float32 and p.is_contiguous() and p.shape[0] == rows):
                raise ValueError("FusedSparseAdam: parameters must be contiguous fp32 ROCm tensors of N rows")
            st = opt.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.preserve_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.preserve_format)
            st["step"] += 1
            t = float(st["step"])
            b1, b2 = grp["betas"]
            if beta1 is None:
                beta1, beta2, eps = float(b1), float(b2), float(grp["eps"])
            elif (float(b1), float(b2), float(grp["eps"])) != (beta1, beta2, eps):
                raise ValueError("FusedSparseAdam: all parameters must share betas and eps")
            g = p.grad if p.grad.is_contiguous() else p.grad.contiguous()
            P.append(p.data_ptr()); G.append(g.data_ptr())
            M.append(st["exp_avg"].data_ptr()); V.append(st["exp_avg_sq"].data_ptr())
            L.append(p.numel() // max(rows, 1))
            S.append(float(grp["lr"]) * (1.0 - beta2 ** t) ** 0.5 / (1.0 - beta1 ** t))
        n = len(P)
        if n == 0 or rows == 0:
            return
        PA = C.c_void_p * n
        call("gsr_sparse_adam_step", n, rows, visible.data_ptr(), PA(*P), PA(*G), PA(*M), PA(*V),
             (C.c_int32 * n)(*L), (C.c_float * n)(*S), beta1, beta2, eps, _raw_stream())

    def zero_grad(self, set_to_none: bool = True) -> None:
        for opt in self.optimizers.values():
            opt.zero_grad(set_to_none=set_to_none)


class _Member:
    def __init__(self, parent: FusedAdam, index: int):
        self.parent, self.index = parent, index

    def step(self):
        if self.index == 0:
            self.parent.step()

    def zero_grad(self, set_to_none: bool = True):
        if self.index == 0:
            self.parent.zero_grad(set_to_none=set_to_none)
