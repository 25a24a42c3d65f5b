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

    # dict-like access so existing code (`optimizers["means"]`, `.values()`) keeps working
    def __getitem__(self, k):
        return self.optimizers[k]

    def __contains__(self, k):
        return k in self.optimizers

    def keys(self):
        return self.optimizers.keys()

    def items(self):
        return self.optimizers.items()

    @torch.no_grad()
    def step(self) -> None:
        ps, gs, ms, vs, numel, ss, bc2 = [], [], [], [], [], [], []
        beta1 = beta2 = eps = None
        for opt in self.optimizers.values():
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
            ps.append(p.data_ptr()); gs.append(g.data_ptr())
            ms.append(st["exp_avg"].data_ptr()); vs.append(st["exp_avg_sq"].data_ptr())
            numel.append(p.numel())
            ss.append(float(grp["lr"]) / (1.0 - beta1 ** t))
            bc2.append((1.0 - beta2 ** t) ** 0.5)
        n = len(ps)
        if n == 0:
            return
        PA = C.c_void_p * n
        call("gsr_adam_step", n, PA(*ps), PA(*gs), PA(*ms), PA(*vs), (C.c_int64 * n)(*numel),
             (C.c_float * n)(*ss), (C.c_float * n)(*bc2), beta1, beta2, eps,
             torch.cuda.current_stream().cuda_stream)

    def zero_grad(self, set_to_none: bool = True) -> None:
        for opt in self.optimizers.values():
            opt.zero_grad(set_to_none=set_to_none)

    def values(self):
        """Iterating `.values()` and calling step()/zero_grad() on each (the
        reference's loop) still works: the first step() performs the fused
        launch for all parameters, the others are no-ops for that iteration."""
        return [_Member(self, i) for i in range(len(self.optimizers))]


class _Member:
    def __init__(self, parent: FusedAdam, index: int):
        self.parent, self.index = parent, index

    def step(self):
        if self.index == 0:
            self.parent.step()

    def zero_grad(self, set_to_none: bool = True):
        if self.index == 0:
            self.parent.zero_grad(set_to_none=set_to_none)
