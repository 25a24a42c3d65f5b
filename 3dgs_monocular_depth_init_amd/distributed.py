"""View-parallel multi-GPU mode: replicated Gaussians, one view per rank per
step, all-reduce (sum) of the Gaussian gradients, identical Adam on every rank.

This is the scheme BASELINE.json's north_star mandates. It differs from the
reference, which shards Gaussians over ranks (runner.py:94-96) and relies on
gsplat's `distributed=True` all-gather + all-to-all (runner.py:359); the
learning-rate / beta / eps scaling rule for the effective batch
BS = batch_size * world_size is the reference's own (runner.py:128-137).

One process per GPU (torch.distributed; backend "nccl" = RCCL over xGMI on
ROCm, "gloo" for the CPU tests). The only collective on the data path is the
gradient all-reduce; nothing else is exchanged.
"""
from __future__ import annotations

from typing import Dict, Iterable, List

import torch
import torch.distributed as dist

PARAM_ORDER = ("shN", "sh0", "means", "quats", "scales", "opacities")


class GradSync:
    """All-reduce the gradients of the six parameter tensors.

    Gradients are summed (the loss of a W-view batch is the sum of the
    per-view losses, as with the reference's batch dimension) unless
    `average=True`. Buckets follow PARAM_ORDER: SH gradients (180 B of the
    236 B per Gaussian) are final first in the backward pass, means last, so
    the large message overlaps the tail of backward when hooks are used.
    """

    def __init__(self, splats, world_size: int, average: bool = False, group=None):
        self.splats = splats
        self.world = world_size
        self.average = average
        self.group = group

    def __call__(self) -> None:
        if self.world <= 1:
            return
        works = []
        for name in PARAM_ORDER:
            if name not in self.splats:
                continue
            g = self.splats[name].grad
            if g is None:
                g = torch.zeros_like(self.splats[name])
                self.splats[name].grad = g
            if not g.is_contiguous():
                g = g.contiguous()
                self.splats[name].grad = g
            works.append((g, dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group, async_op=True)))
        for g, w in works:
            w.wait()
            if self.average:
                g.div_(self.world)


def shard_views(n_views: int, step: int, rank: int, world: int, perm=None) -> int:
    """Camera index rank `rank` renders at `step`: perm[(step*world + rank) % n]."""
    idx = (step * world + rank) % n_views
    return int(perm[idx]) if perm is not None else idx


def fuse_optimizers(splats, optimizers: Dict[str, torch.optim.Optimizer]):
    """Wrap the reference's six per-parameter Adam instances in one fused launch."""
    from .optim import FusedAdam
    return FusedAdam(optimizers)
