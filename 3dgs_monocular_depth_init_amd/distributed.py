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
from ._lib import current_stream as _raw_stream

PARAM_ORDER = ("shN", "sh0", "means", "quats", "scales", "opacities")


class GradSync:
    """All-reduce (sum) the gradients of the six parameter tensors.

    Fast path: a `rendering.GradArena` is registered, so after backward every
    `p.grad` is a view of one flat buffer and ONE collective moves all 59*N
    floats (a single large message is what RCCL's xGMI paths are best at).
    Fallback (grads that are not arena views, e.g. CPU/gloo tests, extra
    autograd consumers): one async all-reduce per tensor, SH first.
    Gradients are summed (the loss of a W-view batch is the sum of per-view
    losses, as with the reference's batch dimension) unless `average=True`.

    Pipelining with the optimizer: after `attach(fused_adam)` the arena is
    reduced as `chunks` asynchronous all-reduces (RCCL runs them in order on
    its own stream) and `FusedAdam.step()` waits chunk by chunk, so the Adam
    update of chunk k overlaps the all-reduce of chunk k+1 (Adam is
    element-wise, so any flat range of the arena can be stepped on its own).
    `finish()` waits for whatever nobody consumed."""

    def __init__(self, splats, world_size: int, average: bool = False, group=None,
                 use_arena: bool = True, force: bool = False, chunks: int = 4):
        self.splats = splats
        self.world = world_size
        self.average = average
        self.group = group
        self.arena = None
        self.use_arena = use_arena
        self.force = force              # issue the collective even at world_size 1 (rehearsal)
        self.chunks = max(1, int(chunks))
        self._consumer = None           # FusedAdam that consumes pending chunks
        self._pending = []              # [(work, start, end)] in arena elements, issue order
        self._maybe_build_arena()

    def attach(self, fused_adam) -> None:
        """Let `fused_adam.step()` consume the chunked all-reduce as it completes."""
        self._consumer = fused_adam
        fused_adam.grad_sync = self

    def chunk_bounds(self):
        """Equal flat ranges of the arena, boundaries on 16-byte multiples."""
        n = self.arena.flat.numel()
        k = self.chunks
        step = (n // k + 3) // 4 * 4
        out, a = [], 0
        while a < n:
            b = min(n, a + step) if len(out) < k - 1 else n
            out.append((a, b))
            a = b
        return out

    def take_pending(self):
        p, self._pending = self._pending, []
        return p

    def finish(self) -> None:
        for work, a, b in self.take_pending():
            work.wait()
            if self.average:
                self.arena.flat[a:b].div_(self.world)

    def _maybe_build_arena(self) -> None:
        if not self.use_arena:
            return
        first = next(iter(self.splats.values()))
        if not first.is_cuda:
            return
        shapes = {k: tuple(p.shape) for k, p in self.splats.items()}
        if self.arena is None or self.arena.shapes != shapes:     # also after densification
            from .rendering import GradArena, set_grad_arena
            self.arena = GradArena(shapes, first.device)
            set_grad_arena(self.arena)

    def __call__(self) -> None:
        if self.arena is not None:
            self.arena.reset()
        if self.world <= 1 and not self.force:
            self._maybe_build_arena()
            return
        names = [n for n in PARAM_ORDER if n in self.splats]
        if self.arena is not None and all(self.arena.owns(n, self.splats[n].grad) for n in names) \
                and len(names) == len(self.arena.offsets):
            if self._consumer is not None and self.chunks > 1:
                self.finish()
                flat = self.arena.flat
                for a, b in self.chunk_bounds():
                    work = dist.all_reduce(flat[a:b], op=dist.ReduceOp.SUM, group=self.group,
                                           async_op=True)
                    self._pending.append((work, a, b))
            else:
                dist.all_reduce(self.arena.flat, op=dist.ReduceOp.SUM, group=self.group)
                if self.average:
                    self.arena.flat.div_(self.world)
        else:
            works = []
            for name in names:
                g = self.splats[name].grad
                if g is None:
                    g = torch.zeros_like(self.splats[name])
                    self.splats[name].grad = g
                if not g.is_contiguous():
                    g = g.contiguous()
                    self.splats[name].grad = g
                works.append((g, dist.all_reduce(g, op=dist.ReduceOp.SUM, group=self.group,
                                                 async_op=True)))
            for g, w in works:
                w.wait()
                if self.average:
                    g.div_(self.world)
        self._maybe_build_arena()


class GatherRowsSync:
    """The same sum of per-view gradients with 6x fewer bytes on the wire.

    The parameter gradients are 59 floats per Gaussian, but what a view contributes is
    determined by the 9 floats per Gaussian its compositing backward leaves in the 64-byte
    rows (d loss / d means2d, conic, opacity, colour). So: every rank packs its rows to 36
    bytes (the 9 floats, zero for invisible pairs, `gsr_pack_grad_rows`), ONE all-gather moves them
    (36 MB per rank at 1 M Gaussians instead of a 236 MB all-reduce), and every rank runs the
    projection backward over the cameras of ALL ranks (`C = world_size`, the batch path of
    `gsr_project_bwd_adam`), which also keeps the Adam update fused in the backward. Every
    rank sums the views in rank order, so the replicas stay bit-identical. The exchange is
    pipelined over `chunks` Gaussian ranges: the all-gather of range k+1 (asynchronous, on RCCL's
    stream) overlaps the projection backward + Adam of range k.

    Usage: `sync = GatherRowsSync(fused_adam, world, rank)`; before each step
    `sync.set_views(camtoworlds_all [W,4,4], Ks_all [W,3,3])` (row r = the camera rank r
    renders in this step, e.g. via `shard_views`); then the usual `train_step` with the local
    camera and `grad_sync=sync`. `close()` unhooks it."""

    def __init__(self, fused_adam, world_size: int, rank: int, group=None, chunks: int = 4,
                 min_chunk: int = 4096, rows: str = "fp32"):
        """rows="fp16" (opt-in): the rows travel as 20 bytes -- a shared exponent and nine IEEE halves,
        `gsr_pack_grad_rows_h` -- instead of 36: 11 significant bits per view-space gradient value,
        inside the 1e-3 relative tolerance of BASELINE.json but no longer the fp32 sum of the
        all-reduce; replicas stay bit-identical (every rank decodes the same bytes)."""
        from .rendering import set_row_exchange
        if rows not in ("fp32", "fp16"):
            raise ValueError(f"rows={rows!r}")
        self.rows = rows
        self.row_stride = 5 if rows == "fp16" else 9          # GSR_PACKED_ROW_H dwords / GSR_PACKED_ROW floats
        self.world, self.rank, self.group = world_size, rank, group
        self.fused = fused_adam
        self.chunks = max(1, int(chunks))
        self.min_chunk = max(64, int(min_chunk))
        fused_adam.fuse_into_backward(True)
        set_row_exchange(self)
        self._views = None
        self._bufs = {}
        self._bufs_n = -1

    def set_views(self, camtoworlds_all, Ks_all) -> None:
        from .rendering import inverse4x4
        assert camtoworlds_all.shape[0] == self.world and Ks_all.shape[0] == self.world
        viewmats, campos = inverse4x4(camtoworlds_all, translation_of="input")
        self._views = (viewmats, Ks_all.float().contiguous(), campos)

    def chunk_bounds(self, N: int):
        """Gaussian ranges of the pipeline, boundaries on multiples of 64 (a wave of the projection
        backward, and 16-byte alignment of every per-Gaussian parameter row incl. the 180-byte shN)."""
        k = min(self.chunks, max(1, N // self.min_chunk))
        step = (N // k + 63) // 64 * 64
        out, a = [], 0
        while a < N:
            b = N if len(out) == k - 1 else min(N, a + step)
            out.append((a, b))
            a = b
        return out

    def exchange(self, rows, radii, N: int):
        """Called by the projection backward: local rows [N,16] + radii -> the cameras of all
        ranks and a list of chunks `(start, count, rows_all [W*count, 9], wait)`; the consumer
        calls `wait()` and then runs the projection backward of that Gaussian range over the W
        cameras. The all-gather of chunk k+1 is in flight (on RCCL's own stream) while chunk k is
        being consumed: both are element-wise in the Gaussian index, so nothing else orders them."""
        from ._lib import call, ptr
        if self._views is None:
            raise RuntimeError("GatherRowsSync.set_views() must be called before every step")
        dev = rows.device
        W = self.world
        st = _raw_stream()
        chunks = []
        if self._bufs_n != N:                   # the Gaussian count changed (densification): the old
            self._bufs.clear()                  # generation's buffers (W*N*36 bytes in all) are dropped
            self._bufs_n = N                    # before the chunk loop, never in the middle of a step
        for a, b in self.chunk_bounds(N):
            n = b - a
            key = (a, n)
            buf = self._bufs.get(key)
            if buf is None:
                buf = self._bufs[key] = (torch.empty(W * n, 5, dtype=torch.int32, device=dev) if self.rows == "fp16"
                                         else torch.empty(W * n, 9, dtype=torch.float32, device=dev))
            mine = buf[self.rank * n:(self.rank + 1) * n]
            call("gsr_pack_grad_rows_h" if self.rows == "fp16" else "gsr_pack_grad_rows", n,
                 rows.data_ptr() + 64 * a, radii.data_ptr() + 8 * a, ptr(mine), st)
            chunks.append((a, n, buf, self._all_gather(buf, mine, a, n)))
        vm, Ks, campos = self._views
        self._views = None
        return chunks, vm, Ks, campos, W

    def _all_gather(self, buf, mine, start: int, n: int):
        """Start the exchange of one Gaussian range: `mine` (= buf[rank*n:(rank+1)*n], this rank's
        packed rows) to every rank, segment r of `buf` <- rank r's rows. Returns the function that
        waits for it. (Its own method so that tests can stand in for the other ranks of a large
        world on one GPU: tests/test_gpu_two_ranks.py::test_world8_chunked_exchange...)"""
        W = self.world
        if W == 1:
            return lambda: None
        if dist.get_backend(self.group) == "nccl":
            return dist.all_gather_into_tensor(buf, mine, group=self.group, async_op=True).wait   # in place
        # gloo (tests): list form, input must not alias the outputs
        outs = [buf[r * n:(r + 1) * n] for r in range(W)]
        return dist.all_gather(outs, mine.clone(), group=self.group, async_op=True).wait

    def __call__(self) -> None:          # train_step's grad_sync hook: nothing left to do
        return None

    def finish(self) -> None:
        return None

    def close(self) -> None:
        from .rendering import set_row_exchange
        set_row_exchange(None)
        self.fused.fuse_into_backward(False)


def shard_views(n_views: int, step: int, rank: int, world: int, perm=None) -> int:
    """Camera index rank `rank` renders at `step`: perm[(step*world + rank) % n]."""
    idx = (step * world + rank) % n_views
    return int(perm[idx]) if perm is not None else idx


def fuse_optimizers(splats, optimizers: Dict[str, torch.optim.Optimizer]):
    """Wrap the reference's six per-parameter Adam instances in one fused launch."""
    from .optim import FusedAdam
    return FusedAdam(optimizers)
