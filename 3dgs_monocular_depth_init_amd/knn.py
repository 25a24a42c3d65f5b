"""Exact K-nearest-neighbour distances on the device (part of A9).

`knn(x, K)` keeps the reference's contract
(/root/reference/gs_init_compare/utils/runner_utils.py:142-146: distances [N,K]
sorted ascending, the point itself first at distance 0), where the reference
runs sklearn's NearestNeighbors on the CPU. `initial_log_scales` is
runner.py:88-91: log of the RMS distance to the 3 nearest neighbours.
"""
from __future__ import annotations

import math

import torch
from torch import Tensor

from ._lib import call, ptr

BRUTE_FORCE_BELOW = 4096


def _st():
    return torch.cuda.current_stream().cuda_stream


@torch.no_grad()
def knn(x: Tensor, K: int = 4) -> Tensor:
    if not x.is_cuda:
        from ._lib import GsrastError
        raise GsrastError("knn: tensor must be on a ROCm device; there is no CPU path")
    if K not in (4, 8):
        raise NotImplementedError("knn kernels are built for K = 4 and K = 8")
    pts = x.detach().contiguous().float()
    N = pts.shape[0]
    out = torch.empty(N, K, dtype=torch.float32, device=pts.device)
    if N <= BRUTE_FORCE_BELOW:
        call("gsr_knn_brute", N, N, K, ptr(pts), ptr(pts), ptr(out), _st())
        return out.to(x.dtype)
    # cell edge from a sample: twice the median K-th neighbour distance
    g = torch.Generator(device="cpu").manual_seed(0)
    sample = pts[torch.randint(0, N, (256,), generator=g).to(pts.device)].contiguous()
    sd = torch.empty(256, K, dtype=torch.float32, device=pts.device)
    call("gsr_knn_brute", 256, N, K, ptr(sample), ptr(pts), ptr(sd), _st())
    lo, hi = pts.min(0).values, pts.max(0).values
    extent = float((hi - lo).max())
    h = max(2.0 * float(sd[:, K - 1].median()), extent / 2.0e6, 1e-12)
    origin = lo.contiguous()
    pending = torch.arange(N, device=pts.device)
    for attempt in range(3):
        qpts = pts if attempt == 0 else pts[pending].contiguous()
        _grid_pass(pts, qpts, pending if attempt else None, origin, h, K, out)
        pending = torch.nonzero(_UNRESOLVED[0]).reshape(-1) if attempt == 0 else \
            pending[torch.nonzero(_UNRESOLVED[0]).reshape(-1)]
        if pending.numel() <= max(64, N // 200):
            break
        h *= 4.0                                   # sparse regions: retry those with coarser cells
    if pending.numel() > 0:                        # isolated points: exact brute force
        q = pts[pending].contiguous()
        res = torch.empty(q.shape[0], K, dtype=torch.float32, device=pts.device)
        call("gsr_knn_brute", q.shape[0], N, K, ptr(q), ptr(pts), ptr(res), _st())
        out[pending] = res
    return out.to(x.dtype)


_UNRESOLVED = [None]
MAX_RING = 3


def _grid_pass(pts: Tensor, queries: Tensor, query_ids, origin: Tensor, h: float, K: int, out: Tensor):
    """Bucket ALL points into cells of edge h and answer `queries` (all points on the
    first pass, the still-unresolved ones afterwards) with the ring search."""
    N = pts.shape[0]
    dev = pts.device
    keys = torch.empty(N, dtype=torch.int64, device=dev)
    call("gsr_knn_cell_keys", N, ptr(pts), ptr(origin), h, ptr(keys), _st())
    skeys, order = torch.sort(keys)
    sorted_pts = pts[order].contiguous()
    ukeys, counts = torch.unique_consecutive(skeys, return_counts=True)
    ustart = torch.zeros(ukeys.numel() + 1, dtype=torch.int64, device=dev)
    ustart[1:] = torch.cumsum(counts, 0)
    if query_ids is None:
        q, qorder, Q = sorted_pts, order, N
        res, unresolved = out, torch.empty(N, dtype=torch.uint8, device=dev)
    else:
        Q = queries.shape[0]
        q, qorder = queries, torch.arange(Q, device=dev)
        res = torch.empty(Q, K, dtype=torch.float32, device=dev)
        unresolved = torch.empty(Q, dtype=torch.uint8, device=dev)
    call("gsr_knn_grid", Q, K, ptr(q), ptr(sorted_pts), ptr(qorder), ptr(ukeys), ptr(ustart), int(ukeys.numel()),
         ptr(origin), h, MAX_RING, ptr(res), ptr(unresolved), _st())
    if query_ids is not None:
        out[query_ids] = res
    _UNRESOLVED[0] = unresolved


def initial_log_scales(points: Tensor, init_scale: float = 1.0) -> Tensor:
    """runner.py:88-91."""
    dist2_avg = (knn(points, 4)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg)
    return torch.log(dist_avg * init_scale).unsqueeze(-1).repeat(1, 3)
