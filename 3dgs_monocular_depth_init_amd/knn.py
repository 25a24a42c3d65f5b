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
from ._lib import current_stream as _raw_stream

BRUTE_FORCE_BELOW = 4096


def _st():
    return _raw_stream()


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


@torch.no_grad()
def knn_neighbors(x: Tensor, K: int):
    """The K nearest OTHER points of every point: (distances fp64 [N,K] ascending, indices int32
    [N,K]) -- what sklearn's `NearestNeighbors.kneighbors()` of the fitted data returns (the query
    point itself dropped, distances in float64). K <= 64, N > K. Exact: ring search over a cell
    grid, queries not proven within the ring cap retried with coarser cells, the last ones against
    one all-containing cell (= brute force)."""
    if not x.is_cuda:
        from ._lib import GsrastError
        raise GsrastError("knn_neighbors: tensor must be on a ROCm device; there is no CPU path")
    pts = x.detach().contiguous().float()
    N, dev = pts.shape[0], pts.device
    if not (1 <= K <= 64 and N > K):
        raise ValueError(f"knn_neighbors: need 1 <= K <= 64 and N > K (K={K}, N={N})")
    lo, hi = pts.min(0).values, pts.max(0).values
    origin = lo.contiguous()
    extent = max(float((hi - lo).max()), 1e-12)
    dist = torch.empty(N, K, dtype=torch.float64, device=dev)
    idx = torch.empty(N, K, dtype=torch.int32, device=dev)
    everything = 4.0 * extent                   # one cell holds the whole cloud: the ring search is a full scan

    def grid_pass(query_ids, h):
        keys = torch.empty(N, dtype=torch.int64, device=dev)
        call("gsr_knn_cell_keys", N, ptr(pts), ptr(origin), h, ptr(keys), _st())
        skeys, order = torch.sort(keys)
        sorted_pts = pts[order].contiguous()
        ukeys, counts = torch.unique_consecutive(skeys, return_counts=True)
        ustart = torch.zeros(ukeys.numel() + 1, dtype=torch.int64, device=dev)
        ustart[1:] = torch.cumsum(counts, 0)
        if query_ids is None:                   # all points, in cell order (neighbouring lanes walk the same cells)
            q, self_pos, qorder, Q = sorted_pts, torch.arange(N, device=dev), order, N
            d, ix = dist, idx
        else:
            Q = query_ids.numel()
            q = pts[query_ids].contiguous()
            inv = torch.empty(N, dtype=torch.int64, device=dev)
            inv[order] = torch.arange(N, device=dev)
            self_pos, qorder = inv[query_ids].contiguous(), None
            d = torch.empty(Q, K, dtype=torch.float64, device=dev)
            ix = torch.empty(Q, K, dtype=torch.int32, device=dev)
        unresolved = torch.empty(Q, dtype=torch.uint8, device=dev)
        call("gsr_knn_grid_idx", Q, K, ptr(q), ptr(self_pos), ptr(sorted_pts), ptr(order), ptr(qorder), ptr(ukeys),
             ptr(ustart), int(ukeys.numel()), ptr(origin), float(h), MAX_RING, ptr(d), ptr(ix), ptr(unresolved), _st())
        if query_ids is not None:
            dist[query_ids] = d
            idx[query_ids] = ix
        return torch.nonzero(unresolved).reshape(-1)

    if N <= BRUTE_FORCE_BELOW:
        left = grid_pass(None, everything)
        assert left.numel() == 0
        return dist, idx
    # cell edge from a sample: 1.3 x the median K-th neighbour distance (the 27 cells around a query
    # then hold ~15 K candidates and prove most queries)
    g = torch.Generator(device="cpu").manual_seed(0)
    sample = torch.randint(0, N, (256,), generator=g).to(dev)
    assert grid_pass(sample, everything).numel() == 0
    h = max(1.3 * float(dist[sample, K - 1].median()), extent / 2.0e6, 1e-12)
    pending = None
    for attempt in range(3):
        left = grid_pass(pending, h)
        pending = left if pending is None else pending[left]
        if pending.numel() <= max(64, N // 200):
            break
        h *= 3.0                                # sparse regions: retry those with coarser cells
    if pending.numel() > 0:
        assert grid_pass(pending, everything).numel() == 0
    return dist, idx


@torch.no_grad()
def local_outlier_factor(x: Tensor, n_neighbors: int = 40, offset: float = -1.5):
    """sklearn.neighbors.LocalOutlierFactor(n_neighbors).fit_predict(x) == -1 on the device
    (contamination="auto": offset -1.5). Returns (outlier mask bool [N], negative_outlier_factor fp64 [N])."""
    N = x.shape[0]
    K = max(1, min(int(n_neighbors), N - 1))            # _lof.py: n_neighbors_ = max(1, min(n_neighbors, n_samples - 1))
    dist, idx = knn_neighbors(x, K)
    dev = dist.device
    lrd = torch.empty(N, dtype=torch.float64, device=dev)
    nof = torch.empty(N, dtype=torch.float64, device=dev)
    out = torch.empty(N, dtype=torch.uint8, device=dev)
    call("gsr_lof", N, K, ptr(dist), ptr(idx), float(offset), ptr(lrd), ptr(nof), ptr(out), _st())
    return out.bool(), nof


def initial_log_scales(points: Tensor, init_scale: float = 1.0) -> Tensor:
    """runner.py:88-91."""
    dist2_avg = (knn(points, 4)[:, 1:] ** 2).mean(dim=-1)
    dist_avg = torch.sqrt(dist2_avg)
    return torch.log(dist_avg * init_scale).unsqueeze(-1).repeat(1, 3)
