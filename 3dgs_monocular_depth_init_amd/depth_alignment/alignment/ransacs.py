"""LO-RANSAC / MSAC scale-shift alignment with batched hypothesis scoring.

Mirror of /root/reference/gs_init_compare/depth_alignment/alignment/ransacs.py:100-189.
The reference runs up to 2 500 Python iterations, each ~10 tiny launches. Here
a chunk of iterations is evaluated in six launches: sample sums -> sample
hypotheses -> their losses over all M points -> inlier-restricted sums ->
local-optimisation hypotheses -> their losses. The sequential accept rule
(ransacs.py:139-162: strict improvement of the SAMPLE loss gates the LO step,
strict improvement of the LO loss gates acceptance, adaptive stop from the
inlier count) is then replayed on the host over the per-hypothesis scalars,
so the decisions are the reference's.

RNG: the reference draws `torch.randperm(num_samples)[:sample_size]` from the
global CPU generator once per executed iteration (line 131). The same calls
are made here; if the replay stops early inside a chunk, the generator state
is rewound to what the reference would have left.
"""
import math
from pathlib import Path
from typing import Optional

import torch

from ..._lib import call, ptr
from ..config import RansacConfig
from ..interface import DepthAlignmentResult, DepthAlignmentStrategy
from .lstsqrs import apply_scale_shift, gather_depth

CHUNK = 256          # largest chunk of iterations evaluated per round trip
FIRST_CHUNK = 16     # chunks grow 16, 32, ... CHUNK: the adaptive stop usually fires within the
                     # first dozens of iterations and every drawn randperm costs host time


def _st():
    return torch.cuda.current_stream().cuda_stream


def _required_samples(inlier_count, total, min_sample_size, confidence):   # ransacs.py:79-91
    """k = log(eta) / log(1 - eps^m). In the reference `inlier_count` is a 0-dim
    int64 TENSOR once a hypothesis has been accepted (ransacs.py:150), so the
    ratio, its power and `1 - x` are evaluated in fp32: for inlier ratios below
    ~1.6 % `1 - ratio**4` rounds to 1.0, log() is 0, the ZeroDivisionError branch
    returns 0 and the loop stops at once. That behaviour is part of the
    reference's results, so the same expression is evaluated on a CPU tensor."""
    if not isinstance(inlier_count, torch.Tensor):
        inlier_count = torch.tensor(int(inlier_count)) if inlier_count else 0
    inlier_ratio = inlier_count / total
    try:
        return math.log(1 - confidence) / math.log(1 - inlier_ratio ** min_sample_size)
    except (ZeroDivisionError, ValueError):
        return 0


def _evaluate_chunk(d, g, sample_idx, thr):
    """For T sampled index sets return per-hypothesis host arrays:
    (h_sample [T,2], loss_r, loss_m, h_lo [T,2], lo_loss_r, lo_loss_m, lo_inliers)."""
    dev = d.device
    T, S = sample_idx.shape
    M = d.numel()
    st = _st()
    idx = sample_idx.to(dev)
    sums = torch.empty(T, 5, dtype=torch.float64, device=dev)
    h_s = torch.empty(T, 2, dtype=torch.float32, device=dev)
    h_lo = torch.empty(T, 2, dtype=torch.float32, device=dev)
    o_r = torch.empty(2, T, dtype=torch.int32, device=dev)
    o_m = torch.empty(2, T, dtype=torch.float32, device=dev)
    o_i = torch.empty(2, T, dtype=torch.int32, device=dev)
    call("gsr_lsq_sums", T, M, 1, ptr(d), ptr(g), ptr(idx), S, None, 0.0, ptr(sums), st)
    call("gsr_solve_scale_shift", T, ptr(sums), ptr(h_s), st)
    call("gsr_ransac_score", T, M, ptr(h_s), ptr(d), ptr(g), thr, ptr(o_r[0]), ptr(o_m[0]),
         ptr(o_i[0]), st)
    call("gsr_lsq_sums", T, M, 2, ptr(d), ptr(g), None, 0, ptr(h_s), thr, ptr(sums), st)
    call("gsr_solve_scale_shift", T, ptr(sums), ptr(h_lo), st)
    call("gsr_ransac_score", T, M, ptr(h_lo), ptr(d), ptr(g), thr, ptr(o_r[1]), ptr(o_m[1]),
         ptr(o_i[1]), st)
    return (h_s.cpu(), o_r[0].cpu(), o_m[0].cpu(), h_lo.cpu(), o_r[1].cpu(), o_m[1].cpu(),
            o_i[1].cpu())


def _align_depth_ransac_generic(predicted_depth, gt_points_camera_coords, gt_depth, loss_name: str,
                                config: RansacConfig, debug_export_dir: Optional[Path] = None,
                                return_stats: bool = False):
    depth_map = predicted_depth.depth.contiguous().float()
    coords = gt_points_camera_coords.contiguous().long()
    g = gt_depth.contiguous().float()
    d = gather_depth(depth_map, coords)
    num_samples = d.numel()
    p = config
    h_best_lo = None
    num_inliers_best_lo = 0
    loss_best_lo = float("inf")
    loss_best_sample = float("inf")
    iteration = -1
    done = False
    base = 0
    required = 0          # _required_samples(0, ...) == 0: ZeroDivisionError branch
    chunk = FIRST_CHUNK
    while base < p.max_iters and not done:
        T = min(chunk, p.max_iters - base)
        chunk = min(2 * chunk, CHUNK)
        rng_state = torch.get_rng_state()
        sample_idx = torch.stack([torch.randperm(num_samples)[: p.sample_size] for _ in range(T)])
        h_s, l_r, l_m, h_lo, lo_r, lo_m, lo_in = _evaluate_chunk(d, g, sample_idx, p.inlier_threshold)
        ls = (l_r if loss_name == "ransac" else l_m).tolist()
        ll = (lo_r if loss_name == "ransac" else lo_m).tolist()
        lin = lo_in.tolist()
        for k in range(T):
            iteration = base + k
            if ls[k] < loss_best_sample:                       # ransacs.py:139
                if ll[k] < loss_best_lo:                       # ransacs.py:145
                    h_best_lo = h_lo[k]
                    loss_best_lo = ll[k]
                    loss_best_sample = ls[k]
                    num_inliers_best_lo = lin[k]
                    required = _required_samples(num_inliers_best_lo, num_samples, p.sample_size,
                                                 p.confidence)
            if required <= iteration and h_best_lo is not None and iteration >= p.min_iters:
                done = True
                if k + 1 < T:      # leave the global RNG where the reference would
                    torch.set_rng_state(rng_state)
                    for _ in range(k + 1):
                        torch.randperm(num_samples)
                break
        base += T
    if h_best_lo is None:
        raise RuntimeError("RANSAC produced no hypothesis (max_iters == 0?)")
    print(f"[RANSAC] Iterations: {iteration}, Inliers: {num_inliers_best_lo}/{num_samples}, "
          f"best scale: {float(h_best_lo[0])}, best shift: {float(h_best_lo[1])}")
    aligned = apply_scale_shift(depth_map, h_best_lo.to(depth_map.device))
    res = DepthAlignmentResult(aligned_depth=aligned, mask=predicted_depth.mask)
    if return_stats:
        return res, dict(iterations=iteration, inliers=int(num_inliers_best_lo),
                         scale=float(h_best_lo[0]), shift=float(h_best_lo[1]))
    return res


class DepthAlignmentRansac(DepthAlignmentStrategy):
    @classmethod
    def align(cls, predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
              debug_export_dir=None, *args, **kwargs):
        return _align_depth_ransac_generic(predicted_depth, sfm_points_camera_coords,
                                           sfm_points_depth, "ransac", _ransac_cfg(config),
                                           debug_export_dir)


class DepthAlignmentMsac(DepthAlignmentStrategy):
    @classmethod
    def align(cls, predicted_depth, sfm_points_camera_coords, sfm_points_depth, config,
              debug_export_dir=None, *args, **kwargs):
        return _align_depth_ransac_generic(predicted_depth, sfm_points_camera_coords,
                                           sfm_points_depth, "msac", _ransac_cfg(config),
                                           debug_export_dir)


def _ransac_cfg(config) -> RansacConfig:
    """Accept the reference's nested Config (config.mdi.alignment.ransac) or a RansacConfig."""
    if isinstance(config, RansacConfig):
        return config
    return config.mdi.alignment.ransac
