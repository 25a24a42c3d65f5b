"""Shared loader for tests/golden/points_golden.npz (outputs of the reference's own
points_from_depth.py / pipeline.py, see tests/golden/make_points_golden.py). Inputs are
stored in the fixture as well (randn / matmul are not bit-reproducible across CPU models)."""
from pathlib import Path

import numpy as np
import torch

from tests.golden.make_points_golden_scene import scene_rgb

G = np.load(Path(__file__).resolve().parent / "golden" / "points_golden.npz")
SUB = (5, 7)           # lattice on which the aligned depth maps were stored


def scene(key: str):
    """Inputs of one fixture case (stored tensors; colours rebuilt by exact integer arithmetic)."""
    if f"{key}_scene_of" in G:
        key = f"b9_{int(G[f'{key}_scene_of'])}"
    depth = t(f"{key}_depth")
    H, W = depth.shape
    sc = {"depth": depth, "rgb": scene_rgb(H, W)}
    if f"{key}_mask" in G:
        sc["mask"] = bits(f"{key}_mask", H * W).view(H, W)
        for k in ("sfm", "P", "K", "c2w"):
            sc[k] = t(f"{key}_{k}")
    return sc


def bits(name: str, n: int) -> torch.Tensor:
    return torch.from_numpy(np.unpackbits(G[name])[:n].astype(bool))


def t(name: str) -> torch.Tensor:
    return torch.from_numpy(G[name])


def b9_cfg(i: int):
    aligner, factor, grad_thr, nsfm = [str(x) for x in G[f"b9_{i}_cfg"]]
    factor = factor if factor == "adaptive" else int(factor)
    grad_thr = None if grad_thr == "None" else float(grad_thr)
    return aligner, factor, grad_thr, bool(int(nsfm))
