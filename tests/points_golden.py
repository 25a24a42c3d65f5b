"""Shared loader for tests/golden/points_golden.npz (outputs of the reference's own
points_from_depth.py / pipeline.py, see tests/golden/make_points_golden.py). Inputs are
regenerated with our seeded scene generator and verified against recorded checksums."""
from pathlib import Path

import numpy as np
import torch

from tests.golden.make_points_golden_scene import camera_scene

G = np.load(Path(__file__).resolve().parent / "golden" / "points_golden.npz")
SUB = (5, 7)           # lattice on which the aligned depth maps were stored


def scene(key: str):
    H, W, M, seed, frac = G[f"{key}_scene"]
    sc = camera_scene(int(H), int(W), int(M), int(seed), frac_outside=float(frac))
    if f"{key}_check" in G:
        chk = [sc[k].double().sum().item() for k in ("depth", "mask", "rgb", "sfm", "P")]
        assert np.allclose(chk, G[f"{key}_check"], rtol=1e-12), "camera_scene() no longer reproduces the fixture inputs"
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
