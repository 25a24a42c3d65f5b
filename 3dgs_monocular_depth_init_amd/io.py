"""On-disk formats either side of the path, so results
interchange with the reference and its viewers.

* checkpoints: `{"step": int, "splats": state_dict}` saved as
  `ckpt_{step}_rank{r}.pt` (/root/reference/gs_init_compare/runner.py:606,637),
  loaded with `weights_only=True` and concatenated over rank shards
  (trainer.py:22-29).
* PLY: the standard 3DGS point-cloud layout the reference writes through
  `gsplat.export_splats(format="ply")` (runner.py:619-635): x y z, nx ny nz
  (zeros), f_dc_0..2, f_rest_*, opacity (logit), scale_0..2 (log), rot_0..3
  (wxyz), little-endian float32. f_rest is channel-major ([3, K-1] flattened),
  as in the original 3DGS viewer format.
* depth cache: one file per image at the reference's path
  cache_dir/model/dataset/{image_name}.pth (monocular_depth_init.py:60-87). The PAYLOAD is
  one-way only: the reference pickles the PredictedDepth object (`torch.save(depth, path)`),
  which a `weights_only=True` load refuses; this build writes and reads a plain dict of the
  dataclass fields instead. A reference-side reader needs `PredictedDepth(**torch.load(p))`
  (shown in INTEGRATION.md); files written by the reference are ignored with a warning and
  the depth is predicted again (predict_depth_or_get_cached_depth's except branch).
"""
from __future__ import annotations

from pathlib import Path
from typing import Dict, Iterable, Optional

import numpy as np
import torch

SPLAT_KEYS = ("means", "scales", "quats", "opacities", "sh0", "shN")


def save_checkpoint(splats, step: int, ckpt_dir, world_rank: int = 0) -> Path:
    path = Path(ckpt_dir) / f"ckpt_{step}_rank{world_rank}.pt"
    path.parent.mkdir(parents=True, exist_ok=True)
    state = {k: v.detach() for k, v in splats.state_dict().items()} if hasattr(splats, "state_dict") \
        else {k: v.detach() for k, v in splats.items()}
    torch.save({"step": step, "splats": state}, path)
    return path


def load_checkpoints(files: Iterable, device="cpu") -> Dict:
    """trainer.py:22-29: load rank shards with weights_only=True and concatenate."""
    ckpts = [torch.load(f, map_location=device, weights_only=True) for f in files]
    splats = {k: torch.cat([c["splats"][k] for c in ckpts]) for k in ckpts[0]["splats"].keys()}
    return {"step": ckpts[0]["step"], "splats": splats}


def export_ply(splats, path) -> Path:
    means = splats["means"].detach().cpu().float().numpy()
    N = means.shape[0]
    sh0 = splats["sh0"].detach().cpu().float().numpy().reshape(N, 3)
    shN = splats["shN"].detach().cpu().float().numpy()                   # [N, K-1, 3]
    f_rest = np.transpose(shN, (0, 2, 1)).reshape(N, -1)                    # channel-major
    cols = [means, np.zeros_like(means), sh0, f_rest,
            splats["opacities"].detach().cpu().float().numpy().reshape(N, 1),
            splats["scales"].detach().cpu().float().numpy(),
            splats["quats"].detach().cpu().float().numpy()]
    names = (["x", "y", "z", "nx", "ny", "nz"] + [f"f_dc_{i}" for i in range(3)]
             + [f"f_rest_{i}" for i in range(f_rest.shape[1])] + ["opacity"]
             + [f"scale_{i}" for i in range(3)] + [f"rot_{i}" for i in range(4)])
    data = np.ascontiguousarray(np.concatenate(cols, axis=1).astype("<f4"))
    assert data.shape[1] == len(names)
    header = "ply\nformat binary_little_endian 1.0\n" + f"element vertex {N}\n" + \
        "".join(f"property float {n}\n" for n in names) + "end_header\n"
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    with open(path, "wb") as f:
        f.write(header.encode("ascii"))
        f.write(data.tobytes())
    return path


def load_ply(path) -> Dict[str, torch.Tensor]:
    with open(path, "rb") as f:
        names, n = [], 0
        while True:
            line = f.readline().decode("ascii").strip()
            if line.startswith("element vertex"):
                n = int(line.split()[-1])
            elif line.startswith("property float"):
                names.append(line.split()[-1])
            elif line == "end_header":
                break
        data = np.frombuffer(f.read(), dtype="<f4").reshape(n, len(names))
    col = {k: i for i, k in enumerate(names)}
    pick = lambda keys: torch.from_numpy(np.stack([data[:, col[k]] for k in keys], 1).copy())
    n_rest = sum(1 for k in names if k.startswith("f_rest_"))
    f_rest = pick([f"f_rest_{i}" for i in range(n_rest)]).reshape(n, 3, n_rest // 3).permute(0, 2, 1)
    return {"means": pick(["x", "y", "z"]), "sh0": pick([f"f_dc_{i}" for i in range(3)])[:, None, :],
            "shN": f_rest.contiguous(), "opacities": pick(["opacity"])[:, 0],
            "scales": pick([f"scale_{i}" for i in range(3)]), "quats": pick([f"rot_{i}" for i in range(4)])}


def depth_cache_path(cache_dir, model_name: str, dataset_name: str, image_name: str) -> Path:
    """monocular_depth_init.py:66-71: cache_dir / model.name / dataset_name / f"{image_name}.pth"
    (the image name keeps its extension, e.g. `DSC0001.JPG.pth`)."""
    return Path(cache_dir) / model_name / dataset_name / f"{image_name}.pth"


def save_predicted_depth(pred, path) -> None:
    path = Path(path)
    path.parent.mkdir(parents=True, exist_ok=True)
    torch.save({k: (v.cpu() if isinstance(v, torch.Tensor) else v) for k, v in vars(pred).items()}, path)


def load_predicted_depth(path, device="cpu"):
    from .depth_prediction.predictors.depth_predictor_interface import PredictedDepth
    d = torch.load(path, map_location=device, weights_only=True)
    return PredictedDepth(**d)
