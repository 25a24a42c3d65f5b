#!/usr/bin/env python3
"""Which path does the tile-list sort take, bucket by bucket? Needs the diagnostic build:
    bash tools/build_variants.sh sortpaths "-DGSR_SORT_COUNT_PATHS=1"
    GSRAST_LIB=3dgs_monocular_depth_init_amd/lib/variants/libgsrast_sortpaths.so python tools/sort_paths.py
One JSON line per scene: the c4 scene (uniform depths), the same on two thin shells (with / without outliers), and the
three scenes of tests/test_gpu_rasterization.py::test_clustered_depths_take_the_equalised_bins."""
import ctypes
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import scenes  # noqa: E402

R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
lib = L.load()
fn = lib.gsr_debug_sort_paths
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
out = (ctypes.c_ulonglong * 8)()
NAMES = ["buckets", "equalised_parked", "equalised_streamed", "several_groups", "networks", "global_network"]


def run(name, sc, vm, K, W, H):
    dev = {k: v.cuda() for k, v in sc.items()}
    col = torch.cat([dev["sh0"], dev["shN"]], 1)
    fn(out, 1)
    with torch.no_grad():
        _, _, meta = R.rasterization(dev["means"], dev["quats"], dev["scales"], dev["opacities"], col, vm.cuda(), K.cuda(), W, H,
                                     sh_degree=1, packed=False)
    fn(out, 1)
    print(json.dumps({"scene": name, "pairs": int(meta["flatten_ids"].shape[0]), **{n: int(out[i]) for i, n in enumerate(NAMES)}}), flush=True)


def shells(sc, g, thick):
    n = len(sc["means"])
    sc["means"][:, 2] = torch.where(torch.rand(n, generator=g) < 0.5, -0.6, 0.6) + thick * torch.randn(n, generator=g)


g = torch.Generator().manual_seed(11)
vm, K = scenes.cameras([7])
sc = scenes.make_scene(1_000_000, 3)
run("c4 uniform", sc, vm, K, 1920, 1080)
shells(sc, g, 0.01)
run("c4 two shells", sc, vm, K, 1920, 1080)
sc["means"][:2000, 2] = -1.0 + 60.0 * torch.rand(2000, generator=g)
run("c4 two shells + outliers", sc, vm, K, 1920, 1080)
for case in ("short_buckets", "long_bucket", "short_buckets_outliers"):
    g = torch.Generator().manual_seed(17)
    N, W, H, box = (18000, 128, 16, (1.8, 0.2, 0.3)) if case == "long_bucket" else (30000, 256, 64, (3.8, 0.9, 0.3))
    sc = scenes.make_scene(N, 12, box=box, scale_mean=0.002)
    shells(sc, g, 0.001)
    if case.endswith("outliers"):
        sc["means"][:60, 2] = -1.5 + 25.0 * torch.rand(60, generator=g)
    vm1 = torch.eye(4)[None].clone()
    vm1[0, 2, 3] = 2.0
    K1 = torch.tensor([[[60.0, 0, W / 2], [0, 60.0, H / 2], [0, 0, 1]]])
    run("test scene " + case, sc, vm1, K1, W, H)
