#!/usr/bin/env python3
"""How many (tile, Gaussian) pairs of the compositing backward contribute NOTHING (no pixel of the tile passes the
alpha >= 1/255 test, so the pair's lane reduction and LDS row are all zeros)? Needs the diagnostic build:
    bash tools/build_variants.sh count "-DGSR_BWD_COUNT_EMPTY=1"
    GSRAST_LIB=3dgs_monocular_depth_init_amd/lib/variants/libgsrast_count.so python tools/bwd_empty_pairs.py
One JSON line for the c4 step, tight lists (the bench's configuration) and gsplat's rectangle-rule lists."""
import ctypes
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import scenes  # noqa: E402

runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
lib = L.load()
N = 1_000_000
sc = scenes.make_scene(N, 0)
splats, _ = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                 opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
vm, K = scenes.cameras([0])
c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
target = torch.rand(1, 1080, 1920, 3, device="cuda")
out = (ctypes.c_ulonglong * 4)()
fn = lib.gsr_debug_bwd_counts
fn.argtypes = [ctypes.POINTER(ctypes.c_ulonglong), ctypes.c_int]
for tight in (True, False):
    cfg = runner.RasterConfig(tight_tiles=tight)
    runner.train_step(splats, None, c2w, K, target, step=10_000, cfg=cfg)      # warm-up (sizes the lists)
    for p in splats.values():
        p.grad = None
    fn(out, 1)
    _, info = runner.train_step(splats, None, c2w, K, target, step=10_000, cfg=cfg)
    fn(out, 1)
    for p in splats.values():
        p.grad = None
    print(json.dumps({"lists": "tight" if tight else "rectangle rule", "listed_pairs": int(info["flatten_ids"].numel()),
                      "pairs_composited_by_the_backward": out[0], "pairs_with_no_valid_pixel": out[1],
                      "fraction_empty": out[1] / max(out[0], 1), "pairs_in_fast_batches": out[2],
                      "fraction_fast": out[2] / max(out[0], 1)}), flush=True)
