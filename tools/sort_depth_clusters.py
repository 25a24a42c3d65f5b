#!/usr/bin/env python3
"""Tile-list kernels on depth distributions the sort's linear depth bins do not like: the c4 scene (1 M Gaussians,
1080p) with its depths (a) uniform in the box, as benchmarked; (b) on two thin shells; (c) on two thin shells plus 0.2 %
outliers spread over 30x the depth range (every bucket's [min, max] is then set by an outlier).
Run under rocprofv3 --kernel-trace --stats, or alone (prints event timings of the whole forward):
   python tools/sort_depth_clusters.py [uniform|shells|shells_outliers]"""
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
case = sys.argv[1] if len(sys.argv) > 1 else "uniform"
N, W, H = 1_000_000, 1920, 1080
sc = scenes.make_scene(N, 3)
g = torch.Generator().manual_seed(11)
if case != "uniform":
    shell = torch.where(torch.rand(N, generator=g) < 0.5, -0.6, 0.6)
    sc["means"][:, 2] = shell + 0.01 * torch.randn(N, generator=g)
if case == "shells_outliers":
    k = N // 500
    sc["means"][:k, 2] = -1.0 + 60.0 * torch.rand(k, generator=g)
vm, K = scenes.cameras([7], width=W, height=H, f=1200.0)
dev = {k: v.cuda() for k, v in sc.items()}
col = torch.cat([dev["sh0"], dev["shN"]], 1)
ts = []
with torch.no_grad():
    for i in range(25):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc, ra, meta = R.rasterization(dev["means"], dev["quats"], dev["scales"], dev["opacities"], col, vm.cuda(), K.cuda(), W, H,
                                       sh_degree=3, packed=False)
        b.record()
        torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
print(json.dumps({"case": case, "pairs": int(meta["flatten_ids"].shape[0]), "forward_ms_median": sorted(ts[5:])[10]}))
