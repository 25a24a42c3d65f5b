"""Projection kernel times vs SH degree (experiment)."""
import importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import scenes
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
sc = {k: v.cuda().requires_grad_(True) for k, v in scenes.make_scene(1_000_000, 0).items()}
vm, K = scenes.cameras([0]); vm, K = vm.cuda(), K.cuda()
w = torch.rand(1, 1080, 1920, 3, device="cuda")
for deg in (0, 1, 2, 3):
    for it in range(6):
        if it == 2:
            lib.TIMERS = {}
        rc, ra, meta = R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]), vm, K, 1920, 1080, sh_degree=deg)
        (rc * w).sum().backward()
        for v in sc.values(): v.grad = None
    torch.cuda.synchronize()
    t = lib.kernel_times_ms(); lib.TIMERS = None
    print("deg", deg, {k: round(v[1], 4) for k, v in t.items() if "project" in k})
