"""Dense-scene timing (experiment): lists per bucket above the LDS sorter's capacity."""
import importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import scenes
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
for (W, H, f) in ((1920, 1080, 1200.0), (960, 540, 600.0), (640, 360, 400.0)):
    sc = {k: v.cuda().requires_grad_(True) for k, v in scenes.make_scene(N, 0).items()}
    vm, K = scenes.cameras([0], width=W, height=H, f=f); vm, K = vm.cuda(), K.cuda()
    w = torch.rand(1, H, W, 3, device="cuda")
    for it in range(5):
        if it == 2:
            lib.TIMERS = {}
        rc, ra, meta = R.rasterization(sc["means"], sc["quats"], sc["scales"], sc["opacities"], (sc["sh0"], sc["shN"]), vm, K, W, H, sh_degree=3)
        (rc * w).sum().backward()
        for v in sc.values(): v.grad = None
    torch.cuda.synchronize()
    t = lib.kernel_times_ms(); lib.TIMERS = None
    off = meta["isect_offsets"].reshape(-1).long()
    I = meta["flatten_ids"].numel()
    L = torch.diff(torch.cat([off, torch.tensor([I], device=off.device)]))
    print(f"{W}x{H}: I={I} max tile list {int(L.max())} mean {float(L.float().mean()):.0f}", {k: round(v[1], 3) for k, v in t.items()})
