"""c2 timing (experiment): 100k Gaussians, 4 cameras per step at 1080p."""
import importlib, sys, time
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import scenes
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
Cn = int(sys.argv[2]) if len(sys.argv) > 2 else 4
sc = scenes.make_scene(N, 0)
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts); fused.fuse_into_backward(True)
vm, K = scenes.cameras(range(0, 100, 100 // 8)); c2w = torch.linalg.inv(vm).contiguous().cuda(); K = K.cuda()
target = torch.rand(Cn, 1080, 1920, 3, device="cuda")
def step(k):
    i = (k * Cn) % 8
    runner.train_step(splats, fused, c2w[i:i+Cn], K[i:i+Cn], target, step=5000 + k)
for k in range(5): step(k)
torch.cuda.synchronize(); lib.TIMERS = {}
t0 = time.perf_counter()
for k in range(20): step(5 + k)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 20
t = lib.kernel_times_ms(); lib.TIMERS = None
print(f"N={N} C={Cn}: {dt*1e3:.3f} ms/step", {k: round(v[1], 3) for k, v in t.items()})
