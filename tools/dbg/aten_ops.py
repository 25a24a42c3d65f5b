"""Which aten ops launch kernels inside one training step (experiment)."""
import importlib, sys
from pathlib import Path
import torch
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from tests import scenes
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
N = 200_000
sc = scenes.make_scene(N, 0)
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"], opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts)
fused.fuse_into_backward(True)
vm, K = scenes.cameras(range(4)); c2w = torch.linalg.inv(vm).cuda(); K = K.cuda()
target = torch.rand(1, 1080, 1920, 3, device="cuda")
for k in range(3):
    runner.train_step(splats, fused, c2w[k:k+1], K[k:k+1], target, step=5000 + k)
torch.cuda.synchronize()
from torch.profiler import profile, ProfilerActivity
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    runner.train_step(splats, fused, c2w[3:4], K[3:4], target, step=5003)
    torch.cuda.synchronize()
for e in prof.events():
    if e.device_type.name == "CPU" and e.name.startswith("aten::") and any(c.name for c in []) is False:
        pass
evs = [e for e in prof.events() if e.name.startswith("aten::") and e.cuda_time_total > 0 or ("Memset" in e.name or "Memcpy" in e.name)]
seen = []
for e in evs:
    st = ""
    if e.stack:
        st = " <- " + " | ".join(s for s in e.stack if "repo" in s)[:200]
    print(f"{e.name:40s} cuda {e.cuda_time_total:7.1f} us shapes{st}")
