"""A few c4 iterations for rocprofv3 counter passes: the HEADLINE step of bench.py (forward, L1, full backward
with the Adam update fused into the projection backward: `project_bwd_kernel<true>`), then the SURVEY 8d step
(six gradient tensors written, no optimizer: `project_bwd_kernel<false>`), so that one pass has counters for both
instantiations. Usage: python3 tools/prof_step.py [N] [iters] [--ssim]"""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import scenes  # noqa: E402

runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
args = [a for a in sys.argv[1:] if not a.startswith("--")]
N = int(args[0]) if len(args) > 0 else 1_000_000
iters = int(args[1]) if len(args) > 1 else 3
ssim = 0.2 if "--ssim" in sys.argv else 0.0
sc = scenes.make_scene(N, 0)
splats, opts = runner.create_splats_with_optimizers(
    sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts)
vm, K = scenes.cameras([0])
c2w = torch.linalg.inv(vm).cuda()
K = K.cuda()
target = torch.rand(1, 1080, 1920, 3, device="cuda")
fused.fuse_into_backward(True)
try:
    for k in range(iters):
        runner.train_step(splats, fused, c2w, K, target, step=10_000 + k, ssim_lambda=ssim)
finally:
    R.set_backward_optimizer(None)
for k in range(iters):
    runner.train_step(splats, None, c2w, K, target, step=10_000, ssim_lambda=ssim)
    for p in splats.values():
        p.grad = None
torch.cuda.synchronize()
print("done")
