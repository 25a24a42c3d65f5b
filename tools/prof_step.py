"""A few c4 iterations (fwd+bwd, no optimizer) for rocprofv3 counter passes."""
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
from tests import scenes  # noqa: E402

runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
N = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 3
sc = scenes.make_scene(N, 0)
splats, _ = runner.create_splats_with_optimizers(
    sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
vm, K = scenes.cameras([0])
c2w = torch.linalg.inv(vm).cuda()
K = K.cuda()
target = torch.rand(1, 1080, 1920, 3, device="cuda")
for k in range(iters):
    runner.train_step(splats, None, c2w, K, target, step=10_000)
    for p in splats.values():
        p.grad = None
torch.cuda.synchronize()
print("done")
