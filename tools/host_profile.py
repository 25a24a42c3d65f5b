#!/usr/bin/env python3
"""cProfile of the Python side of the c4 training step (tiny scene: the GPU is idle, what is timed is the host)."""
import cProfile
import importlib
import pstats
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
N = 2000
sc = scenes.make_scene(N, 0)
W, H = 1920, 1080
vm, K = scenes.cameras(list(range(8)), width=W, height=H)
c2w, K = torch.linalg.inv(vm).cuda(), K.cuda()
target = torch.rand(1, H, W, 3).cuda()
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts)
fused.fuse_into_backward(True)
cfg = runner.RasterConfig()
for i in range(20):
    runner.train_step(splats, fused, c2w[i % 8:i % 8 + 1], K[i % 8:i % 8 + 1], target, step=5000 + i, cfg=cfg)
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for i in range(300):
    runner.train_step(splats, fused, c2w[i % 8:i % 8 + 1], K[i % 8:i % 8 + 1], target, step=5000 + i, cfg=cfg)
torch.cuda.synchronize()
pr.disable()
st = pstats.Stats(pr)
st.sort_stats("cumulative").print_stats(28)
