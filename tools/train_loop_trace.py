#!/usr/bin/env python3
"""Kernel sequence of ONE iteration of runner.train (default config: DefaultStrategy statistics every step, L1 + SSIM,
fused Adam) on a 200 k-Gaussian scene at 1080p, from a rocprofv3 kernel trace:
   cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tl -o tl -- python3 /root/repo/tools/train_loop_trace.py
   python3 /root/repo/tools/train_loop_trace.py --digest /tmp/tl/tl_kernel_trace.csv"""
import csv
import importlib
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
if "--digest" in sys.argv:
    rows = list(csv.DictReader(open(sys.argv[sys.argv.index("--digest") + 1])))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    idx = [i for i, r in enumerate(rows) if "project_fwd_kernel" in r["Kernel_Name"]]
    a, b = idx[-6], idx[-5]
    t0, prev = int(rows[a]["Start_Timestamp"]), None
    small = 0.0
    for r in rows[a:b]:
        s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
        name = r["Kernel_Name"]
        if "gsr::" not in name:
            small += (e - s) / 1e3
        print("%8.1f us  dur %7.1f  gap %6.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev) / 1e3 if prev else 0.0, name[:110]))
        prev = e
    print("iteration span %.1f us, non-gsr kernels %.1f us" % ((int(rows[b]["Start_Timestamp"]) - t0) / 1e3, small))
    sys.exit(0)
import torch  # noqa: E402
from tests import scenes  # noqa: E402

P = "3dgs_monocular_depth_init_amd."
runner = importlib.import_module(P + "runner")
cfgm = importlib.import_module(P + "config")
N, W, H = (200_037 if "--odd" in sys.argv else 200_000), 1920, 1080
sc = scenes.make_scene(N, 3)
vms, Ks = scenes.cameras(range(0, 100, 10), width=W, height=H, f=1200.0)
c2ws = torch.linalg.inv(vms).contiguous().cuda()
Ks = Ks.cuda()
frames = [{"camtoworld": c2ws[i], "K": Ks[i], "image": torch.rand(H, W, 3, device="cuda") * 255.0, "image_id": i} for i in range(len(vms))]
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
cfg = cfgm.Config()
cfg.max_steps = 60
cfg.save_steps, cfg.eval_steps = [], []
cfg.strategy.refine_start_iter, cfg.strategy.refine_stop_iter, cfg.strategy.reset_every = 10 ** 9, 10 ** 9 + 5, 10 ** 9
runner.train(splats, opts, frames, cfg, progress_every=1000)
torch.cuda.synchronize()
