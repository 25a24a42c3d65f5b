#!/usr/bin/env python3
"""The projection backward + Adam over W cameras per Gaussian -- what every rank of the view-parallel row exchange runs
(distributed.GatherRowsSync: rows of all ranks gathered, `gsr_project_bwd_adam(C = W, packed 36-byte rows)`) -- timed on
ONE GPU with synthetic gathered rows (a real step's rows of camera 0, replicated with the other cameras' poses).
    python tools/bench_project_bwd_multi.py [--world 8] [--gaussians 1000000]"""
import argparse
import ctypes as C
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402
from tests import scenes  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--world", default="1,2,4,8")
ap.add_argument("--gaussians", type=int, default=1_000_000)
ap.add_argument("--iters", type=int, default=20)
args = ap.parse_args()
runner = importlib.import_module("3dgs_monocular_depth_init_amd.runner")
D = importlib.import_module("3dgs_monocular_depth_init_amd.distributed")
R = importlib.import_module("3dgs_monocular_depth_init_amd.rendering")
L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
N, Wd, Hd = args.gaussians, 1920, 1080
sc = scenes.make_scene(N, 0)
splats, opts = runner.create_splats_with_optimizers(sc["means"], torch.rand(N, 3), torch.log(sc["scales"]), quats=sc["quats"],
                                                    opacities_logit=torch.logit(sc["opacities"]), shN=sc["shN"])
fused = D.fuse_optimizers(splats, opts)
for Wn in [int(w) for w in args.world.split(",")]:
    vms, Ks = scenes.cameras(range(Wn), width=Wd, height=Hd)
    c2w, Ks = torch.linalg.inv(vms).contiguous().cuda(), Ks.cuda()
    target = torch.rand(1, Hd, Wd, 3, device="cuda")
    # real rows per camera: render each camera once, pack its rows
    packed = torch.empty(Wn * N, 9, dtype=torch.float32, device="cuda")
    seen = []
    orig = R._rows_from_grads

    def spy(*a, **k):
        out = orig(*a, **k)
        seen.append(out[0])
        return out

    R._rows_from_grads = spy
    try:
        for c in range(Wn):
            _, info = runner.train_step(splats, None, c2w[c:c + 1], Ks[c:c + 1], target, step=10_000)
            rows = seen.pop()
            L.call("gsr_pack_grad_rows", N, rows.data_ptr(), info["radii"].data_ptr(), packed[c * N:].data_ptr(),
                   torch.cuda.current_stream().cuda_stream)
            for p in splats.values():
                p.grad = None
    finally:
        R._rows_from_grads = orig
    viewmats, campos = R.inverse4x4(c2w, translation_of="input")
    opac_act = torch.sigmoid(splats["opacities"].detach()).contiguous()
    st = torch.cuda.current_stream().cuda_stream

    def launch():
        P, M, V, ss, bc2, b1, b2, eps = fused.claim((splats["means"], splats["quats"], splats["scales"], splats["opacities"],
                                                     splats["sh0"], splats["shN"]))
        fused._claimed = False
        L.call("gsr_project_bwd_adam", Wn, N, viewmats.data_ptr(), Ks.data_ptr(), campos.data_ptr(), Wd, Hd, 0.3, 3, None,
               packed.data_ptr(), 9, None, None, -1, 3, opac_act.data_ptr(), P, M, V, ss, bc2, b1, b2, eps, st)

    for _ in range(3):
        launch()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(args.iters):
        launch()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / args.iters
    alg = N * (59 * 4 * 6 + 4) + Wn * N * 36
    print(json.dumps({"cameras": Wn, "ms": round(ms, 4), "algorithmic_GB": round(alg / 1e9, 3), "TBps": round(alg / ms / 1e9, 2)}), flush=True)
