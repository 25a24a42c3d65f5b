#!/usr/bin/env python3
"""Where gsr_project_fwd's time goes: the c4 launch (1 M Gaussians, 1 camera 1920x1080, SH degree 3)
timed with HIP events as the step issues it, and with parts of its work switched off through the
C ABI's own optional arguments (no tile counts, no packed records, SH degree 0, no colours at all).
One JSON line per variant: us per launch and the bytes the variant moves (algorithmic).

    python tools/bench_project.py [--gaussians 1000000] [--iters 50]
"""
import argparse
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gaussians", type=int, default=1_000_000)
    ap.add_argument("--iters", type=int, default=50)
    args = ap.parse_args()
    import torch

    from tests import scenes
    pkg = importlib.import_module("3dgs_monocular_depth_init_amd")
    L = pkg._lib
    L.load()
    call, ptr = L.call, L.ptr
    dev = torch.device("cuda", 0)
    N, W, H = args.gaussians, 1920, 1080
    sc = scenes.make_scene(N, 0)
    means, quats = sc["means"].to(dev), sc["quats"].to(dev)
    scales, opac = torch.log(sc["scales"]).to(dev), torch.logit(sc["opacities"]).to(dev)
    sh0, shN = sc["sh0"].to(dev).contiguous(), sc["shN"].to(dev).contiguous()
    vms, Ks = scenes.cameras(range(1), width=W, height=H)
    vms, Ks = vms.to(dev).contiguous(), Ks.to(dev).contiguous()
    campos = torch.linalg.inv(vms)[:, :3, 3].contiguous()
    tw, th = (W + 15) // 16, (H + 15) // 16
    f32, i32 = torch.float32, torch.int32
    radii = torch.empty(1, N, 2, dtype=i32, device=dev)
    means2d = torch.empty(1, N, 2, dtype=f32, device=dev)
    depths = torch.empty(1, N, dtype=f32, device=dev)
    conics = torch.empty(1, N, 3, dtype=f32, device=dev)
    colors = torch.empty(1, N, 3, dtype=f32, device=dev)
    opac_act = torch.empty(N, dtype=f32, device=dev)
    counts = torch.zeros(tw * th, dtype=i32, device=dev)
    records = torch.empty(N, 16, dtype=f32, device=dev)
    st = torch.cuda.current_stream().cuda_stream

    def launch(sh_degree=3, with_counts=True, with_records=True, with_colors=True):
        call("gsr_project_fwd", 1, N, ptr(means), ptr(quats), ptr(scales), ptr(opac), ptr(vms), ptr(Ks),
             ptr(campos), W, H, 0.3, 0.01, 1e10, 0.0, 0, sh_degree if with_colors else -1,
             ptr(sh0) if with_colors else None, 3, ptr(shN) if with_colors else None, 45, ptr(radii),
             ptr(means2d), ptr(depths), ptr(conics), None, ptr(colors) if with_colors else None, 3, -1, 3,
             ptr(opac_act), tw, th, ptr(counts) if with_counts else None,
             ptr(records) if with_records and with_colors else None, st)

    launch()
    torch.cuda.synchronize()
    V = int((radii[0, :, 0] > 0).sum())
    variants = {
        "full": dict(),
        "no_tile_counts": dict(with_counts=False),
        "no_records": dict(with_records=False),
        "sh_degree_0": dict(sh_degree=0),
        "no_colors_no_records": dict(with_colors=False),
        "no_colors_no_records_no_counts": dict(with_colors=False, with_counts=False),
    }
    for name, kw in variants.items():
        for _ in range(5):
            launch(**kw)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(args.iters):
            launch(**kw)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / args.iters
        deg = -1 if not kw.get("with_colors", True) else kw.get("sh_degree", 3)
        rd = 44 * N + (12 + 12 * ((deg + 1) ** 2 - 1)) * V * (deg >= 0)
        wr = (8 + 8 + 4 + 12 + 4) * N + 12 * N * (deg >= 0) \
            + 48 * V * (deg >= 0 and kw.get("with_records", True))
        print(json.dumps({"kernel": "project_fwd_kernel", "variant": name, "us": us, "N": N, "visible": V,
                          "algorithmic_MB": (rd + wr) / 1e6, "TBps": (rd + wr) / us / 1e6}), flush=True)


if __name__ == "__main__":
    main()
