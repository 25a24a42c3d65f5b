#!/usr/bin/env python3
"""Throughput of the MFMA depth network (BASELINE configs c3 = Metric3D-small,
c5 = Metric3D-large) at the reference's input size 616x1064 (3349 tokens), deterministic random
weights (the real ones are a remote download). Prints one JSON line per backbone: ms/image,
dense TFLOP/s of encoder and whole net against the fp16 MFMA peak of MI355X_MICROARCH.md.

    python tools/bench_depthnet.py [--backbones vits,vitl] [--iters 5]
"""
import argparse
import importlib
import json
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
MFMA_FP16_PEAK_TFLOPS = 2500.0      # dense, MI355X_MICROARCH.md


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbones", default="vits,vitl")
    ap.add_argument("--iters", type=int, default=5)
    ap.add_argument("--batches", default="1,2,4", help="batch sizes of inference() (the init loop's predict_depths uses 4)")
    args = ap.parse_args()
    import torch

    from tests import test_gpu_depthnet as T
    from tests.golden import dn_weights as DW
    N = importlib.import_module("3dgs_monocular_depth_init_amd.depth_prediction.predictors.metric3d_net")
    img = DW.image(616, 1064).cuda()
    for bb in args.backbones.split(","):
        cfg = N.CONFIGS[bb]
        net = N.Metric3DNet(T._state(cfg), backbone=bb, device="cuda")
        net.inference({"input": img})                       # warm-up: eager pass + graph capture
        net.flop_count = 0.0
        tok = net.encode(img)
        enc_flops = net.flop_count
        net.decode(tok)
        all_flops, net.flop_count = net.flop_count, None
        torch.cuda.synchronize()

        def timed(fn):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(args.iters):
                fn()
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / args.iters

        t_enc = timed(lambda: net.encode(img))
        t_all = timed(lambda: net.inference({"input": img}))            # graph replay
        net.use_graph = False
        t_eager = timed(lambda: net.inference({"input": img}))
        net.use_graph = True
        d, c, o = net.inference({"input": img})
        by_batch = {}
        for B in [int(b) for b in args.batches.split(",") if int(b) > 1]:
            imgs = torch.cat([img] * B, 0)
            net.inference({"input": imgs})                  # eager pass + capture of the B-image graph
            tB = timed(lambda: net.inference({"input": imgs}))
            by_batch[str(B)] = {"ms_per_image": tB / B * 1e3, "total_tflops": all_flops * B / tB / 1e12,
                                "frac_of_fp16_mfma_peak": all_flops * B / tB / 1e12 / MFMA_FP16_PEAK_TFLOPS}
        print(json.dumps({
            "metric": f"Metric3D-{bb} depth network, 616x1064, fp16 MFMA", "backbone": bb,
            "ms_per_image": t_all * 1e3, "images_per_s": 1.0 / t_all,
            "eager_ms_per_image": t_eager * 1e3, "encoder_ms_eager": t_enc * 1e3,
            "encoder_tflop": enc_flops / 1e12, "total_tflop": all_flops / 1e12,
            "encoder_tflops": enc_flops / t_enc / 1e12, "total_tflops": all_flops / t_all / 1e12,
            "frac_of_fp16_mfma_peak": all_flops / t_all / 1e12 / MFMA_FP16_PEAK_TFLOPS,
            "batched": by_batch,
            "finite": bool(torch.isfinite(d).all() and torch.isfinite(o["prediction_normal"]).all()),
            "weights": "deterministic random (structurally pinned)"}), flush=True)
        del net
        torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
