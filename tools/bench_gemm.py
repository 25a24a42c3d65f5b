#!/usr/bin/env python3
"""Time gsr_dn_gemm / gsr_dn_attention alone at the Metric3D shapes: TFLOP/s per shape vs the fp16
MFMA peak (2.5 PFLOP/s dense, MI355X_MICROARCH.md). One JSON line per shape."""
import importlib
import json
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
PEAK = 2500.0


def timeit(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3


def main():
    st = torch.cuda.current_stream().cuda_stream
    shapes = [(3349, 3072, 1024, "vitl qkv"), (3349, 1024, 1024, "vitl proj"), (3349, 4096, 1024, "vitl fc1"),
              (3349, 1024, 4096, "vitl fc2"), (3349, 1152, 384, "vits qkv"), (3349, 1536, 384, "vits fc1"),
              (4096, 4096, 4096, "4096^3"), (8192, 8192, 8192, "8192^3"), (40964, 256, 2304, "conv 1/4 256->256"),
              (13376, 512, 4608, "conv 1/7 512->512"), (3344, 1024, 9216, "conv 1/14 1024->1024"),
              (40964, 512, 2880, "gru08 zr 320->512 @1/4"), (40964, 256, 2880, "gru08 q 320->256 @1/4"),
              (13376, 512, 3456, "conv 1/7 384->512"), (40964, 256, 1152, "conv 1/4 128->256")]
    for M, N, K, name in shapes:
        A = (torch.randn(M, K, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        t = timeit(lambda: lib.call("gsr_dn_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), None, 0, None, None, 0,
                                    None, 0, out.data_ptr(), N, None, 0, 0, st))
        tf = 2.0 * M * N * K / t / 1e12
        print(json.dumps({"op": "gemm", "shape": name, "M": M, "N": N, "K": K, "us": t * 1e6, "tflops": tf,
                          "frac_of_peak": tf / PEAK}), flush=True)
    for n_tok, heads in ((3349, 16), (3349, 6)):
        D = heads * 64
        qkv = (torch.randn(n_tok, 3 * D, device="cuda")).half()
        n_pad = (n_tok + 63) // 64 * 64
        vt = torch.empty(heads * 64 * n_pad, dtype=torch.float16, device="cuda")
        out = torch.empty(n_tok, D, dtype=torch.float16, device="cuda")
        t = timeit(lambda: lib.call("gsr_dn_attention", n_tok, n_pad, heads, qkv.data_ptr(), 3 * D, vt.data_ptr(),
                                    0.125, out.data_ptr(), D, st))
        tf = 4.0 * n_tok * n_tok * D / t / 1e12
        print(json.dumps({"op": "attention", "n_tok": n_tok, "heads": heads, "us": t * 1e6, "tflops": tf,
                          "frac_of_peak": tf / PEAK}), flush=True)


if __name__ == "__main__":
    main()
