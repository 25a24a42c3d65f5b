#!/usr/bin/env python3
"""gsr_dn_gemm at a few shapes with the core chosen by GSR_DN_GEMM_CORE (1: 128-row tiles, 2: 256x256
ping-pong, 3: 256x256 with 128x128 wave tiles, 0: the dispatcher's choice). One line per shape."""
import importlib
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
st = torch.cuda.current_stream().cuda_stream
shapes = [(4096, 4096, 4096), (8192, 8192, 8192), (3349, 3072, 1024), (3349, 4096, 1024), (3349, 1024, 4096), (3349, 1024, 1024)]
if len(sys.argv) > 1:
    shapes = shapes[:int(sys.argv[1])]
out_l = []
for M, N, K in shapes:
    A = (torch.randn(M, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    f = lambda: lib.call("gsr_dn_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), None, 0, None, None, 0,
                         None, 0, out.data_ptr(), N, None, 0, 0, st)
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    t = e0.elapsed_time(e1) / 20 * 1e-3
    g = lambda: torch.nn.functional.linear(A, W)          # the library (hipBLASLt through torch), same harness
    g(); g()
    torch.cuda.synchronize()
    e0.record()
    for _ in range(20):
        g()
    e1.record()
    torch.cuda.synchronize()
    tl = e0.elapsed_time(e1) / 20 * 1e-3
    out_l.append("%dx%dx%d %.1f us %.0f TF (library %.1f us %.0f TF, ratio %.2f)" %
                 (M, N, K, t * 1e6, 2.0 * M * N * K / t / 1e12, tl * 1e6, 2.0 * M * N * K / tl / 1e12, tl / t))
print(os.environ.get("GSRAST_LIB", "head").split("_")[-1], "core", os.environ.get("GSR_DN_GEMM_CORE", "0"), "\n  " + "\n  ".join(out_l))
