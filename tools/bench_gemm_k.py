#!/usr/bin/env python3
"""gsr_dn_gemm time against K at fixed M x N: the slope is the K-tile time, the intercept the
prologue + epilogue + launch. GSR_DN_GEMM_CORE picks the core."""
import importlib
import os
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
lib = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
st = torch.cuda.current_stream().cuda_stream
for M, N in ((3349, 3072), (3349, 4096), (4096, 4096)):
    row = []
    for K in (128, 512, 1024, 2048, 4096):
        A = (torch.randn(M, K, device="cuda") * 0.5).half()
        W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
        out = torch.empty(M, N, dtype=torch.float16, device="cuda")
        f = lambda: lib.call("gsr_dn_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), None, 0, None, None, 0,
                             None, 0, out.data_ptr(), N, None, 0, 0, st)
        for _ in range(3):
            f()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            f()
        e1.record()
        torch.cuda.synchronize()
        row.append("K=%d %.1f us" % (K, e0.elapsed_time(e1) / 20 * 1e3))
    print("core", os.environ.get("GSR_DN_GEMM_CORE", "0"), "%dx%d" % (M, N), " | ".join(row))
