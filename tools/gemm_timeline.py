#!/usr/bin/env python3
"""Where a gemm8p launch spends its time (diagnostic build -DGSR_GEMM_TIMELINE=1, GSR_DN_GEMM_CORE=4):
per-workgroup s_memrealtime stamps at kernel start / tile 0 landed / main loop done / epilogue done."""
import ctypes as C
import importlib
import sys
from pathlib import Path

import torch

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
L = importlib.import_module("3dgs_monocular_depth_init_amd._lib")
lib = L.load()
fn = lib.gsr_debug_set_gemm_timeline
fn.argtypes = [C.c_void_p]
fn.restype = C.c_int
st = torch.cuda.current_stream().cuda_stream
for M, N, K in ((3349, 3072, 1024), (3349, 3072, 128), (4096, 4096, 4096)):
    nwg = ((M + 255) // 256) * ((N + 255) // 256)
    buf = torch.zeros(nwg, 4, dtype=torch.int64, device="cuda")
    assert fn(buf.data_ptr()) == 0
    A = (torch.randn(M, K, device="cuda") * 0.5).half()
    W = (torch.randn(N, K, device="cuda") / K ** 0.5).half()
    out = torch.empty(M, N, dtype=torch.float16, device="cuda")
    for _ in range(5):
        L.call("gsr_dn_gemm", M, N, K, A.data_ptr(), K, W.data_ptr(), None, 0, None, None, 0, None, 0,
               out.data_ptr(), N, None, 0, 0, st)
    torch.cuda.synchronize()
    b = buf.double().cpu() * 0.01          # us
    t0 = b[:, 0].min()
    print("%dx%dx%d (%d workgroups): first start -> last end %.1f us; start spread %.1f; per workgroup mean: "
          "prologue %.1f, main loop %.1f, epilogue %.1f; whole %.1f (max %.1f)" %
          (M, N, K, nwg, b[:, 3].max() - t0, b[:, 0].max() - t0, (b[:, 1] - b[:, 0]).mean(), (b[:, 2] - b[:, 1]).mean(),
           (b[:, 3] - b[:, 2]).mean(), (b[:, 3] - b[:, 0]).mean(), (b[:, 3] - b[:, 0]).max()))
