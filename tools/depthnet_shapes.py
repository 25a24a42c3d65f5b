#!/usr/bin/env python3
"""List the GEMM / implicit-convolution launches of one Metric3D inference by shape (count, FLOP share),
timing each distinct shape alone (20 launches, events): where the network's GEMM time goes."""
import collections
import os
import importlib
import json
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from tests import test_gpu_depthnet as T  # noqa: E402
from tests.golden import dn_weights as DW  # noqa: E402

N = importlib.import_module("3dgs_monocular_depth_init_amd.depth_prediction.predictors.metric3d_net")
bb = sys.argv[1] if len(sys.argv) > 1 else "vitl"
net = N.Metric3DNet(T._state(N.CONFIGS[bb]), backbone=bb, device="cuda")
img = DW.image(616, 1064).cuda()
seen = collections.OrderedDict()
orig = N.call


def spy(name, *a):
    if name in ("gsr_dn_gemm", "gsr_dn_conv_gemm"):
        key = (name,) + tuple(x for x in a if isinstance(x, int) and not isinstance(x, bool) and abs(x) < (1 << 24))
        e = seen.setdefault(key, {"n": 0, "args": (name,) + a})
        e["n"] += 1
    return orig(name, *a)


N.call = spy
net.graphs_enabled = False if hasattr(net, "graphs_enabled") else None
net.encode(img)
tok = net.encode(img)
seen.clear()
tok = net.encode(img)
net.decode(tok)
N.call = orig
rows = []
for key, e in seen.items():
    a = e["args"]
    f = lambda: orig(*a)
    f(); f()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        f()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 20 * 1e3
    if a[0] == "gsr_dn_gemm":
        M, Nn, K = a[1], a[2], a[3]
        desc = "gemm %dx%dx%d act%d" % (M, Nn, K, a[8])
    else:
        H, W, C, KS, Nn, K = a[1], a[2], a[3], a[6], a[7], a[8]
        M = H * W
        desc = "conv%d %dx%d C%d->%d (M=%d K=%d) act%d" % (KS, H, W, C, Nn, M, K, a[11])
    rows.append((us * e["n"], e["n"], us, 2.0 * M * Nn * K / us / 1e6, desc))
rows.sort(reverse=True)
tot = sum(r[0] for r in rows)
print("%s: %d distinct GEMM shapes, %.2f ms of GEMM launches per image (each shape timed alone)" % (bb, len(rows), tot / 1e3))
if os.environ.get("SHAPES_JSON"):
    print(json.dumps({desc: [n, round(us, 2)] for t, n, us, tf, desc in rows}))
else:
    for t, n, us, tf, desc in rows[:40]:
        print("  %6.0f us total  %3d x %6.1f us  %6.0f TFLOP/s  %s" % (t, n, us, tf, desc))
