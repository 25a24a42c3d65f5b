#!/usr/bin/env python3
"""Decoder time with a boolean attribute of Metric3DNet on / off, same process, interleaved, eager launches:
    python tools/ab_decoder.py concat_free_gru [vitl]"""
import importlib
import sys
import time
from pathlib import Path

ROOT = Path(__file__).resolve().parents[1]
sys.path.insert(0, str(ROOT))
import torch  # noqa: E402

from tests import test_gpu_depthnet as T  # noqa: E402
from tests.golden import dn_weights as DW  # noqa: E402

N = importlib.import_module("3dgs_monocular_depth_init_amd.depth_prediction.predictors.metric3d_net")
attr = sys.argv[1]
for bb in (sys.argv[2:] or ["vitl"]):
    net = N.Metric3DNet(T._state(N.CONFIGS[bb]), backbone=bb, device="cuda", use_graph=False)
    tok = net.encode(DW.image(616, 1064).cuda())
    res = {True: [], False: []}
    for rep in range(4):
        for on in (True, False):
            setattr(net, attr, on)
            net.decode(tok)
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(3):
                net.decode(tok)
            torch.cuda.synchronize()
            res[on].append((time.perf_counter() - t0) / 3 * 1e3)
    print(bb, "decoder ms:", attr, "on", ["%.2f" % x for x in res[True]], "off", ["%.2f" % x for x in res[False]])
