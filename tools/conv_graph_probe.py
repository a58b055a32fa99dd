#!/usr/bin/env python
"""Per-launch time of the 3x3 convolution tile variants INSIDE a replayed HIP graph (20 back-to-back launches per graph): the
eager probe (tools/conv_probe.py) cannot see below the ~12 us of a Python launch.  Run on the GPU box:
    SHAPES=5x64x64x128x128,... VARIANTS=0,-1,151,132 python tools/conv_graph_probe.py      (0 = automatic choice, -1 = LDS-staged kernel)"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
torch.manual_seed(0)
SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ.get(
    "SHAPES", "5x128x128x128x32,5x64x64x128x32,5x64x64x128x128,5x32x32x128x128,1x128x128x64x64,1x64x64x128x64,1x128x128x64x32").split(",")]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "0,-1,141,151,132,142,152,133,143,153").split(",")]
REP = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (10 * REP) * 1e3


if __name__ != "__main__":
    SHAPES = []
for (n, h, w, cin, cout) in SHAPES:
    wt = torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5)
    plan = ops.ConvPlan(wt, torch.randn(cout) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, cin, device=dev).to(dtype)
    out = torch.empty(n, h, w, cout, device=dev, dtype=dtype)
    line = "%2dx%3dx%3d %3d->%3d |" % (n, h, w, cin, cout)
    for v in VARIANTS:
        ops.USE_CONV3_WFRAG = v >= 0
        ops.CONV3_VARIANT = max(v, 0)
        try:
            us = graph_time(lambda: ops.conv2d(x, plan, out=out))
            name = {0: "auto(%d)" % ops.conv3_tiling(n, h, w, cin, cout, plan.cc3), -1: "lds"}.get(v, str(v))
            line += " %s %.1f" % (name, us)
        except Exception as e:   # noqa: BLE001
            line += " %d err" % v
    ops.USE_CONV3_WFRAG, ops.CONV3_VARIANT = True, 0
    print(line, flush=True)
if __name__ == "__main__" and os.environ.get("HEAD"):
    wt = torch.randn(2, 32, 3, 3) / 17.0
    plan = ops.ConvPlan(wt, torch.randn(2), stride=1, pad=1, store_mode=2, dtype=dtype, device=dev)
    x = torch.randn(1, 256, 256, 32, device=dev).to(dtype)
    out = torch.empty(1, 2, 256, 256, device=dev, dtype=torch.float32)
    for flag in (True, False):
        ops.USE_HEAD_CONV = flag
        print("head 32->2 256x256 direct=%s %.1f us" % (flag, graph_time(lambda: ops.conv2d(x, plan))), flush=True)
