#!/usr/bin/env python
"""Fused BasicBlock launch vs the two 3x3 launches on the ResNet-34 layer1 / layer2 shapes of the CoBEVT frame."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
torch.manual_seed(0)


def bench(fn, iters=30):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(iters):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / iters * 1e3


for (n, h, w, c) in [(20, 128, 128, 64), (20, 64, 64, 128), (5, 128, 128, 64)]:
    mk = lambda: torch.randn(c, c, 3, 3) / (3.0 * c ** 0.5)
    p1 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    p2 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, c, device=dev).to(dtype)
    fl = 2 * 2.0 * n * h * w * c * 9 * c
    a = bench(lambda: ops.conv2d(ops.conv2d(x, p1), p2, residual=x))
    b = bench(lambda: ops.basicblock(x, p1, p2))
    print("%dx%dx%dx%d  two launches %7.1f us (%6.1f TF/s)   fused %7.1f us (%6.1f TF/s algorithmic)" %
          (n, h, w, c, a, fl / a / 1e6, b, fl / b / 1e6), flush=True)
    if c == 64:
        for th in (8, 16):
            ops.BASICBLOCK_TILE_ROWS = th
            b = bench(lambda: ops.basicblock(x, p1, p2))
            print("      tile rows %2d: fused %7.1f us" % (th, b), flush=True)
        ops.BASICBLOCK_TILE_ROWS = 0
