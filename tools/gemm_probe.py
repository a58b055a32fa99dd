#!/usr/bin/env python
"""Dense-row GEMM (Linear / 1x1 conv) v1 vs the persistent fragment-ordered v2 on the CoBEVT frame's shapes."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
torch.manual_seed(0)


class LN(object):
    def __init__(self, d):
        self.weight, self.bias, self.eps = torch.ones(d), torch.zeros(d), 1e-5


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


for (m, k, n, ln, res) in [(81920, 128, 128, True, False), (327680, 128, 128, True, False), (81920, 64, 128, False, False),
                           (81920, 128, 32, False, False), (81920, 32, 128, False, True), (20480, 128, 128, True, False),
                           (5120, 128, 384, True, False), (5120, 128, 128, True, False), (81920, 256, 128, False, False)]:
    plan = ops.ConvPlan(torch.randn(n, k) / k ** 0.5, torch.zeros(n), dtype=dtype, device=dev, ln=LN(k) if ln else None)
    x = torch.randn(m, k, device=dev).to(dtype)
    r = torch.randn(m, n, device=dev).to(dtype) if res else None
    byt = (m * k + m * n * (2 if res else 1) + n * k) * 2
    out = {}
    for v2 in (False, True):
        ops.USE_GEMM_ROWS2 = v2
        y = ops.linear(x, plan, residual=r)
        us = bench(lambda: ops.linear(x, plan, residual=r))
        out[v2] = (us, y)
    d = (out[True][1].float() - out[False][1].float()).abs().max().item()
    print("M=%6d K=%3d N=%3d ln=%d res=%d   v1 %6.1f us %5.0f GB/s   v2 %6.1f us %5.0f GB/s   max|diff| %.3g" %
          (m, k, n, ln, res, out[False][0], byt / out[False][0] / 1e3, out[True][0], byt / out[True][0] / 1e3, d), flush=True)
