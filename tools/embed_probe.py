#!/usr/bin/env python
"""bev_embed + to_q GEMM: fused (embedding produced inside the GEMM) vs two launches, level-0 shape of the CoBEVT frame."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
torch.manual_seed(0)
b, n, hw, d = 5, 4, 128 * 128, 128
E = torch.randn(b * n, 4, 4, device=dev)
world = torch.randn(2, hw, device=dev) * 30
w_bev, b_bev, w_cam = torch.randn(d, 2, device=dev), torch.randn(d, device=dev), torch.randn(d, 4, device=dev)
x = torch.randn(b, hw, d, device=dev).to(dtype)


class LN(object):
    weight, bias, eps = torch.ones(d), torch.zeros(d), 1e-5


plan = ops.ConvPlan(torch.randn(128, d) / d ** 0.5, torch.zeros(128), dtype=dtype, device=dev, ln=LN)


def bench(fn, iters=20):
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


f = lambda: ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, x, n, plan)
print("fused        %7.1f us" % bench(f))
ops.USE_EMBED_GEMM = False
print("two launches %7.1f us" % bench(f))
