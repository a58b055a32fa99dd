#!/usr/bin/env python
"""Per-launch time of the dense-row GEMM inside a replayed HIP graph (20 back-to-back launches per graph) on the frame's shapes.
Run on the GPU box: python tools/gemm_graph_probe.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402
from conv_graph_probe import graph_time  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16


class LN(object):
    def __init__(self, d):
        self.weight, self.bias, self.eps = torch.ones(d), torch.zeros(d), 1e-5


if __name__ == "__main__":
    shapes = os.environ.get("GEMM_SHAPES")    # "m,k,n,ln;..."
    for (m, k, n, ln, res) in ([tuple(int(v) for v in t.split(",")) + (False,) for t in shapes.split(";")] if shapes else [(5120, 128, 128, True, False), (5120, 128, 384, True, False), (5120, 128, 256, True, False),
                               (5120, 512, 128, False, False), (20480, 128, 128, True, False), (20480, 256, 128, False, False),
                               (81920, 128, 128, False, False), (81920, 128, 256, True, False), (327680, 128, 128, True, False),
                               (1024, 128, 128, False, False)]):
        plan = ops.ConvPlan(torch.randn(n, k) / k ** 0.5, torch.zeros(n), dtype=dtype, device=dev, ln=LN(k) if ln else None)
        x = torch.randn(m, k, device=dev).to(dtype)
        out = torch.empty(m, n, device=dev, dtype=dtype)
        us = graph_time(lambda: ops.linear(x, plan, out=out))
        byt = (m * k + m * n + n * k) * 2
        print("M=%6d K=%3d N=%3d ln=%d  %6.1f us  %6.0f GB/s" % (m, k, n, ln, us, byt / us / 1e3), flush=True)
