#!/usr/bin/env python
"""Per-launch time of the fused row chain (attention out-projection + skip + pre-norm MLP (+ post-norm, + next projection)) inside a
replayed HIP graph, on the frame's row counts; ROW_CHAIN_ROWS = 0 (32 rows per workgroup) vs 64.  Run on the GPU box."""
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
    torch.manual_seed(0)
    d, hid = 128, 256
    pp = ops.ConvPlan(torch.randn(d, d) / d ** 0.5, torch.zeros(d), dtype=dtype, device=dev)
    p1 = ops.ConvPlan(torch.randn(hid, d) / d ** 0.5, torch.zeros(hid), act=2, dtype=dtype, device=dev, ln=LN(d))
    p2 = ops.ConvPlan(torch.randn(d, hid) / hid ** 0.5, torch.zeros(d), dtype=dtype, device=dev)
    for nxt_n in (0, 128, 384):
        pn = ops.ConvPlan(torch.randn(nxt_n, d) / d ** 0.5, None, dtype=dtype, device=dev, ln=LN(d)) if nxt_n else None
        for m in (1024, 5120, 20480, 81920):
            a = torch.randn(m, d, device=dev).to(dtype)
            skip = torch.randn(m, d, device=dev).to(dtype)
            line = "M=%6d next=%3d |" % (m, nxt_n)
            for rows in (0, 64):
                ops.ROW_CHAIN_ROWS = rows
                us = graph_time(lambda: ops.attn_mlp_chain(a, skip, pp, p1, p2, next_plan=pn))
                line += "  rows%-2d %6.1f us" % (rows or 32, us)
            ops.ROW_CHAIN_ROWS = 0
            print(line, flush=True)
