#!/usr/bin/env python
"""Run the fused BasicBlock kernel a few times on one layer shape (target for rocprofv3 --pmc): bb_one.py N H W C [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

n, h, w, c = [int(v) for v in sys.argv[1:5]]
iters = int(sys.argv[5]) if len(sys.argv) > 5 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
mk = lambda: torch.randn(c, c, 3, 3) / (3.0 * c ** 0.5)
p1 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=torch.bfloat16, device=dev)
p2 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=torch.bfloat16, device=dev)
x = torch.randn(n, h, w, c, device=dev).to(torch.bfloat16)
for _ in range(iters):
    ops.basicblock(x, p1, p2)
torch.cuda.synchronize()
