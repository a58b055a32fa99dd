#!/usr/bin/env python
"""Run one 3x3 convolution shape / tile variant a few times (target for rocprofv3 --pmc):
python tools/conv_one.py N H W CIN COUT VARIANT [iters]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

n, h, w, cin, cout, variant = [int(v) for v in sys.argv[1:7]]
iters = int(sys.argv[7]) if len(sys.argv) > 7 else 5
dev = torch.device("cuda:0")
torch.manual_seed(0)
plan = ops.ConvPlan(torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5), torch.randn(cout) * 0.1, stride=1, pad=1, act=1,
                    dtype=torch.bfloat16, device=dev)
x = torch.randn(n, h, w, cin, device=dev).to(torch.bfloat16)
res = torch.randn(n, h, w, cout, device=dev).to(torch.bfloat16)
if variant < 0:
    ops.USE_CONV3_WFRAG = False
else:
    ops.CONV3_VARIANT = variant
for _ in range(iters):
    ops.conv2d(x, plan, residual=res)
torch.cuda.synchronize()
