#!/usr/bin/env python
"""3x3 / stride-2 convs of ResNet-34 (first conv of layer2/3/4): strip kernel tilings vs the generic implicit GEMM."""
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


for (n, h, w, cin, cout) in [(20, 128, 128, 64, 128), (20, 64, 64, 128, 256), (20, 32, 32, 256, 512)]:
    plan = ops.ConvPlan(torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5), torch.randn(cout) * 0.1, stride=2, pad=1, act=1,
                        dtype=dtype, device=dev)
    x = torch.randn(n, h, w, cin, device=dev).to(dtype)
    fl = 2.0 * n * (h // 2) * (w // 2) * cout * cin * 9
    ops.USE_CONV3_S2 = False
    ref = ops.conv2d(x, plan).float()
    us = bench(lambda: ops.conv2d(x, plan))
    print("%dx%dx%d %d->%d s2  igemm        %7.1f us %7.1f TF/s" % (n, h, w, cin, cout, us, fl / us / 1e6), flush=True)
    ops.USE_CONV3_S2 = True
    for variant in (0, 130, 131, 140, 141, 150, 151, 160, 161):
        ops.CONV3_VARIANT = variant
        y = ops.conv2d(x, plan).float()
        us = bench(lambda: ops.conv2d(x, plan))
        print("%dx%dx%d %d->%d s2  strips %3d   %7.1f us %7.1f TF/s  max|diff| %.3g" %
              (n, h, w, cin, cout, variant, us, fl / us / 1e6, (y - ref).abs().max().item()), flush=True)
    ops.CONV3_VARIANT = 0
