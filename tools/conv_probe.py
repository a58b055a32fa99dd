#!/usr/bin/env python
"""Time every tile variant of the 3x3 convolution kernels on the four ResNet-34 layer shapes of the CoBEVT frame and
check them against each other (run on the GPU box: python tools/conv_probe.py [fp32])."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

dtype = torch.float32 if "fp32" in sys.argv else torch.bfloat16
dev = torch.device("cuda:0")
torch.manual_seed(0)
SHAPES = [(20, 128, 128, 64, 64), (20, 64, 64, 128, 128), (20, 32, 32, 256, 256), (20, 16, 16, 512, 512),
          (5, 128, 128, 128, 32), (1, 32, 32, 128, 128)]
if os.environ.get("SHAPES"):
    SHAPES = [tuple(int(v) for v in t.split("x")) for t in os.environ["SHAPES"].split(",")]
ITERS = 30


def bench(fn):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(ITERS):
        fn()
    t1.record()
    torch.cuda.synchronize()
    return t0.elapsed_time(t1) / ITERS * 1e3


for (n, h, w, cin, cout) in SHAPES:
    wt = torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5)
    bias = torch.randn(cout) * 0.1
    plan = ops.ConvPlan(wt, bias, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, cin, device=dev).to(dtype)
    res = torch.randn(n, h, w, cout, device=dev).to(dtype)
    flops = 2.0 * n * h * w * cout * cin * 9
    ops.USE_CONV3_WFRAG = False
    ref = ops.conv2d(x, plan, residual=res).float()
    us = bench(lambda: ops.conv2d(x, plan, residual=res))
    print("%3dx%3dx%3d %3d->%3d  lds-staged     %7.1f us %7.1f TF/s" % (n, h, w, cin, cout, us, flops / us / 1e6))
    ops.USE_CONV3_WFRAG = True
    for variant in [int(v) for v in os.environ.get('VARIANTS', '130,131,140,141,150,151,160,161').split(',')]:
        ops.CONV3_VARIANT = variant
        try:
            y = ops.conv2d(x, plan, residual=res).float()
        except Exception as e:   # noqa: BLE001
            print("   variant %d: %s" % (variant, e))
            continue
        err = (y - ref).abs().max().item()
        us = bench(lambda: ops.conv2d(x, plan, residual=res))
        print("%3dx%3dx%3d %3d->%3d  wfrag variant %3d %7.1f us %7.1f TF/s   max|diff| vs lds-staged %.3g" %
              (n, h, w, cin, cout, variant, us, flops / us / 1e6, err))
    ops.CONV3_VARIANT = 0
