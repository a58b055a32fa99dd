#!/usr/bin/env python
"""Where a strip-kernel launch spends its wall time: every workgroup records s_memrealtime (100 MHz) at entry and exit
(-DCOBEVT_CONV3_TRACE build of conv3x3.hip, never the product .so), compared with the HIP-event duration of the launch.
Build here:  python tools/conv_rt.py build      Run on the GPU box:  python tools/conv_rt.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
LIB = os.path.join(OUT, "libconv3_rt.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCOBEVT_CONV3_TRACE",
                           os.path.join(ROOT, "cobevt_amd", "csrc", "conv3x3.hip"), "-o", LIB])
    sys.exit(0)

import numpy as np  # noqa: E402
import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
SHAPES = [(20, 32, 32, 256, 256, 150), (20, 16, 16, 512, 512, 151), (20, 64, 64, 128, 128, 150)]
for (n, h, w, cin, cout, variant) in SHAPES:
    wt = torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5)
    plan = ops.ConvPlan(wt, torch.randn(cout) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, cin, device=dev).to(dtype)
    res = torch.randn(n, h, w, cout, device=dev).to(dtype)
    out = torch.empty_like(res)
    dims = (ctypes.c_int * 13)(0, n, h, w, cin, cout, 0, 1, 0, plan.cc3, plan.coutp3, variant, 1)

    def call():
        rc = lib.cobevt_conv3x3_wfrag_nhwc(vp(x.data_ptr()), vp(plan.wfrag.data_ptr()), vp(plan.bias.data_ptr()), vp(res.data_ptr()),
                                           vp(out.data_ptr()), dims, vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0, rc
    for _ in range(5):
        call()
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        call()
    t1.record()
    torch.cuda.synchronize()
    us = t0.elapsed_time(t1) / 20 * 1e3
    rt = (ctypes.c_ulonglong * (3 * 2048))()
    lib.cobevt_conv3_read_rt(rt)
    a = np.array(list(rt), dtype=np.int64).reshape(2048, 3)
    a = a[a[:, 0] > 0]
    start, end, xcc = a[:, 0] / 100.0, a[:, 1] / 100.0, a[:, 2]
    t00 = start.min()
    dur = end - start
    tr = (ctypes.c_ulonglong * 64)()
    lib.cobevt_conv3_read_trace(tr)
    t = list(tr)
    print("%dx%dx%d %d->%d v%d: %.1f us per launch (events, back to back); %d workgroups; inside the kernel: first entry -> last exit %.1f us; "
          "entry skew %.2f us (p50 %.2f), exit skew %.2f us; workgroup lifetime min/p50/max %.1f/%.1f/%.1f us; wave-0 trace %d shader cycles "
          "= %.2f GHz over its lifetime" % (n, h, w, cin, cout, variant, us, len(a), end.max() - t00, start.max() - t00,
                                             float(np.median(start - t00)), end.max() - end.min(), dur.min(), float(np.median(dur)), dur.max(),
                                             t[40] - t[0], (t[40] - t[0]) / max(float(dur[0]), 1e-9) / 1e3), flush=True)
    print("    workgroups per XCC:", np.bincount(xcc.astype(np.int64), minlength=8).tolist())
