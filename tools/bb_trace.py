#!/usr/bin/env python
"""s_memtime phase trace of the fused BasicBlock kernel (block 0 / thread 0; -DCOBEVT_BB_TRACE copy of basicblock.hip, never the
product .so).  Build here:  python tools/bb_trace.py build      Run on the GPU box:  python tools/bb_trace.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "tools", "_probe", "libbb_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DCOBEVT_BB_TRACE",
                           os.path.join(ROOT, "cobevt_amd", "csrc", "basicblock.hip"), "-o", LIB])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
for (n, h, w, c) in [(20, 128, 128, 64), (20, 64, 64, 128)]:
    mk = lambda: torch.randn(c, c, 3, 3) / (3.0 * c ** 0.5)
    p1 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    p2 = ops.ConvPlan(mk(), torch.randn(c) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, c, device=dev).to(dtype)
    out = torch.empty_like(x)
    dims = (ctypes.c_int * 6)(0, n, h, w, c, 0)

    def call():
        rc = lib.cobevt_basicblock_nhwc(vp(x.data_ptr()), vp(p1.wfrag.data_ptr()), vp(p1.bias.data_ptr()), vp(p2.wfrag.data_ptr()),
                                        vp(p2.bias.data_ptr()), vp(out.data_ptr()), dims, vp(torch.cuda.current_stream().cuda_stream))
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
    tr = (ctypes.c_ulonglong * 32)()
    lib.cobevt_bb_read_trace(tr)
    t = list(tr)
    nch = c // 64
    flops = 2 * 2.0 * n * h * w * c * c * 9
    conv1 = [(t[2 + 2 * k] - t[1 + 2 * k]) for k in range(nch)]
    gaps1 = [(t[1 + 2 * k] - (t[2 * k] if k else t[0])) for k in range(nch)]
    print("%dx%dx%d C=%d: %.1f us per launch, %.0f TF/s; block-0 cycles: conv1 fill+barriers before each chunk %s, conv1 chunks %s, "
          "intermediate + residual issue %d, barrier %d, conv2 %d, barrier %d, stage write %d, store pass %d; total %d"
          % (n, h, w, c, us, flops / us / 1e6, gaps1, conv1, t[8] - t[2 * nch], t[9] - t[8], t[10] - t[9], t[11] - t[10], t[12] - t[11],
             t[13] - t[12], t[13] - t[0]), flush=True)
