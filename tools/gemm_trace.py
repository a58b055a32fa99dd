#!/usr/bin/env python
"""s_memtime phase trace of workgroup 0 / thread 0 of the dense-row GEMM (cobevt_linear_rows).
Build here:  python tools/gemm_trace.py build      Run on the GPU box:  python tools/gemm_trace.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
LIB = os.path.join(OUT, "libgemm_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DCOBEVT_GEMM_TRACE", os.path.join(ROOT, "cobevt_amd", "csrc", "gemm_rows.hip"), "-o", LIB])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
for (m, k, n, ln) in [(327680, 128, 128, 1), (81920, 128, 128, 1), (81920, 64, 128, 0), (5120, 128, 128, 1)]:
    class LN(object):
        weight, bias, eps = torch.ones(k), torch.zeros(k), 1e-5
    plan = ops.ConvPlan(torch.randn(n, k) / k ** 0.5, torch.zeros(n), dtype=torch.bfloat16, device=dev, ln=LN if ln else None)
    x = torch.randn(m, k, device=dev).to(torch.bfloat16)
    out = torch.empty(m, n, device=dev, dtype=torch.bfloat16)
    dims = (ctypes.c_long * 16)(0, m, n, k, plan.kp_rows, k, 0, 0, 1, m, 1, m, ln, 1, 1, 1)
    for _ in range(3):
        rc = lib.cobevt_linear_rows(vp(x.data_ptr()), vp(plan.wgt_rows.data_ptr()), vp(plan.bias.data_ptr()), None, None, None,
                                    None, None, vp(out.data_ptr()), dims, ctypes.c_float(1e-5),
                                    vp(torch.cuda.current_stream().cuda_stream))
        assert rc == 0
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(20):
        lib.cobevt_linear_rows(vp(x.data_ptr()), vp(plan.wgt_rows.data_ptr()), vp(plan.bias.data_ptr()), None, None, None,
                               None, None, vp(out.data_ptr()), dims, ctypes.c_float(1e-5), vp(torch.cuda.current_stream().cuda_stream))
    t1.record()
    torch.cuda.synchronize()
    tr = (ctypes.c_ulonglong * 16)()
    lib.cobevt_gemm_read_trace(tr)
    t = list(tr)
    print("M=%d K=%d N=%d ln=%d: %.1f us/launch; workgroup 0 cycles: loads+transform %d, LDS store+barrier %d, MFMA %d, barrier %d, "
          "staging+barrier %d, store pass %d, total %d" % (m, k, n, ln, t0.elapsed_time(t1) / 20 * 1e3, t[1] - t[0], t[2] - t[1],
                                                            t[3] - t[2], t[4] - t[3], t[5] - t[4], t[6] - t[5], t[6] - t[0]), flush=True)
