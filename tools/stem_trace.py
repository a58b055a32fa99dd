#!/usr/bin/env python
"""s_memtime phase trace of the third tile of workgroup 0 of the fused stem + max-pool kernel.
Build here:  python tools/stem_trace.py build      Run on the GPU box:  python tools/stem_trace.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
LIB = os.path.join(OUT, "libstem_trace.so")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                           "-DCOBEVT_STEM_TRACE", os.path.join(ROOT, "cobevt_amd", "csrc", "stem7x7.hip"), "-o", LIB])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
lib = ctypes.CDLL(LIB)
vp = ctypes.c_void_p
n, h, w = 20, 512, 512
plan = ops.ConvPlan(torch.randn(64, 3, 7, 7) * 0.1, torch.zeros(64), stride=2, pad=3, act=1, dtype=torch.bfloat16, device=dev, smallc=True)
x = torch.randn(n, h, w, 3, device=dev)
out = torch.empty(n, h // 4, w // 4, 64, device=dev, dtype=torch.bfloat16)
dims = (ctypes.c_int * 4)(0, n, h, w)
call = lambda: lib.cobevt_stem_conv7x7s2_pool(vp(x.data_ptr()), vp(plan.wgt_stem.data_ptr()), vp(plan.bias.data_ptr()), vp(out.data_ptr()),
                                              dims, vp(torch.cuda.current_stream().cuda_stream))
for _ in range(3):
    assert call() == 0
torch.cuda.synchronize()
t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
t0.record()
for _ in range(20):
    call()
t1.record()
torch.cuda.synchronize()
tr = (ctypes.c_ulonglong * 16)()
lib.cobevt_stem_read_trace(tr)
t = list(tr)
names = ["patch store + barrier", "issue next patch loads", "MFMA", "staging", "barrier", "pool + store", "barrier"]
print("%.1f us/launch; tile 3 of workgroup 0 (cycles): %s; total %d" % (
    t0.elapsed_time(t1) / 20 * 1e3, ", ".join("%s %d" % (nm, t[i + 1] - t[i]) for i, nm in enumerate(names)), t[7] - t[0]))
