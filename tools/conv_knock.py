#!/usr/bin/env python
"""Knock-out study of cobevt_conv3x3_wfrag_nhwc: the same kernel compiled with parts removed (-DCOBEVT_CONV3_KNOCK=mask:
1 no B-fragment loads after the first chunk, 2 MFMAs replaced by one VALU op, 4 no A-fragment LDS reads after the
first chunk, 8 no patch reload) plus an s_memtime trace of block 0 / wave 0 (-DCOBEVT_CONV3_TRACE).
Build here:  python tools/conv_knock.py build      Run on the GPU box:  python tools/conv_knock.py"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
MASKS = [int(m) for m in os.environ.get("MASKS", "0,15,13,2").split(",")]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for m in MASKS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
               "-DCOBEVT_CONV3_KNOCK=%d" % (m % 100), "-DCOBEVT_CONV3_TRACE"] + (["-DCOBEVT_CONV3_SETPRIO"] if m >= 100 else []) + [ os.path.join(ROOT, "cobevt_amd", "csrc", "conv3x3.hip"),
               "-o", os.path.join(OUT, "libconv3_k%d.so" % m)]
        subprocess.check_call(cmd)
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
SHAPES = [(20, 32, 32, 256, 256), (20, 16, 16, 512, 512)] if os.environ.get("SHAPES") == "both" else [(20, 32, 32, 256, 256)]
VARIANTS = [int(v) for v in os.environ.get("VARIANTS", "150").split(",")]
vp = ctypes.c_void_p


def call(lib, x, plan, res, out, variant):
    n, h, w, cin = x.shape
    dims = (ctypes.c_int * 13)(0, n, h, w, cin, plan.cout, 0, 1, 0, plan.cc3, plan.coutp3, variant, 1)
    rc = lib.cobevt_conv3x3_wfrag_nhwc(vp(x.data_ptr()), vp(plan.wfrag.data_ptr()), vp(plan.bias.data_ptr()),
                                       vp(res.data_ptr()), vp(out.data_ptr()), dims, vp(torch.cuda.current_stream().cuda_stream))
    assert rc == 0, rc


for (n, h, w, cin, cout) in SHAPES:
    wt = torch.randn(cout, cin, 3, 3) / (3.0 * cin ** 0.5)
    plan = ops.ConvPlan(wt, torch.randn(cout) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
    x = torch.randn(n, h, w, cin, device=dev).to(dtype)
    res = torch.randn(n, h, w, cout, device=dev).to(dtype)
    out = torch.empty_like(res)
    flops = 2.0 * n * h * w * cout * cin * 9
    for variant in VARIANTS:
        for m in MASKS:
            lib = ctypes.CDLL(os.path.join(OUT, "libconv3_k%d.so" % m))
            for _ in range(3):
                call(lib, x, plan, res, out, variant)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(20):
                call(lib, x, plan, res, out, variant)
            t1.record()
            torch.cuda.synchronize()
            us = t0.elapsed_time(t1) / 20 * 1e3
            line = "%dx%dx%d %d->%d v%d knock %2d: %7.1f us %7.1f TF/s-equiv" % (n, h, w, cin, cout, variant, m, us, flops / us / 1e6)
            if True:
                tr = (ctypes.c_ulonglong * 64)()
                lib.cobevt_conv3_read_trace(tr)
                t = list(tr)
                nst = min(cin // 64 * 9, 36)
                taps = [t[2 + i + 1] - t[2 + i] for i in range(nst - 1)]
                line += "\n    trace (cycles): prologue %d, taps %s, main loop total %d, k-split reduce %d (bias ready %d, round 0 done %d), store pass %d" % (
                    t[1] - t[0], taps, t[38] - t[1], t[39] - t[38], t[57] - t[38], t[58] - t[38], t[40] - t[39])
                if variant >= 100:
                    nch = min(cin // 64, 4)
                    line += "\n    per chunk: tap8 start -> before barrier %s, barrier wait %s, after barrier -> next tap0 %s" % (
                        [t[41 + 2 * c] - t[2 + c * 9 + 8] for c in range(nch)], [t[42 + 2 * c] - t[41 + 2 * c] for c in range(nch)],
                        [t[2 + (c + 1) * 9] - t[42 + 2 * c] for c in range(nch - 1)])
                    line += "\n    chunk-0 barrier arrival per wave (cycles after main loop start, simd): %s" % (
                        [((t[49 + w] & ((1 << 60) - 1)) - t[1], t[49 + w] >> 60) for w in range(8)])
            print(line, flush=True)
