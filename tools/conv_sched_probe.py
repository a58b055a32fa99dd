#!/usr/bin/env python
"""Same-job A/B of the strip kernel's load scheduling (csrc/conv3x3.hip: COBEVT_CONV3_PLOAD_TAP = the tap at which the next chunk's patch /
the residual loads are issued, COBEVT_CONV3_PSTORE_SPREAD = LDS stores of that patch spread over taps 6-8): stand-alone builds of
conv3x3.hip per setting, timed INSIDE replayed graphs (20 back-to-back launches), outputs compared bit for bit with the first build.
    python tools/conv_sched_probe.py build        # here (hipcc cross-compiles)
    python tools/conv_sched_probe.py              # on the GPU box"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
def _b(plt, spread, first=6):
    return ("plt%d_%s" % (plt, ("s%d" % first) if spread else "burst"),
            ["-DCOBEVT_CONV3_PLOAD_TAP=%d" % plt, "-DCOBEVT_CONV3_PSTORE_SPREAD=%d" % spread, "-DCOBEVT_CONV3_PSTORE_FIRST=%d" % first])


BUILDS = [_b(0, 0), _b(4, 1, 6), _b(0, 1, 6), _b(0, 1, 3), _b(0, 1, 4), _b(1, 1, 5), _b(2, 1, 5), _b(3, 1, 6), _b(4, 1, 7), _b(5, 1, 7)]

if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    procs = []
    for name, flags in BUILDS:
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-Wno-pass-failed"] + flags + [
            os.path.join(ROOT, "cobevt_amd", "csrc", "conv3x3.hip"), "-o", os.path.join(OUT, "libconv3_%s.so" % name)]
        procs.append(subprocess.Popen(cmd))
    sys.exit(max(p.wait() for p in procs))

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402

dev = torch.device("cuda:0")
dtype = torch.bfloat16
vp = ctypes.c_void_p
SHAPES = [(20, 32, 32, 256, 256, 150), (20, 16, 16, 512, 512, 0), (20, 64, 64, 128, 128, 150), (20, 128, 128, 64, 64, 151), (5, 64, 64, 128, 64, 0)]
REP = 20


def graph_time(fn):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(REP):
                fn()
        for _ in range(3):
            g.replay()
        torch.cuda.synchronize()
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        t0.record()
        for _ in range(10):
            g.replay()
        t1.record()
        torch.cuda.synchronize()
    return t0.elapsed_time(t1) / (10 * REP) * 1e3


libs = [(name, ctypes.CDLL(os.path.join(OUT, "libconv3_%s.so" % name))) for name, _ in BUILDS]
for rnd in range(2):                                  # two rounds: the order effect (clock, cache state) shows as the spread between them
    for (n, h, w, cin, cout, variant) in SHAPES:
        wt = torch.randn(cout, cin, 3, 3, generator=torch.Generator().manual_seed(1)) / (3.0 * cin ** 0.5)
        plan = ops.ConvPlan(wt, torch.randn(cout) * 0.1, stride=1, pad=1, act=1, dtype=dtype, device=dev)
        v = variant or ops.conv3_tiling(n, h, w, cin, cout, plan.cc3)
        x = torch.randn(n, h, w, cin, device=dev).to(dtype)
        res = torch.randn(n, h, w, cout, device=dev).to(dtype)
        outs = []
        line = "%2dx%3dx%3d %3d->%3d v%d |" % (n, h, w, cin, cout, v)
        for name, lib in libs:
            out = torch.zeros(n, h, w, cout, device=dev, dtype=dtype)
            dims = (ctypes.c_int * 13)(0, n, h, w, cin, plan.cout, 0, 1, 0, plan.cc3, plan.coutp3, v, 1)

            def call():
                rc = lib.cobevt_conv3x3_wfrag_nhwc(vp(x.data_ptr()), vp(plan.wfrag.data_ptr()), vp(plan.bias.data_ptr()), vp(res.data_ptr()),
                                                   vp(out.data_ptr()), dims, vp(torch.cuda.current_stream().cuda_stream))
                assert rc == 0, rc
            us = graph_time(call)
            outs.append(out.clone())
            same = torch.equal(outs[0], outs[-1])
            line += " %s %.2f%s" % (name, us, "" if same else " MISMATCH")
        print(line, flush=True)
