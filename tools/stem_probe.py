#!/usr/bin/env python
"""Fused stem + max-pool kernel: per-launch time inside a replayed HIP graph (20 launches per graph) for builds of
csrc/stem7x7.hip with different compile-time tile geometries, all in one job (box-to-box variance is larger than the effects).
Build here:   python tools/stem_probe.py build "ROWS=3" "ROWS=4" ...     (each spec = space separated -D macros, COBEVT_STEM_ prefix)
On the GPU:   python tools/stem_probe.py            times every tools/_probe/libstem_*.so + checks they agree bit for bit"""
import ctypes
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "tools", "_probe")
SRC = os.path.join(ROOT, "cobevt_amd", "csrc", "stem7x7.hip")
if len(sys.argv) > 1 and sys.argv[1] == "build":
    os.makedirs(OUT, exist_ok=True)
    for spec in sys.argv[2:]:
        src = SRC
        macros = []
        for kv in spec.split():
            if kv.startswith("SRC="):
                src = kv[4:]
            else:
                macros.append("-DCOBEVT_STEM_" + kv)
        name = spec.replace(" ", "_").replace("=", "").replace("/", "_")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared",
                               "-I" + os.path.join(ROOT, "cobevt_amd", "csrc"), "-I" + os.path.join(ROOT, "include")]
                              + macros + [src, "-o", os.path.join(OUT, "libstem_%s.so" % name)])
    sys.exit(0)

import torch  # noqa: E402

sys.path.insert(0, ROOT)
from cobevt_amd import ops  # noqa: E402
from conv_graph_probe import graph_time  # noqa: E402

dev = torch.device("cuda:0")
vp = ctypes.c_void_p
torch.manual_seed(0)
n, h, w = 20, 512, 512
x = torch.randn(n, h, w, 3, device=dev)
wt, bs = torch.randn(64, 3, 7, 7) * 0.1, torch.randn(64) * 0.1
libs = sorted(glob.glob(os.path.join(OUT, "libstem_*.so")))
for dt, code in ((torch.bfloat16, 0), (torch.float32, 1)):
    plan = ops.ConvPlan(wt, bs, stride=2, pad=3, act=1, dtype=dt, device=dev, smallc=True)
    ref = None
    for path in libs:
        lib = ctypes.CDLL(path)
        out = torch.zeros(n, h // 4, w // 4, 64, device=dev, dtype=dt)
        dims = (ctypes.c_int * 4)(code, n, h, w)

        def call():
            rc = lib.cobevt_stem_conv7x7s2_pool(vp(x.data_ptr()), vp(plan.wgt_stem.data_ptr()), vp(plan.bias.data_ptr()), vp(out.data_ptr()),
                                                dims, vp(torch.cuda.current_stream().cuda_stream))
            assert rc == 0, rc
        us = graph_time(call)
        torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        same = torch.equal(ref, out)
        print("%-8s %-40s %.1f us  %s" % (str(dt).split(".")[-1], os.path.basename(path), us, "same" if same else
                                          "DIFFERS max %.3g" % (ref.float() - out.float()).abs().max().item()), flush=True)
        if hasattr(lib, "cobevt_stem_read_trace"):
            tr = (ctypes.c_ulonglong * 16)()
            lib.cobevt_stem_read_trace(tr)
            t = list(tr)
            names = ["patch store + barrier", "issue next patch loads", "MFMA", "staging", "barrier", "pool + store", "barrier"]
            print("    tile 3 of workgroup 0 (cycles): %s; total %d" % (
                ", ".join("%s %d" % (nm, t[i + 1] - t[i]) for i, nm in enumerate(names)), t[7] - t[0]), flush=True)
    # odd sizes: partial tiles in both directions
    for (n2, h2, w2) in ((1, 36, 44), (2, 100, 72)):
        x2 = torch.randn(n2, h2, w2, 3, device=dev)
        outs = []
        for path in libs:
            lib = ctypes.CDLL(path)
            o2 = torch.zeros(n2, h2 // 4, w2 // 4, 64, device=dev, dtype=dt)
            d2 = (ctypes.c_int * 4)(code, n2, h2, w2)
            assert lib.cobevt_stem_conv7x7s2_pool(vp(x2.data_ptr()), vp(plan.wgt_stem.data_ptr()), vp(plan.bias.data_ptr()), vp(o2.data_ptr()),
                                                  d2, vp(torch.cuda.current_stream().cuda_stream)) == 0
            torch.cuda.synchronize()
            outs.append(o2)
        print("  %dx%dx%d agree: %s" % (n2, h2, w2, all(torch.equal(outs[0], o) for o in outs[1:])), flush=True)
