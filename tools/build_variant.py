#!/usr/bin/env python
"""Build a VARIANT of libcobevt_hip.so for same-job A/B runs: the package's sources with extra -D flags, objects and library under
tools/_probe/libs/<name>/ (git-ignored, shipped to the GPU box by gpurun).  Select it with COBEVT_HIP_LIB=<path>.

    python tools/build_variant.py gelu_exact -DCOBEVT_GELU_EXACT=1
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobevt_amd import build as b  # noqa: E402

name, flags = sys.argv[1], sys.argv[2:]
out = os.path.join(ROOT, "tools", "_probe", "libs", name)
os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libcobevt_hip.so")
procs, objs = [], []
for src in b.SOURCES:
    o = os.path.join(out, src.replace(".hip", ".o"))
    objs.append(o)
    procs.append((src, subprocess.Popen([b._hipcc()] + b.FLAGS + flags + ["-c", os.path.join(b.CSRC, src), "-o", o],
                                        stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
for src, p in procs:
    o_, _ = p.communicate()
    if p.returncode:
        raise SystemExit("hipcc failed for %s:\n%s" % (src, o_.decode(errors="replace")))
subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
for o in objs:
    os.remove(o)
print(lib)
