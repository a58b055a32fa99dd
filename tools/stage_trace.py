#!/usr/bin/env python
"""Where a workgroup of the fused swap-fusion stage kernel (csrc/swap_stage.hip) spends its time: a probe build with s_memtime marks
(tools/_probe/libcobevt_hip_stagetrace.so, `--build` on the CPU box), run on the camera config's shape (5 agents, 32 x 32 map, 8 x 8
windows) and printed as per-phase medians in units of the 100 MHz s_memtime counter (x 10 ns).
Usage: python tools/stage_trace.py --build   (here)   /   python tools/stage_trace.py   (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cobevt_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "_probe", "libcobevt_hip_stagetrace.so")

if "--build" in sys.argv:
    sys.path.insert(0, ROOT)
    from cobevt_amd import build as b
    b.build(verbose=False)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "_probe", "swap_stage_trace.o")
    subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DCOBEVT_STAGE_TRACE", "-c", os.path.join(CSRC, "swap_stage.hip"), "-o", obj])
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in b.SOURCES if s != "swap_stage.hip"] + [obj]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB)
    sys.exit(0)

os.environ["COBEVT_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cobevt_amd import host, lib as L, synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda")
lib = L.load()
lib.cobevt_stage_trace_read.restype = ctypes.c_int
lib.cobevt_stage_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
args = dict(input_dim=128, mlp_dim=256, agent_size=5, window_size=8, dim_head=32, drop_out=0.1, depth=3, mask=True)
enc = synth.fill_module_(host.SwapFusionEncoder(args), 0).eval().to(dev)
x = synth.procedural_input("trace.x", (1, 5, 128, 32, 32), 0).to(dev)
mask = torch.ones(1, 32, 32, 1, 5, device=dev)
with host.compute_dtype(torch.bfloat16):
    for _ in range(3):
        enc(x, mask)
    torch.cuda.synchronize()
n = 160
buf = (ctypes.c_ulonglong * (16 * n))()
assert lib.cobevt_stage_trace_read(buf, 16 * n) == 0
full = torch.tensor(list(buf), dtype=torch.float64).reshape(n, 16)
t = full[:, :8]
names = ["tables", "V^T staging", "attention", "chain A (proj)", "chain B-D (MLP)", "store + LN", "next qkv"]
ph = t[:, 1:] - t[:, :-1]
print("last launch of the encoder (160 workgroups), phase medians in s_memtime ticks (10 ns each):")
for i, nm in enumerate(names):
    print("  %-18s median %7.0f   p90 %7.0f" % (nm, ph[:, i].median(), ph[:, i].quantile(0.9)))
print("  workgroup total   median %7.0f ; first start -> last end %7.0f ; start spread %7.0f" %
      ((t[:, 7] - t[:, 0]).median(), t[:, 7].max() - t[:, 0].min(), t[:, 0].max() - t[:, 0].min()))
fine = [("tables barrier -> key-row reads done", 1, 8), ("-> V loads issued", 8, 9), ("-> last K / V load returned", 9, 10),
        ("-> V^T written (+ score MFMAs)", 10, 11), ("-> barrier", 11, 2), ("barrier -> scores + bias + max", 2, 12), ("-> exp, PV, output tile", 12, 3)]
for nm, a, b in fine:
    d = full[:, b] - full[:, a]
    print("  %-42s median %7.0f   p90 %7.0f" % (nm, d.median(), d.quantile(0.9)))
