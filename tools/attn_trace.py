#!/usr/bin/env python
"""Where a workgroup of the K/V-resident attention kernel spends its time: builds a probe copy of the library with s_memtime
marks (tools/_probe/libcobevt_hip_restrace.so, build with `--build` on the CPU box), runs the level-0 launch shapes and prints
per-workgroup phase durations in shader cycles (tables | K/V staging | query-tile loop) plus the launch's start-to-end span.
Usage: python tools/attn_trace.py --build     (here)      /      python tools/attn_trace.py     (GPU box)"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "cobevt_amd", "csrc")
LIB = os.path.join(ROOT, "tools", "_probe", "libcobevt_hip_restrace.so")

if "--build" in sys.argv:
    sys.path.insert(0, ROOT)
    from cobevt_amd import build as b
    b.build(verbose=False)
    os.makedirs(os.path.dirname(LIB), exist_ok=True)
    obj = os.path.join(ROOT, "tools", "_probe", "attention_resident_trace.o")
    subprocess.check_call([b._hipcc()] + b.FLAGS + ["-DCOBEVT_RES_TRACE", "-c", os.path.join(CSRC, "attention_resident.hip"), "-o", obj])
    objs = [os.path.join(CSRC, s.replace(".hip", ".o")) for s in b.SOURCES if s != "attention_resident.hip"] + [obj]
    subprocess.check_call([b._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs)
    print("built", LIB)
    sys.exit(0)

os.environ["COBEVT_HIP_LIB"] = LIB
sys.path.insert(0, ROOT)
import torch  # noqa: E402
from cobevt_amd import lib as L, ops  # noqa: E402

dev = torch.device("cuda")
lib = L.load()
lib.cobevt_res_trace_read.restype = ctypes.c_int
lib.cobevt_res_trace_read.argtypes = [ctypes.c_void_p, ctypes.c_int]


def report(name, nwg):
    torch.cuda.synchronize()
    n = min(nwg, 8192)
    buf = (ctypes.c_ulonglong * (4 * n))()
    rc = lib.cobevt_res_trace_read(buf, 4 * n)
    assert rc == 0, rc
    t = torch.tensor(list(buf), dtype=torch.float64).reshape(n, 4)
    t0 = t[:, 0].min()
    ph = torch.stack([t[:, 1] - t[:, 0], t[:, 2] - t[:, 1], t[:, 3] - t[:, 2]], 1)
    print("%-28s %5d wgs | tables %7.0f  staging %7.0f  loop %8.0f cycles (median) | first start -> last end %9.0f cycles | "
          "start spread: median wg starts at %8.0f, last at %8.0f" %
          (name, n, ph[:, 0].median(), ph[:, 1].median(), ph[:, 2].median(), (t[:, 3].max() - t0), (t[:, 0] - t0).median(), (t[:, 0] - t0).max()))
    if n >= 2048 and os.environ.get("COBEVT_ATTN_PERSIST", "1") != "0":        # persistent workgroups: item = workgroup + 256 x iteration
        for it in range(n // 256):
            sl = ph[it * 256:(it + 1) * 256]
            print("    iteration %d of the persistent workgroups: tables %6.0f staging %6.0f loop %7.0f (median over 256 items)" %
                  (it, sl[:, 0].median(), sl[:, 1].median(), sl[:, 2].median()))
    for q in (0.1, 0.5, 0.9):
        k = int(q * (n - 1))
        order = torch.argsort(t[:, 0])
        i = order[k]
        print("    wg at start-quantile %.1f: start %8.0f tables %6.0f staging %6.0f loop %7.0f" % (q, t[i, 0] - t0, ph[i, 0], ph[i, 1], ph[i, 2]))


def cross(name, B, n, H, W, W1, W2, h, w, w1, w2, kmode, mean, heads=4, qsplit=0):
    d = heads * 32
    nq = n if mean else 1
    q = torch.randn(B, nq, H, W, d, device=dev).to(torch.bfloat16)
    kv = torch.randn(B * n, h, w, 2 * d, device=dev).to(torch.bfloat16)
    out = torch.empty(B, H, W, d, device=dev, dtype=torch.bfloat16)
    qmap, kmap, omap = ops.tokmap(0, nq, H, W, W1, W2), ops.tokmap(kmode, n, h, w, w1, w2), ops.tokmap(0, 1, H, W, W1, W2)
    for _ in range(3):
        ops.window_attention(q, kv, kv, out, qmap, kmap, omap, B, heads, 32 ** -0.5, d, 2 * d, 2 * d, d, koff=0, voff=d,
                             mean_q=mean, variant=0, qsplit=qsplit)
    report(name, B * qmap[6] * qmap[7] * heads * max(qsplit, 1))


def swap(name, L, w, H, mode=0, heads=2):
    """swap-fusion self attention with the 3-D bias + key mask (LiDAR FuseBEVT: 8 agents, 8 x 8 windows, 2 heads)"""
    d = heads * 32
    qkv = torch.randn(1, L, H, H, 3 * d, device=dev).to(torch.bfloat16)
    table = torch.randn((2 * L - 1) * (2 * w - 1) ** 2, heads, device=dev)
    mask = torch.ones(1, H, H, 1, L, device=dev)
    mask[0, :, :, :, L - 2:] = 0
    out = torch.empty(1, L, H, H, d, device=dev, dtype=torch.bfloat16)
    m = ops.tokmap(mode, L, H, H, w, w)
    for _ in range(3):
        ops.window_attention(qkv, qkv, qkv, out, m, m, m, 1, heads, 32 ** -0.5, 3 * d, 3 * d, 3 * d, d, qoff=0, koff=d, voff=2 * d,
                             bias_table=table, bias_L=L, mask=mask, variant=0)
    report(name, (H // w) ** 2 * heads)


if "--lidar" in sys.argv:
    swap("LiDAR window 8 agents 256x256", 8, 8, 256, 0)
    swap("LiDAR grid   8 agents 256x256", 8, 8, 256, 1)
    sys.exit(0)
cross("L0 #1 mean 5 agents", 5, 4, 128, 128, 16, 16, 64, 64, 8, 8, 0, True, qsplit=1)
cross("L0 #2 grid 5 agents", 5, 4, 128, 128, 16, 16, 64, 64, 8, 8, 1, False, qsplit=1)
cross("L1 #1 5 agents qs=2", 5, 4, 64, 64, 16, 16, 32, 32, 8, 8, 0, False, qsplit=2)
