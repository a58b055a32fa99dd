#!/usr/bin/env python
"""A/B of the two attention kernels (K/V-resident vs streaming) on the launch shapes of the bench workloads: time per launch
(HIP events, median of `reps`), algorithmic TFLOP/s, and the largest difference between the two kernels' outputs / against an
fp32 torch reference on a slice.  Usage (GPU box): python tools/attn_probe.py [--reps 30] [--qsplits 1,2,4]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--reps", type=int, default=30)
ap.add_argument("--qsplits", default="0")
ap.add_argument("--cases", default="", help="comma-separated substrings of the case names to run (default: all)")
args = ap.parse_args()
dev = torch.device("cuda")
torch.manual_seed(0)


def timed(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        fn()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    return ts[len(ts) // 2]


def wanted(name):
    return not args.cases or any(c in name for c in args.cases.split(","))


def cross_case(name, B, n, H, W, W1, W2, h, w, w1, w2, kmode, mean, heads=4):
    if not wanted(name):
        return
    d = heads * 32
    nq = n if mean else 1
    q = torch.randn(B, nq, H, W, d, device=dev).to(torch.bfloat16)
    kv = torch.randn(B * n, h, w, 2 * d, device=dev).to(torch.bfloat16)         # keys / values side by side like the product path
    out = [torch.empty(B, H, W, d, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    qmap, kmap, omap = ops.tokmap(0, nq, H, W, W1, W2), ops.tokmap(kmode, n, h, w, w1, w2), ops.tokmap(0, 1, H, W, W1, W2)
    L = qmap[6] * qmap[7]
    Nq, Nk = nq * W1 * W2, n * w1 * w2
    flops = 4.0 * B * L * heads * Nq * Nk * 32

    def run(o, variant, qsplit=0):
        ops.window_attention(q, kv, kv, o, qmap, kmap, omap, B, heads, 32 ** -0.5, d, 2 * d, 2 * d, d, koff=0, voff=d,
                             mean_q=mean, variant=variant, qsplit=qsplit)
    t_stream = timed(lambda: run(out[1], 1), args.reps)
    line = "%-34s Nq %5d Nk %4d L %4d | streaming %7.1f us %6.1f TF/s" % (name, Nq, Nk, B * L * heads, t_stream, flops / t_stream / 1e6)
    for qs in [int(x) for x in args.qsplits.split(",")]:
        t_res = timed(lambda: run(out[0], 0, qs), args.reps)
        diff = (out[0].float() - out[1].float()).abs().max().item()
        line += " | resident(qs=%d) %7.1f us %6.1f TF/s diff %.2e" % (qs, t_res, flops / t_res / 1e6, diff)
    print(line, flush=True)


def swap_case(name, B, Lag, H, W, w, mode, heads):
    if not wanted(name):
        return
    d = heads * 32
    qkv = torch.randn(B, Lag, H, W, 3 * d, device=dev).to(torch.bfloat16)
    table = torch.randn((2 * Lag - 1) * (2 * w - 1) ** 2, heads, device=dev)
    mask = torch.ones(B, H, W, 1, Lag, device=dev)
    mask[:, :, :, :, Lag - 2:] = 0
    mask[:, : H // 3, W // 2:, :, 1] = 0
    out = [torch.empty(B, Lag, H, W, d, device=dev, dtype=torch.bfloat16) for _ in range(2)]
    m = ops.tokmap(mode, Lag, H, W, w, w)
    Lw = m[6] * m[7]
    N = Lag * w * w
    flops = 4.0 * B * Lw * heads * N * N * 32

    def run(o, variant, qsplit=0):
        ops.window_attention(qkv, qkv, qkv, o, m, m, m, B, heads, 32 ** -0.5, 3 * d, 3 * d, 3 * d, d, koff=d, voff=2 * d,
                             bias_table=table, bias_L=Lag, mask=mask, variant=variant, qsplit=qsplit)
    t_stream = timed(lambda: run(out[1], 1), args.reps)
    line = "%-34s N %5d windows*heads %5d | streaming %7.1f us %6.1f TF/s" % (name, N, B * Lw * heads, t_stream, flops / t_stream / 1e6)
    for qs in [int(x) for x in args.qsplits.split(",")]:
        t_res = timed(lambda: run(out[0], 0, qs), args.reps)
        diff = (out[0].float() - out[1].float()).abs().max().item()
        line += " | resident(qs=%d) %7.1f us %6.1f TF/s diff %.2e" % (qs, t_res, flops / t_res / 1e6, diff)
    print(line, flush=True)


cross_case("L0 #1 mean (5 agents)", 5, 4, 128, 128, 16, 16, 64, 64, 8, 8, 0, True)
cross_case("L0 #2 grid keys", 5, 4, 128, 128, 16, 16, 64, 64, 8, 8, 1, False)
cross_case("L1 #1 window keys", 5, 4, 64, 64, 16, 16, 32, 32, 8, 8, 0, False)
cross_case("L1 #2 grid keys", 5, 4, 64, 64, 16, 16, 32, 32, 8, 8, 1, False)
cross_case("L0 #1 mean (2 agents)", 2, 4, 128, 128, 16, 16, 64, 64, 8, 8, 0, True)
cross_case("nuScenes L0 #1 mean 6 cams", 1, 6, 100, 100, 10, 10, 60, 120, 6, 12, 0, True, heads=1)
cross_case("nuScenes L0 #2", 1, 6, 100, 100, 10, 10, 60, 120, 6, 12, 1, False, heads=1)
swap_case("fusion window (5 agents, 32x32)", 1, 5, 32, 32, 8, 0, 4)
swap_case("fusion grid", 1, 5, 32, 32, 8, 1, 4)
swap_case("LiDAR window (8 x 256x256)", 1, 8, 256, 256, 8, 0, 2)
swap_case("LiDAR grid", 1, 8, 256, 256, 8, 1, 2)
