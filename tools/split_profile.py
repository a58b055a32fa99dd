#!/usr/bin/env python
"""Per-kernel-family time of one eager 5-agent frame in the fp32 modes (exact fp32 MFMA vs the split-bf16 matrix path) and bf16:
HIP events around every C-ABI launch (ops.LaunchProfile), side-stream overlap off.  Usage: python tools/split_profile.py [agents]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobevt_amd import host, ops, synth  # noqa: E402
from cobevt_amd.host import pipeline  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 5
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.corpbevt_config(max_cav=max(5, A))
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).eval().to(dev)
batch = {k: v.to(dev) for k, v in synth.opv2v_batch(agents=A, max_cav=cfg["max_cav"], seed=0).items()}
for mode in ("fp32", "fp32_split", "bf16"):
    with host.compute_dtype(mode):
        run = pipeline.CapturedCorpBEVT(model, batch, use_graph=False)
        run.model.overlap_streams = False
        best = None
        for _ in range(3):
            with ops.LaunchProfile() as prof:
                run.eager_step()
            summ, shapes = prof.summary(), prof.summary(by_shape=True)
            tot = sum(d["ms"] for d in summ.values())
            if best is None or tot < best[0]:
                best = (tot, summ, shapes)
        run.model.overlap_streams = True
    tot, summ, shapes = best
    print("== %s: %.3f ms of timed launches per frame" % (mode, tot))
    for fam, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-12s %3d launches %8.3f ms  %7.1f TFLOP/s  %7.1f GB/s" % (fam, d["calls"], d["ms"], d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] else 0,
                                                                             d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] else 0))
    if mode != "bf16":
        for k, d in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])[:14]:
            print("      %-60s %3d x %8.1f us  %7.1f TFLOP/s" % (k[:60], d["calls"], d["ms"] * 1e3 / d["calls"], d["flops"] / (d["ms"] * 1e-3) / 1e12))
