#!/usr/bin/env python
"""Per-kernel-family time of one eager 5-agent frame in the fp32 modes (exact fp32 MFMA vs the split-bf16 matrix path) and bf16:
HIP events around every C-ABI launch (ops.LaunchProfile), side-stream overlap off.  Usage: python tools/split_profile.py [agents] [modes, comma separated]"""
import copy
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobevt_amd import host, ops, synth  # noqa: E402
from cobevt_amd.host import pipeline  # noqa: E402

A = int(sys.argv[1]) if len(sys.argv) > 1 else 5
MODES = sys.argv[2].split(",") if len(sys.argv) > 2 else ["fp32", "fp32_split", "fp32_fast", "bf16"]
torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
cfg = synth.corpbevt_config(max_cav=max(5, A))
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).eval().to(dev)
batch = {k: v.to(dev) for k, v in synth.opv2v_batch(agents=A, max_cav=cfg["max_cav"], seed=0).items()}
for mode in MODES:
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
    with host.compute_dtype(mode):
        rg = pipeline.CapturedCorpBEVT(model, batch, use_graph=True)
        for _ in range(3):
            rg.step()
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(20):
            rg.step()
        ev[1].record()
        torch.cuda.synchronize()
        one = ev[0].elapsed_time(ev[1]) / 20
        rp = pipeline.PipelinedCorpBEVT(model, batch, depth=3)
        for _ in range(8):
            rp.step()
        torch.cuda.synchronize()
        ev[0].record()
        for _ in range(30):
            rp.step()
        ev[1].record()
        torch.cuda.synchronize()
        three = ev[0].elapsed_time(ev[1]) / 30
        del rg, rp
    print("== %s: one frame at a time from a captured graph %.3f ms = %.1f frames/s; three frames in flight %.3f ms per step = %.1f frames/s"
          % (mode, one, 1e3 / one, three, 1e3 / three))
    print("== %s: %.3f ms of timed launches per frame" % (mode, tot))
    for fam, d in sorted(summ.items(), key=lambda kv: -kv[1]["ms"]):
        print("   %-12s %3d launches %8.3f ms  %7.1f TFLOP/s  %7.1f GB/s" % (fam, d["calls"], d["ms"], d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] else 0,
                                                                             d["bytes"] / (d["ms"] * 1e-3) / 1e9 if d["ms"] else 0))
    if mode != "bf16":
        for k, d in sorted(shapes.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("SPLIT_PROFILE_TOP", "14"))]:
            print("      %-60s %3d x %8.1f us  %7.1f TFLOP/s" % (k[:60], d["calls"], d["ms"] * 1e3 / d["calls"], d["flops"] / (d["ms"] * 1e-3) / 1e12))
