#!/usr/bin/env python
"""Per-launch-shape timing of one eager CoBEVT frame (HIP events around every C-ABI call)."""
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import host, ops, synth  # noqa: E402

agents = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dtype = torch.float32 if "fp32" in sys.argv else torch.bfloat16
torch.set_grad_enabled(False)
host.set_compute_dtype(dtype)
dev = torch.device("cuda:0")
cfg = synth.corpbevt_config()
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).eval().to(dev)
batch = {k: v.to(dev) for k, v in synth.opv2v_batch(agents=agents).items()}
for _ in range(2):
    model(dict(batch))
best = None
for _ in range(3):
    with ops.LaunchProfile() as prof:
        model(dict(batch))
    s = prof.summary(by_shape=True)
    tot = sum(d["ms"] for d in s.values())
    if best is None or tot < best[0]:
        best = (tot, s)
tot, s = best
print("timed launches: %.3f ms per frame" % tot)
for k, d in sorted(s.items(), key=lambda kv: -kv[1]["ms"])[:int(os.environ.get("TOP", "40"))]:
    print("%8.1f us %3d calls %7.1f TF/s %7.1f GB/s  %s" % (d["ms"] * 1e3, d["calls"], d["flops"] / d["ms"] / 1e9,
                                                          d["bytes"] / d["ms"] / 1e6, k))
