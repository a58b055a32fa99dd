#!/usr/bin/env python
"""Per-launch-shape time of one eager LiDAR FuseBEVT forward (bench.py --workload lidar's module): HIP events around every C-ABI launch."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from cobevt_amd import host, ops, synth  # noqa: E402

torch.set_grad_enabled(False)
dev = torch.device("cuda:0")
host.set_compute_dtype(torch.bfloat16)
enc = synth.fill_module_(host.SwapFusionEncoder(dict(bench.LIDAR_ARGS)), 0).eval().to(dev)
x, mask = bench.lidar_inputs(dev, seed=0)
for _ in range(2):
    enc(x, mask)
best = None
for _ in range(3):
    with ops.LaunchProfile() as prof:
        enc(x, mask)
    s = prof.summary(by_shape=True)
    tot = sum(d["ms"] for d in s.values())
    if best is None or tot < best[0]:
        best = (tot, s)
print("%.3f ms of timed launches" % best[0])
for k, d in sorted(best[1].items(), key=lambda kv: -kv[1]["ms"]):
    print("   %-64s %2d x %8.1f us  %7.1f GB/s" % (k[:64], d["calls"], d["ms"] * 1e3 / d["calls"], d["bytes"] / (d["ms"] * 1e-3) / 1e9))
