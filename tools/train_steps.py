#!/usr/bin/env python
"""N optimisation steps of the full corpbevt.yaml CorpBEVT (train_camera.py:143-179: forward, VanillaSegLoss, backward, AdamW), for
`rocprofv3 --kernel-trace --stats -- python tools/train_steps.py --agents 5 --steps 4 [--amp]` (profiles/rNN_train*_kernel_trace.txt)."""
import argparse
import copy
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import host, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=5)
ap.add_argument("--steps", type=int, default=4)
ap.add_argument("--amp", action="store_true", help="bf16 autocast region around forward + loss (train_camera.py --half)")
args = ap.parse_args()
cfg = synth.corpbevt_config()
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
batch = {k: v.cuda() for k, v in synth.opv2v_batch(agents=args.agents).items()}
crit = host.VanillaSegLoss({"d_weights": 75.0, "s_weights": 15.0, "l_weights": 50, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"})
opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
gt = {"gt_dynamic": (torch.rand(1, 1, 256, 256, device="cuda") > 0.9).long(), "gt_static": torch.zeros(1, 1, 256, 256, device="cuda", dtype=torch.long)}
for i in range(args.steps):
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=args.amp):
        loss = crit(model(dict(batch)), gt)
    loss.backward()
    opt.step()
torch.cuda.synchronize()
print("steps %d agents %d amp %s final loss %.4f" % (args.steps, args.agents, args.amp, float(loss.detach())))
