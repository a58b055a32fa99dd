#!/usr/bin/env python
"""Which Python lines of the training path launch torch's own (ATen) kernels: one bf16-autocast optimisation step of the full corpbevt.yaml
CorpBEVT under torch.profiler (CPU activity + Python stacks), ATen ops that launch a kernel counted by the innermost cobevt_amd /
tools source line.  The HIP entry points are ctypes calls and do not show up - this lists what is NOT yet a HIP kernel of the package.

    python tools/train_op_audit.py [--agents 2] [--fp32]        # on the GPU box
"""
import argparse
import collections
import copy
import os
import sys

import torch
from torch.profiler import ProfilerActivity, profile

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import host, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=2)
ap.add_argument("--fp32", action="store_true")
args = ap.parse_args()
cfg = synth.corpbevt_config()
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
batch = {k: v.cuda() for k, v in synth.opv2v_batch(agents=args.agents).items()}
crit = host.VanillaSegLoss({"d_weights": 75.0, "s_weights": 15.0, "l_weights": 50, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"})
opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
gt = {"gt_dynamic": (torch.rand(1, 1, 256, 256, device="cuda") > 0.9).long(), "gt_static": torch.zeros(1, 1, 256, 256, device="cuda", dtype=torch.long)}


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
        loss = crit(model(dict(batch)), gt)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU], with_stack=True, record_shapes=False) as prof:
    step()
    torch.cuda.synchronize()

# ATen ops that (normally) launch one kernel each; views / metadata ops are left out
LAUNCHING = ("aten::copy_", "aten::fill_", "aten::zero_", "aten::add", "aten::add_", "aten::mul", "aten::mul_", "aten::sub", "aten::div", "aten::div_",
             "aten::mean", "aten::sum", "aten::neg", "aten::native_dropout", "aten::native_dropout_backward", "aten::index", "aten::index_put_",
             "aten::flip", "aten::cat", "aten::bmm", "aten::mm", "aten::addmm", "aten::linalg_vector_norm", "aten::clamp_min", "aten::where",
             "aten::masked_fill_", "aten::sqrt", "aten::rsqrt", "aten::pow", "aten::exp", "aten::_foreach_add_", "aten::_foreach_mul_",
             "aten::_foreach_addcdiv_", "aten::_foreach_addcmul_", "aten::_foreach_sqrt", "aten::_foreach_div_", "aten::_foreach_lerp_",
             "aten::constant_pad_nd", "aten::relu", "aten::threshold_backward", "aten::sigmoid", "aten::eq", "aten::gt", "aten::lt", "aten::ne")
by_line = collections.Counter()
by_op = collections.Counter()
for ev in prof.events():
    if ev.name not in LAUNCHING:
        continue
    where = "<backward / no package frame>"
    for fr in ev.stack:                      # innermost first
        if "cobevt_amd" in fr or "/tools/" in fr:
            where = fr.split("cobevt_amd/")[-1] if "cobevt_amd/" in fr else fr.split("/tools/")[-1]
            break
    by_line[(where, ev.name)] += 1
    by_op[ev.name] += 1
print("one %s training step, %d agents: %d kernel-launching ATen ops" % ("fp32" if args.fp32 else "bf16-autocast", args.agents, sum(by_op.values())))
print("by op:", ", ".join("%s %d" % kv for kv in by_op.most_common(20)))
print("by source line (innermost package frame):")
for (where, name), cnt in by_line.most_common(70):
    print("  %5d  %-22s %s" % (cnt, name, where))
