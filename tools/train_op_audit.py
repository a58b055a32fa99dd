#!/usr/bin/env python
"""Which Python lines of the training path still launch torch's own (ATen) kernels: one optimisation step of the full corpbevt.yaml
CorpBEVT under a TorchDispatchMode that sees every ATen call, each attributed to the innermost cobevt_amd / tools source line on the
Python stack (ops issued by the autograd engine's built-in backward nodes have no Python frame: they are listed as such).  The HIP
entry points are ctypes calls and do not show up - this lists what is NOT yet a HIP kernel of the package.

    python tools/train_op_audit.py [--agents 2] [--fp32]        # on the GPU box
"""
import argparse
import collections
import copy
import os
import sys
import traceback

import torch
from torch.utils._python_dispatch import TorchDispatchMode

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from cobevt_amd import host, synth   # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--agents", type=int, default=2)
ap.add_argument("--fp32", action="store_true")
ap.add_argument("--top", type=int, default=60)
args = ap.parse_args()
cfg = synth.corpbevt_config()
model = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), 0).train().cuda()
batch = {k: v.cuda() for k, v in synth.opv2v_batch(agents=args.agents).items()}
crit = host.VanillaSegLoss({"d_weights": 75.0, "s_weights": 15.0, "l_weights": 50, "d_coe": 2.0, "s_coe": 0.0, "target": "dynamic"})
opt = torch.optim.AdamW(model.parameters(), lr=2e-4)
gt = {"gt_dynamic": (torch.rand(1, 1, 256, 256, device="cuda") > 0.9).long(), "gt_static": torch.zeros(1, 1, 256, 256, device="cuda", dtype=torch.long)}

# metadata / view ops launch nothing
NO_KERNEL = ("view", "reshape", "permute", "transpose", "expand", "slice", "select", "unsqueeze", "squeeze", "as_strided", "detach", "alias",
             "t.default", "empty", "_unsafe_view", "stride", "size", "is_", "lift_fresh", "split", "unbind", "chunk", "narrow", "unfold",
             "_local_scalar_dense", "item", "record_stream", "_has_compatible_shallow_copy_type", "set_", "resize_", "sym_", "numel", "dim")


class Audit(TorchDispatchMode):
    def __init__(self):
        super().__init__()
        self.by_line = collections.Counter()
        self.by_op = collections.Counter()

    def __torch_dispatch__(self, func, types, args=(), kwargs=None):
        name = str(func).replace("aten.", "")
        if not any(k in name for k in NO_KERNEL):
            where = "<autograd engine / torch internals>"
            for fr in reversed(traceback.extract_stack()[:-1]):
                fn = fr.filename
                if ("cobevt_amd" in fn or os.sep + "tools" + os.sep in fn) and "train_op_audit" not in fn:
                    where = "%s:%d %s" % (fn.split("cobevt_amd" + os.sep)[-1] if "cobevt_amd" in fn else os.path.basename(fn), fr.lineno, fr.name)
                    break
            self.by_line[(where, name)] += 1
            self.by_op[name] += 1
        return func(*args, **(kwargs or {}))


def step():
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cuda", dtype=torch.bfloat16, enabled=not args.fp32):
        loss = crit(model(dict(batch)), gt)
    loss.backward()
    opt.step()


for _ in range(2):
    step()
torch.cuda.synchronize()
audit = Audit()
with audit:
    step()
torch.cuda.synchronize()
print("one %s training step, %d agents: %d ATen calls that (normally) launch a kernel" % ("fp32" if args.fp32 else "bf16-autocast", args.agents, sum(audit.by_op.values())))
print("by op: " + ", ".join("%s %d" % kv for kv in audit.by_op.most_common(24)))
print("by source line (innermost package frame on the Python stack):")
for (where, name), cnt in audit.by_line.most_common(args.top):
    print("  %5d  %-28s %s" % (cnt, name, where))
