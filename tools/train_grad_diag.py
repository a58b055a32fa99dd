#!/usr/bin/env python
"""Which piece of the training path sets the gradient error of the reduced CorpBEVT against torch autograd through the CPU oracle:
the worst parameters (relative to each tensor's scale) with the package's kernels, with the dense projections on the library GEMM,
and with the glue ops (BatchNorm / pooling / shuffles) on torch - each configuration run twice (fp32 atomics make runs differ)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import cases  # noqa: E402
from cobevt_amd import autograd as ag, host, synth  # noqa: E402
import oracle.corpbevt as o_model  # noqa: E402

dev = torch.device("cuda")
cfg = synth.corpbevt_small_config()
cfg["fax"]["self_attn"]["dropout"] = 0.0
cfg["fax_fusion"]["drop_out"] = 0.0
batch = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)


def build():
    m = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), cases.SEED).train().to(dev)
    for mod in m.modules():
        if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
            mod.eval()
    return m


m = build()
sd = {k: v.detach().cpu().clone() for k, v in m.state_dict().items()}
for k, p in m.named_parameters():
    sd[k] = p.detach().cpu().clone().requires_grad_(True)
with torch.enable_grad():
    out_ref = o_model.corpbevt_forward(sd, cfg, dict(batch))["dynamic_seg"]
    w = synth.procedural_input("train.w.diag", tuple(out_ref.shape), cases.SEED)
    (out_ref * w).sum().backward()
floor = 1e-3 * max(float(t.grad.abs().max()) for t in sd.values() if t.grad is not None)


def run(label):
    m.zero_grad(set_to_none=True)
    with torch.enable_grad():
        out = m({k: v.to(dev) for k, v in batch.items()})["dynamic_seg"]
        (out * w.to(dev)).sum().backward()
    errs = []
    for k, p in m.named_parameters():
        ref = sd[k].grad
        if ref is None or p.grad is None:
            continue
        errs.append((float((p.grad.cpu() - ref).abs().max()) / max(float(ref.abs().max()), floor), k))
    errs.sort(reverse=True)
    fwd = float((out.detach().cpu() - out_ref.detach()).abs().max() / out_ref.detach().abs().max())
    print("%-28s forward %.2e | worst: %s" % (label, fwd, ", ".join("%s %.2e" % (k.replace("encoder.encoder.", "enc."), e) for e, k in errs[:4])))


ag.USE_LIBRARY_GEMM, ag.USE_TORCH_GLUE = False, False
run("package kernels")
bns = [(k, mod) for k, mod in m.named_modules() if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm)]
stem = {id(mod) for k, mod in bns if k == "encoder.encoder.bn1"}
others = {id(mod) for k, mod in bns if k != "encoder.encoder.bn1"}


def grads(label):
    m.zero_grad(set_to_none=True)
    with torch.enable_grad():
        out = m({k: v.to(dev) for k, v in batch.items()})["dynamic_seg"]
        (out * w.to(dev)).sum().backward()
    torch.cuda.synchronize()
    return {k: p.grad.detach().clone() for k, p in m.named_parameters() if p.grad is not None}


ag.TORCH_GLUE_BN_IDS = stem | others
ga = grads("all torch BN")
ag.TORCH_GLUE_BN_IDS = others
gb = grads("HIP stem BN only")
gb2 = grads("HIP stem BN only (again)")
print("params whose gradient differs between `all torch BN` and `HIP stem BN only` (rel to the tensor's max):")
for k in ga:
    d = float((ga[k] - gb[k]).abs().max()) / max(float(ga[k].abs().max()), 1e-30)
    d2 = float((gb2[k] - gb[k]).abs().max()) / max(float(gb[k].abs().max()), 1e-30)
    if d > 1e-5 or d2 > 1e-5:
        bad = (ga[k] - gb[k]).abs() > 1e-4 * ga[k].abs().max()
        print("  %-50s diff %.2e (run-to-run %.2e) shape %s ptr %x mismatching elements %d of %d" % (k, d, d2, tuple(ga[k].shape), gb[k].data_ptr(), int(bad.sum()), bad.numel()))
ag.TORCH_GLUE_BN_IDS = set()
run("all HIP BN, torch pool")
ag.TORCH_GLUE_OPS = set()
ag.TORCH_GLUE_BN_IDS = stem
run("torch stem BN")
ag.TORCH_GLUE_BN_IDS = set()
