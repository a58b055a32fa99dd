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


for rep in range(2):
    ag.USE_LIBRARY_GEMM, ag.USE_TORCH_GLUE = False, False
    run("package kernels #%d" % rep)
for rep in range(2):
    ag.USE_LIBRARY_GEMM, ag.USE_TORCH_GLUE = True, False
    run("library GEMM linears #%d" % rep)
for rep in range(2):
    ag.USE_LIBRARY_GEMM, ag.USE_TORCH_GLUE = False, True
    run("torch glue #%d" % rep)
ag.USE_LIBRARY_GEMM, ag.USE_TORCH_GLUE = True, True
run("library GEMM + torch glue")
