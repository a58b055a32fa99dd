#!/usr/bin/env python
"""How often do the HIP BatchNorm + ReLU (x * scale + shift, fused) and torch's batch_norm disagree on the SIGN of a pre-activation,
and by how much do the outputs differ - on the stem of the reduced CorpBEVT (diagnostic for tools/train_grad_diag.py)."""
import copy
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch  # noqa: E402
import cases  # noqa: E402
from cobevt_amd import autograd as ag, host, synth  # noqa: E402

dev = torch.device("cuda")
cfg = synth.corpbevt_small_config()
m = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), cases.SEED).train().to(dev)
for mod in m.modules():
    if isinstance(mod, torch.nn.modules.batchnorm._BatchNorm):
        mod.eval()
batch = synth.opv2v_batch(agents=2, cams=2, image=128, max_cav=3, seed=cases.SEED)
net = m.encoder.encoder
img = batch["inputs"].to(dev)
b, l, n, h, w, c = img.shape
x = img.reshape(b * l * n, h, w, c).permute(0, 3, 1, 2)
with torch.no_grad():
    z = ag.conv2d(x, net.conv1)
    bn = net.bn1
    y_hip = ag.BatchNormActFn.apply(z, None, bn.weight, bn.bias, bn, False, 0)
    y_t = torch.nn.functional.batch_norm(z, bn.running_mean, bn.running_var, bn.weight, bn.bias, False, 0.1, bn.eps)
    y_64 = torch.nn.functional.batch_norm(z.double(), bn.running_mean.double(), bn.running_var.double(), bn.weight.double(), bn.bias.double(),
                                          False, 0.1, bn.eps)
    for name, y in (("HIP", y_hip), ("torch fp32", y_t)):
        d = (y.double() - y_64).abs()
        flips = ((y > 0) != (y_64 > 0)).sum().item()
        print("%-10s vs fp64: max abs diff %.3e (output scale %.3e), sign disagreements %d of %d, elements with |y| < 1e-6: %d" %
              (name, d.max().item(), y_64.abs().max().item(), flips, y.numel(), int((y_64.abs() < 1e-6).sum())))
    print("HIP vs torch fp32: sign disagreements %d, max abs diff %.3e" % (((y_hip > 0) != (y_t > 0)).sum().item(), (y_hip - y_t).abs().max().item()))
    print("exact ties inside 3x3/s2 pooling windows (relu(y) of torch): windows whose two largest entries are equal and > 0:",
          end=" ")
    r = torch.relu(y_t)
    u = torch.nn.functional.unfold(r, 3, padding=1, stride=2).reshape(r.shape[0], r.shape[1], 9, -1)
    top = u.topk(2, dim=2).values
    print(int(((top[:, :, 0] == top[:, :, 1]) & (top[:, :, 0] > 0)).sum()), "of", top[:, :, 0].numel())
