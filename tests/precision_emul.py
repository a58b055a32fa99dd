#!/usr/bin/env python3
"""What precision does a matrix operand need for the north-star's 1e-3?  (measure before building.)

Runs the CPU oracle (the pinned restatement of the reference, fp32) on the full-size corpbevt.yaml frame of the parity tests and
again with every matrix product's operands rounded the way a candidate device mode would round them, and prints the logit
deviation in the norms the tests gate (max|d| / max|ref|, ||d|| / ||ref||, arg-max agreement).  Only products with a contraction
length >= 8 are touched: the camera-geometry products (K = 3 / 4) are fp32 VALU code in every device mode.

    modes:  fp16      both operands rounded to fp16                      (1 f16 MFMA per 16 k: the bf16 kernels' matrix time)
            fp16_w    weights (F.linear / F.conv2d weight; K, V of an attention product) rounded to fp16, the other operand exact
            fp16_a    activations rounded, weights exact
            fp16_st   fp16 storage emulation: operands AND results of every product rounded to fp16
            fp16_w3   fp16_w on the k x k (k > 1) convolutions only - the matrix-bound kernels (3x3 strips, BasicBlocks, stem) -
                      everything else exact; fp16_w1 = the complement (1x1 / Linear / attention products only)
            fp16_we   fp16_w on the ResNet encoder's k x k convolutions only (80 % of the frame's flops)
            fp16_e2   BOTH operands fp16 in the ResNet encoder's k x k convolutions only (would run them at the bf16 kernels' matrix rate)
            fp16_qp   fp16_e2 plus, in the attention products only, the FIRST operand (queries; probabilities) rounded to fp16, keys / values exact
            bf16      both operands rounded to bf16 (calibration: the shipped bf16 mode measures 1.06e-2 / 4.5e-3 on this frame)

This is a tool (it imports oracle/ as the thing evaluated, like bench.py's cpu_baseline leg); the product never does.
usage: python tools/precision_emul.py [--agents 5] [--modes fp16,fp16_w,...] [--seeds 0,1]
"""
import argparse
import copy
import os
import sys
import time

import torch
import torch.nn.functional as F
from torch.overrides import TorchFunctionMode

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def _r(x, dt):
    return x.to(dt).to(torch.float32)


class OperandRounding(TorchFunctionMode):
    """Rounds the operands of F.linear / F.conv2d / matmul as `mode` says; `first` = the activation-side operand."""

    def __init__(self, mode):
        super().__init__()
        self.mode = mode
        self.dt = torch.bfloat16 if mode.startswith("bf16") else torch.float16
        self.n = 0
        self.encoder_ids = set()

    def _pair(self, a, w):
        m = self.mode
        if m == "fp16_qp" and getattr(self, "_in_matmul", False):
            return _r(a, self.dt), w
        if m in ("fp16", "bf16", "fp16_st", "fp16_e2", "fp16_qp"):
            return _r(a, self.dt), _r(w, self.dt)
        if m in ("fp16_w", "fp16_w3", "fp16_w1", "fp16_we"):
            return a, _r(w, self.dt)
        if m == "fp16_a":
            return _r(a, self.dt), w
        raise ValueError(m)

    def _out(self, y):
        return _r(y, self.dt) if self.mode == "fp16_st" else y

    def __torch_function__(self, func, types, args=(), kwargs=None):
        kwargs = kwargs or {}
        if func is F.conv2d:
            x, w = args[0], args[1]
            big = w.shape[2] * w.shape[3] > 1
            skip = (self.mode == "fp16_w3" and not big) or (self.mode == "fp16_w1" and big) or (self.mode in ("fp16_we", "fp16_e2", "fp16_qp") and not (big and id(w) in self.encoder_ids))
            if w.shape[1] * w.shape[2] * w.shape[3] >= 8 and x.dtype == torch.float32 and not skip:
                self.n += 1
                x, w = self._pair(x, w)
                return self._out(func(x, w, *args[2:], **kwargs))
        elif func is F.linear:
            x, w = args[0], args[1]
            if w.shape[-1] >= 8 and x.dtype == torch.float32 and self.mode not in ("fp16_w3", "fp16_we", "fp16_e2", "fp16_qp"):
                self.n += 1
                x, w = self._pair(x, w)
                return self._out(func(x, w, *args[2:], **kwargs))
        elif func in (torch.matmul, torch.Tensor.matmul, torch.Tensor.__matmul__, torch.bmm):
            a, b = args[0], args[1]
            if torch.is_tensor(a) and torch.is_tensor(b) and a.dtype == torch.float32 and a.shape[-1] >= 8 and self.mode not in ("fp16_w3", "fp16_we", "fp16_e2"):
                self.n += 1
                self._in_matmul = True
                # attention products: q k^T (first = queries, second = keys), att v (first = probabilities, second = values);
                # the device kernels feed K / V^T as the MFMA's A operand and Q / P^T as B - "weights" = the key / value side
                try:
                    a, b = self._pair(a, b)
                finally:
                    self._in_matmul = False
                return self._out(func(a, b, *args[2:], **kwargs))
        return func(*args, **kwargs)


def stats(y, ref):
    d = (y - ref).double()
    mx = float(d.abs().max() / ref.abs().max())
    rms = float(d.square().sum().sqrt() / ref.double().square().sum().sqrt())
    agree = float((y.argmax(2) == ref.argmax(2)).float().mean())
    return mx, rms, agree


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--agents", type=int, default=5)
    ap.add_argument("--modes", default="fp16,fp16_w,fp16_a,fp16_st,bf16")
    ap.add_argument("--seeds", default="0")
    ap.add_argument("--threads", type=int, default=8)
    a = ap.parse_args()
    torch.set_num_threads(a.threads)
    from cobevt_amd import host, synth
    from oracle import corpbevt as o_model
    cfg = synth.corpbevt_config()
    for seed in [int(s) for s in a.seeds.split(",")]:
        m = synth.fill_module_(host.CorpBEVT(copy.deepcopy(cfg)), seed).eval()
        synth.balance_seg_head_(m, a.agents) if hasattr(synth, "balance_seg_head_") and os.environ.get("BALANCE") else None
        sd = m.state_dict()
        batch = synth.opv2v_batch(agents=a.agents, seed=seed)
        t0 = time.time()
        with torch.no_grad():
            ref = o_model.corpbevt_forward(sd, cfg, dict(batch))["dynamic_seg"]
        print("seed %d: oracle fp32 frame %.1f s; logits %s max|ref| %.3f" % (seed, time.time() - t0, tuple(ref.shape), float(ref.abs().max())), flush=True)
        for mode in a.modes.split(","):
            om = OperandRounding(mode)
            om.encoder_ids = {id(v) for k, v in sd.items() if k.startswith("encoder.")}
            with torch.no_grad(), om:
                y = o_model.corpbevt_forward(sd, cfg, dict(batch))["dynamic_seg"]
            mx, rms, ag = stats(y, ref)
            print("seed %d  %-8s products %3d  logits max-rel %.3e  rms-rel %.3e  arg-max agreement %.5f" % (seed, mode, om.n, mx, rms, ag), flush=True)


if __name__ == "__main__":
    main()
