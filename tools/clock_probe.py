#!/usr/bin/env python
"""Sustained shader clock / MFMA issue rate probe (tools/clock_probe.hip): what the ingredients of the conv main loop cost.
Build here: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/_probe/libclock_probe.so ; run on the GPU box."""
import ctypes
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libclock_probe.so"))
dev = torch.device("cuda:0")
w = torch.randint(0, 2 ** 31 - 1, (256 * 64 * 4,), dtype=torch.int32, device=dev)
NAMES = {0: "4 acc, constant operands", 1: "5 acc, 5 A regs + 1 B reg", 2: "+ ds_read_b128 per MFMA (A ring)", 3: "+ 1-KB B load per k-group (B ring)"}
for variant in (0, 1, 2, 3):
    for blocks, threads in ((256, 256), (256, 512), (512, 512)):
        iters = 20000
        nacc = 4 if variant == 0 else 5
        per_iter = 3 * nacc
        out = torch.zeros(blocks * 4, dtype=torch.int64, device=dev)
        t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        t0.record()
        rc = lib.clock_probe(ctypes.c_void_p(out.data_ptr()), ctypes.c_void_p(w.data_ptr()), variant, blocks, threads, iters,
                             ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
        t1.record()
        torch.cuda.synchronize()
        assert rc == 0, rc
        o = out.cpu().reshape(blocks, 4)
        cyc, ref = o[:, 0].double().mean().item(), o[:, 1].double().mean().item()
        ms = t0.elapsed_time(t1)
        waves = blocks * threads // 64
        flops = waves * iters * per_iter * 32 * 32 * 16 * 2.0
        wps = threads // 64 / 4.0 * max(1, blocks // 256)
        print("variant %d (%s), %d x %d threads = %.0f waves/SIMD: clock %.2f GHz; %.1f cycles per MFMA per SIMD (%.1f per wave); %.0f TF/s"
              % (variant, NAMES[variant], blocks, threads, wps, cyc / (ref / 100.0) / 1e3, cyc / (iters * per_iter * wps),
                 cyc / (iters * per_iter), flops / (ms * 1e-3) / 1e12), flush=True)
