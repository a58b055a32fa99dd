#!/usr/bin/env python
"""Sustained shader clock / MFMA rate probe (tools/clock_probe.hip).  Run on the GPU box."""
import ctypes
import os
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
lib = ctypes.CDLL(os.path.join(ROOT, "tools", "_probe", "libclock_probe.so"))
dev = torch.device("cuda:0")
for blocks, threads, iters in [(256, 256, 2000), (256, 256, 20000), (256, 256, 200000), (256, 512, 100000), (32, 256, 20000), (1, 64, 20000)]:
    out = torch.zeros(blocks * 4, dtype=torch.int64, device=dev)
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    t0.record()
    rc = lib.clock_probe(ctypes.c_void_p(out.data_ptr()), blocks, threads, iters, ctypes.c_void_p(torch.cuda.current_stream().cuda_stream))
    t1.record()
    torch.cuda.synchronize()
    assert rc == 0
    o = out.cpu().reshape(blocks, 4)
    cyc, ref = o[:, 0].double().mean().item(), o[:, 1].double().mean().item()
    ms = t0.elapsed_time(t1)
    waves = blocks * threads // 64
    flops = waves * iters * 4 * 32 * 32 * 16 * 2.0
    wps = threads // 64 / 4.0     # waves per SIMD in a block (one block per CU when blocks <= 256)
    print("blocks %d x %d thr, %d iters: %.3f ms event; s_memtime %.0f, s_memrealtime %.0f (100 MHz -> %.1f us) => shader clock %.2f GHz; "
          "%.1f cycles per MFMA per SIMD; %.0f TF/s" % (blocks, threads, iters, ms, cyc, ref, ref / 100.0, cyc / (ref / 100.0) / 1e3,
                                                      cyc / (iters * 4 * max(wps, 1.0)), flops / (ms * 1e-3) / 1e12))
