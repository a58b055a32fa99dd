#!/usr/bin/env python
"""Per-launch time of the attention kernels inside a replayed HIP graph on the frame's small-grid shapes: variant 0 = automatic,
1 = streaming kernel (128-key tiles where they apply), 2 = streaming kernel with 64-key tiles.  Run on the GPU box."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cobevt_amd import ops  # noqa: E402
from conv_graph_probe import graph_time  # noqa: E402

dev = torch.device("cuda:0")
if __name__ == "__main__":
    torch.manual_seed(0)
    # (name, mode, ncam, H, W, w1, w2, heads, batch, bias, mask)
    for name, mode, ncam, H, W, w1, w2, heads, B, bias, mask in [
            ("level-2 / global  B5 L1 Nq1024 Nk1024", 0, 1, 32, 32, 32, 32, 4, 5, False, False),
            ("global + 2-D bias B5 L1 Nq1024 Nk1024", 0, 1, 32, 32, 32, 32, 4, 5, True, False),
            ("fusion window     B1 L16 Nq320 Nk320", 0, 5, 32, 32, 8, 8, 4, 1, True, True),
            ("fusion grid       B1 L16 Nq320 Nk320", 1, 5, 32, 32, 8, 8, 4, 1, True, True),
            ("level-1 #2        B5 L16 Nq256 Nk256", 0, 1, 64, 64, 16, 16, 4, 5, False, False),
            ("LiDAR window      B1 L1024 Nq512 Nk512", 0, 8, 256, 256, 8, 8, 2, 1, True, True),
            ("LiDAR grid        B1 L1024 Nq512 Nk512", 1, 8, 256, 256, 8, 8, 2, 1, True, True)]:
        if os.environ.get("ONLY") and os.environ["ONLY"] not in name:
            continue
        d = heads * 32
        tm = ops.tokmap(mode, ncam, H, W, w1, w2)
        rows = B * ncam * H * W
        qkv = torch.randn(rows, 3 * d, device=dev).to(torch.bfloat16)
        out = torch.empty(rows, d, device=dev, dtype=torch.bfloat16)
        table = torch.randn((2 * ncam - 1) * (2 * w1 - 1) * (2 * w2 - 1), heads, device=dev) if bias else None
        mk = torch.ones(B, H, W, ncam, device=dev) if mask else None
        line = name + " |"
        ref = None
        for variant in [int(v) for v in os.environ.get('ATTN_VARIANTS', '0,1,2').split(',')]:
            fn = lambda: ops.window_attention(qkv, qkv, qkv, out, tm, tm, tm, B, heads, 0.17, 3 * d, 3 * d, 3 * d, d, koff=d, voff=2 * d,
                                              bias_table=table, bias_L=ncam, mask=mk, variant=variant, qsplit=int(os.environ.get('QSPLIT', '0')))
            us = graph_time(fn)
            fn()
            torch.cuda.synchronize()
            cur = out.float().clone()
            err = 0.0 if ref is None else (cur - ref).abs().max().item()
            ref = cur if ref is None else ref
            line += "  v%d %.1f us (diff %.2g)" % (variant, us, err)
        print(line, flush=True)
