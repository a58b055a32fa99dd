"""Oracle for FuseBEVT (swap fusion).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows opv2v/opencood/models/fusion_modules/swap_fusion_modules.py and base_transformer.py:102-124.
"""
import numpy as np
import torch
import torch.nn.functional as F


def relative_position_index_3d(agent_size, window_size):
    """swap_fusion_modules.py:63-85 — token t = (l*w + a)*w + b;
    index[t1, t2] = (l1-l2+L-1)(2w-1)^2 + (a1-a2+w-1)(2w-1) + (b1-b2+w-1)."""
    L, w = agent_size, window_size
    l, a, b = np.meshgrid(np.arange(L), np.arange(w), np.arange(w), indexing="ij")
    l, a, b = l.reshape(-1), a.reshape(-1), b.reshape(-1)
    dl = l[:, None] - l[None, :] + L - 1
    da = a[:, None] - a[None, :] + w - 1
    db = b[:, None] - b[None, :] + w - 1
    return (dl * (2 * w - 1) * (2 * w - 1) + da * (2 * w - 1) + db).astype(np.int64)


def swap_attention(sd, pfx, x, mask, dim_head, agent_size, window_size):
    """Attention.forward, swap_fusion_modules.py:87-128.  x (b l X Y w1 w2 d) already LayerNorm'ed;
    mask (b X Y w1 w2 1 l) or None.  Returns same shape as x."""
    b, l, X, Y, w1, w2, d = x.shape
    heads = d // dim_head
    scale = dim_head ** -0.5
    t = x.permute(0, 2, 3, 1, 4, 5, 6).reshape(b * X * Y, l * w1 * w2, d)           # '(b x y) (l w1 w2) d'
    q, k, v = F.linear(t, sd[pfx + "to_qkv.weight"]).chunk(3, dim=-1)
    split = lambda z: z.reshape(z.shape[0], z.shape[1], heads, dim_head).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    q = q * scale                                                                    # scale BEFORE the GEMM (:100)
    sim = torch.matmul(q, k.transpose(-1, -2))
    idx = torch.from_numpy(relative_position_index_3d(agent_size, window_size))
    bias = sd[pfx + "relative_position_bias_table.weight"][idx]                      # i j h
    sim = sim + bias.permute(2, 0, 1)
    if mask is not None:
        # 'b x y w1 w2 e l -> (b x y) e (l w1 w2)'
        m = mask.permute(0, 1, 2, 5, 6, 3, 4).reshape(b * X * Y, mask.shape[5], l * w1 * w2)
        sim = sim.masked_fill(m.unsqueeze(1) == 0, -float("inf"))
    attn = sim.softmax(dim=-1)
    out = torch.matmul(attn, v)                                                      # (bxy) h n dh
    out = out.permute(0, 2, 1, 3).reshape(b * X * Y, l, w1, w2, d)
    out = F.linear(out, sd[pfx + "to_out.0.weight"])
    return out.reshape(b, X, Y, l, w1, w2, d).permute(0, 3, 1, 2, 4, 5, 6)


def _ln(x, sd, key):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def _ffn(sd, pfx, x):
    """FeedForward, base_transformer.py:112-124 (dropout is identity in eval)."""
    y = F.gelu(F.linear(x, sd[pfx + "net.0.weight"], sd[pfx + "net.0.bias"]))
    return F.linear(y, sd[pfx + "net.3.weight"], sd[pfx + "net.3.bias"])


def swap_stage(sd, attn_pfx, ffd_pfx, xl, mask, mode, dim_head, agent_size, window_size):
    """One half of a block on a channels-last map: PreNormResidual(Attention) + PreNormResidual(FeedForward) over the window
    (mode 0, :172-179) or dilated-grid (mode 1, :181-190) partition.  xl (b l h w d); mask (b h w 1 l) | None -> (b l h w d)."""
    w = window_size
    b, l, h, wd, d = xl.shape
    X, Y = h // w, wd // w
    e = mask.shape[3] if mask is not None else 1
    if mode == 0:       # 'b m d (x w1) (y w2) -> b m x y w1 w2 d' ; mask 'b (x w1) (y w2) e l -> b x y w1 w2 e l'
        xp = xl.reshape(b, l, X, w, Y, w, d).permute(0, 1, 2, 4, 3, 5, 6)
        mp = mask.reshape(b, X, w, Y, w, e, l).permute(0, 1, 3, 2, 4, 5, 6) if mask is not None else None
    else:               # 'b m d (w1 x) (w2 y) -> b m x y w1 w2 d' ; mask 'b (w1 x) (w2 y) e l -> b x y w1 w2 e l'
        xp = xl.reshape(b, l, w, X, w, Y, d).permute(0, 1, 3, 5, 2, 4, 6)
        mp = mask.reshape(b, w, X, w, Y, e, l).permute(0, 2, 4, 1, 3, 5, 6) if mask is not None else None
    xp = swap_attention(sd, attn_pfx + "fn.", _ln(xp, sd, attn_pfx + "norm"), mp, dim_head, agent_size, w) + xp
    xp = _ffn(sd, ffd_pfx + "fn.", _ln(xp, sd, ffd_pfx + "norm")) + xp
    if mode == 0:
        return xp.permute(0, 1, 2, 4, 3, 5, 6).reshape(b, l, h, wd, d)
    return xp.permute(0, 1, 4, 2, 5, 3, 6).reshape(b, l, h, wd, d)                    # 'b m x y w1 w2 d -> (w1 x) (w2 y)'


def swap_fusion_block(sd, names, x, mask, dim_head, agent_size, window_size):
    """SwapFusionBlockMask.forward (:165-192) / SwapFusionBlock (:209-225).  x (b l d h w); mask (b h w 1 l)|None.
    names = key prefixes of (window_attention, window_ffd, grid_attention, grid_ffd) PreNormResidual modules."""
    xl = x.permute(0, 1, 3, 4, 2)                                                     # b l h w d
    xl = swap_stage(sd, names[0], names[1], xl, mask, 0, dim_head, agent_size, window_size)
    xl = swap_stage(sd, names[2], names[3], xl, mask, 1, dim_head, agent_size, window_size)
    return xl.permute(0, 1, 4, 2, 3)


def block_names(pfx, i, use_mask):
    base = "%slayers.%d." % (pfx, i)
    if use_mask:
        return [base + "window_attention.", base + "window_ffd.", base + "grid_attention.", base + "grid_ffd."]
    return [base + "block.1.", base + "block.2.", base + "block.5.", base + "block.6."]


def mlp_head(sd, pfx, xl):
    """mlp_head :275-281 on a channels-last map: mean over agents, LayerNorm, Linear.  xl (b l h w d) -> (b h w d)"""
    y = xl.mean(dim=1)
    y = F.layer_norm(y, (y.shape[-1],), sd[pfx + "mlp_head.2.weight"], sd[pfx + "mlp_head.2.bias"], 1e-5)
    return F.linear(y, sd[pfx + "mlp_head.3.weight"], sd[pfx + "mlp_head.3.bias"])


def swap_fusion_encoder(sd, pfx, args, x, mask=None):
    """SwapFusionEncoder.forward, swap_fusion_modules.py:283-286 with mlp_head :275-281.
    x (b m d h w), mask (b h w 1 m) -> (b d h w)."""
    use_mask = bool(args.get("mask", False))
    for i in range(args["depth"]):
        base = "%slayers.%d." % (pfx, i)
        if use_mask:
            names = [base + "window_attention.", base + "window_ffd.", base + "grid_attention.", base + "grid_ffd."]
            x = swap_fusion_block(sd, names, x, mask, args["dim_head"], args["agent_size"], args["window_size"])
        else:
            names = [base + "block.1.", base + "block.2.", base + "block.5.", base + "block.6."]
            x = swap_fusion_block(sd, names, x, None, args["dim_head"], args["agent_size"], args["window_size"])
    y = x.mean(dim=1).permute(0, 2, 3, 1)                                             # b h w d
    y = F.layer_norm(y, (y.shape[-1],), sd[pfx + "mlp_head.2.weight"], sd[pfx + "mlp_head.2.bias"], 1e-5)
    y = F.linear(y, sd[pfx + "mlp_head.3.weight"], sd[pfx + "mlp_head.3.bias"])
    return y.permute(0, 3, 1, 2)
