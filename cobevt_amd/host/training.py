"""train() forwards of the FAX / FuseBEVT transformer blocks (first slice).

The reference's modules are ordinary nn.Modules trained by torch autograd (train_camera.py:143-179).  The inference forwards
of this package run fused bf16 launches over re-laid-out weights, which autograd cannot see through; in train() mode the
blocks below run this fp32 graph instead: LayerNorm, GELU and the gathered attention core are HIP kernels with HIP backward
kernels (cobevt_amd/autograd.py), the dense projections are library GEMMs, dropout is torch's.  Parameters stay the module's
own nn.Parameter containers, so optimizers, state_dicts and the gradient all-reduce (cobevt_amd.dist.GradAllReducer) see the
reference's names.

Covered: swap Attention / PreNormResidual / FeedForward / SwapFusionBlock(Mask) / SwapFusionEncoder
(swap_fusion_modules.py:13-286, base_transformer.py:102-124), FAX CrossWinAttention and the global Attention
(fax_modules.py:93-248).  The convolutional parts (encoders, decoder, Bottlenecks) have no backward kernels yet: their
modules keep raising in train() mode.
"""
import torch

from .. import autograd as ag
from .. import ops
from ..lib import CobevtHipError


def _check(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CobevtHipError("training forward needs ROCm device tensors; the HIP path has no CPU fallback")
        if t.dtype != torch.float32:
            raise CobevtHipError("the training slice is fp32 (got %s)" % t.dtype)


def _mask_f32(mask):
    if mask is None:
        return None
    mk = mask.to(torch.float32)
    return mk if mk.is_contiguous() else mk.contiguous()


def swap_attention(attn, x, mask, mode, norm=None):
    """swap_fusion_modules.Attention on x (b, l, H, W, d) (mode 0 window / 1 grid) or (b, l, X, Y, w1, w2, d) (mode 2);
    norm: the PreNormResidual LayerNorm (then the residual is added too).  swap_fusion_modules.py:87-128."""
    _check(x)
    L, w = attn.window_size[0], attn.window_size[1]
    if mode == 2:
        b, l, X, Y, w1, w2, d = x.shape
        m = (2, l, X * w1, Y * w2, w1, w2, X, Y)
    else:
        b, l, H, W, d = x.shape
        w1 = w2 = w
        m = ops.tokmap(mode, l, H, W, w, w)
    if l != L or w1 != w or w2 != w:
        raise CobevtHipError("swap attention built for %d agents x %dx%d windows, got %d x %dx%d" % (L, w, w, l, w1, w2))
    x = x.contiguous()
    rows = x.numel() // d
    xn = ag.layernorm(x, norm) if norm is not None else x
    qkv = ag.linear(xn.reshape(rows, d), attn.to_qkv)
    a = ag.window_attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], m, m, m, b, attn.heads, attn.scale, rows,
                            bias_table=attn.relative_position_bias_table.weight, bias_L=L, mask=_mask_f32(mask))
    y = ag.dropout(ag.linear(a, attn.to_out[0]), attn.to_out[1].p).reshape(x.shape)
    return y + x if norm is not None else y


def feed_forward(ffd, x, norm=None):
    """base_transformer.FeedForward (:112-124) on (..., d), with the PreNormResidual wrapper when `norm` is given."""
    _check(x)
    xn = ag.layernorm(x.contiguous(), norm) if norm is not None else x
    h = ag.dropout(ag.gelu(ag.linear(xn, ffd.net[0])), ffd.net[2].p)
    y = ag.dropout(ag.linear(h, ffd.net[3]), ffd.net[4].p)
    return y + x if norm is not None else y


def run_stages(stages, x, mask_of):
    for i, (ar, fr, mode) in enumerate(stages):
        x = swap_attention(ar.fn, x, mask_of(i), mode, norm=ar.norm)
        x = feed_forward(fr.fn, x, norm=fr.norm)
    return x


def to_blhwc(x):
    """(b, l, c, h, w) fp32 -> contiguous (b, l, h, w, c) (differentiable)"""
    return x.permute(0, 1, 3, 4, 2).contiguous()


def swap_fusion_block(block, x, mask):
    _check(x)
    y = run_stages(block.stages(), to_blhwc(x), lambda i: mask if block.uses_mask else None)
    return y.permute(0, 1, 4, 2, 3)


def swap_fusion_encoder(enc, x, mask):
    """SwapFusionEncoder.forward (:266-286): x (b, m, d, h, w), mask (b, h, w, 1, m) | None -> (b, d, h, w)."""
    _check(x)
    y = to_blhwc(x)
    for layer in enc.layers:
        y = run_stages(layer.stages(), y, lambda i: mask if layer.uses_mask else None)
    y = y.mean(dim=1)                                                        # Reduce('b m d h w -> b d h w', 'mean')
    y = ag.linear(ag.layernorm(y.contiguous(), enc.mlp_head[2]), enc.mlp_head[3])
    return y.permute(0, 3, 1, 2)


def cross_win_attention(m, q, k, v, skip):
    """CrossWinAttention.forward (fax_modules.py:194-248): q (b n X Y W1 W2 d), k, v (b n x y w1 w2 d), skip (b X Y W1 W2 d)."""
    _check(q, k, v, skip)
    assert k.shape == v.shape
    b, n, X, Y, W1, W2, d = q.shape
    _, nk, kx, ky, w1, w2, _ = k.shape
    assert X * Y == kx * ky
    inner = m.heads * m.dim_head
    qmap = (2, n, X * W1, Y * W2, W1, W2, X, Y)
    kmap = (2, nk, kx * w1, ky * w2, w1, w2, kx, ky)

    def project(seq, t):
        t = t.contiguous()
        return ag.linear(ag.layernorm(t, seq[0]).reshape(-1, t.shape[-1]), seq[1])

    qt, kt, vt = project(m.to_q, q), project(m.to_k, k), project(m.to_v, v)
    # every camera's queries against all cameras' keys; the camera mean (:243) is taken after the projection (:240), outside
    # the kernel, so that autograd sees it
    a = ag.window_attention(qt, kt, vt, qmap, kmap, qmap, b, m.heads, m.scale, qt.shape[0])
    z = ag.linear(a, m.proj).reshape(b, n, X, Y, W1, W2, -1).mean(dim=1)
    return z + skip if skip is not None else z


def global_attention(m, x):
    """FAX Attention.forward (fax_modules.py:137-176): x (b, d, h, w) -> (b, d, h, w)."""
    _check(x)
    b, d, h, w = x.shape
    if h != m.window_size or w != m.window_size:
        raise CobevtHipError("FAX global attention expects a %dx%d map" % (m.window_size, m.window_size))
    t = x.permute(0, 2, 3, 1).contiguous()
    rows = b * h * w
    qkv = ag.linear(t.reshape(rows, d), m.to_qkv)
    tm = ops.tokmap(0, 1, h, w, h, w)
    a = ag.window_attention(qkv[:, :d], qkv[:, d:2 * d], qkv[:, 2 * d:], tm, tm, tm, b, m.heads, m.scale, rows,
                            bias_table=m.rel_pos_bias.weight, bias_L=1)
    # nn.Dropout on the probabilities (attend[1]) has no fused counterpart: only p = 0 is supported in train mode
    if m.attend[1].p > 0:
        raise CobevtHipError("FAX global attention: dropout on the attention probabilities is not built (set dropout = 0)")
    y = ag.dropout(ag.linear(a, m.to_out[0]), m.to_out[1].p)
    return y.reshape(b, h, w, d).permute(0, 3, 1, 2)
