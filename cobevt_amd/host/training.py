"""train() forwards of the FAX / FuseBEVT transformer blocks (first slice).

The reference's modules are ordinary nn.Modules trained by torch autograd (train_camera.py:143-179).  The inference forwards
of this package run fused bf16 launches over re-laid-out weights, which autograd cannot see through; in train() mode the
blocks below run this fp32 graph instead: LayerNorm, GELU and the gathered attention core are HIP kernels with HIP backward
kernels (cobevt_amd/autograd.py), the dense projections are library GEMMs, dropout is torch's.  Parameters stay the module's
own nn.Parameter containers, so optimizers, state_dicts and the gradient all-reduce (cobevt_amd.dist.GradAllReducer) see the
reference's names.

Covered: swap Attention / PreNormResidual / FeedForward / SwapFusionBlock(Mask) / SwapFusionEncoder
(swap_fusion_modules.py:13-286, base_transformer.py:102-124), FAX CrossWinAttention, CrossViewSwapAttention and the global
Attention (fax_modules.py:93-441).  The 3x3-convolutional parts (encoders, decoder, Bottlenecks, down-sampling blocks) have no
backward kernels yet: their modules keep raising in train() mode.
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


def _project(seq, t):
    """nn.Sequential(LayerNorm, Linear) of a cross attention on (..., d) -> (rows, inner)"""
    t = t.contiguous()
    return ag.linear(ag.layernorm(t, seq[0]).reshape(-1, t.shape[-1]), seq[1])


def cross_win_attend(m, q_src, k_src, v_src, qmap, kmap, batch, skip):
    """CrossWinAttention (fax_modules.py:198-248) on token-major sources whose rows the maps address: q_src (b, n, .., d) with
    n = qmap[1] cameras, k_src / v_src (b, nk, .., d); skip (b, .., d) | None in the layout of ONE camera of q_src.  Every
    camera's queries see all cameras' keys; the camera mean (:243) is taken after the projection (:240), outside the kernel, so
    that autograd sees it.  Returns (b, .., dim) in q_src's single-camera layout."""
    n = qmap[1]
    qt, kt, vt = _project(m.to_q, q_src), _project(m.to_k, k_src), _project(m.to_v, v_src)
    a = ag.window_attention(qt, kt, vt, qmap, kmap, qmap, batch, m.heads, m.scale, qt.shape[0])
    z = ag.linear(a, m.proj).reshape((batch, n) + tuple(q_src.shape[2:-1]) + (-1,)).mean(dim=1)
    return z + skip if skip is not None else z


def cross_win_attention(m, q, k, v, skip):
    """CrossWinAttention.forward (fax_modules.py:194-248): q (b n X Y W1 W2 d), k, v (b n x y w1 w2 d), skip (b X Y W1 W2 d)."""
    _check(q, k, v, skip)
    assert k.shape == v.shape
    b, n, X, Y, W1, W2, d = q.shape
    _, nk, kx, ky, w1, w2, _ = k.shape
    assert X * Y == kx * ky
    qmap = (2, n, X * W1, Y * W2, W1, W2, X, Y)
    kmap = (2, nk, kx * w1, ky * w2, w1, w2, kx, ky)
    return cross_win_attend(m, q, k, v, qmap, kmap, b, skip)


def _mlp(x, prenorm, mlp):
    """x + Linear(GELU(Linear(LayerNorm(x))))  (fax_modules.py:411,435)"""
    return x + ag.linear(ag.gelu(ag.linear(ag.layernorm(x.contiguous(), prenorm), mlp[0])), mlp[2])


def _pre_act_conv1x1(seq, x):
    """nn.Sequential(BatchNorm2d, ReLU, Conv2d 1x1) (fax_modules.py:281-292).  The BatchNorm follows ITS OWN .training flag
    (batch statistics + running-stat update, or the frozen running statistics), as torch would; both and the 1x1 convolution
    are library calls - there is no hot kernel here."""
    F = torch.nn.functional
    bn = seq[0]
    y = F.batch_norm(x, bn.running_mean, bn.running_var, bn.weight, bn.bias, bn.training, bn.momentum, bn.eps)
    return F.conv2d(F.relu(y), seq[2].weight, seq[2].bias)


def cross_view_swap_attention(m, index, x, bev, feature, I_inv, E_inv):
    """CrossViewSwapAttention.forward (fax_modules.py:323-441) as a differentiable graph: x (b d H W), feature (b n C h w),
    I_inv (b n 3 3), E_inv (b n 4 4) -> (b d H W).  The camera-geometry embeddings and the BN -> ReLU -> 1x1 projections are
    small library ops; both cross attentions (gathered window / grid attention), the LayerNorms and the GELUs are the HIP
    kernels with their HIP backward."""
    _check(x, feature, I_inv, E_inv)
    F = torch.nn.functional
    b, n, _, h, w = feature.shape
    _, d, H, W = x.shape
    W1, W2 = m.q_win_size
    w1, w2 = m.feat_win_size
    pixel = m.image_plane.reshape(1, 1, 3, h * w)
    c = E_inv[..., -1:]
    c_embed = F.conv2d(c.reshape(b * n, 4, 1, 1), m.cam_embed.weight)                   # (bn) d 1 1
    cam = F.pad(I_inv @ pixel, (0, 0, 0, 1), value=1)                                   # b n 4 hw
    dd = (E_inv @ cam).reshape(b * n, 4, h, w)
    img_embed = F.conv2d(dd, m.img_embed.weight) - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    if m.bev_embed_flag:
        grid = getattr(bev, "grid%d" % index)
        bev_embed = F.conv2d(grid[:2][None], m.bev_embed.weight, m.bev_embed.bias) - c_embed
        bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
        query = bev_embed.reshape(b, n, d, H, W) + x[:, None]
    else:
        query = x[:, None]
    feat = feature.reshape(b * n, -1, h, w)
    key = img_embed if m.feature_proj is None else img_embed + _pre_act_conv1x1(m.feature_proj, feat)
    val = _pre_act_conv1x1(m.feature_linear, feat)
    hp, wp = m._padded_hw(h, w)
    if (hp, wp) != (h, w):
        key, val = F.pad(key, (0, wp - w, 0, hp - h)), F.pad(val, (0, wp - w, 0, hp - h))
    # channels-last token matrices; the window / grid partitions are index arithmetic inside the attention kernels
    key_l = key.reshape(b, n, d, hp, wp).permute(0, 1, 3, 4, 2).contiguous()
    val_l = val.reshape(b, n, d, hp, wp).permute(0, 1, 3, 4, 2).contiguous()
    query_l = query.permute(0, 1, 3, 4, 2).contiguous()
    x_l = x.permute(0, 2, 3, 1).contiguous()
    nq = query_l.shape[1]
    qmap = ops.tokmap(0, nq, H, W, W1, W2)
    kwin, kgrid = ops.tokmap(0, n, hp, wp, w1, w2), ops.tokmap(1, n, hp, wp, w1, w2)
    if qmap[6] * qmap[7] != kwin[6] * kwin[7]:
        raise CobevtHipError("query windows %dx%d != key windows %dx%d" % (qmap[6], qmap[7], kwin[6], kwin[7]))
    out = cross_win_attend(m.cross_win_attend_1, query_l, key_l, val_l, qmap, kwin, b, x_l if m.skip else None)
    out = _mlp(out, m.prenorm_1, m.mlp_1)
    # local-to-global: the reference repeats the query over the n cameras (:417); the copies are identical, one is enough
    q2 = out[:, None]
    out = cross_win_attend(m.cross_win_attend_2, q2, key_l, val_l, ops.tokmap(0, 1, H, W, W1, W2), kgrid, b,
                           out if m.skip else None)
    out = _mlp(out, m.prenorm_2, m.mlp_2)
    out = ag.layernorm(out.contiguous(), m.postnorm)
    return out.permute(0, 3, 1, 2)


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
