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

_F = torch.nn.functional


def _check(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise CobevtHipError("training forward needs ROCm device tensors; the HIP path has no CPU fallback")
        # inside an autocast region the torch ops between the HIP Functions produce half tensors; the Functions cast them back
        if t.dtype != torch.float32 and not (torch.is_autocast_enabled() and t.dtype in (torch.float16, torch.bfloat16)):
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
    xn = ag.layernorm(x, norm, for_projection=True) if norm is not None else x
    qkv = ag.linear(xn.reshape(rows, d), attn.to_qkv)
    a = ag.window_self_attention(qkv, m, b, attn.heads, attn.scale, rows, bias_table=attn.relative_position_bias_table.weight, bias_L=L,
                                 mask=_mask_f32(mask))
    y = ag.dropout(ag.linear(a, attn.to_out[0]), attn.to_out[1].p).reshape(x.shape)
    return y + x if norm is not None else y


def feed_forward(ffd, x, norm=None):
    """base_transformer.FeedForward (:112-124) on (..., d), with the PreNormResidual wrapper when `norm` is given."""
    _check(x)
    xn = ag.layernorm(x.contiguous(), norm, for_projection=True) if norm is not None else x
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
    y = ag.linear(ag.layernorm(y.contiguous(), enc.mlp_head[2], for_projection=True), enc.mlp_head[3])
    return y.permute(0, 3, 1, 2)


def _project(seq, t):
    """nn.Sequential(LayerNorm, Linear) of a cross attention on (..., d) -> (rows, inner)"""
    t = t.contiguous()
    return ag.linear(ag.layernorm(t, seq[0], for_projection=True).reshape(-1, t.shape[-1]), seq[1])


def cross_win_attend(m, q_src, k_src, v_src, qmap, kmap, batch, skip):
    """CrossWinAttention (fax_modules.py:198-248) on token-major sources whose rows the maps address: q_src (b, n, .., d) with
    n = qmap[1] cameras, k_src / v_src (b, nk, .., d); skip (b, .., d) | None in the layout of ONE camera of q_src.  Every
    camera's queries see all cameras' keys; the camera mean (:243) is taken after the projection (:240), outside the kernel, so
    that autograd sees it.  Returns (b, .., dim) in q_src's single-camera layout."""
    n = qmap[1]
    qt, kt, vt = _project(m.to_q, q_src), _project(m.to_k, k_src), _project(m.to_v, v_src)
    a = ag.window_attention(qt, kt, vt, qmap, kmap, qmap, batch, m.heads, m.scale, qt.shape[0])
    z = ag.group_mean(ag.linear(a, m.proj).reshape((batch, n) + tuple(q_src.shape[2:-1]) + (-1,)))
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
    return x + ag.linear(ag.gelu(ag.linear(ag.layernorm(x.contiguous(), prenorm, for_projection=True), mlp[0])), mlp[2])


def _pre_act_conv1x1(seq, x):
    """nn.Sequential(BatchNorm2d, ReLU, Conv2d 1x1) (fax_modules.py:281-292).  The BatchNorm follows ITS OWN .training flag
    (batch statistics + running-stat update, or the frozen running statistics), as torch would; the 1x1 convolution is the
    training conv (implicit-GEMM forward / input gradient, cobevt_conv_wgrad); BN + ReLU is one HIP Function (csrc/train_glue.hip)."""
    return ag.conv2d(ag.batch_norm_act(x, seq[0], relu=True), seq[2])


# the 2- / 4-channel geometry embeddings are 1x1 convolutions over the channel axis: the package's own implicit-GEMM kernel (forward, input
# gradient) and weight-gradient kernel through ag.linear - K = 4 is one 16-byte fp32 chunk, the 2-channel BEV grid is zero-padded to 4.
# The 3x3 / 4x4 camera-matrix products in front of them are broadcast multiply-adds (no trainable operand, no library GEMM).
def _pointwise(t, conv):
    k = t.shape[1]
    rows = t.permute(0, 2, 3, 1).float()                                            # (N, h, w, k) channels-last rows
    w2 = conv.weight.reshape(conv.weight.shape[0], k)
    if k % 4:
        rows, w2 = _F.pad(rows, (0, 4 - k % 4)), _F.pad(w2, (0, 4 - k % 4))
    y = ag.linear_weight(rows, w2, conv.bias)
    return y.permute(0, 3, 1, 2)


def _matmul_small(a, bmat):                                                          # (..., r, k) @ (..., k, m) with k <= 4
    return (a[..., :, :, None] * bmat[..., None, :, :]).sum(-2)


def cross_view_swap_attention(m, index, x, bev, feature, I_inv, E_inv):
    """CrossViewSwapAttention.forward (fax_modules.py:323-441) as a differentiable graph: x (b d H W), feature (b n C h w),
    I_inv (b n 3 3), E_inv (b n 4 4) -> (b d H W).  The camera-geometry embeddings (1 x 1 convolutions of 2 / 4 channels) and the
    BN -> ReLU -> 1x1 projections run the training convolution kernels; both cross attentions (gathered window / grid attention), the
    LayerNorms and the GELUs are the HIP kernels with their HIP backward; what is left to torch is elementwise (the L2 normalisation,
    the 3x3 / 4x4 camera-matrix products as broadcast multiply-adds, residual adds)."""
    _check(x, feature, I_inv, E_inv)
    F = torch.nn.functional
    b, n, _, h, w = feature.shape
    _, d, H, W = x.shape
    W1, W2 = m.q_win_size
    w1, w2 = m.feat_win_size
    pixel = m.image_plane.reshape(1, 1, 3, h * w)
    c = E_inv[..., -1:]
    c_embed = _pointwise(c.reshape(b * n, 4, 1, 1), m.cam_embed)                         # (bn) d 1 1
    cam = F.pad(_matmul_small(I_inv, pixel), (0, 0, 0, 1), value=1)                      # b n 4 hw
    dd = _matmul_small(E_inv, cam).reshape(b * n, 4, h, w)
    img_l = None
    if ag.USE_FAX_BEV_QUERY and ag.fax_img_embed_fusable(dd, m.img_embed) and d == 128:
        # the image embedding, normalised, channels-last: one kernel per direction (csrc/train_fax.hip); as an NCHW-shaped view below
        img_l = ag.fax_img_embed(dd, m.img_embed, c_embed.reshape(b * n, d))
        img_embed = img_l.permute(0, 3, 1, 2)
    else:
        img_embed = _pointwise(dd, m.img_embed) - c_embed
        img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    x_l = x.permute(0, 2, 3, 1).contiguous()
    query_l = None
    if m.bev_embed_flag:
        grid = getattr(bev, "grid%d" % index)
        if ag.USE_FAX_BEV_QUERY and ag.fax_bev_query_fusable(x_l, m.bev_embed, n):
            # embedding, normalisation, + x and the channels-last layout in one kernel per direction (csrc/train_fax.hip)
            query_l = ag.fax_bev_query(x_l, grid[:2], m.bev_embed, c_embed.reshape(b * n, d), n)
        else:
            bev_embed = _pointwise(grid[:2][None], m.bev_embed) - c_embed
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
    if query_l is None:
        query_l = query.permute(0, 1, 3, 4, 2).contiguous()
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
    # nn.Dropout on the probabilities (attend[1], fax_modules.py:114,161) happens inside the attention kernels
    a = ag.window_self_attention(qkv, tm, b, m.heads, m.scale, rows, bias_table=m.rel_pos_bias.weight, bias_L=1,
                                 drop_p=m.attend[1].p if m.attend[1].training else 0.0)
    y = ag.dropout(ag.linear(a, m.to_out[0]), m.to_out[1].p)
    return y.reshape(b, h, w, d).permute(0, 3, 1, 2)


# ----------------------------------------------------------------------------------------------
# the convolutional half: encoder, Bottlenecks, down-sampling, decoder, heads, STTF - and the whole model
# ----------------------------------------------------------------------------------------------
# Convolutions AND dense projections: the implicit-GEMM kernel forward and for the input gradient, cobevt_conv_wgrad(_blocked) for
# the weight gradient (cobevt_amd.autograd.Conv2dFn / linear).  BatchNorm (each container's own .training flag: batch statistics and
# running-stat updates, or the frozen statistics) fused with the residual add and the ReLU behind it, max-pooling, nearest
# up-sampling, PixelUnshuffle and the STTF warp are HIP kernels in both directions too (csrc/train_glue.hip).  What is left to
# torch: elementwise residual adds / means, layout permutes, dropout masks, the tiny camera-geometry embeddings and the optimiser.
# Tensors are (N, C, H, W)-shaped in channels-last memory.


def _conv_bn(x, conv, bn=None, relu=False, residual=None):
    """conv -> BatchNorm (its own .training flag) [-> + residual] [-> ReLU]; the BN / add / ReLU tail is ONE HIP Function"""
    y = ag.conv2d(x, conv)
    if bn is not None:
        return ag.batch_norm_act(y, bn, residual=residual, relu=relu)
    if residual is not None:
        y = y + residual
    return _F.relu(y) if relu else y


def basic_block(blk, x):
    """torchvision BasicBlock.forward as reached from resnet_ms.py:67-74"""
    identity = x if blk.downsample is None else _conv_bn(x, blk.downsample[0], blk.downsample[1])
    y = _conv_bn(x, blk.conv1, blk.bn1, relu=True)
    return _conv_bn(y, blk.conv2, blk.bn2, relu=True, residual=identity)


def resnet_encoder(enc, input_images):
    """ResnetEncoder.forward (resnet_ms.py:46-89): (B, L, M, H, W, 3) -> list of (B, L, M, C, h, w)"""
    _check(input_images)
    b, l, m, h, w, c = input_images.shape
    net = enc.encoder
    x = input_images.reshape(b * l * m, h, w, c).permute(0, 3, 1, 2)          # channels-last memory, no copy
    x = _conv_bn(x, net.conv1, net.bn1, relu=True)
    x = ag.max_pool3x3s2(x)
    results = []
    for layer in (net.layer1, net.layer2, net.layer3, net.layer4):
        for blk in layer:
            x = basic_block(blk, x)
        results.append(x.reshape(b, l, m, *x.shape[1:]))
    return [results[i] for i in enc.idx_pick] if isinstance(enc.idx_pick, list) else results[enc.idx_pick]


def bottleneck(blk, x):
    """torchvision Bottleneck(c, c // 4) (fax_modules.py:10,472)"""
    y = _conv_bn(x, blk.conv1, blk.bn1, relu=True)
    y = _conv_bn(y, blk.conv2, blk.bn2, relu=True)
    return _conv_bn(y, blk.conv3, blk.bn3, relu=True, residual=x)


def downsample(ds, x):
    """fax_modules.py:476-489: conv3x3 -> PixelUnshuffle(2) -> conv3x3 -> BN -> ReLU -> conv1x1 -> BN"""
    y = ag.pixel_unshuffle2(ag.conv2d(x, ds[0]))
    y = _conv_bn(y, ds[2], ds[3], relu=True)
    return _conv_bn(y, ds[5], ds[6])


def fax_module(m, batch):
    """FAXModule.forward (fax_modules.py:497-521): batch['features'] list of (b, l, n, C, h, w) -> (b, l, d, H, W)"""
    feats = batch["features"]
    b, l, n = feats[0].shape[:3]
    # (no gradient flows into the camera matrices; the device kernel has no host round trip, unlike torch.linalg.inv's error check)
    I_inv = ops.invert_small(batch["intrinsic"].reshape(b * l * n, 3, 3).to(torch.float32)).reshape(b * l, n, 3, 3)
    E_inv = m._extrinsic(batch["extrinsic"].reshape(b * l, n, 4, 4).to(torch.float32))
    prior = m.bev_embedding.get_prior()
    x = prior[None].expand(b * l, *prior.shape)
    for i, (cross_view, feature, layer) in enumerate(zip(m.cross_views, feats, m.layers)):
        feature = feature.reshape(b * l, n, *feature.shape[3:])
        x = cross_view_swap_attention(cross_view, i, x.contiguous(), m.bev_embedding, feature.contiguous(), I_inv, E_inv)
        for blk in layer:
            x = bottleneck(blk, x)
        if i < len(m.cross_views) - 1:
            x = downsample(m.downsample_layers[i][0], x)
    if m.self_attn is not None:
        x = global_attention(m.self_attn, x.contiguous())
    return x.reshape(b, l, *x.shape[1:])


def naive_decoder(dec, x):
    """NaiveDecoder.forward (naive_decoder.py:62-91) on (N, C, H, W)"""
    for i in range(dec.num_layer - 1, -1, -1):
        x = _conv_bn(x, dec.convs[("upconv", i, 0)], dec.convs[("norm", i, 0)], relu=True)
        x = ag.upsample_nearest2(x)
        x = _conv_bn(x, dec.convs[("upconv", i, 1)], dec.convs[("norm", i, 1)], relu=True)
    return x


def bev_seg_head(head, x, b, l):
    """BevSegHead.forward (bev_seg_head.py:35-61): x ((b l), C, H, W) -> both maps (b, l, classes, H, W)"""
    def run(conv):
        y = ag.conv2d(x, conv)
        return y.reshape(b, l, *y.shape[1:])
    if head.target == "dynamic":
        dyn = run(head.dynamic_head)
        return {"static_seg": torch.zeros_like(dyn), "dynamic_seg": dyn}
    if head.target == "static":
        sta = run(head.static_head)
        return {"static_seg": sta, "dynamic_seg": torch.zeros_like(sta)}
    return {"static_seg": run(head.static_head), "dynamic_seg": run(head.dynamic_head)}


def naive_compressor(comp, x):
    """NaiveCompressor.forward (naive_compress.py:22-31) on (N, C, H, W): three conv3x3 + BatchNorm + ReLU stages"""
    _check(x)
    x = _conv_bn(x, comp.encoder[0], comp.encoder[1], relu=True)
    x = _conv_bn(x, comp.decoder[0], comp.decoder[1], relu=True)
    return _conv_bn(x, comp.decoder[3], comp.decoder[4], relu=True)


def sttf_warp(x, tm, discrete_ratio, downsample_rate):
    """STTF.forward (corpbevt.py:28-64): x (B, L, C, H, W) -> (B, L, H, W, C) in the ego frame, differentiable in x: the inference
    warp kernel forward, its adjoint backward (cobevt_amd.autograd.SttfWarpFn).  The pose algebra and the sampling stay in fp32
    inside an autocast region (the reference needs an fp16-safe inverse and a `.half()` hack there,
    torch_transformation_utils.py:137-157,354)."""
    with torch.autocast("cuda", enabled=False):
        xl = x.float().permute(0, 1, 3, 4, 2).contiguous()
        return ag.sttf_warp(xl, tm.to(device=x.device, dtype=torch.float32).contiguous(), None, x.shape[1], discrete_ratio, downsample_rate)


def fuse_and_decode(model, f, transformation_matrix, record_len, record_len_host=None):
    """the cross-agent part of CorpBEVT.forward (corpbevt.py:119-145): f (N, C, H, W) per-agent BEV features.  record_len_host:
    the agent counts as Python ints when record_len lives on the device (a captured training step cannot read it back)"""
    if model.compression:                               # corpbevt.py:119-121
        f = naive_compressor(model.naive_compressor, f)
    dev = f.device
    tm = transformation_matrix.to(device=dev, dtype=torch.float32)
    rl = torch.as_tensor(record_len).to(device=dev, dtype=torch.int32)
    # regroup (fuse_utils.py:8-61) + warp in one HIP Function: no host read of record_len, adjoint kernel in backward
    with torch.autocast("cuda", enabled=False):
        w = ag.sttf_warp(f.float().permute(0, 2, 3, 1).contiguous(), tm.contiguous(), rl, model.max_cav, model.discrete_ratio,
                         model.downsample_rate)                                                   # b l h w c
    # the ROI / agent mask carries no gradient: the inference kernel computes it
    with torch.no_grad():
        _, com_mask, cav_mask = ops.sttf_warp(f.detach().float().permute(0, 2, 3, 1).contiguous(), tm.contiguous(), None, model.discrete_ratio,
                                              model.downsample_rate, want_mask=model.use_roi_mask, record_len=rl,
                                              max_cav=model.max_cav)
        if not model.use_roi_mask:
            b, l, h, ww, _ = w.shape
            com_mask = cav_mask[:, None, None, None, :].expand(b, h, ww, 1, l).contiguous()
    fused = swap_fusion_encoder(model.fusion_net, w.permute(0, 1, 4, 2, 3).contiguous(), com_mask)           # b c h w
    y = naive_decoder(model.decoder, fused)
    return bev_seg_head(model.seg_head, y, y.shape[0], 1)


def corpbevt(model, batch_dict):
    """CorpBEVT.forward (corpbevt.py:104-145) as a differentiable graph"""
    feats = resnet_encoder(model.encoder, batch_dict["inputs"])
    batch_dict.update({"features": feats})
    f = fax_module(model.fax, batch_dict).squeeze(1)
    return fuse_and_decode(model, f, batch_dict["transformation_matrix"], batch_dict["record_len"], batch_dict.get("record_len_host"))


# ----------------------------------------------------------------------------------------------
# the CVT baselines: cvt_modules.py CrossAttention / CrossViewAttention / CrossViewModule, the per-pixel agent
# attention of base_transformer.py, and the models cross_view_transformer{,_swap_fuse,_fcooper,_att_fuse}.py in train() mode
# ----------------------------------------------------------------------------------------------
def cvt_cross_attention(m, q, k, v, skip):
    """cvt_modules.CrossAttention.forward (:116-170) on token-major sources: q (b, n, Q, d), k / v (b, n, K, d), skip (b, Q, d) | None
    -> (b, Q, d).  Camera c's query copy scores camera c's keys and ONE softmax runs over the keys of all cameras (:142-153).
    With s_c the scores of camera c,   softmax over (c, K)  =  softmax_K(s_c) * softmax_c(lse_c),   lse_c = log sum_K exp(s_c):
    the attention kernels run once with the cameras as the windows of a stored-partitioned map (the per-camera outputs and their
    log-sum-exp), a softmax over the n log-sum-exps merges them - both differentiable (WindowAttentionFn takes the lse gradient)."""
    _check(q, k, v, skip)
    b, n, Q, d = q.shape
    K = k.shape[2]
    heads = m.heads
    qt, kt, vt = _project(m.to_q, q), _project(m.to_k, k), _project(m.to_v, v)
    # mode-2 maps with ncam = 1, X * Y = n windows, w1 * w2 tokens: row = (b * n + camera) * tokens + token
    qmap, kmap = (2, 1, n * Q, 1, Q, 1, n, 1), (2, 1, n * K, 1, K, 1, n, 1)
    if Q >= 256 or K >= 256:                 # a window side is < 256 in the token-coordinate packing: factor the token count
        def sides(t):
            for a in (128, 64, 32, 16, 8, 4, 2):
                if t % a == 0 and t // a < 256:
                    return t // a, a
            raise CobevtHipError("CVT cross attention: cannot factor %d tokens into a window below 256 x 256" % t)
        (q1, q2), (k1, k2) = sides(Q), sides(K)
        qmap, kmap = (2, 1, n * q1, q2, q1, q2, n, 1), (2, 1, n * k1, k2, k1, k2, n, 1)
    a, lse = ag.window_attention(qt, kt, vt, qmap, kmap, qmap, b, heads, m.scale, qt.shape[0], return_lse=True)   # (b n Q, inner), (b, n, heads, Q)
    wts = torch.softmax(lse, dim=1).permute(0, 1, 3, 2)                                     # (b, n, Q, heads)
    a = (a.reshape(b, n, Q, heads, 32) * wts[..., None]).sum(dim=1).reshape(b * Q, heads * 32)
    z = ag.linear(a, m.proj).reshape(b, Q, d)
    if skip is not None:
        z = z + skip
    z = ag.layernorm(z.contiguous(), m.prenorm)
    z = z + ag.linear(ag.gelu(ag.linear(z, m.mlp[0])), m.mlp[2])
    return ag.layernorm(z.contiguous(), m.postnorm)


def cvt_cross_view_attention(m, x, bev, feature, I_inv, E_inv):
    """cvt_modules.CrossViewAttention.forward (:217-283): x (b d H W), feature (b n C h w), I_inv (b n 3 3), E_inv (b n 4 4) -> (b d H W)"""
    _check(x, feature, I_inv, E_inv)
    b, n, _, h, w = feature.shape
    _, d, H, W = x.shape
    pixel = m.image_plane.reshape(1, 1, 3, h * w)
    c = E_inv[..., -1:]
    c_embed = _pointwise(c.reshape(b * n, 4, 1, 1), m.cam_embed)                            # (bn) d 1 1
    cam = _F.pad(_matmul_small(I_inv, pixel), (0, 0, 0, 1), value=1)                        # b n 4 hw
    dd = _matmul_small(E_inv, cam).reshape(b * n, 4, h, w)
    img_embed = _pointwise(dd, m.img_embed) - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    bev_embed = _pointwise(bev.grid[:2][None], m.bev_embed) - c_embed
    bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
    query = bev_embed.reshape(b, n, d, H, W) + x[:, None]
    feat = feature.reshape(b * n, -1, h, w)
    key = img_embed if m.feature_proj is None else img_embed + _pre_act_conv1x1(m.feature_proj, feat)
    val = _pre_act_conv1x1(m.feature_linear, feat)
    tok = lambda t, hh, ww: t.reshape(b, n, d, hh * ww).permute(0, 1, 3, 2).contiguous()       # (b, n, tokens, d)
    skip = x.reshape(b, d, H * W).permute(0, 2, 1) if m.skip else None
    z = cvt_cross_attention(m.cross_attend, tok(query, H, W), tok(key, h, w), tok(val, h, w), skip)
    return z.reshape(b, H, W, d).permute(0, 3, 1, 2)


def cvt_cross_view_module(m, batch):
    """cvt_modules.CrossViewModule.forward (:311-327): batch['features'] list of (b, l, n, C, h, w) -> (b, l, d, H, W)"""
    feats = batch["features"]
    b, l, n = feats[0].shape[:3]
    I_inv = ops.invert_small(batch["intrinsic"].reshape(b * l * n, 3, 3).to(torch.float32)).reshape(b * l, n, 3, 3)
    E_inv = batch["extrinsic"].reshape(b * l, n, 4, 4).to(torch.float32)                  # used un-inverted (:316-317)
    prior = m.bev_embedding.get_prior()
    x = prior[None].expand(b * l, *prior.shape)
    for cross_view, feature, layer in zip(m.cross_views, feats, m.layers):
        feature = feature.reshape(b * l, n, *feature.shape[3:])
        x = cvt_cross_view_attention(cross_view, x.contiguous(), m.bev_embedding, feature.contiguous(), I_inv, E_inv)
        for blk in layer:
            x = bottleneck(blk, x)
    return x.reshape(b, l, *x.shape[1:])


def cvt_encode_agents(model, batch_dict):
    """images -> (b, l, d, H, W) per-agent BEV features (cross_view_transformer.py:36-45)"""
    feats = resnet_encoder(model.encoder, batch_dict["inputs"])
    batch_dict.update({"features": feats})
    return cvt_cross_view_module(model.cvm, batch_dict)


def cross_view_transformer(model, batch_dict):
    """CrossViewTransformer.forward (cross_view_transformer.py:36-51) as a differentiable graph"""
    b, l = batch_dict["inputs"].shape[:2]
    f = cvt_encode_agents(model, batch_dict)
    y = naive_decoder(model.decoder, f.reshape(b * l, *f.shape[2:]))
    return bev_seg_head(model.seg_head, y, b, l)


def cav_attention(attn, x, mask, norm):
    """base_transformer.CavAttention (:127-172) behind its PreNorm, residual added: x (b, l, h, w, c), mask (b, h, w, 1, l) | None.  The
    gathered window attention with 1 x 1 windows: the tokens of a window are the l agents' features at that pixel."""
    _check(x)
    b, l, h, w, c = x.shape
    x = x.contiguous()
    rows = x.numel() // c
    inner = attn.heads * 32
    qkv = ag.linear(ag.layernorm(x, norm, for_projection=True).reshape(rows, c), attn.to_qkv)
    m = ops.tokmap(0, l, h, w, 1, 1)
    mk = None if mask is None else mask.to(torch.float32).expand(b, h, w, 1, l).reshape(b, h, w, l).contiguous()
    a = ag.window_self_attention(qkv, m, b, attn.heads, attn.scale, rows, mask=mk)
    y = ag.dropout(ag.linear(a, attn.to_out[0]), attn.to_out[1].p).reshape(x.shape)
    return y + x


def base_transformer(bt, x, mask):
    """BaseTransformer.forward (base_transformer.py:342-362): x (b, l, h, w, c), mask (b, h, w, 1, l) -> the ego agent's map (b, h, w, c)"""
    for attn, ff in bt.encoder.layers:
        x = cav_attention(attn.fn, x, mask, attn.norm)
        x = feed_forward(ff.fn, x, norm=ff.norm)
    return x[:, 0]


def cvt_fuse_and_decode(model, f, transformation_matrix, record_len):
    """the V2V tail of the CVT fusion baselines (cross_view_transformer_{swap_fuse,fcooper,att_fuse}.py): regroup + STTF warp (+ ROI
    mask), the model's fusion, NaiveDecoder, BevSegHead.  f (N, C, H, W) per-agent BEV features."""
    dev = f.device
    tm = transformation_matrix.to(device=dev, dtype=torch.float32)
    rl = torch.as_tensor(record_len).to(device=dev, dtype=torch.int32)
    with torch.autocast("cuda", enabled=False):
        fl = f.float().permute(0, 2, 3, 1).contiguous()
        w = ag.sttf_warp(fl, tm.contiguous(), rl, model.max_cav, model.discrete_ratio, model.downsample_rate)     # b l h w c
    with torch.no_grad():
        _, com_mask, cav_mask = ops.sttf_warp(fl.detach(), tm.contiguous(), None, model.discrete_ratio, model.downsample_rate,
                                              want_mask=model.use_roi_mask, record_len=rl, max_cav=model.max_cav)
        if not model.use_roi_mask:
            b, l, h, ww, _ = w.shape
            com_mask = cav_mask[:, None, None, None, :].expand(b, h, ww, 1, l).contiguous()
    fused = model._fuse_train(w, com_mask)                                                       # (b, c, h, w)
    y = naive_decoder(model.decoder, fused)
    return bev_seg_head(model.seg_head, y, y.shape[0], 1)


# ----------------------------------------------------------------------------------------------
# the nuScenes SinBEVT model (nuscenes/cross_view_transformer/model/{cvt,encoder_pyramid_axial,decoder}.py and
# backbones/efficientnet.py over efficientnet-pytorch's MBConvBlock) in train() mode
# ----------------------------------------------------------------------------------------------
def _drop_connect(x, p, training):
    """efficientnet_pytorch.utils.drop_connect: per-sample stochastic depth on the block's output before the identity skip"""
    if not training or not p:
        return x
    keep = 1.0 - p
    mask = torch.floor(keep + torch.rand((x.shape[0], 1, 1, 1), dtype=x.dtype, device=x.device))
    return x / keep * mask


def mbconv(blk, x, drop_rate):
    """efficientnet-pytorch MBConvBlock.forward (restated in oracle/efficientnet.py:mbconv; host mirror nuscenes/efficientnet.py): 1x1 expand
    + BN + swish, depthwise k x k ("same" static padding) + BN + swish, squeeze-and-excitation, 1x1 project + BN, drop-connect + identity
    skip.  Convolutions, BatchNorm, swish and the depthwise convolution are HIP kernels in both directions; the squeeze-and-excitation
    vector algebra ((N, C) tensors) and the gate multiply are elementwise torch ops."""
    inp = x
    if blk.expand != 1:
        x = ag.swish(ag.batch_norm_act(ag.conv2d(x, blk._expand_conv), blk._bn0))
    x = ag.depthwise_conv2d(x, blk._depthwise_conv, blk.pad)
    x = ag.swish(ag.batch_norm_act(x, blk._bn1))
    mid, sq = blk._se_reduce.in_channels, blk._se_reduce.out_channels
    s = x.float().mean(dim=(2, 3))                                                      # (N, mid)
    s = (s[:, None, :] * blk._se_reduce.weight.reshape(1, sq, mid)).sum(-1) + blk._se_reduce.bias
    s = s * torch.sigmoid(s)
    s = (s[:, None, :] * blk._se_expand.weight.reshape(1, mid, sq)).sum(-1) + blk._se_expand.bias
    x = x * torch.sigmoid(s).to(x.dtype)[:, :, None, None]
    x = ag.batch_norm_act(ag.conv2d(x, blk._project_conv), blk._bn2)
    if blk.stride == 1 and blk.cin == blk.cout:
        x = _drop_connect(x, drop_rate, blk.training) + inp
    return x


def efficientnet_extractor(m, x):
    """EfficientNetExtractor.forward (backbones/efficientnet.py:85-96): x (N, 3, H, W) normalised images -> the picked feature maps.  (The
    reference wraps every layer in torch.utils.checkpoint in training mode - a memory / recompute trade that does not change the function.)"""
    _check(x)
    stem = m.layers[0]
    p0, p1 = m._stem_pad
    y = ag.conv2d(_F.pad(x, (p0, p1, p0, p1)), stem[0])
    y = ag.swish(ag.batch_norm_act(y, stem[1]))
    result = [y]
    for group in list(m.layers)[1:]:
        for blk, a in zip(group, group.args):
            y = mbconv(blk, y, a[0] if a else 0.0)
        result.append(y)
    return [result[i] for i in m.idx_pick]


def pyramid_axial_encoder(m, batch):
    """PyramidAxialEncoder.forward (encoder_pyramid_axial.py:534-558): image (b, n, 3, h, w), intrinsics, extrinsics -> (b, d, H, W)"""
    image = batch["image"]
    _check(image, batch["intrinsics"], batch["extrinsics"])
    b, n = image.shape[:2]
    I_inv = ops.invert_small(batch["intrinsics"].reshape(b * n, 3, 3).to(torch.float32)).reshape(b, n, 3, 3)
    E_inv = ops.invert_small(batch["extrinsics"].reshape(b * n, 4, 4).to(torch.float32)).reshape(b, n, 4, 4)     # inverted in the model (:538-539)
    feats = efficientnet_extractor(m.backbone, (image.flatten(0, 1) - m.norm.mean) / m.norm.std)
    prior = m.bev_embedding.get_prior()
    x = prior[None].expand(b, *prior.shape)
    for i, (cross_view, feature, layer) in enumerate(zip(m.cross_views, feats, m.layers)):
        feature = feature.reshape(b, n, *feature.shape[1:])
        x = cross_view_swap_attention(cross_view, i, x.contiguous(), m.bev_embedding, feature.contiguous(), I_inv, E_inv)
        for blk in layer:
            x = bottleneck(blk, x)
        if i < len(m.cross_views) - 1:
            x = downsample(m.downsample_layers[i][0], x)
    return x


def nusc_decoder_block(blk, x, skip):
    """DecoderBlock.forward (decoder.py:27-36): bilinear x2 (align_corners) -> conv3x3 + BN + ReLU -> conv1x1 + BN, + the 1x1 projection of the
    decoder's INPUT map resized (nearest) to the new size, ReLU"""
    h, w = x.shape[2:]
    big = ag.resize_bilinear(x, 2 * h, 2 * w)
    hidden = _conv_bn(big, blk.conv[1], blk.conv[2], relu=True)
    branch = None
    if blk.up is not None:
        branch = ag.conv2d(skip, blk.up)
        f = (2 * h) // branch.shape[2]
        if branch.shape[2] * f == 2 * h and branch.shape[3] * f == 2 * w and f & (f - 1) == 0:
            while branch.shape[2] < 2 * h:               # nearest by an integer power of two = nearest x2, repeated
                branch = ag.upsample_nearest2(branch)
        else:
            branch = _F.interpolate(branch, (2 * h, 2 * w))
    return _conv_bn(hidden, blk.conv[4], blk.conv[5], relu=True, residual=branch)


def nusc_decoder(dec, x):
    y = x
    for blk in dec.layers:                               # every block's skip branch reads the decoder INPUT (decoder.py:55-61)
        y = nusc_decoder_block(blk, y, x)
    return y


def nusc_cross_view_transformer(model, batch):
    """nuScenes CrossViewTransformer.forward (cvt.py:35-40) as a differentiable graph"""
    bev = pyramid_axial_encoder(model.encoder, batch)
    y = nusc_decoder(model.decoder, bev)
    hidden = _conv_bn(y, model.to_logits[0], model.to_logits[1], relu=True)
    logits = ag.conv2d(hidden, model.to_logits[3])
    return {name: logits[:, lo:hi] for name, (lo, hi) in model.outputs.items()}


# ----------------------------------------------------------------------------------------------
# the pairwise-warp fusions of the V2VNet / DiscoNet baselines (fusion_modules/v2v_fuse.py:47-144, disconet_fuse.py:82-168) in train()
# mode.  As in the inference path the maps stay in their original orientation: the warp undoes the reference's transposition + flip,
# the 3x3 convolutions run with re-indexed taps W~[u][v] = W[v][2 - u] (a differentiable view of the parameter).
# ----------------------------------------------------------------------------------------------
def _flipped_taps(w):
    return w.transpose(2, 3).flip(2)


def _agent_index(rl, n):
    """sample b and slot i of each of the n un-grouped agent rows, on the device (no host round trip)"""
    rl64 = rl.to(torch.int64)
    ends = torch.cumsum(rl64, 0)
    a = torch.arange(n, device=rl.device)
    b_of = torch.searchsorted(ends, a, right=True)
    return b_of, a - (ends - rl64)[b_of]


def _pairwise_inputs(m, x, record_len, pairwise_t_matrix):
    rl = torch.as_tensor(record_len).to(device=x.device, dtype=torch.int32)
    pw = pairwise_t_matrix.to(device=x.device, dtype=torch.float32).contiguous()
    return rl, pw


def _warp_pairs(m, feats, pw, rl, l):
    """feats (N, C, H, W) -> nb (B, L, L, H, W, C) differentiable in feats, roi (B, L, L, H, W) (no gradient)"""
    with torch.autocast("cuda", enabled=False):
        fl = feats.float().permute(0, 2, 3, 1).contiguous()
        nb = ag.PairwiseWarpFn.apply(fl, pw, rl, l, m.discrete_ratio, m.downsample_rate)
        with torch.no_grad():
            _, roi = ops.pairwise_warp(fl.detach(), pw, rl, l, m.discrete_ratio, m.downsample_rate)
    return nb, roi


def v2vnet_fusion(m, x, record_len, pairwise_t_matrix):
    """V2VNetFusion.forward (v2v_fuse.py:47-144): x (sum(record_len), C, H, W) -> (B, H, W, C)"""
    _check(x)
    rl, pw = _pairwise_inputs(m, x, record_len, pairwise_t_matrix)
    b, l = pw.shape[:2]
    n, c, h, w = x.shape
    b_of, i_of = _agent_index(rl, n)
    valid = (torch.arange(l, device=x.device)[None, :] < rl[:, None])[b_of]                       # (n, l): slot j holds an agent
    count = rl[b_of].to(torch.float32)[:, None, None, None]
    wt = _flipped_taps(m.msg_cnn.weight)
    cell = m.conv_gru.cell_list[0]
    feats = x
    for _ in range(m.num_iteration):
        nb, roi = _warp_pairs(m, feats, pw, rl, l)
        # message = msg_cnn(cat[neighbour, ego]): the ego half (and the bias) is the same for every neighbour
        msg = ag.conv2d_weight(nb.reshape(b * l * l, h, w, c).permute(0, 3, 1, 2), wt[:, :c], None, 1, 1)
        ego = ag.conv2d_weight(feats, wt[:, c:], m.msg_cnn.bias, 1, 1)
        msg_a = msg.reshape(b, l, l, c, h, w)[b_of, i_of]                                            # (n, l, c, h, w)
        val = (msg_a + ego[:, None]) * roi[b_of, i_of][:, :, None]
        if m.agg_operator == "avg":
            agg = (val * valid[:, :, None, None, None]).sum(1) / count
        else:
            agg = val.masked_fill(~valid[:, :, None, None, None], float("-inf")).max(1)[0]
        if m.gru_flag:
            # one ConvGRU step from h = 0 (convgru.py:57-78): h' = sigmoid(update) * tanh(candidate), both from the 2C input channels only
            wg, wc = _flipped_taps(cell.conv_gates.weight), _flipped_taps(cell.conv_can.weight)
            gw = torch.cat([wg[c:2 * c, :2 * c], wc[:, :2 * c]], dim=0)
            gb = torch.cat([cell.conv_gates.bias[c:2 * c], cell.conv_can.bias], dim=0)
            g = ag.conv2d_weight(torch.cat([feats.float(), agg.float()], dim=1), gw, gb, 1, 1)
            feats = torch.sigmoid(g[:, :c]) * torch.tanh(g[:, c:])
        else:
            feats = feats + agg
    rl64 = rl.to(torch.int64)
    out = feats.index_select(0, torch.cumsum(rl64, 0) - rl64)
    return ag.linear(out.permute(0, 2, 3, 1), m.mlp)


def disconet_fusion(m, x, record_len, pairwise_t_matrix, record_len_host=None):
    """DiscoNetFusion.forward (disconet_fuse.py:82-168): x (sum(record_len), C, H, W) -> (B, H, W, C).  PixelWeightedFusionSoftmax runs once per
    target agent on its N_b neighbours, as in the reference - its BatchNorms then see the same batches (statistics over N_b x H x W, running
    statistics updated call by call) - which needs the agent counts on the host (the reference reads them there too, :98-104)."""
    _check(x)
    rl, pw = _pairwise_inputs(m, x, record_len, pairwise_t_matrix)
    lens = [int(v) for v in (record_len_host if record_len_host is not None else record_len)]
    b, l = pw.shape[:2]
    n, c, h, w = x.shape
    f = m.pixel_weighted_fusion
    feats = x
    for _ in range(m.num_iteration):
        nb, roi = _warp_pairs(m, feats, pw, rl, l)
        rows, a = [], 0
        for bi, nb_count in enumerate(lens):
            for i in range(nb_count):
                neigh = nb[bi, i, :nb_count].permute(0, 3, 1, 2)                                     # (N_b, c, h, w)
                mask = roi[bi, i, :nb_count][:, None]                                                 # (N_b, 1, h, w)
                y = torch.cat([neigh, feats[a][None].float().expand(nb_count, c, h, w)], dim=1)
                y = _conv_bn(y, f.conv1_1, f.bn1_1, relu=True)
                y = _conv_bn(y, f.conv1_2, f.bn1_2, relu=True)
                y = _conv_bn(y, f.conv1_3, f.bn1_3, relu=True)
                s = _F.relu(ag.conv2d(y, f.conv1_4)).float()
                if m.use_mask:
                    s = s.masked_fill(mask == 0, float("-inf"))
                rows.append((torch.softmax(s, dim=0) * neigh * mask).sum(0))
                a += 1
        feats = torch.stack(rows, 0)
    rl64 = rl.to(torch.int64)
    out = feats.index_select(0, torch.cumsum(rl64, 0) - rl64)
    return ag.linear(out.permute(0, 2, 3, 1), m.mlp)


def cvt_pairwise_model(model, batch_dict):
    """CrossViewTransformerV2VNet / DiscoNet .forward (cross_view_transformer_{v2vnet,disconet}.py:41-68) as a differentiable graph"""
    f = cvt_encode_agents(model, batch_dict).squeeze(1)
    fused = model._fuse_train(f, batch_dict["record_len"], batch_dict["pairwise_t_matrix"], batch_dict.get("record_len_host"))   # (B, H, W, C)
    y = naive_decoder(model.decoder, fused.permute(0, 3, 1, 2))
    return bev_seg_head(model.seg_head, y, y.shape[0], 1)
