"""Oracle for the CVT per-agent encoder and the baseline fusion models built on it.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows opv2v/opencood/models/sub_modules/cvt_modules.py (BEVEmbedding :41-90, CrossAttention :93-170, CrossViewAttention
:173-283, CrossViewModule :286-327) and the model files cross_view_transformer.py:14-51 (single agent),
cross_view_transformer_swap_fuse.py:63-131 (CVT + swap fusion), cross_view_transformer_fcooper.py:62-129 (CVT + F-Cooper
max-out, fusion_modules/f_cooper_fuse.py:30-36), cross_view_transformer_att_fuse.py:62-131 (CVT + per-pixel agent attention:
base_transformer.py CavAttention :127-172, BaseEncoder :321-339, BaseTransformer :342-362).  Plain torch, functional over a flat
state_dict.
"""
import torch
import torch.nn.functional as F

from .corpbevt import bev_seg_head, naive_decoder
from .fax import _lin, _ln, _pre_act_conv1x1, generate_grid, get_view_matrix, image_plane
from .resnet import bottleneck_forward, resnet_encoder
from .sttf import regroup, roi_and_cav_mask, sttf
from .swap_fusion import swap_fusion_encoder


def bev_grid(bev_height, bev_width, h_meters, w_meters, offset, decoder_blocks, **_unused):
    """cvt_modules.py:63-83 — (3, h, w) ego-frame coordinates of the BEV cells; each decoder block doubles the map."""
    h, w = bev_height // (2 ** len(decoder_blocks)), bev_width // (2 ** len(decoder_blocks))
    g = generate_grid(h, w)
    g[0] = bev_width * g[0]
    g[1] = bev_height * g[1]
    V_inv = torch.FloatTensor(get_view_matrix(bev_height, bev_width, h_meters, w_meters, offset)).inverse()
    return (V_inv @ g.reshape(3, h * w)).reshape(3, h, w)


def cross_attention(sd, pfx, q, k, v, skip, heads, dim_head):
    """CrossAttention.forward, cvt_modules.py:116-170.  q (b n d H W), k, v (b n d h w), skip (b d H W) | None -> (b d H W).
    Per-camera queries against their own camera's keys, ONE softmax over the keys of all cameras (:148-150)."""
    b, n, d, H, W = q.shape
    scale = dim_head ** -0.5
    q = q.permute(0, 1, 3, 4, 2).reshape(b, n, H * W, d)
    k = k.permute(0, 1, 3, 4, 2).reshape(b, n, -1, d)
    v = v.permute(0, 1, 3, 4, 2).reshape(b, -1, d)                                  # b (n h w) d
    q = _lin(_ln(q, sd, pfx + "to_q.0"), sd, pfx + "to_q.1")
    k = _lin(_ln(k, sd, pfx + "to_k.0"), sd, pfx + "to_k.1")
    v = _lin(_ln(v, sd, pfx + "to_v.0"), sd, pfx + "to_v.1")
    q = q.reshape(b, n, H * W, heads, dim_head).permute(0, 3, 1, 2, 4)              # b m n Q dh
    k = k.reshape(b, n, -1, heads, dim_head).permute(0, 3, 1, 2, 4)                 # b m n K dh
    v = v.reshape(b, -1, heads, dim_head).permute(0, 2, 1, 3)                       # b m (n K) dh
    dot = scale * torch.matmul(q, k.transpose(-1, -2))                              # b m n Q K
    dot = dot.permute(0, 1, 3, 2, 4).reshape(b, heads, H * W, -1)                   # 'b n Q K -> b Q (n K)'
    att = dot.softmax(dim=-1)
    a = torch.matmul(att, v)                                                        # b m Q dh
    a = a.permute(0, 2, 1, 3).reshape(b, H * W, heads * dim_head)
    z = _lin(a, sd, pfx + "proj")
    if skip is not None:
        z = z + skip.permute(0, 2, 3, 1).reshape(b, H * W, d)
    z = _ln(z, sd, pfx + "prenorm")
    z = z + _lin(F.gelu(_lin(z, sd, pfx + "mlp.0")), sd, pfx + "mlp.2")
    z = _ln(z, sd, pfx + "postnorm")
    return z.reshape(b, H, W, d).permute(0, 3, 1, 2)


def cross_view_attention(sd, pfx, cfg, x, grid, feature, I_inv, E_inv):
    """CrossViewAttention.forward, cvt_modules.py:217-283.  x (b d H W); grid (3 H W); feature (b n C h w)."""
    b, n, _, h, w = feature.shape
    _, d, H, W = x.shape
    pixel = image_plane(h, w, cfg["image_height"], cfg["image_width"])
    c = E_inv[..., -1:]
    c_embed = F.conv2d(c.reshape(b * n, 4, 1, 1), sd[pfx + "cam_embed.weight"])
    cam = I_inv @ pixel.reshape(1, 1, 3, h * w)
    cam = F.pad(cam, (0, 0, 0, 1), value=1)
    dd = (E_inv @ cam).reshape(b * n, 4, h, w)
    d_embed = F.conv2d(dd, sd[pfx + "img_embed.weight"])
    img_embed = d_embed - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)
    w_embed = F.conv2d(grid[:2][None], sd[pfx + "bev_embed.weight"], sd[pfx + "bev_embed.bias"])
    bev_embed = w_embed - c_embed
    bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
    query = bev_embed.reshape(b, n, d, H, W) + x[:, None]
    feature_flat = feature.reshape(b * n, -1, h, w)
    if cfg.get("no_image_features", False):
        key_flat = img_embed
    else:
        key_flat = img_embed + _pre_act_conv1x1(feature_flat, sd, pfx + "feature_proj")
    val_flat = _pre_act_conv1x1(feature_flat, sd, pfx + "feature_linear")
    return cross_attention(sd, pfx + "cross_attend.", query, key_flat.reshape(b, n, d, h, w), val_flat.reshape(b, n, d, h, w),
                           x if cfg.get("skip", True) else None, cfg["heads"], cfg["dim_head"])


def cross_view_module(sd, pfx, cfg, features, intrinsic, extrinsic):
    """CrossViewModule.forward, cvt_modules.py:311-327.  features: list of (b l n C h w) -> (b l d H W); the extrinsics are
    used un-inverted (:316-317) like in the FAX module."""
    b, l, n = features[0].shape[:3]
    I_inv = intrinsic.reshape(b * l, n, 3, 3).inverse()
    E_inv = extrinsic.reshape(b * l, n, 4, 4)
    grid = bev_grid(**cfg["bev_embedding"])
    x = sd[pfx + "bev_embedding.learned_features"]
    x = x[None].expand(b * l, *x.shape)
    for i, feature in enumerate(features):
        feature = feature.reshape(b * l, n, *feature.shape[3:])
        x = cross_view_attention(sd, pfx + "cross_views.%d." % i, cfg["cross_view"], x, grid, feature, I_inv, E_inv)
        for j in range(cfg["middle"][i]):
            x = bottleneck_forward(sd, pfx + "layers.%d.%d." % (i, j), x)
    return x.reshape(b, l, *x.shape[1:])


def encode_agents(sd, config, batch):
    feats = resnet_encoder(sd, "encoder.encoder.", config["encoder"], batch["inputs"])
    return cross_view_module(sd, "cvm.", config["cvm"], feats, batch["intrinsic"], batch["extrinsic"])


def cross_view_transformer_forward(sd, config, batch):
    """CrossViewTransformer.forward, cross_view_transformer.py:36-51 (single-agent / late-fusion CVT baseline)."""
    b, l = batch["inputs"].shape[:2]
    f = encode_agents(sd, config, batch)
    y = naive_decoder(sd, "decoder.", config["decoder"], f)
    return bev_seg_head(sd, "seg_head.", config["target"], y.reshape(-1, *y.shape[2:]), b, l)


def _warp(sd, config, batch):
    f = encode_agents(sd, config, batch).squeeze(1)
    tm, record_len = batch["transformation_matrix"], batch["record_len"]
    g, mask = regroup(f, record_len, config["max_cav"])
    st = config["sttf"]
    w = sttf(g, tm, st["resolution"], st["downsample_rate"])                        # b l h w c
    if st["use_roi_mask"]:
        com_mask = roi_and_cav_mask(w.shape, mask, tm, st["resolution"], st["downsample_rate"])
    else:
        com_mask = mask[:, None, None, None, :].to(w.dtype)
    return w, com_mask


def _decode(sd, config, fused):
    """fused (b, c, h, w) -> output dict"""
    y = naive_decoder(sd, "decoder.", config["decoder"], fused[:, None])
    yb = y.reshape(-1, *y.shape[2:])
    return bev_seg_head(sd, "seg_head.", config["target"], yb, yb.shape[0], 1)


def cross_view_transformer_swap_fuse_forward(sd, config, batch):
    """CrossViewTransformerSwapFuse.forward, cross_view_transformer_swap_fuse.py:92-131."""
    w, com_mask = _warp(sd, config, batch)
    fused = swap_fusion_encoder(sd, "fusion_net.", config["swap_fusion"], w.permute(0, 1, 4, 2, 3), com_mask)
    return _decode(sd, config, fused)


def cross_view_transformer_fcooper_forward(sd, config, batch):
    """CrossViewTransformerFcooper.forward, cross_view_transformer_fcooper.py:93-129: max over the max_cav slots (zero-padded
    agents take part: SpatialFusionMask ignores the mask, f_cooper_fuse.py:30-36)."""
    w, _ = _warp(sd, config, batch)
    fused = w.max(dim=1)[0].permute(0, 3, 1, 2)
    return _decode(sd, config, fused)


def cav_attention(sd, pfx, x, mask, heads):
    """CavAttention.forward, base_transformer.py:143-172: at every pixel, attention over the agents.  x (b l h w c) already
    LayerNorm'ed; mask (b h w 1 l) (0 = agent not visible at that pixel, as a KEY) -> (b l h w c)."""
    b, l, h, w, c = x.shape
    t = x.permute(0, 2, 3, 1, 4)                                                    # b h w l c
    q, k, v = F.linear(t, sd[pfx + "to_qkv.weight"]).chunk(3, dim=-1)
    inner = q.shape[-1]
    dh = inner // heads
    scale = dh ** -0.5
    split = lambda z: z.reshape(b, h, w, l, heads, dh).permute(0, 4, 1, 2, 3, 5)    # b m h w l c
    q, k, v = split(q), split(k), split(v)
    att = torch.matmul(q, k.transpose(-1, -2)) * scale                              # b m h w i j
    att = att.masked_fill(mask.unsqueeze(1) == 0, -float("inf"))                    # (b 1 h w 1 l)
    att = att.softmax(dim=-1)
    out = torch.matmul(att, v)                                                      # b m h w l c
    out = out.permute(0, 2, 3, 4, 1, 5).reshape(b, h, w, l, inner)
    out = F.linear(out, sd[pfx + "to_out.0.weight"], sd[pfx + "to_out.0.bias"])
    return out.permute(0, 3, 1, 2, 4)


def base_transformer(sd, pfx, args, x, mask):
    """BaseTransformer.forward (:357-362) over BaseEncoder (:335-339): depth x [PreNorm(CavAttention) + x, PreNorm(FeedForward) + x],
    then the ego agent's map.  x (b l h w c), mask (b h w 1 l) -> (b h w c)."""
    for i in range(args["depth"]):
        a, f = "%sencoder.layers.%d.0." % (pfx, i), "%sencoder.layers.%d.1." % (pfx, i)
        x = cav_attention(sd, a + "fn.", _ln(x, sd, a + "norm"), mask, args["heads"]) + x
        y = _ln(x, sd, f + "norm")
        x = _lin(F.gelu(_lin(y, sd, f + "fn.net.0")), sd, f + "fn.net.3") + x
    return x[:, 0]


def cross_view_transformer_att_fuse_forward(sd, config, batch):
    """CrossViewTransformerAttFuse.forward, cross_view_transformer_att_fuse.py:96-131."""
    w, com_mask = _warp(sd, config, batch)
    fused = base_transformer(sd, "fusion_net.", config["base_transformer"], w, com_mask)      # b h w c
    return _decode(sd, config, fused.permute(0, 3, 1, 2))
