"""Oracle for the FAX cross-view pyramid (SinBEVT core).  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows opv2v/opencood/models/sub_modules/fax_modules.py (OPV2V flavour) and, through the `flavour` switch,
nuscenes/cross_view_transformer/model/encoder_pyramid_axial.py (nuScenes flavour).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .resnet import bottleneck_forward, bn_eval


# ---------------------------------------------------------------------------------------------
# init-time buffers
# ---------------------------------------------------------------------------------------------
def generate_grid(height, width):
    """fax_modules.py:13-21 — (3, h, w): x in [0,1] along width, y in [0,1] along height, ones."""
    xs = torch.linspace(0, 1, width)
    ys = torch.linspace(0, 1, height)
    gx = xs[None, :].expand(height, width)
    gy = ys[:, None].expand(height, width)
    return torch.stack([gx, gy, torch.ones(height, width)], 0).contiguous()


def get_view_matrix(h=200, w=200, h_meters=100.0, w_meters=100.0, offset=0.0):
    """fax_modules.py:24-35"""
    sh = h / h_meters
    sw = w / w_meters
    return [[0., -sw, w / 2.], [-sh, 0., h * offset + h / 2.], [0., 0., 1.]]


def bev_grids(bev_height, bev_width, h_meters, w_meters, offset, upsample_scales, **_unused):
    """fax_modules.py:63-81 — list of (3, H_i, W_i) ego-frame coordinates per pyramid level."""
    V_inv = torch.FloatTensor(get_view_matrix(bev_height, bev_width, h_meters, w_meters, offset)).inverse()
    grids = []
    for scale in upsample_scales:
        h, w = bev_height // scale, bev_width // scale
        g = generate_grid(h, w)
        g[0] = bev_width * g[0]
        g[1] = bev_height * g[1]
        g = (V_inv @ g.reshape(3, h * w)).reshape(3, h, w)
        grids.append(g)
    return grids


def image_plane(feat_height, feat_width, image_height, image_width):
    """fax_modules.py:275-277 — (3, h, w) pixel coordinates of the feature-map cells in the input image."""
    g = generate_grid(feat_height, feat_width)
    g[0] *= image_width
    g[1] *= image_height
    return g


# ---------------------------------------------------------------------------------------------
# window / grid partition index maps (integer, bit-exact) — fax_modules.py:399-404,417-424
# ---------------------------------------------------------------------------------------------
def window_partition_index(H, W, w1, w2):
    """[l = x*Y + y][t = i*w2 + j] -> flat pixel (x*w1+i)*W + (y*w2+j)   ('(x w1) (y w2) -> x y w1 w2')"""
    X, Y = H // w1, W // w2
    x, y, i, j = np.meshgrid(np.arange(X), np.arange(Y), np.arange(w1), np.arange(w2), indexing="ij")
    return ((x * w1 + i) * W + (y * w2 + j)).reshape(X * Y, w1 * w2).astype(np.int64)


def grid_partition_index(H, W, w1, w2):
    """[l = x*Y + y][t = i*w2 + j] -> flat pixel (i*X+x)*W + (j*Y+y)     ('(w1 x) (w2 y) -> x y w1 w2')"""
    X, Y = H // w1, W // w2
    x, y, i, j = np.meshgrid(np.arange(X), np.arange(Y), np.arange(w1), np.arange(w2), indexing="ij")
    return ((i * X + x) * W + (j * Y + y)).reshape(X * Y, w1 * w2).astype(np.int64)


def _window_partition(x, w1, w2):
    """(b, n, H, W, d) -> (b, n, X, Y, w1, w2, d)"""
    b, n, H, W, d = x.shape
    return x.reshape(b, n, H // w1, w1, W // w2, w2, d).permute(0, 1, 2, 4, 3, 5, 6)


def _grid_partition(x, w1, w2):
    """(b, n, H, W, d) -> (b, n, X, Y, w1, w2, d) with pixel (i*X + x, j*Y + y)"""
    b, n, H, W, d = x.shape
    return x.reshape(b, n, w1, H // w1, w2, W // w2, d).permute(0, 1, 3, 5, 2, 4, 6)


def _window_reverse(x):
    """(b, X, Y, w1, w2, d) -> (b, X*w1, Y*w2, d)"""
    b, X, Y, w1, w2, d = x.shape
    return x.permute(0, 1, 3, 2, 4, 5).reshape(b, X * w1, Y * w2, d)


def _ln(x, sd, key):
    return F.layer_norm(x, (x.shape[-1],), sd[key + ".weight"], sd[key + ".bias"], 1e-5)


def _lin(x, sd, key):
    return F.linear(x, sd[key + ".weight"], sd.get(key + ".bias"))


# ---------------------------------------------------------------------------------------------
def cross_win_attention(sd, pfx, q, k, v, skip, heads, dim_head):
    """fax_modules.py:198-248.  q (b n X Y W1 W2 d); k, v (b n x y w1 w2 d); skip (b X Y W1 W2 d) or None."""
    assert k.shape == v.shape
    b, n, X, Y, W1, W2, d = q.shape
    _, nk, kx, ky, w1, w2, _ = k.shape
    assert X * Y == kx * ky
    scale = dim_head ** -0.5
    # 'b n x y w1 w2 d -> b (x y) (n w1 w2) d'
    q = q.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, X * Y, n * W1 * W2, d)
    k = k.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, kx * ky, nk * w1 * w2, d)
    v = v.permute(0, 2, 3, 1, 4, 5, 6).reshape(b, kx * ky, nk * w1 * w2, d)
    q = _lin(_ln(q, sd, pfx + "to_q.0"), sd, pfx + "to_q.1")
    k = _lin(_ln(k, sd, pfx + "to_k.0"), sd, pfx + "to_k.1")
    v = _lin(_ln(v, sd, pfx + "to_v.0"), sd, pfx + "to_v.1")

    def heads_first(t):  # 'b l Q (m d) -> b m l Q d'
        return t.reshape(t.shape[0], t.shape[1], t.shape[2], heads, dim_head).permute(0, 3, 1, 2, 4)

    q, k, v = heads_first(q), heads_first(k), heads_first(v)
    dot = scale * torch.matmul(q, k.transpose(-1, -2))       # scale AFTER the GEMM (:226)
    att = dot.softmax(dim=-1)
    a = torch.matmul(att, v)                                  # b m l Q dh
    a = a.permute(0, 2, 3, 1, 4).reshape(b, X * Y, n * W1 * W2, heads * dim_head)
    a = a.reshape(b, X, Y, n, W1, W2, heads * dim_head).permute(0, 3, 1, 2, 4, 5, 6)  # b n X Y W1 W2 d
    z = _lin(a, sd, pfx + "proj")
    z = z.mean(1)
    if skip is not None:
        z = z + skip
    return z


def _pad_divisible(x, win_h, win_w):
    """fax_modules.py:315-321 — zero pad bottom/right of (.., h, w) to multiples of the window."""
    h, w = x.shape[-2:]
    h_pad, w_pad = ((h + win_h) // win_h) * win_h, ((w + win_w) // win_w) * win_w
    padh = h_pad - h if h % win_h != 0 else 0
    padw = w_pad - w if w % win_w != 0 else 0
    return F.pad(x, (0, padw, 0, padh), value=0)


def _pre_act_conv1x1(x, sd, key):
    """nn.Sequential(BatchNorm2d, ReLU, Conv2d 1x1 no bias) — fax_modules.py:281-292"""
    y = F.relu(bn_eval(x, sd, key + ".0"))
    return F.conv2d(y, sd[key + ".2.weight"])


def cross_view_swap_attention(sd, pfx, cfg, index, x, grid, feature, I_inv, E_inv):
    """fax_modules.py:323-441.  x (b d H W); grid (3 H W) bev.grid{index}; feature (b n C h w); I_inv (b n 3 3);
    E_inv (b n 4 4).  cfg keys: image_height, image_width, q_win_size, feat_win_size, heads, dim_head,
    bev_embedding_flag, skip, no_image_features."""
    b, n, _, h, w = feature.shape
    _, d, H, W = x.shape
    heads, dim_head = cfg["heads"][index], cfg["dim_head"][index]
    qw, fw = cfg["q_win_size"][index], cfg["feat_win_size"][index]
    use_skip = cfg.get("skip", True)

    pixel = image_plane(h, w, cfg["image_height"], cfg["image_width"])            # 3 h w
    c = E_inv[..., -1:]                                                            # b n 4 1
    c_embed = F.conv2d(c.reshape(b * n, 4, 1, 1), sd[pfx + "cam_embed.weight"])    # (bn) d 1 1
    cam = I_inv @ pixel.reshape(1, 1, 3, h * w)                                    # b n 3 hw
    cam = F.pad(cam, (0, 0, 0, 1), value=1)                                        # b n 4 hw
    dd = (E_inv @ cam).reshape(b * n, 4, h, w)
    d_embed = F.conv2d(dd, sd[pfx + "img_embed.weight"])
    img_embed = d_embed - c_embed
    img_embed = img_embed / (img_embed.norm(dim=1, keepdim=True) + 1e-7)

    if cfg["bev_embedding_flag"][index]:
        w_embed = F.conv2d(grid[:2][None], sd[pfx + "bev_embed.weight"], sd[pfx + "bev_embed.bias"])  # 1 d H W
        bev_embed = w_embed - c_embed
        bev_embed = bev_embed / (bev_embed.norm(dim=1, keepdim=True) + 1e-7)
        query = bev_embed.reshape(b, n, d, H, W) + x[:, None]
    else:
        query = x[:, None]                                                         # b 1 d H W

    feature_flat = feature.reshape(b * n, -1, h, w)
    if cfg.get("no_image_features", False):
        key_flat = img_embed
    else:
        key_flat = img_embed + _pre_act_conv1x1(feature_flat, sd, pfx + "feature_proj")
    val_flat = _pre_act_conv1x1(feature_flat, sd, pfx + "feature_linear")
    key = _pad_divisible(key_flat.reshape(b, n, d, h, w), fw[0], fw[1])
    val = _pad_divisible(val_flat.reshape(b, n, d, h, w), fw[0], fw[1])

    to_last = lambda t: t.permute(0, 1, 3, 4, 2)                                   # b n d h w -> b n h w d
    key_l, val_l, query_l = to_last(key), to_last(val), to_last(query)
    x_l = x.permute(0, 2, 3, 1)                                                    # b H W d

    # local-to-local
    q1 = _window_partition(query_l, qw[0], qw[1])
    k1 = _window_partition(key_l, fw[0], fw[1])
    v1 = _window_partition(val_l, fw[0], fw[1])
    skip1 = _window_partition(x_l[:, None], qw[0], qw[1])[:, 0] if use_skip else None
    out = _window_reverse(cross_win_attention(sd, pfx + "cross_win_attend_1.", q1, k1, v1, skip1, heads, dim_head))
    out = out + _lin(F.gelu(_lin(_ln(out, sd, pfx + "prenorm_1"), sd, pfx + "mlp_1.0")), sd, pfx + "mlp_1.2")

    # local-to-global: queries window-partitioned, keys/values grid-partitioned
    x_skip = out
    q2 = _window_partition(out[:, None].expand(b, n, H, W, d), qw[0], qw[1])
    k2 = _grid_partition(key_l, fw[0], fw[1])
    v2 = _grid_partition(val_l, fw[0], fw[1])
    skip2 = _window_partition(x_skip[:, None], qw[0], qw[1])[:, 0] if use_skip else None
    out = _window_reverse(cross_win_attention(sd, pfx + "cross_win_attend_2.", q2, k2, v2, skip2, heads, dim_head))
    out = out + _lin(F.gelu(_lin(_ln(out, sd, pfx + "prenorm_2"), sd, pfx + "mlp_2.0")), sd, pfx + "mlp_2.2")
    out = _ln(out, sd, pfx + "postnorm")
    return out.permute(0, 3, 1, 2)                                                 # b d H W


def rel_pos_index_2d(window_size):
    """fax_modules.py:123-128 — [(i1 j1), (i2 j2)] -> (i1-i2+w-1)(2w-1) + (j1-j2+w-1)"""
    pos = np.arange(window_size)
    gi, gj = np.meshgrid(pos, pos, indexing="ij")
    gi, gj = gi.reshape(-1), gj.reshape(-1)
    di = gi[:, None] - gi[None, :] + window_size - 1
    dj = gj[:, None] - gj[None, :] + window_size - 1
    return (di * (2 * window_size - 1) + dj).astype(np.int64)


def global_attention(sd, pfx, x, dim_head, window_size):
    """fax_modules.py:132-176 — full self-attention over (h w) tokens with 2-D relative position bias."""
    b, dim, height, width = x.shape
    heads = dim // dim_head
    scale = dim_head ** -0.5
    t = x.permute(0, 2, 3, 1).reshape(b, height * width, dim)
    q, k, v = F.linear(t, sd[pfx + "to_qkv.weight"]).chunk(3, dim=-1)
    split = lambda z: z.reshape(b, -1, heads, dim_head).permute(0, 2, 1, 3)
    q, k, v = split(q), split(k), split(v)
    q = q * scale
    sim = torch.matmul(q, k.transpose(-1, -2))
    idx = torch.from_numpy(rel_pos_index_2d(window_size))
    bias = sd[pfx + "rel_pos_bias.weight"][idx]                                    # i j h
    sim = sim + bias.permute(2, 0, 1)
    attn = sim.softmax(dim=-1)
    out = torch.matmul(attn, v)                                                    # b m (h w) d
    out = out.permute(0, 2, 1, 3).reshape(b, height, width, dim)
    out = F.linear(out, sd[pfx + "to_out.0.weight"])
    return out.permute(0, 3, 1, 2)


def _downsample(sd, pfx, x):
    """fax_modules.py:476-489 — conv3x3 -> PixelUnshuffle(2) -> conv3x3 -> BN -> ReLU -> conv1x1 -> BN"""
    y = F.conv2d(x, sd[pfx + "0.0.weight"], padding=1)
    y = F.pixel_unshuffle(y, 2)
    y = F.conv2d(y, sd[pfx + "0.2.weight"], padding=1)
    y = F.relu(bn_eval(y, sd, pfx + "0.3"))
    y = F.conv2d(y, sd[pfx + "0.5.weight"])
    return bn_eval(y, sd, pfx + "0.6")


def fax_module(sd, pfx, cfg, features, intrinsic, extrinsic, invert_extrinsic=False, final_self_attn=True, taps=None):
    """FAXModule.forward, fax_modules.py:497-521.  features: list of (b l n C h w); intrinsic (b l n 3 3);
    extrinsic (b l n 4 4).  Returns (b l d H W).  nuScenes flavour (encoder_pyramid_axial.py:534-558):
    invert_extrinsic=True, final_self_attn=False."""
    b, l, n = features[0].shape[:3]
    I_inv = intrinsic.reshape(b * l, n, 3, 3).inverse()
    E_inv = extrinsic.reshape(b * l, n, 4, 4)
    if invert_extrinsic:
        E_inv = E_inv.inverse()
    grids = bev_grids(**cfg["bev_embedding"])
    cva_cfg = dict(cfg["cross_view"])
    cva_cfg.update(cfg["cross_view_swap"])
    x = sd[pfx + "bev_embedding.learned_features"]
    x = x[None].expand(b * l, *x.shape)
    for i, feature in enumerate(features):
        feature = feature.reshape(b * l, n, *feature.shape[3:])
        x = cross_view_swap_attention(sd, pfx + "cross_views.%d." % i, cva_cfg, i, x, grids[i], feature, I_inv, E_inv)
        for j in range(cfg["middle"][i]):
            x = bottleneck_forward(sd, pfx + "layers.%d.%d." % (i, j), x)
        if taps is not None:                      # per-level BEV query (b*l, d, H_i, W_i) in front of the down-sampling layer
            taps["level%d" % i] = x
        if i < len(features) - 1:
            x = _downsample(sd, pfx + "downsample_layers.%d." % i, x)
    if final_self_attn:
        x = global_attention(sd, pfx + "self_attn.", x, cfg["self_attn"]["dim_head"], cfg["self_attn"]["window_size"])
    return x.reshape(b, l, *x.shape[1:])
