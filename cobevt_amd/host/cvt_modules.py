"""CVT per-agent encoder behind the reference's module API — mirror of opv2v/opencood/models/sub_modules/cvt_modules.py
(BEVEmbedding :41-90, CrossAttention :93-170, CrossViewAttention :173-283, CrossViewModule :286-327): same class names,
constructor arguments, state_dict keys and forward contracts.  The baseline fusion models (cross_view_transformer*.py) run on it
(SURVEY.md §8f rank 4).  Everything is the FAX hot path's kernels: ray / BEV embeddings, pre-activation 1x1 projections,
LayerNorm-folded dense-row GEMMs, the fused Bottleneck - and `cobevt_window_attention` in its camera-paired mode (mean_q = 2):
one window = the whole map, camera c's query copy scores camera c's keys, ONE softmax over all cameras' keys (:142-153)."""
import torch
import torch.nn as nn

from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from .fax_modules import ResNetBottleNeck, generate_grid, get_view_matrix
from .runtime import HipModule


class BEVEmbedding(nn.Module):
    """cvt_modules.py:41-90 — one grid (non-persistent buffer) at bev / 2^len(decoder_blocks) + the learned prior."""

    def __init__(self, dim, sigma, bev_height, bev_width, h_meters, w_meters, offset, decoder_blocks):
        super().__init__()
        h, w = bev_height // (2 ** len(decoder_blocks)), bev_width // (2 ** len(decoder_blocks))
        grid = generate_grid(h, w).squeeze(0)
        grid[0] = bev_width * grid[0]
        grid[1] = bev_height * grid[1]
        V_inv = torch.FloatTensor(get_view_matrix(bev_height, bev_width, h_meters, w_meters, offset)).inverse()
        grid = (V_inv @ grid.reshape(3, h * w)).reshape(3, h, w)
        self.register_buffer("grid", grid, persistent=False)
        self.learned_features = nn.Parameter(sigma * torch.randn(dim, h, w))

    def get_prior(self):
        return self.learned_features


class CrossAttention(HipModule):
    """cvt_modules.py:93-170."""

    def __init__(self, dim, heads, dim_head, qkv_bias, norm=nn.LayerNorm):
        super().__init__()
        if dim_head != 32:
            raise CobevtHipError("the HIP attention kernel is built for dim_head = 32")
        self.scale = dim_head ** -0.5
        self.heads, self.dim_head = heads, dim_head
        self.to_q = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_k = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_v = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.proj = nn.Linear(heads * dim_head, dim)
        self.prenorm = norm(dim)
        self.mlp = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.postnorm = norm(dim)

    def attend_tokens(self, q, k, v, skip, H, W, h, w):
        """q (b, n, H*W, d), k / v (b*n, h, w, d) channels-last compute dtype, skip (b, H, W, d) | None -> (b, H, W, d)"""
        b, n = q.shape[:2]
        inner = self.heads * self.dim_head
        qt = ops.linear(q, rt.linear_plan(self, "q", self.to_q[1], ln=self.to_q[0]))
        kt = ops.linear(k, rt.linear_plan(self, "k", self.to_k[1], ln=self.to_k[0]))
        vt = ops.linear(v, rt.linear_plan(self, "v", self.to_v[1], ln=self.to_v[0]))
        a = torch.empty((b, H, W, inner), device=qt.device, dtype=qt.dtype)
        qmap, kmap, omap = ops.tokmap(0, n, H, W, H, W), ops.tokmap(0, n, h, w, h, w), ops.tokmap(0, 1, H, W, H, W)
        ops.window_attention(qt, kt, vt, a, qmap, kmap, omap, b, self.heads, self.scale, inner, inner, inner, inner, mean_q=2)
        z = ops.linear(a, rt.linear_plan(self, "proj", self.proj), residual=skip)
        z = rt.layernorm(self, "prenorm", self.prenorm, z)                  # z = prenorm(z); z = z + mlp(z); z = postnorm(z)
        t = ops.linear(z, rt.linear_plan(self, "mlp0", self.mlp[0], act=2))
        z = ops.linear(t, rt.linear_plan(self, "mlp2", self.mlp[2]), residual=z)
        return rt.layernorm(self, "postnorm", self.postnorm, z)

    def forward(self, q, k, v, skip=None):
        """q (b n d H W); k, v (b n d h w); skip (b d H W) -> (b d H W)"""
        self._require_inference(q, k, v, skip)
        b, n, d, H, W = q.shape
        h, w = k.shape[-2:]
        ql = rt.to_nhwc(q.reshape(b * n, d, H, W)).reshape(b, n, H * W, d)
        kl, vl = rt.to_nhwc(k.reshape(b * n, d, h, w)), rt.to_nhwc(v.reshape(b * n, d, h, w))
        sk = rt.to_nhwc(skip) if skip is not None else None
        return rt.like_input(rt.nchw_view(self.attend_tokens(ql, kl, vl, sk, H, W, h, w)), q)


class CrossViewAttention(HipModule):
    """cvt_modules.py:173-283."""

    def __init__(self, feat_height, feat_width, feat_dim, dim, config):
        super().__init__()
        image_plane = generate_grid(feat_height, feat_width)[None]
        image_plane[:, :, 0] *= config["image_width"]
        image_plane[:, :, 1] *= config["image_height"]
        self.register_buffer("image_plane", image_plane, persistent=False)
        self.feature_linear = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        if config["no_image_features"]:
            self.feature_proj = None
        else:
            self.feature_proj = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.bev_embed = nn.Conv2d(2, dim, 1)
        self.img_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cam_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cross_attend = CrossAttention(dim, config["heads"], config["dim_head"], config["qkv_bias"])
        self.skip = config["skip"]
        self.dim = dim

    def forward_nhwc(self, x, bev, feature, I_inv, E_inv, batch):
        """x (b, H, W, d); feature (b*n, h, w, C) compute dtype; I_inv (b*n, 3, 3), E_inv (b*n, 4, 4) fp32 -> (b, H, W, d)"""
        bn, h, w, _ = feature.shape
        b, H, W, d = x.shape
        n = bn // batch
        dt = feature.dtype
        plane = rt.f32_param(self, "plane", self.image_plane, (3, h * w))
        w_img = rt.f32_param(self, "w_img", self.img_embed.weight, (d, 4))
        w_cam = rt.f32_param(self, "w_cam", self.cam_embed.weight, (d, 4))
        img = ops.ray_embed(I_inv, E_inv, plane, w_img, w_cam, h * w, d, dt).reshape(bn, h, w, d)
        if self.feature_proj is not None:
            key = ops.conv2d(feature, rt.conv_plan(self, "fproj", self.feature_proj[2], pre_bn=self.feature_proj[0]), residual=img)
        else:
            key = img
        val = ops.conv2d(feature, rt.conv_plan(self, "flin", self.feature_linear[2], pre_bn=self.feature_linear[0]))
        world = rt.f32_param(self, "world", bev.grid[:2], (2, H * W))
        w_bev = rt.f32_param(self, "w_bev", self.bev_embed.weight, (d, 2))
        b_bev = rt.f32_param(self, "b_bev", self.bev_embed.bias)
        query = ops.bev_embed(E_inv, world, w_bev, b_bev, w_cam, x.reshape(b, H * W, d), n)          # (b, n, HW, d)
        skip = (x if x.is_contiguous() else x.contiguous()) if self.skip else None
        return self.cross_attend.attend_tokens(query, key, val, skip, H, W, h, w)

    def forward(self, x, bev, feature, I_inv, E_inv):
        """x (b, d, H, W); feature (b, n, C, h, w); I_inv (b, n, 3, 3); E_inv (b, n, 4, 4) -> (b, d, H, W)"""
        self._require_inference(x, feature, I_inv, E_inv)
        b, n = feature.shape[:2]
        f = rt.to_nhwc(feature.reshape(b * n, *feature.shape[2:]))
        Ii = I_inv.reshape(b * n, 3, 3).to(torch.float32).contiguous()
        Ei = E_inv.reshape(b * n, 4, 4).to(torch.float32).contiguous()
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x), bev, f, Ii, Ei, b)), x)


class CrossViewModule(HipModule):
    """cvt_modules.py:286-327."""

    def __init__(self, config):
        super().__init__()
        middle, dim = config["middle"], config["dim"]
        self.backbone_output_shape = config["backbone_output_shape"]
        assert len(middle) == len(self.backbone_output_shape)
        cross_views, layers = [], []
        for feat_shape, num_layers in zip(self.backbone_output_shape, middle):
            _, _, _, feat_dim, feat_height, feat_width = tuple(feat_shape)
            cross_views.append(CrossViewAttention(feat_height, feat_width, feat_dim, dim, config["cross_view"]))
            layers.append(nn.Sequential(*[ResNetBottleNeck(dim) for _ in range(num_layers)]))
        self.bev_embedding = BEVEmbedding(dim, **config["bev_embedding"])
        self.cross_views = nn.ModuleList(cross_views)
        self.layers = nn.ModuleList(layers)

    def forward_features(self, feats, I_inv, E_inv, batch):
        """feats: list of (batch*n, h, w, C) channels-last -> (batch, H, W, d) channels-last"""
        dt = rt.get_compute_dtype()
        prior = self._plan("prior", [self.bev_embedding.learned_features],
                           lambda d_, dev: self.bev_embedding.learned_features.detach().permute(1, 2, 0).to(dt).contiguous())
        x = prior[None].expand(batch, *prior.shape)                  # stride-0 batch view: bev_embed broadcasts it
        for cross_view, feature, layer in zip(self.cross_views, feats, self.layers):
            x = cross_view.forward_nhwc(x, self.bev_embedding, feature, I_inv, E_inv, batch)
            for blk in layer:
                x = blk.forward_nhwc(x)
        return x

    def forward(self, batch):
        b, l, n = batch["inputs"].shape[:3]
        intrinsic, extrinsic = batch["intrinsic"], batch["extrinsic"]
        self._require_inference(intrinsic, extrinsic, *batch["features"])
        I_inv = ops.invert_small(intrinsic.reshape(b * l * n, 3, 3))             # cvt_modules.py:314-315
        E_inv = extrinsic.reshape(b * l * n, 4, 4).to(torch.float32).contiguous()   # un-inverted, :316-317
        feats = [rt.to_nhwc(f.reshape(b * l * n, *f.shape[3:])) for f in batch["features"]]
        x = rt.nchw_view(self.forward_features(feats, I_inv, E_inv, b * l))
        return x.reshape(b, l, *x.shape[1:])
