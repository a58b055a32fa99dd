"""MI355X-native FAX cross-view pyramid behind the reference's module API.

Mirror of opv2v/opencood/models/sub_modules/fax_modules.py: same class names, constructor signatures,
state_dict keys and forward contracts.  The nn.Linear / nn.Conv2d / nn.LayerNorm /
nn.BatchNorm2d children are parameter containers only; the arithmetic is HIP kernels (cobevt_amd/ops.py).
Internally activations are channels-last (b, H, W, d) in the compute dtype and the window / grid partitions
are index arithmetic inside the attention kernel; the public forwards accept/return the reference's shapes as
(zero-copy where possible) views.
"""
import torch
import torch.nn as nn

from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from . import training
from .runtime import HipModule


def generate_grid(height, width):
    """fax_modules.py:13-21 -> (1, 3, h, w)"""
    xs = torch.linspace(0, 1, width)
    ys = torch.linspace(0, 1, height)
    gx = xs[None, :].expand(height, width)
    gy = ys[:, None].expand(height, width)
    return torch.stack([gx, gy, torch.ones(height, width)], 0)[None].contiguous()


def get_view_matrix(h=200, w=200, h_meters=100.0, w_meters=100.0, offset=0.0):
    """fax_modules.py:24-35"""
    sh = h / h_meters
    sw = w / w_meters
    return [[0., -sw, w / 2.], [-sh, 0., h * offset + h / 2.], [0., 0., 1.]]


class Bottleneck(HipModule):
    """torchvision.models.resnet.Bottleneck(inplanes, planes) container + HIP forward (used as
    ResNetBottleNeck(c) = Bottleneck(c, c // 4), fax_modules.py:10,472)."""
    expansion = 4

    def __init__(self, inplanes, planes):
        super().__init__()
        width = planes
        self.conv1 = nn.Conv2d(inplanes, width, 1, bias=False)
        self.bn1 = nn.BatchNorm2d(width)
        self.conv2 = nn.Conv2d(width, width, 3, 1, 1, bias=False)
        self.bn2 = nn.BatchNorm2d(width)
        self.conv3 = nn.Conv2d(width, planes * self.expansion, 1, bias=False)
        self.bn3 = nn.BatchNorm2d(planes * self.expansion)
        self.relu = nn.ReLU(inplace=True)
        self.downsample = None

    def entry_plan(self):
        """conv1 + bn1 + ReLU as a row-local GEMM plan: the producer of x may compute it in its own launch
        (ops.attn_mlp_chain next_plan) and hand the result to forward_nhwc(x, y1=...)."""
        return rt.conv_plan(self, "c1", self.conv1, self.bn1, act=1)

    def fused_plan(self):
        """the whole block as one launch (bf16, Bottleneck(128, 32)): ops.BottleneckPlan, or None when the kernel does not apply"""
        if self.conv1.weight.shape[:2] != (32, 128) or self.downsample is not None:
            return None
        return self._plan("fused", rt.module_tensors(self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3),
                          lambda dt, dev: ops.BottleneckPlan(self.conv1, self.bn1, self.conv2, self.bn2, self.conv3, self.bn3, dev))

    def fusable(self, x):
        return ops.bottleneck_fusable(x) and self.conv1.weight.shape[:2] == (32, 128)

    def forward_nhwc(self, x, y1=None):
        if y1 is None and self.fusable(x):
            return ops.bottleneck(x, self.fused_plan())
        p1, p2, p3 = self.entry_plan(), rt.conv_plan(self, "c2", self.conv2, self.bn2, act=1), rt.conv_plan(self, "c3", self.conv3, self.bn3, act=1)
        if ops.bottleneck_f32_fusable(x, p1, p2, p3, y1):       # fp32 storage: one launch (csrc/bottleneck_f32.hip), conv1 from the producer if it made it
            return ops.bottleneck_f32(x, p1, p2, p3, y1)
        y = ops.conv2d(x, p1) if y1 is None else y1
        y = ops.conv2d(y, p2)
        return ops.conv2d(y, p3, residual=x)

    def forward(self, x):
        if self.training:
            return training.bottleneck(self, x)
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)


ResNetBottleNeck = lambda c: Bottleneck(c, c // 4)  # noqa: E731


class BEVEmbedding(nn.Module):
    """fax_modules.py:38-90 — init-time grids (non-persistent buffers) + learned prior."""

    def __init__(self, dim, sigma, bev_height, bev_width, h_meters, w_meters, offset, upsample_scales):
        super().__init__()
        V_inv = torch.FloatTensor(get_view_matrix(bev_height, bev_width, h_meters, w_meters, offset)).inverse()
        for i, scale in enumerate(upsample_scales):
            h, w = bev_height // scale, bev_width // scale
            grid = generate_grid(h, w).squeeze(0)
            grid[0] = bev_width * grid[0]
            grid[1] = bev_height * grid[1]
            grid = (V_inv @ grid.reshape(3, h * w)).reshape(3, h, w)
            self.register_buffer("grid%d" % i, grid, persistent=False)
        self.learned_features = nn.Parameter(
            sigma * torch.randn(dim, bev_height // upsample_scales[0], bev_width // upsample_scales[0]))

    def get_prior(self):
        return self.learned_features


class Attention(HipModule):
    """FAX global self-attention with 2-D relative position bias, fax_modules.py:93-176."""

    def __init__(self, dim, dim_head=32, dropout=0., window_size=25):
        super().__init__()
        assert (dim % dim_head) == 0, "dimension should be divisible by dimension per head"
        if dim_head != 32:
            raise CobevtHipError("the HIP attention kernel is built for dim_head = 32")
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.window_size = window_size
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attend = nn.Sequential(nn.Softmax(dim=-1), nn.Dropout(dropout))
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(dropout))
        self.rel_pos_bias = nn.Embedding((2 * window_size - 1) ** 2, self.heads)
        pos = torch.arange(window_size)
        gi, gj = torch.meshgrid(pos, pos, indexing="ij")
        grid = torch.stack([gi.reshape(-1), gj.reshape(-1)], -1)              # (i j) c
        rel_pos = grid[:, None, :] - grid[None, :, :] + window_size - 1
        rel_pos_indices = (rel_pos * torch.tensor([2 * window_size - 1, 1])).sum(dim=-1)
        self.register_buffer("rel_pos_indices", rel_pos_indices, persistent=False)

    def forward_nhwc(self, x, out=None):
        b, h, w, d = x.shape
        if h != self.window_size or w != self.window_size:
            raise CobevtHipError("FAX global attention expects a %dx%d map" % (self.window_size, self.window_size))
        qkv = ops.linear(x, rt.linear_plan(self, "qkv", self.to_qkv))
        a = torch.empty((b, h, w, d), device=x.device, dtype=x.dtype)
        m = ops.tokmap(0, 1, h, w, h, w)
        table = rt.f32_param(self, "bias", self.rel_pos_bias.weight)
        ops.window_attention(qkv, qkv, qkv, a, m, m, m, b, self.heads, self.scale, 3 * d, 3 * d, 3 * d, d, koff=d,
                             voff=2 * d, bias_table=table, bias_L=1)
        return ops.linear(a, rt.linear_plan(self, "out", self.to_out[0]), out=out)

    def forward(self, x):
        if self.training:
            return training.global_attention(self, x)
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)


class CrossWinAttention(HipModule):
    """fax_modules.py:179-248."""

    def __init__(self, dim, heads, dim_head, qkv_bias, rel_pos_emb=False, norm=nn.LayerNorm):
        super().__init__()
        if dim_head != 32:
            raise CobevtHipError("the HIP attention kernel is built for dim_head = 32")
        self.scale = dim_head ** -0.5
        self.heads = heads
        self.dim_head = dim_head
        self.rel_pos_emb = rel_pos_emb
        self.to_q = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_k = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.to_v = nn.Sequential(norm(dim), nn.Linear(dim, heads * dim_head, bias=qkv_bias))
        self.proj = nn.Linear(heads * dim_head, dim)

    def add_rel_pos_emb(self, x):
        return x

    def _project(self, name, seq, x, out=None):
        return ops.linear(x, rt.linear_plan(self, name, seq[1], ln=seq[0]), out=out)

    def project_kv(self, k_src, v_src, out_k=None, out_v=None):
        """LayerNorm + Linear of the key / value tokens (independent of the query -> can run ahead on a side stream).
        out_k / out_v: optional preallocated result buffers (a pipelined caller keeps them across steps)."""
        return self._project("k", self.to_k, k_src, out_k), self._project("v", self.to_v, v_src, out_v)

    def q_plan(self):
        """LayerNorm + Linear of the query as a plan (so the producer of the query rows can compute it in its launch)."""
        return rt.linear_plan(self, "q", self.to_q[1], ln=self.to_q[0])

    def attend_projected(self, q_src, kt, vt, qmap, kmap, omap, batch, out_shape, qt=None, ldkv=None, kvoff=0):
        """Query projection (unless `qt` already holds it) + fused window attention on already projected keys / values;
        returns the head-merged attention output BEFORE self.proj.  ldkv / kvoff: row stride and column offset of this
        attention's keys / values when kt / vt hold the projections of several attentions side by side."""
        inner = self.heads * self.dim_head
        if qt is None:
            qt = self._project("q", self.to_q, q_src)
        ld = inner if ldkv is None else ldkv
        a = torch.empty(tuple(out_shape) + (inner,), device=qt.device, dtype=qt.dtype)
        ops.window_attention(qt, kt, vt, a, qmap, kmap, omap, batch, self.heads, self.scale, inner, ld, ld, inner,
                             koff=kvoff, voff=kvoff, mean_q=qmap[1] > 1)
        return a

    def attend_core(self, q_src, k_src, v_src, qmap, kmap, omap, batch, out_shape):
        kt, vt = self.project_kv(k_src, v_src)
        return self.attend_projected(q_src, kt, vt, qmap, kmap, omap, batch, out_shape)

    def attend(self, q_src, k_src, v_src, qmap, kmap, omap, batch, skip, out_shape):
        """q_src/k_src/v_src: token-major tensors whose rows the maps address; returns proj(attn) (+skip)."""
        a = self.attend_core(q_src, k_src, v_src, qmap, kmap, omap, batch, out_shape)
        return ops.linear(a, rt.linear_plan(self, "proj", self.proj), residual=skip)

    def forward(self, q, k, v, skip=None):
        """q: (b n X Y W1 W2 d); k, v: (b n x y w1 w2 d); skip (b X Y W1 W2 d) -> (b X Y W1 W2 d)"""
        if self.training:
            return training.cross_win_attention(self, q, k, v, skip)
        self._require_inference(q, k, v, skip)
        assert k.shape == v.shape
        b, n, X, Y, W1, W2, _ = q.shape
        _, nk, kx, ky, w1, w2, _ = k.shape
        assert X * Y == kx * ky
        qmap = (2, n, X * W1, Y * W2, W1, W2, X, Y)
        kmap = (2, nk, kx * w1, ky * w2, w1, w2, kx, ky)
        omap = (2, 1, X * W1, Y * W2, W1, W2, X, Y)
        sk = rt.as_compute(skip) if skip is not None else None
        z = self.attend(rt.as_compute(q), rt.as_compute(k), rt.as_compute(v), qmap, kmap, omap, b, sk,
                        (b, X, Y, W1, W2))
        return rt.like_input(z, q)


class CrossViewSwapAttention(HipModule):
    """fax_modules.py:251-441."""

    def __init__(self, feat_height, feat_width, feat_dim, dim, index, image_height, image_width, qkv_bias, q_win_size,
                 feat_win_size, heads, dim_head, bev_embedding_flag, rel_pos_emb=False, no_image_features=False,
                 skip=True, norm=nn.LayerNorm):
        super().__init__()
        image_plane = generate_grid(feat_height, feat_width)[None]
        image_plane[:, :, 0] *= image_width
        image_plane[:, :, 1] *= image_height
        self.register_buffer("image_plane", image_plane, persistent=False)
        self.feature_linear = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        if no_image_features:
            self.feature_proj = None
        else:
            self.feature_proj = nn.Sequential(nn.BatchNorm2d(feat_dim), nn.ReLU(), nn.Conv2d(feat_dim, dim, 1, bias=False))
        self.bev_embed_flag = bev_embedding_flag[index]
        if self.bev_embed_flag:
            self.bev_embed = nn.Conv2d(2, dim, 1)
        self.img_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.cam_embed = nn.Conv2d(4, dim, 1, bias=False)
        self.q_win_size = q_win_size[index]
        self.feat_win_size = feat_win_size[index]
        self.rel_pos_emb = rel_pos_emb
        self.cross_win_attend_1 = CrossWinAttention(dim, heads[index], dim_head[index], qkv_bias)
        self.cross_win_attend_2 = CrossWinAttention(dim, heads[index], dim_head[index], qkv_bias)
        self.skip = skip
        self.prenorm_1 = norm(dim)
        self.prenorm_2 = norm(dim)
        self.mlp_1 = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.mlp_2 = nn.Sequential(nn.Linear(dim, 2 * dim), nn.GELU(), nn.Linear(2 * dim, dim))
        self.postnorm = norm(dim)
        self.dim = dim

    def _padded_hw(self, h, w):
        """pad_divisble, fax_modules.py:315-321"""
        wh, ww = self.feat_win_size
        hp = ((h + wh) // wh) * wh if h % wh != 0 else h
        wp = ((w + ww) // ww) * ww if w % ww != 0 else w
        return hp, wp

    def _proj_mlp(self, name, attn, a, skip, prenorm, mlp, postnorm=None, next_plan=None):
        """proj(a) + skip -> x + mlp(prenorm(x)) -> postnorm : one fused launch in bf16 mode (ops.attn_mlp_chain);
        with next_plan also returns next_plan(result) (same launch when it fits)."""
        post = None
        if postnorm is not None:
            post = (rt.f32_param(self, name + ".post.w", postnorm.weight), rt.f32_param(self, name + ".post.b", postnorm.bias),
                    postnorm.eps)
        return ops.attn_mlp_chain(a, skip, rt.linear_plan(attn, "proj", attn.proj),
                                  rt.linear_plan(self, name + ".0", mlp[0], act=2, ln=prenorm),
                                  rt.linear_plan(self, name + ".2", mlp[2]), post, next_plan=next_plan)

    def prepare_kv(self, feature, I_inv, E_inv, batch, out=None):
        """Everything that depends only on the image features and camera geometry (fax_modules.py:346-358,377-396 and
        the to_k / to_v projections of both attentions): ray embedding, key / value 1x1 projections, zero padding to
        the window grid, LayerNorm + Linear of K and V.  Independent of the BEV query, so FAXModule may run it ahead
        of time on a side stream.  feature (b*n,h,w,C); I_inv (b*n,3,3), E_inv (b*n,4,4) fp32."""
        bn, h, w, _ = feature.shape
        d, dt = self.dim, feature.dtype
        n = bn // batch
        w1, w2 = self.feat_win_size
        hp, wp = self._padded_hw(h, w)
        padded = (hp, wp) != (h, w)
        plane = rt.f32_param(self, "plane", self.image_plane, (3, h * w))
        w_img = rt.f32_param(self, "w_img", self.img_embed.weight, (d, 4))
        w_cam = rt.f32_param(self, "w_cam", self.cam_embed.weight, (d, 4))
        img = ops.ray_embed(I_inv, E_inv, plane, w_img, w_cam, h * w, d, dt).reshape(bn, h, w, d)

        def kv_buffer(tag):
            """zero-padded (bn, hp, wp, d) map whose interior the convolution's epilogue writes.  The pad rows / columns are
            zeroed ONCE: the buffer is a per-module scratch map (only read by the two GEMMs below, in this call), so a
            replayed HIP graph holds no fill kernel for it"""
            if not padded:
                return None
            return self._plan("kvpad.%s.%dx%dx%d" % (tag, bn, hp, wp), [self.img_embed.weight],
                              lambda dt_, dev: torch.zeros((bn, hp, wp, d), device=feature.device, dtype=dt))

        o = out if out is not None else {}      # optional preallocated {"kk", "vv"} buffers
        # to_k of both attentions read the same `key` rows (to_v: `val`): one GEMM each with the two weight matrices
        # stacked (each LayerNorm's affine folded into its half; the normalisation itself is shared) -> (.., 2*inner);
        # attention #1 reads columns [0, inner), attention #2 columns [inner, 2*inner) through its row stride
        pk, pv = self._pair_plan("k"), self._pair_plan("v")
        plan_key = rt.conv_plan(self, "fproj", self.feature_proj[2], pre_bn=self.feature_proj[0]) if self.feature_proj is not None else None
        plan_val = rt.conv_plan(self, "flin", self.feature_linear[2], pre_bn=self.feature_linear[0])
        if (not padded and plan_key is not None and ops.proj_chain_fusable(feature, plan_key, pk, img)
                and ops.proj_chain_fusable(feature, plan_val, pv, None)):
            # 128-channel features (pyramid level 0: 3/4 of the key / value rows of a frame): projection + ray embedding ->
            # LayerNorm -> stacked to_k (to_v) in one launch per operand; `key` / `val` stay in LDS (csrc/row_chain.hip)
            kk = ops.proj_chain(feature, plan_key, pk, residual=img, out_next=o.get("kk"))
            vv = ops.proj_chain(feature, plan_val, pv, out_next=o.get("vv"))
            return {"n": n, "hp": hp, "wp": wp, "kk": kk, "vv": vv}
        if not padded and plan_key is not None and ops.proj_chain_kv_fusable(feature, plan_key, plan_val, pk, pv, img):
            # 256- / 512-channel features (pyramid levels 1 and 2): both operands' chains in ONE launch (csrc/proj_chain_k.hip)
            kk, vv = ops.proj_chain_kv(feature, plan_key, plan_val, pk, pv, residual=img, out_k=o.get("kk"), out_v=o.get("vv"))
            return {"n": n, "hp": hp, "wp": wp, "kk": kk, "vv": vv}
        if self.feature_proj is not None:
            key = ops.conv2d(feature, plan_key, residual=img, out=kv_buffer("key"))
        elif padded:
            key = ops.copy_into_interior(img, kv_buffer("key"))
        else:
            key = img
        val = ops.conv2d(feature, plan_val, out=kv_buffer("val"))
        kk = ops.linear(key, pk, out=o.get("kk"))
        vv = ops.linear(val, pv, out=o.get("vv"))
        return {"n": n, "hp": hp, "wp": wp, "kk": kk, "vv": vv}

    def _pair_plan(self, which):
        a1, a2 = getattr(self.cross_win_attend_1, "to_" + which), getattr(self.cross_win_attend_2, "to_" + which)
        if a1[0].eps != a2[0].eps:
            raise CobevtHipError("the two cross attentions normalise their keys / values with different eps")

        def build(dt, dev):
            ws, bs = [], []
            for ln, lin in ((a1[0], a1[1]), (a2[0], a2[1])):
                w = lin.weight.detach().double().cpu()
                b = lin.bias.detach().double().cpu() if lin.bias is not None else torch.zeros(w.shape[0], dtype=torch.float64)
                g, be = ln.weight.detach().double().cpu(), ln.bias.detach().double().cpu()
                bs.append(b + w @ be)
                ws.append(w * g[None, :])
            return ops.ConvPlan(torch.cat(ws), torch.cat(bs), dtype=dt, device=dev, ln_folded_eps=a1[0].eps)
        return self._plan("pair_" + which, rt.module_tensors(a1[0], a1[1], a2[0], a2[1]), build)

    def forward_query(self, index, x, bev, E_inv, kv, next_plan=None):
        """The query side of fax_modules.py:360-441 given prepare_kv()'s result.  x (b,H,W,d) -> (b,H,W,d).
        next_plan: the row-local plan that consumes the result next (the following ResNetBottleNeck's conv1); when
        given, returns (result, next_plan(result))."""
        b, H, W, d = x.shape
        n, hp, wp = kv["n"], kv["hp"], kv["wp"]
        W1, W2 = self.q_win_size
        w1, w2 = self.feat_win_size
        if self.bev_embed_flag:
            grid = getattr(bev, "grid%d" % index)
            world = rt.f32_param(self, "world%d" % index, grid[:2], (2, H * W))
            w_bev = rt.f32_param(self, "w_bev", self.bev_embed.weight, (d, 2))
            b_bev = rt.f32_param(self, "b_bev", self.bev_embed.bias)
            w_cam = rt.f32_param(self, "w_cam", self.cam_embed.weight, (d, 4))
            # query = x + bev embedding, then to_q: one launch, the (b, n, HW, d) query is never materialised
            q1 = ops.bev_embed_linear(E_inv, world, w_bev, b_bev, w_cam, x.reshape(b, H * W, d), n,
                                      self.cross_win_attend_1.q_plan())
            query, nq = None, n
        else:
            x = x.contiguous()
            query, nq, q1 = x, 1, None
        qmap_n = ops.tokmap(0, nq, H, W, W1, W2)
        qmap_1 = ops.tokmap(0, 1, H, W, W1, W2)
        kwin = ops.tokmap(0, n, hp, wp, w1, w2)
        kgrid = ops.tokmap(1, n, hp, wp, w1, w2)
        if qmap_1[6] * qmap_1[7] != kwin[6] * kwin[7]:
            raise CobevtHipError("query windows %dx%d != key windows %dx%d" % (qmap_1[6], qmap_1[7], kwin[6], kwin[7]))
        # local-to-local: window queries x window keys; per-camera queries are averaged in-kernel
        inner = self.cross_win_attend_1.heads * self.cross_win_attend_1.dim_head
        a = self.cross_win_attend_1.attend_projected(query, kv["kk"], kv["vv"], qmap_n, kwin, qmap_1, b, (b, H, W), qt=q1,
                                                     ldkv=2 * inner, kvoff=0)
        y, q2 = self._proj_mlp("mlp1", self.cross_win_attend_1, a, x if self.skip else None, self.prenorm_1, self.mlp_1,
                               next_plan=self.cross_win_attend_2.q_plan())
        # local-to-global: the n query replicas of the reference are identical -> one copy
        a = self.cross_win_attend_2.attend_projected(y, kv["kk"], kv["vv"], qmap_1, kgrid, qmap_1, b, (b, H, W), qt=q2,
                                                     ldkv=2 * inner, kvoff=inner)
        return self._proj_mlp("mlp2", self.cross_win_attend_2, a, y if self.skip else None, self.prenorm_2, self.mlp_2,
                              self.postnorm, next_plan=next_plan)

    def forward_nhwc(self, index, x, bev, feature, I_inv, E_inv):
        """x (b,H,W,d); feature (b*n,h,w,C) compute dtype; I_inv (b*n,3,3), E_inv (b*n,4,4) fp32 -> (b,H,W,d)"""
        return self.forward_query(index, x, bev, E_inv, self.prepare_kv(feature, I_inv, E_inv, x.shape[0]))

    def forward(self, index, x, bev, feature, I_inv, E_inv):
        """x (b,d,H,W); feature (b,n,C,h,w); I_inv (b,n,3,3); E_inv (b,n,4,4) -> (b,d,H,W)"""
        if self.training:
            return training.cross_view_swap_attention(self, index, x, bev, feature, I_inv, E_inv)
        self._require_inference(x, feature, I_inv, E_inv)
        b, n = feature.shape[:2]
        f = rt.to_nhwc(feature.reshape(b * n, *feature.shape[2:]))
        Ii = I_inv.reshape(b * n, 3, 3).to(torch.float32).contiguous()
        Ei = E_inv.reshape(b * n, 4, 4).to(torch.float32).contiguous()
        y = self.forward_nhwc(index, rt.to_nhwc(x), bev, f, Ii, Ei)
        return rt.like_input(rt.nchw_view(y), x)


class _Downsample(HipModule):
    """The inner nn.Sequential of FAXModule.downsample_layers[i] (fax_modules.py:477-489):
    conv3x3 -> PixelUnshuffle(2) -> conv3x3 -> BN -> ReLU -> conv1x1 -> BN, indices 0,1,2,3,4,5,6."""

    def __init__(self, dim_in, dim_mid, dim_out):
        super().__init__()
        mods = [nn.Conv2d(dim_in, dim_mid, kernel_size=3, stride=1, padding=1, bias=False), nn.PixelUnshuffle(2),
                nn.Conv2d(dim_out, dim_out, 3, padding=1, bias=False), nn.BatchNorm2d(dim_out), nn.ReLU(inplace=True),
                nn.Conv2d(dim_out, dim_out, 1, padding=0, bias=False), nn.BatchNorm2d(dim_out)]
        for i, m in enumerate(mods):
            self.add_module(str(i), m)

    def __getitem__(self, i):
        return getattr(self, str(i))

    def forward_nhwc(self, x, out=None):
        y = ops.conv2d(x, rt.conv_plan(self, "c0", self[0], store_mode=1))          # conv + PixelUnshuffle(2)
        y = ops.conv2d(y, rt.conv_plan(self, "c2", self[2], self[3], act=1))
        return ops.conv2d(y, rt.conv_plan(self, "c5", self[5], self[6]), out=out)


class FAXModule(HipModule):
    """fax_modules.py:444-521."""

    # OPV2V flavour; the nuScenes encoder overrides these (encoder_pyramid_axial.py:515,532,539)
    _downsample_div = 4

    def __init__(self, config):
        super().__init__()
        middle = config["middle"]
        dim = config["dim"]
        self.backbone_output_shape = config["backbone_output_shape"]
        assert len(middle) == len(self.backbone_output_shape)
        cross_view = config["cross_view"]
        cross_view_swap = config["cross_view_swap"]
        cross_views, layers, downsample_layers = [], [], []
        for i, (feat_shape, num_layers) in enumerate(zip(self.backbone_output_shape, middle)):
            _, _, _, feat_dim, feat_height, feat_width = tuple(feat_shape)
            cross_views.append(CrossViewSwapAttention(feat_height, feat_width, feat_dim, dim[i], i, **cross_view,
                                                      **cross_view_swap))
            layers.append(nn.Sequential(*[ResNetBottleNeck(dim[i]) for _ in range(num_layers)]))
            if i < len(middle) - 1:
                downsample_layers.append(nn.Sequential(_Downsample(dim[i], dim[i] // self._downsample_div, dim[i + 1])))
        self.bev_embedding = BEVEmbedding(dim[0], **config["bev_embedding"])
        self.cross_views = nn.ModuleList(cross_views)
        self.layers = nn.ModuleList(layers)
        self.downsample_layers = nn.ModuleList(downsample_layers)
        self.self_attn = Attention(dim[-1], **config["self_attn"])

    taps = None      # set to a dict to receive intermediate tensors (tests compare them against the oracle's)

    def forward_features(self, features, I_inv, E_inv, batch, kv=None, levels=None, x=None, out=None):
        """features: list of (batch*n, h, w, C) channels-last; returns (batch, H, W, d) channels-last.
        kv: optional list of callables returning each level's prepare_kv() result (computed ahead on side streams).
        levels = (first, last): run only pyramid levels first..last-1 (the global self-attention belongs to the last
        level), starting from `x` when first > 0 - the pieces a frame pipeline runs on different streams.
        out: optional preallocated (batch, H, W, d) buffer for the result (written by the last kernel, no copy)."""
        nlev = len(self.cross_views)
        first, last = (0, nlev) if levels is None else levels
        if first == 0:
            dt = rt.get_compute_dtype()
            prior = self._plan("prior", [self.bev_embedding.learned_features],
                               lambda d_, dev: self.bev_embedding.learned_features.detach().permute(1, 2, 0).to(dt).contiguous())
            x = prior[None].expand(batch, *prior.shape)     # stride-0 batch view: the kernels broadcast it, no copy
        for i, (cross_view, feature, layer) in enumerate(zip(self.cross_views, features, self.layers)):
            if i < first or i >= last:
                continue
            kvi = kv[i]() if kv is not None else cross_view.prepare_kv(feature, I_inv, E_inv, batch)
            blocks = list(layer)
            y1 = None
            if blocks and not (x.dtype == torch.bfloat16 and ops.USE_BOTTLENECK and x.shape[-1] == 128
                               and blocks[0].conv1.weight.shape[:2] == (32, 128)):
                # conv1 of the first Bottleneck rides in the row chain's launch (the fused Bottleneck kernel computes it itself)
                x, y1 = cross_view.forward_query(i, x, self.bev_embedding, E_inv, kvi, next_plan=blocks[0].entry_plan())
            else:
                x = cross_view.forward_query(i, x, self.bev_embedding, E_inv, kvi)
            for j, blk in enumerate(blocks):
                x = blk.forward_nhwc(x, y1=y1 if j == 0 else None)
            if self.taps is not None:                   # tests: the level's BEV query (batch, H_i, W_i, d) channels-last
                self.taps["level%d" % i] = x
            if i < len(self.cross_views) - 1:
                final = out is not None and i == last - 1 and not (self.self_attn is not None and last == nlev)
                x = self.downsample_layers[i][0].forward_nhwc(x, out=out if final else None)
        if self.self_attn is not None and last == nlev:
            x = self.self_attn.forward_nhwc(x, out=out)
        if out is not None and x.data_ptr() != out.data_ptr():
            out.copy_(x)
            x = out
        return x

    def forward(self, batch):
        if self.training:
            return training.fax_module(self, batch)
        b, l, n = batch["inputs"].shape[:3]
        intrinsic, extrinsic = batch["intrinsic"], batch["extrinsic"]
        self._require_inference(intrinsic, extrinsic, *batch["features"])
        I_inv = ops.invert_small(intrinsic.reshape(b * l * n, 3, 3))            # fax_modules.py:500-501
        E_inv = self._extrinsic(extrinsic.reshape(b * l * n, 4, 4).to(torch.float32)).contiguous()
        feats = [rt.to_nhwc(f.reshape(b * l * n, *f.shape[3:])) for f in batch["features"]]
        x = self.forward_features(feats, I_inv, E_inv, b * l)
        x = rt.nchw_view(x)
        return x.reshape(b, l, *x.shape[1:])

    @staticmethod
    def _extrinsic(e):
        return e  # OPV2V passes camera->ego un-inverted (fax_modules.py:502-503)
