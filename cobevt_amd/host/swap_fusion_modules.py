"""MI355X-native FuseBEVT (swap fusion) behind the reference's module API.

Mirror of opv2v/opencood/models/fusion_modules/swap_fusion_modules.py (class names, constructor arguments,
state_dict keys incl. the persistent `relative_position_index` buffer, forward contracts).  Agents x window
tokens are gathered straight from the (b, l, h, w, d) channels-last tensor by the attention kernel for both the
window and the dilated-grid pass; the 3-D relative position bias is computed from token coordinates in-kernel
(the index buffer is kept only for state_dict compatibility).
"""
import torch
import torch.nn as nn

from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from . import training
from .base_transformer import FeedForward, PreNormResidual
from .runtime import HipModule


class Attention(HipModule):
    """swap_fusion_modules.py:13-128."""

    def __init__(self, dim, dim_head=32, dropout=0., agent_size=6, window_size=7):
        super().__init__()
        assert (dim % dim_head) == 0, "dimension should be divisible by dimension per head"
        if dim_head != 32:
            raise CobevtHipError("the HIP attention kernel is built for dim_head = 32")
        self.heads = dim // dim_head
        self.scale = dim_head ** -0.5
        self.window_size = [agent_size, window_size, window_size]
        self.to_qkv = nn.Linear(dim, dim * 3, bias=False)
        self.attend = nn.Sequential(nn.Softmax(dim=-1))
        self.to_out = nn.Sequential(nn.Linear(dim, dim, bias=False), nn.Dropout(dropout))
        L, w = agent_size, window_size
        self.relative_position_bias_table = nn.Embedding((2 * L - 1) * (2 * w - 1) * (2 * w - 1), self.heads)
        # persistent buffer of the reference (:63-85): token t = (l*w + a)*w + b
        cl, ca, cb = torch.meshgrid(torch.arange(L), torch.arange(w), torch.arange(w), indexing="ij")
        cl, ca, cb = cl.reshape(-1), ca.reshape(-1), cb.reshape(-1)
        index = ((cl[:, None] - cl[None, :] + L - 1) * (2 * w - 1) * (2 * w - 1)
                 + (ca[:, None] - ca[None, :] + w - 1) * (2 * w - 1) + (cb[:, None] - cb[None, :] + w - 1))
        self.register_buffer("relative_position_index", index)

    def stage_bias_table(self):
        """the relative-position table as the single-launch stage kernel reads it: one column per head, [heads][rows padded to 4],
        in the base-2 domain (x log2 e), as a flat zero-padded image of ops.SWAP_STAGE_BIAS_FLOATS floats - the kernel copies the
        whole image into LDS with unconditional 16-byte loads and gathers (query term - key term) from it"""
        def build(dt, dev):
            t = self.relative_position_bias_table.weight.detach().to(device=dev, dtype=torch.float32).t().contiguous()
            rows = t.shape[1]
            pad = (-rows) % 4
            if pad:
                t = torch.nn.functional.pad(t, (0, pad))
            flat = (t * 1.4426950408889634).reshape(-1)
            if flat.numel() > ops.SWAP_STAGE_BIAS_FLOATS:
                return None
            return torch.nn.functional.pad(flat, (0, ops.SWAP_STAGE_BIAS_FLOATS - flat.numel())).contiguous()
        return self._plan("stage_table", [self.relative_position_bias_table.weight], build)

    def qkv_plan(self, ln=None):
        """to_qkv (with the PreNormResidual LayerNorm `ln` folded in) as a plan the producer of the rows may run."""
        return rt.linear_plan(self, "qkv", self.to_qkv, ln=ln)

    def forward_fused(self, xn, residual=None, mask=None, mode=2, ln=None, core_only=False, qkv=None):
        """xn (compute dtype; LayerNorm'ed already, or raw with ln=<nn.LayerNorm container> to fuse it): mode 2 -> (b l X Y w1 w2 d) partitioned, mask (b X Y w1 w2 1 l);
        mode 0 (window) / 1 (grid) -> (b l H W d), mask (b H W 1 l).  Returns to_out(attn) (+ residual).
        qkv: to_qkv(ln(xn)) when the producer of xn already computed it."""
        L, w = self.window_size[0], self.window_size[1]
        if mode == 2:
            b, l, X, Y, w1, w2, d = xn.shape
            m = (2, l, X * w1, Y * w2, w1, w2, X, Y)
        else:
            b, l, H, W, d = xn.shape
            w1 = w2 = w
            m = ops.tokmap(mode, l, H, W, w, w)
        if l != L or w1 != w or w2 != w:
            raise CobevtHipError("swap attention built for %d agents x %dx%d windows, got %d x %dx%d" % (L, w, w, l, w1, w2))
        if qkv is None:
            qkv = ops.linear(xn, self.qkv_plan(ln))
        out = torch.empty(xn.shape, device=xn.device, dtype=xn.dtype)
        table = rt.f32_param(self, "table", self.relative_position_bias_table.weight)
        mk = None
        if mask is not None:
            mk = mask.to(torch.float32)
            mk = mk if mk.is_contiguous() else mk.contiguous()
        ops.window_attention(qkv, qkv, qkv, out, m, m, m, b, self.heads, self.scale, 3 * d, 3 * d, 3 * d, d, koff=d,
                             voff=2 * d, bias_table=table, bias_L=L, mask=mk)
        if core_only:
            return out
        return ops.linear(out, rt.linear_plan(self, "out", self.to_out[0]), residual=residual)

    def forward(self, x, mask=None):
        """x: (b, l, X, Y, w1, w2, c); mask: (b, X, Y, w1, w2, 1, l) or None"""
        if self.training:
            return training.swap_attention(self, x, mask, 2)
        self._require_inference(x, mask)
        return rt.like_input(self.forward_fused(rt.as_compute(x), mask=mask, mode=2), x)


def _to_blhwc(x):
    """(b, l, c, h, w)-shaped -> contiguous (b, l, h, w, c) compute dtype"""
    b, l, c, h, w = x.shape
    return rt.to_nhwc(x.reshape(b * l, c, h, w)).reshape(b, l, h, w, c)


def _from_blhwc(x):
    return x.permute(0, 1, 4, 2, 3)


def _attn_ffd(attn_res, ffd_res, x, mask, mode, qkv=None, next_attn=None):
    """PreNormResidual(Attention) followed by PreNormResidual(FeedForward) on (b, l, h, w, d):
    attention core, then to_out + residual + LayerNorm + FeedForward + residual as one fused row chain.
    qkv: this attention's to_qkv(norm(x)) if the previous stage already produced it; next_attn: the PreNormResidual
    (Attention) that consumes the result - its norm + to_qkv then ride in this stage's launch and (x, qkv) is returned."""
    attn, ffd = attn_res.fn, ffd_res.fn
    nxt = next_attn.fn.qkv_plan(next_attn.norm) if next_attn is not None else None
    plan_p = rt.linear_plan(attn, "out", attn.to_out[0])
    plan_1 = rt.linear_plan(ffd, "fc1", ffd.net[0], act=2, ln=ffd_res.norm)
    plan_2 = rt.linear_plan(ffd, "fc2", ffd.net[3])
    if x.dim() == 5 and mode in (0, 1):
        b, l, H, W, d = x.shape
        L, w = attn.window_size[0], attn.window_size[1]
        if l == L and H % w == 0 and W % w == 0:
            tmap = ops.tokmap(mode, l, H, W, w, w)
            mk = None
            if mask is not None:
                mk = mask.to(torch.float32)
                mk = mk if mk.is_contiguous() else mk.contiguous()
            if qkv is None and x.dtype == torch.bfloat16:
                qkv = ops.linear(x, attn.qkv_plan(attn_res.norm))
            table = attn.stage_bias_table() if attn.heads == 4 else None
            if qkv is not None and table is not None and ops.swap_stage_fusable(qkv, x, tmap, attn.heads, plan_p, plan_1, plan_2, nxt, mk):
                # the whole half in ONE launch: attention core, to_out + residual, pre-norm FeedForward + residual, and the
                # next half's LayerNorm + to_qkv (csrc/swap_stage.hip)
                out, qn = ops.swap_stage(qkv, x, tmap, b, attn.heads, attn.scale, table, L, mk, plan_p, plan_1, plan_2, nxt)
                return (out, qn) if nxt is not None else out
    a = attn.forward_fused(x, mask=mask, mode=mode, ln=attn_res.norm, core_only=True, qkv=qkv)
    return ops.attn_mlp_chain(a, x, plan_p, plan_1, plan_2, next_plan=nxt)


def _run_stages(stages, x, mask_of):
    """stages: [(PreNormResidual(Attention), PreNormResidual(FeedForward), mode)] applied in order."""
    qkv = None
    for i, (ar, fr, mode) in enumerate(stages):
        nxt = stages[i + 1][0] if i + 1 < len(stages) else None
        r = _attn_ffd(ar, fr, x, mask_of(i), mode, qkv=qkv, next_attn=nxt)
        x, qkv = r if nxt is not None else (r, None)
    return x


class SwapFusionBlockMask(HipModule):
    """swap_fusion_modules.py:131-192."""

    def __init__(self, input_dim, mlp_dim, dim_head, window_size, agent_size, drop_out):
        super().__init__()
        self.window_size = window_size
        self.window_attention = PreNormResidual(input_dim, Attention(input_dim, dim_head, drop_out, agent_size, window_size))
        self.window_ffd = PreNormResidual(input_dim, FeedForward(input_dim, mlp_dim, drop_out))
        self.grid_attention = PreNormResidual(input_dim, Attention(input_dim, dim_head, drop_out, agent_size, window_size))
        self.grid_ffd = PreNormResidual(input_dim, FeedForward(input_dim, mlp_dim, drop_out))

    def stages(self):
        return [(self.window_attention, self.window_ffd, 0), (self.grid_attention, self.grid_ffd, 1)]

    uses_mask = True

    def forward_blhwc(self, x, mask):
        return _run_stages(self.stages(), x, lambda i: mask)

    def forward(self, x, mask):
        """x: (b, l, c, h, w); mask: (b, h, w, 1, l)"""
        if self.training:
            return training.swap_fusion_block(self, x, mask)
        self._require_inference(x, mask)
        return rt.like_input(_from_blhwc(self.forward_blhwc(_to_blhwc(x), mask)), x)


class SwapFusionBlock(HipModule):
    """swap_fusion_modules.py:195-230 — nn.Sequential `block` with parametrised entries at 1, 2, 5, 6."""

    def __init__(self, input_dim, mlp_dim, dim_head, window_size, agent_size, drop_out):
        super().__init__()
        self.block = nn.Sequential(
            nn.Identity(),
            PreNormResidual(input_dim, Attention(input_dim, dim_head, drop_out, agent_size, window_size)),
            PreNormResidual(input_dim, FeedForward(input_dim, mlp_dim, drop_out)),
            nn.Identity(),
            nn.Identity(),
            PreNormResidual(input_dim, Attention(input_dim, dim_head, drop_out, agent_size, window_size)),
            PreNormResidual(input_dim, FeedForward(input_dim, mlp_dim, drop_out)),
            nn.Identity())

    def stages(self):
        return [(self.block[1], self.block[2], 0), (self.block[5], self.block[6], 1)]

    uses_mask = False

    def forward_blhwc(self, x, mask=None):
        return _run_stages(self.stages(), x, lambda i: None)

    def forward(self, x, mask=None):
        if self.training:
            return training.swap_fusion_block(self, x, None)
        self._require_inference(x)
        return rt.like_input(_from_blhwc(self.forward_blhwc(_to_blhwc(x))), x)


class SwapFusionEncoder(HipModule):
    """swap_fusion_modules.py:233-286."""

    def __init__(self, args):
        super().__init__()
        self.layers = nn.ModuleList([])
        self.depth = args["depth"]
        input_dim, mlp_dim = args["input_dim"], args["mlp_dim"]
        agent_size, window_size = args["agent_size"], args["window_size"]
        drop_out, dim_head = args["drop_out"], args["dim_head"]
        self.mask = bool(args["mask"]) if "mask" in args else False
        for _ in range(self.depth):
            cls = SwapFusionBlockMask if self.mask else SwapFusionBlock
            self.layers.append(cls(input_dim, mlp_dim, dim_head, window_size, agent_size, drop_out))
        # Reduce('b m d h w -> b d h w', 'mean'), Rearrange, LayerNorm, Linear, Rearrange
        self.mlp_head = nn.Sequential(nn.Identity(), nn.Identity(), nn.LayerNorm(input_dim),
                                      nn.Linear(input_dim, input_dim), nn.Identity())

    def forward_blhwc(self, x, mask=None):
        """x (b, l, h, w, d) channels-last compute dtype -> (b, h, w, d)"""
        stages, masks = [], []
        for layer in self.layers:
            st = layer.stages()
            stages += st
            masks += [mask if layer.uses_mask else None] * len(st)
        x = _run_stages(stages, x, lambda i: masks[i])
        b, l, h, w, d = x.shape
        ln = self.mlp_head[2]
        # mean over the agents -> LayerNorm -> Linear: one launch (the mean and the normalisation happen while the GEMM stages its rows)
        y = ops.mean_ln_linear(x.reshape(b, l, h * w, d), rt.linear_plan(self, "head.fc_ln", self.mlp_head[3], ln=ln))
        if y is None:
            y = ops.mean_layernorm(x.reshape(b, l, h * w, d), rt.f32_param(self, "head.ln.w", ln.weight),
                                   rt.f32_param(self, "head.ln.b", ln.bias), ln.eps)
            y = ops.linear(y, rt.linear_plan(self, "head.fc", self.mlp_head[3]))
        return y.reshape(b, h, w, -1)

    def forward(self, x, mask=None):
        """x: (b, m, d, h, w); mask: (b, h, w, 1, m) -> (b, d, h, w).  In train() mode: the differentiable fp32 graph of
        host/training.py (HIP attention / LayerNorm / GELU kernels in both directions)."""
        if self.training:
            return training.swap_fusion_encoder(self, x, mask)
        self._require_inference(x, mask)
        return rt.like_input(rt.nchw_view(self.forward_blhwc(_to_blhwc(x), mask)), x)


def sharded_stages(encoder):
    """The encoder as the stage list cobevt_amd.dist.RowShardedFuseBEVT runs over row-sharded maps: [(mode, fn)] with
    fn(x (b, l, h_local, W, d) compute dtype, mask (b, h_local, W, 1, l) | None) -> same shape, and head(x) -> (b, h_local, W, d).
    Window passes see bands of whole windows, grid passes the (i, x_local) re-ordered rows that form whole grid groups -
    both are plain window / dilated-grid partitions of the LOCAL map, so the kernels are the single-GPU ones."""
    stages = []
    for layer in encoder.layers:
        for ar, fr, mode in layer.stages():
            def fn(x, mask, ar=ar, fr=fr, mode=mode, use=layer.uses_mask):
                return _attn_ffd(ar, fr, x.contiguous(), mask if use else None, mode)
            stages.append((mode, fn))

    def head(x):
        b, l, h, w, d = x.shape
        ln = encoder.mlp_head[2]
        y = ops.mean_layernorm(x.contiguous().reshape(b, l, h * w, d), rt.f32_param(encoder, "head.ln.w", ln.weight),
                               rt.f32_param(encoder, "head.ln.b", ln.bias), ln.eps)
        y = ops.linear(y, rt.linear_plan(encoder, "head.fc", encoder.mlp_head[3]))
        return y.reshape(b, h, w, d)
    return stages, head
