"""PreNormResidual / FeedForward — mirror of opv2v/opencood/models/base_transformer.py:102-124 (the parts of that file on
the FAX hot path) - and PreNorm / CavAttention / BaseEncoder / BaseTransformer (:91-99,127-172,321-362), the per-pixel agent
attention of the CVT + AttFuse baseline.  HGTCavAttention and the V2X-ViT temporal encodings stay out."""
import torch
import torch.nn as nn

from ..lib import CobevtHipError

from .. import ops
from . import runtime as rt
from . import training
from .runtime import HipModule


class PreNormResidual(HipModule):
    """fn(LayerNorm(x)) + x.  The residual add is fused into fn's last GEMM when fn offers `forward_fused`."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward_fused(self, x, **kwargs):
        """x: contiguous channels-last tensor in the compute dtype."""
        return self.fn.forward_fused(x, residual=x, ln=self.norm, **kwargs)

    def forward(self, x, **kwargs):
        if self.training:
            if isinstance(self.fn, FeedForward):
                return training.feed_forward(self.fn, x, norm=self.norm)
            return training.swap_attention(self.fn, x, kwargs.get("mask"), 2, norm=self.norm)
        self._require_inference(x)
        return rt.like_input(self.forward_fused(rt.as_compute(x), **kwargs), x)


class FeedForward(HipModule):
    """Linear -> GELU -> Dropout -> Linear -> Dropout (dropout is identity at inference)."""

    def __init__(self, dim, hidden_dim, dropout=0.):
        super().__init__()
        self.net = nn.Sequential(nn.Linear(dim, hidden_dim), nn.GELU(), nn.Dropout(dropout),
                                 nn.Linear(hidden_dim, dim), nn.Dropout(dropout))

    def forward_fused(self, x, residual=None, ln=None):
        t = ops.linear(x, rt.linear_plan(self, "fc1", self.net[0], act=2, ln=ln))
        return ops.linear(t, rt.linear_plan(self, "fc2", self.net[3]), residual=residual)

    def forward(self, x):
        if self.training:
            return training.feed_forward(self, x)
        self._require_inference(x)
        return rt.like_input(self.forward_fused(rt.as_compute(x)), x)


class PreNorm(HipModule):
    """fn(LayerNorm(x)) (base_transformer.py:91-99); the residual is added by the caller (BaseEncoder)."""

    def __init__(self, dim, fn):
        super().__init__()
        self.norm = nn.LayerNorm(dim)
        self.fn = fn

    def forward(self, x, **kwargs):
        self._require_inference(x)
        return rt.like_input(self.fn.forward_fused(rt.as_compute(x), ln=self.norm, **kwargs), x)


class CavAttention(HipModule):
    """At every BEV pixel, multi-head attention over the agents (base_transformer.py:127-172).  It is the gathered window
    attention kernel with 1 x 1 windows: tokens of a window = the L agents' features at that pixel, key mask = com_mask."""

    def __init__(self, dim, heads, dim_head=64, dropout=0.1):
        super().__init__()
        if dim_head != 32:
            raise CobevtHipError("the HIP attention kernel is built for dim_head = 32")
        inner_dim = heads * dim_head
        self.heads = heads
        self.scale = dim_head ** -0.5
        self.attend = nn.Softmax(dim=-1)
        self.to_qkv = nn.Linear(dim, inner_dim * 3, bias=False)
        self.to_out = nn.Sequential(nn.Linear(inner_dim, dim), nn.Dropout(dropout))

    def forward_fused(self, x, residual=None, ln=None, mask=None):
        """x (b, l, h, w, c) compute dtype (raw, with ln = the PreNorm LayerNorm to fold into to_qkv); mask (b, h, w, 1, l) or
        (b, l)-broadcastable; -> to_out(attention) (+ residual)"""
        b, l, h, w, c = x.shape
        inner = self.heads * 32
        qkv = ops.linear(x, rt.linear_plan(self, "qkv", self.to_qkv, ln=ln))
        out = torch.empty((b, l, h, w, inner), device=x.device, dtype=x.dtype)
        m = ops.tokmap(0, l, h, w, 1, 1)
        mk = None
        if mask is not None:
            mk = mask.to(torch.float32).expand(b, h, w, 1, l).reshape(b, h, w, l).contiguous()
        ops.window_attention(qkv, qkv, qkv, out, m, m, m, b, self.heads, self.scale, 3 * inner, 3 * inner, 3 * inner, inner,
                             koff=inner, voff=2 * inner, mask=mk)
        return ops.linear(out, rt.linear_plan(self, "out", self.to_out[0]), residual=residual)

    def forward(self, x, mask, prior_encoding=None):
        """x (B, L, H, W, C); mask (B, H, W, 1, L) (or (B, 1, 1, 1, L)) -> (B, L, H, W, C)"""
        self._require_inference(x, mask)
        return rt.like_input(self.forward_fused(rt.as_compute(x), mask=mask), x)


class BaseEncoder(HipModule):
    """base_transformer.py:321-339: depth x [PreNorm(CavAttention) + x, PreNorm(FeedForward) + x]"""

    def __init__(self, dim, depth, heads, dim_head, mlp_dim, dropout=0.):
        super().__init__()
        self.layers = nn.ModuleList([])
        for _ in range(depth):
            self.layers.append(nn.ModuleList([PreNorm(dim, CavAttention(dim, heads=heads, dim_head=dim_head, dropout=dropout)),
                                              PreNorm(dim, FeedForward(dim, mlp_dim, dropout=dropout))]))

    def forward_blhwc(self, x, mask):
        for attn, ff in self.layers:
            x = attn.fn.forward_fused(x, residual=x, ln=attn.norm, mask=mask)
            x = ff.fn.forward_fused(x, residual=x, ln=ff.norm)
        return x

    def forward(self, x, mask):
        self._require_inference(x, mask)
        return rt.like_input(self.forward_blhwc(rt.as_compute(x), mask), x)


class BaseTransformer(HipModule):
    """base_transformer.py:342-362: the encoder, then the ego agent's map.  x (B, L, H, W, C) -> (B, H, W, C)"""

    def __init__(self, args):
        super().__init__()
        self.encoder = BaseEncoder(args["dim"], args["depth"], args["heads"], args["dim_head"], args["mlp_dim"], args["dropout"])

    def forward_blhwc(self, x, mask):
        return self.encoder.forward_blhwc(x, mask)[:, 0]

    def forward(self, x, mask):
        self._require_inference(x, mask)
        return rt.like_input(self.forward_blhwc(rt.as_compute(x), mask), x)
