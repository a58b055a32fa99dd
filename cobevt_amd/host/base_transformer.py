"""PreNormResidual / FeedForward — mirror of opv2v/opencood/models/base_transformer.py:102-124 (the only parts
of that file on the FAX hot path; CavAttention / HGT / BaseTransformer are V2X-ViT baselines, out of scope)."""
import torch.nn as nn

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
