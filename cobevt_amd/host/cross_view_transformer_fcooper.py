"""CrossViewTransformerFcooper (CVT per agent + F-Cooper max-out) — mirror of
opv2v/opencood/models/cross_view_transformer_fcooper.py:62-129 with SpatialFusionMask (fusion_modules/f_cooper_fuse.py:30-36):
the fused map is the element-wise maximum over the max_cav agent slots (zero-padded slots take part, the mask is not used)."""
import torch.nn as nn

from .. import ops
from .cross_view_transformer_swap_fuse import _CvtFusionBase


class SpatialFusionMask(nn.Module):
    """x (B, L, H, W, C) channels-last device tensor -> max over L (f_cooper_fuse.py:30-36)"""

    def forward(self, x):
        return ops.agent_max(x.contiguous())


class CrossViewTransformerFcooper(_CvtFusionBase):
    def __init__(self, config):
        super().__init__(config)
        self.fusion_net = SpatialFusionMask()

    def _fuse(self, x, com_mask):
        return self.fusion_net(x)

    def _fuse_train(self, x, com_mask):
        # the reference's own op (f_cooper_fuse.py:30-36: torch.max over the agent slots, its gradient goes to the arg-max agent)
        return x.max(dim=1)[0].permute(0, 3, 1, 2)
