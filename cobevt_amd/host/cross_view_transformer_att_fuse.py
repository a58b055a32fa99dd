"""CrossViewTransformerAttFuse (CVT per agent + per-pixel agent attention) — mirror of
opv2v/opencood/models/cross_view_transformer_att_fuse.py:62-131 (cvt_att_fuse.yaml): regroup + STTF warp + ROI mask in one kernel,
BaseTransformer (base_transformer.py:342-362) over the agents at every BEV pixel, NaiveDecoder, BevSegHead."""
from . import training
from .base_transformer import BaseTransformer
from .cross_view_transformer_swap_fuse import _CvtFusionBase


class CrossViewTransformerAttFuse(_CvtFusionBase):
    def __init__(self, config):
        super().__init__(config)
        self.fusion_net = BaseTransformer(config["base_transformer"])

    def _fuse(self, x, com_mask):
        return self.fusion_net.forward_blhwc(x, com_mask).contiguous()

    def _fuse_train(self, x, com_mask):
        return training.base_transformer(self.fusion_net, x, com_mask).permute(0, 3, 1, 2)
