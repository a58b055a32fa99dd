"""CrossViewTransformerV2VNet (CVT per agent + V2VNet message passing) — mirror of
opv2v/opencood/models/cross_view_transformer_v2vnet.py:13-68 (cvt_v2vnet.yaml)."""
from . import runtime as rt
from . import training
from .cross_view_transformer import CrossViewTransformer
from .v2v_fuse import V2VNetFusion


class _CvtPairwiseBase(CrossViewTransformer):
    """encoder + cvm + decoder + head of CrossViewTransformer; the fusion consumes batch['pairwise_t_matrix']"""

    def __init__(self, config):
        super().__init__(config)
        self.downsample_rate = config["sttf"]["downsample_rate"]
        self.discrete_ratio = config["sttf"]["resolution"]
        self.use_roi_mask = config["sttf"]["use_roi_mask"]

    def forward(self, batch_dict):
        if self.training:                       # train_camera.py:143-179: the differentiable graph of host/training.py
            return training.cvt_pairwise_model(self, batch_dict)
        feats = self.encode_agents(batch_dict)                                   # (N, H, W, C)
        fused = self.fusion_net.forward_nhwc(feats, batch_dict["record_len"], batch_dict["pairwise_t_matrix"])
        y = self.decoder.forward_nhwc(fused)
        return self.seg_head(rt.nchw_view(y), y.shape[0], 1)


class CrossViewTransformerV2VNet(_CvtPairwiseBase):
    def __init__(self, config):
        super().__init__(config)
        self.fusion_net = V2VNetFusion(config["v2vnet_fusion"])

    def _fuse_train(self, f, record_len, pairwise_t_matrix, record_len_host=None):
        return training.v2vnet_fusion(self.fusion_net, f, record_len, pairwise_t_matrix)
