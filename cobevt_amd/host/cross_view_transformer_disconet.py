"""CrossViewTransformerDiscoNet (CVT per agent + DiscoNet pixel-weighted fusion) — mirror of
opv2v/opencood/models/cross_view_transformer_disconet.py:14-68 (cvt_disconet.yaml)."""
from . import training
from .cross_view_transformer_v2vnet import _CvtPairwiseBase
from .v2v_fuse import DiscoNetFusion


class CrossViewTransformerDiscoNet(_CvtPairwiseBase):
    def __init__(self, config):
        super().__init__(config)
        self.fusion_net = DiscoNetFusion(config["disconet_fusion"])

    def _fuse_train(self, f, record_len, pairwise_t_matrix, record_len_host=None):
        return training.disconet_fusion(self.fusion_net, f, record_len, pairwise_t_matrix, record_len_host)
