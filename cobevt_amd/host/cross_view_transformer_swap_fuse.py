"""CrossViewTransformerSwapFuse (CVT per agent + FuseBEVT swap fusion) — mirror of
opv2v/opencood/models/cross_view_transformer_swap_fuse.py:63-131 (cvt_swap_fuse.yaml).  Same tail as CorpBEVT: regroup + STTF
warp + ROI mask in one kernel, SwapFusionEncoder, NaiveDecoder, BevSegHead."""
import torch

from .. import ops
from . import runtime as rt
from . import training
from .bev_seg_head import BevSegHead
from .corpbevt import STTF
from .cross_view_transformer import CrossViewTransformer
from .swap_fusion_modules import SwapFusionEncoder


class _CvtFusionBase(CrossViewTransformer):
    """encoder + cvm + decoder + head of CrossViewTransformer plus the V2V tail shared by the fusion baselines"""

    def __init__(self, config):
        super().__init__(config)
        self.max_cav = config["max_cav"]
        self.downsample_rate = config["sttf"]["downsample_rate"]
        self.discrete_ratio = config["sttf"]["resolution"]
        self.use_roi_mask = config["sttf"]["use_roi_mask"]
        self.sttf = STTF(config["sttf"])

    def _warp(self, feats, transformation_matrix, record_len):
        """(N, H, W, C) -> warped (B, L, H, W, C), com_mask (B, H, W, 1, L)"""
        dev = feats.device
        rl = torch.as_tensor(record_len).to(device=dev, dtype=torch.int32)
        tm = transformation_matrix.to(device=dev, dtype=torch.float32).contiguous()
        x, com_mask, cav_mask = ops.sttf_warp(feats, tm, None, self.discrete_ratio, self.downsample_rate,
                                              want_mask=self.use_roi_mask, record_len=rl, max_cav=self.max_cav)
        if not self.use_roi_mask:
            b, l, h, w, _ = x.shape
            com_mask = cav_mask[:, None, None, None, :].expand(b, h, w, 1, l).contiguous()
        return x, com_mask

    def _fuse(self, x, com_mask):
        raise NotImplementedError

    def _fuse_train(self, x, com_mask):
        """train() mode: x (B, L, H, W, C) fp32 autograd tensor, com_mask (B, H, W, 1, L) -> fused (B, C, H, W)"""
        raise NotImplementedError

    def fuse_and_decode(self, feats, transformation_matrix, record_len):
        x, com_mask = self._warp(feats, transformation_matrix, record_len)
        y = self.decoder.forward_nhwc(self._fuse(x, com_mask))       # (B, 8H, 8W, C')
        return self.seg_head(rt.nchw_view(y), y.shape[0], 1)

    def forward(self, batch_dict):
        if self.training:                       # train_camera.py:143-179: the differentiable graph of host/training.py
            f = training.cvt_encode_agents(self, batch_dict).squeeze(1)
            return training.cvt_fuse_and_decode(self, f, batch_dict["transformation_matrix"], batch_dict["record_len"])
        feats = self.encode_agents(batch_dict)
        return self.fuse_and_decode(feats, batch_dict["transformation_matrix"], batch_dict["record_len"])


class CrossViewTransformerSwapFuse(_CvtFusionBase):
    def __init__(self, config):
        super().__init__(config)
        self.fusion_net = SwapFusionEncoder(config["swap_fusion"])

    def _fuse(self, x, com_mask):
        return self.fusion_net.forward_blhwc(x, com_mask)

    def _fuse_train(self, x, com_mask):
        return training.swap_fusion_encoder(self.fusion_net, x.permute(0, 1, 4, 2, 3).contiguous(), com_mask)
