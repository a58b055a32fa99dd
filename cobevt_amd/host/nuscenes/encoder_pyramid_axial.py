"""PyramidAxialEncoder — mirror of nuscenes/cross_view_transformer/model/encoder_pyramid_axial.py:475-558.
Same FAX operators as the OPV2V tree (cobevt_amd/host/fax_modules.py); the differences the reference has between
its two trees are parameterised here: both camera matrices are inverted in the model (:538-539), the downsample
block's first conv is dim -> dim//2 (:515), there is no final self-attention (:532,556), images are (b,n,3,h,w)
and normalised inside the encoder (:489,541)."""
import torch
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from ..fax_modules import (BEVEmbedding, CrossViewSwapAttention, FAXModule, ResNetBottleNeck, _Downsample)
from ..runtime import HipModule


class Normalize(HipModule):
    """encoder_pyramid_axial.py:41-49 — (x - mean) / std with non-persistent buffers."""

    def __init__(self, mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225]):
        super().__init__()
        self.register_buffer("mean", torch.tensor(mean)[None, :, None, None], persistent=False)
        self.register_buffer("std", torch.tensor(std)[None, :, None, None], persistent=False)

    def forward(self, x):
        self._require_inference(x)
        scale = self._plan("scale", [self.std], lambda dt, dev: (1.0 / self.std.double()).float().reshape(-1).contiguous())
        shift = self._plan("shift", [self.std, self.mean],
                           lambda dt, dev: (-self.mean.double() / self.std.double()).float().reshape(-1).contiguous())
        return ops.channel_affine(x, scale, shift)


class PyramidAxialEncoder(FAXModule):
    _downsample_div = 2

    def __init__(self, backbone, cross_view, cross_view_swap, bev_embedding, self_attn, dim, middle=[2, 2], scale=1.0):
        HipModule.__init__(self)
        if scale < 1.0:
            raise NotImplementedError("feature down-scaling (scale < 1) is not used by cvt_pyramid_axial.yaml")
        self.norm = Normalize()
        self.backbone = backbone
        assert len(self.backbone.output_shapes) == len(middle)
        cross_views, layers, downsample_layers = [], [], []
        for i, (feat_shape, num_layers) in enumerate(zip(self.backbone.output_shapes, middle)):
            _, feat_dim, feat_height, feat_width = tuple(feat_shape)
            cross_views.append(CrossViewSwapAttention(feat_height, feat_width, feat_dim, dim[i], i, **cross_view,
                                                      **cross_view_swap))
            layers.append(nn.Sequential(*[ResNetBottleNeck(dim[i]) for _ in range(num_layers)]))
            if i < len(middle) - 1:
                downsample_layers.append(nn.Sequential(_Downsample(dim[i], dim[i] // self._downsample_div, dim[i + 1])))
        self.bev_embedding = BEVEmbedding(dim[0], **bev_embedding)
        self.cross_views = nn.ModuleList(cross_views)
        self.layers = nn.ModuleList(layers)
        self.downsample_layers = nn.ModuleList(downsample_layers)
        self.self_attn = None      # commented out in the reference (:532, :556)

    def forward(self, batch):
        """batch: image (b,n,3,h,w), intrinsics (b,n,3,3), extrinsics (b,n,4,4) -> (b, d, H, W) channels-last view"""
        image = batch["image"]
        self._require_inference(image, batch["intrinsics"], batch["extrinsics"])
        b, n = image.shape[:2]
        I_inv = ops.invert_small(batch["intrinsics"].reshape(b * n, 3, 3))
        E_inv = ops.invert_small(batch["extrinsics"].reshape(b * n, 4, 4))
        feats = self.backbone(self.norm(image.flatten(0, 1)))
        feats = [rt.to_nhwc(f) for f in feats]
        return rt.nchw_view(self.forward_features(feats, I_inv, E_inv, b))
