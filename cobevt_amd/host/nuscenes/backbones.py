"""Backbone contract of PyramidAxialEncoder: any module with `.output_shapes` (list of (1, C, h, w)) whose forward
maps normalised images (b*n, 3, H, W) to that list of feature maps (efficientnet.py:24-110).

The real backbone of the shipped config is EfficientNetExtractor (efficientnet.py in this package: EfficientNet-B4 of
efficientnet-pytorch 0.7.1 restated from its published definition, "parity unpinned" for that third-party arithmetic).
FeatureMapBackbone returns fixed feature maps of the shapes the shipped config produces (the extractor's outputs at
224x480: (32,56,120), (56,28,60), (112,14,30)); it is what the reference-generated fixture gv11 was made with (the
reference itself could only be run with stand-in features here), so the FAX encoder / decoder can be replayed against the
reference's own outputs."""
import torch
import torch.nn as nn

NUSCENES_B4_SHAPES = [(1, 32, 56, 120), (1, 56, 28, 60), (1, 112, 14, 30)]


class FeatureMapBackbone(nn.Module):
    def __init__(self, features):
        """features: list of (b*n, C, h, w) tensors returned verbatim by forward."""
        super().__init__()
        self.output_shapes = [torch.Size((1,) + tuple(f.shape[1:])) for f in features]
        for i, f in enumerate(features):
            self.register_buffer("feature%d" % i, f, persistent=False)

    def forward(self, x):
        return [getattr(self, "feature%d" % i) for i in range(len(self.output_shapes))]
