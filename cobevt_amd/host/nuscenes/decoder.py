"""Decoder / DecoderBlock — mirror of nuscenes/cross_view_transformer/model/decoder.py:6-61."""
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from ..runtime import HipModule


class DecoderBlock(HipModule):
    def __init__(self, in_channels, out_channels, skip_dim, residual, factor):
        super().__init__()
        dim = out_channels // factor
        self.conv = nn.Sequential(
            nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
            nn.Conv2d(in_channels, dim, 3, padding=1, bias=False),
            nn.BatchNorm2d(dim),
            nn.ReLU(inplace=True),
            nn.Conv2d(dim, out_channels, 1, padding=0, bias=False),
            nn.BatchNorm2d(out_channels))
        self.up = nn.Conv2d(skip_dim, out_channels, 1) if residual else None
        self.relu = nn.ReLU(inplace=True)

    def forward_nhwc(self, x, skip):
        n, h, w, _ = x.shape
        y = ops.resize_nhwc(x, 2 * h, 2 * w, "bilinear")
        y = ops.conv2d(y, rt.conv_plan(self, "c1", self.conv[1], self.conv[2], act=1))
        up = None
        if self.up is not None:
            up = ops.conv2d(skip, rt.conv_plan(self, "up", self.up))
            up = ops.resize_nhwc(up, 2 * h, 2 * w, "nearest")
        return ops.conv2d(y, rt.conv_plan(self, "c4", self.conv[4], self.conv[5], act=1), residual=up)

    def forward(self, x, skip):
        self._require_inference(x, skip)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x), rt.to_nhwc(skip))), x)


class Decoder(HipModule):
    def __init__(self, dim, blocks, residual=True, factor=2):
        super().__init__()
        layers = []
        channels = dim
        for out_channels in blocks:
            layers.append(DecoderBlock(channels, out_channels, dim, residual, factor))
            channels = out_channels
        self.layers = nn.Sequential(*layers)
        self.out_channels = channels

    def forward_nhwc(self, x):
        y = x
        for layer in self.layers:
            y = layer.forward_nhwc(y, x)
        return y

    def forward(self, x):
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)
