"""Decoder / DecoderBlock of the nuScenes model (nuscenes/cross_view_transformer/model/decoder.py:6-61): each block doubles
the BEV map (bilinear, align_corners) -> 3x3 conv + BN + ReLU -> 1x1 conv + BN, adds a 1x1 projection of the decoder's
INPUT map resized (nearest) to the new size, ReLU.  Same constructor arguments and state_dict keys (`layers.<i>.conv.1/2/4/5`,
`layers.<i>.up`).  On the device: resize kernel, 3x3 kernel, and the 1x1 through the dense-row GEMM with the skip branch as
its residual and the final ReLU in its epilogue."""
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from ..runtime import HipModule


class DecoderBlock(HipModule):
    def __init__(self, in_channels, out_channels, skip_dim, residual, factor):
        super().__init__()
        mid = out_channels // factor
        stages = [nn.Upsample(scale_factor=2, mode="bilinear", align_corners=True),
                  nn.Conv2d(in_channels, mid, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(mid), nn.ReLU(inplace=True),
                  nn.Conv2d(mid, out_channels, kernel_size=1, padding=0, bias=False), nn.BatchNorm2d(out_channels)]
        self.conv = nn.Sequential(*stages)
        self.up = nn.Conv2d(skip_dim, out_channels, kernel_size=1) if residual else None
        self.relu = nn.ReLU(inplace=True)

    def forward_nhwc(self, x, skip):
        _, h, w, _ = x.shape
        big = ops.resize_nhwc(x, 2 * h, 2 * w, "bilinear")
        hidden = ops.conv2d(big, rt.conv_plan(self, "c1", self.conv[1], self.conv[2], act=1))
        branch = None
        if self.up is not None:                      # F.interpolate(up(skip), size) with the default nearest mode
            branch = ops.resize_nhwc(ops.conv2d(skip, rt.conv_plan(self, "up", self.up)), 2 * h, 2 * w, "nearest")
        return ops.conv2d(hidden, rt.conv_plan(self, "c4", self.conv[4], self.conv[5], act=1), residual=branch)

    def forward(self, x, skip):
        self._require_inference(x, skip)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x), rt.to_nhwc(skip))), x)


class Decoder(HipModule):
    def __init__(self, dim, blocks, residual=True, factor=2):
        super().__init__()
        widths = [dim] + list(blocks)
        self.layers = nn.Sequential(*[DecoderBlock(cin, cout, dim, residual, factor) for cin, cout in zip(widths[:-1], widths[1:])])
        self.out_channels = widths[-1]

    def forward_nhwc(self, x):
        y = x
        for block in self.layers:                    # every block's skip branch reads the decoder INPUT
            y = block.forward_nhwc(y, x)
        return y

    def forward(self, x):
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)
