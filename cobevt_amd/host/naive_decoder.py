"""NaiveDecoder — mirror of opv2v/opencood/models/sub_modules/naive_decoder.py (ModuleList `decoder` with
conv/BN/ReLU triples ordered from the deepest stage; nearest x2 up-sampling between the two convs of a stage
is folded into the second conv's gather)."""
from collections import OrderedDict

import torch.nn as nn

from .. import ops
from . import runtime as rt
from . import training
from .runtime import HipModule


class NaiveDecoder(HipModule):
    def __init__(self, params):
        super().__init__()
        self.num_ch_dec = params["num_ch_dec"]
        self.num_layer = params["num_layer"]
        self.input_dim = params["input_dim"]
        assert len(self.num_ch_dec) == self.num_layer
        self.convs = OrderedDict()
        for i in range(self.num_layer - 1, -1, -1):
            num_ch_in = self.input_dim if i == self.num_layer - 1 else self.num_ch_dec[i + 1]
            num_ch_out = self.num_ch_dec[i]
            self.convs[("upconv", i, 0)] = nn.Conv2d(num_ch_in, num_ch_out, 3, 1, 1)
            self.convs[("norm", i, 0)] = nn.BatchNorm2d(num_ch_out)
            self.convs[("relu", i, 0)] = nn.ReLU(True)
            self.convs[("upconv", i, 1)] = nn.Conv2d(num_ch_out, num_ch_out, 3, 1, 1)
            self.convs[("norm", i, 1)] = nn.BatchNorm2d(num_ch_out)
            self.convs[("relu", i, 1)] = nn.ReLU(True)
        self.decoder = nn.ModuleList(list(self.convs.values()))

    def forward_nhwc(self, x):
        for i in range(self.num_layer - 1, -1, -1):
            x = ops.conv2d(x, rt.conv_plan(self, "u%d0" % i, self.convs[("upconv", i, 0)], self.convs[("norm", i, 0)], act=1))
            x = ops.conv2d(x, rt.conv_plan(self, "u%d1" % i, self.convs[("upconv", i, 1)], self.convs[("norm", i, 1)], act=1,
                                           upsample=True))
        return x

    def forward(self, x):
        """(B, L, C1, H, W) -> (B, L, C2, 8H, 8W)"""
        if self.training:
            b, l = x.shape[:2]
            y = training.naive_decoder(self, x.reshape(b * l, *x.shape[2:]))
            return y.reshape(b, l, *y.shape[1:])
        self._require_inference(x)
        b, l, c, h, w = x.shape
        y = rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x.reshape(b * l, c, h, w))))
        return rt.like_input(y.reshape(b, l, *y.shape[1:]), x)
