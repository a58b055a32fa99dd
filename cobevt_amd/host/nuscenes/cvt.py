"""CrossViewTransformer — mirror of nuscenes/cross_view_transformer/model/cvt.py:4-40."""
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from ..runtime import HipModule


class CrossViewTransformer(HipModule):
    def __init__(self, encoder, decoder, dim_last=64, outputs={"bev": [0, 1]}):
        super().__init__()
        dim_total = 0
        dim_max = 0
        for _, (start, stop) in outputs.items():
            assert start < stop
            dim_total += stop - start
            dim_max = max(dim_max, stop)
        assert dim_max == dim_total
        self.encoder = encoder
        self.decoder = decoder
        self.outputs = outputs
        self.to_logits = nn.Sequential(
            nn.Conv2d(self.decoder.out_channels, dim_last, 3, padding=1, bias=False),
            nn.BatchNorm2d(dim_last),
            nn.ReLU(inplace=True),
            nn.Conv2d(dim_last, dim_max, 1))

    def forward(self, batch):
        x = self.encoder(batch)                                  # (b, d, H, W) channels-last view
        y = self.decoder.forward_nhwc(rt.to_nhwc(x))
        z = ops.conv2d(y, rt.conv_plan(self, "l0", self.to_logits[0], self.to_logits[1], act=1))
        z = ops.conv2d(z, rt.conv_plan(self, "l3", self.to_logits[3], store_mode=2))    # (b, dim_max, H, W) fp32
        return {k: z[:, start:stop] for k, (start, stop) in self.outputs.items()}
