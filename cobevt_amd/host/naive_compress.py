"""NaiveCompressor — mirror of opv2v/opencood/models/sub_modules/naive_compress.py:5-31: the channel bottleneck CorpBEVT
puts on the per-agent BEV features before they are shared when `compression > 0` (corpbevt.py:79-81,119-121): one
3x3 conv + BN + ReLU down to C / ratio channels (what would be transmitted), two 3x3 conv + BN + ReLU back up.
Sequential containers `encoder` / `decoder` with the reference's indices (state_dict keys encoder.0/1, decoder.0/1/3/4);
BatchNorm eps 1e-3.  Three launches of the 3x3 kernels with BatchNorm and ReLU folded into weights / epilogue."""
import torch.nn as nn

from .. import ops
from . import runtime as rt
from .runtime import HipModule


class NaiveCompressor(HipModule):
    def __init__(self, input_dim, compress_raito):
        super().__init__()
        mid = input_dim // compress_raito
        self.encoder = nn.Sequential(nn.Conv2d(input_dim, mid, kernel_size=3, stride=1, padding=1),
                                     nn.BatchNorm2d(mid, eps=1e-3, momentum=0.01), nn.ReLU())
        self.decoder = nn.Sequential(nn.Conv2d(mid, input_dim, kernel_size=3, stride=1, padding=1),
                                     nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01), nn.ReLU(),
                                     nn.Conv2d(input_dim, input_dim, kernel_size=3, stride=1, padding=1),
                                     nn.BatchNorm2d(input_dim, eps=1e-3, momentum=0.01), nn.ReLU())

    def forward_nhwc(self, x):
        x = ops.conv2d(x, rt.conv_plan(self, "enc", self.encoder[0], self.encoder[1], act=1))
        x = ops.conv2d(x, rt.conv_plan(self, "dec0", self.decoder[0], self.decoder[1], act=1))
        return ops.conv2d(x, rt.conv_plan(self, "dec1", self.decoder[3], self.decoder[4], act=1))

    def forward(self, x):
        """(N, C, H, W) -> (N, C, H, W) (channels-last view)"""
        if self.training:               # the differentiable graph (host/training.py): train_camera.py:143-179
            from . import training
            return training.naive_compressor(self, x)
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)
