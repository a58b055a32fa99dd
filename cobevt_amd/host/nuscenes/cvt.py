"""CrossViewTransformer — the nuScenes model wrapper (nuscenes/cross_view_transformer/model/cvt.py:4-40): encoder -> decoder ->
a 3x3 conv + BN + ReLU + 1x1 conv head whose output channels are sliced into named maps.  Same constructor arguments and
state_dict keys (`to_logits.0 / .1 / .3`); the head runs as two launches on the decoder's channels-last output, the logits
are written planar fp32 by the second one."""
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from .. import training
from ..runtime import HipModule


def _head_width(outputs):
    """The named slices must tile [0, width) without gaps or overlaps (the reference checks sum of widths == largest stop)."""
    widths = [stop - start for start, stop in outputs.values()]
    top = max(stop for _, stop in outputs.values())
    if min(widths) <= 0 or sum(widths) != top:
        raise AssertionError("outputs %r do not partition the head's channels" % (outputs,))
    return top


class CrossViewTransformer(HipModule):
    def __init__(self, encoder, decoder, dim_last=64, outputs={"bev": [0, 1]}):
        super().__init__()
        width = _head_width(outputs)
        self.encoder, self.decoder, self.outputs = encoder, decoder, outputs
        head = [nn.Conv2d(decoder.out_channels, dim_last, kernel_size=3, padding=1, bias=False), nn.BatchNorm2d(dim_last),
                nn.ReLU(inplace=True), nn.Conv2d(dim_last, width, kernel_size=1)]
        self.to_logits = nn.Sequential(*head)

    def forward(self, batch):
        """batch: image / intrinsics / extrinsics -> {name: (b, stop - start, H, W) fp32 logits}"""
        if self.training:                   # model_module.py:35-60 training_step: the differentiable graph of host/training.py
            return training.nusc_cross_view_transformer(self, batch)
        bev = self.decoder.forward_nhwc(rt.to_nhwc(self.encoder(batch)))
        hidden = ops.conv2d(bev, rt.conv_plan(self, "l0", self.to_logits[0], self.to_logits[1], act=1))
        logits = ops.conv2d(hidden, rt.conv_plan(self, "l3", self.to_logits[3], store_mode=2))       # (b, width, H, W) fp32
        return {name: logits[:, lo:hi] for name, (lo, hi) in self.outputs.items()}
