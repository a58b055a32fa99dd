"""BevSegHead — mirror of opv2v/opencood/models/sub_modules/bev_seg_head.py (incl. its `if / if / else`
constructor quirk: target='dynamic' creates BOTH heads, :14-33).  Logits are written fp32 NCHW by the conv
kernel's epilogue."""
import torch
import torch.nn as nn

from .. import ops
from . import runtime as rt
from . import training
from .runtime import HipModule


class BevSegHead(HipModule):
    def __init__(self, target, input_dim, output_class):
        super().__init__()
        self.target = target
        if self.target == "dynamic":
            self.dynamic_head = nn.Conv2d(input_dim, output_class, kernel_size=3, padding=1)
        if self.target == "static":
            self.static_head = nn.Conv2d(input_dim, output_class, kernel_size=3, padding=1)
        else:
            self.dynamic_head = nn.Conv2d(input_dim, output_class, kernel_size=3, padding=1)
            self.static_head = nn.Conv2d(input_dim, output_class, kernel_size=3, padding=1)

    def _head(self, name, conv, x, b, l):
        y = ops.conv2d(x, rt.conv_plan(self, name, conv, store_mode=2))     # (b*l, classes, H, W) fp32
        return y.reshape(b, l, *y.shape[1:])

    def _zeros_like(self, t):
        """the all-zero map of the head the config does not train (bev_seg_head.py:34-41): one cached read-only buffer
        per shape instead of a fill kernel per frame"""
        conv = self.dynamic_head if self.target == "dynamic" else self.static_head
        return self._plan("zeros:%s" % (tuple(t.shape),), [conv.weight],
                          lambda dt, dev: torch.zeros(tuple(t.shape), device=t.device, dtype=t.dtype))

    def forward(self, x, b, l):
        """x: ((b l), C, H, W) -> {'static_seg', 'dynamic_seg'} each (b, l, classes, H, W) fp32"""
        if self.training:
            return training.bev_seg_head(self, x, b, l)
        self._require_inference(x)
        xn = rt.to_nhwc(x)
        if self.target == "dynamic":
            dynamic_map = self._head("dyn", self.dynamic_head, xn, b, l)
            static_map = self._zeros_like(dynamic_map)
        elif self.target == "static":
            static_map = self._head("sta", self.static_head, xn, b, l)
            dynamic_map = self._zeros_like(static_map)
        else:
            dynamic_map = self._head("dyn", self.dynamic_head, xn, b, l)
            static_map = self._head("sta", self.static_head, xn, b, l)
        return {"static_seg": static_map, "dynamic_seg": dynamic_map}
