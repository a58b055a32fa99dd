"""EfficientNetExtractor — mirror of nuscenes/cross_view_transformer/model/backbones/efficientnet.py:24-96 (constructor
arguments, `.output_shapes`, `.layer_names`, `.idx_pick`, forward(x) -> list of feature maps, state_dict keys
`layers.<g>.<j>.<_expand_conv|_bn0|_depthwise_conv|_bn1|_se_reduce|_se_expand|_project_conv|_bn2>...`).

The reference wraps `efficientnet_pytorch.EfficientNet.from_pretrained(model_name)` (third-party 0.7.1, not in the reference
tree, not in this image, and a download): the network definition here is restated from the package's published one
(block table, MBConvBlock, TensorFlow-"same" static padding fixed from the nominal 380 / 224 pixel resolution, BatchNorm
eps 1e-3) — see oracle/efficientnet.py for the arithmetic and the "parity unpinned" caveat — and weights are whatever the
caller loads (a checkpoint of the reference model loads key for key) or the module's random initialisation.

What IS reference behaviour and is reproduced: layers = [stem] + the block groups of the aliases BELOW the highest one
requested (`range(idx_max)`, :62-66), every layer's output collected, `idx_pick` indexing that list (:73, :85-96) — so
asking for reduction_2..4 yields the maps of reduction_1..3, the shapes `PyramidAxialEncoder` is configured for.

Device path (channels-last inside): stem 3x3/s2 through the generic implicit GEMM (3 input channels), the 1x1 expand /
project convolutions through the dense-row GEMM with BatchNorm folded (+ swish / + identity skip in the epilogue), depthwise
k x k + BatchNorm + swish, squeeze (deterministic spatial mean), excitation and gating in csrc/depthwise.hip."""
import math

import torch
import torch.nn as nn

from ... import ops
from .. import runtime as rt
from ..runtime import HipModule

MODELS = {
    "efficientnet-b0": [("reduction_1", (0, 2)), ("reduction_2", (2, 4)), ("reduction_3", (4, 6)), ("reduction_4", (6, 12))],
    "efficientnet-b4": [("reduction_1", (0, 3)), ("reduction_2", (3, 7)), ("reduction_3", (7, 11)), ("reduction_4", (11, 23))],
}
_PARAMS = {"efficientnet-b0": (1.0, 1.0, 224), "efficientnet-b4": (1.4, 1.8, 380)}        # width, depth, nominal resolution
_B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
              (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]                               # repeats, k, stride, expand, in, out
_BN_EPS = 1e-3
_BN_MOM = 0.01            # 1 - batch_norm_momentum (0.99) of efficientnet-pytorch's global params: only the running-stat update of train() sees it
_SWISH = 3


def _round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def _same_pad(image, kernel, stride):
    total = max((math.ceil(image / stride) - 1) * stride + kernel - image, 0)
    return total // 2, total - total // 2


class MBConvBlock(HipModule):
    """Mobile inverted bottleneck with squeeze-and-excitation; attribute names are efficientnet-pytorch's (state_dict keys)."""

    def __init__(self, cin, cout, kernel, stride, expand, image):
        super().__init__()
        mid = cin * expand
        self.cin, self.cout, self.stride, self.expand = cin, cout, stride, expand
        self.pad = _same_pad(image, kernel, stride)
        if expand != 1:
            self._expand_conv = nn.Conv2d(cin, mid, 1, bias=False)
            self._bn0 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOM)
        self._depthwise_conv = nn.Conv2d(mid, mid, kernel, stride=stride, groups=mid, bias=False)
        self._bn1 = nn.BatchNorm2d(mid, eps=_BN_EPS, momentum=_BN_MOM)
        squeezed = max(1, int(cin * 0.25))
        self._se_reduce = nn.Conv2d(mid, squeezed, 1)
        self._se_expand = nn.Conv2d(squeezed, mid, 1)
        self._project_conv = nn.Conv2d(mid, cout, 1, bias=False)
        self._bn2 = nn.BatchNorm2d(cout, eps=_BN_EPS, momentum=_BN_MOM)

    def forward_nhwc(self, x):
        inp = x
        if self.expand != 1:
            x = ops.conv2d(x, rt.conv_plan(self, "expand", self._expand_conv, bn=self._bn0, act=_SWISH))
        dw = self._plan("depthwise", rt.module_tensors(self._depthwise_conv, self._bn1),
                        lambda dt, dev: ops.DepthwisePlan(self._depthwise_conv.weight, bn=self._bn1, stride=self.stride, pad=self.pad,
                                                          act=_SWISH, dtype=dt, device=dev))
        x = ops.depthwise_conv(x, dw)
        mid, sq = self._se_reduce.in_channels, self._se_reduce.out_channels
        gate = ops.se_gate(ops.spatial_mean(x),
                           rt.f32_param(self, "se.w1", self._se_reduce.weight, (sq, mid)), rt.f32_param(self, "se.b1", self._se_reduce.bias),
                           rt.f32_param(self, "se.w2", self._se_expand.weight, (mid, sq)), rt.f32_param(self, "se.b2", self._se_expand.bias))
        x = ops.channel_gate(x, gate)
        skip = inp if (self.stride == 1 and self.cin == self.cout) else None      # drop-connect is a training-time op
        return ops.conv2d(x, rt.conv_plan(self, "project", self._project_conv, bn=self._bn2), residual=skip)

    def forward(self, x, drop_connect_rate=None):
        """(N, C, H, W) -> (N, C', H', W') (channels-last view); the rate is accepted and unused, as in eval mode"""
        self._require_inference(x)
        return rt.like_input(rt.nchw_view(self.forward_nhwc(rt.to_nhwc(x))), x)


class SequentialWithArgs(nn.Sequential):
    """efficientnet.py:99-112 (the per-block drop-connect rates are training-only arguments)"""

    def __init__(self, *layers_args):
        super().__init__(*[layer for layer, _ in layers_args])
        self.args = [args for _, args in layers_args]

    def forward_nhwc(self, x):
        for blk in self:
            x = blk.forward_nhwc(x)
        return x

    def forward(self, x):
        for blk, a in zip(self, self.args):
            x = blk(x, *a)
        return x


class _Stem(nn.Sequential):
    """Sequential(conv_stem, bn0, swish) of the reference (:60) - keys layers.0.0.weight, layers.0.1.*"""

    def __init__(self, cout):
        super().__init__(nn.Conv2d(3, cout, 3, stride=2, bias=False), nn.BatchNorm2d(cout, eps=_BN_EPS, momentum=_BN_MOM), nn.Identity())


class EfficientNetExtractor(HipModule):
    def __init__(self, layer_names, image_height, image_width, model_name="efficientnet-b4"):
        super().__init__()
        assert model_name in MODELS
        names = [k for k, _ in MODELS[model_name]]
        assert all(k in names for k in layer_names)
        layer_to_idx = {k: names.index(k) for k in layer_names}
        idx_max = max(layer_to_idx.values())

        width, depth, res = _PARAMS[model_name]
        self._stem_pad = _same_pad(res, 3, 2)
        size = math.ceil(res / 2)
        table = []                                                   # every block of the network: constructor arguments
        for (rep, k, s, e, ci, co) in _B0_STAGES:
            ci, co = _round_filters(ci, width), _round_filters(co, width)
            for r in range(int(math.ceil(depth * rep))):
                stride, cin = (s, ci) if r == 0 else (1, co)
                table.append((cin, co, k, stride, e, size))
                size = math.ceil(size / stride)
        drop = 0.2 / len(table)                                      # drop_connect_rate / number of blocks (:58), unused in eval
        blocks = [_Stem(_round_filters(32, width))]
        for idx in range(idx_max):                                   # the reference's range(idx_max): see module docstring
            lo, hi = MODELS[model_name][idx][1]
            blocks.append(SequentialWithArgs(*[(MBConvBlock(*table[i]), [i * drop]) for i in range(lo, hi)]))
        self.layers = nn.Sequential(*blocks)
        self.layer_names = layer_names
        self.idx_pick = [layer_to_idx[name] for name in layer_names]
        # the reference measures the shapes with a dummy forward (:76-79); the arithmetic is static, so they are computed
        h, w = self._conv_out(image_height, self._stem_pad, 3, 2), self._conv_out(image_width, self._stem_pad, 3, 2)
        shapes = [(1, blocks[0][0].out_channels, h, w)]
        for group in blocks[1:]:
            for blk in group:
                k = blk._depthwise_conv.kernel_size[0]
                h, w = self._conv_out(h, blk.pad, k, blk.stride), self._conv_out(w, blk.pad, k, blk.stride)
            shapes.append((1, group[-1].cout, h, w))
        self.output_shapes = [torch.Size(shapes[i]) for i in self.idx_pick]

    @staticmethod
    def _conv_out(size, pad, kernel, stride):
        return (size + pad[0] + pad[1] - kernel) // stride + 1

    def forward(self, x):
        """x: (N, 3, H, W) normalised images -> [ (N, C_i, h_i, w_i) channels-last views ] in layer_names order"""
        self._require_inference(x)
        stem = self.layers[0]
        plan = self._plan("stem", rt.module_tensors(stem[0], stem[1]),
                          lambda dt, dev: ops.ConvPlan(stem[0].weight, None, bn=stem[1], stride=2, pad=self._stem_pad[0],
                                                       pad_br=self._stem_pad[1], act=_SWISH, dtype=dt, device=dev, smallc=True))
        y = ops.conv2d(ops.to_nhwc(x, torch.float32), plan)
        result = [y]
        for group in list(self.layers)[1:]:
            y = group.forward_nhwc(y)
            result.append(y)
        return [rt.nchw_view(result[i]) for i in self.idx_pick]
