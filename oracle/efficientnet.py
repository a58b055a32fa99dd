"""CPU restatement of the nuScenes image backbone: EfficientNetExtractor (nuscenes/cross_view_transformer/model/backbones/
efficientnet.py:24-96) over EfficientNet-B4 / B0 of efficientnet-pytorch 0.7.1 — TEST INFRASTRUCTURE ONLY (see
oracle/__init__.py).

The extractor is reference code (efficientnet.py in the reference tree); the network it wraps is third-party
(`efficientnet-pytorch==0.7.1`, nuscenes/requirements.txt:13) and absent from both the reference tree and this image, so
its arithmetic is restated here from the package's published definition and is **parity unpinned** (SURVEY.md §8c):
  * block table of EfficientNet-B0 scaled by (width, depth) = (1.4, 1.8) for B4, filters rounded to multiples of 8,
    repeats rounded up (efficientnet_pytorch/utils.py round_filters / round_repeats);
  * MBConvBlock (efficientnet_pytorch/model.py): 1x1 expand + BN + swish (skipped when expand_ratio == 1), k x k
    depthwise + BN + swish, squeeze-and-excitation (global mean, 1x1 reduce with bias, swish, 1x1 expand with bias,
    sigmoid gate), 1x1 project + BN, identity skip when stride 1 and the filter counts match (drop-connect is a
    training-time op);
  * Conv2dStaticSamePadding: TensorFlow-style "same" padding fixed at construction from the model's NOMINAL image size
    (380 for B4, 224 for B0), extra pixel at the bottom / right, applied as a zero pad before a padding-free conv;
  * BatchNorm eps 1e-3.
What the extractor itself does (and what IS reference behaviour): layers = [stem] + one group of blocks per alias below the
highest requested one, `range(idx_max)` — so the group of the highest alias is never built and asking for
reduction_2..4 returns the outputs of reduction_1..3 (efficientnet.py:62-73, forward :85-96).  state_dict keys follow
from the module tree: layers.0.0 = stem conv, layers.0.1 = stem BN, layers.<g>.<j>._expand_conv / _bn0 / _depthwise_conv /
_bn1 / _se_reduce / _se_expand / _project_conv / _bn2.
"""
import math

import torch
import torch.nn.functional as F

# efficientnet.py:8-21 (block index ranges of each alias)
MODELS = {
    "efficientnet-b0": [("reduction_1", (0, 2)), ("reduction_2", (2, 4)), ("reduction_3", (4, 6)), ("reduction_4", (6, 12))],
    "efficientnet-b4": [("reduction_1", (0, 3)), ("reduction_2", (3, 7)), ("reduction_3", (7, 11)), ("reduction_4", (11, 23))],
}
# (width, depth, nominal resolution) ; efficientnet_pytorch/utils.py efficientnet_params
PARAMS = {"efficientnet-b0": (1.0, 1.0, 224), "efficientnet-b4": (1.4, 1.8, 380)}
# B0 stages: (repeats, kernel, stride, expand, in, out) ; se_ratio 0.25 everywhere
B0_STAGES = [(1, 3, 1, 1, 32, 16), (2, 3, 2, 6, 16, 24), (2, 5, 2, 6, 24, 40), (3, 3, 2, 6, 40, 80), (3, 5, 1, 6, 80, 112),
             (4, 5, 2, 6, 112, 192), (1, 3, 1, 6, 192, 320)]
BN_EPS = 1e-3


def round_filters(filters, width, divisor=8):
    filters *= width
    new = max(divisor, int(filters + divisor / 2) // divisor * divisor)
    if new < 0.9 * filters:
        new += divisor
    return int(new)


def block_table(model_name):
    """-> (stem filters, [dict(kernel, stride, expand, cin, cout, se, image)] per block, nominal image size after the stem)"""
    width, depth, res = PARAMS[model_name]
    stem = round_filters(32, width)
    size = math.ceil(res / 2)                      # after the stride-2 stem
    blocks = []
    for (rep, k, s, e, ci, co) in B0_STAGES:
        ci, co, rep = round_filters(ci, width), round_filters(co, width), int(math.ceil(depth * rep))
        for r in range(rep):
            stride, cin = (s, ci) if r == 0 else (1, co)
            blocks.append(dict(kernel=k, stride=stride, expand=e, cin=cin, cout=co, se=max(1, int(cin * 0.25)), image=size))
            size = math.ceil(size / stride)
    return stem, blocks, res


def same_pad(image, kernel, stride):
    """(before, after) zero padding of Conv2dStaticSamePadding for a square nominal image"""
    out = math.ceil(image / stride)
    total = max((out - 1) * stride + kernel - image, 0)
    return total // 2, total - total // 2


def _bn(x, sd, prefix):
    return F.batch_norm(x, sd[prefix + "running_mean"], sd[prefix + "running_var"], sd[prefix + "weight"], sd[prefix + "bias"],
                        False, 0.0, BN_EPS)


def _swish(x):
    return x * torch.sigmoid(x)


def _conv_same(x, w, b, image, stride, groups=1):
    k = w.shape[-1]
    p0, p1 = same_pad(image, k, stride)
    return F.conv2d(F.pad(x, (p0, p1, p0, p1)), w, b, stride=stride, groups=groups)


def mbconv(x, sd, prefix, blk):
    inp = x
    if blk["expand"] != 1:
        x = _swish(_bn(F.conv2d(x, sd[prefix + "_expand_conv.weight"]), sd, prefix + "_bn0."))
    x = _conv_same(x, sd[prefix + "_depthwise_conv.weight"], None, blk["image"], blk["stride"], groups=x.shape[1])
    x = _swish(_bn(x, sd, prefix + "_bn1."))
    s = x.mean(dim=(2, 3), keepdim=True)
    s = _swish(F.conv2d(s, sd[prefix + "_se_reduce.weight"], sd[prefix + "_se_reduce.bias"]))
    s = F.conv2d(s, sd[prefix + "_se_expand.weight"], sd[prefix + "_se_expand.bias"])
    x = torch.sigmoid(s) * x
    x = _bn(F.conv2d(x, sd[prefix + "_project_conv.weight"]), sd, prefix + "_bn2.")
    if blk["stride"] == 1 and blk["cin"] == blk["cout"]:
        x = x + inp
    return x


def extractor_layout(layer_names, model_name):
    """-> (groups = [(first block, last block + 1)] actually built, idx_pick into [stem, group 0, group 1, ...])"""
    names = [k for k, _ in MODELS[model_name]]
    assert all(k in names for k in layer_names)
    idx_max = max(names.index(k) for k in layer_names)
    groups = [MODELS[model_name][i][1] for i in range(idx_max)]         # range(idx_max): the reference's off-by-one
    return groups, [names.index(k) for k in layer_names]


def efficientnet_extractor(sd, prefix, layer_names, x, model_name="efficientnet-b4"):
    """sd: flat state_dict with the extractor's keys under `prefix`; x: (N, 3, H, W) normalised images -> list of maps"""
    stem, blocks, res = block_table(model_name)
    groups, idx_pick = extractor_layout(layer_names, model_name)
    x = _conv_same(x, sd[prefix + "layers.0.0.weight"], None, res, 2)
    x = _swish(_bn(x, sd, prefix + "layers.0.1."))
    result = [x]
    for g, (lo, hi) in enumerate(groups):
        for j, bi in enumerate(range(lo, hi)):
            x = mbconv(x, sd, "%slayers.%d.%d." % (prefix, g + 1, j), blocks[bi])
        result.append(x)
    return [result[i] for i in idx_pick]
