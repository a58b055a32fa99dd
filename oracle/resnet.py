"""Oracle for the image encoder.  TEST INFRASTRUCTURE — see oracle/__init__.py.

ResnetEncoder (opv2v/opencood/models/backbones/resnet_ms.py:46-89) wraps torchvision's ResNet, which is NOT in
the reference tree (torchvision 0.12.0 per opv2v/README.md:21).  BasicBlock / Bottleneck / the stem are
restated here from torchvision's public definition (He et al. 2015; conv-bn-relu ordering, stride on the
first 3x3 of a BasicBlock and on the 3x3 of a Bottleneck ["ResNet v1.5"], 1x1-conv+BN downsample on the
identity when shape changes).  Parity with torchvision's own arithmetic is unpinned (library absent); the
golden vectors use a stand-in with this same structure.
"""
import torch
import torch.nn.functional as F

RESNET_BLOCKS = {18: [2, 2, 2, 2], 34: [3, 4, 6, 3]}


def bn_eval(x, sd, key, eps=1e-5):
    """BatchNorm2d in eval mode with running statistics."""
    return F.batch_norm(x, sd[key + ".running_mean"], sd[key + ".running_var"], sd[key + ".weight"],
                        sd[key + ".bias"], False, 0.0, eps)


def bottleneck_forward(sd, pfx, x, stride=1):
    """torchvision Bottleneck(inplanes, planes) used as ResNetBottleNeck(c) = Bottleneck(c, c // 4)
    (fax_modules.py:10,472): 1x1 -> BN -> ReLU -> 3x3 -> BN -> ReLU -> 1x1 -> BN -> (+identity) -> ReLU."""
    identity = x
    out = F.relu(bn_eval(F.conv2d(x, sd[pfx + "conv1.weight"]), sd, pfx + "bn1"))
    out = F.relu(bn_eval(F.conv2d(out, sd[pfx + "conv2.weight"], stride=stride, padding=1), sd, pfx + "bn2"))
    out = bn_eval(F.conv2d(out, sd[pfx + "conv3.weight"]), sd, pfx + "bn3")
    if (pfx + "downsample.0.weight") in sd:
        identity = bn_eval(F.conv2d(x, sd[pfx + "downsample.0.weight"], stride=stride), sd, pfx + "downsample.1")
    return F.relu(out + identity)


def basic_block_forward(sd, pfx, x, stride):
    identity = x
    out = F.relu(bn_eval(F.conv2d(x, sd[pfx + "conv1.weight"], stride=stride, padding=1), sd, pfx + "bn1"))
    out = bn_eval(F.conv2d(out, sd[pfx + "conv2.weight"], padding=1), sd, pfx + "bn2")
    if (pfx + "downsample.0.weight") in sd:
        identity = bn_eval(F.conv2d(x, sd[pfx + "downsample.0.weight"], stride=stride), sd, pfx + "downsample.1")
    return F.relu(out + identity)


def resnet_encoder(sd, pfx, cfg, input_images):
    """ResnetEncoder.forward, resnet_ms.py:46-89.  input (B, L, M, H, W, 3) channels-last ->
    [(B, L, M, C, h, w)] picked by cfg['id_pick'].  pfx is the key prefix of the torchvision model
    (e.g. 'encoder.encoder.')."""
    b, l, m, h, w, c = input_images.shape
    x = input_images.reshape(b * l * m, h, w, c).permute(0, 3, 1, 2).contiguous()
    x = F.relu(bn_eval(F.conv2d(x, sd[pfx + "conv1.weight"], stride=2, padding=3), sd, pfx + "bn1"))
    x = F.max_pool2d(x, kernel_size=3, stride=2, padding=1)
    results = []
    for li, nblocks in enumerate(RESNET_BLOCKS[cfg["num_layers"]]):
        for j in range(nblocks):
            stride = 2 if (li > 0 and j == 0) else 1
            x = basic_block_forward(sd, "%slayer%d.%d." % (pfx, li + 1, j), x, stride)
        results.append(x.reshape(b, l, m, *x.shape[1:]))
    pick = cfg["id_pick"]
    if isinstance(pick, list):
        return [results[i] for i in pick]
    return results[pick]
