"""Oracle for the nuScenes SinBEVT flavour.  TEST INFRASTRUCTURE — see oracle/__init__.py.

Follows nuscenes/cross_view_transformer/model/encoder_pyramid_axial.py:475-558 (PyramidAxialEncoder; the FAX operators
themselves are the ones restated in oracle/fax.py — the two reference trees share them),
decoder.py:6-61 and cvt.py:4-40.  The backbone is an input (feature maps), EfficientNet is not restated.
"""
import torch
import torch.nn.functional as F

from .fax import fax_module
from .resnet import bn_eval


def normalize(image, mean=(0.485, 0.456, 0.406), std=(0.229, 0.224, 0.225)):
    """Normalize, encoder_pyramid_axial.py:41-49"""
    m = torch.tensor(mean)[None, :, None, None]
    s = torch.tensor(std)[None, :, None, None]
    return (image - m) / s


def pyramid_axial_encoder(sd, pfx, cfg, features, intrinsics, extrinsics):
    """PyramidAxialEncoder.forward (:534-558) after the backbone.  features: list of (b*n, C, h, w); intrinsics (b,n,3,3);
    extrinsics (b,n,4,4) -> (b, d, H, W).  cfg: dict(dim, middle, cross_view, cross_view_swap, bev_embedding, self_attn)."""
    b, n = intrinsics.shape[:2]
    feats = [f.reshape(b, 1, n, *f.shape[1:]) for f in features]
    x = fax_module(sd, pfx, cfg, feats, intrinsics[:, None], extrinsics[:, None], invert_extrinsic=True,
                   final_self_attn=False)
    return x[:, 0]


def decoder_block(sd, pfx, x, skip, residual=True):
    """DecoderBlock.forward, decoder.py:27-36"""
    y = F.interpolate(x, scale_factor=2, mode="bilinear", align_corners=True)
    y = F.relu(bn_eval(F.conv2d(y, sd[pfx + "conv.1.weight"], padding=1), sd, pfx + "conv.2"))
    y = bn_eval(F.conv2d(y, sd[pfx + "conv.4.weight"]), sd, pfx + "conv.5")
    if residual:
        up = F.conv2d(skip, sd[pfx + "up.weight"], sd[pfx + "up.bias"])
        up = F.interpolate(up, y.shape[-2:])
        y = y + up
    return F.relu(y)


def decoder(sd, pfx, nblocks, x, residual=True):
    """Decoder.forward, decoder.py:55-61"""
    y = x
    for i in range(nblocks):
        y = decoder_block(sd, "%slayers.%d." % (pfx, i), y, x, residual)
    return y


def cross_view_transformer(sd, cfg, nblocks, outputs, features, intrinsics, extrinsics):
    """CrossViewTransformer.forward, cvt.py:35-40 (encoder -> decoder -> to_logits -> channel slices)."""
    x = pyramid_axial_encoder(sd, "encoder.", cfg, features, intrinsics, extrinsics)
    y = decoder(sd, "decoder.", nblocks, x)
    z = F.relu(bn_eval(F.conv2d(y, sd["to_logits.0.weight"], padding=1), sd, "to_logits.1"))
    z = F.conv2d(z, sd["to_logits.3.weight"], sd["to_logits.3.bias"])
    return {k: z[:, a:b] for k, (a, b) in outputs.items()}
