"""Oracle for the model-level forwards.  TEST INFRASTRUCTURE — see oracle/__init__.py.

CorpBEVT (opv2v/opencood/models/corpbevt.py:104-145), FaxFusedTransformer
(opv2v/opencood/models/fax_fused_transformer.py:34-48), NaiveDecoder (sub_modules/naive_decoder.py:62-91),
BevSegHead (sub_modules/bev_seg_head.py:35-61) and the mIoU metric (opv2v/opencood/utils/seg_utils.py:25-51).
"""
import numpy as np
import torch
import torch.nn.functional as F

from .fax import fax_module
from .resnet import bn_eval, resnet_encoder
from .sttf import regroup, roi_and_cav_mask, sttf
from .swap_fusion import swap_fusion_encoder


def naive_decoder(sd, pfx, params, x):
    """naive_decoder.py:62-91 — x (B, L, C, H, W).  ModuleList order: for stage i = num_layer-1 .. 0:
    [conv, bn, relu, conv, bn, relu] -> indices 6*s + {0,1,3,4}, s = 0 for the deepest stage."""
    b, l, c, h, w = x.shape
    x = x.reshape(b * l, c, h, w)
    for s in range(params["num_layer"]):
        k = pfx + "decoder.%d"
        x = F.conv2d(x, sd[(k % (6 * s)) + ".weight"], sd[(k % (6 * s)) + ".bias"], padding=1)
        x = F.relu(bn_eval(x, sd, k % (6 * s + 1)))
        x = F.interpolate(x, scale_factor=2, mode="nearest")
        x = F.conv2d(x, sd[(k % (6 * s + 3)) + ".weight"], sd[(k % (6 * s + 3)) + ".bias"], padding=1)
        x = F.relu(bn_eval(x, sd, k % (6 * s + 4)))
    return x.reshape(b, l, *x.shape[1:])


def bev_seg_head(sd, pfx, target, x, b, l):
    """bev_seg_head.py:35-61 — x ((b l), C, H, W) -> {'static_seg','dynamic_seg'} each (b, l, classes, H, W)."""
    def head(name):
        y = F.conv2d(x, sd[pfx + name + ".weight"], sd[pfx + name + ".bias"], padding=1)
        return y.reshape(b, l, *y.shape[1:])
    if target == "dynamic":
        dyn = head("dynamic_head")
        return {"static_seg": torch.zeros_like(dyn), "dynamic_seg": dyn}
    if target == "static":
        sta = head("static_head")
        return {"static_seg": sta, "dynamic_seg": torch.zeros_like(sta)}
    return {"static_seg": head("static_head"), "dynamic_seg": head("dynamic_head")}


def naive_compressor(sd, prefix, x):
    """NaiveCompressor.forward, sub_modules/naive_compress.py:5-31: (conv3x3 + BN(eps 1e-3) + ReLU) x 3, C -> C/r -> C -> C."""
    for conv, bn in (("encoder.0", "encoder.1"), ("decoder.0", "decoder.1"), ("decoder.3", "decoder.4")):
        x = F.conv2d(x, sd[prefix + conv + ".weight"], sd[prefix + conv + ".bias"], padding=1)
        x = F.relu(F.batch_norm(x, sd[prefix + bn + ".running_mean"], sd[prefix + bn + ".running_var"], sd[prefix + bn + ".weight"],
                                sd[prefix + bn + ".bias"], False, 0.0, 1e-3))
    return x


def encode_agents(sd, config, batch, taps=None):
    """Per-agent part of CorpBEVT.forward (corpbevt.py:112-117): encoder + FAX -> (N, C, H, W).  taps: optional dict that
    receives the per-level FAX outputs (tests compare intermediate tensors, not only the logits)."""
    feats = resnet_encoder(sd, "encoder.encoder.", config["encoder"], batch["inputs"])
    f = fax_module(sd, "fax.", config["fax"], feats, batch["intrinsic"], batch["extrinsic"], taps=taps)
    return f.squeeze(1)


def fuse_and_decode(sd, config, f, tm, record_len, return_intermediates=False):
    """Cross-agent part of CorpBEVT.forward (corpbevt.py:119-145): regroup, STTF, mask, swap fusion, decoder, head."""
    if config["compression"] > 0:                         # corpbevt.py:119-121
        f = naive_compressor(sd, "naive_compressor.", f)
    g, mask = regroup(f, record_len, config["max_cav"])
    st = config["sttf"]
    w = sttf(g, tm, st["resolution"], st["downsample_rate"])                       # b l h w c
    if st["use_roi_mask"]:
        com_mask = roi_and_cav_mask(w.shape, mask, tm, st["resolution"], st["downsample_rate"])
    else:
        com_mask = mask[:, None, None, None, :].to(w.dtype)
    fused = swap_fusion_encoder(sd, "fusion_net.", config["fax_fusion"], w.permute(0, 1, 4, 2, 3), com_mask)
    y = naive_decoder(sd, "decoder.", config["decoder"], fused[:, None])
    yb = y.reshape(-1, *y.shape[2:])
    out = bev_seg_head(sd, "seg_head.", config["target"], yb, yb.shape[0], 1)
    if return_intermediates:
        out = dict(out)
        out.update({"fax": f, "regroup": g, "cav_mask": mask, "sttf": w, "com_mask": com_mask, "fused": fused})
    return out


def corpbevt_forward(sd, config, batch, return_intermediates=False):
    """CorpBEVT.forward, corpbevt.py:104-145."""
    taps = {} if return_intermediates else None
    f = encode_agents(sd, config, batch, taps=taps)
    out = fuse_and_decode(sd, config, f, batch["transformation_matrix"], batch["record_len"], return_intermediates)
    if return_intermediates:
        out.update({"fax_" + k: v for k, v in taps.items()})
    return out


def fax_fused_transformer_forward(sd, config, batch):
    """FaxFusedTransformer.forward, fax_fused_transformer.py:34-48."""
    x = batch["inputs"]
    b, l = x.shape[:2]
    feats = resnet_encoder(sd, "encoder.encoder.", config["encoder"], x)
    f = fax_module(sd, "fax.", config["fax"], feats, batch["intrinsic"], batch["extrinsic"])
    y = naive_decoder(sd, "decoder.", config["decoder"], f)
    yb = y.reshape(-1, *y.shape[2:])
    return bev_seg_head(sd, "seg_head.", config["target"], yb, b, l)


def mean_iu(eval_segm, gt_segm):
    """mean_IU, opv2v/opencood/utils/seg_utils.py:25-51 — per-class IoU list over the union of classes present
    in prediction and ground truth (sorted); a class missing from either map scores 0."""
    eval_segm, gt_segm = np.asarray(eval_segm), np.asarray(gt_segm)
    assert eval_segm.shape == gt_segm.shape
    classes = np.union1d(np.unique(eval_segm), np.unique(gt_segm))
    ious = []
    for c in classes:
        e, g = eval_segm == c, gt_segm == c
        if e.sum() == 0 or g.sum() == 0:
            ious.append(0.0)
            continue
        n_ii = np.logical_and(e, g).sum()
        ious.append(float(n_ii) / float(g.sum() + e.sum() - n_ii))
    return ious
