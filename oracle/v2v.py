"""Oracle for the pairwise-warp fusion baselines: V2VNet and DiscoNet on the CVT per-agent encoder.  TEST INFRASTRUCTURE — see
oracle/__init__.py.

Follows opv2v/opencood/models/fusion_modules/v2v_fuse.py:16-144 (with sub_modules/convgru.py:8-78,118-176),
fusion_modules/disconet_fuse.py:16-168, torch_transformation_utils.py:77-105 (get_rotated_roi) and the model files
cross_view_transformer_v2vnet.py:13-68, cross_view_transformer_disconet.py:13-68.  Plain torch, functional over a flat state_dict.
"""
import torch
import torch.nn.functional as F

from .cvt import _decode, encode_agents
from .resnet import bn_eval
from .sttf import discretized_matrix, transformation_matrix, warp_affine


def rotated_roi(shape, correction_matrix):
    """get_rotated_roi, torch_transformation_utils.py:77-105: nearest-neighbour warp of a map of ones by the DISCRETISED pairwise
    matrices as they are (no re-centring).  shape (B, L, C, H, W); correction_matrix (B*L, 2, 3) -> (B, L, C, H, W)."""
    B, L, C, H, W = shape
    ones = torch.ones((B * L, 1, H, W), dtype=correction_matrix.dtype)
    roi = warp_affine(ones, correction_matrix, (H, W), mode="nearest")
    return torch.repeat_interleave(roi, C, dim=1).reshape(B, L, C, H, W)


def _pairwise(x, record_len, pairwise_t_matrix, args):
    """the common head of both fusions: per-sample feature lists, discretised (B, L, L, 2, 3) matrices, (B, L, L, 1, H, W) ROI"""
    _, C, H, W = x.shape
    B, L = pairwise_t_matrix.shape[:2]
    lens = [int(v) for v in record_len]
    split, off = [], 0
    for n in lens:
        split.append(x[off:off + n])
        off += n
    pm = discretized_matrix(pairwise_t_matrix.reshape(-1, L, 4, 4), args["resolution"], args["downsample_rate"]).reshape(B, L, L, 2, 3)
    roi = rotated_roi((B * L, L, 1, H, W), pm.reshape(B * L * L, 2, 3)).reshape(B, L, L, 1, H, W)
    return split, lens, pm, roi


def _neighbours(feats, pm_b, i, N, H, W):
    """features of the N agents of one sample in agent i's frame, in the reference's transposed + flipped layout
    (v2v_fuse.py:89-103): returns (flipped features (N, C, W, H), warped neighbours (N, C, H, W) of that layout)"""
    f = feats.permute(0, 1, 3, 2).flip(3)                                           # 'b c h w -> b c w h', flip
    T = transformation_matrix(pm_b[:N, i], (H, W))
    return f, warp_affine(f, T, (H, W))


def conv_gru_step(sd, pfx, x):
    """ConvGRU.forward (convgru.py:118-176) as the fusions call it: one layer, a length-1 sequence, hidden state zero
    (ConvGRUCell.forward :57-78 with h_cur = 0).  x (1, 2C, H, W) -> (1, C, H, W)."""
    c = sd[pfx + "cell_list.0.conv_can.weight"].shape[0]
    h = torch.zeros(x.shape[0], c, x.shape[2], x.shape[3], dtype=x.dtype)
    comb = torch.cat([x, h], dim=1)
    gates = F.conv2d(comb, sd[pfx + "cell_list.0.conv_gates.weight"], sd[pfx + "cell_list.0.conv_gates.bias"], padding=1)
    gamma, beta = torch.split(gates, c, dim=1)
    reset, update = torch.sigmoid(gamma), torch.sigmoid(beta)
    comb = torch.cat([x, reset * h], dim=1)
    cnm = torch.tanh(F.conv2d(comb, sd[pfx + "cell_list.0.conv_can.weight"], sd[pfx + "cell_list.0.conv_can.bias"], padding=1))
    return (1 - update) * h + update * cnm


def v2vnet_fusion(sd, pfx, args, x, record_len, pairwise_t_matrix):
    """V2VNetFusion.forward, v2v_fuse.py:47-144.  x (sum(record_len), C, H, W) -> (B, H, W, C)."""
    _, C, H, W = x.shape
    feats, lens, pm, roi = _pairwise(x, record_len, pairwise_t_matrix, args)
    for _ in range(args["num_iteration"]):
        updated = []
        for b, N in enumerate(lens):
            rows = []
            for i in range(N):
                mask = roi[b, :N, i]                                                # (N, 1, H, W), NOT transposed / flipped
                f, nb = _neighbours(feats[b], pm[b], i, N, H, W)
                ego = f[i][None].repeat(N, 1, 1, 1)
                msg = F.conv2d(torch.cat([nb, ego], dim=1), sd[pfx + "msg_cnn.weight"], sd[pfx + "msg_cnn.bias"], padding=1) * mask
                if args["agg_operator"] == "avg":
                    agg = msg.mean(dim=0)
                elif args["agg_operator"] == "max":
                    agg = msg.max(dim=0)[0]
                else:
                    raise ValueError("agg_operator has wrong value")
                if args["gru_flag"]:
                    out = conv_gru_step(sd, pfx + "conv_gru.", torch.cat([f[i], agg], dim=0)[None])[0]
                else:
                    out = f[i] + agg
                rows.append(out.flip(2).permute(0, 2, 1)[None])                     # flip, 'c w h -> c h w'
            updated.append(torch.cat(rows, dim=0))
        feats = updated
    out = torch.cat([f[0][None] for f in feats], dim=0)
    return F.linear(out.permute(0, 2, 3, 1), sd[pfx + "mlp.weight"], sd[pfx + "mlp.bias"])


def pixel_weighted_fusion(sd, pfx, x, mask):
    """PixelWeightedFusionSoftmax.forward, disconet_fuse.py:35-42: 1x1 conv + BN + ReLU x3, 1x1 conv + ReLU, masked softmax over
    the agents (dim 0)."""
    y = x
    for k in ("1_1", "1_2", "1_3"):
        y = F.relu(bn_eval(F.conv2d(y, sd[pfx + "conv%s.weight" % k], sd[pfx + "conv%s.bias" % k]), sd, pfx + "bn" + k))
    y = F.relu(F.conv2d(y, sd[pfx + "conv1_4.weight"], sd[pfx + "conv1_4.bias"]))
    if mask is not None:
        y = y.masked_fill(mask == 0, -float("inf"))
    return y.softmax(dim=0)


def disconet_fusion(sd, pfx, args, x, record_len, pairwise_t_matrix):
    """DiscoNetFusion.forward, disconet_fuse.py:82-168 (cnn / msg_cnn / conv_gru are constructed but never called there)."""
    _, C, H, W = x.shape
    feats, lens, pm, roi = _pairwise(x, record_len, pairwise_t_matrix, args)
    for _ in range(args["num_iteration"]):
        updated = []
        for b, N in enumerate(lens):
            rows = []
            for i in range(N):
                mask = roi[b, :N, i]
                f, nb = _neighbours(feats[b], pm[b], i, N, H, W)
                ego = f[i][None].repeat(N, 1, 1, 1)
                wgt = pixel_weighted_fusion(sd, pfx + "pixel_weighted_fusion.", torch.cat([nb, ego], dim=1),
                                            mask if args["use_mask"] else None)
                out = (wgt * nb * mask).sum(0)
                rows.append(out.flip(2).permute(0, 2, 1)[None])
            updated.append(torch.cat(rows, dim=0))
        feats = updated
    out = torch.cat([f[0][None] for f in feats], dim=0)
    return F.linear(out.permute(0, 2, 3, 1), sd[pfx + "mlp.weight"], sd[pfx + "mlp.bias"])


def _model(fusion, key):
    def forward(sd, config, batch):
        f = encode_agents(sd, config, batch).squeeze(1)                             # (N, C, H, W)
        fused = fusion(sd, "fusion_net.", config[key], f, batch["record_len"], batch["pairwise_t_matrix"])
        return _decode(sd, config, fused.permute(0, 3, 1, 2))
    return forward


cross_view_transformer_v2vnet_forward = _model(v2vnet_fusion, "v2vnet_fusion")          # cross_view_transformer_v2vnet.py:41-68
cross_view_transformer_disconet_forward = _model(disconet_fusion, "disconet_fusion")    # cross_view_transformer_disconet.py:41-68
