"""Hyper-parameters + input recipes of the golden-vector cases.

Shared by make_golden.py (runs the REFERENCE, only possible in the build container) and by the tests (run the
oracle / the HIP modules anywhere).  Inputs are procedural (cobevt_amd.synth.procedural_input), weights are
procedural (cobevt_amd.synth.fill_module_), so a fixture file only needs the expected outputs.
"""
import math

import numpy as np
import torch

from cobevt_amd import synth
from cobevt_amd.synth import procedural_input

SEED = 0

# GV1 — (H, W, w1, w2) pairs that occur in the configs (OPV2V, nuScenes incl. padded maps) + odd shapes
INDEX_MAP_SHAPES = [
    (128, 128, 16, 16), (64, 64, 16, 16), (32, 32, 32, 32), (64, 64, 8, 8), (32, 32, 8, 8), (16, 16, 16, 16),
    (100, 100, 10, 10), (50, 50, 10, 10), (25, 25, 25, 25), (60, 120, 6, 12), (30, 60, 6, 12), (14, 30, 14, 30),
    (24, 16, 8, 8), (18, 24, 6, 12), (8, 8, 4, 4),
]
REL_POS_3D = [(3, 4), (5, 8)]          # (agent_size, window_size)
REL_POS_2D = [8, 32]

# GV2 — CrossWinAttention
CROSS_WIN = {
    "a": dict(dim=32, heads=1, dim_head=32, qkv_bias=True, q=(1, 1, 2, 3, 4, 4, 32), kv=(1, 1, 2, 3, 2, 3, 32), skip=True),
    "b": dict(dim=128, heads=4, dim_head=32, qkv_bias=True, q=(2, 3, 2, 2, 4, 4, 128), kv=(2, 3, 2, 2, 3, 6, 128), skip=False),
    "c": dict(dim=128, heads=4, dim_head=32, qkv_bias=False, q=(1, 1, 1, 1, 8, 8, 128), kv=(1, 1, 1, 1, 4, 4, 128), skip=True),
}


def cross_win_inputs(name):
    c = CROSS_WIN[name]
    q = procedural_input("gv2.%s.q" % name, c["q"], SEED)
    k = procedural_input("gv2.%s.k" % name, c["kv"], SEED)
    v = procedural_input("gv2.%s.v" % name, c["kv"], SEED)
    skip = procedural_input("gv2.%s.skip" % name, (c["q"][0],) + tuple(c["q"][2:]), SEED) if c["skip"] else None
    return q, k, v, skip


def camera_geometry(bn, n, image_h, image_w):
    """Pin-hole intrinsics + yawed camera->ego extrinsics, (bn//n, n, ...) shaped fp32."""
    f = image_w / 2.0
    intr = np.array([[f, 0, image_w / 2.0], [0, f, image_h / 2.0], [0, 0, 1]], dtype=np.float64)
    axes = np.array([[0, 0, 1, 0], [-1, 0, 0, 0], [0, -1, 0, 0], [0, 0, 0, 1]], dtype=np.float64)
    ext = np.zeros((bn // n, n, 4, 4))
    for b in range(bn // n):
        for k in range(n):
            a = math.radians(35.0 + 97.0 * k + 11.0 * b)
            rz = np.array([[math.cos(a), -math.sin(a), 0, 0], [math.sin(a), math.cos(a), 0, 0], [0, 0, 1, 0], [0, 0, 0, 1]])
            t = np.eye(4)
            t[:3, 3] = (1.2 + 0.1 * k, 0.3 * b, 1.6)
            ext[b, k] = rz @ t @ axes
    I = np.broadcast_to(intr, (bn // n, n, 3, 3)).copy()
    return torch.from_numpy(I.astype(np.float32)), torch.from_numpy(ext.astype(np.float32))


# GV3 — CrossViewSwapAttention
CVSA = {
    "plumbing": dict(
        b=1, n=1, feat=(56, 28, 28), dim=128, index=0, bev=(64, 64), image=(224, 224),
        bev_embedding=dict(sigma=1.0, bev_height=64, bev_width=64, h_meters=100, w_meters=100, offset=0.0,
                           upsample_scales=[1]),
        kwargs=dict(qkv_bias=True, q_win_size=[[16, 16]], feat_win_size=[[7, 7]], heads=[4], dim_head=[32],
                    bev_embedding_flag=[True], rel_pos_emb=False, no_image_features=False, skip=True)),
    "padded": dict(
        b=2, n=2, feat=(32, 14, 15), dim=64, index=1, bev=(24, 16), image=(112, 120),
        bev_embedding=dict(sigma=1.0, bev_height=24, bev_width=16, h_meters=50, w_meters=40, offset=0.0,
                           upsample_scales=[1, 1]),
        kwargs=dict(qkv_bias=True, q_win_size=[[8, 8], [8, 8]], feat_win_size=[[6, 12], [6, 12]], heads=[2, 2],
                    dim_head=[32, 32], bev_embedding_flag=[True, False], rel_pos_emb=False, no_image_features=False,
                    skip=True)),
}


def cvsa_inputs(name):
    c = CVSA[name]
    x = procedural_input("gv3.%s.x" % name, (c["b"], c["dim"]) + tuple(c["bev"]), SEED)
    feat = procedural_input("gv3.%s.feature" % name, (c["b"], c["n"]) + tuple(c["feat"]), SEED)
    I, E = camera_geometry(c["b"] * c["n"], c["n"], *c["image"])
    return x, feat, I.inverse(), E


# GV4 — FAXModule (reduced; verified to run on the reference)
FAX_SMALL = dict(
    b=1, l=2, n=2, image=64,
    config=dict(
        dim=[32, 32, 32], middle=[2, 2, 2],
        backbone_output_shape=[(1, 1, 1, 16, 8, 8), (1, 1, 1, 32, 4, 4), (1, 1, 1, 64, 2, 2)],
        bev_embedding=dict(sigma=1.0, bev_height=32, bev_width=32, h_meters=100, w_meters=100, offset=0.0,
                           upsample_scales=[2, 4, 8]),
        cross_view=dict(image_height=64, image_width=64, no_image_features=False, skip=True, heads=[1, 1, 1],
                        dim_head=[32, 32, 32], qkv_bias=True),
        cross_view_swap=dict(rel_pos_emb=False, q_win_size=[[4, 4], [4, 4], [4, 4]],
                             feat_win_size=[[2, 2], [2, 2], [2, 2]], bev_embedding_flag=[True, False, False]),
        self_attn=dict(dim_head=32, dropout=0.1, window_size=4)))


def fax_small_inputs():
    c = FAX_SMALL
    feats = [procedural_input("gv4.feature%d" % i, (c["b"], c["l"], c["n"]) + tuple(s[3:]), SEED)
             for i, s in enumerate(c["config"]["backbone_output_shape"])]
    I, E = camera_geometry(c["b"] * c["l"] * c["n"], c["n"], c["image"], c["image"])
    shape5 = (c["b"], c["l"], c["n"])
    return {"inputs": torch.zeros(shape5 + (c["image"], c["image"], 3)), "features": feats,
            "intrinsic": I.reshape(shape5 + (3, 3)), "extrinsic": E.reshape(shape5 + (4, 4))}


# GV5 — swap fusion
SWAP = dict(dim=64, dim_head=32, agent_size=3, window_size=4, mlp_dim=128, depth=2, b=2, hw=8)


def swap_inputs():
    c = SWAP
    x = procedural_input("gv5.x", (c["b"], c["agent_size"], c["dim"], c["hw"], c["hw"]), SEED)
    mask = torch.ones(c["b"], c["hw"], c["hw"], 1, c["agent_size"])
    mask[1, :, :, :, 2] = 0                       # a padded agent in sample 1
    mask[0, :3, :, :, 1] = 0                      # a partially visible ROI for agent 1 of sample 0
    mask[0, :, 6:, :, 2] = 0
    return x, mask


# GV6 — regroup / STTF / ROI mask
STTF = dict(resolution=1.5625, downsample_rate=8, C=8)


def sttf_inputs(h, w):
    L = 4
    x = procedural_input("gv6.x.%dx%d" % (h, w), (1, L, STTF["C"], h, w), SEED)
    cell = STTF["resolution"] * STTF["downsample_rate"]
    tm = np.tile(np.eye(4), (1, L, 1, 1))
    tm[0, 1, 0, 3] = cell                          # +1 cell along x
    a = math.radians(10.0)
    tm[0, 2, :2, :2] = [[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]]
    tm[0, 2, :2, 3] = (2.3 * cell, -1.4 * cell)
    a = math.radians(-33.0)
    tm[0, 3, :2, :2] = [[math.cos(a), -math.sin(a)], [math.sin(a), math.cos(a)]]
    tm[0, 3, :2, 3] = (-0.6 * cell, 3.2 * cell)
    cav = torch.tensor([[1, 1, 1, 0]], dtype=torch.int64)
    return x, torch.from_numpy(tm.astype(np.float32)), cav


# GV7 — decoder + head
DECODER = dict(input_dim=32, num_layer=3, num_ch_dec=[8, 16, 32])

# GV9 — FAX global attention
GLOBAL_ATTN = dict(dim=64, dim_head=32, window_size=8, b=2)

# GV10 — ResnetEncoder
RESNET = {18: dict(num_layers=18, pretrained=False, image_width=64, image_height=64, id_pick=[1, 2, 3]),
          34: dict(num_layers=34, pretrained=False, image_width=64, image_height=64, id_pick=[1, 2, 3])}


# GV11 — nuScenes SinBEVT (BASELINE config[1] shapes: 6 cams, features of EfficientNet-B4 reduction_2..4 at 224x480,
# 200x200 BEV, config/model/cvt_pyramid_axial.yaml) with the backbone replaced by fixed feature maps
NUSCENES = synth.nuscenes_config()


def nuscenes_inputs():
    return synth.nuscenes_inputs("gv11", SEED)


# ---- data formats either side of the hot path: pre-processor, collate, post-processor, scores ----
PRE_POST = dict(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225], image_hw=(12, 16), seg_hw=(24, 20))


def pre_post_inputs():
    """Deterministic inputs of gv12 (numpy RandomState is stable across numpy versions)."""
    import numpy as np
    rs = np.random.RandomState(1234)
    h, w = PRE_POST["image_hw"]
    sh, sw = PRE_POST["seg_hw"]
    out = {"image_u8": rs.randint(0, 256, size=(h, w, 3)).astype(np.uint8)}
    # logits with exact ties, near ties, large magnitudes and a constant pixel
    for name, c in (("static", 3), ("dynamic", 2)):
        x = (rs.standard_normal((2, c, sh, sw)) * 3.0).astype(np.float32)
        x[0, :, 0, 0] = 1.25                    # all classes tie -> class 0
        x[0, 1:, 0, 1] = 2.5
        x[0, 0, 0, 1] = -1.0                    # classes 1.. tie -> class 1
        x[1, :, 1, 0] = np.float32(80.0)        # large equal logits
        x[1, 0, 1, 1], x[1, c - 1, 1, 1] = np.float32(1.0), np.nextafter(np.float32(1.0), np.float32(2.0))   # one ulp apart
        x[1, 0, 2, 2], x[1, 1, 2, 2] = np.float32(-90.0), np.float32(60.0)                                   # exp underflow
        out[name + "_logits"] = x
    # label maps: road / lane renderings and prediction / ground-truth pairs (one with a class missing from the prediction)
    out["road"] = (rs.rand(sh, sw) > 0.5).astype(np.float64)
    out["lane"] = (rs.rand(sh, sw) > 0.8).astype(np.float64)
    out["bev_bgr"] = (rs.randint(0, 256, size=(sh, sw, 3)) * (rs.rand(sh, sw, 1) > 0.6)).astype(np.uint8)
    out["bev_bgr"][0, :6] = [[0, 0, 0], [4, 0, 0], [5, 0, 0], [1, 0, 1], [2, 0, 1], [0, 1, 0]]     # around the gray > 0 threshold
    pairs = []
    for k, ncls in enumerate((3, 2, 3, 3)):
        pred = rs.randint(0, ncls, size=(sh, sw)).astype(np.int64)
        gt = rs.randint(0, ncls, size=(sh, sw)).astype(np.int64)
        if k == 2:
            pred[pred == 2] = 0                 # class 2 never predicted
        if k == 3:
            gt[:] = 1                           # single-class ground truth
        pairs.append((pred, gt))
    out["pairs"] = pairs
    # two scenarios with 2 and 3 agents, 4 cameras of 6x8 pixels, max_cav 5
    samples = []
    for agents in (2, 3):
        samples.append({"ego": {
            "camera_data": rs.rand(agents, 4, 6, 8, 3),
            "camera_intrinsic": rs.rand(agents, 4, 3, 3),
            "camera_extrinsic": rs.rand(agents, 4, 4, 4),
            "gt_static": rs.randint(0, 3, size=(1, sh, sw)),
            "gt_dynamic": rs.randint(0, 2, size=(1, sh, sw)),
            "transformation_matrix": rs.rand(5, 4, 4),
            "pairwise_t_matrix": rs.rand(5, 5, 4, 4),
        }})
    out["samples"] = samples
    return out


# loss arguments of the shipped configs: corpbevt.yaml:117-123 (dynamic) and corpbevt_static.yaml:115-122 (static), plus "both"
SEG_LOSS = [dict(target="dynamic", d_weights=75.0, s_weights=15.0, d_coe=2.0, s_coe=0.0),
            dict(target="static", d_weights=75.0, s_weights=2.0, l_weights=4.0, d_coe=2.0, s_coe=1.0),
            dict(target="both", d_weights=10.0, s_weights=3.0, d_coe=0.5, s_coe=1.5)]


def seg_loss_inputs():
    """logits (b, 1, c, h, w) and ground truth (b, 1, h, w) for the three loss configurations"""
    import numpy as np
    rs = np.random.RandomState(4321)
    b, h, w = 3, 20, 28
    return {"dynamic_seg": (rs.standard_normal((b, 1, 2, h, w)) * 2.5).astype(np.float32),
            "static_seg": (rs.standard_normal((b, 1, 3, h, w)) * 2.5).astype(np.float32),
            "gt_dynamic": rs.randint(0, 2, size=(b, 1, h, w)).astype(np.int64),
            "gt_static": rs.randint(0, 3, size=(b, 1, h, w)).astype(np.int64)}


# nuScenes IoU metric: config/experiment/cvt_nuscenes_vehicle.yaml groups label channels [4..11] into one output channel with
# min_visibility 2; a two-channel grouping exercises the bit masks
IOU_METRIC = [dict(label_indices=[[4, 5, 6, 7, 8, 9, 10, 11]], min_visibility=2, channels=1),
              dict(label_indices=[[0, 1], [2, 3]], min_visibility=None, channels=2)]


def iou_metric_inputs(channels):
    """two update() calls: (pred (b, channels, h, w), bev labels (b, 12, h, w), visibility (b, h, w))"""
    import numpy as np
    rs = np.random.RandomState(777 + channels)
    out = []
    for b in (2, 1):
        pred = (rs.standard_normal((b, channels, 40, 56)) * 2.0).astype(np.float32)
        pred[0, 0, 0, :4] = [0.0, np.log(0.4 / 0.6), -np.log(0.4 / 0.6), 1e-7]          # on and next to the thresholds
        bev = (rs.rand(b, 12, 40, 56) > 0.93).astype(np.float32)
        vis = rs.choice(np.array([1, 2, 3, 4, 255], dtype=np.uint8), size=(b, 40, 56))
        out.append((pred, bev, vis))
    return out


# nuScenes losses: config/loss/* use BinarySegmentationLoss(label_indices, min_visibility=2, alpha=-1, gamma=2) and
# CenterLoss(min_visibility=2, alpha=-1, gamma=2); a weighted (alpha >= 0) variant exercises the other branch
FOCAL_LOSS = [dict(kind="bev", label_indices=[[4, 5, 6, 7, 8, 9, 10, 11]], min_visibility=2, alpha=-1.0, gamma=2.0),
              dict(kind="bev", label_indices=[[0, 1]], min_visibility=None, alpha=0.25, gamma=1.5),
              dict(kind="center", min_visibility=2, alpha=-1.0, gamma=2.0),
              dict(kind="center", min_visibility=None, alpha=0.75, gamma=2.0)]


def focal_loss_inputs():
    import numpy as np
    rs = np.random.RandomState(99)
    b, h, w = 2, 40, 56
    return {"bev_pred": (rs.standard_normal((b, 1, h, w)) * 2.5).astype(np.float32),
            "center_pred": (rs.standard_normal((b, 1, h, w)) * 2.5).astype(np.float32),
            "bev": (rs.rand(b, 12, h, w) > 0.9).astype(np.float32),
            "center": np.clip(rs.rand(b, 1, h, w) * 1.2 - 0.1, 0, 1).astype(np.float32),
            "visibility": rs.choice(np.array([1, 2, 3, 4, 255], dtype=np.uint8), size=(b, h, w))}
