"""CPU restatement of the data formats either side of the hot path — TEST INFRASTRUCTURE ONLY
(see oracle/__init__.py): the product never imports this.

Follows, function by function:
  standardize_rgb      opv2v/opencood/data_utils/pre_processor/rgb_preprocessor.py:16-33,35-44 (channel swap, /255.,
                       (x - mean) / std in float64)
  resize_linear_float  the resize step :46-55 = cv2.resize(INTER_LINEAR): the published sampling rule (pixel centres aligned,
                       fx = (d + 0.5) * scale - 0.5, border taps clamped) evaluated in float64 - cv2 is absent from the build
                       image, so this is "parity unpinned"; it checks the product's fixed-point restatement to 1 LSB
  bgr_to_gray_u8       cv2.cvtColor(BGR2GRAY) on uint8 as camera_bev_postprocessor.py:33 uses it: OpenCV's published
                       fixed-point formula (B*1868 + G*9617 + R*4899 + 8192) >> 14.  Third-party (opencv-python, not
                       pinned by the reference's requirements): parity unpinned.
  generate_label       camera_bev_postprocessor.py:24-37
  merge_label          camera_bev_postprocessor.py:39-53
  softmax_argmax       camera_bev_postprocessor.py:55-59
  post_process_train   camera_bev_postprocessor.py:61-89
  mean_iu              opv2v/opencood/utils/seg_utils.py:25-50 (+ helpers :57-112)
  mean_precision       seg_utils.py:6-21
  cal_iou_training     seg_utils.py:115-155 (returns inside the loop: only sample 0 is scored)
  collate_batch        opv2v/opencood/data_utils/datasets/camera_only/intermediate_fusion_dataset.py:231-317
  find_last_checkpoint / load_saved_model   opv2v/opencood/tools/train_utils.py:24-65
  vanilla_seg_loss     opv2v/opencood/loss/vanilla_seg_loss.py:7-76 (forward; nn.CrossEntropyLoss(weight) = weighted mean)
  iou_metric           nuscenes/cross_view_transformer/metrics.py:7-72 (BaseIoUMetric / IoUMetric update + compute)
  sigmoid_focal_loss   fvcore.nn.sigmoid_focal_loss (third-party, absent: restated from its published definition, parity unpinned)
  binary_segmentation_loss / center_loss   nuscenes/cross_view_transformer/losses.py:27-84 (forward)
"""
import glob
import os
import re

import numpy as np
import torch


def standardize_rgb(image_u8, mean, std, bgr2rgb):
    img = image_u8[..., ::-1] if bgr2rgb else image_u8
    x = np.asarray(img, dtype=np.float64) / 255.0
    return (x - np.asarray(mean, dtype=np.float64)) / np.asarray(std, dtype=np.float64)


def resize_linear_float(image, width, height):
    img = np.asarray(image, dtype=np.float64)
    h, w = img.shape[:2]
    out = np.zeros((height, width) + img.shape[2:], dtype=np.float64)
    for dy in range(height):
        fy = (dy + 0.5) * h / height - 0.5
        y0 = int(np.floor(fy))
        wy = fy - y0
        if y0 < 0:
            y0, wy = 0, 0.0
        if y0 >= h - 1:
            y0, wy = h - 1, 0.0
        y1 = min(y0 + 1, h - 1)
        for dx in range(width):
            fx = (dx + 0.5) * w / width - 0.5
            x0 = int(np.floor(fx))
            wx = fx - x0
            if x0 < 0:
                x0, wx = 0, 0.0
            if x0 >= w - 1:
                x0, wx = w - 1, 0.0
            x1 = min(x0 + 1, w - 1)
            top = img[y0, x0] * (1 - wx) + img[y0, x1] * wx
            bot = img[y1, x0] * (1 - wx) + img[y1, x1] * wx
            out[dy, dx] = top * (1 - wy) + bot * wy
    return out


def bgr_to_gray_u8(bgr):
    b = bgr[..., 0].astype(np.int64)
    g = bgr[..., 1].astype(np.int64)
    r = bgr[..., 2].astype(np.int64)
    return ((b * 1868 + g * 9617 + r * 4899 + (1 << 13)) >> 14).astype(np.uint8)


def generate_label(bev_map_bgr_u8):
    gray = bgr_to_gray_u8(bev_map_bgr_u8).astype(np.float64) / 255.0
    return np.where(gray > 0, 1.0, gray)


def merge_label(road_map, lane_map):
    out = np.zeros(road_map.shape[:2], dtype=np.float64)
    out[road_map == 1] = 1
    out[lane_map == 1] = 2
    return out


def softmax_argmax(seg_logits):
    prob = torch.softmax(seg_logits.float(), dim=1)
    return prob, torch.argmax(prob, dim=1)


def post_process_train(output_dict):
    sp, sm = softmax_argmax(output_dict["static_seg"][:, 0])
    dp, dm = softmax_argmax(output_dict["dynamic_seg"][:, 0])
    out = dict(output_dict)
    out.update({"static_prob": sp, "static_map": sm, "dynamic_prob": dp, "dynamic_map": dm})
    return out


def _classes(*maps):
    return np.unique(np.concatenate([np.unique(m) for m in maps]))


def mean_iu(pred, gt):
    if pred.shape[:2] != gt.shape[:2]:
        raise ValueError("DiffDim: Different dimensions of matrices!")
    res = []
    for c in _classes(pred, gt):
        pm, gm = pred == c, gt == c
        n_pred, n_gt = int(pm.sum()), int(gm.sum())
        if n_pred == 0 or n_gt == 0:
            res.append(0)
            continue
        inter = int(np.logical_and(pm, gm).sum())
        res.append(inter / (n_gt + n_pred - inter))
    return res


def mean_precision(pred, gt):
    if pred.shape[:2] != gt.shape[:2]:
        raise ValueError("DiffDim: Different dimensions of matrices!")
    res = []
    for c in np.unique(gt):
        pm, gm = pred == c, gt == c
        n_pred = int(pm.sum())
        res.append(0.0 if n_pred == 0 else int(np.logical_and(pm, gm).sum()) / float(n_pred))
    return res


def cal_iou_training(batch_dict, output_dict):
    as_int = lambda t: np.asarray(t.detach().cpu().numpy(), dtype=np.int64)
    gt_static = as_int(batch_dict["ego"]["gt_static"])[0, 0]
    gt_dynamic = as_int(batch_dict["ego"]["gt_dynamic"])[0, 0]
    return mean_iu(as_int(output_dict["dynamic_map"])[0], gt_dynamic), mean_iu(as_int(output_dict["static_map"])[0], gt_static)


def collate_batch(batch, train=True):
    if not train:
        assert len(batch) == 1
    egos = [b["ego"] for b in batch]
    for e in egos:
        assert e["camera_data"].shape[0] == e["camera_intrinsic"].shape[0] == e["camera_extrinsic"].shape[0]
    cat = lambda key: torch.from_numpy(np.concatenate([e[key] for e in egos], axis=0)).unsqueeze(1).float()
    stack = lambda key: torch.from_numpy(np.stack([e[key] for e in egos]))
    return {"ego": {
        "inputs": cat("camera_data"),
        "extrinsic": cat("camera_extrinsic"),
        "intrinsic": cat("camera_intrinsic"),
        "gt_static": stack("gt_static").long(),
        "gt_dynamic": stack("gt_dynamic").long(),
        "transformation_matrix": stack("transformation_matrix").float(),
        "pairwise_t_matrix": stack("pairwise_t_matrix").float(),
        "record_len": torch.from_numpy(np.array([e["camera_data"].shape[0] for e in egos], dtype=int)),
    }}


def find_last_checkpoint(save_dir):
    epochs = [int(re.findall(".*epoch(.*).pth.*", f)[0]) for f in glob.glob(os.path.join(save_dir, "*epoch*.pth"))]
    return max(epochs) if epochs else 0


def load_saved_model(saved_path, model):
    assert os.path.exists(saved_path), "{} not found".format(saved_path)
    epoch = find_last_checkpoint(saved_path)
    if epoch > 0:
        state = torch.load(os.path.join(saved_path, "net_epoch%d.pth" % epoch), map_location="cpu")
        model.load_state_dict(state, strict=False)
    return epoch, model


def vanilla_seg_loss(args, output_dict, gt_dict):
    """-> dict(total_loss, static_loss, dynamic_loss) of 0-d tensors"""
    import torch.nn.functional as F
    flat = lambda t: t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])
    zero = torch.tensor(0)
    static_loss, dynamic_loss = zero, zero
    if args["target"] != "static":
        dynamic_loss = F.cross_entropy(flat(output_dict["dynamic_seg"]).float(), flat(gt_dict["gt_dynamic"]),
                                       weight=torch.tensor([1.0, args["d_weights"]]))
    if args["target"] != "dynamic":
        static_loss = F.cross_entropy(flat(output_dict["static_seg"]).float(), flat(gt_dict["gt_static"]),
                                      weight=torch.tensor([1.0, args["s_weights"], args.get("l_weights", 50)]))
    return {"total_loss": args["s_coe"] * static_loss + args["d_coe"] * dynamic_loss, "static_loss": static_loss,
            "dynamic_loss": dynamic_loss}


def iou_metric(updates, label_indices, min_visibility, thresholds=(0.4, 0.5)):
    """updates: list of (pred (b, c, h, w) logits, batch dict with 'bev' (b, n, h, w) and 'visibility' (b, h, w)) ->
    (tp, fp, fn float tensors per threshold, {'@0.40': iou, ...})"""
    thr = torch.FloatTensor(list(thresholds))
    tp, fp, fn = torch.zeros_like(thr), torch.zeros_like(thr), torch.zeros_like(thr)
    for pred, batch in updates:
        label = torch.cat([batch["bev"][:, idx].max(1, keepdim=True).values for idx in label_indices], 1)
        if min_visibility is not None:
            mask = (batch["visibility"] >= min_visibility)[:, None].expand_as(pred)
            pred, label = pred[mask], label[mask]
        p = pred.detach().sigmoid().reshape(-1)[:, None] >= thr[None]
        l = label.detach().bool().reshape(-1)[:, None]
        tp += (p & l).sum(0)
        fp += (p & ~l).sum(0)
        fn += (~p & l).sum(0)
    ious = tp / (tp + fp + fn + 1e-7)
    return tp, fp, fn, {"@%.2f" % t.item(): i.item() for t, i in zip(thr, ious)}


def sigmoid_focal_loss(inputs, targets, alpha=-1.0, gamma=2.0):
    """element-wise (reduction 'none')"""
    import torch.nn.functional as F
    p = torch.sigmoid(inputs)
    ce = F.binary_cross_entropy_with_logits(inputs, targets, reduction="none")
    p_t = p * targets + (1 - p) * (1 - targets)
    loss = ce * ((1 - p_t) ** gamma)
    if alpha >= 0:
        loss = (alpha * targets + (1 - alpha) * (1 - targets)) * loss
    return loss


def binary_segmentation_loss(pred, batch, label_indices=None, min_visibility=None, alpha=-1.0, gamma=2.0):
    if isinstance(pred, dict):
        pred = pred["bev"]
    label = batch["bev"]
    if label_indices is not None:
        label = torch.cat([label[:, idx].max(1, keepdim=True).values for idx in label_indices], 1)
    loss = sigmoid_focal_loss(pred, label, alpha, gamma)
    if min_visibility is not None:
        loss = loss[(batch["visibility"] >= min_visibility)[:, None]]
    return loss.mean()


def center_loss(pred, batch, min_visibility=None, alpha=-1.0, gamma=2.0):
    loss = sigmoid_focal_loss(pred["center"], batch["center"], alpha, gamma)
    if min_visibility is not None:
        loss = loss[(batch["visibility"] >= min_visibility)[:, None]]
    return loss.mean()
