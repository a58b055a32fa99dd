"""Segmentation scores — mirror of opv2v/opencood/utils/seg_utils.py: mean_IU :25-50, mean_precision :6-21,
cal_iou_training :115-155.

The reference copies both maps to the host and builds one (H, W) float mask per class per map; here the label maps
stay on the GPU and one launch (cobevt_seg_class_counts) reduces them to the three integers per class the formulas use,
so only a few dozen bytes cross PCIe.  The results are the same Python lists (integer counts are exact, the division is
the same float64 division).  Host numpy arrays are accepted too and take the same formulas on counts made with numpy
(that is the reference's own arithmetic, not a fallback for a GPU kernel: no GPU tensor is involved)."""
import numpy as np
import torch

from .. import ops
from ..lib import CobevtHipError


class EvalSegErr(Exception):
    def __init__(self, value):
        self.value = value

    def __str__(self):
        return repr(self.value)


def _counts(eval_segm, gt_segm):
    """-> (classes present in either map (sorted), {class: (n_ii, t_i, n_ij)})"""
    if eval_segm.shape[-2:] != gt_segm.shape[-2:]:
        raise EvalSegErr("DiffDim: Different dimensions of matrices!")
    if torch.is_tensor(eval_segm) and eval_segm.is_cuda:
        gt_dev = gt_segm if torch.is_tensor(gt_segm) else torch.as_tensor(np.asarray(gt_segm))
        gt_dev = gt_dev.to(eval_segm.device)
        hi = int(max(eval_segm.max().item(), gt_dev.max().item())) + 1
        if hi > 8 or min(eval_segm.min().item(), gt_dev.min().item()) < 0:
            raise CobevtHipError("seg_utils: labels must lie in [0, 8) for the GPU path")
        table = ops.seg_class_counts(eval_segm.reshape(1, *eval_segm.shape[-2:]), gt_dev.reshape(1, *gt_dev.shape[-2:]), hi)[0]
        table = table.numpy()
        per_class = {c: tuple(int(v) for v in table[c]) for c in range(hi)}
    else:
        pred, gt = np.asarray(eval_segm), np.asarray(gt_segm)
        per_class = {}
        for c in np.union1d(np.unique(pred), np.unique(gt)):
            pm, gm = pred == c, gt == c
            per_class[c.item()] = (int(np.logical_and(pm, gm).sum()), int(gm.sum()), int(pm.sum()))
    present = sorted(c for c, (_, t_i, n_ij) in per_class.items() if t_i > 0 or n_ij > 0)
    return present, per_class


def mean_IU(eval_segm, gt_segm):
    """per class present in either map: n_ii / (t_i + n_ij - n_ii), 0 when the class is missing from one of them; the
    list is indexed by position among the present classes, as in the reference (:25-50)"""
    present, per_class = _counts(eval_segm, gt_segm)
    iu = [0] * len(present)
    for i, c in enumerate(present):
        n_ii, t_i, n_ij = per_class[c]
        if n_ij == 0 or t_i == 0:
            continue
        iu[i] = n_ii / (t_i + n_ij - n_ii)
    return iu


def mean_precision(eval_segm, gt_segm):
    """per class present in the ground truth: n_ii / n_ij, 0 where the prediction never says that class (:6-21)"""
    _, per_class = _counts(eval_segm, gt_segm)
    out = []
    for c in sorted(c for c, (_, t_i, _) in per_class.items() if t_i > 0):
        n_ii, _, n_ij = per_class[c]
        out.append(0.0 if n_ij == 0 else n_ii / float(n_ij))
    return out


def cal_iou_training(batch_dict, output_dict):
    """(iou_dynamic, iou_static) of sample 0 — the reference returns from inside its batch loop (:134-155), so only the
    first sample of a batch is ever scored; reproduced.  gt_* are (B, 1, H, W); *_map are (B, H, W)."""
    gt_static = batch_dict["ego"]["gt_static"][0, 0]
    gt_dynamic = batch_dict["ego"]["gt_dynamic"][0, 0]
    iou_dynamic = mean_IU(output_dict["dynamic_map"][0], gt_dynamic)
    iou_static = mean_IU(output_dict["static_map"][0], gt_static)
    return iou_dynamic, iou_static
