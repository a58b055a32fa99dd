"""IoU metric of the nuScenes experiments — mirror of nuscenes/cross_view_transformer/metrics.py:7-72 (`BaseIoUMetric`,
`IoUMetric`: `update(pred, batch)`, `compute()` -> {'@0.40': iou, '@0.50': iou}) without the torchmetrics base class (absent
here; `reset()` and the tp / fp / fn state tensors are kept).  The prediction never leaves the GPU: one launch per update
(cobevt_iou_counts) adds the per-threshold integer counts, including the label-channel grouping (`label_indices`) and the
visibility mask."""
import torch

from ... import ops


class BaseIoUMetric(object):
    def __init__(self, thresholds=[0.4, 0.5]):
        self.thresholds = torch.FloatTensor(thresholds)
        self.reset()

    def reset(self):
        self.tp = torch.zeros_like(self.thresholds)
        self.fp = torch.zeros_like(self.thresholds)
        self.fn = torch.zeros_like(self.thresholds)
        self._counts = None

    def _add(self, counts):
        """counts: (T, 3) int64 host tensor"""
        c = counts.to(self.tp.dtype)
        self.tp += c[:, 0]
        self.fp += c[:, 1]
        self.fn += c[:, 2]

    def update(self, pred, label):
        """pred: logits, label: same shape, non-zero = positive (:22-31)"""
        pred = pred.detach().reshape(1, 1, -1).float().contiguous()
        label = label.detach().reshape(1, 1, -1).float().contiguous()
        self._add(ops.iou_counts(pred, label, None, [[0]], self.thresholds, None))

    def compute(self):
        ious = self.tp / (self.tp + self.fp + self.fn + 1e-7)
        return {"@%.2f" % t.item(): i.item() for t, i in zip(self.thresholds, ious)}


class IoUMetric(BaseIoUMetric):
    def __init__(self, label_indices, min_visibility=None):
        super().__init__()
        self.label_indices = label_indices
        self.min_visibility = min_visibility

    def update(self, pred, batch):
        """pred: {'bev': (b, c, h, w) logits} or the tensor; batch['bev'] (b, n, h, w) labels, batch['visibility'] (b, h, w)"""
        if isinstance(pred, dict):
            pred = pred["bev"]
        b, c, h, w = pred.shape
        label = batch["bev"]
        vis = batch["visibility"] if self.min_visibility is not None else None
        self._add(ops.iou_counts(pred.detach().float().reshape(b, c, h * w), label.detach().float().reshape(b, label.shape[1], h * w),
                                 None if vis is None else vis.reshape(b, h * w), self.label_indices, self.thresholds, self.min_visibility))
