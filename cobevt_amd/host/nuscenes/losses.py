"""Losses of the nuScenes experiments, forward only — mirror of nuscenes/cross_view_transformer/losses.py:10-116
(`SigmoidFocalLoss`, `BinarySegmentationLoss`, `CenterLoss`, `MultipleLoss`): same constructor arguments and call
signatures, the value computed where the logits are (cobevt_sigmoid_focal_loss: label grouping, visibility mask and the mean
in one pass).  fvcore's `sigmoid_focal_loss` (third-party, absent here) is restated from its published definition.
When the prediction requires grad the value comes from cobevt_amd.autograd.SigmoidFocalLossFn (same forward kernel; its gradient w.r.t. the
logits is cobevt_sigmoid_focal_loss_bwd), so `loss.backward()` of the reference's training_step (model_module.py:35-60) works."""
import logging

import torch

from ... import autograd as ag
from ... import ops


def _focal_mean(pred, label, vis, label_indices, min_visibility, alpha, gamma):
    """pred (b, c, hw) logits; the differentiable path when pred carries a graph, the plain forward kernel otherwise"""
    if not (torch.is_grad_enabled() and pred.requires_grad):
        return ops.sigmoid_focal_loss_mean(pred.detach(), label.detach(), vis, label_indices, min_visibility, alpha, gamma)
    dev = pred.device
    lab = label.detach().to(device=dev, dtype=torch.float32).contiguous()
    soft = label_indices is None
    masks = None
    if not soft:
        masks = torch.tensor([sum(1 << int(l) for l in g) for g in label_indices], dtype=torch.int64).to(torch.int32).to(dev)
    v, mv = None, -1
    if min_visibility is not None:
        v, mv = vis.to(device=dev, dtype=torch.uint8).contiguous(), int(min_visibility)
    return ag.SigmoidFocalLossFn.apply(pred.float(), lab, v, masks, (mv, float(alpha), float(gamma), bool(soft)))

logger = logging.getLogger(__name__)


class SigmoidFocalLoss(object):
    def __init__(self, alpha=-1.0, gamma=2.0, reduction="mean"):
        if reduction != "mean":
            raise ValueError("only the mean reduction is implemented on the device (the subclasses of the reference use "
                             "'none' followed by a masked mean, which is what they compute here too)")
        self.alpha, self.gamma, self.reduction = alpha, gamma, reduction

    def __call__(self, pred, label):
        return self.forward(pred, label)

    def forward(self, pred, label):
        """pred, label: same shape (b, c, h, w)"""
        b, c, h, w = pred.shape
        return _focal_mean(pred.reshape(b, c, h * w), label.reshape(b, c, h * w), None, None, None, self.alpha, self.gamma)


class BinarySegmentationLoss(SigmoidFocalLoss):
    def __init__(self, label_indices=None, min_visibility=None, alpha=-1.0, gamma=2.0):
        super().__init__(alpha=alpha, gamma=gamma)
        self.label_indices = label_indices
        self.min_visibility = min_visibility

    def forward(self, pred, batch):
        if isinstance(pred, dict):
            pred = pred["bev"]
        b, c, h, w = pred.shape
        label = batch["bev"]
        vis = batch["visibility"].reshape(b, h * w) if self.min_visibility is not None else None
        return _focal_mean(pred.reshape(b, c, h * w), label.reshape(b, label.shape[1], h * w), vis, self.label_indices, self.min_visibility,
                           self.alpha, self.gamma)


class CenterLoss(SigmoidFocalLoss):
    def __init__(self, min_visibility=None, alpha=-1.0, gamma=2.0):
        super().__init__(alpha=alpha, gamma=gamma)
        self.min_visibility = min_visibility

    def forward(self, pred, batch):
        pred, label = pred["center"], batch["center"]
        b, c, h, w = pred.shape
        vis = batch["visibility"].reshape(b, h * w) if self.min_visibility is not None else None
        return _focal_mean(pred.reshape(b, c, h * w), label.reshape(b, c, h * w), vis, None, self.min_visibility, self.alpha, self.gamma)


class MultipleLoss(dict):
    """losses = MultipleLoss({'bev': BinarySegmentationLoss(...), 'bev_weight': 1.0}); total, parts = losses(pred, batch)"""

    def __init__(self, modules_or_weights):
        weights = {k.replace("_weight", ""): v for k, v in modules_or_weights.items() if isinstance(v, float)}
        modules = {k: v for k, v in modules_or_weights.items() if not isinstance(v, float)}
        for key in modules:
            if key not in weights:
                logger.warning("Weight for %s was not specified.", key)
                weights[key] = 1.0
        assert modules.keys() == weights.keys()
        super().__init__(modules)
        self._weights = weights

    def __call__(self, pred, batch):
        outputs = {k: v(pred, batch) for k, v in self.items()}
        return sum(self._weights[k] * o for k, o in outputs.items()), outputs
