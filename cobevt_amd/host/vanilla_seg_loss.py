"""VanillaSegLoss — forward-only mirror of opv2v/opencood/loss/vanilla_seg_loss.py:7-76: the class-weighted cross entropy
the reference reports as validation loss (train_camera.py:182-196 under no_grad) and optimises in training.  Same
constructor arguments (`d_weights`, `s_weights`, optional `l_weights` (default 50), `d_coe`, `s_coe`, `target`), same
`forward(output_dict, gt_dict)` -> total loss and `loss_dict` entries; the cross entropies run on the logits where they are
(cobevt_weighted_cross_entropy).  When a prediction requires grad (training, train_camera.py:166-173) the same kernel runs behind
an autograd Function whose backward is cobevt_weighted_cross_entropy_bwd; otherwise the result carries no graph."""
import torch

from .. import autograd as ag
from .. import ops


class VanillaSegLoss(object):
    def __init__(self, args):
        self.d_weights = args["d_weights"]
        self.s_weights = args["s_weights"]
        self.l_weights = 50 if "l_weights" not in args else args["l_weights"]
        self.d_coe = args["d_coe"]
        self.s_coe = args["s_coe"]
        self.target = args["target"]
        self.static_weight = torch.tensor([1.0, self.s_weights, self.l_weights], dtype=torch.float32)
        self.dynamic_weight = torch.tensor([1.0, self.d_weights], dtype=torch.float32)
        self.loss_dict = {}
        self._dev_weights = {}          # (name, device) -> class weights on that device (uploaded once: no per-step host -> device copy)

    def __call__(self, output_dict, gt_dict):
        return self.forward(output_dict, gt_dict)

    def forward(self, output_dict, gt_dict):
        """output_dict: static_seg / dynamic_seg (b, l, c, h, w); gt_dict: gt_static / gt_dynamic (b, l, h, w) integer maps"""
        static_pred, dynamic_pred = output_dict["static_seg"], output_dict["dynamic_seg"]
        static_loss = torch.zeros((), dtype=torch.int64, device=static_pred.device)
        dynamic_loss = torch.zeros((), dtype=torch.int64, device=dynamic_pred.device)
        flat = lambda t: t.reshape(t.shape[0] * t.shape[1], *t.shape[2:])

        def ce(pred, gt, name):
            key = (name, str(pred.device))
            if key not in self._dev_weights:
                self._dev_weights[key] = getattr(self, name).to(pred.device)
            fn = ag.weighted_cross_entropy if (pred.requires_grad and torch.is_grad_enabled()) else ops.weighted_cross_entropy
            return fn(flat(pred), flat(gt).to(pred.device), self._dev_weights[key])
        if self.target != "static":
            dynamic_loss = ce(dynamic_pred, gt_dict["gt_dynamic"], "dynamic_weight")
        if self.target != "dynamic":
            static_loss = ce(static_pred, gt_dict["gt_static"], "static_weight")
        total_loss = self.s_coe * static_loss + self.d_coe * dynamic_loss
        # values for logging (vanilla_seg_loss.py:72-76 prints .item() of them): detached, so that the record of the last step does not
        # keep that step's whole autograd graph alive until the next one
        self.loss_dict.update({"total_loss": total_loss.detach(), "static_loss": static_loss.detach(), "dynamic_loss": dynamic_loss.detach()})
        return total_loss
