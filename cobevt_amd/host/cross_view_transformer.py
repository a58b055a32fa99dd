"""CrossViewTransformer (single-agent / late-fusion CVT baseline) — mirror of
opv2v/opencood/models/cross_view_transformer.py:14-51 (hypes_yaml/opcamera/cvt.yaml: core_method cross_view_transformer)."""
from . import runtime as rt
from . import training
from .bev_seg_head import BevSegHead
from .cvt_modules import CrossViewModule
from .naive_decoder import NaiveDecoder
from .resnet_ms import ResnetEncoder
from .runtime import HipModule


class CrossViewTransformer(HipModule):
    def __init__(self, config):
        super().__init__()
        self.encoder = ResnetEncoder(config["encoder"])
        cvm_params = config["cvm"]
        cvm_params["backbone_output_shape"] = self.encoder.output_shapes
        self.cvm = CrossViewModule(cvm_params)
        self.decoder = NaiveDecoder(config["decoder"])
        self.target = config["target"]
        self.seg_head = BevSegHead(self.target, config["seg_head_dim"], config["output_class"])

    def encode_agents(self, batch_dict):
        """images -> (N, H, W, C) channels-last per-agent BEV features"""
        x = self.encoder(batch_dict["inputs"])
        batch_dict.update({"features": x})                          # reference side effect (:41)
        f = self.cvm(batch_dict)                                    # (b, l, C, H, W) channels-last view
        return rt.to_nhwc(f.reshape(-1, *f.shape[2:]))

    def forward(self, batch_dict):
        if self.training:                       # train_camera.py:143-179: the differentiable graph of host/training.py
            return training.cross_view_transformer(self, batch_dict)
        b, l = batch_dict["inputs"].shape[:2]
        y = self.decoder.forward_nhwc(self.encode_agents(batch_dict))
        return self.seg_head(rt.nchw_view(y), b, l)
