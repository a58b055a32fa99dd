"""FaxFusedTransformer (SinBEVT on OPV2V, no fusion) — mirror of
opv2v/opencood/models/fax_fused_transformer.py:13-48."""
from . import runtime as rt
from .bev_seg_head import BevSegHead
from .fax_modules import FAXModule
from .naive_decoder import NaiveDecoder
from .resnet_ms import ResnetEncoder
from .runtime import HipModule


class FaxFusedTransformer(HipModule):
    def __init__(self, config):
        super().__init__()
        self.encoder = ResnetEncoder(config["encoder"])
        cvm_params = config["fax"]
        cvm_params["backbone_output_shape"] = self.encoder.output_shapes
        self.fax = FAXModule(cvm_params)
        self.decoder = NaiveDecoder(config["decoder"])
        self.target = config["target"]
        self.seg_head = BevSegHead(self.target, config["seg_head_dim"], config["output_class"])

    def _train_tail(self, x, b, l):
        y = self.decoder(x)                                       # (b, l, C', 8H, 8W)
        return self.seg_head(y.reshape(b * l, *y.shape[2:]), b, l)

    def forward(self, batch_dict):
        x = batch_dict["inputs"]
        b, l = x.shape[:2]
        x = self.encoder(x)
        batch_dict.update({"features": x})
        x = self.fax(batch_dict)                                  # (b, l, C, H, W)
        if self.training:                                         # the sub-modules ran their differentiable fp32 graphs
            return self._train_tail(x, b, l)
        y = self.decoder.forward_nhwc(rt.to_nhwc(x.reshape(b * l, *x.shape[2:])))
        return self.seg_head(rt.nchw_view(y), b, l)
