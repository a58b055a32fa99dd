"""CorpBEVT (CoBEVT = SinBEVT per agent + FuseBEVT across agents) — mirror of
opv2v/opencood/models/corpbevt.py: STTF :22-64, CorpBEVT :67-145 (constructor config keys, state_dict keys,
forward(batch_dict) -> {'static_seg', 'dynamic_seg'}, the `batch_dict['features']` side effect :113)."""
import os

import torch

from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from . import training
from .bev_seg_head import BevSegHead
from .fax_modules import FAXModule
from .naive_compress import NaiveCompressor
from .naive_decoder import NaiveDecoder
from .resnet_ms import ResnetEncoder
from .runtime import HipModule
from .swap_fusion_modules import SwapFusionEncoder


class STTF(HipModule):
    """Warp every agent's BEV feature map into the ego frame (corpbevt.py:22-64); the affine matrices
    (torch_transformation_utils.py:108-134,160-297), affine_grid + bilinear grid_sample (:317-355) and the
    transpose / flip sandwich are one kernel."""

    def __init__(self, args):
        super().__init__()
        self.discrete_ratio = args["resolution"]
        self.downsample_rate = args["downsample_rate"]

    def warp_blhwc(self, x, spatial_correction_matrix, cav_mask=None, want_mask=False):
        tm = spatial_correction_matrix.to(device=x.device, dtype=torch.float32).contiguous()
        return ops.sttf_warp(x, tm, cav_mask, self.discrete_ratio, self.downsample_rate, want_mask=want_mask)

    def forward(self, x, spatial_correction_matrix):
        """x: (B, L, C, H, W) -> (B, L, H, W, C)"""
        if self.training:
            return training.sttf_warp(x, spatial_correction_matrix.to(x.device), self.discrete_ratio, self.downsample_rate)
        self._require_inference(x)
        b, l, c, h, w = x.shape
        xl = rt.to_nhwc(x.reshape(b * l, c, h, w)).reshape(b, l, h, w, c)
        y, _ = self.warp_blhwc(xl, spatial_correction_matrix)
        return rt.like_input(y, x)


class CorpBEVT(HipModule):
    def __init__(self, config):
        super().__init__()
        self.max_cav = config["max_cav"]
        self.encoder = ResnetEncoder(config["encoder"])
        fax_params = config["fax"]
        fax_params["backbone_output_shape"] = self.encoder.output_shapes
        self.fax = FAXModule(fax_params)
        # corpbevt.py:79-83.  0 in corpbevt.yaml:58; corpbevt_static.yaml has no such key at all (the reference raises a
        # KeyError there) - read as 0
        if config.get("compression", 0) > 0:
            self.compression = True
            self.naive_compressor = NaiveCompressor(128, config["compression"])
        else:
            self.compression = False
        self.downsample_rate = config["sttf"]["downsample_rate"]
        self.discrete_ratio = config["sttf"]["resolution"]
        self.use_roi_mask = config["sttf"]["use_roi_mask"]
        self.sttf = STTF(config["sttf"])
        self.fusion_net = SwapFusionEncoder(config["fax_fusion"])
        self.decoder = NaiveDecoder(config["decoder"])
        self.target = config["target"]
        self.seg_head = BevSegHead(self.target, config["seg_head_dim"], config["output_class"])

    def fuse_and_decode(self, feats, transformation_matrix, record_len):
        """feats: (N, H, W, C) channels-last per-agent BEV features (what V2V sharing transmits) ->
        output dict.  Split out so the multi-GPU path can all-gather `feats` first (cobevt_amd/dist.py)."""
        dev = feats.device
        if self.compression:                                # what each agent would transmit and the receiver's reconstruction
            feats = self.naive_compressor.forward_nhwc(feats)
        rl = torch.as_tensor(record_len).to(device=dev, dtype=torch.int32)
        tm = transformation_matrix.to(device=dev, dtype=torch.float32).contiguous()
        # regroup (fuse_utils.py:8-61) + STTF warp + ROI mask in one launch -> (B, L, H, W, C), (B, H, W, 1, L), (B, L)
        x, com_mask, cav_mask = ops.sttf_warp(feats, tm, None, self.discrete_ratio, self.downsample_rate,
                                              want_mask=self.use_roi_mask, record_len=rl, max_cav=self.max_cav)
        if not self.use_roi_mask:
            b, l, h, w, _ = x.shape
            com_mask = cav_mask[:, None, None, None, :].expand(b, h, w, 1, l).contiguous()
        fused = self.fusion_net.forward_blhwc(x, com_mask)                      # (B, H, W, C)
        if self.taps is not None:
            self.taps.update({"feats": feats, "sttf": x, "com_mask": com_mask, "fused": fused})
        y = self.decoder.forward_nhwc(fused)                                    # (B, 8H, 8W, C')
        b = y.shape[0]
        return self.seg_head(rt.nchw_view(y), b, 1)

    taps = None              # set to a dict to receive intermediate tensors (tests compare them against the oracle's)
    overlap_streams = True   # run each level's key/value path on a side HIP stream under the remaining encoder stages
    overlap_kv = os.environ.get("COBEVT_OVERLAP_KV", "1") != "0"

    def encode_trunk(self, batch_dict, kv_out=None, stage_hook=None):
        """Stage 1 of the per-agent SinBEVT: the camera encoder and everything of the FAX pyramid that depends only on
        the images and the camera geometry (ray embedding, feature projections, K/V projections of both attentions of
        every level).  Returns the state `fax_query` needs: {"kv": [per-level dict], "E_inv", "batch"}.

        Schedule: the key/value side of pyramid level i depends only on encoder stage id_pick[i], not on the BEV query,
        so it is forked onto a side stream as soon as that stage is done and overlaps the remaining ResNet stages (whose
        160-320 workgroups leave CUs idle); the query path joins it right before the level's first attention.
        Captured as parallel graph branches."""
        self._require_inference(batch_dict["inputs"], batch_dict["intrinsic"], batch_dict["extrinsic"])
        pick = self.encoder.idx_pick
        images = batch_dict["inputs"]
        b, l, n = images.shape[:3]
        fax = self.fax
        I_inv = ops.invert_small(batch_dict["intrinsic"].reshape(b * l * n, 3, 3))
        E_inv = fax._extrinsic(batch_dict["extrinsic"].reshape(b * l * n, 4, 4).to(torch.float32)).contiguous()
        main = torch.cuda.current_stream()
        side = self._plan("side_streams", [fax.bev_embedding.learned_features],
                          lambda dt, dev: [torch.cuda.Stream(device=dev) for _ in range(len(pick))])
        feats, kv = {}, {}
        for stage, x in self.encoder.stages_nhwc(images):
            if stage_hook is not None:
                stage_hook(stage)            # a pipeline may order the encoder's next stage behind its other branches (A/B knob)
            if stage not in pick:
                continue
            level = pick.index(stage)
            feats[level] = x
            s = side[level] if self.overlap_kv else main          # overlap_kv False: K/V work in line on the encoder's stream
            s.wait_stream(main)
            with torch.cuda.stream(s):
                kv[level] = fax.cross_views[level].prepare_kv(x, I_inv, E_inv, b * l,
                                                              out=kv_out[level] if kv_out is not None else None)
            x.record_stream(s)
            # I_inv / E_inv are allocated on the main stream and read by ray_embed on the side stream: without this the
            # caching allocator may hand I_inv's block to a later main-stream allocation while the side stream still reads it
            I_inv.record_stream(s)
            E_inv.record_stream(s)
            for t in kv[level].values():
                if torch.is_tensor(t):
                    t.record_stream(main)
        v = [rt.nchw_view(feats[i]) for i in range(len(pick))]
        batch_dict.update({"features": [t.reshape(b, l, n, *t.shape[1:]) for t in v]})    # reference side effect (:113)
        return {"kv": [kv[i] for i in range(len(pick))], "E_inv": E_inv, "batch": b * l, "side": side}

    def fax_query(self, state, joined=True, levels=None, x=None, out=None):
        """Stage 2: the BEV-query side of the FAX pyramid on the K/V state of `encode_trunk` -> (N, H, W, C) channels-last
        BEV features (the tensor V2V sharing transmits).  joined=False: the state's tensors are already complete on the
        current stream (they come from an earlier pipeline step), no side-stream join.  levels / x: run only part of the pyramid
        (FAXModule.forward_features) - a deeper pipeline puts level 0 and levels 1.. on different streams.
        out: optional preallocated result buffer (a pipeline's ring slot) the last kernel writes into."""
        kv, side = state["kv"], state.get("side")
        main = torch.cuda.current_stream()

        def getter(level):
            def get():
                if joined and side is not None:
                    main.wait_stream(side[level])
                return kv[level]
            return get

        return self.fax.forward_features([None] * len(kv), None, state["E_inv"], state["batch"],
                                         kv=[getter(i) for i in range(len(kv))], levels=levels, x=x, out=out)

    def encode_agents(self, batch_dict):
        """Per-agent SinBEVT: images -> (N, H, W, C) channels-last BEV features (the tensor V2V sharing transmits).
        Agents are a pure batch dimension here (corpbevt.py:112-117), which is what the multi-GPU path shards."""
        pick = self.encoder.idx_pick
        if not (self.overlap_streams and isinstance(pick, list) and batch_dict["inputs"].is_cuda):
            x = self.encoder(batch_dict["inputs"])
            batch_dict.update({"features": x})
            x = self.fax(batch_dict)                    # (N, 1, C, H, W) channels-last view
            return rt.to_nhwc(x.squeeze(1))
        return self.fax_query(self.encode_trunk(batch_dict))

    graph_plans = None       # set by enable_graphs(): host.pipeline.AgentCountPlans serving `forward` in eval() mode

    def enable_graphs(self, enabled=True, max_plans=8):
        """Serve eval-mode `model(batch)` from captured HIP graphs: one plan per frame shape (agent count, cameras, image size),
        captured the first time the shape is seen and replayed afterwards (host.pipeline.AgentCountPlans) - the drop-in call of
        inference_camera.py:56 then costs one graph replay instead of ~95 Python-issued launches (3.3 -> 2.1 ms on the 5-agent
        frame).  The returned tensors are the plan's static output buffers: consume (or clone) them before the next call with
        the same shape.  The `batch_dict['features']` side effect (corpbevt.py:113) is not reproduced in this mode."""
        if enabled:
            from .pipeline import AgentCountPlans
            self.graph_plans = AgentCountPlans(self, max_plans=max_plans)
        else:
            self.graph_plans = None
        return self

    def forward(self, batch_dict):
        if self.training:               # the differentiable fp32 graph (host/training.py): train_camera.py:143-179
            return training.corpbevt(self, batch_dict)
        if self.graph_plans is not None and not torch.cuda.is_current_stream_capturing() and not self.graph_plans.busy:
            return self.graph_plans.step(batch_dict)
        feats = self.encode_agents(batch_dict)
        return self.fuse_and_decode(feats, batch_dict["transformation_matrix"], batch_dict["record_len"])
