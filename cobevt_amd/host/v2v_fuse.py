"""V2VNet and DiscoNet fusion behind the reference's module API — mirrors of
opv2v/opencood/models/fusion_modules/v2v_fuse.py:16-144 (V2VNetFusion), fusion_modules/disconet_fuse.py:16-168
(PixelWeightedFusionSoftmax, DiscoNetFusion) and sub_modules/convgru.py:8-78,81-176 (ConvGRUCell / ConvGRU as parameter
containers: the fusions only ever run one step from a zero hidden state).

The reference loops over samples, target agents and source agents in Python on transposed + flipped copies of the maps.  Here one
iteration is a handful of launches over all (sample, target, source) triples at once, on the un-grouped channels-last agent batch
in its original orientation (csrc/pairwise_fusion.hip); the 3x3 convolutions the reference applies in the flipped domain run with
re-indexed taps, conv(flipT(x), W) = flipT(conv(x, W~)) with W~[u][v] = W[v][2 - u].
"""
import torch
import torch.nn as nn

from .. import ops
from ..lib import CobevtHipError
from . import runtime as rt
from .runtime import HipModule


class ConvGRUCell(nn.Module):
    """convgru.py:8-55: parameter container (conv_gates: in + hidden -> 2 hidden, conv_can: in + hidden -> hidden)"""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.height, self.width = input_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.hidden_dim = hidden_dim
        self.bias = bias
        self.conv_gates = nn.Conv2d(input_dim + hidden_dim, 2 * hidden_dim, kernel_size, padding=self.padding, bias=bias)
        self.conv_can = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=self.padding, bias=bias)


class ConvGRU(nn.Module):
    """convgru.py:81-176: parameter container; one layer is what the fusion configs build (cvt_v2vnet.yaml:75-79)"""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, num_layers, batch_first=False, bias=True,
                 return_all_layers=False):
        super().__init__()
        if not isinstance(hidden_dim, list):
            hidden_dim = [hidden_dim] * num_layers
        if not (isinstance(kernel_size, list) and all(isinstance(k, (list, tuple)) for k in kernel_size)):
            kernel_size = [kernel_size] * num_layers
        if not len(kernel_size) == len(hidden_dim) == num_layers:
            raise ValueError("Inconsistent list length.")
        if num_layers != 1 or tuple(kernel_size[0]) != (3, 3):
            raise CobevtHipError("the HIP ConvGRU step is built for one layer of 3x3 convolutions")
        self.cell_list = nn.ModuleList([ConvGRUCell(input_size, input_dim if i == 0 else hidden_dim[i - 1], hidden_dim[i],
                                                    kernel_size[i], bias) for i in range(num_layers)])


def _flipped_domain_taps(w):
    """W~[u][v] = W[v][2 - u]: the 3x3 kernel that does in the original orientation what W does on the transposed + flipped map"""
    return w.detach().transpose(2, 3).flip(2)


def _ego_rows(record_len):
    """row of each sample's first (ego) agent in the un-grouped batch, on the device (no host round trip)"""
    rl = record_len.to(torch.int64)
    return torch.cumsum(rl, 0) - rl


class _PairwiseFusion(HipModule):
    def _setup(self, args):
        self.discrete_ratio = args["resolution"]
        self.downsample_rate = args["downsample_rate"]
        self.num_iteration = args["num_iteration"]
        self.gru_flag = args["gru_flag"]
        self.agg_operator = args["agg_operator"]

    def _inputs(self, x, record_len, pairwise_t_matrix):
        rl = torch.as_tensor(record_len).to(device=x.device, dtype=torch.int32)
        pw = pairwise_t_matrix.to(device=x.device, dtype=torch.float32).contiguous()
        return rl, pw

    def _head(self, feats, rl):
        out = feats.index_select(0, _ego_rows(rl))
        return ops.linear(out, rt.linear_plan(self, "mlp", self.mlp))

    def forward(self, x, record_len, pairwise_t_matrix, prior_encoding=None):
        """x (sum(record_len), C, H, W) -> (B, H, W, C)"""
        self._require_inference(x)
        return rt.like_input(self.forward_nhwc(rt.to_nhwc(x), record_len, pairwise_t_matrix), x)


class V2VNetFusion(_PairwiseFusion):
    """v2v_fuse.py:16-144"""

    def __init__(self, args):
        super().__init__()
        c = args["in_channels"]
        g = args["conv_gru"]
        self._setup(args)
        if self.agg_operator not in ("avg", "max"):
            raise ValueError("agg_operator has wrong value")
        self.msg_cnn = nn.Conv2d(c * 2, c, kernel_size=3, stride=1, padding=1)
        self.conv_gru = ConvGRU(input_size=(g["H"], g["W"]), input_dim=c * 2, hidden_dim=[c], kernel_size=g["kernel_size"],
                                num_layers=g["num_layers"], batch_first=True, bias=True, return_all_layers=False)
        self.mlp = nn.Linear(c, c)
        self.channels = c

    def _plans(self):
        c = self.channels
        cell = self.conv_gru.cell_list[0]
        tensors = rt.module_tensors(self.msg_cnn, cell)

        def build(dt, dev):
            w = _flipped_domain_taps(self.msg_cnn.weight)
            # message = msg_cnn(cat[neighbour, ego]): the ego half (and the bias) is the same for every neighbour
            nb = ops.ConvPlan(w[:, :c], None, stride=1, pad=1, dtype=dt, device=dev)
            ego = ops.ConvPlan(w[:, c:], self.msg_cnn.bias, stride=1, pad=1, dtype=dt, device=dev)
            # one step from h = 0 (convgru.py:57-78): h' = sigmoid(update) * tanh(candidate), both from the 2C input channels only
            wg, wc = _flipped_domain_taps(cell.conv_gates.weight), _flipped_domain_taps(cell.conv_can.weight)
            gw = torch.cat([wg[c:2 * c, :2 * c], wc[:, :2 * c]], dim=0)
            gb = torch.cat([cell.conv_gates.bias.detach()[c:2 * c], cell.conv_can.bias.detach()], dim=0)
            gru = ops.ConvPlan(gw, gb, stride=1, pad=1, dtype=dt, device=dev)
            return nb, ego, gru
        return self._plan("v2v", tensors, build)

    def forward_nhwc(self, x, record_len, pairwise_t_matrix):
        """x (N, H, W, C) channels-last compute dtype -> (B, H, W, C)"""
        rl, pw = self._inputs(x, record_len, pairwise_t_matrix)
        b, l = pw.shape[:2]
        n, h, w, c = x.shape
        p_nb, p_ego, p_gru = self._plans()
        feats = x.contiguous()
        for _ in range(self.num_iteration):
            nb, roi = ops.pairwise_warp(feats, pw, rl, l, self.discrete_ratio, self.downsample_rate)
            msg = ops.conv2d(nb.reshape(b * l * l, h, w, c), p_nb).reshape(b, l, l, h, w, c)
            ego = ops.conv2d(feats, p_ego)
            agg = ops.agent_message_reduce(msg, ego, roi, rl, self.agg_operator)
            if self.gru_flag:
                feats = ops.gru_zero_state(ops.conv2d(torch.cat([feats, agg], dim=-1), p_gru))
            else:
                feats = feats + agg
        return self._head(feats, rl)


class PixelWeightedFusionSoftmax(nn.Module):
    """disconet_fuse.py:16-42: parameter container"""

    def __init__(self, channel):
        super().__init__()
        self.conv1_1 = nn.Conv2d(channel * 2, 128, kernel_size=1, stride=1, padding=0)
        self.bn1_1 = nn.BatchNorm2d(128)
        self.conv1_2 = nn.Conv2d(128, 32, kernel_size=1, stride=1, padding=0)
        self.bn1_2 = nn.BatchNorm2d(32)
        self.conv1_3 = nn.Conv2d(32, 8, kernel_size=1, stride=1, padding=0)
        self.bn1_3 = nn.BatchNorm2d(8)
        self.conv1_4 = nn.Conv2d(8, 1, kernel_size=1, stride=1, padding=0)
        self.softmax = nn.Softmax(dim=0)


class DiscoNetFusion(_PairwiseFusion):
    """disconet_fuse.py:45-168 (cnn / msg_cnn / conv_gru are constructed there but never called: containers only)"""

    def __init__(self, args):
        super().__init__()
        c = args["in_channels"]
        g = args["conv_gru"]
        self._setup(args)
        self.use_temporal_encoding = args["use_temporal_encoding"]
        self.use_mask = args["use_mask"]
        self.cnn = nn.Conv2d(c + 1, c, kernel_size=3, stride=1, padding=1)
        self.msg_cnn = nn.Conv2d(c * 2, c, kernel_size=3, stride=1, padding=1)
        self.conv_gru = ConvGRU(input_size=(g["H"], g["W"]), input_dim=c * 2, hidden_dim=[c], kernel_size=g["kernel_size"],
                                num_layers=g["num_layers"], batch_first=True, bias=True, return_all_layers=False)
        self.mlp = nn.Linear(c, c)
        self.pixel_weighted_fusion = PixelWeightedFusionSoftmax(c)
        self.channels = c

    def _plans(self):
        c = self.channels
        f = self.pixel_weighted_fusion

        def build(dt, dev):
            def fold(conv, bn):
                s = bn.weight.detach().double() / torch.sqrt(bn.running_var.detach().double() + bn.eps)
                t = bn.bias.detach().double() - s * bn.running_mean.detach().double()
                return conv.weight.detach().double() * s[:, None, None, None], conv.bias.detach().double() * s + t
            w1, b1 = fold(f.conv1_1, f.bn1_1)
            # layer 1 on cat[neighbour, ego]: the ego half (+ bias) is computed once per agent and added as the residual
            p1_nb = ops.ConvPlan(w1[:, :c], None, act=1, dtype=dt, device=dev)
            p1_ego = ops.ConvPlan(w1[:, c:], b1, dtype=dt, device=dev)
            w2, b2 = fold(f.conv1_2, f.bn1_2)
            w3, b3 = fold(f.conv1_3, f.bn1_3)
            p2 = ops.ConvPlan(w2, b2, act=1, dtype=dt, device=dev)
            p3 = ops.ConvPlan(w3, b3, act=1, dtype=dt, device=dev)
            # 8 -> 1 scores, padded to 8 output columns (column 0 is the score)
            w4 = torch.zeros(8, 8, 1, 1, dtype=torch.float64)
            b4 = torch.zeros(8, dtype=torch.float64)
            w4[0] = f.conv1_4.weight.detach().double()[0]
            b4[0] = f.conv1_4.bias.detach().double()[0]
            p4 = ops.ConvPlan(w4, b4, act=1, dtype=dt, device=dev)
            return p1_nb, p1_ego, p2, p3, p4
        return self._plan("disco", rt.module_tensors(f), build)

    def forward_nhwc(self, x, record_len, pairwise_t_matrix):
        rl, pw = self._inputs(x, record_len, pairwise_t_matrix)
        b, l = pw.shape[:2]
        n, h, w, c = x.shape
        p1_nb, p1_ego, p2, p3, p4 = self._plans()
        # agent row of every (sample, target) pair (clamped for absent targets: their rows are never used)
        rl64 = rl.to(torch.int64)
        tgt = torch.minimum(torch.arange(l, device=x.device)[None, :], (rl64 - 1).clamp_min(0)[:, None]) + _ego_rows(rl)[:, None]
        tgt = tgt[:, :, None].expand(b, l, l).reshape(-1)
        feats = x.contiguous()
        for _ in range(self.num_iteration):
            nb, roi = ops.pairwise_warp(feats, pw, rl, l, self.discrete_ratio, self.downsample_rate)
            e1 = ops.conv2d(feats, p1_ego)                                       # (N, H, W, 128)
            t = ops.conv2d(nb.reshape(b * l * l, h, w, c), p1_nb, residual=e1.index_select(0, tgt))
            t = ops.conv2d(ops.conv2d(ops.conv2d(t, p2), p3), p4)                # (B L L, H, W, 8), column 0 = ReLU'ed score
            feats = ops.agent_softmax_sum(t.reshape(-1, 8), nb, roi, rl, n, self.use_mask)
        return self._head(feats, rl)
