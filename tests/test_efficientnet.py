"""nuScenes image backbone: EfficientNetExtractor over EfficientNet-B4.

The extractor wrapper is reference code (backbones/efficientnet.py); the network is efficientnet-pytorch 0.7.1, absent from
the reference tree and from this image -> restated from its published definition in oracle/efficientnet.py, "parity
unpinned" for the third-party arithmetic.  What IS pinned by the reference: the layer selection quirk and the resulting
feature shapes (probed on the reference: (32,56,120), (56,28,60), (112,14,30) for 224x480 images).
CPU: structure / state_dict keys / oracle vs a plain-torch forward over the host module's own nn containers.
GPU (-m gpu): the MBConv kernels vs torch, the extractor vs the oracle (fp32 1e-3 rel, bf16 1e-2 rel), and the whole
nuScenes SinBEVT with the real backbone in front."""
import copy

import numpy as np
import pytest
import torch
import torch.nn.functional as F

import cases
from cobevt_amd import host, ops
from cobevt_amd.host.nuscenes.efficientnet import EfficientNetExtractor, MBConvBlock
from cobevt_amd.synth import fill_module_, procedural_input
import oracle.efficientnet as o_eff
from util import BF16, BF16_FLOOR, assert_close, rel_err

torch.set_grad_enabled(False)
LAYERS = ["reduction_2", "reduction_3", "reduction_4"]          # config/model/cvt_pyramid_axial.yaml:19


def _torch_forward(m, x):
    """the host module's nn containers driven by plain torch ops (an independent wiring of the same state_dict)"""
    def conv_same(conv, x, pad):
        return F.conv2d(F.pad(x, (pad[0], pad[1], pad[0], pad[1])), conv.weight, conv.bias, stride=conv.stride, groups=conv.groups)
    sw = lambda t: t * torch.sigmoid(t)
    stem = m.layers[0]
    x = sw(stem[1](conv_same(stem[0], x, m._stem_pad)))
    res = [x]
    for group in list(m.layers)[1:]:
        for blk in group:
            inp = x
            if blk.expand != 1:
                x = sw(blk._bn0(blk._expand_conv(x)))
            x = sw(blk._bn1(conv_same(blk._depthwise_conv, x, blk.pad)))
            s = blk._se_expand(sw(blk._se_reduce(x.mean((2, 3), keepdim=True))))
            x = blk._bn2(blk._project_conv(torch.sigmoid(s) * x))
            if blk.stride == 1 and blk.cin == blk.cout:
                x = x + inp
        res.append(x)
    return [res[i] for i in m.idx_pick]


def test_extractor_structure_and_reference_quirk():
    m = EfficientNetExtractor(LAYERS, 224, 480)
    assert [tuple(s) for s in m.output_shapes] == [(1,) + tuple(s) for s in cases.NUSCENES["feature_shapes"]]
    assert m.idx_pick == [1, 2, 3] and len(m.layers) == 4                 # stem + the groups of reduction_1..3 only
    assert [len(g) for g in list(m.layers)[1:]] == [3, 4, 4]
    sd = m.state_dict()
    assert "layers.0.0.weight" in sd and tuple(sd["layers.0.0.weight"].shape) == (48, 3, 3, 3)
    assert "layers.1.0._expand_conv.weight" not in sd                      # expand ratio 1 in the first stage
    assert tuple(sd["layers.1.2._expand_conv.weight"].shape) == (144, 24, 1, 1)
    assert tuple(sd["layers.2.3._depthwise_conv.weight"].shape) == (192, 1, 5, 5)
    assert tuple(sd["layers.3.3._se_reduce.weight"].shape) == (14, 336, 1, 1) and tuple(sd["layers.3.3._project_conv.weight"].shape) == (112, 336, 1, 1)
    # stride-2 depthwise convs of a 380-pixel nominal model: pads (0,1) for k=3 on even sizes, (2,2) for k=5 on 95
    assert m._stem_pad == (0, 1) and m.layers[1][2].pad == (0, 1) and m.layers[2][3].pad == (2, 2) and m.layers[3][3].pad == (0, 1)
    # B0 aliases of the docstring example (efficientnet.py:31-36 with the off-by-one: reduction_1 and reduction_3 maps)
    m0 = EfficientNetExtractor(["reduction_2", "reduction_4"], 224, 480, "efficientnet-b0")
    assert [tuple(s) for s in m0.output_shapes] == [(1, 24, 56, 120), (1, 80, 14, 30)]
    with pytest.raises(AssertionError):
        EfficientNetExtractor(["reduction_5"], 224, 480)


def test_oracle_matches_plain_torch_forward():
    m = fill_module_(EfficientNetExtractor(LAYERS, 64, 96), cases.SEED).eval()
    x = procedural_input("eff.x", (2, 3, 64, 96), cases.SEED, -2.0, 2.0)
    ref = _torch_forward(m, x)
    got = o_eff.efficientnet_extractor(m.state_dict(), "", LAYERS, x)
    for a, b, s in zip(got, ref, m.output_shapes):
        assert tuple(a.shape[1:]) == tuple(s[1:])
        assert rel_err(a, b) <= 1e-5
    stem, blocks, res = o_eff.block_table("efficientnet-b4")
    assert (stem, len(blocks), res) == (48, 32, 380)
    assert [b["cout"] for b in blocks][::4] == [24, 32, 56, 112, 160, 160, 272, 272] and blocks[-1]["cout"] == 448


def test_extractor_has_no_cpu_fallback():
    from cobevt_amd.lib import CobevtHipError
    m = EfficientNetExtractor(LAYERS, 64, 96).eval()
    with pytest.raises(CobevtHipError):
        m(torch.zeros(1, 3, 64, 96))


# ------------------------------------------------------------------------------------------------------------------
MODES = [(torch.float32, 1e-4), (torch.bfloat16, 2e-2)]


def _rnd(t, dtype):
    return t.to(dtype).float()


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", MODES)
@pytest.mark.parametrize("k,stride,pad,c,hw", [(3, 1, (1, 1), 48, (20, 28)), (3, 2, (0, 1), 144, (30, 44)), (5, 2, (2, 2), 192, (15, 19)),
                                                (5, 1, (2, 2), 336, (14, 30)), (3, 2, (0, 1), 24, (7, 9)), (5, 2, (1, 2), 64, (16, 16))])
def test_depthwise_conv_kernel(cuda, dtype, tol, k, stride, pad, c, hw):
    n, (h, w) = 3, hw
    x = procedural_input("dw.x", (n, c, h, w), 0, -2, 2)
    wt = procedural_input("dw.w", (c, 1, k, k), 0) / k
    bn = torch.nn.BatchNorm2d(c, eps=1e-3).eval()
    bn.weight.copy_(0.8 + 0.4 * procedural_input("dw.g", (c,), 0, 0, 1)); bn.bias.copy_(procedural_input("dw.b", (c,), 0, -0.3, 0.3))
    bn.running_mean.copy_(procedural_input("dw.m", (c,), 0, -0.2, 0.2)); bn.running_var.copy_(0.5 + procedural_input("dw.v", (c,), 0, 0, 1))
    plan = ops.DepthwisePlan(wt, bn=bn, stride=stride, pad=pad, act=3, dtype=dtype, device=cuda)
    y = ops.depthwise_conv(x.permute(0, 2, 3, 1).contiguous().to(cuda).to(dtype), plan)
    z = bn(F.conv2d(F.pad(_rnd(x, dtype), (pad[0], pad[1], pad[0], pad[1])), wt, None, stride=stride, groups=c))
    ref = (z * torch.sigmoid(z)).permute(0, 2, 3, 1)
    assert tuple(y.shape) == tuple(ref.shape)
    assert rel_err(y, ref) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_squeeze_excite_kernels(cuda, dtype):
    n, h, w, c, cs = 3, 14, 30, 336, 14
    x = procedural_input("se.x", (n, h, w, c), 0, -2, 2)
    xd = x.to(cuda).to(dtype)
    mean = ops.spatial_mean(xd)
    ref_mean = _rnd(x, dtype).double().mean((1, 2)).float()
    assert (mean.cpu() - ref_mean).abs().max().item() <= 2e-6
    assert torch.equal(mean, ops.spatial_mean(xd))                        # fixed summation order: bit-reproducible
    w1, b1 = procedural_input("se.w1", (cs, c), 0) / c ** 0.5, procedural_input("se.b1", (cs,), 0, -0.2, 0.2)
    w2, b2 = procedural_input("se.w2", (c, cs), 0) / cs ** 0.5, procedural_input("se.b2", (c,), 0, -0.2, 0.2)
    gate = ops.se_gate(mean, w1.to(cuda), b1.to(cuda), w2.to(cuda), b2.to(cuda))
    r = F.linear(mean.cpu(), w1, b1)
    ref_gate = torch.sigmoid(F.linear(r * torch.sigmoid(r), w2, b2))
    assert (gate.cpu() - ref_gate).abs().max().item() <= 2e-6
    y = ops.channel_gate(xd, gate)
    ref = _rnd(x, dtype) * gate.cpu()[:, None, None, :]
    assert rel_err(y, ref) <= (1e-6 if dtype == torch.float32 else 5e-3)
    m1 = ops.spatial_mean(procedural_input("se.x1", (2, 1, 1, 24), 0).to(cuda).to(dtype))      # one pixel, one partial chunk group
    assert tuple(m1.shape) == (2, 24)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", MODES)
def test_swish_epilogue_of_the_gemm_paths(cuda, dtype, tol):
    """act = 3 in the dense-row GEMM (1x1 expand conv + BN + swish, with and without the identity skip) and in the generic
    implicit GEMM (the 3-channel stem with TensorFlow-"same" padding (0, 1))"""
    n, h, w, cin, cout = 2, 12, 20, 56, 336
    x = procedural_input("sw.x", (n, h, w, cin), 0, -2, 2)
    wt = procedural_input("sw.w", (cout, cin, 1, 1), 0) / cin ** 0.5
    plan = ops.ConvPlan(wt, None, act=3, dtype=dtype, device=cuda)
    assert plan.wgt_rows is not None
    y = ops.conv2d(x.to(cuda).to(dtype), plan)
    z = F.conv2d(_rnd(x, dtype).permute(0, 3, 1, 2), _rnd(wt, dtype))
    assert rel_err(y, (z * torch.sigmoid(z)).permute(0, 2, 3, 1)) <= tol
    img = procedural_input("sw.img", (n, 23, 30, 3), 0, -2, 2)
    ws = procedural_input("sw.ws", (48, 3, 3, 3), 0) / 27 ** 0.5
    sp = ops.ConvPlan(ws, None, stride=2, pad=0, pad_br=1, act=3, dtype=dtype, device=cuda, smallc=True)
    ys = ops.conv2d(img.to(cuda), sp)
    zs = F.conv2d(F.pad(img.permute(0, 3, 1, 2), (0, 1, 0, 1)), _rnd(ws, dtype), stride=2)
    assert tuple(ys.shape) == (n, 11, 15, 48) == tuple(zs.permute(0, 2, 3, 1).shape)
    assert rel_err(ys, (zs * torch.sigmoid(zs)).permute(0, 2, 3, 1)) <= max(tol, 1e-2 if dtype == torch.bfloat16 else tol)


@pytest.mark.gpu
@pytest.mark.parametrize("dtype,tol", [(torch.float32, 1e-3), (torch.bfloat16, BF16_FLOOR)])   # restated third-party backbone: no reference fixture, BASELINE.md's 1e-2
def test_mbconv_block_and_extractor_vs_oracle(cuda, dtype, tol):
    blk = fill_module_(MBConvBlock(32, 32, 3, 1, 6, 95), cases.SEED).eval()
    x = procedural_input("mb.x", (2, 32, 17, 23), cases.SEED, -2, 2)
    ref = o_eff.mbconv(x, {"b." + k: v for k, v in blk.state_dict().items()}, "b.",
                       dict(kernel=3, stride=1, expand=6, cin=32, cout=32, se=8, image=95))
    with host.compute_dtype(dtype):
        y = blk.to(cuda)(x.to(cuda))
    assert y.dtype == torch.float32
    assert_close(y, ref.numpy(), tol, "MBConvBlock")
    m = fill_module_(EfficientNetExtractor(LAYERS, 224, 480), cases.SEED).eval()
    img = procedural_input("eff.img", (6, 3, 224, 480), cases.SEED, -2.0, 2.0)
    refs = o_eff.efficientnet_extractor(m.state_dict(), "", LAYERS, img)
    m = m.to(cuda)
    with host.compute_dtype(dtype):
        outs = m(img.to(cuda))
    for i, (o, r, s) in enumerate(zip(outs, refs, m.output_shapes)):
        assert tuple(o.shape[1:]) == tuple(s[1:]) == tuple(r.shape[1:])
        assert_close(o, r.numpy(), tol, "EfficientNetExtractor map %d" % i)


@pytest.mark.gpu
def test_nuscenes_sinbevt_with_the_real_backbone(cuda):
    """images -> Normalize -> EfficientNetExtractor -> PyramidAxialEncoder -> Decoder -> heads, all on the device, vs the same
    pipeline assembled from the oracles (fp32 parity mode 1e-3 rel; bf16: util.bf16_gate)"""
    from cobevt_amd.host import nuscenes as nu
    import oracle.nuscenes as o_nu
    c = cases.NUSCENES
    _, image, intr, ext = cases.nuscenes_inputs()
    backbone = EfficientNetExtractor(LAYERS, *c["image"])
    enc = nu.PyramidAxialEncoder(backbone, **copy.deepcopy(c["encoder"]))
    model = fill_module_(nu.CrossViewTransformer(enc, nu.Decoder(**c["decoder"]), c["dim_last"], c["outputs"]), cases.SEED).eval()
    sd = model.state_dict()
    feats = o_eff.efficientnet_extractor(sd, "encoder.backbone.", LAYERS, o_nu.normalize(image.flatten(0, 1)))
    ref = o_nu.cross_view_transformer(sd, c["encoder"], len(c["decoder"]["blocks"]), c["outputs"], feats, intr, ext)
    model = model.to(cuda)
    batch = {"image": image.to(cuda), "intrinsics": intr.to(cuda), "extrinsics": ext.to(cuda)}
    # bf16: gated like the reference's own bf16-autocast deviation of the same model behind its backbone (gv18 "nuScenes SinBEVT")
    for dtype, tol in ((torch.float32, 1e-3), (torch.bfloat16, BF16)):
        with host.compute_dtype(dtype):
            out = model(batch)
        assert tuple(out["bev"].shape) == (1, 1, 200, 200)
        for k in ref:
            assert_close(out[k], ref[k].numpy(), tol, "nuScenes SinBEVT with EfficientNet-B4 [%s] %s" % (k, dtype), case="nuScenes SinBEVT." + k)
