"""Kernel-level parity: every C-ABI entry point against a plain PyTorch fp32 CPU computation of the same op.

Tolerances: fp32 mode 2e-4 of the output scale (exact-fp32 MFMA, different summation order);
bf16 mode 1e-2 of the output scale (BASELINE.md section 2's bf16 gate; a single kernel's output rounding is 2^-9), against the
fp32 result on bf16-rounded operands.
"""
import math

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from cobevt_amd import ops
from cobevt_amd.synth import procedural_input
import oracle.fax as o_fax
import oracle.sttf as o_sttf
import oracle.swap_fusion as o_swap

pytestmark = pytest.mark.gpu

DTYPES = [torch.float32, torch.bfloat16]


def tol(dtype):
    return 2e-4 if dtype == torch.float32 else 1e-2


def rnd(t, dtype):
    """value the kernel actually sees (bf16 rounding of operands in bf16 mode)"""
    return t.to(dtype).to(torch.float32)


def check(got, ref, dtype, what, scale=None):
    got = got.detach().float().cpu()
    assert got.shape == ref.shape, "%s: shape %s vs %s" % (what, tuple(got.shape), tuple(ref.shape))
    assert torch.isfinite(got).all(), "%s: non-finite output" % what
    s = ref.abs().max().item() if scale is None else scale
    err = (got - ref).abs().max().item()
    assert err <= tol(dtype) * max(s, 1e-6), "%s [%s]: max|err| %.3e vs scale %.3e" % (what, dtype, err, s)


def nhwc(x):  # (N,C,H,W) cpu -> (N,H,W,C) contiguous
    return x.permute(0, 2, 3, 1).contiguous()


class FakeBN(object):
    def __init__(self, c, key):
        self.weight = 0.8 + 0.4 * procedural_input(key + ".w", (c,), 0, 0, 1)
        self.bias = procedural_input(key + ".b", (c,), 0, -0.2, 0.2)
        self.running_mean = procedural_input(key + ".m", (c,), 0, -0.3, 0.3)
        self.running_var = 0.6 + 0.8 * procedural_input(key + ".v", (c,), 0, 0, 1)
        self.eps = 1e-5

    def apply(self, x):
        return F.batch_norm(x, self.running_mean, self.running_var, self.weight, self.bias, False, 0.0, self.eps)


def _conv_case(cuda, dtype, name, n, cin, h, w, cout, k, stride, pad, bias=True, bn=False, act=0, residual=False,
               pre_bn=False, upsample=False, store_mode=0, smallc=False, out_pad=None):
    x = procedural_input(name + ".x", (n, cin, h, w), 0)
    wt = procedural_input(name + ".w", (cout, cin, k, k), 0) * math.sqrt(3.0 / (cin * k * k))
    b = procedural_input(name + ".b", (cout,), 0, -0.3, 0.3) if bias else None
    bnm = FakeBN(cout, name + ".bn") if bn else None
    pbn = FakeBN(cin, name + ".pbn") if pre_bn else None
    plan = ops.ConvPlan(wt, b, bn=bnm, pre_bn=pbn, pre_relu=pre_bn, stride=stride, pad=pad, act=act, upsample=upsample,
                        store_mode=store_mode, dtype=dtype, device=cuda, smallc=smallc)
    # reference on the operands the kernel sees
    xin = x if smallc else rnd(x, dtype)
    if smallc and dtype == torch.bfloat16:
        xin = rnd(x, dtype)  # the kernel rounds the fp32 image to bf16 when packing the A tile
    wref = plan.wgt.float().cpu()[:, :plan.K].reshape(cout, k, k, cin).permute(0, 3, 1, 2)
    bref = plan.bias.cpu() if plan.bias is not None else None
    xr = xin
    if pre_bn:
        xr = rnd(F.relu(pbn.apply(xr)), dtype)
    if upsample:
        xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    ref = F.conv2d(xr, wref, bref, stride=stride, padding=pad)
    res = None
    if residual:
        res = procedural_input(name + ".res", tuple(ref.shape), 0)
        ref = ref + rnd(res, dtype)
    if act == 1:
        ref = F.relu(ref)
    elif act == 2:
        ref = F.gelu(ref)
    xd = nhwc(x).to(cuda) if smallc else nhwc(x).to(cuda).to(dtype)
    rd = nhwc(res).to(cuda).to(dtype) if residual else None
    out = None
    if out_pad is not None:
        out = torch.zeros((n, out_pad[0], out_pad[1], cout), device=cuda, dtype=dtype)
    y = ops.conv2d(xd, plan, residual=rd, out=out)
    torch.cuda.synchronize()
    if store_mode == 0 and out_pad is None:
        check(y.permute(0, 3, 1, 2), ref, dtype, name)
    elif store_mode == 0:
        refp = F.pad(ref, (0, out_pad[1] - ref.shape[3], 0, out_pad[0] - ref.shape[2]))
        check(y.permute(0, 3, 1, 2), refp, dtype, name)
    elif store_mode == 1:
        check(y.permute(0, 3, 1, 2), F.pixel_unshuffle(ref, 2), dtype, name)
    elif store_mode == 2:
        assert y.dtype == torch.float32
        check(y, ref, dtype, name)
    else:
        assert y.dtype == torch.float32
        check(y.permute(0, 3, 1, 2), ref, dtype, name)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_3x3_bias_relu(cuda, dtype):
    _conv_case(cuda, dtype, "c1", 2, 64, 24, 24, 64, 3, 1, 1, act=1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_3x3_s2_bn_residual_relu(cuda, dtype):
    _conv_case(cuda, dtype, "c2", 2, 64, 32, 32, 128, 3, 2, 1, bias=False, bn=True, act=1, residual=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,n,h,w", [(64, 128, 3, 32, 32), (128, 256, 2, 17, 37), (256, 96, 1, 16, 16), (64, 64, 2, 9, 64)])
def test_conv_3x3_s2_strip_kernel(cuda, dtype, cin, cout, n, h, w):
    """stride-2 3x3 conv through the strip kernel (parity-split patch), ragged output maps, 1-4 channel chunks, both cout
    tile widths; and the generic implicit GEMM on the same case"""
    c1 = cin if dtype == torch.bfloat16 else cin // 2
    assert ops.USE_CONV3_S2
    _conv_case(cuda, dtype, "s2_%d_%d" % (cin, cout), n, c1, h, w, cout, 3, 2, 1, bias=False, bn=True, act=1, residual=True)
    ops.USE_CONV3_S2 = False
    try:
        _conv_case(cuda, dtype, "s2g_%d_%d" % (cin, cout), n, c1, h, w, cout, 3, 2, 1, bias=False, bn=True, act=1)
    finally:
        ops.USE_CONV3_S2 = True


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_1x1_big_tile_gelu(cuda, dtype):
    # M = 32768, Cout = 256 -> 512 tiles of 128x128
    _conv_case(cuda, dtype, "c3", 8, 32, 64, 64, 256, 1, 1, 0, act=2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_7x7_s2_image_stem(cuda, dtype):
    _conv_case(cuda, dtype, "c4", 2, 3, 64, 64, 64, 7, 2, 3, bias=False, bn=True, act=1, smallc=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,w", [(3, 128, 96), (1, 38, 50)])
def test_stem_space_to_depth_kernel(cuda, dtype, n, h, w):
    """stem7x7.hip (ragged tiles, several images) vs torch, and vs the generic small-Cin igemm on the same plan"""
    assert ops.USE_STEM
    _conv_case(cuda, dtype, "st%d" % h, n, 3, h, w, 64, 7, 2, 3, bias=False, bn=True, act=1, smallc=True)
    x = procedural_input("stg.x", (n, 3, h, w), 0)
    wt = procedural_input("stg.w", (64, 3, 7, 7), 0) * math.sqrt(3.0 / 147)
    plan = ops.ConvPlan(wt, None, bn=FakeBN(64, "stg.bn"), stride=2, pad=3, act=1, dtype=dtype, device=cuda, smallc=True)
    assert plan.wgt_stem is not None
    xd = nhwc(x).to(cuda)
    a = ops.conv2d(xd, plan)
    ops.USE_STEM = False
    try:
        b = ops.conv2d(xd, plan)
    finally:
        ops.USE_STEM = True
    check(a, b.float().cpu(), dtype, "stem7x7 vs igemm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,h,w", [(3, 128, 96), (1, 36, 52), (2, 64, 64)])
def test_stem_fused_with_maxpool(cuda, dtype, n, h, w):
    """conv1 + bn1 + relu + maxpool(3, 2, 1) in one launch == stem kernel followed by the pool kernel (bit for bit:
    rounding commutes with max), and vs torch; ragged pooled tiles"""
    x = procedural_input("sp.x", (n, 3, h, w), 0)
    wt = procedural_input("sp.w", (64, 3, 7, 7), 0) * math.sqrt(3.0 / 147)
    bn = FakeBN(64, "sp.bn")
    plan = ops.ConvPlan(wt, None, bn=bn, stride=2, pad=3, act=1, dtype=dtype, device=cuda, smallc=True)
    xd = nhwc(x).to(cuda)
    assert ops.USE_STEM_POOL
    y = ops.stem_pool(xd, plan)
    ops.USE_STEM_POOL = False
    try:
        y2 = ops.stem_pool(xd, plan)
    finally:
        ops.USE_STEM_POOL = True
    assert y.shape == (n, h // 4, w // 4, 64)
    from cobevt_amd import lib as _lib
    if _lib.get_variant() == "f32s":
        # split-bf16 matrix path: the two kernels hand (patch, weights) to the MFMA in opposite operand roles, and the split form adds
        # its four cross terms in an operand-dependent order - equal to rounding, not bit for bit
        assert (y - y2).abs().max().item() <= 2e-5 * y2.abs().max().item()
    else:
        assert torch.equal(y, y2), "fused stem+pool differs from stem -> pool in %d values" % (y != y2).sum()
    wref = plan.wgt.float().cpu()[:, :plan.K].reshape(64, 7, 7, 3).permute(0, 3, 1, 2)
    ref = F.max_pool2d(F.relu(F.conv2d(rnd(x, dtype), wref, plan.bias.cpu(), stride=2, padding=3)), 3, 2, 1)
    check(y.permute(0, 3, 1, 2), ref, dtype, "stem+pool vs torch")


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_1x1_preact_bn_relu_padded_out(cuda, dtype):
    _conv_case(cuda, dtype, "c5", 2, 56, 14, 15, 128, 1, 1, 0, bias=False, pre_bn=True, out_pad=(18, 24))


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_3x3_nearest_upsample(cuda, dtype):
    _conv_case(cuda, dtype, "c6", 2, 32, 8, 8, 16, 3, 1, 1, bn=True, act=1, upsample=True)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_3x3_pixel_unshuffle(cuda, dtype):
    _conv_case(cuda, dtype, "c7", 1, 32, 16, 16, 8, 3, 1, 1, bias=False, store_mode=1)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_3x3_head_nchw_fp32(cuda, dtype):
    _conv_case(cuda, dtype, "c8", 2, 8, 16, 16, 2, 3, 1, 1, store_mode=2)


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv_1x1_s2_downsample(cuda, dtype):
    """BasicBlock downsample path: dense-row GEMM with a strided row gather (and the generic igemm on the same plan)"""
    _conv_case(cuda, dtype, "c9", 2, 64, 16, 16, 128, 1, 2, 0, bias=False, bn=True)
    _conv_case(cuda, dtype, "c9b", 3, 128, 15, 21, 256, 1, 2, 0, bias=False, bn=True, act=1)
    ops.USE_GEMM_ROWS = False
    try:
        _conv_case(cuda, dtype, "c9c", 2, 64, 16, 16, 128, 1, 2, 0, bias=False, bn=True)
    finally:
        ops.USE_GEMM_ROWS = True


# ---- the LDS-patch 3x3 kernel (conv3x3.hip): both tile configs, chunk widths, ragged tiles, every epilogue
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,h,w", [(128, 128, 16, 32), (256, 192, 9, 21), (64, 64, 20, 20), (32, 32, 33, 17),
                                          (96, 48, 8, 8)])
def test_conv3x3_patch_kernel_shapes(cuda, dtype, cin, cout, h, w):
    assert ops.USE_CONV3X3
    _conv_case(cuda, dtype, "p%d_%d" % (cin, cout), 2, cin, h, w, cout, 3, 1, 1, bias=False, bn=True, act=1, residual=True)


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("variant", [130, 131, 140, 141, 150, 151, 160, 161])
def test_conv3x3_wfrag_tile_variants(cuda, dtype, variant):
    """cobevt_conv3x3_wfrag_nhwc (fragment-ordered weights): every tile shape, ragged tiles and cout tails, 1-3 channel
    chunks, residual / up-sampled input / PixelUnshuffle epilogues, against torch"""
    assert ops.USE_CONV3_WFRAG
    cc = 64 if dtype == torch.bfloat16 else 32
    cout = 160
    ops.CONV3_VARIANT = variant
    try:
        for i, (nch, h, w, kw) in enumerate([(1, 19, 37, dict(bn=True, act=1, residual=True)),
                                             (3, 16, 16, dict(act=2)),
                                             (2, 9, 12, dict(bias=False, upsample=True, act=1)),
                                             (2, 12, 20, dict(bias=False, store_mode=1))]):
            plan_probe = ops.ConvPlan(torch.zeros(cout, cc * nch, 3, 3), None, stride=1, pad=1, dtype=dtype, device=cuda)
            assert plan_probe.wfrag is not None and plan_probe.coutp3 % 128 == 0
            _conv_case(cuda, dtype, "wf%d_%d" % (variant, i), 2, cc * nch, h, w, cout, 3, 1, 1, **kw)
    finally:
        ops.CONV3_VARIANT = 0


@pytest.mark.parametrize("variant", [130, 131, 140, 141, 150, 151])
@pytest.mark.parametrize("c,cin2,n,h2,w2", [(256, 128, 3, 24, 40), (128, 64, 2, 19, 37), (256, 256, 1, 12, 12)])
def test_conv3x3_with_projection_shortcut(cuda, variant, c, cin2, n, h2, w2):
    """cobevt_conv3x3_ds_wfrag_nhwc: the second 3x3 of a down-sampling BasicBlock with the 1x1 / stride-2 shortcut as extra one-tap chunks
    (layer3.0 / layer4.0), every tile shape, odd source sizes and ragged strips: vs torch on the same rounded operands (the shortcut
    rounded to bf16 before the add, as its own launch stores it) and vs the two launches it replaces (fp32 summation order only)."""
    dtype = torch.bfloat16
    x = procedural_input("ds.x", (n, cin2, h2, w2), variant, -1, 1)
    ho, wo = (h2 - 1) // 2 + 1, (w2 - 1) // 2 + 1
    y = procedural_input("ds.y", (n, c, ho, wo), variant, -1, 1)
    w2_ = procedural_input("ds.w2", (c, c, 3, 3), 0) * math.sqrt(3.0 / (9 * c))
    wd = procedural_input("ds.wd", (c, cin2, 1, 1), 0) * math.sqrt(3.0 / cin2)
    bn2, bnd = FakeBN(c, "ds.bn2"), FakeBN(c, "ds.bnd")
    p1 = ops.ConvPlan(torch.zeros(c, cin2, 3, 3), None, bn=FakeBN(c, "ds.bn1"), stride=2, pad=1, act=1, dtype=dtype, device=cuda)
    p2 = ops.ConvPlan(w2_, None, bn=bn2, stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    pd = ops.ConvPlan(wd, None, bn=bnd, stride=2, pad=0, act=0, dtype=dtype, device=cuda)
    xd, yd = nhwc(x).to(cuda).to(dtype), nhwc(y).to(cuda).to(dtype)
    ops.CONV3_VARIANT = variant
    try:
        assert ops.conv3_ds_fusable(xd, p1, p2, pd) == variant
        out = ops.conv3_ds(yd, xd, p2, pd, variant)
        two = ops.conv2d(yd, p2, residual=ops.conv2d(xd, pd))
    finally:
        ops.CONV3_VARIANT = 0
    torch.cuda.synchronize()
    # the folded weights as the plans hold them (BatchNorm scale folded in, then rounded to bf16)
    s2, sh2 = ops.bn_affine(bn2)
    sd, shd = ops.bn_affine(bnd)
    w2r = rnd(w2_ * s2.cpu().float().view(-1, 1, 1, 1), dtype)
    wdr = rnd(wd * sd.cpu().float().view(-1, 1, 1, 1), dtype)
    ref = F.relu(F.conv2d(rnd(y, dtype), w2r, sh2.cpu().float(), 1, 1) + rnd(F.conv2d(rnd(x, dtype), wdr, shd.cpu().float(), 2, 0), dtype))
    check(out, nhwc(ref), dtype, "conv3x3 + projection shortcut v%d" % variant)
    diff = (out.float() - two.float()).abs()
    assert diff.max().item() <= 2.0 ** -7 * ref.abs().max().item(), diff.max().item()      # rare one-ulp flips of the bf16 stores ...
    assert (diff > 0).float().mean().item() < 0.02                                          # ... and only those


@pytest.mark.parametrize("variant", [113, 123, 133, 143, 153])
def test_conv3x3_wfrag_four_wave_32_cout_tiles(cuda, variant):
    """the four-wave / 32-cout-tile form (bf16, stride 1; what layers with <= 32 output channels are routed to): ragged strips, a
    cout tail over several tiles, 1-3 channel chunks (single-buffered patch refill), residual / up-sampled / PixelUnshuffle epilogues"""
    ops.CONV3_VARIANT = variant
    try:
        for i, (nch, h, w, cout, kw) in enumerate([(1, 19, 37, 32, dict(bn=True, act=1, residual=True)),
                                                   (3, 16, 16, 24, dict(act=2)),
                                                   (2, 9, 12, 72, dict(bias=False, upsample=True, act=1)),
                                                   (2, 12, 20, 32, dict(bias=False, store_mode=1))]):
            _conv_case(cuda, torch.bfloat16, "wf4_%d_%d" % (variant, i), 2, 64 * nch, h, w, cout, 3, 1, 1, **kw)
    finally:
        ops.CONV3_VARIANT = 0
    # the automatic choice takes this form for cout <= 32, and for wider layers only where the eight-wave grid stays below the CU count
    # (one-image decoder maps, the small FAX maps: one or two strips per workgroup - the launch lasts one workgroup lifetime)
    assert ops.conv3_tiling(5, 128, 128, 128, 32, 64) % 10 == 3 and ops.conv3_tiling(5, 64, 64, 128, 32, 64) % 10 == 3
    assert ops.conv3_tiling(1, 32, 32, 128, 128, 64) == 113 and ops.conv3_tiling(1, 128, 128, 64, 64, 64) == 123
    assert ops.conv3_tiling(5, 128, 128, 128, 32, 32, bf16=False) == 0 and ops.conv3_tiling(5, 64, 64, 128, 128, 64) % 10 != 3
    assert ops.conv3_tiling(20, 32, 32, 256, 256, 64) == 150 and ops.conv3_tiling(20, 16, 16, 512, 512, 64) == 151
    with pytest.raises(ops.CobevtHipError):                       # not built for fp32
        ops.CONV3_VARIANT = 153
        try:
            _conv_case(cuda, torch.float32, "wf4_fp32", 1, 32, 8, 16, 32, 3, 1, 1)
        finally:
            ops.CONV3_VARIANT = 0


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_wfrag_matches_lds_staged(cuda, dtype):
    """the two 3x3 kernels on one plan (the LDS-staged one stays the path for 64-byte channel chunks)"""
    cin = 256 if dtype == torch.bfloat16 else 128
    x = procedural_input("wl.x", (3, cin, 32, 32), 0)
    wt = procedural_input("wl.w", (256, cin, 3, 3), 0) * math.sqrt(3.0 / (cin * 9))
    plan = ops.ConvPlan(wt, None, bn=FakeBN(256, "wl.bn"), stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    assert plan.wfrag is not None and plan.wgt3 is not None
    xd = nhwc(x).to(cuda).to(dtype)
    a = ops.conv2d(xd, plan)
    ops.USE_CONV3_WFRAG = False
    try:
        b = ops.conv2d(xd, plan)
    finally:
        ops.USE_CONV3_WFRAG = True
    check(a, b.float().cpu(), dtype, "conv3x3 wfrag vs lds-staged")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cin,cout,h,w", [(128, 128, 16, 32), (256, 192, 9, 21), (64, 64, 20, 20)])
def test_conv3x3_lds_staged_kernel_shapes(cuda, dtype, cin, cout, h, w):
    ops.USE_CONV3_WFRAG = False
    try:
        _conv_case(cuda, dtype, "ps%d_%d" % (cin, cout), 2, cin, h, w, cout, 3, 1, 1, bias=False, bn=True, act=1, residual=True)
    finally:
        ops.USE_CONV3_WFRAG = True


@pytest.mark.parametrize("tile_rows", [0, 8, 16])
@pytest.mark.parametrize("n,h,w", [(2, 32, 32), (1, 21, 37), (3, 16, 48), (1, 5, 3)])
def test_bottleneck_fused(cuda, n, h, w, tile_rows):
    """cobevt_bottleneck_nhwc (FAX ResNetBottleNeck(128), fax_modules.py:10,472) against torchvision's Bottleneck arithmetic in
    fp32 on the bf16-rounded operands with the intermediates rounded where the kernel rounds them, and against the three
    unfused launches; ragged maps cover the zero padding and the partial tiles."""
    import torch.nn as nn
    from cobevt_amd.synth import fill_module_

    class B(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(128, 32, 1, bias=False), nn.BatchNorm2d(32)
            self.conv2, self.bn2 = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32)
            self.conv3, self.bn3 = nn.Conv2d(32, 128, 1, bias=False), nn.BatchNorm2d(128)
    m = fill_module_(B(), 5).eval()
    dtype = torch.bfloat16
    plan = ops.BottleneckPlan(m.conv1, m.bn1, m.conv2, m.bn2, m.conv3, m.bn3, device=cuda)
    x = procedural_input("bnk.x", (n, 128, h, w), 0)
    xd = nhwc(x).to(cuda).to(dtype)
    y = ops.bottleneck(xd, plan, tile_rows)
    torch.cuda.synchronize()
    # reference on what the kernel sees: folded weights rounded to bf16, intermediates rounded to bf16
    def fold(conv, bn):
        sc, sh = ops.bn_affine(bn)
        return (conv.weight.double() * sc[:, None, None, None]).float().to(dtype).float(), sh.float()
    (w1, b1), (w2, b2), (w3, b3) = fold(m.conv1, m.bn1), fold(m.conv2, m.bn2), fold(m.conv3, m.bn3)
    xr = rnd(x, dtype)
    y1 = rnd(F.relu(F.conv2d(xr, w1, b1)), dtype)
    y2 = rnd(F.relu(F.conv2d(y1, w2, b2, padding=1)), dtype)
    ref = F.relu(F.conv2d(y2, w3, b3) + xr)
    check(y.permute(0, 3, 1, 2), ref, dtype, "fused bottleneck %dx%dx%d rows=%d" % (n, h, w, tile_rows))
    # the three-launch path computes the same thing
    p1 = ops.ConvPlan(m.conv1.weight, None, bn=m.bn1, act=1, dtype=dtype, device=cuda)
    p2 = ops.ConvPlan(m.conv2.weight, None, bn=m.bn2, stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    p3 = ops.ConvPlan(m.conv3.weight, None, bn=m.bn3, act=1, dtype=dtype, device=cuda)
    z = ops.conv2d(ops.conv2d(ops.conv2d(xd, p1), p2), p3, residual=xd)
    torch.cuda.synchronize()
    assert (y.float() - z.float()).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("mode", ["fp32", "fp32_split"])
@pytest.mark.parametrize("with_y1", [False, True])
@pytest.mark.parametrize("n,h,w", [(2, 32, 32), (1, 21, 37), (3, 16, 48), (1, 5, 3), (5, 128, 128)])
def test_bottleneck_fp32_storage(cuda, n, h, w, with_y1, mode):
    """cobevt_bottleneck_f32_nhwc (round 6): the FAX ResNetBottleNeck(128) in fp32 storage as ONE launch - conv1 computed on the halo
    region by the kernel, or taken from the producer (y1) - against the three launches it replaces (same library, 2e-5) and fp64 torch
    (2e-4); ragged maps cover the zero padding and the partial tiles, the last case is the level-0 map"""
    import torch.nn as nn
    from cobevt_amd import host
    from cobevt_amd.synth import fill_module_

    class B(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(128, 32, 1, bias=False), nn.BatchNorm2d(32)
            self.conv2, self.bn2 = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32)
            self.conv3, self.bn3 = nn.Conv2d(32, 128, 1, bias=False), nn.BatchNorm2d(128)
    m = fill_module_(B(), 5).eval()
    f32 = torch.float32
    p1 = ops.ConvPlan(m.conv1.weight, None, bn=m.bn1, act=1, dtype=f32, device=cuda)
    p2 = ops.ConvPlan(m.conv2.weight, None, bn=m.bn2, stride=1, pad=1, act=1, dtype=f32, device=cuda)
    p3 = ops.ConvPlan(m.conv3.weight, None, bn=m.bn3, act=1, dtype=f32, device=cuda)
    x = procedural_input("bnk.x", (n, 128, h, w), 0)
    xd = nhwc(x).to(cuda)
    with host.compute_dtype(mode):
        y1 = ops.conv2d(xd, p1) if with_y1 else None
        assert ops.bottleneck_f32_fusable(xd, p1, p2, p3, y1)
        with ops.LaunchProfile() as prof:
            y = ops.bottleneck_f32(xd, p1, p2, p3, y1)
        assert sum(d["calls"] for d in prof.summary().values()) == 1
        z = ops.conv2d(ops.conv2d(ops.conv2d(xd, p1), p2), p3, residual=xd)
    torch.cuda.synchronize()

    def fold(conv, bn):
        sc, sh = ops.bn_affine(bn)
        return conv.weight.double() * sc[:, None, None, None].double(), sh.double()
    (w1, b1), (w2, b2), (w3, b3) = fold(m.conv1, m.bn1), fold(m.conv2, m.bn2), fold(m.conv3, m.bn3)
    xr = x.double()
    ref = F.relu(F.conv2d(F.relu(F.conv2d(F.relu(F.conv2d(xr, w1, b1)), w2, b2, padding=1)), w3, b3) + xr)
    s = ref.abs().max().item()
    yd = y.permute(0, 3, 1, 2).double().cpu()
    assert yd.shape == ref.shape and torch.isfinite(yd).all()
    assert (yd - ref).abs().max().item() <= 2e-4 * s
    assert (y.double() - z.double()).abs().max().item() <= 2e-5 * s


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c,n,h,w", [(64, 2, 24, 40), (128, 3, 16, 16), (64, 1, 13, 21), (128, 2, 9, 35)])
def test_basicblock_fused(cuda, dtype, c, n, h, w):
    """relu(conv2(relu(conv1(x))) + x) in ONE launch (intermediate map in LDS) vs the two 3x3 launches vs torch;
    ragged tiles exercise the zero padding of the intermediate at the image border"""
    x = procedural_input("bb.x", (n, c, h, w), 0)
    w1 = procedural_input("bb.w1", (c, c, 3, 3), 0) * math.sqrt(3.0 / (c * 9))
    w2 = procedural_input("bb.w2", (c, c, 3, 3), 0) * math.sqrt(3.0 / (c * 9))
    bn1, bn2 = FakeBN(c, "bb.bn1"), FakeBN(c, "bb.bn2")
    p1 = ops.ConvPlan(w1, None, bn=bn1, stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    p2 = ops.ConvPlan(w2, None, bn=bn2, stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    xd = nhwc(x).to(cuda).to(dtype)
    assert ops.basicblock_fusable(xd, p1, p2)
    y = ops.basicblock(xd, p1, p2)
    y2 = ops.conv2d(ops.conv2d(xd, p1), p2, residual=xd)
    wr1 = p1.wgt.float().cpu()[:, :p1.K].reshape(c, 3, 3, c).permute(0, 3, 1, 2)
    wr2 = p2.wgt.float().cpu()[:, :p2.K].reshape(c, 3, 3, c).permute(0, 3, 1, 2)
    mid = rnd(F.relu(F.conv2d(rnd(x, dtype), wr1, p1.bias.cpu(), padding=1)), dtype)
    ref = F.relu(F.conv2d(mid, wr2, p2.bias.cpu(), padding=1) + rnd(x, dtype))
    check(y.permute(0, 3, 1, 2), ref, dtype, "fused basicblock vs torch")
    s = ref.abs().max().item()
    assert (y.float() - y2.float()).abs().max().item() <= (1e-2 if dtype == torch.bfloat16 else 1e-5) * s


@pytest.mark.parametrize("n,h,w", [(2, 128, 128), (1, 26, 44), (3, 8, 32), (1, 2, 2)])
def test_dsblock_fused(cuda, n, h, w):
    """layer2's first BasicBlock (64 -> 128, stride 2, projection shortcut) in ONE launch vs the three launches vs torch;
    ragged tiles exercise the zero padding of the input patch and of the intermediate at the image border"""
    dtype = torch.bfloat16
    x = procedural_input("ds.x", (n, 64, h, w), 0)
    w1 = procedural_input("ds.w1", (128, 64, 3, 3), 0) * math.sqrt(3.0 / (64 * 9))
    w2 = procedural_input("ds.w2", (128, 128, 3, 3), 0) * math.sqrt(3.0 / (128 * 9))
    wd = procedural_input("ds.wd", (128, 64, 1, 1), 0) * math.sqrt(3.0 / 64)
    p1 = ops.ConvPlan(w1, None, bn=FakeBN(128, "ds.bn1"), stride=2, pad=1, act=1, dtype=dtype, device=cuda)
    p2 = ops.ConvPlan(w2, None, bn=FakeBN(128, "ds.bn2"), stride=1, pad=1, act=1, dtype=dtype, device=cuda)
    pd = ops.ConvPlan(wd, None, bn=FakeBN(128, "ds.bnd"), stride=2, pad=0, act=0, dtype=dtype, device=cuda)
    xd = nhwc(x).to(cuda).to(dtype)
    assert ops.dsblock_fusable(xd, p1, p2, pd)
    y = ops.dsblock(xd, p1, p2, pd)
    assert y.shape == (n, h // 2, w // 2, 128)
    y2 = ops.conv2d(ops.conv2d(xd, p1), p2, residual=ops.conv2d(xd, pd))
    wr1 = p1.wgt.float().cpu()[:, :p1.K].reshape(128, 3, 3, 64).permute(0, 3, 1, 2)
    wr2 = p2.wgt.float().cpu()[:, :p2.K].reshape(128, 3, 3, 128).permute(0, 3, 1, 2)
    wrd = pd.wgt_rows.float().cpu()[:, :64].reshape(128, 64, 1, 1)
    xr = rnd(x, dtype)
    mid = rnd(F.relu(F.conv2d(xr, wr1, p1.bias.cpu(), stride=2, padding=1)), dtype)
    ref = F.relu(F.conv2d(mid, wr2, p2.bias.cpu(), padding=1) + F.conv2d(xr, wrd, pd.bias.cpu(), stride=2))
    check(y.permute(0, 3, 1, 2), ref, dtype, "fused down-sampling basicblock vs torch")
    s = ref.abs().max().item()
    assert (y.float() - y2.float()).abs().max().item() <= 1e-2 * s
    # the host block takes the fused path for exactly this shape
    ops.USE_DSBLOCK = False
    try:
        assert not ops.dsblock_fusable(xd, p1, p2, pd)
    finally:
        ops.USE_DSBLOCK = True


@pytest.mark.parametrize("dtype", DTYPES)
def test_conv3x3_matches_generic_igemm(cuda, dtype):
    """same plan through both kernels (the generic implicit GEMM is the fallback for padded outputs / stride 2)"""
    x = procedural_input("pg.x", (3, 128, 24, 40), 0)
    wt = procedural_input("pg.w", (64, 128, 3, 3), 0) * math.sqrt(3.0 / (128 * 9))
    plan = ops.ConvPlan(wt, None, bn=FakeBN(64, "pg.bn"), stride=1, pad=1, act=1, upsample=True, dtype=dtype, device=cuda)
    assert plan.wgt3 is not None
    xd = nhwc(x).to(cuda).to(dtype)
    a = ops.conv2d(xd, plan)
    ops.USE_CONV3X3 = False
    try:
        b = ops.conv2d(xd, plan)
    finally:
        ops.USE_CONV3X3 = True
    check(a, b.float().cpu(), dtype, "conv3x3 vs igemm")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,n,rows,ln,res", [(128, 128, 1000, True, False), (128, 384, 130, True, True), (32, 128, 257, False, True),
                                             (256, 128, 300, False, False), (64, 96, 3000, True, False)])
def test_gemm_rows_persistent_wfrag(cuda, dtype, k, n, rows, ln, res):
    """cobevt_linear_rows_wfrag (persistent workgroups, fragment-ordered weights) == cobevt_linear_rows on the same plan"""
    if ln and dtype == torch.float32 and k > 64:
        k = 64
    x = procedural_input("g2.x", (rows, k), 0).to(cuda).to(dtype)
    w = procedural_input("g2.w", (n, k), 0) * math.sqrt(3.0 / k)
    b = procedural_input("g2.b", (n,), 0, -0.2, 0.2)

    class LN(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("g2.g", (k,), 0, 0, 1), procedural_input("g2.be", (k,), 0, -0.2, 0.2), 1e-5
    plan = ops.ConvPlan(w, b, act=2, dtype=dtype, device=cuda, ln=LN if ln else None)
    r = procedural_input("g2.r", (rows, n), 0).to(cuda).to(dtype) if res else None
    y1 = ops.linear(x, plan, residual=r)
    ops.USE_GEMM_ROWS2 = True
    try:
        y2 = ops.linear(x, plan, residual=r)
    finally:
        ops.USE_GEMM_ROWS2 = False
    s = y1.float().abs().max().item()
    assert (y1.float() - y2.float()).abs().max().item() <= (1e-2 if dtype == torch.bfloat16 else 1e-5) * s


@pytest.mark.parametrize("dtype", DTYPES)
def test_linear_ragged_rows(cuda, dtype):
    x = procedural_input("l1.x", (3, 100, 128), 0)
    w = procedural_input("l1.w", (384, 128), 0) * math.sqrt(3.0 / 128)
    b = procedural_input("l1.b", (384,), 0, -0.2, 0.2)
    plan = ops.ConvPlan(w, b, dtype=dtype, device=cuda)
    y = ops.linear(x.to(cuda).to(dtype), plan)
    ref = F.linear(rnd(x, dtype), plan.wgt.float().cpu()[:, :128], b)
    check(y, ref, dtype, "linear")
    # residual + gelu-free second GEMM 384 -> 128
    w2 = procedural_input("l1.w2", (128, 384), 0) * math.sqrt(3.0 / 384)
    plan2 = ops.ConvPlan(w2, None, dtype=dtype, device=cuda)
    z = ops.linear(y, plan2, residual=x.to(cuda).to(dtype))
    ref2 = F.linear(y.float().cpu(), plan2.wgt.float().cpu()[:, :384]) + rnd(x, dtype)
    check(z, ref2, dtype, "linear+residual")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("k,n,rows", [(128, 128, 1000), (128, 384, 130), (32, 128, 257), (256, 128, 300), (64, 96, 64)])
def test_gemm_rows_fused_layernorm(cuda, dtype, k, n, rows):
    """LayerNorm -> Linear (+bias, GELU, residual) in one launch vs separate torch ops; and vs the generic igemm."""
    x = procedural_input("gr.x", (rows, k), 0, -2, 3)
    w = procedural_input("gr.w", (n, k), 0) * math.sqrt(3.0 / k)
    b = procedural_input("gr.b", (n,), 0, -0.2, 0.2)
    g = 0.8 + 0.4 * procedural_input("gr.g", (k,), 0, 0, 1)
    be = procedural_input("gr.be", (k,), 0, -0.2, 0.2)
    res = procedural_input("gr.res", (rows, n), 0)
    class LN(object):
        weight, bias, eps = g, be, 1e-5
    plan = ops.ConvPlan(w, b, act=2, dtype=dtype, device=cuda, ln=LN)
    assert plan.wgt_rows is not None and plan.has_ln
    xd, rd = x.to(cuda).to(dtype), res.to(cuda).to(dtype)
    y = ops.linear(xd, plan, residual=rd)
    # the kernel normalises (no affine), rounds to the compute dtype, and multiplies by the folded weights
    xhat = rnd(F.layer_norm(rnd(x, dtype), (k,), None, None, 1e-5), dtype)
    ref = F.gelu(F.linear(xhat, plan.wgt_rows.float().cpu()[:, :k], plan.bias.cpu()) + rnd(res, dtype))
    check(y, ref, dtype, "gemm_rows LN k=%d n=%d" % (k, n))
    # and agrees with the un-folded definition LayerNorm(x) @ W^T + b to the mode's tolerance
    ref2 = F.gelu(F.linear(F.layer_norm(rnd(x, dtype), (k,), g, be, 1e-5), rnd(w, dtype), b) + rnd(res, dtype))
    check(y, ref2, dtype, "gemm_rows LN (unfolded definition)", scale=ref2.abs().max().item() * (1.0 if dtype == torch.float32 else 2.0))
    ops.USE_GEMM_ROWS = False
    try:
        y2 = ops.linear(xd, plan, residual=rd)       # separate normalisation kernel + generic implicit GEMM
    finally:
        ops.USE_GEMM_ROWS = True
    check(y, y2.float().cpu(), dtype, "gemm_rows vs igemm")


@pytest.mark.parametrize("n,rows,act", [(192, 65536 + 37, 0), (64, 70000, 2), (160, 65536, 0)])
def test_ln_linear64_big_map(cuda, n, rows, act):
    """LayerNorm -> Linear on a big 64-channel map (>= 65,536 rows, the LiDAR encoder's first to_qkv) takes the independent-waves kernel
    (ln_linear64.hip) inside cobevt_linear_rows_small_k: vs torch, and vs the generic dense-row kernel the same rows take in a smaller call
    (rounding of the normalised row to bf16 may differ by one ulp where the two LayerNorm sums round differently, nothing more)."""
    dtype, k = torch.bfloat16, 64
    x = procedural_input("l64.x", (rows, k), 0, -2, 3)
    w = procedural_input("l64.w", (n, k), 0) * math.sqrt(3.0 / k)
    b = procedural_input("l64.b", (n,), 0, -0.2, 0.2)
    g = 0.8 + 0.4 * procedural_input("l64.g", (k,), 0, 0, 1)
    be = procedural_input("l64.be", (k,), 0, -0.2, 0.2)

    class LN(object):
        weight, bias, eps = g, be, 1e-5
    plan = ops.ConvPlan(w, b, act=act, dtype=dtype, device=cuda, ln=LN)
    xd = x.to(cuda).to(dtype)
    y = ops.linear(xd, plan)
    assert y.shape == (rows, n)
    xhat = rnd(F.layer_norm(rnd(x, dtype), (k,), None, None, 1e-5), dtype)
    ref = F.linear(xhat, plan.wgt_rows.float().cpu()[:, :k], plan.bias.cpu())
    ref = F.gelu(ref) if act == 2 else ref
    check(y, ref, dtype, "ln_linear64 n=%d" % n)
    for lo in (0, rows - 4133):                            # the same rows through the generic kernel (below the size threshold)
        y2 = ops.linear(xd[lo:lo + 4133].contiguous(), plan)
        d = (y[lo:lo + 4133].float() - y2.float()).abs().max().item()
        assert d <= 2.0 ** -7 * ref.abs().max().item(), d
    # nothing written past the last row: a tail block's dead lanes must not store
    buf = torch.full((rows + 64, n), 7.0, dtype=dtype, device=cuda)
    ops.linear(xd, plan, out=buf[:rows])
    assert (buf[rows:] == 7.0).all() and torch.equal(buf[:rows], y)


@pytest.mark.parametrize("c,rows,batch,nn_", [(128, 96, 3, 128), (64, 50, 4, 0), (128, 1000, 2, 0)])
def test_attn_mlp_chain_broadcast_skip(cuda, c, rows, batch, nn_):
    """skip = one (rows, C) slice shared by every batch entry (the learned BEV prior at the first pyramid level,
    fax_modules.py:509-510), passed as a stride-0 view: bit-identical to the materialised repeat"""
    dtype, hd = torch.bfloat16, 2 * c
    mk = lambda key, shape, fan: procedural_input(key, shape, 0) * math.sqrt(3.0 / fan)
    g1, be1 = 0.8 + 0.4 * procedural_input("cb.g1", (c,), 0, 0, 1), procedural_input("cb.be1", (c,), 0, -0.2, 0.2)

    class LN1(object):
        weight, bias, eps = g1, be1, 1e-5
    pp = ops.ConvPlan(mk("cb.wp", (c, c), c), None, dtype=dtype, device=cuda)
    p1 = ops.ConvPlan(mk("cb.w1", (hd, c), c), procedural_input("cb.b1", (hd,), 0, -0.2, 0.2), act=2, dtype=dtype, device=cuda, ln=LN1)
    p2 = ops.ConvPlan(mk("cb.w2", (c, hd), hd), procedural_input("cb.b2", (c,), 0, -0.2, 0.2), dtype=dtype, device=cuda)
    pn = ops.ConvPlan(mk("cb.wn", (nn_, c), c), None, dtype=dtype, device=cuda, ln=LN1) if nn_ else None
    a = procedural_input("cb.a", (batch, rows, c), 0, -2, 2).to(cuda).to(dtype)
    prior = procedural_input("cb.s", (rows, c), 0, -1, 1).to(cuda).to(dtype)
    view = prior[None].expand(batch, rows, c)
    assert ops.batch_broadcast(view)
    yb = ops.attn_mlp_chain(a, view, pp, p1, p2, None, next_plan=pn)
    yc = ops.attn_mlp_chain(a, view.contiguous(), pp, p1, p2, None, next_plan=pn)
    if nn_:
        assert torch.equal(yb[0], yc[0]) and torch.equal(yb[1], yc[1])
    else:
        assert torch.equal(yb, yc)
    ops.USE_ROW_CHAIN = False
    try:
        y3 = ops.attn_mlp_chain(a, view, pp, p1, p2, None)            # the three-GEMM path copes with the view too
    finally:
        ops.USE_ROW_CHAIN = True
    y = yb[0] if nn_ else yb
    assert (y.float() - y3.float()).abs().max().item() <= 3e-2 * y3.float().abs().max().item()


@pytest.mark.parametrize("c,hd,rows,post,proj_bias,skip", [(128, 256, 1000, True, True, True), (128, 256, 130, False, False, True),
                                                           (64, 128, 70, False, False, True), (32, 64, 333, True, True, False)])
def test_attn_mlp_chain_fused(cuda, c, hd, rows, post, proj_bias, skip):
    """proj + skip -> pre-norm MLP + residual -> post-norm in ONE launch (bf16) vs the three-GEMM path vs torch fp32."""
    dtype = torch.bfloat16
    a = procedural_input("ch.a", (rows, c), 0, -2, 2)
    sk = procedural_input("ch.s", (rows, c), 0, -1, 1) if skip else None
    mk = lambda key, shape, fan: procedural_input(key, shape, 0) * math.sqrt(3.0 / fan)
    wp, w1, w2 = mk("ch.wp", (c, c), c), mk("ch.w1", (hd, c), c), mk("ch.w2", (c, hd), hd)
    bp = procedural_input("ch.bp", (c,), 0, -0.2, 0.2) if proj_bias else None
    b1, b2 = procedural_input("ch.b1", (hd,), 0, -0.2, 0.2), procedural_input("ch.b2", (c,), 0, -0.2, 0.2)
    g1, be1 = 0.8 + 0.4 * procedural_input("ch.g1", (c,), 0, 0, 1), procedural_input("ch.be1", (c,), 0, -0.2, 0.2)
    g2, be2 = 0.8 + 0.4 * procedural_input("ch.g2", (c,), 0, 0, 1), procedural_input("ch.be2", (c,), 0, -0.2, 0.2)

    class LN1(object):
        weight, bias, eps = g1, be1, 1e-5
    pp = ops.ConvPlan(wp, bp, dtype=dtype, device=cuda)
    p1 = ops.ConvPlan(w1, b1, act=2, dtype=dtype, device=cuda, ln=LN1)
    p2 = ops.ConvPlan(w2, b2, dtype=dtype, device=cuda)
    post_ln = (g2.to(cuda), be2.to(cuda), 1e-5) if post else None
    ad = a.to(cuda).to(dtype)
    sd = sk.to(cuda).to(dtype) if skip else None
    assert ops.USE_ROW_CHAIN
    y = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln)
    ops.USE_ROW_CHAIN = False
    try:
        y3 = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln)
    finally:
        ops.USE_ROW_CHAIN = True
    # torch fp32 definition on the bf16-rounded operands
    yy = F.linear(rnd(a, dtype), rnd(wp, dtype), bp) + (rnd(sk, dtype) if skip else 0)
    zz = yy + F.linear(F.gelu(F.linear(F.layer_norm(yy, (c,), g1, be1, 1e-5), rnd(w1, dtype), b1)), rnd(w2, dtype), b2)
    ref = F.layer_norm(zz, (c,), g2, be2, 1e-5) if post else zz
    s = ref.abs().max().item()
    e_f, e_3 = (y.float().cpu() - ref).abs().max().item(), (y3.float().cpu() - ref).abs().max().item()
    assert e_f <= 3e-2 * s, "fused chain: %.3e vs scale %.3e" % (e_f, s)
    assert e_3 <= 3e-2 * s
    assert (y.float() - y3.float()).abs().max().item() <= 3e-2 * s


@pytest.mark.parametrize("c,rows,nn_,next_ln,next_act,post", [(128, 1000, 384, True, 0, False), (128, 130, 128, True, 0, False),
                                                              (128, 333, 32, False, 1, True), (64, 70, 200, True, 2, True)])
def test_attn_mlp_chain_next_projection(cuda, c, rows, nn_, next_ln, next_act, post):
    """the row-local GEMM that consumes the chain's output (to_qkv behind a LayerNorm, to_q, a Bottleneck's conv1 + BN + ReLU)
    fused into the same launch: `out` identical to the plain chain, `next` vs the separate dense-row GEMM and torch"""
    dtype, hd = torch.bfloat16, 2 * c
    a = procedural_input("cn.a", (rows, c), 0, -2, 2)
    sk = procedural_input("cn.s", (rows, c), 0, -1, 1)
    mk = lambda key, shape, fan: procedural_input(key, shape, 0) * math.sqrt(3.0 / fan)
    wp, w1, w2, wn = mk("cn.wp", (c, c), c), mk("cn.w1", (hd, c), c), mk("cn.w2", (c, hd), hd), mk("cn.wn", (nn_, c), c)
    b1, b2 = procedural_input("cn.b1", (hd,), 0, -0.2, 0.2), procedural_input("cn.b2", (c,), 0, -0.2, 0.2)
    bn_ = procedural_input("cn.bn", (nn_,), 0, -0.2, 0.2)

    class LN1(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("cn.g1", (c,), 0, 0, 1), procedural_input("cn.be1", (c,), 0, -0.2, 0.2), 1e-5

    class LNn(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("cn.gn", (c,), 0, 0, 1), procedural_input("cn.ben", (c,), 0, -0.2, 0.2), 1e-5
    g2, be2 = 0.8 + 0.4 * procedural_input("cn.g2", (c,), 0, 0, 1), procedural_input("cn.be2", (c,), 0, -0.2, 0.2)
    pp = ops.ConvPlan(wp, None, dtype=dtype, device=cuda)
    p1 = ops.ConvPlan(w1, b1, act=2, dtype=dtype, device=cuda, ln=LN1)
    p2 = ops.ConvPlan(w2, b2, dtype=dtype, device=cuda)
    pn = ops.ConvPlan(wn, bn_, act=next_act, dtype=dtype, device=cuda, ln=LNn if next_ln else None)
    post_ln = (g2.to(cuda), be2.to(cuda), 1e-5) if post else None
    ad, sd = a.to(cuda).to(dtype), sk.to(cuda).to(dtype)
    assert ops.USE_ROW_CHAIN and ops.USE_CHAIN_NEXT and ops.chain_next_fusable(pn, c)
    y0 = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln)
    y, nx = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)
    assert torch.equal(y, y0), "the next projection must not change the chain's own output"
    assert nx.shape == (rows, nn_)
    ops.USE_CHAIN_NEXT = False
    try:
        y_b, nx_b = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)      # separate dense-row GEMM launch
    finally:
        ops.USE_CHAIN_NEXT = True
    assert torch.equal(y_b, y0)
    xin = y0.float().cpu()
    if next_ln:
        xin = F.layer_norm(xin, (c,), LNn.weight, LNn.bias, 1e-5)
    ref = F.linear(xin, rnd(wn, dtype), bn_)
    ref = F.relu(ref) if next_act == 1 else (F.gelu(ref) if next_act == 2 else ref)
    s = ref.abs().max().item()
    assert (nx.float().cpu() - ref).abs().max().item() <= 3e-2 * s
    assert (nx.float() - nx_b.float()).abs().max().item() <= 1e-2 * s


@pytest.mark.parametrize("mode", ["fp32", "fp32_split"])
@pytest.mark.parametrize("rows,batch,nn_,next_ln,next_act,post,proj_bias,skip", [
    (1000, 1, 384, True, 0, False, True, "full"), (130, 1, 0, False, 0, True, False, "full"), (333, 1, 32, False, 1, True, True, "none"),
    (96, 3, 128, True, 2, False, False, "broadcast"), (5120, 1, 256, True, 0, False, True, "full")])
def test_attn_mlp_chain_fp32_storage(cuda, mode, rows, batch, nn_, next_ln, next_act, post, proj_bias, skip):
    """csrc/row_chain_f32.hip (round 6): the chain for fp32 storage (C = 128, hidden 256) in ONE launch - exact fp32 MFMA and the
    split-bf16 matrix path - against the separate dense-row launches it replaces (same library) and fp64 torch: ragged row counts,
    with / without projection bias, skip (full, broadcast over the batch, none), post-LayerNorm and the next projection (LayerNorm / ReLU /
    GELU, 32-384 columns).  Gates: 2e-4 of the output scale like every fp32 kernel test; fused vs separate 2e-5."""
    from cobevt_amd import host
    dtype, c, hd = torch.float32, 128, 256
    a = procedural_input("cf.a", (batch, rows, c), 0, -2, 2)
    sk = None if skip == "none" else procedural_input("cf.s", (rows, c) if skip == "broadcast" else (batch, rows, c), 0, -1, 1)
    mk = lambda key, shape, fan: procedural_input(key, shape, 0) * math.sqrt(3.0 / fan)
    wp, w1, w2 = mk("cf.wp", (c, c), c), mk("cf.w1", (hd, c), c), mk("cf.w2", (c, hd), hd)
    bp = procedural_input("cf.bp", (c,), 0, -0.2, 0.2) if proj_bias else None
    b1, b2 = procedural_input("cf.b1", (hd,), 0, -0.2, 0.2), procedural_input("cf.b2", (c,), 0, -0.2, 0.2)

    class LN1(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("cf.g1", (c,), 0, 0, 1), procedural_input("cf.be1", (c,), 0, -0.2, 0.2), 1e-5

    class LNn(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("cf.gn", (c,), 0, 0, 1), procedural_input("cf.ben", (c,), 0, -0.2, 0.2), 1e-5
    g2, be2 = 0.8 + 0.4 * procedural_input("cf.g2", (c,), 0, 0, 1), procedural_input("cf.be2", (c,), 0, -0.2, 0.2)
    pp = ops.ConvPlan(wp, bp, dtype=dtype, device=cuda)
    p1 = ops.ConvPlan(w1, b1, act=2, dtype=dtype, device=cuda, ln=LN1)
    p2 = ops.ConvPlan(w2, b2, dtype=dtype, device=cuda)
    pn = None
    if nn_:
        wn, bn_ = mk("cf.wn", (nn_, c), c), procedural_input("cf.bn", (nn_,), 0, -0.2, 0.2)
        pn = ops.ConvPlan(wn, bn_, act=next_act, dtype=dtype, device=cuda, ln=LNn if next_ln else None)
    post_ln = (g2.to(cuda), be2.to(cuda), 1e-5) if post else None
    ad = a.to(cuda)
    sd = None if sk is None else (sk.to(cuda)[None].expand(batch, rows, c) if skip == "broadcast" else sk.to(cuda))
    with host.compute_dtype(mode):
        assert ops.USE_ROW_CHAIN and ops.USE_ROW_CHAIN_F32
        with ops.LaunchProfile() as prof:
            r = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)
        assert sum(d["calls"] for d in prof.summary().values()) == 1, "the fp32 chain must be ONE launch: %s" % prof.summary()
        ops.USE_ROW_CHAIN_F32 = False
        try:
            r3 = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)
        finally:
            ops.USE_ROW_CHAIN_F32 = True
    y, nx = (r if nn_ else (r, None))
    y3, nx3 = (r3 if nn_ else (r3, None))
    d64 = torch.float64
    yy = F.linear(a.to(d64), wp.to(d64), None if bp is None else bp.to(d64)) + (0 if sk is None else sk.to(d64))
    zz = yy + F.linear(F.gelu(F.linear(F.layer_norm(yy, (c,), LN1.weight.to(d64), LN1.bias.to(d64), 1e-5), w1.to(d64), b1.to(d64))), w2.to(d64), b2.to(d64))
    ref = F.layer_norm(zz, (c,), g2.to(d64), be2.to(d64), 1e-5) if post else zz
    s = ref.abs().max().item()
    assert y.shape == ref.shape and torch.isfinite(y).all()
    assert (y.double().cpu() - ref).abs().max().item() <= 2e-4 * s
    assert (y.double() - y3.double()).abs().max().item() <= 2e-5 * s
    if nn_:
        xin = F.layer_norm(ref, (c,), LNn.weight.to(d64), LNn.bias.to(d64), 1e-5) if next_ln else ref
        rn = F.linear(xin, wn.to(d64), bn_.to(d64))
        rn = F.relu(rn) if next_act == 1 else (F.gelu(rn) if next_act == 2 else rn)
        sn = rn.abs().max().item()
        assert nx.shape == rn.shape
        assert (nx.double().cpu() - rn).abs().max().item() <= 2e-4 * sn
        assert (nx.double() - nx3.double()).abs().max().item() <= 2e-5 * sn


@pytest.mark.parametrize("rows,nn_,next_ln,next_act,post", [(16384 + 40, 192, True, 0, False), (20000, 0, False, 0, True),
                                                            (16384, 64, False, 1, True), (33000, 128, True, 2, False)])
def test_row_chain64_wave_level_kernel(cuda, rows, nn_, next_ln, next_act, post):
    """row_chain64.hip (64-channel rows, hidden 128: the LiDAR FuseBEVT chain as independent waves, activations in registers in
    accumulator k order, weights in LDS) against the barrier-phased generic kernel (ROW_CHAIN_ROWS = 32 pins it) and fp32 torch, on
    ragged row counts, with / without the next projection, its LayerNorm / activation and the post-LayerNorm"""
    dtype, c, hd = torch.bfloat16, 64, 128
    a = procedural_input("c64.a", (rows, c), 0, -2, 2)
    sk = procedural_input("c64.s", (rows, c), 0, -1, 1)
    mk = lambda key, shape, fan: procedural_input(key, shape, 0) * math.sqrt(3.0 / fan)
    wp, w1, w2 = mk("c64.wp", (c, c), c), mk("c64.w1", (hd, c), c), mk("c64.w2", (c, hd), hd)
    bp, b1, b2 = [procedural_input("c64.b%d" % i, (n,), 0, -0.2, 0.2) for i, n in enumerate((c, hd, c))]

    class LN1(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("c64.g1", (c,), 0, 0, 1), procedural_input("c64.be1", (c,), 0, -0.2, 0.2), 1e-5

    class LNn(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("c64.gn", (c,), 0, 0, 1), procedural_input("c64.ben", (c,), 0, -0.2, 0.2), 1e-5
    g2, be2 = 0.8 + 0.4 * procedural_input("c64.g2", (c,), 0, 0, 1), procedural_input("c64.be2", (c,), 0, -0.2, 0.2)
    pp = ops.ConvPlan(wp, bp, dtype=dtype, device=cuda)
    p1 = ops.ConvPlan(w1, b1, act=2, dtype=dtype, device=cuda, ln=LN1)
    p2 = ops.ConvPlan(w2, b2, dtype=dtype, device=cuda)
    pn = None
    if nn_:
        wn, bn_ = mk("c64.wn", (nn_, c), c), procedural_input("c64.bn", (nn_,), 0, -0.2, 0.2)
        pn = ops.ConvPlan(wn, bn_, act=next_act, dtype=dtype, device=cuda, ln=LNn if next_ln else None)
    post_ln = (g2.to(cuda), be2.to(cuda), 1e-5) if post else None
    ad, sd = a.to(cuda).to(dtype), sk.to(cuda).to(dtype)
    with ops.LaunchProfile() as prof:
        res = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)
    assert len(prof.records) == 1
    keep = ops.ROW_CHAIN_ROWS
    ops.ROW_CHAIN_ROWS = 32
    try:
        gen = ops.attn_mlp_chain(ad, sd, pp, p1, p2, post_ln, next_plan=pn)
    finally:
        ops.ROW_CHAIN_ROWS = keep
    y, nx = res if nn_ else (res, None)
    yg, nxg = gen if nn_ else (gen, None)
    # fp32 torch over the bf16-rounded operands, with the same roundings of y / LN(y) / hidden the kernels store
    yr = rnd(F.linear(rnd(a, dtype), rnd(wp, dtype), bp) + rnd(sk, dtype), dtype)
    hid = rnd(F.gelu(F.linear(rnd(F.layer_norm(yr, (c,), None, None, 1e-5), dtype), p1.wgt_rows.float().cpu()[:, :c], p1.bias.cpu())), dtype)
    z = F.linear(hid, rnd(w2, dtype), b2) + yr
    ref = F.layer_norm(z, (c,), g2, be2, 1e-5) if post else z
    s = ref.abs().max().item()
    assert (y.float().cpu() - ref).abs().max().item() <= 1e-2 * s
    assert (y.float() - yg.float()).abs().max().item() <= 1e-2 * s
    if nn_:
        xin = y.float().cpu()
        if next_ln:
            xin = F.layer_norm(xin, (c,), LNn.weight, LNn.bias, 1e-5)
        rn = F.linear(xin, rnd(wn, dtype), bn_)
        rn = F.relu(rn) if next_act == 1 else (F.gelu(rn) if next_act == 2 else rn)
        sn = rn.abs().max().item()
        assert nx.shape == (rows, nn_)
        assert (nx.float().cpu() - rn).abs().max().item() <= 1e-2 * sn
        assert (nx.float() - nxg.float()).abs().max().item() <= 1e-2 * sn


# ---------------------------------------------------------------------------------------------
def _attn_ref(q, k, v, scale, bias=None, key_mask=None):
    """q (G, Nq, dh), k/v (G, Nk, dh) fp32 -> (G, Nq, dh)"""
    s = torch.matmul(q, k.transpose(-1, -2)) * scale
    if bias is not None:
        s = s + bias
    if key_mask is not None:
        s = s.masked_fill(key_mask[:, None, :] == 0, -float("inf"))
    return torch.matmul(s.softmax(-1), v)


def test_attention_index_maps_bit_exact_on_gpu(cuda):
    """The integer part of the attention kernels - token -> row (window / grid partition and reverse) and the relative-position
    table index - dumped by the device functions the kernels use, int32-equal to the REFERENCE's einops / index buffers
    (tests/golden/gv1_index_maps.npz: fax_modules.py:399-404,417-424,123-130; swap_fusion_modules.py:63-85) for every shape of
    the shipped configs, including non-square windows, several cameras per window and a batch offset."""
    import cases
    from util import golden
    g = golden("gv1_index_maps")
    B, ncam = 2, 3
    for (H, W, w1, w2) in cases.INDEX_MAP_SHAPES:
        for mode, key in ((0, "win"), (1, "grid")):
            ref = torch.from_numpy(g["%s_%d_%d_%d_%d" % (key, H, W, w1, w2)].astype(np.int64))       # [l][t] -> flat pixel
            L, ws = ref.shape
            # token t = cam * ws + local (camera-major inside a window, fax_modules.py:211-213); row = (b*ncam + cam)*H*W + pixel
            plane = (torch.arange(B)[:, None] * ncam + torch.arange(ncam)[None, :]) * (H * W)              # [b][cam]
            want = (plane[:, None, :, None] + ref[None, :, None, :]).reshape(B, L, ncam * ws).to(torch.int32)
            got = ops.attention_index_map(ops.tokmap(mode, ncam, H, W, w1, w2), B, cuda).cpu()
            assert torch.equal(got, want), "%s partition %dx%d / %dx%d" % (key, H, W, w1, w2)
        # mode 2 (rows already stored partitioned, the CrossWinAttention.forward API): the identity
        m2 = (2, ncam, H, W, w1, w2, H // w1, W // w2)
        got = ops.attention_index_map(m2, B, cuda).cpu()
        L, ws = (H // w1) * (W // w2), w1 * w2
        b_, l_, c_, t_ = torch.meshgrid(torch.arange(B), torch.arange(L), torch.arange(ncam), torch.arange(ws), indexing="ij")
        want = (((b_ * ncam + c_) * L + l_) * ws + t_).reshape(B, L, ncam * ws).to(torch.int32)
        assert torch.equal(got, want)
    for (L, w) in cases.REL_POS_3D:
        m = ops.tokmap(0, L, w, w, w, w)
        got = ops.attention_bias_index(m, m, L, cuda).cpu()
        assert torch.equal(got, torch.from_numpy(g["rel3d_%d_%d" % (L, w)]).to(torch.int32)), "3-D rel-pos index L=%d w=%d" % (L, w)
        mg = ops.tokmap(1, L, 4 * w, 4 * w, w, w)           # the grid pass uses the same window-local coordinates
        assert torch.equal(ops.attention_bias_index(mg, mg, L, cuda).cpu(), got)
    m8 = ops.tokmap(0, 1, 8, 8, 8, 8)
    assert torch.equal(ops.attention_bias_index(m8, m8, 1, cuda).cpu(), torch.from_numpy(g["rel2d_8"]).to(torch.int32))
    m32 = ops.tokmap(0, 1, 32, 32, 32, 32)
    i32 = ops.attention_bias_index(m32, m32, 1, cuda).cpu()
    assert torch.equal(i32[::37, ::41], torch.from_numpy(g["rel2d_32_sample"]).to(torch.int32))
    assert int(i32.to(torch.int64).sum()) == int(g["rel2d_32_sum"][0])
    # LiDAR FuseBEVT table (8 agents, window 8: 3375 rows) against the oracle's restatement of the same buffer
    ml = ops.tokmap(0, 8, 8, 8, 8, 8)
    assert torch.equal(ops.attention_bias_index(ml, ml, 8, cuda).cpu(),
                       torch.from_numpy(o_swap.relative_position_index_3d(8, 8)).to(torch.int32))


@pytest.mark.parametrize("variant", [0, 1])          # 0 = K/V-resident kernel (bf16, 216 keys padded to 256), 1 = streaming
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("kmode", [0, 1])
@pytest.mark.parametrize("mean_q", [True, False])
def test_cross_window_attention(cuda, dtype, kmode, mean_q, variant):
    B, n, heads, dh = 2, 3, 4, 32
    d = heads * dh
    H = W = 8
    W1 = W2 = 4          # 2x2 query windows
    h, w, w1, w2 = 12, 24, 6, 12   # 2x2 key windows of 6x12
    nq = n if mean_q else 1
    q = procedural_input("att.q", (B, nq, H, W, d), 0)
    k = procedural_input("att.k", (B, n, h, w, d), 0)
    v = procedural_input("att.v", (B, n, h, w, d), 0)
    scale = dh ** -0.5
    qd, kd, vd = [t.to(cuda).to(dtype) for t in (q, k, v)]
    out = torch.empty((B, H, W, d), device=cuda, dtype=dtype)
    qmap = ops.tokmap(0, nq, H, W, W1, W2)
    kmap = ops.tokmap(kmode, n, h, w, w1, w2)
    omap = ops.tokmap(0, 1, H, W, W1, W2)
    ops.window_attention(qd, kd, vd, out, qmap, kmap, omap, B, heads, scale, d, d, d, d, mean_q=mean_q, variant=variant)
    torch.cuda.synchronize()
    # reference through the oracle's partitions
    qp = o_fax._window_partition(rnd(q, dtype), W1, W2)                     # b nq X Y W1 W2 d
    part = o_fax._window_partition if kmode == 0 else o_fax._grid_partition
    kp, vp = part(rnd(k, dtype), w1, w2), part(rnd(v, dtype), w1, w2)
    X, Y = H // W1, W // W2
    qf = qp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, nq * W1 * W2, heads, dh).permute(0, 3, 1, 2, 4)
    kf = kp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, n * w1 * w2, heads, dh).permute(0, 3, 1, 2, 4)
    vf = vp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, n * w1 * w2, heads, dh).permute(0, 3, 1, 2, 4)
    a = torch.matmul((torch.matmul(qf, kf.transpose(-1, -2)) * scale).softmax(-1), vf)   # b m l Q dh
    a = a.permute(0, 2, 3, 1, 4).reshape(B, X, Y, nq, W1, W2, d).mean(3)                    # b X Y W1 W2 d
    ref = o_fax._window_reverse(a)
    check(out, ref, dtype, "cross attention kmode=%d mean=%s" % (kmode, mean_q))


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("n,H,W,h,w", [(3, 8, 8, 5, 7), (2, 16, 8, 16, 16), (4, 4, 4, 24, 24)])
def test_camera_paired_attention(cuda, dtype, n, H, W, h, w):
    """mean_q = 2 (CVT CrossAttention, cvt_modules.py:142-153): camera c's query copy scores camera c's keys, one softmax over all
    cameras' keys; key counts per camera that are not a multiple of the 64-key tile (35), equal to several tiles (256), and
    ragged (576 = 9 tiles); keys addressed without a table (one window = the whole map)."""
    B, heads, dh = 2, 2, 32
    d = heads * dh
    q = procedural_input("pair.q", (B, n, H * W, d), n)
    k = procedural_input("pair.k", (B * n, h, w, d), n)
    v = procedural_input("pair.v", (B * n, h, w, d), n)
    out = torch.empty((B, H, W, d), device=cuda, dtype=dtype)
    qmap, kmap, omap = ops.tokmap(0, n, H, W, H, W), ops.tokmap(0, n, h, w, h, w), ops.tokmap(0, 1, H, W, H, W)
    ops.window_attention(q.to(cuda).to(dtype), k.to(cuda).to(dtype), v.to(cuda).to(dtype), out, qmap, kmap, omap, B, heads,
                         dh ** -0.5, d, d, d, d, mean_q=2)
    torch.cuda.synchronize()
    qf = rnd(q, dtype).reshape(B, n, H * W, heads, dh).permute(0, 3, 1, 2, 4)            # b m n Q dh
    kf = rnd(k, dtype).reshape(B, n, h * w, heads, dh).permute(0, 3, 1, 2, 4)            # b m n K dh
    vf = rnd(v, dtype).reshape(B, n * h * w, heads, dh).permute(0, 2, 1, 3)              # b m (n K) dh
    dot = dh ** -0.5 * torch.matmul(qf, kf.transpose(-1, -2))                            # b m n Q K
    att = dot.permute(0, 1, 3, 2, 4).reshape(B, heads, H * W, n * h * w).softmax(-1)
    ref = torch.matmul(att, vf).permute(0, 2, 1, 3).reshape(B, H, W, d)
    check(out, ref, dtype, "camera-paired attention n=%d keys/cam=%d" % (n, h * w))


@pytest.mark.parametrize("dtype", DTYPES)
def test_attention_prepartitioned_long_keys(cuda, dtype):
    """mode-2 (stored partitioned) maps, Nq = 1024 rows in one window, Nk = 1024 keys (16 key tiles)."""
    B, heads, dh = 1, 2, 32
    d = heads * dh
    q = procedural_input("att2.q", (B, 1, 1, 1, 32, 32, d), 0)
    k = procedural_input("att2.k", (B, 4, 1, 1, 16, 16, d), 0)
    v = procedural_input("att2.v", (B, 4, 1, 1, 16, 16, d), 0)
    out = torch.empty((B, 1, 1, 32, 32, d), device=cuda, dtype=dtype)
    qmap = (2, 1, 32, 32, 32, 32, 1, 1)
    kmap = (2, 4, 16, 16, 16, 16, 1, 1)
    ops.window_attention(q.to(cuda).to(dtype), k.to(cuda).to(dtype), v.to(cuda).to(dtype), out, qmap, kmap, qmap, B,
                         heads, dh ** -0.5, d, d, d, d)
    torch.cuda.synchronize()
    qf = rnd(q, dtype).reshape(1024, heads, dh).permute(1, 0, 2)
    kf = rnd(k, dtype).reshape(1024, heads, dh).permute(1, 0, 2)
    vf = rnd(v, dtype).reshape(1024, heads, dh).permute(1, 0, 2)
    ref = _attn_ref(qf, kf, vf, dh ** -0.5).permute(1, 0, 2).reshape(B, 1, 1, 32, 32, d)
    check(out, ref, dtype, "prepartitioned attention")


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("mode", [0, 1])
def test_swap_attention_bias_mask(cuda, dtype, mode):
    """fused qkv (ld = 3d), 3-D relative position bias, key mask, window / grid partition over (agent, h, w)."""
    B, L, w, heads, dh = 2, 3, 4, 2, 32
    d = heads * dh
    H = W = 8
    qkv = procedural_input("sw.qkv", (B, L, H, W, 3 * d), 0)
    table = procedural_input("sw.table", ((2 * L - 1) * (2 * w - 1) ** 2, heads), 0)
    mask = torch.ones(B, H, W, 1, L)
    mask[1, :, :, :, 2] = 0
    mask[0, :3, :, :, 1] = 0
    mask[0, :, 6:, :, 2] = 0
    out = torch.empty((B, L, H, W, d), device=cuda, dtype=dtype)
    m = ops.tokmap(mode, L, H, W, w, w)
    qd = qkv.to(cuda).to(dtype)
    ops.window_attention(qd, qd, qd, out, m, m, m, B, heads, dh ** -0.5, 3 * d, 3 * d, 3 * d, d, qoff=0, koff=d,
                         voff=2 * d, bias_table=table.to(cuda), bias_L=L, mask=mask.to(cuda).contiguous())
    torch.cuda.synchronize()
    x = rnd(qkv, dtype)
    part = (lambda t: t.reshape(B, L, H // w, w, W // w, w, -1).permute(0, 1, 2, 4, 3, 5, 6)) if mode == 0 else \
           (lambda t: t.reshape(B, L, w, H // w, w, W // w, -1).permute(0, 1, 3, 5, 2, 4, 6))
    xp = part(x)                                                       # b l X Y w1 w2 3d
    X = Y = H // w
    t = xp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B * X * Y, L * w * w, 3 * d)
    qf, kf, vf = [z.reshape(B * X * Y, L * w * w, heads, dh).permute(0, 2, 1, 3) for z in t.chunk(3, -1)]
    idx = torch.from_numpy(o_swap.relative_position_index_3d(L, w))
    bias = table[idx].permute(2, 0, 1)
    mp = mask.reshape(B, H // w, w, W // w, w, 1, L).permute(0, 1, 3, 2, 4, 5, 6) if mode == 0 else \
        mask.reshape(B, w, H // w, w, W // w, 1, L).permute(0, 2, 4, 1, 3, 5, 6)
    mk = mp.permute(0, 1, 2, 5, 6, 3, 4).reshape(B * X * Y, 1, L * w * w)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * dh ** -0.5 + bias
    s = s.masked_fill(mk.unsqueeze(1) == 0, -float("inf"))
    o = torch.matmul(s.softmax(-1), vf).permute(0, 2, 1, 3).reshape(B, X, Y, L, w, w, d).permute(0, 3, 1, 2, 4, 5, 6)
    if mode == 0:
        ref = o.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, L, H, W, d)
    else:
        ref = o.permute(0, 1, 4, 2, 5, 3, 6).reshape(B, L, H, W, d)
    check(out, ref, dtype, "swap attention mode=%d" % mode)


@pytest.mark.parametrize("qsplit", [0, 1, 2, 4])
@pytest.mark.parametrize("mode", [0, 1])
@pytest.mark.parametrize("L,w,H,use_mask,outlier", [(5, 8, 16, True, False), (8, 8, 16, True, False), (2, 8, 24, False, False), (3, 8, 16, True, False),
                                                    (8, 8, 16, True, True), (8, 8, 32, False, True)])
def test_resident_attention_bias_mask(cuda, L, w, H, use_mask, outlier, mode, qsplit):
    """K/V-resident kernel with the 3-D relative-position bias (+ key mask) on swap-fusion shapes: 320 keys (the 5-agent
    fusion), 512 keys (LiDAR: 8 waves per workgroup), 128 and 192 keys (padded), every query split - against the fp32 torch
    computation AND the streaming kernel (same bf16 operands: equal to rounding).  outlier: a few keys scaled so that their scores
    leave the range the eight-wave variants' branch-free pass accepts against a carried reference (2^+-40 .. 2^-80 in the row sum)
    in both directions - those tasks are redone by the exact per-tile loop."""
    B, heads, dh = 2, 2, 32
    d = heads * dh
    W = H
    dtype = torch.bfloat16
    if outlier and qsplit == 0 and B * (H // w) ** 2 * heads < 256:
        pytest.skip("automatic split: fewer (window, head) items than CUs go to the streaming kernel")
    qkv = procedural_input("rsw.qkv", (B, L, H, W, 3 * d), L)
    if outlier:
        qkv[0, L - 1, 5, 3, d:2 * d] *= 60.0        # the last agent's key tile: a late spike
        qkv[0, 0, 9, 9, d:2 * d] *= 45.0            # the first: an early one
        qkv[1, 3, 2, 12, d:2 * d] *= 80.0           # (a lane's next query sees the spike with the other sign: the carried reference is then
        #                                             far ABOVE its scores, the row sum underflows, the task is redone as well)
    table = procedural_input("rsw.table", ((2 * L - 1) * (2 * w - 1) ** 2, heads), L)
    mask = torch.ones(B, H, W, 1, L)
    mask[1, :, :, :, L - 1] = 0
    mask[0, :3, :, :, 1] = 0
    mask[0, :, W - 5:, :, L - 1] = 0
    outs = [torch.empty((B, L, H, W, d), device=cuda, dtype=dtype) for _ in range(2)]
    m = ops.tokmap(mode, L, H, W, w, w)
    qd = qkv.to(cuda).to(dtype)
    mk = mask.to(cuda).contiguous() if use_mask else None
    for o, variant in zip(outs, (0, 1)):
        ops.window_attention(qd, qd, qd, o, m, m, m, B, heads, dh ** -0.5, 3 * d, 3 * d, 3 * d, d, qoff=0, koff=d,
                             voff=2 * d, bias_table=table.to(cuda), bias_L=L, mask=mk, variant=variant, qsplit=qsplit)
    torch.cuda.synchronize()
    x = rnd(qkv, dtype)
    X = Y = H // w
    part = (lambda t: t.reshape(B, L, X, w, Y, w, -1).permute(0, 1, 2, 4, 3, 5, 6)) if mode == 0 else \
           (lambda t: t.reshape(B, L, w, X, w, Y, -1).permute(0, 1, 3, 5, 2, 4, 6))
    t = part(x).permute(0, 2, 3, 1, 4, 5, 6).reshape(B * X * Y, L * w * w, 3 * d)
    qf, kf, vf = [z.reshape(B * X * Y, L * w * w, heads, dh).permute(0, 2, 1, 3) for z in t.chunk(3, -1)]
    bias = table[torch.from_numpy(o_swap.relative_position_index_3d(L, w))].permute(2, 0, 1)
    if outlier:
        # the resident kernel stages K pre-multiplied by scale * log2(e), rounded to bf16 once more: with logits in the hundreds that
        # rounding (2^-9 of sum |q_i k_i|) moves a softmax weight by tens of per cent, so the reference takes the operand the kernel sees
        c = dh ** -0.5 * math.log2(math.e)
        kf = rnd(kf * c, dtype) / c
    sc = torch.matmul(qf, kf.transpose(-1, -2)) * dh ** -0.5 + bias
    if use_mask:
        mp = mask.reshape(B, X, w, Y, w, 1, L).permute(0, 1, 3, 2, 4, 5, 6) if mode == 0 else \
            mask.reshape(B, w, X, w, Y, 1, L).permute(0, 2, 4, 1, 3, 5, 6)
        mk_ = mp.permute(0, 1, 2, 5, 6, 3, 4).reshape(B * X * Y, 1, L * w * w)
        sc = sc.masked_fill(mk_.unsqueeze(1) == 0, -float("inf"))
    o = torch.matmul(sc.softmax(-1), vf).permute(0, 2, 1, 3).reshape(B, X, Y, L, w, w, d).permute(0, 3, 1, 2, 4, 5, 6)
    ref = o.permute(0, 1, 2, 4, 3, 5, 6).reshape(B, L, H, W, d) if mode == 0 else o.permute(0, 1, 4, 2, 5, 3, 6).reshape(B, L, H, W, d)
    check(outs[0], ref, dtype, "resident swap attention L=%d mode=%d qsplit=%d" % (L, mode, qsplit))
    if outlier:
        return                                       # (the streaming kernel rounds its operands differently, see above)
    check(outs[1], ref, dtype, "streaming swap attention L=%d mode=%d" % (L, mode))
    assert (outs[0].float() - outs[1].float()).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("nq_cams,mean", [(4, True), (1, False)])
def test_resident_attention_level0_shape_and_outlier_keys(cuda, nq_cams, mean):
    """The level-0 launch shape of the bench (256 queries x 4 cameras x 256 keys per window) at reduced batch, with score
    outliers that force the deferred softmax rescale both ways: a key tile whose scores exceed the running maximum by far more
    than the 2^8 threshold late in the window, and one early; against the fp32 torch result."""
    B, n, heads, dh = 1, 4, 4, 32
    d = heads * dh
    H = W = 32
    W1 = W2 = 16
    h = w = 16
    w1 = w2 = 8
    q = procedural_input("l0.q", (B, nq_cams, H, W, d), 0)
    k = procedural_input("l0.k", (B, n, h, w, d), 0)
    v = procedural_input("l0.v", (B, n, h, w, d), 0)
    k[0, 3, 5, 3] *= 60.0          # camera 3 = the last key tile of its window: a late spike
    k[0, 0, 9, 9] *= 45.0          # camera 0 = the first key tile: an early spike
    scale = dh ** -0.5
    dtype = torch.bfloat16
    qd, kd, vd = [t.to(cuda).to(dtype) for t in (q, k, v)]
    out = torch.empty((B, H, W, d), device=cuda, dtype=dtype)
    qmap, kmap, omap = ops.tokmap(0, nq_cams, H, W, W1, W2), ops.tokmap(0, n, h, w, w1, w2), ops.tokmap(0, 1, H, W, W1, W2)
    ops.window_attention(qd, kd, vd, out, qmap, kmap, omap, B, heads, scale, d, d, d, d, mean_q=mean, variant=0)
    torch.cuda.synchronize()
    qp = o_fax._window_partition(rnd(q, dtype), W1, W2)
    kp, vp = o_fax._window_partition(rnd(k, dtype), w1, w2), o_fax._window_partition(rnd(v, dtype), w1, w2)
    X, Y = H // W1, W // W2
    qf = qp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, nq_cams * W1 * W2, heads, dh).permute(0, 3, 1, 2, 4)
    kf = kp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, n * w1 * w2, heads, dh).permute(0, 3, 1, 2, 4)
    vf = vp.permute(0, 2, 3, 1, 4, 5, 6).reshape(B, X * Y, n * w1 * w2, heads, dh).permute(0, 3, 1, 2, 4)
    a = torch.matmul((torch.matmul(qf, kf.transpose(-1, -2)) * scale).softmax(-1), vf)
    a = a.permute(0, 2, 3, 1, 4).reshape(B, X, Y, nq_cams, W1, W2, d).mean(3)
    check(out, o_fax._window_reverse(a), dtype, "level-0 resident attention with outlier keys")


@pytest.mark.parametrize("dtype", DTYPES)
def test_agent_max(cuda, dtype):
    x = procedural_input("amax.x", (2, 5, 9, 7, 32), 0)
    y = ops.agent_max(x.to(cuda).to(dtype))
    torch.cuda.synchronize()
    assert torch.equal(y.float().cpu(), rnd(x, dtype).max(dim=1)[0])


@pytest.mark.parametrize("dtype", DTYPES)
def test_global_attention_2d_bias(cuda, dtype):
    B, ws, heads, dh = 2, 8, 2, 32
    d = heads * dh
    qkv = procedural_input("ga.qkv", (B, ws, ws, 3 * d), 0)
    table = procedural_input("ga.table", ((2 * ws - 1) ** 2, heads), 0)
    out = torch.empty((B, ws, ws, d), device=cuda, dtype=dtype)
    m = ops.tokmap(0, 1, ws, ws, ws, ws)
    qd = qkv.to(cuda).to(dtype)
    ops.window_attention(qd, qd, qd, out, m, m, m, B, heads, dh ** -0.5, 3 * d, 3 * d, 3 * d, d, koff=d, voff=2 * d,
                         bias_table=table.to(cuda), bias_L=1)
    torch.cuda.synchronize()
    t = rnd(qkv, dtype).reshape(B, ws * ws, 3 * d)
    qf, kf, vf = [z.reshape(B, ws * ws, heads, dh).permute(0, 2, 1, 3) for z in t.chunk(3, -1)]
    bias = table[torch.from_numpy(o_fax.rel_pos_index_2d(ws))].permute(2, 0, 1)
    s = torch.matmul(qf, kf.transpose(-1, -2)) * dh ** -0.5 + bias
    ref = torch.matmul(s.softmax(-1), vf).permute(0, 2, 1, 3).reshape(B, ws, ws, d)
    check(out, ref, dtype, "global attention")


# ---------------------------------------------------------------------------------------------
@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("c", [32, 128, 256])
def test_layernorm(cuda, dtype, c):
    x = procedural_input("ln.x%d" % c, (5, 37, c), 0, -2, 3)
    g = 0.8 + 0.4 * procedural_input("ln.g%d" % c, (c,), 0, 0, 1)
    b = procedural_input("ln.b%d" % c, (c,), 0, -0.2, 0.2)
    y = ops.layernorm(x.to(cuda).to(dtype), g.to(cuda), b.to(cuda))
    check(y, F.layer_norm(rnd(x, dtype), (c,), g, b, 1e-5), dtype, "layernorm C=%d" % c)


@pytest.mark.parametrize("dtype", DTYPES)
def test_layernorm_without_affine(cuda, dtype):
    x = procedural_input("ln0.x", (3, 50, 128), 0, -2, 3)
    y = ops.layernorm(x.to(cuda).to(dtype), None, None)
    check(y, F.layer_norm(rnd(x, dtype), (128,), None, None, 1e-5), dtype, "layernorm (no affine)")


@pytest.mark.parametrize("dtype", DTYPES)
def test_mean_layernorm(cuda, dtype):
    x = procedural_input("mln.x", (2, 3, 20, 64), 0, -2, 3)
    g = 0.8 + 0.4 * procedural_input("mln.g", (64,), 0, 0, 1)
    b = procedural_input("mln.b", (64,), 0, -0.2, 0.2)
    y = ops.mean_layernorm(x.to(cuda).to(dtype), g.to(cuda), b.to(cuda))
    check(y, F.layer_norm(rnd(x, dtype).mean(1), (64,), g, b, 1e-5), dtype, "mean+layernorm")


@pytest.mark.parametrize("dtype", DTYPES)
def test_maxpool(cuda, dtype):
    x = procedural_input("mp.x", (2, 64, 18, 22), 0)
    y = ops.maxpool3x3s2(nhwc(x).to(cuda).to(dtype))
    check(y.permute(0, 3, 1, 2), F.max_pool2d(rnd(x, dtype), 3, 2, 1), dtype, "maxpool")


@pytest.mark.parametrize("dtype", DTYPES)
def test_layout_roundtrip(cuda, dtype):
    x = procedural_input("lay.x", (2, 24, 5, 7), 0)
    y = ops.to_nhwc(x.to(cuda), dtype)
    check(y, rnd(nhwc(x), dtype), dtype, "to_nhwc")
    assert ops.to_nhwc(y.permute(0, 3, 1, 2), dtype).data_ptr() == y.data_ptr()      # zero-copy for channels-last views
    z = ops.from_nhwc(y, torch.float32)
    check(z, rnd(x, dtype), dtype, "from_nhwc")
    sl = x.to(cuda)[:, 4:20]                                                          # a strided (non-contiguous) view
    check(ops.to_nhwc(sl, dtype), rnd(nhwc(x[:, 4:20]), dtype), dtype, "to_nhwc strided")


@pytest.mark.parametrize("dtype", DTYPES)
def test_regroup(cuda, dtype):
    x = procedural_input("rg.x", (5, 6, 6, 8), 0)
    rl = torch.tensor([2, 3], dtype=torch.int32)
    y, mask = ops.regroup(x.to(cuda).to(dtype), rl.to(cuda), 4)
    ref, rmask = o_sttf.regroup(rnd(x, dtype), rl, 4)
    check(y, ref, dtype, "regroup")
    assert torch.equal(mask.cpu(), rmask.float())


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("hw", [(16, 16), (12, 16)])
def test_sttf_warp_and_mask(cuda, dtype, hw):
    import cases
    h, w = hw
    x, tm, cav = cases.sttf_inputs(h, w)                                     # x: (1, L, C, h, w)
    s = cases.STTF
    xd = x.permute(0, 1, 3, 4, 2).contiguous().to(cuda).to(dtype)           # (1, L, h, w, C)
    y, com = ops.sttf_warp(xd, tm.to(cuda), cav.float().to(cuda), s["resolution"], s["downsample_rate"])
    ref = o_sttf.sttf(rnd(x, dtype), tm, s["resolution"], s["downsample_rate"])
    check(y, ref, dtype, "sttf warp %dx%d" % (h, w), scale=1.0)
    refm = o_sttf.roi_and_cav_mask(tuple(ref.shape), cav, tm, s["resolution"], s["downsample_rate"])
    assert torch.equal(com.cpu(), refm.float()), "ROI mask differs in %d cells" % (com.cpu() != refm.float()).sum()
    g = np.load(__import__("os").path.join(__import__("os").path.dirname(cases.__file__), "gv6_sttf_regroup.npz"))
    if dtype == torch.float32:
        check(y, torch.from_numpy(g["sttf_%dx%d" % (h, w)]), dtype, "sttf vs golden", scale=1.0)
        assert np.array_equal(com.cpu().numpy(), g["mask_%dx%d" % (h, w)])


@pytest.mark.parametrize("b,n,H,W,bcast", [(2, 3, 16, 24, False), (5, 4, 32, 32, True), (1, 1, 8, 4, False)])
def test_bev_query_wave_level_kernel(cuda, b, n, H, W, bcast):
    """bev_query.hip (the 128 -> 128 query side of FAX cross attention #1 in one launch: embedding, + x, bf16 rounding, LayerNorm,
    to_q; fax_modules.py:370-375,387-388,201) against the embedding kernel + GEMM (same roundings: equal to bf16 rounding of the
    result) and against an fp32 torch evaluation of the reference formulas"""
    import cases
    d, dtype = 128, torch.bfloat16
    _, E = cases.camera_geometry(b * n, n, 112, 120)
    Ec = E.reshape(b * n, 4, 4).contiguous()
    world = procedural_input("bq.world", (2, H * W), 0, -40, 40)
    w_bev, b_bev, w_cam = procedural_input("bq.wb", (d, 2), 0), procedural_input("bq.bb", (d,), 0), procedural_input("bq.wc", (d, 4), 0)
    x = procedural_input("bq.x", (1 if bcast else b, H * W, d), 0)
    wq = procedural_input("bq.wq", (d, d), 0) * math.sqrt(3.0 / d)
    bq = procedural_input("bq.bq", (d,), 0, -0.2, 0.2)

    class LN(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("bq.g", (d,), 0, 0, 1), procedural_input("bq.be", (d,), 0, -0.2, 0.2), 1e-5
    plan = ops.ConvPlan(wq, bq, dtype=dtype, device=cuda, ln=LN)
    xd = x.to(cuda).to(dtype)
    xin = xd.expand(b, H * W, d) if bcast else xd
    args = (Ec.to(cuda), world.to(cuda), w_bev.to(cuda), b_bev.to(cuda), w_cam.to(cuda))
    keep = (ops.USE_EMBED_GEMM, ops.USE_EMBED_GEMM3)
    try:
        ops.USE_EMBED_GEMM, ops.USE_EMBED_GEMM3 = False, False
        two = ops.bev_embed_linear(*args, xin, n, plan)
        ops.USE_EMBED_GEMM3 = True
        with ops.LaunchProfile() as prof:
            one = ops.bev_embed_linear(*args, xin, n, plan)
        assert len(prof.records) == 1, "the fused query must be ONE launch"
    finally:
        ops.USE_EMBED_GEMM, ops.USE_EMBED_GEMM3 = keep
    assert one.shape == (b, n, H * W, d) and one.dtype == dtype
    # fp32 reference on the bf16-rounded x / weights: query = normalise(w_bev . world + b_bev - w_cam . c) + x ; LN ; Linear
    c = Ec[:, :, 3]                                                        # (b n, 4): camera position column
    emb = (w_bev @ world)[None] + b_bev[None, :, None] - (c @ w_cam.t())[:, :, None]        # (bn, d, hw)
    emb = emb / (emb.norm(dim=1, keepdim=True) + 1e-7)
    xr = rnd(x, dtype)
    q = emb.permute(0, 2, 1).reshape(b, n, H * W, d) + (xr[0][None, None] if bcast else xr[:, None])
    q = rnd(q, dtype)                                                      # the query as the unfused path stores it
    qn = F.layer_norm(q, (d,), LN.weight, LN.bias, LN.eps)
    ref = qn @ wq.t() + bq
    s = ref.abs().max().item()
    assert (one.float().cpu() - ref).abs().max().item() <= 1e-2 * s
    assert (one.float() - two.float()).abs().max().item() <= 1e-2 * s


@pytest.mark.parametrize("dtype", DTYPES)
def test_bev_embed_fused_into_q_projection(cuda, dtype):
    """to_q(LayerNorm(x + bev embedding)) with the embedding produced inside the GEMM == bev_embed kernel + GEMM"""
    import cases
    b, n = 2, 3
    d = 128 if dtype == torch.bfloat16 else 64
    H, W = 16, 24                                                           # hw = 384 (multiple of 128)
    _, E = cases.camera_geometry(b * n, n, 112, 120)
    E = E.reshape(b * n, 4, 4).contiguous().to(cuda)
    world = procedural_input("be.world", (2, H * W), 0, -40, 40).to(cuda)
    w_bev = procedural_input("be.wb", (d, 2), 0).to(cuda)
    b_bev = procedural_input("be.bb", (d,), 0).to(cuda)
    w_cam = procedural_input("be.wc", (d, 4), 0).to(cuda)
    x = procedural_input("be.x", (b, H * W, d), 0).to(cuda).to(dtype)
    wq = procedural_input("be.wq", (96, d), 0) * math.sqrt(3.0 / d)

    class LN(object):
        weight, bias, eps = 0.8 + 0.4 * procedural_input("be.g", (d,), 0, 0, 1), procedural_input("be.be", (d,), 0, -0.2, 0.2), 1e-5
    plan = ops.ConvPlan(wq, procedural_input("be.bq", (96,), 0, -0.2, 0.2), dtype=dtype, device=cuda, ln=LN)
    assert ops.ln_fusable(plan)
    keep = (ops.USE_EMBED_GEMM, ops.USE_EMBED_GEMM3)
    ops.USE_EMBED_GEMM = ops.USE_EMBED_GEMM3 = False
    try:
        y2 = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, x, n, plan)    # bev_embed kernel + GEMM
        ops.USE_EMBED_GEMM = True
        y = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, x, n, plan)     # produced inside the 128-row GEMM
    finally:
        ops.USE_EMBED_GEMM, ops.USE_EMBED_GEMM3 = keep
    assert y.shape == (b, n, H * W, 96)
    s = y2.float().abs().max().item()
    assert (y.float() - y2.float()).abs().max().item() <= (1e-2 if dtype == torch.bfloat16 else 1e-5) * s
    if dtype == torch.bfloat16:
        # produced inside the 32-row GEMM, also with the batch-broadcast prior of pyramid level 0
        prior = x[0]
        view = prior[None].expand(b, H * W, d)
        yr = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, view, n, plan)       # two launches
        ops.USE_EMBED_GEMM3 = 2                   # (2 = also for widths the wave-level kernel does not take)
        try:
            y3 = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, x, n, plan)
            yb = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, view, n, plan)
            yc = ops.bev_embed_linear(E, world, w_bev, b_bev, w_cam, view.contiguous(), n, plan)
        finally:
            ops.USE_EMBED_GEMM3 = keep[1]
        assert y3.shape == (b, n, H * W, 96)
        assert (y3.float() - y2.float()).abs().max().item() <= 1e-2 * s
        assert torch.equal(yb, yc)
        assert (yb.float() - yr.float()).abs().max().item() <= 1e-2 * yr.float().abs().max().item()


@pytest.mark.parametrize("dtype", DTYPES)
def test_regroup_fused_into_sttf_warp(cuda, dtype):
    """record_len-aware warp (regroup + STTF + ROI mask in one launch) == regroup kernel followed by the warp kernel,
    bit for bit, for ragged record_len (absent agents -> zeros / mask 0)"""
    import cases
    h, w, c, L = 12, 16, 16, 4
    rl = torch.tensor([3, 1, 4], dtype=torch.int32)
    n = int(rl.sum())
    x = procedural_input("rgs.x", (n, h, w, c), 0).to(cuda).to(dtype)
    _, tm1, _ = cases.sttf_inputs(h, w)                                      # (1, L0, 4, 4)
    tm = tm1[:, :1].expand(3, L, 4, 4).clone()
    for b in range(3):
        for l in range(L):
            tm[b, l] = tm1[0, (b + l) % tm1.shape[1]]
    tm = tm.contiguous().to(cuda)
    s = cases.STTF
    xg, cav = ops.regroup(x, rl.to(cuda), L)
    y_ref, com_ref = ops.sttf_warp(xg, tm, cav, s["resolution"], s["downsample_rate"])
    y, com, cav2 = ops.sttf_warp(x, tm, None, s["resolution"], s["downsample_rate"], record_len=rl.to(cuda), max_cav=L)
    assert torch.equal(cav2, cav) and torch.equal(y, y_ref) and torch.equal(com, com_ref)


@pytest.mark.parametrize("dtype", DTYPES)
def test_ray_and_bev_embed(cuda, dtype):
    import cases
    b, n, d, h, w, H, W = 2, 2, 64, 14, 15, 24, 16
    I, E = cases.camera_geometry(b * n, n, 112, 120)
    I_inv = I.inverse().reshape(b * n, 3, 3).contiguous()
    E = E.reshape(b * n, 4, 4).contiguous()
    w_img = procedural_input("re.wimg", (d, 4), 0)
    w_cam = procedural_input("re.wcam", (d, 4), 0)
    plane = o_fax.image_plane(h, w, 112, 120)
    y = ops.ray_embed(I_inv.to(cuda), E.to(cuda), plane.reshape(3, -1).contiguous().to(cuda), w_img.to(cuda),
                      w_cam.to(cuda), h * w, d, dtype)
    c_embed = F.conv2d(E[:, :, 3].reshape(b * n, 4, 1, 1), w_cam.reshape(d, 4, 1, 1))
    cam = F.pad(I_inv[:, None] @ plane.reshape(1, 1, 3, h * w), (0, 0, 0, 1), value=1)[:, 0]
    dd = (E @ cam).reshape(b * n, 4, h, w)
    emb = F.conv2d(dd, w_img.reshape(d, 4, 1, 1)) - c_embed
    emb = emb / (emb.norm(dim=1, keepdim=True) + 1e-7)
    check(y.reshape(b * n, h, w, d), nhwc(emb), dtype, "ray embed", scale=1.0)

    grid = o_fax.bev_grids(1.0, H, W, 50, 40, 0.0, [1])[0] if False else \
        o_fax.bev_grids(bev_height=H, bev_width=W, h_meters=50, w_meters=40, offset=0.0, upsample_scales=[1])[0]
    w_bev = procedural_input("re.wbev", (d, 2), 0)
    b_bev = procedural_input("re.bbev", (d,), 0, -0.2, 0.2)
    x = procedural_input("re.x", (b, H, W, d), 0)
    q = ops.bev_embed(E.to(cuda), grid[:2].reshape(2, -1).contiguous().to(cuda), w_bev.to(cuda), b_bev.to(cuda),
                      w_cam.to(cuda), x.reshape(b, H * W, d).to(cuda).to(dtype), n)
    we = F.conv2d(grid[:2][None], w_bev.reshape(d, 2, 1, 1), b_bev) - c_embed
    we = we / (we.norm(dim=1, keepdim=True) + 1e-7)
    ref = nhwc(we).reshape(b, n, H, W, d) + rnd(x, dtype)[:, None]
    check(q.reshape(b, n, H, W, d), ref, dtype, "bev embed")
    # the first pyramid level's x is one learned prior repeated over the batch: a stride-0 view, never materialised
    prior = x[0].reshape(H * W, d).to(cuda).to(dtype)
    args = (E.to(cuda), grid[:2].reshape(2, -1).contiguous().to(cuda), w_bev.to(cuda), b_bev.to(cuda), w_cam.to(cuda))
    qb = ops.bev_embed(*args, prior[None].expand(b, H * W, d), n)
    qc = ops.bev_embed(*args, prior[None].expand(b, H * W, d).contiguous(), n)
    assert torch.equal(qb, qc)


def test_small_block_gemm_row_tiles_agree(cuda):
    """cobevt_linear_rows_small_k with 32- and 64-row workgroups: the same MFMAs on the same operands, bit-identical results
    (K = 128 with LayerNorm, K = 320 with the BN -> ReLU pre-activation and a ragged last row tile)"""
    dtype = torch.bfloat16
    for (m, k, n_, ln, pre) in ((1000, 128, 256, True, False), (333, 320, 96, False, True)):
        x = procedural_input("g3.x", (m, k), 0, -2, 2).to(cuda).to(dtype)
        w = procedural_input("g3.w", (n_, k), 0) * math.sqrt(3.0 / k)

        class LN(object):
            weight, bias, eps = 0.8 + 0.4 * procedural_input("g3.g", (k,), 0, 0, 1), procedural_input("g3.b", (k,), 0, -0.2, 0.2), 1e-5
        bn = None
        if pre:
            bn = torch.nn.BatchNorm2d(k).eval()
            bn.weight.data.copy_(0.8 + 0.4 * procedural_input("g3.bg", (k,), 0, 0, 1))
            bn.running_mean.data.copy_(procedural_input("g3.bm", (k,), 0, -0.3, 0.3))
        plan = ops.ConvPlan(w, procedural_input("g3.bias", (n_,), 0, -0.2, 0.2), dtype=dtype, device=cuda, ln=LN if ln else None,
                            pre_bn=bn, pre_relu=pre)
        keep = ops.GEMM_ROWS3_ROWS64_MIN_M
        try:
            ops.GEMM_ROWS3_ROWS64_MIN_M = 0
            y32 = ops.linear(x, plan)
            ops.GEMM_ROWS3_ROWS64_MIN_M = 1
            y64 = ops.linear(x, plan)
        finally:
            ops.GEMM_ROWS3_ROWS64_MIN_M = keep
        assert torch.equal(y32, y64)
        ops.USE_GEMM_ROWS3 = False
        try:
            y1 = ops.linear(x, plan)                        # the 128 x 128-tile kernel
        finally:
            ops.USE_GEMM_ROWS3 = True
        assert (y32.float() - y1.float()).abs().max().item() <= 1e-2 * y1.float().abs().max().item()


@pytest.mark.parametrize("dtype", DTYPES)
@pytest.mark.parametrize("cout,cin,h,w", [(2, 32, 256, 256), (3, 32, 37, 21), (1, 64, 16, 16), (4, 128, 20, 33)])
def test_head_conv_direct_kernel(cuda, dtype, cout, cin, h, w):
    """BevSegHead's 3x3 convs (2 / 3 classes) on the direct small-Cout kernel: against torch and against the implicit-GEMM path"""
    g = torch.Generator().manual_seed(4)
    wt, b = torch.randn(cout, cin, 3, 3, generator=g) / (3 * cin ** 0.5), torch.randn(cout, generator=g)
    x = torch.randn(2, h, w, cin, generator=g)
    plan = ops.ConvPlan(wt, b, stride=1, pad=1, store_mode=2, dtype=dtype, device=cuda)
    assert (plan.wgt_head is not None) == (cin * (4 if dtype == torch.float32 else 2) <= 256)      # one pixel row <= 256 bytes
    xd = x.to(cuda).to(dtype)
    y = ops.conv2d(xd, plan)
    ops.USE_HEAD_CONV = False
    try:
        y_igemm = ops.conv2d(xd, plan)
    finally:
        ops.USE_HEAD_CONV = True
    ref = torch.nn.functional.conv2d(xd.float().permute(0, 3, 1, 2), wt.to(cuda).to(dtype).float(), b.to(cuda), padding=1)
    assert y.dtype == torch.float32 and tuple(y.shape) == (2, cout, h, w)
    from util import assert_close
    assert_close(y, ref, 1e-4 if dtype == torch.float32 else 2e-3, "head conv vs torch")
    assert_close(y, y_igemm, 1e-4 if dtype == torch.float32 else 2e-3, "head conv vs implicit GEMM")


@pytest.mark.parametrize("agents,window,hw,mlp,use_mask", [(5, 8, 32, 256, True), (5, 8, 32, 256, False), (3, 8, 16, 128, True),
                                                           (5, 4, 16, 256, True), (2, 8, 24, 256, True)])
def test_swap_fusion_stage_single_launch(cuda, agents, window, hw, mlp, use_mask):
    """cobevt_swap_fusion_stage (one launch per SwapFusionBlock half: attention + row chain + next to_qkv) against the oracle
    (swap_fusion_modules.py:87-128,165-192, base_transformer.py:102-124) and against the two-launch path it replaces - the
    camera config's shape (5 agents x 8x8 windows = 320 keys, 32x32 map), key counts that need padding to the 32-key tile
    (3 x 64 = 192, 5 x 16 = 80, 2 x 64 = 128), hidden widths of one and two 128-column passes, with and without the key mask
    (incl. a fully masked agent and a half-masked one)."""
    import cases
    from cobevt_amd import host, synth
    from cobevt_amd.synth import fill_module_
    from util import rel_err
    args = dict(input_dim=128, mlp_dim=mlp, agent_size=agents, window_size=window, dim_head=32, drop_out=0.1, depth=2, mask=use_mask)
    enc = fill_module_(host.SwapFusionEncoder(args), cases.SEED).eval()
    x = synth.procedural_input("stage.x", (2, agents, 128, hw, hw), cases.SEED, -2.0, 2.0)
    mask = None
    if use_mask:
        mask = torch.ones(2, hw, hw, 1, agents)
        mask[0, :, :, :, agents - 1] = 0                         # an absent agent
        ii, jj = torch.meshgrid(torch.arange(hw), torch.arange(hw), indexing="ij")
        mask[1, :, :, 0, 1] = (jj > ii // 2).float()             # an agent whose warped map covers part of the ego's
    ref = o_swap.swap_fusion_encoder(enc.state_dict(), "", args, x, mask)
    enc = enc.to(cuda)
    xm = (x.to(cuda), mask.to(cuda) if use_mask else None)
    with host.compute_dtype(torch.bfloat16):
        launches = []
        with ops.LaunchProfile() as prof:
            y = enc(*xm)
        launches = sorted(prof.summary())
        assert "swap_stage" in launches and "attention" not in launches and "row_chain" not in launches, launches
        old = ops.USE_SWAP_STAGE
        ops.USE_SWAP_STAGE = False
        try:
            y2 = enc(*xm)
        finally:
            ops.USE_SWAP_STAGE = old
    e, e2, d = rel_err(y, ref), rel_err(y2, ref), rel_err(y, y2)
    print("swap stage %s: fused vs oracle %.2e, two-launch vs oracle %.2e, fused vs two-launch %.2e" % ((agents, window, hw, mlp, use_mask), e, e2, d))
    assert e <= 1e-2 and e <= 1.5 * e2 + 2e-3, (e, e2)


@pytest.mark.parametrize("rows,with_res", [(4096, True), (4096, False), (1000, True), (40960, True), (32768 + 77, False)])
def test_projection_chain_single_launch(cuda, rows, with_res):      # (>= 32768 rows: the wave-level kernel, proj_chain128.hip)
    """cobevt_proj_chain (BN -> ReLU -> 1x1 conv (+ ray embedding) -> LayerNorm -> stacked to_k / to_v without materialising the key
    / value map; fax_modules.py:281-292,377-396,201-205) against the two launches it replaces and against fp32 torch"""
    import torch.nn as nn
    from cobevt_amd import host
    from cobevt_amd.host import runtime as rt
    g = torch.Generator().manual_seed(4)
    bn, conv = nn.BatchNorm2d(128), nn.Conv2d(128, 128, 1, bias=False)
    ln, lin = nn.LayerNorm(128), nn.Linear(128, 256, bias=True)
    with torch.no_grad():
        bn.weight.copy_(torch.rand(128, generator=g) + 0.5); bn.bias.copy_(torch.randn(128, generator=g) * 0.2)
        bn.running_mean.copy_(torch.randn(128, generator=g) * 0.3); bn.running_var.copy_(torch.rand(128, generator=g) + 0.5)
        ln.weight.copy_(torch.rand(128, generator=g) + 0.5); ln.bias.copy_(torch.randn(128, generator=g) * 0.2)
    x = torch.randn(rows, 128, generator=g)
    res = torch.randn(rows, 128, generator=g) if with_res else None
    xb = x.to(torch.bfloat16).to(cuda)
    rb = res.to(torch.bfloat16).to(cuda) if with_res else None
    owner = host.runtime.HipModule()
    owner.bn, owner.conv, owner.ln, owner.lin = bn.eval(), conv, ln, lin
    owner = owner.to(cuda)
    with host.compute_dtype(torch.bfloat16):
        pp = rt.conv_plan(owner, "p", owner.conv, pre_bn=owner.bn)
        pn = rt.linear_plan(owner, "n", owner.lin, ln=owner.ln)
        assert ops.proj_chain_fusable(xb, pp, pn, rb)
        fused = ops.proj_chain(xb, pp, pn, residual=rb)
        key = ops.conv2d(xb.reshape(1, 1, rows, 128), pp, residual=None if rb is None else rb.reshape(1, 1, rows, 128)).reshape(rows, 128)
        two = ops.linear(key, pn)
    with torch.no_grad():                                  # fp32 torch on the device over the same bf16-rounded inputs
        y = torch.relu(owner.bn(xb.float().t().reshape(1, 128, rows, 1))).reshape(128, rows).t() @ owner.conv.weight.reshape(128, 128).t()
        if with_res:
            y = y + rb.float()
        ref = owner.lin(owner.ln(y.to(torch.bfloat16).float())).cpu()
    s = float(ref.abs().max())
    assert (fused.float().cpu() - ref).abs().max().item() <= 1e-2 * s
    assert (fused.float() - two.float()).abs().max().item() <= 1e-2 * s


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
@pytest.mark.parametrize("use_bias,ncam", [(True, 1), (False, 4)])
def test_attention_key_split_matches_single_pass(cuda, dtype, use_bias, ncam):
    """cobevt_window_attention_ksplit (keys of a window shared out over 2 / 4 workgroups per query tile + log-sum-exp merge) against the
    single-pass streaming kernel and a dense fp32 softmax: the FAX global attention shape (one 32 x 32 window = 1024 keys, 2-D
    relative position bias, fax_modules.py:137-171) and the level-2 cross attention shape (1024 queries x 4 cameras x 16 x 16 keys,
    fax_modules.py:211-237)"""
    g = torch.Generator().manual_seed(6)
    B, heads, d = 2, 4, 128
    if ncam == 1:
        qmap = kmap = ops.tokmap(0, 1, 32, 32, 32, 32)
        nq = nk = 1024
        rows_q = rows_k = B * 1024
    else:
        qmap, kmap = ops.tokmap(0, 1, 32, 32, 32, 32), ops.tokmap(0, ncam, 16, 16, 16, 16)
        nq, nk = 1024, ncam * 256
        rows_q, rows_k = B * 1024, B * ncam * 256
    q = (torch.randn(rows_q, d, generator=g) * 1.5).to(dtype).to(cuda)
    k = (torch.randn(rows_k, d, generator=g) * 1.5).to(dtype).to(cuda)
    v = torch.randn(rows_k, d, generator=g).to(dtype).to(cuda)
    table = (torch.randn(63 * 63, heads, generator=g) * 0.5).to(cuda) if use_bias else None
    outs = {}
    for ks in (0, 2, 4):
        out = torch.empty(rows_q, d, device=cuda, dtype=dtype)
        with ops.LaunchProfile() as prof:
            ops.window_attention(q, k, v, out, qmap, kmap, qmap, B, heads, 32 ** -0.5, d, d, d, d, bias_table=table, bias_L=1, ksplit=ks)
        name = list(prof.summary(by_shape=True))[0]
        assert ("ks%d" % ks in name) == (ks > 1), name
        outs[ks] = out.float().cpu()
    # dense reference: per (batch, head) softmax(q k^T / sqrt(32) + bias) v over the window's keys in kernel token order
    qf, kf, vf = q.float().cpu().reshape(B, nq, heads, 32), k.float().cpu().reshape(B, nk, heads, 32), v.float().cpu().reshape(B, nk, heads, 32)
    s = torch.einsum("bqhd,bkhd->bhqk", qf, kf) * 32 ** -0.5
    if use_bias:
        idx = ops.attention_bias_index(qmap, kmap, 1, cuda).cpu().long()
        s = s + table.cpu()[idx].permute(2, 0, 1)[None]
    ref = torch.einsum("bhqk,bkhd->bqhd", s.softmax(-1), vf).reshape(B * nq, d)
    tol_ = 1e-2 if dtype == torch.bfloat16 else 1e-4
    scale = float(ref.abs().max())
    if dtype == torch.bfloat16 and not use_bias:
        # the same windows with K / V resident in LDS (attention_resident.hip: plain bf16 windows of 513 .. 1024 keys, opt-in switch)
        saved = ops.USE_ATTN_BIG_RESIDENT
        ops.USE_ATTN_BIG_RESIDENT = True
        try:
            out = torch.empty(rows_q, d, device=cuda, dtype=dtype)
            with ops.LaunchProfile() as prof:
                ops.window_attention(q, k, v, out, qmap, kmap, qmap, B, heads, 32 ** -0.5, d, d, d, d)
            assert "ks" not in list(prof.summary(by_shape=True))[0]
            outs["resident"] = out.float().cpu()
        finally:
            ops.USE_ATTN_BIG_RESIDENT = saved
    for ks, o in outs.items():
        assert float((o - ref).abs().max()) <= tol_ * scale, (ks, float((o - ref).abs().max()) / scale)


# ----------------------------------------------------------------------------------------------------------------------
# fp32 storage on the split-bf16 matrix path (libcobevt_hip_f32s.so: csrc/common.hpp COBEVT_F32_SPLIT, cobevt_amd/build.py).
# The same kernel tests as exact fp32, at the same 2e-4 tolerance, with the second library selected.
# ----------------------------------------------------------------------------------------------------------------------
def _split_cases():
    f32 = torch.float32
    return [
        ("conv3x3_lds_staged", lambda c: test_conv_3x3_bias_relu(c, f32)),
        ("conv3x3_s2_residual", lambda c: test_conv_3x3_s2_bn_residual_relu(c, f32)),           # -> generic implicit GEMM in this library
        ("conv3x3_s2_128_256", lambda c: test_conv_3x3_s2_strip_kernel(c, f32, 128, 256, 2, 17, 37)),
        ("conv1x1_gelu", lambda c: test_conv_1x1_big_tile_gelu(c, f32)),
        ("stem_pool", lambda c: test_stem_fused_with_maxpool(c, f32, 2, 64, 64)),
        ("stem", lambda c: test_stem_space_to_depth_kernel(c, f32, 1, 38, 50)),
        ("conv3x3_upsample", lambda c: test_conv_3x3_nearest_upsample(c, f32)),
        ("conv3x3_unshuffle", lambda c: test_conv_3x3_pixel_unshuffle(c, f32)),
        ("conv3x3_strips_256_192", lambda c: test_conv3x3_patch_kernel_shapes(c, f32, 256, 192, 9, 21)),
        ("conv3x3_strips_variant_150", lambda c: test_conv3x3_wfrag_tile_variants(c, f32, 150)),
        ("basicblock_64", lambda c: test_basicblock_fused(c, f32, 64, 2, 24, 40)),
        ("basicblock_128", lambda c: test_basicblock_fused(c, f32, 128, 3, 16, 16)),
        ("gemm_rows_layernorm", lambda c: test_gemm_rows_fused_layernorm(c, f32, 128, 384, 130)),
        ("linear_ragged", lambda c: test_linear_ragged_rows(c, f32)),
        ("cross_window_attention", lambda c: test_attention_prepartitioned_long_keys(c, f32)),
        ("swap_attention_bias_mask", lambda c: test_swap_attention_bias_mask(c, f32, 1)),
        ("global_attention", lambda c: test_global_attention_2d_bias(c, f32)),
        ("bev_embed_q_projection", lambda c: test_bev_embed_fused_into_q_projection(c, f32)),
    ]


@pytest.mark.parametrize("name", [n for n, _ in _split_cases()])
def test_split_bf16_matrix_path_kernels(cuda, name):
    from cobevt_amd import host, lib
    fn = dict(_split_cases())[name]
    with host.compute_dtype("fp32_split"):
        assert lib.get_variant() == "f32s"
        fn(cuda)
    assert lib.get_variant() == "" and host.get_compute_mode() == "bf16"


def test_split_bf16_matrix_path_is_the_one_that_runs(cuda):
    """the two libraries must give DIFFERENT low-order bits on the same GEMM (else the second library is not what ran), both
    within fp32-parity distance of the fp64 product, the split one within its 2^-17-per-operand bound"""
    from cobevt_amd import host
    x = procedural_input("split.x", (1, 512, 128), 0, -2.0, 2.0)
    w = procedural_input("split.w", (256, 128), 0, -1.0, 1.0)
    ref = (x.double() @ w.double().t())
    plan = ops.ConvPlan(w, None, dtype=torch.float32, device=cuda)
    xd = x.to(cuda)
    exact = ops.linear(xd, plan).cpu().double()
    with host.compute_dtype("fp32_split"):
        split = ops.linear(xd, plan).cpu().double()
    scale = ref.abs().max()
    e_exact, e_split = ((exact - ref).abs().max() / scale).item(), ((split - ref).abs().max() / scale).item()
    assert not torch.equal(exact, split)
    assert e_exact <= 2e-6 and e_split <= 4e-5, (e_exact, e_split)
    # error model: every product carries <= 2 * 2^-17 relative error -> |err| <= 2^-16 * sum |x||w|
    bound = (x.abs().double() @ w.abs().double().t()) * 2.0 ** -16 + 1e-6 * scale
    assert ((split - ref).abs() <= bound).all()


# ----------------------------------------------------------------------------------------------------------------------
# fp32 storage, ONE fp16 MFMA per piece (libcobevt_hip_f32h.so: csrc/common.hpp COBEVT_F32_SPLIT == 2; round 6): the ResNet encoder's
# library under host.set_compute_dtype("fp32_fast").  The arithmetic is pinned, not just bounded: a kernel must equal the fp64 convolution
# with the folded weights ROUNDED TO fp16 and the activations either UNROUNDED (the (hi, lo) form: fp16 pairs = 22 bits - the stem, the
# 64-cout strip tiles, the dense-row kernel) or ROUNDED TO fp16 as well (the packed form: two k-groups per MFMA - the 128-cout strip tiles
# and the BasicBlocks; tests/precision_emul.py fp16_e2 measures the same end-to-end error for both), to 3e-5 of the output scale - and must
# differ from the split-bf16 library, or the third library is not what ran.
# ----------------------------------------------------------------------------------------------------------------------
def _h(w):
    return w.to(torch.float16).to(torch.float32)


def _enc_lib():
    from cobevt_amd import host, lib

    class _Scope(object):
        def __enter__(self):
            self.a = host.compute_dtype("fp32_fast")
            self.a.__enter__()
            self.b = lib.encoder_scope()
            self.b.__enter__()
            assert lib.get_variant() == "f32h"

        def __exit__(self, *e):
            self.b.__exit__(*e)
            self.a.__exit__(*e)
    return _Scope()


def _close_f16w(y, ref, what, rel=3e-5):
    y, ref = y.detach().double().cpu(), ref.double()
    assert y.shape == ref.shape and torch.isfinite(y).all(), what
    e = ((y - ref).abs().max() / ref.abs().max()).item()
    assert e <= rel, "%s: %.3e of the output scale (the fp16-weight product should be met to %.0e)" % (what, e, rel)
    return e


@pytest.mark.parametrize("cin,cout,n,h,w,stride", [(256, 192, 2, 9, 21, 1), (256, 256, 5, 32, 32, 1), (512, 512, 3, 16, 16, 1), (128, 128, 2, 20, 24, 1),
                                                  (128, 256, 2, 17, 37, 2), (64, 128, 2, 24, 40, 2), (256, 512, 3, 16, 32, 2)])
def test_fp16_weight_matrix_path_conv3x3(cuda, cin, cout, n, h, w, stride):
    """the strip / LDS-staged 3x3 kernels (stride 1 and 2, ragged strips, cout tails, BN folded, residual + ReLU) in the third library"""
    from cobevt_amd import host
    f32 = torch.float32
    x = procedural_input("f16w.x%d" % cin, (n, cin, h, w), 0, -2.0, 2.0)
    wt = procedural_input("f16w.w%d_%d" % (cin, cout), (cout, cin, 3, 3), 0) * math.sqrt(3.0 / (cin * 9))
    plan = ops.ConvPlan(wt, None, bn=FakeBN(cout, "f16w.bn%d" % cout), stride=stride, pad=1, act=1, dtype=f32, device=cuda)
    wref = plan.wgt.float().cpu()[:, :plan.K].reshape(cout, 3, 3, cin).permute(0, 3, 1, 2)
    ho, wo = (h - 1) // stride + 1, (w - 1) // stride + 1
    res = procedural_input("f16w.res", (n, cout, ho, wo), 0)
    ref = F.relu(F.conv2d(x.double(), _h(wref).double(), plan.bias.cpu().double(), stride=stride, padding=1) + res.double())
    ref_packed = F.relu(F.conv2d(_h(x).double(), _h(wref).double(), plan.bias.cpu().double(), stride=stride, padding=1) + res.double())
    ref_exact = F.relu(F.conv2d(x.double(), wref.double(), plan.bias.cpu().double(), stride=stride, padding=1) + res.double())
    xd, rd = nhwc(x).to(cuda), nhwc(res).to(cuda)
    with _enc_lib():
        y = ops.conv2d(xd, plan, residual=rd)
    with host.compute_dtype("fp32_split"):
        ys = ops.conv2d(xd, plan, residual=rd)
    yd = y.permute(0, 3, 1, 2).double().cpu()
    e_pair, e_packed = [((yd - r).abs().max() / r.abs().max()).item() for r in (ref, ref_packed)]
    assert min(e_pair, e_packed) <= 3e-5, "conv3x3 %d->%d s%d matches neither defined form: (hi, lo) activations %.2e, packed fp16 %.2e" % (cin, cout, stride, e_pair, e_packed)
    assert not torch.equal(y, ys)
    # against the UNROUNDED operands the result is off by their fp16 rounding: <= 2 x 2^-11 sum |x||w| per output, and visibly so
    bound = F.conv2d(x.abs().double(), wref.abs().double(), None, stride=stride, padding=1) * 2.0 ** -10 + 1e-6 * ref.abs().max()
    d = (y.permute(0, 3, 1, 2).double().cpu() - ref_exact).abs()
    assert (d <= bound).all() and d.max() > 2e-5 * ref_exact.abs().max()


@pytest.mark.parametrize("c,n,h,w", [(64, 2, 24, 40), (128, 3, 16, 16), (64, 1, 13, 21), (128, 2, 9, 35)])
def test_fp16_weight_matrix_path_basicblock(cuda, c, n, h, w):
    """the fused BasicBlock in the packed form: both convolutions with fp16 weights AND fp16 activations (the intermediate map rounded to
    fp16 where conv2 reads it), fp32 accumulation, the residual added from the fp32 input"""
    f32 = torch.float32
    x = procedural_input("bb.x", (n, c, h, w), 0)
    w1 = procedural_input("bb.w1", (c, c, 3, 3), 0) * math.sqrt(3.0 / (c * 9))
    w2 = procedural_input("bb.w2", (c, c, 3, 3), 0) * math.sqrt(3.0 / (c * 9))
    p1 = ops.ConvPlan(w1, None, bn=FakeBN(c, "bb.bn1"), stride=1, pad=1, act=1, dtype=f32, device=cuda)
    p2 = ops.ConvPlan(w2, None, bn=FakeBN(c, "bb.bn2"), stride=1, pad=1, act=1, dtype=f32, device=cuda)
    xd = nhwc(x).to(cuda)
    with _enc_lib():
        assert ops.basicblock_fusable(xd, p1, p2)
        y = ops.basicblock(xd, p1, p2)
    wr1 = _h(p1.wgt.float().cpu()[:, :p1.K].reshape(c, 3, 3, c).permute(0, 3, 1, 2)).double()
    wr2 = _h(p2.wgt.float().cpu()[:, :p2.K].reshape(c, 3, 3, c).permute(0, 3, 1, 2)).double()
    mid = F.relu(F.conv2d(_h(x).double(), wr1, p1.bias.cpu().double(), padding=1))
    ref = F.relu(F.conv2d(_h(mid.float()).double(), wr2, p2.bias.cpu().double(), padding=1) + x.double())
    # (the intermediate is rounded from the kernel's fp32 accumulation, the reference's from fp64: a value on a rounding boundary may
    #  fall the other way - one fp16 ulp of one input of conv2, far below the gate)
    _close_f16w(y.permute(0, 3, 1, 2), ref, "fused basicblock %d" % c, rel=6e-5)


@pytest.mark.parametrize("n,h,w", [(3, 128, 96), (1, 36, 52)])
def test_fp16_weight_matrix_path_stem(cuda, n, h, w):
    """stem conv 7x7 / 2 + BN + ReLU + max-pool: image pieces split when the patch is staged, weights converted once into LDS; fp32
    image and uint8 frames (the table value is what gets split) give the same bits"""
    f32 = torch.float32
    x8 = (procedural_input("sp8.x", (n, h, w, 3), 0, 0.0, 1.0) * 255).round().clamp(0, 255).to(torch.uint8)
    lut = torch.stack([(torch.arange(256, dtype=torch.float64) / 255 - m) / sd for m, sd in ((0.485, 0.229), (0.456, 0.224), (0.406, 0.225))]).float()
    x = torch.stack([lut[c][x8[..., c].long()] for c in range(3)], -1)                 # (n, h, w, 3) normalised image
    wt = procedural_input("sp.w", (64, 3, 7, 7), 0) * math.sqrt(3.0 / 147)
    plan = ops.ConvPlan(wt, None, bn=FakeBN(64, "sp.bn"), stride=2, pad=3, act=1, dtype=f32, device=cuda, smallc=True)
    with _enc_lib():
        y = ops.stem_pool(x.to(cuda), plan)
        y8 = ops.stem_pool_u8(x8.to(cuda), lut.to(cuda), plan)
    assert torch.equal(y, y8)
    wref = _h(plan.wgt.float().cpu()[:, :plan.K].reshape(64, 7, 7, 3).permute(0, 3, 1, 2)).double()
    ref = F.max_pool2d(F.relu(F.conv2d(x.permute(0, 3, 1, 2).double(), wref, plan.bias.cpu().double(), stride=2, padding=3)), 3, 2, 1)
    _close_f16w(y.permute(0, 3, 1, 2), ref, "stem + pool")


def test_fp16_weight_matrix_path_downsample_1x1(cuda):
    """the 1x1 / stride-2 projection shortcut of a down-sampling BasicBlock (dense-row kernel, weights as the SECOND MFMA operand)"""
    f32 = torch.float32
    x = procedural_input("ds1.x", (2, 128, 18, 26), 0, -2.0, 2.0)
    wt = procedural_input("ds1.w", (256, 128, 1, 1), 0) * math.sqrt(3.0 / 128)
    plan = ops.ConvPlan(wt, None, bn=FakeBN(256, "ds1.bn"), stride=2, pad=0, act=0, dtype=f32, device=cuda)
    with _enc_lib():
        y = ops.conv2d(nhwc(x).to(cuda), plan)
    wref = _h(plan.wgt.float().cpu()[:, :plan.K].reshape(256, 1, 1, 128).permute(0, 3, 1, 2)).double()
    ref = F.conv2d(x.double(), wref, plan.bias.cpu().double(), stride=2)
    _close_f16w(y.permute(0, 3, 1, 2), ref, "1x1 stride-2 shortcut")


@pytest.mark.parametrize("k,rows", [(256, 20480), (512, 5120), (512, 1000), (384, 77), (256, 31)])
def test_projection_chain_key_and_value_single_launch(cuda, k, rows):
    """cobevt_proj_chain_kv (csrc/proj_chain_k.hip): BN -> ReLU -> 1x1 conv K -> 128 (+ ray embedding on the key side) -> LayerNorm ->
    stacked to_k | to_v of both attentions, key and value side in ONE launch (fax_modules.py:281-292,377-396,201-205), against the
    four launches it replaces and against fp32 torch on the same bf16-rounded operands; ragged row counts, every chunk count"""
    import torch.nn as nn
    from cobevt_amd import host
    from cobevt_amd.host import runtime as rt
    g = torch.Generator().manual_seed(7)
    owner = host.runtime.HipModule()
    for tag in ("k", "v"):
        bn, conv, ln, lin = nn.BatchNorm2d(k), nn.Conv2d(k, 128, 1, bias=False), nn.LayerNorm(128), nn.Linear(128, 256, bias=True)
        with torch.no_grad():
            bn.weight.copy_(torch.rand(k, generator=g) + 0.5); bn.bias.copy_(torch.randn(k, generator=g) * 0.2)
            bn.running_mean.copy_(torch.randn(k, generator=g) * 0.3); bn.running_var.copy_(torch.rand(k, generator=g) + 0.5)
            ln.weight.copy_(torch.rand(128, generator=g) + 0.5); ln.bias.copy_(torch.randn(128, generator=g) * 0.2)
            conv.weight.copy_(torch.randn(128, k, 1, 1, generator=g) * (1.0 / k) ** 0.5)
        for nm, mod in (("bn", bn.eval()), ("conv", conv), ("ln", ln), ("lin", lin)):
            setattr(owner, nm + tag, mod)
    owner = owner.to(cuda)
    x = torch.randn(rows, k, generator=g)
    res = torch.randn(rows, 128, generator=g)
    xb, rb = x.to(torch.bfloat16).to(cuda), res.to(torch.bfloat16).to(cuda)
    with host.compute_dtype(torch.bfloat16):
        ppk, ppv = rt.conv_plan(owner, "pk", owner.convk, pre_bn=owner.bnk), rt.conv_plan(owner, "pv", owner.convv, pre_bn=owner.bnv)
        pnk, pnv = rt.linear_plan(owner, "nk", owner.link, ln=owner.lnk), rt.linear_plan(owner, "nv", owner.linv, ln=owner.lnv)
        assert ops.proj_chain_kv_fusable(xb, ppk, ppv, pnk, pnv, rb)
        kk, vv = ops.proj_chain_kv(xb, ppk, ppv, pnk, pnv, residual=rb)
        key = ops.conv2d(xb.reshape(1, 1, rows, k), ppk, residual=rb.reshape(1, 1, rows, 128)).reshape(rows, 128)
        val = ops.conv2d(xb.reshape(1, 1, rows, k), ppv).reshape(rows, 128)
        kk4, vv4 = ops.linear(key, pnk), ops.linear(val, pnv)
    torch.cuda.synchronize()
    with torch.no_grad():
        refs = []
        for tag, r in (("k", rb), ("v", None)):
            bn, conv, ln, lin = (getattr(owner, nm + tag) for nm in ("bn", "conv", "ln", "lin"))
            a = torch.relu(bn(xb.float().t().reshape(1, k, rows, 1))).reshape(k, rows).t().to(torch.bfloat16).float()
            y = a @ conv.weight.reshape(128, k).to(torch.bfloat16).float().t()
            if r is not None:
                y = y + r.float()
            refs.append(lin(ln(y.to(torch.bfloat16).float())).cpu())
    for got, four, ref, what in ((kk, kk4, refs[0], "key side"), (vv, vv4, refs[1], "value side")):
        s = float(ref.abs().max())
        assert tuple(got.shape) == (rows, 256) and torch.isfinite(got.float()).all()
        assert (got.float().cpu() - ref).abs().max().item() <= 1e-2 * s, what
        assert (got.float() - four.float()).abs().max().item() <= 1e-2 * s, what
