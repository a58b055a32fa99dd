"""Host-side weight re-layouts of cobevt_amd.ops.ConvPlan / DepthwisePlan checked against the plain index formulas the kernels
assume (include/cobevt_hip.h).  CPU only: the plans are built with device="cpu"; nothing is launched."""
import numpy as np
import torch

from cobevt_amd import ops


def _w(shape, seed):
    g = torch.Generator().manual_seed(seed)
    # integers: exact in bf16, so the layouts can be compared element for element
    return torch.randint(-8, 9, shape, generator=g).float()


def test_dense_row_weights_and_fragment_order():
    """[Cout][Kp] rows and the MFMA fragment order [N_p/32][Kp/16][64 lanes][8] with lane = 32*half + n%32 holding elements
    [16*kgroup + 8*half, +8) of row n (cobevt_attn_mlp_chain / cobevt_linear_rows_small_k)"""
    for (cout, k) in ((96, 128), (40, 56), (130, 320)):
        w = _w((cout, k), cout + k)
        plan = ops.ConvPlan(w, None, dtype=torch.bfloat16, device="cpu")
        kp = plan.kp_rows
        assert kp == (k + 127) // 128 * 128
        rows = plan.wgt_rows.float()
        assert tuple(rows.shape) == (cout, kp) and torch.equal(rows[:, :k], w) and rows[:, k:].abs().sum() == 0
        frag = plan.wfrag_rows.float()                        # (N_p/32, Kp/16, 2, 32, 8)
        npad = (cout + 127) // 128 * 128
        assert frag.numel() == npad * kp
        frag = frag.reshape(npad // 32, kp // 16, 64, 8)
        ref = torch.zeros(npad, kp)
        ref[:cout, :k] = w
        for tile in range(npad // 32):
            for g in (0, kp // 16 - 1):
                for lane in (0, 17, 32, 63):
                    n, half = tile * 32 + lane % 32, lane // 32
                    assert torch.equal(frag[tile, g, lane], ref[n, 16 * g + 8 * half:16 * g + 8 * half + 8])


def test_layernorm_and_batchnorm_folding():
    """W' = W diag(gamma), b' = b + W beta (LayerNorm in front of a Linear); W' = s W, b' = s b + t (eval BatchNorm after a conv)"""
    w, b = _w((16, 32), 1) / 8, _w((16,), 2) / 8

    class LN(object):
        weight, bias, eps = 1 + _w((32,), 3) / 16, _w((32,), 4) / 16, 1e-5
    plan = ops.ConvPlan(w, b, dtype=torch.float32, device="cpu", ln=LN)
    assert plan.has_ln and plan.ln_eps == 1e-5
    assert torch.allclose(plan.wgt_rows[:, :32], w * LN.weight[None, :], atol=1e-6)
    assert torch.allclose(plan.bias, b + w @ LN.bias, atol=1e-6)
    bn = torch.nn.BatchNorm2d(16).eval()
    bn.weight.data.copy_(1 + _w((16,), 5) / 16); bn.bias.data.copy_(_w((16,), 6) / 16)
    bn.running_mean.data.copy_(_w((16,), 7) / 16); bn.running_var.data.copy_(1 + _w((16,), 8).abs() / 16)
    wc = _w((16, 8, 3, 3), 9) / 8
    plan = ops.ConvPlan(wc, None, bn=bn, stride=1, pad=1, dtype=torch.float32, device="cpu")
    s = bn.weight / torch.sqrt(bn.running_var + bn.eps)
    t = bn.bias - bn.running_mean * s
    k = 9 * 8
    ref = (wc * s[:, None, None, None]).permute(0, 2, 3, 1).reshape(16, k)             # k = (kh*3 + kw)*Cin + c
    assert torch.allclose(plan.wgt[:, :k], ref, atol=1e-6) and torch.allclose(plan.bias, t, atol=1e-6)


def test_conv3x3_fragment_order():
    """[Cout_p/32][Cin/cc][9 taps][KG][64 lanes][8]: lane = 32*half + cout%32 holds channels [chunk*cc + 16*g + 8*half, +8) of
    tap (kh, kw) (cobevt_conv3x3_wfrag_nhwc, cobevt_basicblock_nhwc)"""
    cout, cin = 64, 128
    w = _w((cout, cin, 3, 3), 11)
    plan = ops.ConvPlan(w, None, stride=1, pad=1, dtype=torch.bfloat16, device="cpu")
    assert plan.cc3 == 64 and plan.coutp3 == 128 and plan.wfrag is not None
    frag = plan.wfrag.float().reshape(plan.coutp3 // 32, cin // 64, 9, 4, 64, 8)
    for tile in range(plan.coutp3 // 32):
        for chunk in range(cin // 64):
            for tap in (0, 4, 8):
                for g in (0, 3):
                    for lane in (0, 31, 32, 63):
                        n, half = tile * 32 + lane % 32, lane // 32
                        c0 = chunk * 64 + 16 * g + 8 * half
                        ref = w[n, c0:c0 + 8, tap // 3, tap % 3] if n < cout else torch.zeros(8)
                        assert torch.equal(frag[tile, chunk, tap, g, lane], ref)
    # the stride-2 form keeps the fragments and drops the LDS-staged layout
    p2 = ops.ConvPlan(w, None, stride=2, pad=1, dtype=torch.bfloat16, device="cpu")
    assert p2.wfrag is not None and p2.wgt3 is None and p2.out_hw(64, 64) == (32, 32)


def test_asymmetric_padding_and_depthwise_plan():
    """TensorFlow-"same" padding of the EfficientNet stem (pad 0 before, 1 after) and the depthwise [tap][C] layout"""
    ws = _w((48, 3, 3, 3), 21)
    sp = ops.ConvPlan(ws, None, stride=2, pad=0, pad_br=1, act=3, dtype=torch.bfloat16, device="cpu", smallc=True)
    assert sp.out_hw(224, 480) == (112, 240) and sp.out_hw(23, 30) == (11, 15)
    assert sp.wgt3 is None and sp.wgt_rows is None and sp.wgt_stem is None                     # generic implicit GEMM only
    wd = _w((24, 1, 5, 5), 22)
    dw = ops.DepthwisePlan(wd, bn=None, stride=2, pad=(2, 2), act=3, dtype=torch.bfloat16, device="cpu")
    assert dw.out_hw(95, 95) == (48, 48) and tuple(dw.wgt.shape) == (25, 24)
    assert torch.equal(dw.wgt[7], wd[:, 0, 1, 2]) and dw.bias.abs().sum() == 0


def test_conv3_tiling_picks_whole_workgroups_per_cu():
    """the strip count per workgroup is chosen so the grid is a whole number of workgroups per CU on the frame's shapes"""
    for (n, ho, wo, cin, cout) in ((20, 32, 32, 256, 256), (20, 16, 16, 512, 512)):
        v = ops.conv3_tiling(n, ho, wo, cin, cout, 64)
        assert v >= 100
        mt, bn64 = (v - 100) // 10, (v - 100) % 10
        strips = n * ((ho + 1) // 2) * ((wo + 15) // 16)
        blocks = (strips + mt - 1) // mt * ((cout + (64 if bn64 else 128) - 1) // (64 if bn64 else 128))
        assert blocks % 256 == 0, (v, blocks)
