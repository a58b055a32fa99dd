"""Host-side weight re-layouts of cobevt_amd.ops.ConvPlan / DepthwisePlan checked against the plain index formulas the kernels
assume (include/cobevt_hip.h).  CPU only: the plans are built with device="cpu"; nothing is launched."""
import numpy as np
import torch

from cobevt_amd import ops
from cobevt_amd.synth import procedural_input


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


def test_bottleneck_fragment_tables():
    """ops.BottleneckPlan: the three fragment tables reproduce the folded convolutions when contracted the way bottleneck.hip does
    (lane = 32 * half + output row; W3's contraction slots in the accumulator-register order of conv2's result)."""
    import torch.nn as nn
    import torch.nn.functional as F
    from cobevt_amd.synth import fill_module_

    class B(nn.Module):
        def __init__(self):
            super().__init__()
            self.conv1, self.bn1 = nn.Conv2d(128, 32, 1, bias=False), nn.BatchNorm2d(32)
            self.conv2, self.bn2 = nn.Conv2d(32, 32, 3, 1, 1, bias=False), nn.BatchNorm2d(32)
            self.conv3, self.bn3 = nn.Conv2d(32, 128, 1, bias=False), nn.BatchNorm2d(128)
    m = fill_module_(B(), 3).eval()
    plan = ops.BottleneckPlan(m.conv1, m.bn1, m.conv2, m.bn2, m.conv3, m.bn3, device="cpu")
    f1, f2, f3 = plan.w1.float(), plan.w2.float(), plan.w3.float()
    assert tuple(f1.shape) == (8, 64, 8) and tuple(f2.shape) == (9, 2, 64, 8) and tuple(f3.shape) == (4, 2, 64, 8)
    x = procedural_input("bn.x", (1, 128, 6, 7), 0)
    xr = x.to(torch.bfloat16).float()
    lane = torch.arange(64)
    row, half = lane % 32, lane // 32
    j = torch.arange(8)
    # conv1: acc[m][pix] = sum_g sum_lane(row == m) sum_j f1[g][lane][j] * x[pix][16 g + 8 half + j]
    xp = xr[0].permute(1, 2, 0).reshape(-1, 128)                                   # [pix][c]
    y1 = torch.zeros(32, xp.shape[0])
    for g in range(8):
        ch = 16 * g + 8 * half[:, None] + j[None, :]                               # [lane][j]
        contrib = torch.einsum("lj,plj->lp", f1[g], xp[:, ch])                     # [lane][pix]
        y1.index_add_(0, row, contrib)
    s1, sh1 = ops.bn_affine(m.bn1)
    ref1 = F.conv2d(xr, (m.conv1.weight.double() * s1[:, None, None, None]).float().to(torch.bfloat16).float())
    assert torch.allclose(y1, ref1[0].reshape(32, -1), atol=1e-4)
    # conv3 with the permuted contraction index: y2 registers of lane (pix, half): r = 8 u + j <-> channel 16u + (j&3) + 8(j>>2) + 4 half
    y2 = procedural_input("bn.y2", (5, 32), 0).to(torch.bfloat16).float()          # [pix][mid]
    out = torch.zeros(128, 5)
    for ct in range(4):
        for u in range(2):
            ch = 16 * u + (j & 3)[None, :] + 8 * (j >> 2)[None, :] + 4 * half[:, None]
            contrib = torch.einsum("lj,plj->lp", f3[ct, u], y2[:, ch])
            out.index_add_(0, 32 * ct + row, contrib)
    s3, _ = ops.bn_affine(m.bn3)
    w3 = (m.conv3.weight.double()[:, :, 0, 0] * s3[:, None]).float().to(torch.bfloat16).float()
    assert torch.allclose(out, w3 @ y2.t(), atol=1e-4)
    # conv2: tap-major fragments, standard slot order
    y1p = procedural_input("bn.y1", (32, 5, 6), 0).to(torch.bfloat16).float()      # [c][h][w]
    s2, _ = ops.bn_affine(m.bn2)
    w2 = (m.conv2.weight.double() * s2[:, None, None, None]).float().to(torch.bfloat16).float()
    ref2 = F.conv2d(y1p[None], w2, padding=1)[0]                                    # [32][5][6]
    pad = F.pad(y1p, (1, 1, 1, 1))
    acc = torch.zeros(32, 5, 6)
    for tap in range(9):
        dy, dx = tap // 3, tap % 3
        win = pad[:, dy:dy + 5, dx:dx + 6].reshape(32, -1).t()                      # [pix][c]
        for u in range(2):
            ch = 16 * u + 8 * half[:, None] + j[None, :]
            contrib = torch.einsum("lj,plj->lp", f2[tap, u], win[:, ch])
            acc.reshape(32, -1).index_add_(0, row, contrib)
    assert torch.allclose(acc, ref2, atol=1e-4)


def test_flipped_domain_taps_identity():
    """host/v2v_fuse.py runs the 3x3 convolutions that the reference applies to the TRANSPOSED + FLIPPED maps
    (v2v_fuse.py:89-93, 'b c h w -> b c w h' + flip) on the original orientation with re-indexed taps: the identity
    conv(flipT(x), W) == flipT(conv(x, W~)), W~[u][v] = W[v][2 - u], checked with torch on the CPU."""
    import torch.nn.functional as F
    from cobevt_amd.host.v2v_fuse import _flipped_domain_taps
    g = torch.Generator().manual_seed(11)
    x = torch.randn(2, 5, 7, 7, generator=g)
    w = torch.randn(4, 5, 3, 3, generator=g)
    flip_t = lambda t: t.permute(0, 1, 3, 2).flip(3)
    ref = F.conv2d(flip_t(x), w, padding=1)
    got = flip_t(F.conv2d(x, _flipped_domain_taps(w), padding=1))
    assert torch.allclose(got, ref, atol=1e-5)
    # and back: un-flipping the reference's result gives the plain-orientation convolution
    assert torch.allclose(ref.flip(3).permute(0, 1, 3, 2), F.conv2d(x, _flipped_domain_taps(w), padding=1), atol=1e-5)


def test_blocked_weight_gradient_operands():
    """autograd._blocked_operands: the operand layout cobevt_conv_wgrad_blocked reads ([n][row][block of 8 pixels][channel][8], x
    zero-padded by `pad`, dy rows padded to an even block count) - checked on the CPU by evaluating the weight gradient from the
    blocked tensors exactly as the kernel indexes them (tap (a, b): x pixel ox + b of padded row oy + a) against torch's conv2d
    autograd, for 3x3 / pad 1 and 1x1 / pad 0 and a width that is not a multiple of 8"""
    import torch.nn.functional as F
    from cobevt_amd import autograd as ag
    g = torch.Generator().manual_seed(5)
    for k, pad, (n, cin, cout, h, w) in ((3, 1, (2, 5, 4, 6, 11)), (1, 0, (1, 3, 6, 4, 16))):
        x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
        wt = torch.randn(cout, cin, k, k, generator=g, requires_grad=True)
        with torch.enable_grad():                       # (other modules of the suite switch gradients off globally)
            y = F.conv2d(x, wt, padding=pad)
            dy = torch.randn(y.shape, generator=g)
            (y * dy).sum().backward()
        xl, dyl = x.detach().permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
        xb, db, hp, nxb, ndb = ag._blocked_operands(xl, dyl, k, pad)
        ho, wo = dyl.shape[1:3]
        assert xb.shape == (n, hp, nxb, cin, 8) and db.shape == (n, ho, ndb, cout, 8)
        assert ndb % 2 == 0 and nxb >= ndb + (k > 1) and hp >= ho + k - 1
        # un-block: pixel p of a row lives in block p // 8, slot p % 8
        xrow = xb.permute(0, 1, 3, 2, 4).reshape(n, hp, cin, nxb * 8)          # [n][padded row][c][padded pixel]
        drow = db.permute(0, 1, 3, 2, 4).reshape(n, ho, cout, ndb * 8)         # [n][oy][o][pixel]
        assert torch.equal(drow[..., :wo], dyl.permute(0, 1, 3, 2)) and not drow[..., wo:].any()
        dw = torch.zeros(cout, cin, k, k)
        npx = ndb * 8                                                         # the kernel walks every pixel of the padded dy rows
        for a in range(k):
            for b in range(k):
                win = xrow[:, a:a + ho, :, b:b + npx]                           # x pixel ox + b of padded row oy + a
                dw[:, :, a, b] = torch.einsum("nyop,nycp->oc", drow, win)
        assert torch.allclose(dw, wt.grad, atol=1e-4), (k, (dw - wt.grad).abs().max())


def test_strided_blocked_weight_gradient_operands():
    """The stride-2 and stem forms of cobevt_conv_wgrad_blocked (modes 1 / 2): the operand layout autograd._blocked_x_general specifies
    ([n][row][block][plane][channel][8]; slot j of plane q of block b = padded input pixel sx (8 b + j) + q - pad), evaluated on the CPU
    exactly as the kernels index it, against torch's conv2d autograd:
      mode 1, 3x3 / stride 2 / pad 1: tap column 0 = plane 0 pixel ox, 1 = plane 1 pixel ox, 2 = plane 0 pixel ox + 1; tap row a = padded row 2 oy + a
      mode 1, 1x1 / stride 2 / pad 0: one plane (the even columns), row 2 oy
      mode 2, 7x7 / stride 2 / pad 3 on 3 channels: tap column b = plane b, pixel ox; tap row a = padded row 2 oy + a"""
    import torch.nn.functional as F
    from cobevt_amd import autograd as ag
    g = torch.Generator().manual_seed(9)
    for k, stride, pad, (n, cin, cout, h, w) in ((3, 2, 1, (2, 5, 4, 9, 13)), (1, 2, 0, (1, 6, 3, 8, 15)), (7, 2, 3, (2, 3, 5, 12, 18)),
                                                 (3, 2, 1, (1, 8, 8, 16, 32))):
        mode = ag.wgrad_blocked_mode(k, stride, pad, cin)
        assert mode == (2 if k == 7 else 1)
        x = torch.randn(n, cin, h, w, generator=g, requires_grad=True)
        wt = torch.randn(cout, cin, k, k, generator=g, requires_grad=True)
        with torch.enable_grad():
            y = F.conv2d(x, wt, stride=stride, padding=pad)
            dy = torch.randn(y.shape, generator=g)
            (y * dy).sum().backward()
        xl, dyl = x.detach().permute(0, 2, 3, 1).contiguous(), dy.permute(0, 2, 3, 1).contiguous()
        ho, wo = dyl.shape[1:3]
        ndb, nxb, hp, planes, sx = ag._blocked_geometry(h, w, ho, wo, k, pad, stride, mode)
        assert ndb % 2 == 0 and hp >= (ho - 1) * stride + k and nxb >= ndb + (1 if (mode == 1 and k == 3) else 0)
        xb = ag._blocked_x_general(xl, hp, nxb, pad, planes, sx)             # [n][hp][nxb][planes][cin][8]
        db = ag._blocked_x_general(dyl, ho, ndb, 0, 1, 1)                    # [n][ho][ndb][1][cout][8]
        assert xb.shape == (n, hp, nxb, planes, cin, 8) and db.shape == (n, ho, ndb, 1, cout, 8)
        xrow = xb.permute(0, 1, 3, 4, 2, 5).reshape(n, hp, planes, cin, nxb * 8)      # [n][padded row][plane][c][pixel slot]
        drow = db[:, :, :, 0].permute(0, 1, 3, 2, 4).reshape(n, ho, cout, ndb * 8)     # [n][oy][o][pixel]
        assert not drow[..., wo:].any()
        npx = ndb * 8
        rows = torch.arange(ho) * stride
        dw = torch.zeros(cout, cin, k, k)
        for a in range(k):
            xr = xrow[:, rows + a]                                                      # padded row oy * stride + a
            for b in range(k):
                if mode == 2:
                    win = xr[:, :, b, :, :npx]
                elif k == 1:
                    win = xr[:, :, 0, :, :npx]
                else:
                    win = xr[:, :, b & 1, :, (b >> 1):(b >> 1) + npx]
                dw[:, :, a, b] = torch.einsum("nyop,nycp->oc", drow, win)
        assert torch.allclose(dw, wt.grad, atol=2e-4), (k, stride, (dw - wt.grad).abs().max())
