// Operand preparation of the training convolutions (gfx950): what the autograd path used to assemble from 4-8 torch ops per
// convolution and step (cast, permute, pad, flip, contiguous - ~1200 tiny launches per CorpBEVT step, profiles/r03_train_amp_kernel_trace.txt).
//   conv_weight_rows: the fp32 master weight (Cout, Cin, kh, kw) of an nn.Conv2d / nn.Linear -> the implicit-GEMM kernel's weight rows
//     for the forward pass ([Cout][Kpad], k = (r * kw + s) * Cin + c) AND for the input gradient (the same convolution with the taps
//     flipped and the channel roles swapped: [Cin][Kpad'], k' = (r * kw + s) * Cout + o of w[o][c][kh-1-r][kw-1-s]) in the compute
//     type, in one launch - the rows follow the parameter through optimizer steps, the parameter itself stays fp32 (no autograd cast node).
//   wgrad_block_operand: a channels-last bf16 map (N, H, W, C) -> cobevt_conv_wgrad_blocked's operand [n][row][block of 8 pixels][plane][c][8],
//     zero-padded (top / left by `pad`, bottom / right up to the block counts) - an 8 x 8 transpose per thread, 16-byte accesses both ways.
//     Slot j of plane q of block b holds input pixel sx (8 b + j) + q - pad_left: one plane and sx = 1 for a stride-1 convolution, the
//     even / odd columns (sx = 2) for a stride-2 one, the k tap columns of the 3-channel stem as k planes.
// The reference gets all of this from cuDNN under torch autograd (train_camera.py:143-179).  HBM-bound copies.
#include "common.hpp"

namespace cobevt {
namespace {

constexpr int kThreads = 256;

template <typename T>
__global__ __launch_bounds__(kThreads) void conv_weight_rows_kernel(const float* __restrict__ w, T* __restrict__ fwd, T* __restrict__ dgr,
                                                                    int Cout, int Cin, int kh, int kw, int kpf, int kpd) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    const long nf = fwd ? (long)Cout * kpf : 0, nd = dgr ? (long)Cin * kpd : 0;
    if (i < nf) {
        const int o = (int)(i / kpf), k = (int)(i - (long)o * kpf);
        float v = 0.f;
        if (k < kh * kw * Cin) {
            const int tap = k / Cin, c = k - tap * Cin;
            const int r = tap / kw, s = tap - r * kw;
            v = w[(((long)o * Cin + c) * kh + r) * kw + s];
        }
        store_elem<T>(fwd, (size_t)i, v);
    } else if (i - nf < nd) {
        const long j = i - nf;
        const int c = (int)(j / kpd), k = (int)(j - (long)c * kpd);
        float v = 0.f;
        if (k < kh * kw * Cout) {
            const int tap = k / Cout, o = k - tap * Cout;
            const int r = tap / kw, s = tap - r * kw;
            v = w[(((long)o * Cin + c) * kh + (kh - 1 - r)) * kw + (kw - 1 - s)];
        }
        store_elem<T>(dgr, (size_t)j, v);
    }
}

// ---- 3x3 convolutions on the inference kernels (conv3x3.hip): the master weight -> the two bf16 layouts those kernels read, for the
// forward pass (O = Cout, I = Cin, value w[O][I][tap]) or for the input gradient (the same convolution with the channel roles swapped and
// the taps flipped: O = Cin, I = Cout, value w[I][O][8 - tap]).
//   frag:  [Op/32][I/64][9 taps][4 k-groups][2 halves][32][8]  (ops.ConvPlan.wfrag: MFMA B fragments, Op = O rounded up to 128, zero rows)
//   rows3: [O][I/64][9 taps][64]                               (ops.ConvPlan.wgt3: the LDS-staged kernel's rows)
// One thread per 16-byte piece; pieces [0, nfrag) then [nfrag, nfrag + nrows).
struct Conv3OperandJob {
    uint4* frag;      // nullable
    uint4* rows3;     // nullable
    long nfrag, nrows;
    int O, I, dgrad;
};

__device__ __forceinline__ void conv3_operand_piece(const float* __restrict__ w, const Conv3OperandJob& j, int Cin_w, long i) {
    int o, i0, tap;
    uint4* dst;
    if (i < j.nfrag) {
        dst = j.frag + i;
        const int n = (int)(i & 31), half = (int)((i >> 5) & 1), kg = (int)((i >> 6) & 3);
        long t = i >> 8;
        tap = (int)(t % 9);
        t /= 9;
        const int chunks = j.I >> 6;
        const int chunk = (int)(t % chunks), ct = (int)(t / chunks);
        o = ct * 32 + n;
        i0 = chunk * 64 + kg * 16 + half * 8;
    } else {
        i -= j.nfrag;
        dst = j.rows3 + i;
        const int jj = (int)(i & 7);
        long t = i >> 3;
        tap = (int)(t % 9);
        t /= 9;
        const int chunks = j.I >> 6;
        const int chunk = (int)(t % chunks);
        o = (int)(t / chunks);
        i0 = chunk * 64 + jj * 8;
    }
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        // w is (Cout, Cin_w, 3, 3): forward w[o][i0 + e][tap], input gradient w[i0 + e][o][8 - tap]
        const long idx = j.dgrad ? (((long)(i0 + e) * Cin_w + o) * 9 + (8 - tap)) : (((long)o * Cin_w + (i0 + e)) * 9 + tap);
        v[e] = o < j.O ? w[idx] : 0.f;
    }
    *dst = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

// pieces of job a, then of job b (the forward convolution's operand and its input gradient's: one launch per convolution and step)
__global__ __launch_bounds__(kThreads) void conv3_weight_operands_kernel(const float* __restrict__ w, Conv3OperandJob a, Conv3OperandJob b, int Cin_w) {
    long i = (long)blockIdx.x * kThreads + threadIdx.x;
    const long na = a.nfrag + a.nrows, nb = b.nfrag + b.nrows;
    if (i < na) conv3_operand_piece(w, a, Cin_w, i);
    else if (i - na < nb) conv3_operand_piece(w, b, Cin_w, i - na);
}

// ---- dense projections on the inference row-GEMM kernel (gemm_rows3.hip): the fp32 master weight (N, K) -> its fragment table
// [Np/32][Kp/16][2 halves][32][8] bf16 (ops.ConvPlan.wfrag_rows: Kp = K rounded up to 128, Np = N rounded up to 128, zero padding), for the
// projection itself (rows = output features) or, transposed, for its input gradient dx = dy W (rows = input features, contraction over N).
__global__ __launch_bounds__(kThreads) void linear_weight_frags_kernel(const float* __restrict__ w, uint4* __restrict__ frag_f, uint4* __restrict__ frag_t,
                                                                       int N, int K, long pieces_f, long pieces_t) {
    long i = (long)blockIdx.x * kThreads + threadIdx.x;
    // pieces [0, pieces_f): the projection's table (rows N, contraction K, value w[r][c]); then the transposed one (rows K, contraction N, w[c][r])
    const bool transpose = i >= pieces_f;
    if (transpose) i -= pieces_f;
    if (transpose ? i >= pieces_t : i >= pieces_f) return;
    const int R = transpose ? K : N, C = transpose ? N : K;
    const int Cp = (C + 127) / 128 * 128;
    const int n = (int)(i & 31), half = (int)((i >> 5) & 1);
    long t = i >> 6;
    const int kgs = Cp >> 4;
    const int kg = (int)(t % kgs), tile = (int)(t / kgs);
    const int r = tile * 32 + n, c0 = kg * 16 + half * 8;
    float v[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = c0 + e;
        v[e] = (r < R && c < C) ? (transpose ? w[(long)c * K + r] : w[(long)r * K + c]) : 0.f;
    }
    (transpose ? frag_t : frag_f)[i] = make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}

// one thread = (n, padded row, block, 8-channel group): 8 pixels x 8 channels in, 8 channels x 8 pixels out
__global__ __launch_bounds__(kThreads) void wgrad_block8_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, long items,
                                                                int H, int W, int C, int Hp, int NB, int pt, int pl, int P, int sx) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int G = C >> 3;
    const int g = (int)(gid % G);
    long t = gid / G;
    const int q = (int)(t % P);
    t /= P;
    const int b = (int)(t % NB);
    t /= NB;
    const int y = (int)(t % Hp);
    const long n = t / Hp;
    const int sy = y - pt;
    uint4 px[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cx = (b * 8 + j) * sx + q - pl;
        const bool ok = sy >= 0 && sy < H && cx >= 0 && cx < W;
        // clamped address + select: the load stays unconditional (a branch around it serialises the eight round trips)
        const long off = (((n * H + (ok ? sy : 0)) * W + (ok ? cx : 0)) * C) + g * 8;
        const uint4 v = *(const uint4*)(src + off);
        px[j] = ok ? v : make_uint4(0, 0, 0, 0);
    }
    // 8 x 8 transpose of 16-bit elements: out row e (channel g*8 + e) = element e of every pixel
    const uint32_t* pw = (const uint32_t*)px;                  // pw[j * 4 + m] = channels 2m, 2m + 1 of pixel j
    uint16_t* d = dst + ((((n * Hp + y) * NB + b) * P + q) * (long)C + g * 8) * 8;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        uint32_t o[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const uint32_t lo = pw[(2 * m) * 4 + (e >> 1)], hi = pw[(2 * m + 1) * 4 + (e >> 1)];
            o[m] = (e & 1) ? ((lo >> 16) | (hi & 0xffff0000u)) : ((lo & 0xffffu) | (hi << 16));
        }
        *(uint4*)(d + e * 8) = make_uint4(o[0], o[1], o[2], o[3]);
    }
}

// any channel count: one thread per (n, padded row, block, channel)
__global__ __launch_bounds__(kThreads) void wgrad_block1_kernel(const uint16_t* __restrict__ src, uint16_t* __restrict__ dst, long items,
                                                                int H, int W, int C, int Hp, int NB, int pt, int pl, int P, int sx) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int c = (int)(gid % C);
    long t = gid / C;
    const int q = (int)(t % P);
    t /= P;
    const int b = (int)(t % NB);
    t /= NB;
    const int y = (int)(t % Hp);
    const long n = t / Hp;
    const int sy = y - pt;
    uint16_t v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const int cx = (b * 8 + j) * sx + q - pl;
        const bool ok = sy >= 0 && sy < H && cx >= 0 && cx < W;
        const uint16_t x = src[(((n * H + (ok ? sy : 0)) * W + (ok ? cx : 0)) * C) + c];
        v[j] = ok ? x : (uint16_t)0;
    }
    uint4 o;
    o.x = v[0] | ((uint32_t)v[1] << 16); o.y = v[2] | ((uint32_t)v[3] << 16);
    o.z = v[4] | ((uint32_t)v[5] << 16); o.w = v[6] | ((uint32_t)v[7] << 16);
    *(uint4*)(dst + gid * 8) = o;
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_conv_weight_rows(const float* w, void* rows_fwd, void* rows_dgrad, const int* dims, hipStream_t stream) {
    // dims: [dtype, Cout, Cin, kh, kw, Kpad_fwd, Kpad_dgrad]
    if (!w || !dims || (!rows_fwd && !rows_dgrad)) return COBEVT_ERR_ARG;
    const int dtype = dims[0], Cout = dims[1], Cin = dims[2], kh = dims[3], kw = dims[4], kpf = dims[5], kpd = dims[6];
    if (Cout < 1 || Cin < 1 || kh < 1 || kw < 1) return COBEVT_ERR_SHAPE;
    if ((rows_fwd && kpf < kh * kw * Cin) || (rows_dgrad && kpd < kh * kw * Cout)) return COBEVT_ERR_SHAPE;
    const long total = (rows_fwd ? (long)Cout * kpf : 0) + (rows_dgrad ? (long)Cin * kpd : 0);
    if (total > 0x7fffffffL * (long)kThreads) return COBEVT_ERR_SHAPE;
    const dim3 grid((unsigned)((total + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(conv_weight_rows_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, w, (bf16_t*)rows_fwd, (bf16_t*)rows_dgrad, Cout, Cin, kh, kw, kpf, kpd);
    else if (dtype == 1) hipLaunchKernelGGL(conv_weight_rows_kernel<float>, grid, dim3(kThreads), 0, stream, w, (float*)rows_fwd, (float*)rows_dgrad, Cout, Cin, kh, kw, kpf, kpd);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

static bool conv3_job(Conv3OperandJob& j, void* frag, void* rows3, int Cout, int Cin, int dgrad) {
    j.frag = (uint4*)frag; j.rows3 = (uint4*)rows3; j.dgrad = dgrad;
    j.O = dgrad ? Cin : Cout; j.I = dgrad ? Cout : Cin;
    j.nfrag = 0; j.nrows = 0;
    if (!frag && !rows3) return true;
    if (j.I % 64 != 0) return false;
    const int Op = (j.O + 127) / 128 * 128;
    j.nfrag = frag ? (long)Op * j.I * 9 / 8 : 0;
    j.nrows = rows3 ? (long)j.O * j.I * 9 / 8 : 0;
    return true;
}

extern "C" int cobevt_conv3_weight_operands(const float* w, void* frag, void* rows3, const int* dims, hipStream_t stream) {
    // dims: [Cout, Cin, dgrad]; bf16 outputs (either nullable).  Forward: O = Cout, I = Cin; input gradient: O = Cin, I = Cout.
    if (!w || !dims || (!frag && !rows3)) return COBEVT_ERR_ARG;
    const int Cout = dims[0], Cin = dims[1], dgrad = dims[2];
    if (Cout < 1 || Cin < 1 || (dgrad != 0 && dgrad != 1)) return COBEVT_ERR_SHAPE;
    Conv3OperandJob a, b;
    if (!conv3_job(a, frag, rows3, Cout, Cin, dgrad) || !conv3_job(b, nullptr, nullptr, Cout, Cin, 0)) return COBEVT_ERR_SHAPE;
    const long total = a.nfrag + a.nrows;
    if (total > 0x7fffffffL * (long)kThreads) return COBEVT_ERR_SHAPE;
    hipLaunchKernelGGL(conv3_weight_operands_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, w, a, b, Cin);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// both directions in one launch: outs = [fwd frag, fwd rows3, dgrad frag, dgrad rows3], each nullable
extern "C" int cobevt_conv3_weight_operands2(const float* w, void* const* outs, const int* dims, hipStream_t stream) {
    // dims: [Cout, Cin]
    if (!w || !dims || !outs || (!outs[0] && !outs[1] && !outs[2] && !outs[3])) return COBEVT_ERR_ARG;
    const int Cout = dims[0], Cin = dims[1];
    if (Cout < 1 || Cin < 1) return COBEVT_ERR_SHAPE;
    Conv3OperandJob a, b;
    if (!conv3_job(a, outs[0], outs[1], Cout, Cin, 0) || !conv3_job(b, outs[2], outs[3], Cout, Cin, 1)) return COBEVT_ERR_SHAPE;
    const long total = a.nfrag + a.nrows + b.nfrag + b.nrows;
    if (total > 0x7fffffffL * (long)kThreads) return COBEVT_ERR_SHAPE;
    hipLaunchKernelGGL(conv3_weight_operands_kernel, dim3((unsigned)((total + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, w, a, b, Cin);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_linear_weight_frags(const float* w, void* frag, void* frag_t, const int* dims, hipStream_t stream) {
    // dims: [N, K]; frag (nullable): the projection's table, (Np / 32) x (Kp / 16) x 64 x 16 bytes; frag_t (nullable): the transposed one,
    // (Kp / 32) x (Np / 16) x 64 x 16 bytes; Np, Kp = N, K rounded up to 128.  One launch for both.
    if (!w || (!frag && !frag_t) || !dims) return COBEVT_ERR_ARG;
    const int N = dims[0], K = dims[1];
    if (N < 1 || K < 1) return COBEVT_ERR_SHAPE;
    const long Np = (N + 127) / 128 * 128, Kp = (K + 127) / 128 * 128;
    const long pf = frag ? Np * Kp / 8 : 0, pt = frag_t ? Np * Kp / 8 : 0;
    hipLaunchKernelGGL(linear_weight_frags_kernel, dim3((unsigned)((pf + pt + kThreads - 1) / kThreads)), dim3(kThreads), 0, stream, w, (uint4*)frag,
                       (uint4*)frag_t, N, K, pf, pt);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_wgrad_block_operand(const void* src, void* dst, const int* dims, hipStream_t stream) {
    // dims: [N, H, W, C, Hp, NB, pad_top, pad_left, P, sx]; src bf16 (N, H, W, C) -> dst bf16 [N][Hp][NB][P][C][8]
    if (!src || !dst || !dims) return COBEVT_ERR_ARG;
    const int N = dims[0], H = dims[1], W = dims[2], C = dims[3], Hp = dims[4], NB = dims[5], pt = dims[6], pl = dims[7], P = dims[8], sx = dims[9];
    if (N < 1 || H < 1 || W < 1 || C < 1 || Hp < 1 || NB < 1 || pt < 0 || pl < 0 || P < 1 || sx < 1) return COBEVT_ERR_SHAPE;
    const bool g8 = C % 8 == 0 && ((size_t)src % 16) == 0;
    const long items = (long)N * Hp * NB * P * (g8 ? C >> 3 : C);
    if (items > 0x7fffffffL * (long)kThreads) return COBEVT_ERR_SHAPE;
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (g8) hipLaunchKernelGGL(wgrad_block8_kernel, grid, dim3(kThreads), 0, stream, (const uint16_t*)src, (uint16_t*)dst, items, H, W, C, Hp, NB, pt, pl, P, sx);
    else hipLaunchKernelGGL(wgrad_block1_kernel, grid, dim3(kThreads), 0, stream, (const uint16_t*)src, (uint16_t*)dst, items, H, W, C, Hp, NB, pt, pl, P, sx);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
