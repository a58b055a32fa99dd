// Implicit-GEMM convolution / linear layer on MFMA (gfx950).
//
// One kernel family covers every dense contraction on the CoBEVT hot path:
//   * ResNet-34 encoder convs (7x7 s2, 3x3 s1/s2, 1x1 s2)            reference: opv2v/opencood/models/backbones/resnet_ms.py:67-74
//   * FAX 1x1 feature projections with pre-activation BN+ReLU        reference: .../sub_modules/fax_modules.py:281-292,379-384
//   * every nn.Linear on token-major activations (a 1x1 "conv")      reference: fax_modules.py:189-193,311-312; swap_fusion_modules.py:45-53
//   * Bottleneck / downsample convs incl. PixelUnshuffle(2)          reference: fax_modules.py:472-489
//   * NaiveDecoder 3x3 convs with the nearest x2 up-sampling folded into the gather, BevSegHead
//                                                                    reference: naive_decoder.py:78-87, bev_seg_head.py:35-61
//
// Data layout: activations NHWC (token-major, channels contiguous); weights [Cout][Kpad] with
// k = (kh*Kw + kw)*Cin + c (K contiguous), zero padded to a multiple of one K-tile.  BatchNorm (eval) is
// folded into weights/bias on the host; the epilogue applies bias, residual add, ReLU / exact GELU.
//
// GEMM view: M = N*Ho*Wo output pixels, N = Cout, K = Kh*Kw*Cin.  Block tile BM x BN, K-tile = 64 bytes
// per row (32 bf16 / 16 fp32), 256 threads = 4 waves, each wave owns a WM x WN sub-tile made of 32x32
// MFMA tiles.  Global -> register -> LDS staging with two LDS buffers (one barrier per K-tile): the
// gather for tile t+1 is issued before the MFMAs of tile t.  LDS rows are padded to 80 bytes so the
// 16-lane ds_read_b128 groups hit distinct banks.
#include "common.hpp"

namespace cobevt {

struct IgemmParams {
    const void* in;
    const void* wgt;
    const float* bias;
    const void* residual;
    const float* pre_scale;
    const float* pre_shift;
    const int* klut;
    void* out;
    int N, H, W, Cin;
    int Ho, Wo, Cout;
    int Kh, Kw, stride, pad;
    int K, Kpad;
    int upsample;
    int pre_relu;
    int act;
    int store_mode;
    int out_H, out_W;
    int M;
};

constexpr int kRowBytes = 80;  // 64 data + 16 pad
constexpr int kTileBytes = 64;

// Slow-path stores: padded output maps, fp32 outputs, PixelUnshuffle(2), NCHW logits.  Takes scalars, not the
// params struct: passing IgemmParams by reference forced the whole struct into scratch memory and turned every
// p.field access of the main loop into a memory round trip (236 VMEM reads per wave instead of 12, rocprofv3 PMC).
template <typename T>
__device__ __noinline__ void store_generic(void* out, int store_mode, int Ho, int Wo, int out_H, int out_W, int Cout,
                                           int m, int col, float v) {
    const int hw = Ho * Wo;
    const int n = m / hw, rem = m - n * hw;
    const int oh = rem / Wo, ow = rem - oh * Wo;
    if (store_mode == 0) {
        store_elem<T>((T*)out, (((size_t)n * out_H + oh) * out_W + ow) * Cout + col, v);
    } else if (store_mode == 3) {
        ((float*)out)[(((size_t)n * out_H + oh) * out_W + ow) * Cout + col] = v;
    } else if (store_mode == 1) {  // PixelUnshuffle(2): channel = c*4 + (oh&1)*2 + (ow&1)
        const size_t o = (((size_t)n * (Ho >> 1) + (oh >> 1)) * (Wo >> 1) + (ow >> 1)) * (size_t)(Cout * 4) +
                         col * 4 + (oh & 1) * 2 + (ow & 1);
        store_elem<T>((T*)out, o, v);
    } else {  // 2: NCHW fp32
        ((float*)out)[(((size_t)n * Cout + col) * Ho + oh) * Wo + ow] = v;
    }
}

template <typename T, int BM, int BN, int WM, int WN, bool SMALLC>
__global__ __launch_bounds__(256) void igemm_kernel(IgemmParams p) {
    constexpr int BKE = kTileBytes / Elem<T>::kBytes;
    constexpr int CH = Elem<T>::kChunk;
    constexpr int A_IT = (BM * 4) / 256;
    constexpr int B_IT = (BN * 4 + 255) / 256;
    constexpr int TM = WM / 32, TN = WN / 32;
    constexpr int WAVES_N = BN / WN;
    static_assert((BM / WM) * (BN / WN) == 4, "4 waves per block");
    static_assert(A_IT >= 1, "BM >= 64");

    __shared__ __attribute__((aligned(16))) unsigned char smem[2][(BM + BN) * kRowBytes];

    // XCD-aware bijective remap of the 1-D grid: consecutive logical tiles (same M-tile, all N-tiles,
    // then the next M-tile) land on the same XCD so the A panel is re-read from that XCD's L2.
    const int nblk = gridDim.x;
    const int ntn = (p.Cout + BN - 1) / BN;
    int logical;
    {
        const int bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (logical / ntn) * BM;
    const int n0 = (logical % ntn) * BN;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int j = tid & 3;         // 16-byte chunk within the K-tile row
    const int rowp = tid >> 2;     // 0..63

    // ---- per-thread A row state
    int a_n[A_IT], a_ih0[A_IT], a_iw0[A_IT];
    bool a_ok[A_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
        const int m = m0 + rowp + 64 * it;
        a_ok[it] = m < p.M;
        const int mm = a_ok[it] ? m : 0;
        const int hw = p.Ho * p.Wo;
        const int n = mm / hw, rem = mm - n * hw;
        const int oh = rem / p.Wo, ow = rem - oh * p.Wo;
        a_n[it] = n;
        a_ih0[it] = oh * p.stride - p.pad;
        a_iw0[it] = ow * p.stride - p.pad;
    }
    const int Hv = p.upsample ? 2 * p.H : p.H;
    const int Wv = p.upsample ? 2 * p.W : p.W;

    // tap state of this thread's chunk column (shared by all of its rows)
    int kc, kh, kw;
    {
        const int k0 = j * CH;
        const int tap = k0 / p.Cin;
        kc = k0 - tap * p.Cin;
        kh = tap / p.Kw;
        kw = tap - kh * p.Kw;
    }

    const int nk = p.Kpad / BKE;
    uint4 a_reg[A_IT], b_reg[B_IT];

    auto load_tile = [&](int kt) {
        if constexpr (SMALLC) {
            const float* in = (const float*)p.in;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                float v[8];
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int code = p.klut[kt * BKE + j * CH + e];
                    float x = 0.f;
                    if (code >= 0 && a_ok[it]) {
                        const int ckh = code >> 20, ckw = (code >> 10) & 1023, cc = code & 1023;
                        const int ih = a_ih0[it] + ckh, iw = a_iw0[it] + ckw;
                        if (ih >= 0 && ih < p.H && iw >= 0 && iw < p.W)
                            x = in[(((size_t)a_n[it] * p.H + ih) * p.W + iw) * p.Cin + cc];
                    }
                    v[e] = x;
                }
                a_reg[it] = f32_to_chunk<T>(v);
            }
        } else {
            const T* in = (const T*)p.in;
#pragma unroll
            for (int it = 0; it < A_IT; ++it) {
                const int ih = a_ih0[it] + kh, iw = a_iw0[it] + kw;
                const bool ok = a_ok[it] && kh < p.Kh && ih >= 0 && ih < Hv && iw >= 0 && iw < Wv;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (ok) {
                    const int sh = p.upsample ? (ih >> 1) : ih, sw = p.upsample ? (iw >> 1) : iw;
                    v = *(const uint4*)(in + ((((size_t)a_n[it] * p.H + sh) * p.W + sw) * p.Cin + kc));
                    if (p.pre_scale) {
                        float f[8];
                        chunk_to_f32<T>(v, f);
#pragma unroll
                        for (int e = 0; e < CH; ++e) {
                            f[e] = f[e] * p.pre_scale[kc + e] + p.pre_shift[kc + e];
                            if (p.pre_relu) f[e] = fmaxf(f[e], 0.f);
                        }
                        v = f32_to_chunk<T>(f);
                    }
                }
                a_reg[it] = v;
            }
        }
        const T* wg = (const T*)p.wgt;
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int br = rowp + 64 * it;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (br < BN && n0 + br < p.Cout)
                v = *(const uint4*)(wg + ((size_t)(n0 + br) * p.Kpad + kt * BKE + j * CH));
            b_reg[it] = v;
        }
        // advance the tap to the next K-tile
        kc += BKE;
        while (kc >= p.Cin) {
            kc -= p.Cin;
            if (++kw == p.Kw) { kw = 0; ++kh; }
        }
    };
    auto store_tile = [&](int buf) {
        unsigned char* As = smem[buf];
        unsigned char* Bs = smem[buf] + BM * kRowBytes;
#pragma unroll
        for (int it = 0; it < A_IT; ++it)
            *(uint4*)(As + (rowp + 64 * it) * kRowBytes + j * 16) = stage_x_piece<T>(a_reg[it]);
#pragma unroll
        for (int it = 0; it < B_IT; ++it) {
            const int br = rowp + 64 * it;
            if (br < BN) *(uint4*)(Bs + br * kRowBytes + j * 16) = stage_ws_piece<T>(b_reg[it]);
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int frag_off = (lane & 31) * kRowBytes + (lane >> 5) * 16;

    load_tile(0);
    store_tile(0);
    __syncthreads();
    for (int kt = 0; kt < nk; ++kt) {
        const int buf = kt & 1;
        if (kt + 1 < nk) load_tile(kt + 1);
        const unsigned char* As = smem[buf] + wm * WM * kRowBytes + frag_off;
        const unsigned char* Bs = smem[buf] + (BM + wn * WN) * kRowBytes + frag_off;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            uint4 af[TM], bfr[TN];
#pragma unroll
            for (int a = 0; a < TM; ++a) af[a] = *(const uint4*)(As + a * 32 * kRowBytes + g * 32);
#pragma unroll
            for (int b = 0; b < TN; ++b) bfr[b] = *(const uint4*)(Bs + b * 32 * kRowBytes + g * 32);
#pragma unroll
            for (int a = 0; a < TM; ++a)
#pragma unroll
                for (int b = 0; b < TN; ++b) mfma_kgroup_ss<T>(af[a], bfr[b], acc[a][b]);    // A = gathered pixels, B = weights (both staged: common.hpp)
        }
        if (kt + 1 < nk) store_tile(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: bias, residual, activation, layout-aware store
    const bool plain = p.store_mode == 0 && p.out_H == p.Ho && p.out_W == p.Wo;
#pragma unroll
    for (int b = 0; b < TN; ++b) {
        const int col = n0 + wn * WN + b * 32 + (lane & 31);
        const bool col_ok = col < p.Cout;
        const float bias = (p.bias && col_ok) ? p.bias[col] : 0.f;
#pragma unroll
        for (int a = 0; a < TM; ++a) {
            const int mbase = m0 + wm * WM + a * 32 + 4 * (lane >> 5);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = mbase + (r & 3) + 8 * (r >> 2);
                float v = acc[a][b][r] + bias;
                if (col_ok && m < p.M) {
                    if (p.residual) v += load_elem<T>((const T*)p.residual, (size_t)m * p.Cout + col);
                    if (p.act == 1) v = fmaxf(v, 0.f);
                    else if (p.act == 2) v = gelu_t<T>(v);
                    else if (p.act >= 3) v = apply_act<T>(v, p.act);
                    if (plain) store_elem<T>((T*)p.out, (size_t)m * p.Cout + col, v);
                    else store_generic<T>(p.out, p.store_mode, p.Ho, p.Wo, p.out_H, p.out_W, p.Cout, m, col, v);
                }
            }
        }
    }
}

template <typename T, bool SMALLC>
static int launch_igemm(const IgemmParams& p, hipStream_t stream) {
    const int M = p.M;
    auto blocks = [&](int bm, int bn) { return ((M + bm - 1) / bm) * ((p.Cout + bn - 1) / bn); };
    if (p.Cout > 64) {
        // a 128x128 tile needs enough tiles to fill 256 CUs; otherwise halve the N tile
        if (blocks(128, 128) >= 512) {
            hipLaunchKernelGGL((igemm_kernel<T, 128, 128, 64, 64, SMALLC>), dim3(blocks(128, 128)), dim3(256), 0, stream, p);
        } else {
            hipLaunchKernelGGL((igemm_kernel<T, 128, 64, 64, 32, SMALLC>), dim3(blocks(128, 64)), dim3(256), 0, stream, p);
        }
    } else if (p.Cout > 32) {
        hipLaunchKernelGGL((igemm_kernel<T, 128, 64, 64, 32, SMALLC>), dim3(blocks(128, 64)), dim3(256), 0, stream, p);
    } else {
        hipLaunchKernelGGL((igemm_kernel<T, 128, 32, 32, 32, SMALLC>), dim3(blocks(128, 32)), dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_conv2d_nhwc(const void* in, const void* wgt, const float* bias, const void* residual,
                                  const float* pre_scale, const float* pre_shift, const int* klut, void* out,
                                  const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W, Cin, Ho, Wo, Cout, Kh, Kw, stride, pad, K, Kpad, upsample, pre_relu, act,
    //        store_mode, out_H, out_W, in_is_f32_smallc]
    if (!in || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    IgemmParams p;
    const int dtype = dims[0];
    p.in = in; p.wgt = wgt; p.bias = bias; p.residual = residual;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.klut = klut; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cin = dims[4];
    p.Ho = dims[5]; p.Wo = dims[6]; p.Cout = dims[7];
    p.Kh = dims[8]; p.Kw = dims[9]; p.stride = dims[10]; p.pad = dims[11];
    p.K = dims[12]; p.Kpad = dims[13];
    p.upsample = dims[14]; p.pre_relu = dims[15]; p.act = dims[16]; p.store_mode = dims[17];
    p.out_H = dims[18]; p.out_W = dims[19];
    const int smallc = dims[20];
    p.M = p.N * p.Ho * p.Wo;
    if (p.M <= 0 || p.Cout <= 0 || p.K <= 0) return COBEVT_ERR_SHAPE;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    const int bke = dtype == 0 ? 32 : 16, ch = dtype == 0 ? 8 : 4;
    if (p.Kpad % bke != 0 || p.Kpad < p.K) return COBEVT_ERR_SHAPE;
    if (p.store_mode < 0 || p.store_mode > 3) return COBEVT_ERR_ARG;
    if (p.store_mode == 1 && ((p.Ho | p.Wo) & 1)) return COBEVT_ERR_SHAPE;
    if ((p.store_mode == 1 || p.store_mode == 2) && p.residual) return COBEVT_ERR_UNSUPPORTED;
    if (smallc) {
        if (!klut || p.pre_scale || p.upsample) return COBEVT_ERR_ARG;
        if (p.Cin > 1023 || p.Kh > 1023 || p.Kw > 1023) return COBEVT_ERR_SHAPE;
        return dtype == 0 ? launch_igemm<bf16_t, true>(p, stream) : launch_igemm<float, true>(p, stream);
    }
    if (p.Cin % ch != 0) return COBEVT_ERR_SHAPE;
    if (p.K != p.Kh * p.Kw * p.Cin) return COBEVT_ERR_SHAPE;
    return dtype == 0 ? launch_igemm<bf16_t, false>(p, stream) : launch_igemm<float, false>(p, stream);
}
