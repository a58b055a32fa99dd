// 3x3 / stride 1 / pad 1 convolution on MFMA with an LDS-resident input patch (gfx950).
//
// The generic implicit GEMM (igemm.hip) re-gathers the input once per filter tap; for the 3x3 convolutions that
// carry ~85 % of the CoBEVT frame's FLOPs (ResNet-34 BasicBlocks reached from
// opv2v/opencood/models/backbones/resnet_ms.py:67-74, FAX Bottlenecks / downsample blocks fax_modules.py:472-489,
// NaiveDecoder naive_decoder.py:78-87) that is 9x redundant L2->LDS traffic and leaves only 4-8 MFMAs between
// barriers.  Here a workgroup owns a TH x 16 tile of output pixels of one image and BN output channels:
//   for each 128-byte channel chunk (64 bf16 / 32 fp32 channels):
//       stage the (TH+2) x 18 pixel input patch of that chunk in LDS once (zero filled outside the image = padding;
//       optionally reading a nearest-x2 up-sampled view of the stored input, i.e. F.interpolate folded in);
//       for each of the 9 taps: stream the [BN][chunk] weight slice through a double-buffered LDS tile and issue
//       16 MFMAs (32x32 tiles, 2x2 per wave) per wave whose A fragments are read straight out of the patch at
//       the tap's pixel offset.
// Weights are laid out [Cout][chunk][tap][chunk channels] on the host so a tap's slice is contiguous.
// Epilogue as in igemm.hip: folded-BN bias, residual add, ReLU, NHWC store or PixelUnshuffle(2) store.
#include "common.hpp"

namespace cobevt {

struct Conv3Params {
    const void* in;
    const void* wgt;
    const float* bias;
    const void* residual;
    void* out;
    int N, H, W, Cin;   // stored input dims
    int Ho, Wo, Cout;   // output dims (= virtual input dims: 2H x 2W when upsample)
    int upsample;
    int act;
    int store_mode;     // 0 NHWC, 1 PixelUnshuffle(2) NHWC
    int tiles_y, tiles_x, tiles_n;
};

// KG = 32-byte k-groups per chunk (4 -> 128-byte chunks, 2 -> 64-byte chunks for Cin = 32 bf16)
template <typename T, int BN, int KG> struct Conv3Cfg {
    static constexpr int TW = 16;
    static constexpr int TH = BN == 64 ? 16 : 8;               // 256 or 128 output pixels per workgroup
    static constexpr int PH = TH + 2, PW = TW + 2;
    static constexpr int PSTR = KG * 32 + 16;                  // patch pixel stride: odd multiple of 16 bytes
    // patch row stride rounded to the 256-byte LDS bank row: the two spatial rows of a 32-pixel MFMA tile then hit
    // complementary 16-byte slots (slot = 9*px mod 16) -> conflict-free ds_read_b128 A fragments
    static constexpr int PROW = (PW * PSTR + 255) / 256 * 256;
    static constexpr int WSTR = KG * 32 + 16;                  // weight row stride (bytes)
    static constexpr int PATCH_BYTES = PH * PROW;
    static constexpr int W_BYTES = BN * WSTR;
    static constexpr int SSTR = BN * 4 + 16;                   // fp32 staging row stride of the epilogue
    static constexpr int STAGE_BYTES = TH * TW * SSTR;
    static constexpr int MAIN_BYTES = PATCH_BYTES + 2 * W_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
};

constexpr int kConv3Threads = 512;   // 8 waves: 2 per SIMD from one workgroup, 4 with two workgroups per CU

template <typename T, int BN, int KG>
__global__ __launch_bounds__(kConv3Threads) void conv3x3_kernel(Conv3Params p) {
    using C = Conv3Cfg<T, BN, KG>;
    constexpr int NT = kConv3Threads;
    constexpr int CH = Elem<T>::kChunk;                  // elements per 16 bytes
    constexpr int CC = KG * 32 / Elem<T>::kBytes;        // channels per chunk
    constexpr int TW = C::TW, TH = C::TH, PH = C::PH, PW = C::PW;
    constexpr int PSTR = C::PSTR, PROW = C::PROW, WSTR = C::WSTR;
    constexpr int PIECES = 2 * KG;                       // 16-byte pieces per pixel / weight row
    constexpr int PATCH_ITEMS = PH * PW * PIECES;
    constexpr int P_IT = (PATCH_ITEMS + NT - 1) / NT;
    constexpr int W_ITEMS = BN * PIECES;
    constexpr int W_IT = (W_ITEMS + NT - 1) / NT;
    constexpr int WAVES_N = BN / 64;                     // 1 or 2 ; each wave owns 32 pixels x 64 couts

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wbuf = smem + C::PATCH_BYTES;         // 2 x [BN][WSTR]

    // XCD-aware bijective remap (same as igemm.hip): n-tiles of one spatial tile are adjacent on one XCD
    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tn = logical % p.tiles_n;
    int rest = logical / p.tiles_n;
    const int tx = rest % p.tiles_x; rest /= p.tiles_x;
    const int ty = rest % p.tiles_y;
    const int img = rest / p.tiles_y;
    const int oy0 = ty * TH, ox0 = tx * TW, n0 = tn * BN;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;  // wave rows 2*wm, 2*wm+1 of the tile
    const int nchunk = p.Cin / CC;
    const T* in = (const T*)p.in;
    const T* wg = (const T*)p.wgt;

    // ---- addresses computed once.  Every prefetch load below is UNCONDITIONAL (invalid items read a clamped, valid
    // address and are zeroed by a select when they are written to LDS): with loads under per-lane or uniform branches
    // LLVM's waitcnt insertion falls back to s_waitcnt vmcnt(0) at the control-flow merge, i.e. it also waits for the
    // prefetch issued in the same step and every tap serialises on a global-load latency (seen in the ISA).
    long pgoff[P_IT];
    int plds[P_IT];
    bool pzero[P_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int item = tid + it * NT;
        pgoff[it] = 0;
        plds[it] = -1;
        pzero[it] = true;
        if (item < PATCH_ITEMS) {
            const int pix = item / PIECES, j = item - pix * PIECES;
            const int py = pix / PW, px = pix - py * PW;
            plds[it] = py * PROW + px * PSTR + j * 16;
            const int vy = oy0 - 1 + py, vx = ox0 - 1 + px;         // coordinates in the (virtual) input
            if (vy >= 0 && vy < p.Ho && vx >= 0 && vx < p.Wo) {
                const int sy = p.upsample ? (vy >> 1) : vy, sx = p.upsample ? (vx >> 1) : vx;
                pgoff[it] = (((long)img * p.H + sy) * p.W + sx) * p.Cin + j * CH;
                pzero[it] = false;
            }
        }
    }
    // weights: [Cout][chunk][tap][CC] -> the slice of step s = chunk*9+tap starts at row_base + s*CC
    const T* wptr[W_IT];
    int wlds[W_IT];
    bool wzero[W_IT];
#pragma unroll
    for (int it = 0; it < W_IT; ++it) {
        const int item = (tid + it * NT) % W_ITEMS;
        const int row = item / PIECES, j = item - row * PIECES;
        const bool mine = tid + it * NT < W_ITEMS;
        wlds[it] = mine ? row * WSTR + j * 16 : -1;
        wzero[it] = n0 + row >= p.Cout;
        const int crow = n0 + row < p.Cout ? n0 + row : p.Cout - 1;
        wptr[it] = wg + ((size_t)crow * nchunk * 9 * CC + j * CH);
    }

    uint4 preg[P_IT];
    uint4 wregA[W_IT], wregB[W_IT];

    auto load_patch = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) preg[it] = *(const uint4*)(in + pgoff[it] + chunk * CC);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (plds[it] >= 0) *(uint4*)(patch + plds[it]) = pzero[it] ? make_uint4(0, 0, 0, 0) : preg[it];
    };
    auto load_w = [&](uint4 (&wreg)[W_IT], int step) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) wreg[it] = *(const uint4*)(wptr[it] + step * CC);
    };
    auto store_w = [&](const uint4 (&wreg)[W_IT], int buf) {
        unsigned char* wb = wbuf + buf * C::W_BYTES;
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (wlds[it] >= 0) *(uint4*)(wb + wlds[it]) = wzero[it] ? make_uint4(0, 0, 0, 0) : wreg[it];
    };

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;

    const int abase = (2 * wm + (ql >> 4)) * PROW + (ql & 15) * PSTR + h * 16;
    const int bbase = (wn * 64 + ql) * WSTR + h * 16;
    const int nsteps = nchunk * 9;

    // one tap: weights two steps ahead in registers (RI: issue for step+2, RS: holds step+1), two LDS weight buffers
    auto body = [&](int step, int chunk, int tap, uint4 (&RI)[W_IT], uint4 (&RS)[W_IT], int buf) {
        const bool next_chunk = chunk + 1 < nchunk;
        load_w(RI, step + 2 < nsteps ? step + 2 : nsteps - 1);       // always issued (clamped): keeps the loop branch-free
        if (tap == 6 && next_chunk) load_patch(chunk + 1);
        const int kh = tap / 3, kw = tap - kh * 3;
        const unsigned char* pa = patch + kh * PROW + kw * PSTR + abase;
        const unsigned char* pb = wbuf + buf * C::W_BYTES + bbase;
#pragma unroll
        for (int g = 0; g < KG; ++g) {
            const uint4 af = *(const uint4*)(pa + g * 32);
            uint4 bf[2];
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[b] = *(const uint4*)(pb + b * 32 * WSTR + g * 32);
#pragma unroll
            for (int b = 0; b < 2; ++b) mfma_kgroup<T>(af, bf[b], acc[b]);
        }
        if (step + 1 < nsteps) store_w(RS, buf ^ 1);
        if (tap == 8 && next_chunk) {
            __syncthreads();          // every wave finished reading the old patch
            store_patch();
        }
        __syncthreads();
    };

    load_patch(0);
    load_w(wregA, 0);
    store_patch();
    store_w(wregA, 0);
    load_w(wregB, nsteps > 1 ? 1 : 0);
    __syncthreads();
    {
        int chunk = 0, tap = 0;
        for (int step = 0; step < nsteps; step += 2) {
            body(step, chunk, tap, wregA, wregB, 0);
            if (++tap == 9) { tap = 0; ++chunk; }
            if (step + 1 < nsteps) {
                body(step + 1, chunk, tap, wregB, wregA, 1);
                if (++tap == 9) { tap = 0; ++chunk; }
            }
        }
    }

    // ---- epilogue: stage acc + bias as fp32 [pixel][cout] in LDS, then 16-byte coalesced residual / store passes
    float* stage = (float*)smem;
    constexpr int SROW = C::SSTR / 4;                    // floats per staged pixel row
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int cl = wn * 64 + b * 32 + ql;            // column inside the BN tile
        const float bias = (p.bias && n0 + cl < p.Cout) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = acc_row(r, lane);
            const int px = (2 * wm + (row >> 4)) * TW + (row & 15);
            stage[px * SROW + cl] = acc[b][r] + bias;
        }
    }
    __syncthreads();
    T* out = (T*)p.out;
    if (p.store_mode == 0) {
        constexpr int CPP = BN / CH;                     // 16-byte output chunks per pixel
        for (int item = tid; item < TH * TW * CPP; item += kConv3Threads) {
            const int px = item / CPP, cj = item - px * CPP;
            const int oy = oy0 + px / TW, ox = ox0 + (px % TW);
            const int col = n0 + cj * CH;
            if (oy >= p.Ho || ox >= p.Wo || col >= p.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = stage[px * SROW + cj * CH + e];
            const size_t o = (((size_t)img * p.Ho + oy) * p.Wo + ox) * p.Cout + col;
            if (col + CH <= p.Cout && (p.Cout % CH) == 0) {
                if (p.residual) {
                    float rv[8];
                    chunk_to_f32<T>(*(const uint4*)((const T*)p.residual + o), rv);
#pragma unroll
                    for (int e = 0; e < CH; ++e) v[e] += rv[e];
                }
#pragma unroll
                for (int e = 0; e < CH; ++e) v[e] = p.act == 1 ? fmaxf(v[e], 0.f) : (p.act == 2 ? gelu_erf(v[e]) : v[e]);
                *(uint4*)(out + o) = f32_to_chunk<T>(v);
            } else {                                      // ragged Cout tail (Cout not a multiple of the chunk)
                for (int e = 0; e < CH && col + e < p.Cout; ++e) {
                    float x = v[e];
                    if (p.residual) x += load_elem<T>((const T*)p.residual, o + e);
                    x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_erf(x) : x);
                    store_elem<T>(out, o + e, x);
                }
            }
        }
    } else {
        // PixelUnshuffle(2): out[(oy/2, ox/2)][c*4 + (oy&1)*2 + (ox&1)] = conv[(oy, ox)][c]   (no residual here)
        constexpr int QP = (TH / 2) * (TW / 2);          // output pixels per tile
        const int ocn = BN * 4;                          // output channels produced by this tile
        const int cpp = ocn / CH;
        for (int item = tid; item < QP * cpp; item += kConv3Threads) {
            const int qp = item / cpp, cj = item - qp * cpp;
            const int qy = qp / (TW / 2), qx = qp % (TW / 2);
            const int oy2 = (oy0 >> 1) + qy, ox2 = (ox0 >> 1) + qx;
            if (oy2 >= (p.Ho >> 1) || ox2 >= (p.Wo >> 1)) continue;
            float v[8];
            bool any = false;
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int oc = cj * CH + e, c = oc >> 2, q = oc & 3;
                const int px = (2 * qy + (q >> 1)) * TW + 2 * qx + (q & 1);
                float x = stage[px * SROW + c];
                x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_erf(x) : x);
                v[e] = x;
                any |= (n0 + c) < p.Cout;
            }
            if (!any) continue;
            const size_t o = (((size_t)img * (p.Ho >> 1) + oy2) * (p.Wo >> 1) + ox2) * (size_t)(p.Cout * 4) + (size_t)n0 * 4 + cj * CH;
            if (n0 + ((cj * CH + CH - 1) >> 2) < p.Cout && ((p.Cout * 4) % CH) == 0) *(uint4*)(out + o) = f32_to_chunk<T>(v);
            else for (int e = 0; e < CH; ++e) if (n0 + ((cj * CH + e) >> 2) < p.Cout) store_elem<T>(out, o + e, v[e]);
        }
    }
}

template <typename T, int BN, int KG>
static int launch_conv3(Conv3Params p, hipStream_t stream) {
    using C = Conv3Cfg<T, BN, KG>;
    p.tiles_y = (p.Ho + C::TH - 1) / C::TH;
    p.tiles_x = (p.Wo + 15) / 16;
    p.tiles_n = (p.Cout + BN - 1) / BN;
    const long blocks = (long)p.N * p.tiles_y * p.tiles_x * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    constexpr size_t lds = C::LDS_BYTES;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)conv3x3_kernel<T, BN, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        attr_set = true;
    }
    hipLaunchKernelGGL((conv3x3_kernel<T, BN, KG>), dim3((unsigned)blocks), dim3(kConv3Threads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_conv3x3_nhwc(const void* in, const void* wgt, const float* bias, const void* residual, void* out,
                                   const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W, Cin, Cout, upsample, act, store_mode, chunk_channels]
    if (!in || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    Conv3Params p;
    const int dtype = dims[0];
    p.in = in; p.wgt = wgt; p.bias = bias; p.residual = residual; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cin = dims[4]; p.Cout = dims[5];
    p.upsample = dims[6]; p.act = dims[7]; p.store_mode = dims[8];
    const int cc = dims[9];
    p.Ho = p.upsample ? 2 * p.H : p.H;
    p.Wo = p.upsample ? 2 * p.W : p.W;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.Cin < 1 || p.Cout < 1) return COBEVT_ERR_SHAPE;
    if (p.store_mode != 0 && p.store_mode != 1) return COBEVT_ERR_ARG;
    if (p.store_mode == 1 && (p.residual || ((p.Ho | p.Wo) & 1))) return COBEVT_ERR_UNSUPPORTED;
    if (p.Cin % cc != 0) return COBEVT_ERR_SHAPE;
    const int kg = cc * (dtype == 0 ? 2 : 4) / 32;      // 32-byte k-groups per chunk
    const bool wide = p.Cout > 64;
    if (dtype == 0) {
        if (kg == 4) return wide ? launch_conv3<bf16_t, 128, 4>(p, stream) : launch_conv3<bf16_t, 64, 4>(p, stream);
        if (kg == 2) return wide ? launch_conv3<bf16_t, 128, 2>(p, stream) : launch_conv3<bf16_t, 64, 2>(p, stream);
    } else {
        if (kg == 4) return wide ? launch_conv3<float, 128, 4>(p, stream) : launch_conv3<float, 64, 4>(p, stream);
        if (kg == 2) return wide ? launch_conv3<float, 128, 2>(p, stream) : launch_conv3<float, 64, 2>(p, stream);
    }
    return COBEVT_ERR_SHAPE;
}
