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
#include <type_traits>

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
    // cobevt_conv3x3_ds_wfrag_nhwc: the 1x1 / stride-2 projection shortcut of a down-sampling BasicBlock as extra one-tap channel chunks
    const void* in2 = nullptr;   // x, (N, H2, W2, Cin2): the block's input, read at (2 oy, 2 ox)
    const float* bias2 = nullptr;   // the shortcut's bias (its folded BatchNorm shift)
    int H2 = 0, W2 = 0, Cin2 = 0;
};

// NHWC store pass shared by the three kernels: the fp32 staging tile [NPX pixels][BN couts] -> + residual -> activation
// -> 16-byte stores.  Trip count and item mapping are compile-time, so everything is unrolled and every residual load of
// a thread is in flight before the first is consumed (a rolled loop serialises one global round trip per item).  It is
// split in two so a kernel can issue the residual loads (prepare) well before the accumulators are final: with one
// workgroup per CU all CUs reach their epilogue together and the residual reads + output writes of the whole layer hit
// HBM as one burst (7-8k cycles for a 160 x 128 tile in the s_memtime trace); loads issued a channel chunk earlier
// arrive while the MFMA loop still runs.  coord(px, img, oy, ox) returns false for pixels outside the image / past
// the last strip.  off[] < 0 marks an item with nothing to store.
template <typename T, int NT, int NPX, int BN> struct Conv3Store {
    static constexpr int CH = Elem<T>::kChunk;
    static constexpr int CPP = BN / CH;
    static constexpr int ITEMS = NPX * CPP;
    static constexpr int NIT = (ITEMS + NT - 1) / NT;
    long off[NIT];
    uint4 rres[NIT];

    template <typename Coord>
    __device__ __forceinline__ void prepare(const Conv3Params& p, int tid, int n0, Coord coord) {
        const bool vec = (p.Cout % CH) == 0;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = tid + i * NT;
            const int px = item / CPP, cj = item - px * CPP;
            const int col = n0 + cj * CH;
            int img = 0, oy = 0, ox = 0;
            rres[i] = make_uint4(0, 0, 0, 0);
            const bool ok = coord(px < NPX ? px : 0, img, oy, ox) & (item < ITEMS) & (col < p.Cout);
            const long o = (((long)img * p.Ho + oy) * p.Wo + ox) * p.Cout + col;
            off[i] = ok ? o : -1;
        }
        // all residual loads of the thread in flight together: unconditional (address clamped to element 0 for the
        // lanes that store nothing) - a load under a divergent branch makes LLVM wait for vmcnt(0) right after it, which
        // serialised these HBM round trips (3.4k cycles in the s_memtime trace of the strip kernel)
        if (p.residual && vec) {
#pragma unroll
            for (int i = 0; i < NIT; ++i) rres[i] = *(const uint4*)((const T*)p.residual + (off[i] < 0 ? 0 : off[i]));
        }
    }

    __device__ __forceinline__ void finish(const Conv3Params& p, const float* stage, int srow, int tid, int n0) {
        T* out = (T*)p.out;
        const bool vec = (p.Cout % CH) == 0;
        // all staging reads of the thread first (unconditional, row clamped), then the arithmetic and the stores: under
        // the per-item `off < 0` branch every item paid its own LDS round trip
        float sv[NIT][CH];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = tid + i * NT;
            const int px = item / CPP, cj = item - px * CPP;
            const float* src = stage + (px < NPX ? px : 0) * srow + cj * CH;
#pragma unroll
            for (int e = 0; e < CH; ++e) sv[i][e] = src[e];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (off[i] < 0) continue;
            const int item = tid + i * NT;
            const int px = item / CPP, cj = item - px * CPP;
            float v[8], rv[8];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = sv[i][e];
            if (vec) {
                if (p.residual) {
                    chunk_to_f32<T>(rres[i], rv);
#pragma unroll
                    for (int e = 0; e < CH; ++e) v[e] += rv[e];
                }
#pragma unroll
                for (int e = 0; e < CH; ++e) v[e] = p.act == 1 ? fmaxf(v[e], 0.f) : (p.act == 2 ? gelu_t<T>(v[e]) : v[e]);
                *(uint4*)(out + off[i]) = f32_to_chunk<T>(v);
            } else {                                      // ragged Cout (not a multiple of the 16-byte chunk): scalar
                const int col = n0 + cj * CH;
                for (int e = 0; e < CH && col + e < p.Cout; ++e) {
                    float x = v[e];
                    if (p.residual) x += load_elem<T>((const T*)p.residual, off[i] + e);
                    x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_t<T>(x) : x);
                    store_elem<T>(out, off[i] + e, x);
                }
            }
        }
    }
};

template <typename T, int NT, int NPX, int BN, typename Coord>
__device__ __forceinline__ void conv3_store_nhwc(const Conv3Params& p, const float* stage, int srow, int tid, int n0, Coord coord) {
    Conv3Store<T, NT, NPX, BN> st;
    st.prepare(p, tid, n0, coord);
    st.finish(p, stage, srow, tid, n0);
}

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
    float ebias[2];                                      // the epilogue's two bias values per lane, fetched up front
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int c = n0 + wn * 64 + b * 32 + ql;
        ebias[b] = p.bias ? p.bias[p.bias && c < p.Cout ? c : 0] : 0.f;
        if (c >= p.Cout) ebias[b] = 0.f;
    }

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
            if (plds[it] >= 0) *(uint4*)(patch + plds[it]) = pzero[it] ? make_uint4(0, 0, 0, 0) : stage_x_piece<T>(preg[it]);
    };
    auto load_w = [&](uint4 (&wreg)[W_IT], int step) {
#pragma unroll
        for (int it = 0; it < W_IT; ++it) wreg[it] = *(const uint4*)(wptr[it] + step * CC);
    };
    auto store_w = [&](const uint4 (&wreg)[W_IT], int buf) {
        unsigned char* wb = wbuf + buf * C::W_BYTES;
#pragma unroll
        for (int it = 0; it < W_IT; ++it)
            if (wlds[it] >= 0) *(uint4*)(wb + wlds[it]) = wzero[it] ? make_uint4(0, 0, 0, 0) : stage_ws_piece<T>(wreg[it]);
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
            for (int b = 0; b < 2; ++b) mfma_kgroup_ss<T>(af, bf[b], acc[b]);    // A = pixels, B = weights (both staged: common.hpp)
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
        const float bias = ebias[b];
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
        conv3_store_nhwc<T, kConv3Threads, TH * TW, BN>(p, stage, SROW, tid, n0, [&](int px, int& im, int& oy, int& ox) {
            im = img; oy = oy0 + px / TW; ox = ox0 + (px % TW);
            return oy < p.Ho && ox < p.Wo;
        });
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
                x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_t<T>(x) : x);
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
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)conv3x3_kernel<T, BN, KG>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((conv3x3_kernel<T, BN, KG>), dim3((unsigned)blocks), dim3(kConv3Threads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------------------------
// Fragment-ordered weights (cobevt_conv3x3_wfrag_nhwc).
//
// The kernel above stages every tap's weight slice through LDS and pays one workgroup barrier per tap.  Here the host
// lays the weights out in MFMA B-fragment order
//     [Cout/32][Cin/cc][9 taps][KG k-groups][64 lanes][16 bytes]
// so the weight operand of one (32-cout tile, tap, k-group) is a single fully coalesced 1-KB wave load that goes
// straight from L2 into registers: no weight LDS traffic and no per-tap barrier.  LDS only holds the input patch (double
// buffered across channel chunks -> one barrier per nine taps) and the fragments of the tap two steps ahead are
// prefetched into a register ring while the current tap runs.
// ---------------------------------------------------------------------------------------------------------------------
// Where the global loads of the NEXT chunk's patch (and of the store pass's residual) are issued inside a chunk, and how their LDS
// stores go out.  s_waitcnt vmcnt is in-order: every weight-fragment wait behind such a load is also a wait for it, and these come
// from HBM / Infinity Cache (1.5-2k cycles under load) while the fragments are L2 hits - issued in front of tap 0 (the round-2 code)
// the first fragment wait that covers them is tap 2's, ~1.1k cycles later; issued at tap T, right behind tap T + 2's fragment request,
// it is tap T + 3's.  And the six ds_write_b128 per thread of that patch used to go out as ONE burst in tap 6, where both waves of
// every SIMD sat in the LDS store path together: spread in equal shares over taps FIRST .. 8 they hide between the MFMAs.  Same-job A/B
// (bit-identical outputs): 20 back-to-back launches of ONE layer in a replayed graph, i.e. input and weights hot in L2 / Infinity Cache
// (tools/conv_sched_probe.py, profiles/r03_conv_sched_probe.txt): 256 -> 256 on 20 x 32 x 32 27.3 -> 25.4 us, 512 -> 512 on 20 x 16 x 16
// 31.9 -> 25.9 us, 128 -> 128 on 20 x 64 x 64 34-37 -> 32 us (the spread stores are most of it; any load tap in 0..5 with any first store
// tap in 3..7 within 1 %).  INSIDE THE FRAME, where every launch reads what the previous kernel just wrote (L2 is written back and
// invalidated at kernel boundaries, so the patch and the residual come from the Infinity Cache and the same kernels take 29.8 us
// instead of 25.4), the gain is 1-1.5 % per launch (30.22 -> 29.77 us, 28.82 -> 28.57 us, kernel trace of both builds in one job) and
// the frame time does not move (5 alternating bench runs): the hot-cache microbenchmark overstated it.
#ifndef COBEVT_CONV3_PLOAD_TAP
#define COBEVT_CONV3_PLOAD_TAP 3
#endif
#ifndef COBEVT_CONV3_PSTORE_SPREAD
#define COBEVT_CONV3_PSTORE_SPREAD 1
#endif
#ifndef COBEVT_CONV3_PSTORE_FIRST
#define COBEVT_CONV3_PSTORE_FIRST 6      // spread form: the stores go out in equal shares over taps FIRST .. 8
#endif
#ifndef COBEVT_CONV3_CONT
#define COBEVT_CONV3_CONT 1           // the A-operand ring runs on across channel chunks (bf16, stride 1, eight waves); 0 = restart it behind every chunk barrier
#endif
#ifndef COBEVT_CONV3_CONT_LEAD
#define COBEVT_CONV3_CONT_LEAD 3      // CONT: taps between the request of the next chunk's patch and its first LDS store (as PLOAD_TAP 3 -> PSTORE_FIRST 6)
#endif
#ifndef COBEVT_CONV3_KNOCK
#define COBEVT_CONV3_KNOCK 0          // tools/conv_probe.py builds knock-out copies of this file (never the product .so)
#endif
#ifdef COBEVT_CONV3_TRACE
__device__ unsigned long long cobevt_conv3_trace[64];
__device__ unsigned long long cobevt_conv3_rt[3 * 2048];    // per workgroup: s_memrealtime (100 MHz) at entry / exit, XCC id
#define COBEVT_TRACE_MARK(i) do { if (logical == 0 && tid == 0) cobevt_conv3_trace[(i)] = __builtin_readcyclecounter(); } while (0)
#define COBEVT_TRACE_RT(i) do { if (threadIdx.x == 0 && blockIdx.x < 2048) cobevt_conv3_rt[blockIdx.x * 3 + (i)] = (i) == 2 ? (unsigned long long)(__builtin_amdgcn_s_getreg((20 << 0) | (0 << 6) | (3 << 11)) & 15) : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define COBEVT_TRACE_MARK(i) do {} while (0)
#define COBEVT_TRACE_RT(i) do {} while (0)
#endif

// ---------------------------------------------------------------------------------------------------------------------
// Strip tiling.
//
// Measured with s_memtime, one 8-wave workgroup with a fixed 128 px x 128 cout tile already keeps its CU's MFMA pipes
// ~75-85 % busy (a tap of 64 MFMAs takes 500-700 cycles against 512 of pure MFMA issue); what it loses is the tail:
// ResNet-34's layers on the 20 camera images of a 5-agent frame make 320 or 160 such tiles for 256 CUs, so the kernel
// lasts two workgroup lifetimes at 62 % occupancy.  Here the pixel tile is MT independent STRIPS of 2 rows x 16
// pixels (one 32-row MFMA tile each, with its own 4 x 18 pixel halo patch), strips are numbered across
// (image, row pair, column block), and MT is picked per launch so that the grid is a whole number of workgroups per
// CU (MT = 5 on 20 images: 256 or 512 workgroups).  The 8 waves are WN cout tiles x KS k-splits: every wave walks all
// MT strips for its 32 couts and its share of each tap's four k-groups, so no B fragment is loaded twice in a
// workgroup; the k-split partial sums meet in the fp32 staging tile of the epilogue.
// ---------------------------------------------------------------------------------------------------------------------
// S = convolution stride.  S = 2 (the first conv of a down-sampling BasicBlock, resnet_ms.py:67-74): a strip of 2 x 16
// output pixels needs 5 x 33 input pixels; they are stored de-interleaved by column parity ([row][parity][17 pixels]) so
// that a tap's A fragments are again 16 consecutive pixels at the 144-byte stride (conflict-free ds_read_b128), and the
// patch is single-buffered (25.6 KB per strip; these layers have 1-4 channel chunks, the refill is exposed 0-3 times).
// NW = waves per workgroup.  8: one workgroup per CU, patch double-buffered across channel chunks.  4 (stride 1, one 32-cout tile x
// four k-splits): for layers with <= 32 output channels (FAX down-sampling 128 -> 32, decoder tails), where a 64-cout tile wastes
// half its MFMAs - the patch is single-buffered (46 KB), two workgroups share a CU and fill each other's barrier / refill gaps.
// In-graph: 128 -> 32 on 5 x 128 x 128 22.1 -> 12.9 us, on 5 x 64 x 64 13.3 -> 7.1 us (tools/conv_graph_probe.py).  The same
// half-size form with 64-cout tiles measured SLOWER than the eight-wave kernel on the 256- / 512-channel layers (29.2 vs 27.3 us:
// twice the patch fills, exposed patch stores) and is not built.
template <typename T, int MT, int WN, int KS, int S = 1, int NW = 8> struct Conv3SCfg {
    static constexpr int NT = 64 * NW, KG = 4, KGW = KG / KS;
    static constexpr bool HALF = NW == 4;
    static constexpr int BN = WN * 32;
    static constexpr int PW = S == 1 ? 18 : 33;               // input pixels per patch row
    static constexpr int SROWS = S == 1 ? 4 : 5;              // input rows per strip
    static constexpr int PSTR = KG * 32 + 16;
    static constexpr int PLANE = 17 * PSTR;                   // S = 2: one column-parity plane of a patch row
    static constexpr int PROW = ((S == 1 ? PW * PSTR : 2 * PLANE) + 255) / 256 * 256;
    static constexpr int STRIP_BYTES = SROWS * PROW;
    static constexpr int PATCH_BYTES = MT * STRIP_BYTES;
    static constexpr int SSTR = BN * 4 + 16;
    static constexpr int STAGE_BYTES = MT * 32 * SSTR;
    static constexpr int MAIN_BYTES = (S == 1 && !HALF ? 2 : 1) * PATCH_BYTES;
    static constexpr int LDS_BYTES = MAIN_BYTES > STAGE_BYTES ? MAIN_BYTES : STAGE_BYTES;
    static_assert(WN * KS == NW && (NW == 8 || (NW == 4 && S == 1)), "eight waves, or four (stride 1)");
};

// DS (bf16, stride 1, eight waves): out = act(conv3x3(in) + bias + round(conv1x1/s2(in2) + bias2)) - the second convolution of a
// down-sampling BasicBlock with the block's projection shortcut in the same launch: behind the nine-tap chunks of `in` a wave computes
// the complete shortcut of some of its (strip, 32 couts) tiles from x(2 oy, 2 ox) (Cin2 / 64 chunks x four k-groups, one 16-register
// accumulator), rounds it as its own launch would store it and adds it to its partial sum; the shortcut's weight fragments follow the
// 3x3's in the table ([tile][Cin/64 * 9 + Cin2/64 steps]).
template <typename T, int MT, int WN, int KS, int S, int NW, bool DS = false>
__global__ __launch_bounds__(64 * NW, NW == 4 ? 2 : 1) void conv3x3_strips_kernel(Conv3Params p) {
    static_assert(!DS || (Elem<T>::kIsBf16 && S == 1 && NW == 8), "the shortcut form is bf16 / stride 1 / eight waves");
    using C = Conv3SCfg<T, MT, WN, KS, S, NW>;
    constexpr bool HALF = C::HALF;
    constexpr int NT = C::NT, KG = C::KG, KGW = C::KGW, BN = C::BN, PW = C::PW;
    constexpr int CH = Elem<T>::kChunk;
    constexpr int CC = KG * 32 / Elem<T>::kBytes;
    constexpr int PSTR = C::PSTR, PROW = C::PROW, STRIP = C::STRIP_BYTES;
    constexpr int PIECES = 2 * KG;
    constexpr int STRIP_ITEMS = C::SROWS * PW * PIECES;          // 576 16-byte pieces per strip patch
    constexpr int PATCH_ITEMS = MT * STRIP_ITEMS;
    constexpr int P_IT = S == 1 ? (PATCH_ITEMS + NT - 1) / NT : 1;   // S = 2 refills the patch without resident registers
    constexpr int PF = 2, R = 3;
    // third library, fp32 storage, an even number of k-groups per wave and tap: the patch holds ONE fp16 per activation and a 16-byte
    // A operand spans two k-groups (common.hpp kXPack) - KGA operand groups per tap instead of KGW
    constexpr bool PACK = kXPack<T, KGW>;
    // second library, same condition: the three-term form on k-group pairs - hi and lo operands of a pair as two 16-byte reads, an operand
    // ring of two slots instead of three (the registers), three MFMAs per pair (common.hpp kXPack3)
    constexpr bool PACK3 = kXPack3<T, KGW>;
    constexpr int KGA = (PACK || PACK3) ? KGW / 2 : KGW;
    constexpr int RA = PACK3 ? 2 : 3;                 // slots of the A-operand ring
    // CONT: the ring does not restart at a chunk boundary.  A chunk is NG = 9 KGA operand groups, a multiple of the ring's three slots, so
    // group NG + i of chunk c IS group i of chunk c + 1 in the same slot; the read of group m is issued while group m - 2 multiplies, so
    // the chunk barrier (next patch complete, this one no longer read) sits in front of group NG - 2 = the first group of tap TB: the
    // operands of the chunk's last two groups are in registers by then, and the reads issued under them take the next chunk's patch.
    constexpr bool CONT = COBEVT_CONV3_CONT && S == 1 && !HALF && Elem<T>::kIsBf16 && (KGA == 1 || KGA == 2) && MT <= 5;   // MT = 6: the carried ring spills
    constexpr int TB = CONT ? (9 * KGA - 2) / KGA : 9;               // the tap the chunk barrier stands in front of (8, or 7 for KGA = 1)

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    COBEVT_TRACE_RT(0);
    COBEVT_TRACE_RT(2);

    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tn = logical % p.tiles_n;
    const int q0 = (logical / p.tiles_n) * MT;                   // first strip of this workgroup
    const int n0 = tn * BN;
    const int per_img = p.tiles_y * p.tiles_x;
    const int nstrips = p.N * per_img;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wn = wave % WN, ks = wave / WN;
    const int nchunk = p.Cin / CC;
    const int nchunk2 = DS ? p.Cin2 / CC : 0;
    const int nsteps = nchunk * 9 + nchunk2;                     // fragment steps per cout tile (DS: the shortcut's one-tap steps behind the 3x3's)
    const T* in = (const T*)p.in;

    // (image, strip row, strip column) of the workgroup's MT strips, divided once by MT lanes: a runtime integer division
    // is ~50 VALU instructions and the patch / store address set-up used to do two per 16-byte piece (the s_memtime
    // trace showed 3.2k cycles in the store set-up alone)
    __shared__ __attribute__((aligned(16))) int stab[8 * 4];
    __shared__ __attribute__((aligned(16))) float sbias[BN];               // this tile's bias, fetched now: the epilogue must not start with a global round trip
    __shared__ __attribute__((aligned(16))) float sbias2[DS ? BN : 4];     // (DS) the projection shortcut's bias
    if (tid >= NT - BN) {
        const int c = n0 + tid - (NT - BN);
        sbias[tid - (NT - BN)] = (p.bias && c < p.Cout) ? p.bias[c] : 0.f;
        if (DS) sbias2[tid - (NT - BN)] = (p.bias2 && c < p.Cout) ? p.bias2[c] : 0.f;
    }
    if (tid < MT) {
        const int q = q0 + tid;
        const bool ok = q < nstrips;
        const int img = ok ? q / per_img : 0, rem = ok ? q - img * per_img : 0;
        const int sy = rem / p.tiles_x, sx = rem - sy * p.tiles_x;
        *(int4*)&stab[tid * 4] = make_int4(img, sy, sx, ok ? 1 : 0);
    }
    __syncthreads();

    int pgoff[P_IT];        // element offsets (the host entry guarantees the input has < 2^31 elements)
    int plds[P_IT];         // < 0: nothing to write; otherwise bit 30 set = write zeros
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int item = tid + it * NT;
        pgoff[it] = 0;
        plds[it] = -1;
        if (S == 1 && item < PATCH_ITEMS) {
            const int s = item / STRIP_ITEMS, r = item - s * STRIP_ITEMS;
            const int pix = r / PIECES, j = r - pix * PIECES;
            const int py = pix / PW, px = pix - py * PW;
            int lds = s * STRIP + py * PROW + px * PSTR + (PACK ? packed_piece_offset(j) : PACK3 ? packed3_piece_offset(j) : j * 16);
            const int4 sc = *(const int4*)&stab[s * 4];
            const int img = sc.x, sy = sc.y, sx = sc.z;
            const int vy = sy * 2 - 1 + py, vx = sx * 16 - 1 + px;
            const bool valid = (sc.w != 0) & (vy >= 0) & (vy < p.Ho) & (vx >= 0) & (vx < p.Wo);
            const int yy = p.upsample ? (vy >> 1) : vy, xx = p.upsample ? (vx >> 1) : vx;
            pgoff[it] = valid ? ((img * p.H + yy) * p.W + xx) * p.Cin + j * CH : 0;
            plds[it] = valid ? lds : (lds | (1 << 30));
        }
    }
    const uint4* wq = (const uint4*)p.wgt + ((size_t)(n0 / 32 + wn) * nsteps) * (KG * 64) + ks * KGW * 64 + lane;

    uint4 preg[P_IT];
    auto load_patch = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) preg[it] = *(const uint4*)(in + pgoff[it] + chunk * CC);
    };
    auto store_patch = [&](unsigned char* dst) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (plds[it] >= 0) {
                if constexpr (PACK) *(uint2*)(dst + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint2(0, 0) : pack_f16_hi(preg[it]);
                else if constexpr (PACK3) store_piece_packed3(dst + (plds[it] & 0x3fffffff), preg[it], (plds[it] >> 30) != 0);
                else *(uint4*)(dst + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint4(0, 0, 0, 0) : stage_x_piece<T>(preg[it]);
            }
    };
    constexpr int PSF = CONT ? TB - 3 : COBEVT_CONV3_PSTORE_FIRST;   // first tap that carries stores of the next chunk's patch
    constexpr int PSN = TB - PSF;                                    // taps that carry stores
    auto store_patch_part = [&](unsigned char* dst, int part) {      // one share of the pieces (part = 0 .. PSN - 1, compile-time after unrolling)
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (it % PSN == part && plds[it] >= 0) {
                if constexpr (PACK) *(uint2*)(dst + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint2(0, 0) : pack_f16_hi(preg[it]);
                else if constexpr (PACK3) store_piece_packed3(dst + (plds[it] & 0x3fffffff), preg[it], (plds[it] >> 30) != 0);
                else *(uint4*)(dst + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint4(0, 0, 0, 0) : stage_x_piece<T>(preg[it]);
            }
    };
    auto load_b = [&](uint4 (&b)[KGW], int step) {
#pragma unroll
        for (int g = 0; g < KGW; ++g) b[g] = wq[(size_t)step * (KG * 64) + g * 64];
    };
    // S = 2: fill the single patch buffer for one channel chunk, eight 16-byte pieces in flight per thread at a time
    auto fill_patch_s2 = [&](int chunk) {
        constexpr int BATCH = 8;
        for (int base = 0; base < PATCH_ITEMS; base += BATCH * NT) {
            uint4 v[BATCH];
            int dst[BATCH];
#pragma unroll
            for (int u = 0; u < BATCH; ++u) {
                const int item = base + u * NT + tid;
                dst[u] = -1;
                v[u] = make_uint4(0, 0, 0, 0);
                if (item < PATCH_ITEMS) {
                    const int st = item / STRIP_ITEMS, r = item - st * STRIP_ITEMS;
                    const int pix = r / PIECES, j = r - pix * PIECES;
                    const int py = pix / PW, px = pix - py * PW;
                    dst[u] = st * STRIP + py * PROW + (px & 1) * C::PLANE + (px >> 1) * PSTR + (PACK ? packed_piece_offset(j) : PACK3 ? packed3_piece_offset(j) : j * 16);
                    const int4 sc = *(const int4*)&stab[st * 4];
                    if (sc.w) {
                        const int img = sc.x, sy = sc.y, sx = sc.z;
                        const int vy = sy * 4 - 1 + py, vx = sx * 32 - 1 + px;
                        if (vy >= 0 && vy < p.H && vx >= 0 && vx < p.W)
                            v[u] = *(const uint4*)(in + ((img * p.H + vy) * p.W + vx) * p.Cin + j * CH + chunk * CC);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < BATCH; ++u)
                if (dst[u] >= 0) {                                                        // split(0) = 0: the padding stays zero
                    if constexpr (PACK) *(uint2*)(patch + dst[u]) = pack_f16_hi(v[u]);
                    else if constexpr (PACK3) store_piece_packed3(patch + dst[u], v[u], false);
                    else *(uint4*)(patch + dst[u]) = stage_x_piece<T>(v[u]);
                }
        }
    };

    f32x16 acc[MT];
#pragma unroll
    for (int a = 0; a < MT; ++a)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;

    const int abase = (ql >> 4) * S * PROW + (ql & 15) * PSTR + (PACK3 ? h * 32 + ks * KGA * 64 : h * 16 + ks * KGA * 32);

    uint4 bq[R][KGW];
    COBEVT_TRACE_MARK(0);
    if (S == 1) load_patch(0);
#pragma unroll
    for (int s = 0; s < PF; ++s) load_b(bq[s], s < nsteps ? s : nsteps - 1);
    // The KS k-split partials of the accumulators meet in the fp32 staging tile [MT*32 pixels][BN] (it reuses the patch memory).  With the
    // operands swapped a lane holds, per accumulator tile, one pixel (lane & 31) and four runs of four consecutive couts
    // (8k + 4*(lane>>5) + 0..3), i.e. 16-byte staging accesses instead of sixteen scalar ones.  The KS rounds rotate over
    // the accumulator tiles - in round r the waves of k-split ks handle the tiles a with (a - r) mod KS == ks, storing
    // partial + bias in round 0 and adding into the tile afterwards - so all eight waves work in every round (with
    // "split kk stores / adds its whole accumulator in round kk" four, then two, of the eight idled: 4.3k-5.8k cycles).
    float* stage = (float*)smem;
    constexpr int SROW = C::SSTR / 4;
    auto reduce_ksplit = [&](const float* bias_lds, bool trace) {
        const int c0 = wn * 32 + 4 * h;
        float4 bias4[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) bias4[k] = *(const float4*)&bias_lds[c0 + 8 * k];
        if (trace) COBEVT_TRACE_MARK(57);
#pragma unroll
        for (int r = 0; r < KS; ++r) {
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if (((a + KS - r) % KS) != ks) continue;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    float4* d = (float4*)(stage + (a * 32 + ql) * SROW + c0 + 8 * k);
                    float4 v = make_float4(acc[a][4 * k], acc[a][4 * k + 1], acc[a][4 * k + 2], acc[a][4 * k + 3]);
                    if (r == 0) {
                        v.x += bias4[k].x; v.y += bias4[k].y; v.z += bias4[k].z; v.w += bias4[k].w;
                    } else {
                        const float4 o = *d;
                        v.x += o.x; v.y += o.y; v.z += o.z; v.w += o.w;
                    }
                    *d = v;
                }
            }
            __syncthreads();
            if (trace && r == 0) COBEVT_TRACE_MARK(58);
        }
    };
    if (S == 1) store_patch(patch);
    else fill_patch_s2(0);
    __syncthreads();
    COBEVT_TRACE_MARK(1);
    constexpr bool EARLY_RES = S == 1 && Elem<T>::kIsBf16 && MT <= 5;
    Conv3Store<T, NT, MT * 32, BN> st;
    auto coord = [&](int px, int& im, int& oy, int& ox) {
        const int4 sc = *(const int4*)&stab[(px >> 5) * 4];      // branch-free: one 16-byte LDS read, no dependent waits
        im = sc.x;
        oy = sc.y * 2 + ((px >> 4) & 1); ox = sc.z * 16 + (px & 15);
        return (sc.w != 0) & (oy < p.Ho) & (ox < p.Wo);
    };
    constexpr int PLT = (S == 1 && !HALF) ? (CONT ? PSF - COBEVT_CONV3_CONT_LEAD : COBEVT_CONV3_PLOAD_TAP) : 0;
    uint4 af[RA][MT];                                    // the A-operand ring (CONT: carried across chunks)
    uint4 al[PACK3 ? RA : 1][PACK3 ? MT : 1];            // three-term form: the lo operands of the pair
    auto read_a_from = [&](const unsigned char* pb, int slot, int n) {   // n = group index inside the chunk (compile-time after unrolling)
        const int t2 = n / KGA, g2 = n - t2 * KGA;
        const int kh2 = t2 / 3, kw2 = t2 - kh2 * 3;
        const int toff = S == 1 ? kh2 * PROW + kw2 * PSTR : kh2 * PROW + (kw2 & 1) * C::PLANE + (kw2 >> 1) * PSTR;
        const unsigned char* pn = pb + toff + abase + g2 * (PACK3 ? 64 : 32);
#pragma unroll
        for (int a = 0; a < MT; ++a) {
            af[slot][a] = *(const uint4*)(pn + a * STRIP);
            if constexpr (PACK3) al[slot][a] = *(const uint4*)(pn + a * STRIP + 16);
        }
    };
    if (CONT) {
        read_a_from(patch, 0, 0);
        read_a_from(patch, 1, 1);
        __builtin_amdgcn_sched_barrier(0);
    }
    auto run_chunk = [&](int chunk, auto last_tag) {
        constexpr bool LAST = decltype(last_tag)::value;          // the peeled last chunk: its residual loads ride at tap PLT too
        const bool more = chunk + 1 < nchunk;
        unsigned char* pbuf = patch + (S == 1 && !HALF ? (chunk & 1) * C::PATCH_BYTES : 0);
        unsigned char* pother = patch + (S == 1 && !HALF ? ((chunk & 1) ^ 1) * C::PATCH_BYTES : 0);
        if (S == 1 && PLT == 0 && !(COBEVT_CONV3_KNOCK & 8)) load_patch(more ? chunk + 1 : chunk);   // unconditional (clamped): counted vmcnt
        // A fragments run two k-groups ahead of the MFMAs in a three-slot register ring (9 * KGW groups per chunk, a
        // multiple of 3, so slots are static); sched_group_barrier pins the issue order "one ds_read, one MFMA", i.e. a
        // fragment is requested 2 * MT MFMAs (>= 320 cycles) before its first use.  A wave alone on its SIMD then
        // keeps the MFMA pipe busy (the s_memtime trace showed the younger wave of each SIMD finishing a chunk ~2k
        // cycles after the older one with the A reads only one MFMA ahead of their use).
        constexpr int NG = 9 * KGA;                      // A-operand groups of this wave per chunk
        auto read_a = [&](int slot, int n) { read_a_from(pbuf, slot, n); };
        if (!CONT) {
            read_a(0, 0);
            if (RA == 3) read_a(1, 1);
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int step = chunk * 9 + tap;
            if (chunk < 4) COBEVT_TRACE_MARK(2 + chunk * 9 + tap);
            if (CONT && tap == TB) {                     // the next chunk's patch is complete; nobody reads this one any more (see CONT)
                __syncthreads();
                __builtin_amdgcn_sched_barrier(0);
            }
#ifdef COBEVT_CONV3_SETPRIO
            // issue arbitration favours the older wave of a SIMD; alternate it per tap so both waves of a SIMD reach
            // the chunk barrier together instead of the younger one finishing its taps alone
            if ((tap ^ ks) & 1) __builtin_amdgcn_s_setprio(3); else __builtin_amdgcn_s_setprio(0);
#endif
            if (!(COBEVT_CONV3_KNOCK & 1) || chunk == 0)
                load_b(bq[(tap + PF) % R], step + PF < nsteps ? step + PF : nsteps - 1);
            if (S == 1 && PLT > 0 && tap == PLT) {
                if (LAST) { if (EARLY_RES && p.store_mode == 0) st.prepare(p, tid, n0, coord); }
                else if (!(COBEVT_CONV3_KNOCK & 8)) load_patch(more ? chunk + 1 : chunk);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < KGA; ++g) {
                const int n = tap * KGA + g;
                if (n + RA - 1 < NG) { if (!(COBEVT_CONV3_KNOCK & 4) || chunk == 0) read_a((n + RA - 1) % RA, n + RA - 1); }
                else if (CONT && !LAST && !(COBEVT_CONV3_KNOCK & 4)) read_a_from(pother, (n + RA - 1) % RA, n + RA - 1 - NG);
                uint4 wpk = make_uint4(0, 0, 0, 0), wlo = make_uint4(0, 0, 0, 0);
                if constexpr (PACK) wpk = pack_f16_pair(bq[tap % R][(2 * g) % KGW], bq[tap % R][(2 * g + 1) % KGW]);
                if constexpr (PACK3) split_w_pair(bq[tap % R][(2 * g) % KGW], bq[tap % R][(2 * g + 1) % KGW], wpk, wlo);
#pragma unroll
                for (int a = 0; a < MT; ++a) {
                    if (COBEVT_CONV3_KNOCK & 2) acc[a][0] += __uint_as_float((af[n % RA][a].x ^ bq[tap % R][g].x) & 0x3fffffffu);
                    else if constexpr (PACK) mfma_f16_packed(wpk, af[n % RA][a], acc[a]);
                    else if constexpr (PACK3) mfma_3term(wpk, wlo, af[n % RA][a], al[PACK3 ? n % RA : 0][PACK3 ? a : 0], acc[a]);
                    else mfma_kgroup_xs<T>(bq[tap % R][g], af[n % RA][a], acc[a]);   // D = W X^T: rows = couts, cols = pixels
                }
                if (Elem<T>::kIsBf16 && (n + 2 < NG || (CONT && !LAST))) {
#pragma unroll
                    for (int a = 0; a < MT; ++a) {
                        __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);   // one ds_read ...
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // ... then one MFMA
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the other buffer has been free since the last barrier: write the next chunk's patch under taps 7-8
            if (S == 1 && !HALF && more && !LAST && !(COBEVT_CONV3_KNOCK & 8)) {
                if (COBEVT_CONV3_PSTORE_SPREAD || CONT) {
                    if (tap >= PSF && tap < TB) { store_patch_part(pother, tap - PSF); __builtin_amdgcn_sched_barrier(0); }
                } else if (tap == 6) {
                    store_patch(pother);
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        if (chunk < 4) COBEVT_TRACE_MARK(41 + 2 * chunk);
#ifdef COBEVT_CONV3_TRACE
        if (logical == 0 && lane == 0 && chunk == 0) {       // per-wave barrier arrival + SIMD placement
            cobevt_conv3_trace[49 + wave] = __builtin_readcyclecounter();
            cobevt_conv3_trace[57 + wave > 63 ? 63 : 57 + wave] = 0;
            unsigned hw = __builtin_amdgcn_s_getreg((4 << 0) | (0 << 6) | (31 << 11));   // HW_REG_HW_ID, all 32 bits
            cobevt_conv3_trace[49 + wave] |= (unsigned long long)((hw >> 4) & 3) << 60;
        }
#endif
        if (!CONT) __syncthreads();
        if (HALF && more) {                          // single patch buffer: every wave is done reading it, write the next chunk
            store_patch(pbuf);
            __syncthreads();
        }
        if (chunk < 4) COBEVT_TRACE_MARK(42 + 2 * chunk);
    };
    // the last chunk is peeled so the residual loads of the store pass can be issued (unconditionally, keeping the
    // vmcnt waits counted) one chunk of MFMAs before they are needed
    // (DS) the projection shortcut: x(2 oy, 2 ox) of the workgroup's MT * 32 output pixels, Cin2 / 64 chunks, in a compact LDS image
    // [chunk][pixel][PSTR] over the (by then dead) patch memory.  Items = (strip, pixel 0..31, 16-byte piece); the first two chunks'
    // pieces are requested in front of the last nine-tap chunk (its patch-prefetch registers are free) and written behind it.
    constexpr int XI = MT * 32 * PIECES, X_IT = DS ? (XI + NT - 1) / NT : 1, XCH = MT * 32 * PSTR;
    int xoff[X_IT], xlds[X_IT];
    const T* in2 = (const T*)p.in2;
    uint4 xr[2][X_IT];
    auto load_x = [&](uint4 (&x)[X_IT], int e) {
#pragma unroll
        for (int it = 0; it < X_IT; ++it) x[it] = *(const uint4*)(in2 + xoff[it] + e * CC);
    };
    auto store_x = [&](const uint4 (&x)[X_IT], int e) {
#pragma unroll
        for (int it = 0; it < X_IT; ++it)
            if (xlds[it] >= 0) *(uint4*)(patch + e * XCH + (xlds[it] & 0x3fffffff)) = (xlds[it] >> 30) ? make_uint4(0, 0, 0, 0) : x[it];
    };
    if (S == 1) {
        for (int chunk = 0; chunk + 1 < nchunk; ++chunk) run_chunk(chunk, std::false_type{});
        if constexpr (DS) {
#pragma unroll
            for (int it = 0; it < X_IT; ++it) {
                const int item = tid + it * NT;
                xoff[it] = 0;
                xlds[it] = -1;
                if (item < XI) {
                    const int pi = item / PIECES, j = item - pi * PIECES;          // pi = strip * 32 + pixel of the strip
                    const int4 sc = *(const int4*)&stab[(pi >> 5) * 4];
                    const int oy = sc.y * 2 + ((pi >> 4) & 1), ox = sc.z * 16 + (pi & 15);
                    const bool valid = (sc.w != 0) & (oy < p.Ho) & (ox < p.Wo);
                    xoff[it] = valid ? ((sc.x * p.H2 + 2 * oy) * p.W2 + 2 * ox) * p.Cin2 + j * CH : 0;
                    xlds[it] = (pi * PSTR + j * 16) | (valid ? 0 : (1 << 30));
                }
            }
            load_x(xr[0], 0);
            load_x(xr[1], nchunk2 > 1 ? 1 : 0);
        }
        if (PLT == 0 && EARLY_RES && p.store_mode == 0) st.prepare(p, tid, n0, coord);
        __builtin_amdgcn_sched_barrier(0);           // keep the residual loads up here
        run_chunk(nchunk - 1, std::true_type{});
    } else {
        // single patch buffer: refill between two barriers (run_chunk ends with one); ONE call site keeps the fully
        // unrolled body within the unroller's budget (with two, LLVM left the tap loop rolled and the fragment rings
        // went to scratch: 3x slower)
#pragma unroll 1
        for (int chunk = 0; chunk < nchunk; ++chunk) {
            if (chunk > 0) {
                fill_patch_s2(chunk);
                __syncthreads();
            }
            run_chunk(chunk, std::false_type{});
        }
    }
    if constexpr (DS) {
        // Every wave is past the last chunk barrier: nobody reads the patch any more.  The shortcut is NOT k-split: the waves of k-split ks
        // take the strips a with a % KS == ks and all four k-groups of every chunk, so a wave holds the COMPLETE shortcut sum of its
        // (strip, 32 couts) tiles, rounds it (+ its bias) to the storage type - exactly what the separate launch stores - and adds it to
        // its own partial sum of the 3x3; the epilogue's reduction does the rest.  The result differs from the three-launch block by fp32
        // summation order only (no bf16 parity gate moves); one 16-register accumulator at a time.
        for (int e0 = 0; e0 < nchunk2; e0 += 2) {
            if (e0 > 0) {                                        // (layers with more than two shortcut chunks: exposed loads)
                load_x(xr[0], e0);
                load_x(xr[1], e0 + 1 < nchunk2 ? e0 + 1 : e0);
            }
            store_x(xr[0], e0);
            if (e0 + 1 < nchunk2) store_x(xr[1], e0 + 1);
        }
        __syncthreads();
        {
            const int c0 = wn * 32 + 4 * h;
            float4 bias4[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) bias4[k] = *(const float4*)&sbias2[c0 + 8 * k];
            const uint4* wd = (const uint4*)p.wgt + ((size_t)(n0 / 32 + wn) * nsteps + nchunk * 9) * (KG * 64) + lane;   // all k-groups
            // the shortcut's weight fragments of this wave's cout tile, all (<= 4) chunks, once: the rings of the main loop are dead,
            // the registers are there, and a load in front of every strip's MFMAs would be an exposed L2 round trip each time
            uint4 bw[4][KG];
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (e < nchunk2) {
#pragma unroll
                    for (int g = 0; g < KG; ++g) bw[e][g] = wd[(size_t)e * (KG * 64) + g * 64];
                }
#pragma unroll
            for (int a = 0; a < MT; ++a) {
                if ((a % KS) != ks) continue;
                f32x16 dacc;
#pragma unroll
                for (int r = 0; r < 16; ++r) dacc[r] = 0.f;
                const unsigned char* xa = patch + (a * 32 + ql) * PSTR + h * 16;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (e >= nchunk2) continue;
                    uint4 ax[KG];
#pragma unroll
                    for (int g = 0; g < KG; ++g) ax[g] = *(const uint4*)(xa + e * XCH + g * 32);
#pragma unroll
                    for (int g = 0; g < KG; ++g) mfma_kgroup_xs<T>(bw[e][g], ax[g], dacc);
                }
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const uint32_t lo = pack_bf2(dacc[4 * k] + bias4[k].x, dacc[4 * k + 1] + bias4[k].y);
                    const uint32_t hi = pack_bf2(dacc[4 * k + 2] + bias4[k].z, dacc[4 * k + 3] + bias4[k].w);
                    acc[a][4 * k] += __uint_as_float(lo << 16);
                    acc[a][4 * k + 1] += __uint_as_float(lo & 0xffff0000u);
                    acc[a][4 * k + 2] += __uint_as_float(hi << 16);
                    acc[a][4 * k + 3] += __uint_as_float(hi & 0xffff0000u);
                }
            }
        }
        __syncthreads();                                         // the staging tile of the epilogue reuses this memory
    }
    COBEVT_TRACE_MARK(38);

    // ---- epilogue: the k-split partials meet in the staging tile (reduce_ksplit above), then the store pass
    reduce_ksplit(sbias, true);
    COBEVT_TRACE_MARK(39);
    T* out = (T*)p.out;
    if (p.store_mode == 0) {
        if (!EARLY_RES) st.prepare(p, tid, n0, coord);
        st.finish(p, stage, SROW, tid, n0);
    } else {
        // PixelUnshuffle(2): a strip (2 rows x 16 columns) is one output row of 8 pixels with 4*BN channels of this tile
        const int cpp = BN * 4 / CH;
        for (int item = tid; item < MT * 8 * cpp; item += NT) {
            const int qp = item / cpp, cj = item - qp * cpp;
            const int s = qp >> 3, qx = qp & 7;
            const int4 sc = *(const int4*)&stab[s * 4];
            if (!sc.w) continue;
            const int img = sc.x, sy = sc.y, sx = sc.z;
            const int oy2 = sy, ox2 = sx * 8 + qx;
            if (oy2 >= (p.Ho >> 1) || ox2 >= (p.Wo >> 1)) continue;
            float v[8];
            bool any = false;
#pragma unroll
            for (int e = 0; e < CH; ++e) {
                const int oc = cj * CH + e, c = oc >> 2, qd = oc & 3;
                const int px = s * 32 + (qd >> 1) * 16 + 2 * qx + (qd & 1);
                float x = stage[px * SROW + c];
                x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_t<T>(x) : x);
                v[e] = x;
                any |= (n0 + c) < p.Cout;
            }
            if (!any) continue;
            const size_t o = (((size_t)img * (p.Ho >> 1) + oy2) * (p.Wo >> 1) + ox2) * (size_t)(p.Cout * 4) + (size_t)n0 * 4 + cj * CH;
            if (n0 + ((cj * CH + CH - 1) >> 2) < p.Cout && ((p.Cout * 4) % CH) == 0) *(uint4*)(out + o) = f32_to_chunk<T>(v);
            else for (int e = 0; e < CH; ++e) if (n0 + ((cj * CH + e) >> 2) < p.Cout) store_elem<T>(out, o + e, v[e]);
        }
    }
    COBEVT_TRACE_MARK(40);
    COBEVT_TRACE_RT(1);
}

template <typename T, int MT, int WN, int KS, int S, int NW = 8, bool DS = false>
static int launch_conv3s(Conv3Params p, int coutp, hipStream_t stream) {
    using C = Conv3SCfg<T, MT, WN, KS, S, NW>;
    if ((long)p.N * p.H * p.W * p.Cin >= 0x7fffffffL) return COBEVT_ERR_UNSUPPORTED;   // 32-bit patch offsets
    if (DS && (long)p.N * p.H2 * p.W2 * p.Cin2 >= 0x7fffffffL) return COBEVT_ERR_UNSUPPORTED;
    if (DS && (long)(p.Cin2 / 64) * MT * 32 * C::PSTR > (long)C::LDS_BYTES) return COBEVT_ERR_UNSUPPORTED;   // the shortcut's x image lives in the patch memory
    p.tiles_y = (p.Ho + 1) / 2;
    p.tiles_x = (p.Wo + 15) / 16;
    p.tiles_n = (p.Cout + C::BN - 1) / C::BN;
    if (p.tiles_n * C::BN > coutp) return COBEVT_ERR_SHAPE;
    const long nstrips = (long)p.N * p.tiles_y * p.tiles_x;
    const long blocks = (nstrips + MT - 1) / MT * p.tiles_n;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    constexpr size_t lds = C::LDS_BYTES;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)conv3x3_strips_kernel<T, MT, WN, KS, S, NW, DS>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    hipLaunchKernelGGL((conv3x3_strips_kernel<T, MT, WN, KS, S, NW, DS>), dim3((unsigned)blocks), dim3(C::NT), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

template <typename T, int MT>
static int launch_conv3s_bn(const Conv3Params& p, int coutp, int bn64, int stride, hipStream_t stream) {
    // code 3: 32-cout tiles in four-wave workgroups, two per CU; bf16, stride 1
    if constexpr (Elem<T>::kIsBf16 && MT <= 5) {
        if (bn64 == 3 && stride == 1) return launch_conv3s<T, MT, 1, 4, 1, 4>(p, coutp, stream);
    }
    if (bn64 > 1) return COBEVT_ERR_UNSUPPORTED;
    if (stride == 2) return bn64 ? launch_conv3s<T, MT, 2, 4, 2>(p, coutp, stream) : launch_conv3s<T, MT, 4, 2, 2>(p, coutp, stream);
    return bn64 ? launch_conv3s<T, MT, 2, 4, 1>(p, coutp, stream) : launch_conv3s<T, MT, 4, 2, 1>(p, coutp, stream);
}

template <typename T>
static int dispatch_conv3f(const Conv3Params& p, int kg, int coutp, int variant, int stride, hipStream_t stream) {
    if (kg != 4) return COBEVT_ERR_UNSUPPORTED;
    // variant = 100 + 10*MT + code; code 0 = 128-cout tiles, 1 = 64-cout tiles (8 waves), 3 = 32-cout tiles in four-wave workgroups
    // (bf16, stride 1, MT <= 5); 0 = a safe default
    if (variant == 0) variant = p.Cout <= 64 ? 151 : 150;
    const int mt = (variant - 100) / 10, bn64 = (variant - 100) % 10;
    if (variant < 100 || bn64 > 3) return COBEVT_ERR_ARG;
    if constexpr (Elem<T>::kIsBf16) {                      // one / two strips per four-wave workgroup: tiny grids (a 1-image decoder map), code 3 only
        if (bn64 == 3 && stride == 1 && mt == 1) return launch_conv3s<T, 1, 1, 4, 1, 4>(p, coutp, stream);
        if (bn64 == 3 && stride == 1 && mt == 2) return launch_conv3s<T, 2, 1, 4, 1, 4>(p, coutp, stream);
    }
    switch (mt) {
        case 3: return launch_conv3s_bn<T, 3>(p, coutp, bn64, stride, stream);
        case 4: return launch_conv3s_bn<T, 4>(p, coutp, bn64, stride, stream);
        case 5: return launch_conv3s_bn<T, 5>(p, coutp, bn64, stride, stream);
        case 6: return launch_conv3s_bn<T, 6>(p, coutp, bn64, stride, stream);
        default: return COBEVT_ERR_ARG;
    }
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

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_conv3x3_wfrag_nhwc(const void* in, const void* wfrag, const float* bias, const void* residual, void* out,
                                         const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W, Cin, Cout, upsample, act, store_mode, chunk_channels, padded_cout, variant, stride]
    if (!in || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    Conv3Params p;
    const int dtype = dims[0];
    p.in = in; p.wgt = wfrag; p.bias = bias; p.residual = residual; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cin = dims[4]; p.Cout = dims[5];
    p.upsample = dims[6]; p.act = dims[7]; p.store_mode = dims[8];
    const int cc = dims[9], coutp = dims[10], variant = dims[11], stride = dims[12];
    if (stride != 1 && stride != 2) return COBEVT_ERR_ARG;
    if (stride == 2 && (p.upsample || p.store_mode != 0)) return COBEVT_ERR_UNSUPPORTED;
    p.Ho = stride == 2 ? (p.H - 1) / 2 + 1 : (p.upsample ? 2 * p.H : p.H);
    p.Wo = stride == 2 ? (p.W - 1) / 2 + 1 : (p.upsample ? 2 * p.W : p.W);
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.Cin < 1 || p.Cout < 1 || cc < 1) return COBEVT_ERR_SHAPE;
    if (p.store_mode != 0 && p.store_mode != 1) return COBEVT_ERR_ARG;
    if (p.store_mode == 1 && (p.residual || ((p.Ho | p.Wo) & 1))) return COBEVT_ERR_UNSUPPORTED;
    if (p.Cin % cc != 0 || coutp % 32 != 0 || coutp < p.Cout) return COBEVT_ERR_SHAPE;
    const int kg = cc * (dtype == 0 ? 2 : 4) / 32;
    return dtype == 0 ? dispatch_conv3f<bf16_t>(p, kg, coutp, variant, stride, stream)
                      : dispatch_conv3f<float>(p, kg, coutp, variant, stride, stream);
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_conv3x3_ds_wfrag_nhwc(const void* in, const void* in2, const void* wfrag, const float* bias, const float* bias2, void* out,
                                            const int* dims, hipStream_t stream) {
    // dims: [dtype(0), N, H, W, Cin, Cout, act, padded_cout, variant, H2, W2, Cin2]
    if (!in || !in2 || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    Conv3Params p;
    p.in = in; p.wgt = wfrag; p.bias = bias; p.bias2 = bias2; p.residual = nullptr; p.out = out; p.in2 = in2;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cin = dims[4]; p.Cout = dims[5];
    p.upsample = 0; p.act = dims[6]; p.store_mode = 0;
    const int coutp = dims[7], variant = dims[8];
    p.H2 = dims[9]; p.W2 = dims[10]; p.Cin2 = dims[11];
    p.Ho = p.H; p.Wo = p.W;
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.Cin < 64 || p.Cin % 64 || p.Cout < 1 || p.Cin2 < 64 || p.Cin2 % 64) return COBEVT_ERR_SHAPE;
    if (p.Cin2 > 256) return COBEVT_ERR_UNSUPPORTED;           // at most four shortcut chunks (their weight fragments live in registers)
    if ((p.H2 - 1) / 2 + 1 != p.H || (p.W2 - 1) / 2 + 1 != p.W) return COBEVT_ERR_SHAPE;      // in2 is the stride-2 source of the in / out grid
    if (coutp % 32 != 0 || coutp < p.Cout) return COBEVT_ERR_SHAPE;
    if (p.act < 0 || p.act > 2) return COBEVT_ERR_ARG;
    const int mt = (variant - 100) / 10, bn64 = (variant - 100) % 10;
    if (variant < 100 || bn64 > 1) return COBEVT_ERR_ARG;
    switch (mt * 2 + bn64) {
        case 6: return launch_conv3s<bf16_t, 3, 4, 2, 1, 8, true>(p, coutp, stream);
        case 7: return launch_conv3s<bf16_t, 3, 2, 4, 1, 8, true>(p, coutp, stream);
        case 8: return launch_conv3s<bf16_t, 4, 4, 2, 1, 8, true>(p, coutp, stream);
        case 9: return launch_conv3s<bf16_t, 4, 2, 4, 1, 8, true>(p, coutp, stream);
        case 10: return launch_conv3s<bf16_t, 5, 4, 2, 1, 8, true>(p, coutp, stream);
        case 11: return launch_conv3s<bf16_t, 5, 2, 4, 1, 8, true>(p, coutp, stream);
        default: return COBEVT_ERR_UNSUPPORTED;
    }
}

#ifdef COBEVT_CONV3_TRACE
extern "C" int cobevt_conv3_read_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_conv3_trace), sizeof(unsigned long long) * 64) == hipSuccess ? 0 : 1;
}
extern "C" int cobevt_conv3_read_rt(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_conv3_rt), sizeof(unsigned long long) * 3 * 2048) == hipSuccess ? 0 : 1;
}
#endif
