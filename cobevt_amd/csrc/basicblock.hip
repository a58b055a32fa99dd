// Fused ResNet BasicBlock (stride 1, no downsample) on MFMA (gfx950):
//
//     out = ReLU( conv3x3_2( ReLU( conv3x3_1(x) + b1 ) ) + b2 + x )        (eval BatchNorms folded into the weights / biases)
//
// i.e. torchvision.models.resnet.BasicBlock.forward as reached from opv2v/opencood/models/backbones/resnet_ms.py:67-74
// (layer1 / layer2 of ResNet-34 on the camera images: 64 and 128 channels on 128x128 and 64x64 maps).  As two launches
// of the 3x3 kernel these layers are neither MFMA- nor HBM-bound (40 us each against a 20 / 10 us HBM floor): every launch
// pays a patch prologue, nine barriered taps and an fp32-staged epilogue for 72 MFMAs per wave, and the intermediate map
// (42 MB on the 64-channel level) is written and read back with its halo.  Here a workgroup owns a TH x 16 output tile (TH = 16 / 8):
//   1. the (TH+4) x 20 input patch (2-pixel halo) goes to LDS chunk by chunk (128 bytes of channels);
//   2. conv1 is evaluated on the (TH+2) x 18 region the second conv needs (its pixels linearised into 32-row MFMA
//      tiles), + b1, ReLU, rounded to the storage type exactly as the unfused path stores it, and written to a second
//      LDS patch - zeros where the region leaves the image, which is conv2's zero padding;
//   3. conv2 runs on that patch, + b2 + residual (the block input, re-read coalesced from global), ReLU, 16-byte stores.
// The intermediate never reaches HBM, one launch / prologue / epilogue instead of two, at the price of recomputing the
// halo of conv1 (324 instead of 256 pixels at TH = 16).  Weights come in MFMA fragment order straight from L2 (same table as
// cobevt_conv3x3_wfrag_nhwc), prefetched two taps ahead; D = W . X^T so a lane owns one pixel and runs of four couts
// (8-byte LDS writes for the intermediate, 16-byte for the staging).  8 waves = cout tiles x pixel-tile groups.
#include "common.hpp"

namespace cobevt {

struct BasicBlockParams {
    const void* in;
    const uint4* w1;        // fragment-ordered [C_p/32][C/cc][9][4][64]
    const float* b1;
    const uint4* w2;
    const float* b2;
    void* out;
    int N, H, W;
    int tiles_y, tiles_x;
    int ntiles;             // N * tiles_y * tiles_x; the grid may be smaller (persistent workgroups walk tiles logical, logical + grid, ...)
};

// Persistent workgroups: one per CU slot, walking the tiles logical, logical + grid, ...  A tile used to be a workgroup: load the input
// patch (a cold fetch - the previous kernel wrote it, and L2 is invalidated at kernel boundaries: 3.5-3.9k cycles of the 23k / 37k of a
// tile in the s_memtime trace, tools/bb_trace.py), conv1, conv2, store, exit, and the NEXT workgroup of the CU (LDS admits one) started
// with the same cold fetch.  Now the next tile's patch is requested right behind conv2's last weight fragments and lands under the
// epilogue (stage write, barrier, store pass: ~4k cycles); only the LDS stores of it remain in front of the next conv1.
#ifndef COBEVT_BB_PERSIST
#define COBEVT_BB_PERSIST 1
#endif

template <typename T, int C, int TH_> struct BBCfg {
    static constexpr int NT = 512;
    static constexpr int EB = Elem<T>::kBytes, CH = Elem<T>::kChunk;
    static constexpr int CC = 128 / EB;                  // channels per 128-byte chunk
    static constexpr int NCH = C / CC;                   // channel chunks
    static constexpr int NCT = C / 32;                   // 32-cout tiles
    static constexpr int NPG = 8 / NCT;                  // waves sharing one cout tile
    static constexpr int TH = TH_, TW = 16;
    static constexpr int R1H = TH + 2, R1W = TW + 2, R1 = R1H * R1W;   // conv1 region: (TH + 2) x 18 pixels
    static constexpr int N1 = (R1 + 31) / 32;            // MFMA pixel tiles of the region (6 for TH = 8, 11 for 16)
    static constexpr int T1W = (N1 + NPG - 1) / NPG;     // conv1 pixel tiles per wave
    static constexpr int N2 = TH * TW / 32;              // MFMA pixel tiles of the output
    static constexpr int T2W = N2 / NPG;                 // conv2 pixel tiles per wave
    static constexpr int P1H = TH + 4, P1W = TW + 4;     // input patch (TH + 4) x 20
    static constexpr int PSTR1 = 128 + 16;
    // LDS row pitches chosen for conflict-free ds_read_b128 A fragments (16-byte slots, 16 per 256-byte bank row; a pixel
    // advances 9 slots in patch1 and (C*EB/16 + 1), an odd number, in patch2):
    //  * conv1 reads 32 CONSECUTIVE region pixels p = 18*ry + rx: with a pitch of 194 slots (== 18*9 mod 16) the slot of a
    //    region pixel is 9*p + const for every p, i.e. it keeps advancing across the row wrap -> any 16 lanes of a read
    //    group hit 16 distinct slots (a 180-slot pitch measured 44 % of the LDS-active cycles as bank conflicts);
    //  * conv2 reads 2 rows x 16 columns: pitch == 0 mod 16 slots, the two rows cover complementary slots.
    static constexpr int PITCH1 = 194 * 16;                                        // >= P1W * PSTR1 = 2880
    static constexpr int PATCH1 = P1H * PITCH1;
    static constexpr int PSTR2 = C * EB + 16;
    static constexpr int PITCH2 = (R1W * PSTR2 + 255) / 256 * 256;
    static constexpr int PATCH2 = R1H * PITCH2;
    static_assert(PITCH1 >= P1W * PSTR1, "patch1 pitch");
    static constexpr int SSTR = C * 4 + 16;
    static constexpr int STAGE = TH * TW * SSTR;
    static constexpr int MAIN = PATCH1 + PATCH2;
    static constexpr int LDS = MAIN > STAGE ? MAIN : STAGE;
    static_assert(NCT >= 2 && NCT <= 8 && 8 % NCT == 0 && N2 % NPG == 0, "wave layout");
    // persistent workgroups (see above): the bf16 64-channel 16-row shape only.  Same-job kernel traces of both builds inside the frame
    // (gpurun_out/r03ad): 64 channels 69.47 -> 67.93 us per launch, 128 channels 69.22 -> 70.87 us (slower: its second chunk's refill and
    // the longer conv2 leave less to hide, and 231 registers instead of 195), whole frame 557.7 / 558.1 -> 559.3 / 559.1 frames/s, i.e.
    // nothing; the fp32 parity mode and the two-per-CU 8-row tile spill registers in the tile loop.
    static constexpr bool PERSIST = COBEVT_BB_PERSIST && Elem<T>::kIsBf16 && C == 64 && TH_ == 16;
};

// One 128-byte channel chunk of a 3x3 convolution out of an LDS patch: 9 taps x 4 k-groups, NTW pixel tiles per wave.
// The weight fragments of tap + 2 are requested at the start of every tap (register ring bq, 9 % 3 == 0 keeps the slots
// static) and the A fragments run two k-groups ahead of the MFMAs in a second ring, with the issue order pinned
// ("one ds_read, one MFMA"): with two waves per SIMD an LDS round trip in front of every MFMA is the whole runtime.
template <typename T, int NTW>
__device__ __forceinline__ void bb_conv_chunk(const unsigned char* patch, const int (&aoff)[NTW], const bool (&ok)[NTW],
                                              int row_pitch, int pstr, const uint4* wsrc, int step0, int nstep,
                                              uint4 (&bq)[3][4], f32x16 (&acc)[NTW], int odd_off = -1) {
    constexpr bool PACK = kXPack<T, 4>;                         // third library: one fp16 per activation, an A operand spans two k-groups (common.hpp)
    constexpr bool PACK3 = kXPack3<T, 4>;                       // second library: the three-term form on k-group pairs (hi and lo operands, two ring slots)
    constexpr int KGA = (PACK || PACK3) ? 2 : 4, NG = 9 * KGA, RA = PACK3 ? 2 : 3;
    uint4 af[RA][NTW];
    uint4 al[PACK3 ? RA : 1][PACK3 ? NTW : 1];
    auto read_a = [&](int slot, int n) {                        // n = tap * KGA + operand group (compile-time after unrolling)
        const int tap = n / KGA, g = n % KGA;
        // odd_off >= 0: a stride-2 convolution out of a patch whose rows are de-interleaved by column parity ([even columns][odd
        // columns]; the lane base steps two patch rows / one plane pixel per output pixel): tap column 0 / 1 / 2 = even plane,
        // odd plane, even plane + 1 pixel
        const int kw = tap % 3;
        const int off = (tap / 3) * row_pitch + (odd_off < 0 ? kw * pstr : (kw == 1 ? odd_off : (kw == 2 ? pstr : 0))) + g * (PACK3 ? 64 : 32);      // row_pitch in bytes
#pragma unroll
        for (int t = 0; t < NTW; ++t)
            if (ok[t]) {
                af[slot][t] = *(const uint4*)(patch + aoff[t] + off);
                if constexpr (PACK3) al[slot][t] = *(const uint4*)(patch + aoff[t] + off + 16);
            }
    };
    read_a(0, 0);
    if (RA == 3) read_a(1, 1);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int tap = 0; tap < 9; ++tap) {
        {
            const int step = step0 + tap + 2;
            const uint4* src = wsrc + (size_t)(step < nstep ? step : nstep - 1) * 256;
#pragma unroll
            for (int g = 0; g < 4; ++g) bq[(tap + 2) % 3][g] = src[g * 64];
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < KGA; ++g) {
            const int n = tap * KGA + g;
            if (n + RA - 1 < NG) read_a((n + RA - 1) % RA, n + RA - 1);
            uint4 wpk = make_uint4(0, 0, 0, 0), wlo = make_uint4(0, 0, 0, 0);
            if constexpr (PACK) wpk = pack_f16_pair(bq[tap % 3][(2 * g) & 3], bq[tap % 3][(2 * g + 1) & 3]);
            if constexpr (PACK3) split_w_pair(bq[tap % 3][(2 * g) & 3], bq[tap % 3][(2 * g + 1) & 3], wpk, wlo);
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                if (ok[t]) {
                    if constexpr (PACK) mfma_f16_packed(wpk, af[n % RA][t], acc[t]);
                    else if constexpr (PACK3) mfma_3term(wpk, wlo, af[n % RA][t], al[PACK3 ? n % RA : 0][PACK3 ? t : 0], acc[t]);
                    else mfma_kgroup_xs<T>(bq[tap % 3][g], af[n % RA][t], acc[t]);   // D = W . X^T: lane <-> pixel, registers <-> couts
                }
            if (Elem<T>::kIsBf16 && n + 2 < NG) {
#pragma unroll
                for (int t = 0; t < NTW; ++t) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

#ifdef COBEVT_BB_TRACE       // tools/bb_trace.py builds a copy of this file with s_memtime marks (never the product .so)
__device__ unsigned long long cobevt_bb_trace[32];
#define COBEVT_BB_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) cobevt_bb_trace[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define COBEVT_BB_MARK(i) do {} while (0)
#endif

template <typename T, int C, int TH_>
__global__ __launch_bounds__(512, (C == 64 && TH_ == 8 && Elem<T>::kIsBf16) ? 4 : (TH_ == 12 ? 1 : 2)) void basicblock_kernel(BasicBlockParams p) {
    using G = BBCfg<T, C, TH_>;
    constexpr int NT = G::NT, CH = G::CH, CC = G::CC, NCH = G::NCH, NCT = G::NCT, NPG = G::NPG;
    constexpr int TH = G::TH, TW = G::TW, R1W = G::R1W, R1 = G::R1, N1 = G::N1, T1W = G::T1W, T2W = G::T2W;
    constexpr int P1W = G::P1W, PSTR1 = G::PSTR1, PSTR2 = G::PSTR2;
    constexpr int PIECES = 8;                                   // 16-byte pieces per pixel of a 128-byte chunk
    constexpr int P1_ITEMS = G::P1H * P1W * PIECES;             // 1920
    constexpr int P_IT = (P1_ITEMS + NT - 1) / NT;              // 4
    constexpr int NSTEP = NCH * 9;
    constexpr int R = 3;

    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch1 = smem;
    unsigned char* patch2 = smem + G::PATCH1;

    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    int tile = logical, img, oy0, ox0;
    auto set_tile = [&](int t) {
        const int tx = t % p.tiles_x, ty = (t / p.tiles_x) % p.tiles_y;
        img = t / (p.tiles_x * p.tiles_y);
        oy0 = ty * TH;
        ox0 = tx * TW;
    };
    set_tile(tile);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int ct = wave % NCT, pg = wave / NCT;                 // cout tile, pixel-tile group
    const T* in = (const T*)p.in;

    // ---- input patch addressing (12 x 20 pixels, origin (oy0 - 2, ox0 - 2)), once per tile
    int pgoff[P_IT], plds[P_IT];
    auto patch_addr = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int item = tid + it * NT;
            pgoff[it] = 0;
            plds[it] = -1;
            if (item < P1_ITEMS) {
                const int pix = item / PIECES, j = item - pix * PIECES;
                const int py = pix / P1W, px = pix - py * P1W;
                const int lds = py * G::PITCH1 + px * PSTR1 + (kXPack<T, 4> ? packed_piece_offset(j) : kXPack3<T, 4> ? packed3_piece_offset(j) : j * 16);
                const int iy = oy0 - 2 + py, ix = ox0 - 2 + px;
                const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                if (inside) pgoff[it] = ((img * p.H + iy) * p.W + ix) * C + j * CH;
                plds[it] = inside ? lds : (lds | (1 << 30));
            }
        }
    };
    patch_addr();
    uint4 preg[P_IT];
    auto load_patch = [&](int chunk) {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) preg[it] = *(const uint4*)(in + pgoff[it] + chunk * CC);
        __builtin_amdgcn_sched_barrier(0);
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (plds[it] >= 0)
            {
                if constexpr (kXPack<T, 4>) *(uint2*)(patch1 + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint2(0, 0) : pack_f16_hi(preg[it]);
                else if constexpr (kXPack3<T, 4>) store_piece_packed3(patch1 + (plds[it] & 0x3fffffff), preg[it], (plds[it] >> 30) != 0);
                else *(uint4*)(patch1 + (plds[it] & 0x3fffffff)) = (plds[it] >> 30) ? make_uint4(0, 0, 0, 0) : stage_x_piece<T>(preg[it]);
            }
    };
    // weight fragments of this wave's cout tile: step s = chunk * 9 + tap -> 4 k-groups x 64 lanes
    uint4 bq[R][4];
    auto load_b = [&](uint4 (&b)[4], const uint4* wsrc, int step) {        // wsrc = this wave's cout tile, this lane
        const uint4* src = wsrc + (size_t)(step < NSTEP ? step : NSTEP - 1) * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };

    // ---- conv1 on the 10 x 18 region: pixel p = 32 * tile + ql of the row-major region, tiles pg, pg + NPG, ...
    int a1[T1W];                     // LDS offset of this lane's region pixel inside patch1 (tap (0,0), k-group 0)
    bool t1_ok[T1W];
#pragma unroll
    for (int t = 0; t < T1W; ++t) {
        const int tile = pg + t * NPG;
        t1_ok[t] = tile < N1;                                   // wave-uniform
        int pr = tile * 32 + ql;
        if (pr >= R1) pr = R1 - 1;                              // padding lanes of the last tile compute a duplicate
        const int ry = pr / R1W, rx = pr - ry * R1W;
        a1[t] = ry * G::PITCH1 + rx * PSTR1 + h * (kXPack3<T, 4> ? 32 : 16);
    }
    f32x16 acc1[T1W];

    const uint4* w1base = p.w1 + (size_t)ct * NSTEP * 256 + lane;
    const uint4* w2base = p.w2 + (size_t)ct * NSTEP * 256 + lane;
    const uint4* w1src = w1base;
    const uint4* w2src = w2base;
    COBEVT_BB_MARK(0);
    load_patch(0);
    load_b(bq[0], w1src, 0);
    load_b(bq[1], w1src, 1);
    // both folded-BN biases into LDS now (visible after the chunk loop's barriers): neither epilogue starts with a global
    // round trip
    __shared__ __attribute__((aligned(16))) float sbias[2 * C];
    if (tid < 2 * C) {
        const float* b = tid < C ? p.b1 : p.b2;
        sbias[tid] = b ? b[tid < C ? tid : tid - C] : 0.f;
    }
    // conv2's pixel tiles: 2 rows x 16 columns each (tile independent)
    int a2[T2W];
    bool t2_ok[T2W];
#pragma unroll
    for (int t = 0; t < T2W; ++t) {
        const int tl = pg + t * NPG;
        a2[t] = (tl * 2 + (ql >> 4)) * G::PITCH2 + (ql & 15) * PSTR2 + h * (kXPack3<T, 4> ? 32 : 16);
        t2_ok[t] = true;
    }
#pragma unroll 1
  for (;;) {                                                    // ---- one tile per iteration
    // the weight pointers through an opaque copy per iteration: as loop invariants LLVM hoists all 2 x 36 per-step fragment addresses
    // out of the tile loop (144 VGPRs of 64-bit pointers -> 256 VGPRs + scratch spills instead of 153)
    w1src = w1base;
    w2src = w2base;
    asm volatile("" : "+v"(w1src), "+v"(w2src));
#pragma unroll
    for (int t = 0; t < T1W; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk) {
        if (chunk > 0) __syncthreads();                         // every wave finished reading the previous chunk
        store_patch();
        __syncthreads();
        if (chunk + 1 < NCH) load_patch(chunk + 1);
        COBEVT_BB_MARK(1 + 2 * chunk);
        bb_conv_chunk<T, T1W>(patch1, a1, t1_ok, G::PITCH1, PSTR1, w1src, chunk * 9, NSTEP, bq, acc1);
        COBEVT_BB_MARK(2 + 2 * chunk);
    }
    // conv2's first fragments while the intermediate is written (the ring slots 0 / 1 are free: NSTEP % 3 == 0)
    load_b(bq[0], w2src, 0);
    load_b(bq[1], w2src, 1);
    // ---- intermediate: ReLU(conv1 + b1) rounded to T -> patch2 [region pixel][C] ; zeros outside the image
    {
        const int c0 = ct * 32 + 4 * h;
        float4 bias[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) bias[k] = *(const float4*)&sbias[c0 + 8 * k];
#pragma unroll
        for (int t = 0; t < T1W; ++t) {
            const int pr = (pg + t * NPG) * 32 + ql;
            if (!t1_ok[t] || pr >= R1) continue;
            const int ry = pr / R1W, rx = pr - ry * R1W;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[4] = {acc1[t][4 * k] + bias[k].x, acc1[t][4 * k + 1] + bias[k].y, acc1[t][4 * k + 2] + bias[k].z,
                              acc1[t][4 * k + 3] + bias[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e], 0.f) : 0.f;
                unsigned char* d = patch2 + ry * G::PITCH2 + rx * PSTR2 + (c0 + 8 * k) * Elem<T>::kBytes;
                if constexpr (Elem<T>::kIsBf16) *(uint2*)d = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else if constexpr (kXPack<T, 4>)        // channels c0 + 8k .. +3 = piece 2k + h of 128-byte chunk ct: its 8 bytes of the chunk's packed image
                    *(uint2*)(patch2 + ry * G::PITCH2 + rx * PSTR2 + ct * 128 + packed_piece_offset(2 * k + h)) = make_uint2(pack_h2(v[0], v[1]), pack_h2(v[2], v[3]));
                else if constexpr (kXPack3<T, 4>)
                    store_piece_packed3(patch2 + ry * G::PITCH2 + rx * PSTR2 + ct * 128 + packed3_piece_offset(2 * k + h),
                                        make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])), false);
                else *(uint4*)d = stage_x_piece<T>(make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
            }
        }
    }
    // residual = the block input at the output pixels, 16-byte coalesced, in flight under conv2
    constexpr int CPP = C / CH;
    constexpr int S_ITEMS = TH * TW * CPP;
    constexpr int S_IT = S_ITEMS / NT;
    static_assert(S_ITEMS % NT == 0, "store pass");
    int soff[S_IT];
    uint4 rres[S_IT];
#pragma unroll
    for (int i = 0; i < S_IT; ++i) {
        const int item = tid + i * NT;
        const int px = item / CPP, cj = item - px * CPP;
        const int oy = oy0 + px / TW, ox = ox0 + (px % TW);
        const bool ok = (oy < p.H) & (ox < p.W);
        soff[i] = ok ? ((img * p.H + oy) * p.W + ox) * C + cj * CH : -1;
        rres[i] = *(const uint4*)(in + (ok ? soff[i] : 0));      // unconditional (clamped): no vmcnt(0) after each load
    }
    __builtin_amdgcn_sched_barrier(0);
    COBEVT_BB_MARK(8);
    __syncthreads();                                            // patch2 complete
    COBEVT_BB_MARK(9);

    // ---- conv2 on the 8 x 16 tile: output pixel tiles pg, pg + NPG, ...
    f32x16 acc[T2W];
#pragma unroll
    for (int t = 0; t < T2W; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
#pragma unroll 1
    for (int chunk = 0; chunk < NCH; ++chunk)
        bb_conv_chunk<T, T2W>(patch2 + chunk * 128, a2, t2_ok, G::PITCH2, PSTR2, w2src, chunk * 9, NSTEP, bq, acc);
    COBEVT_BB_MARK(10);
    // the next tile's patch: requested now (every weight fragment of this tile has been requested, so nothing queues behind these
    // cold loads in the in-order vmcnt), it lands under the epilogue
    const int next = tile + (int)gridDim.x;
    const bool has_next = G::PERSIST && next < p.ntiles;
    if (has_next) {
        set_tile(next);
        patch_addr();
        load_patch(0);
    }
    __syncthreads();                                            // patch2 no longer read: stage over the patches
    COBEVT_BB_MARK(11);

    // ---- epilogue: fp32 staging [128 pixels][C], then + residual, ReLU, 16-byte stores
    float* stage = (float*)smem;
    constexpr int SROW = G::SSTR / 4;
    {
        const int c0 = ct * 32 + 4 * h;
#pragma unroll
        for (int t = 0; t < T2W; ++t) {
            const int px = (pg + t * NPG) * 32 + ql;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 b = *(const float4*)&sbias[C + c0 + 8 * k];
                *(float4*)(stage + px * SROW + c0 + 8 * k) =
                    make_float4(acc[t][4 * k] + b.x, acc[t][4 * k + 1] + b.y, acc[t][4 * k + 2] + b.z, acc[t][4 * k + 3] + b.w);
            }
        }
    }
    __syncthreads();
    COBEVT_BB_MARK(12);
    T* out = (T*)p.out;
    float sv[S_IT][CH];                                          // every staging read of the thread in flight at once
#pragma unroll
    for (int i = 0; i < S_IT; ++i) {
        const int item = tid + i * NT;
        const int px = item / CPP, cj = item - px * CPP;
#pragma unroll
        for (int e = 0; e < CH; ++e) sv[i][e] = stage[px * SROW + cj * CH + e];
    }
    if (has_next) __syncthreads();                              // the staging tile has been read: the next patch may overwrite it
#pragma unroll
    for (int i = 0; i < S_IT; ++i) {
        if (soff[i] < 0) continue;
        float v[8], rv[8];
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = sv[i][e];
        chunk_to_f32<T>(rres[i], rv);
#pragma unroll
        for (int e = 0; e < CH; ++e) v[e] = fmaxf(v[e] + rv[e], 0.f);
        *(uint4*)(out + soff[i]) = f32_to_chunk<T>(v);
    }
    COBEVT_BB_MARK(13);
    if (!has_next) break;
    tile = next;
    load_b(bq[0], w1src, 0);
    load_b(bq[1], w1src, 1);
  }
}

template <typename T, int C, int TH_>
static int launch_basicblock(BasicBlockParams p, hipStream_t stream) {
    using G = BBCfg<T, C, TH_>;
    p.tiles_y = (p.H + G::TH - 1) / G::TH;
    p.tiles_x = (p.W + G::TW - 1) / G::TW;
    long blocks = (long)p.N * p.tiles_y * p.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    p.ntiles = (int)blocks;
    static cobevt::PerDeviceOnce attr_once;
    static int cu_slots[16];
    int dev = 0;
    (void)hipGetDevice(&dev);
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)basicblock_kernel<T, C, TH_>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        const int per_cu = (160 * 1024) / (G::LDS + 2048) < 1 ? 1 : (160 * 1024) / (G::LDS + 2048);     // workgroups of this kernel a CU holds
        cu_slots[dev & 15] = cus * (per_cu > 2 ? 2 : per_cu);
    }
    if (G::PERSIST && cu_slots[dev & 15] > 0 && blocks > cu_slots[dev & 15]) blocks = cu_slots[dev & 15];
    hipLaunchKernelGGL((basicblock_kernel<T, C, TH_>), dim3((unsigned)blocks), dim3(G::NT), G::LDS, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// The first BasicBlock of ResNet layer2 (stride 2, 64 -> 128 channels, projection shortcut; torchvision BasicBlock + downsample as
// reached from resnet_ms.py:67-74) in one launch (bf16):
//
//     out = ReLU( conv3x3_2( ReLU( conv3x3_1/s2(x) + b1 ) ) + b2 + conv1x1/s2(x) + b_ds )
//
// As three launches it cost 35 + 40 + 17 us per 5-agent frame (the stride-2 conv with a single 64-channel chunk is all prologue and
// epilogue: 0.13 of the matrix peak; the shortcut map is written and read back).  Same scheme as basicblock_kernel with a 4 x 16
// output tile: the (2 * 6 + 1) x 37 input patch in LDS with its rows de-interleaved by column parity, so a tap's A fragments are 32
// consecutive region pixels at the 144-byte pitch again; conv1 on the 6 x 18 region conv2 needs -> + b1, ReLU, bf16 -> second patch;
// conv2 out of that patch; the shortcut is four more k-groups into conv2's accumulators, read at the centre taps of the first patch
// (x[2 oy][2 ox]) - it is never rounded to bf16 on its own; epilogue straight from the registers (D = W X^T: a lane owns one pixel
// and runs of couts; one v_permlane32_swap per register pairs them into 16-byte stores).  1280 workgroups on the 20 x 64 x 64 maps =
// five per CU.
struct DsBlockParams {
    const void* in;
    const uint4* w1;        // fragment-ordered [4][1][9][4][64]   conv1: 64 -> 128, stride 2
    const float* b1;
    const uint4* w2;        // fragment-ordered [4][2][9][4][64]   conv2: 128 -> 128
    const float* b2;
    const uint4* wds;       // dense-row fragments [4 tiles][8 k-groups][64] of the 1x1 / stride-2 shortcut (K = 64 padded to 128)
    const float* bds;
    void* out;
    int N, H, W;            // input map (even sides)
    int Ho, Wo;
    int tiles_y, tiles_x;
};

struct DsCfg {
    static constexpr int TH = 4, TW = 16, R1H = TH + 2, R1W = TW + 2, R1 = R1H * R1W;      // conv1 region 6 x 18 = 108 pixels
    static constexpr int N1 = (R1 + 31) / 32, NPG = 2, T1W = N1 / NPG;                      // 4 pixel tiles, 2 per wave
    static constexpr int P1H = 2 * R1H + 1, P1W = 2 * R1W + 1;                              // 13 x 37 input pixels
    static constexpr int PSTR1 = 128 + 16;
    static constexpr int ODD = ((P1W + 1) / 2) * PSTR1;                                     // odd-column plane behind the 19 even columns
    // pitch: >= 37 * 144 and == 16 * (8 k + 1), so that two patch rows (one region row) advance the 16-byte slot by 2 = 18 * 9 mod 16:
    // the slot of region pixel p is 9 p + const across the row wrap -> conflict-free ds_read_b128 for any 32 consecutive pixels
    static constexpr int PITCH1 = 337 * 16;
    static constexpr int PATCH1 = P1H * PITCH1;                                             // 70,096 B
    static constexpr int PSTR2 = 256 + 16;
    static constexpr int PITCH2 = (R1W * PSTR2 + 255) / 256 * 256;
    static constexpr int PATCH2 = R1H * PITCH2;                                             // 30,720 B, over the first patch once conv1 is done
    static constexpr int XCH = PATCH2;                                                      // partial-sum exchange behind it: 8 waves x 4 KB
    static constexpr int SC = PATCH1;                                                       // the shortcut's pixels x[2 oy][2 ox]: [64][144 B]
    static constexpr int LDS = PATCH1 + TH * TW * PSTR1;                                    // 79,312 B (+ 1 KB of biases): two workgroups per CU
    static_assert(PITCH1 >= P1W * PSTR1 && N1 % NPG == 0 && XCH + 8 * 4096 <= PATCH1, "geometry");
};

__global__ __launch_bounds__(512, 2) void dsblock_kernel(DsBlockParams p) {
    using G = DsCfg;
    using T = bf16_t;
    constexpr int NT = 512, PIECES = 8;
    constexpr int P1_ITEMS = G::P1H * G::P1W * PIECES;          // 3848
    constexpr int P_IT = (P1_ITEMS + NT - 1) / NT;              // 8
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch1 = smem;
    unsigned char* patch2 = smem;

    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tx_ = logical % p.tiles_x, ty_ = (logical / p.tiles_x) % p.tiles_y, img = logical / (p.tiles_x * p.tiles_y);
    const int oy0 = ty_ * G::TH, ox0 = tx_ * G::TW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int ct = wave & 3, pg = wave >> 2;                    // cout tile; conv1: pixel-tile group, conv2: half of the reduction
    const T* in = (const T*)p.in;

    // ---- input patch: rows 2 oy0 - 3 .., columns 2 ox0 - 3 .. (all loads of a thread in flight together, zeros outside the image)
    {
        uint4 preg[P_IT];
        int plds[P_IT];
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int item = tid + it * NT;
            plds[it] = -1;
            int goff = 0;
            if (item < P1_ITEMS) {
                const int pix = item / PIECES, j = item - pix * PIECES;
                const int py = pix / G::P1W, pc = pix - py * G::P1W;
                const int iy = 2 * oy0 - 3 + py, ix = 2 * ox0 - 3 + pc;
                const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const int lds = py * G::PITCH1 + (pc & 1) * G::ODD + (pc >> 1) * G::PSTR1 + j * 16;
                if (inside) goff = ((img * p.H + iy) * p.W + ix) * 64 + j * 8;
                plds[it] = inside ? lds : (lds | (1 << 30));
            }
            preg[it] = *(const uint4*)(in + goff);              // unconditional (element 0 for the padding items)
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (plds[it] >= 0) {
                const int lds = plds[it] & 0x3fffffff;
                const uint4 v = (plds[it] >> 30) ? make_uint4(0, 0, 0, 0) : preg[it];
                *(uint4*)(patch1 + lds) = v;
                // the centre taps x[2 oy][2 ox] once more, compactly, for the shortcut after the patch has been overwritten
                const int py = lds / G::PITCH1, rem = lds - py * G::PITCH1;
                if ((py & 1) && py >= 3 && py <= 2 * G::TH + 1 && rem >= G::ODD + G::PSTR1 && rem < G::ODD + (G::TW + 1) * G::PSTR1) {
                    const int tx = (rem - G::ODD) / G::PSTR1 - 1, jj = (rem - G::ODD) % G::PSTR1;
                    *(uint4*)(smem + G::SC + (((py - 3) >> 1) * G::TW + tx) * G::PSTR1 + jj) = v;
                }
            }
    }
    uint4 bq[3][4];
    auto load_b = [&](uint4 (&b)[4], const uint4* wsrc, int step, int nstep) {
        const uint4* src = wsrc + (size_t)(step < nstep ? step : nstep - 1) * 256;
#pragma unroll
        for (int g = 0; g < 4; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    const uint4* w1src = p.w1 + (size_t)ct * 9 * 256 + lane;
    const uint4* w2src = p.w2 + (size_t)ct * 18 * 256 + lane;
    load_b(bq[0], w1src, 0, 9);
    load_b(bq[1], w1src, 1, 9);
    __shared__ __attribute__((aligned(16))) float sbias[256];   // b1 | b2 + b_ds
    if (tid < 256) sbias[tid] = tid < 128 ? (p.b1 ? p.b1[tid] : 0.f) : ((p.b2 ? p.b2[tid - 128] : 0.f) + (p.bds ? p.bds[tid - 128] : 0.f));
    __syncthreads();

    // ---- conv1 (stride 2) on the 6 x 18 region: region pixel 32 * tile + ql, tiles pg, pg + 2
    f32x16 acc1[G::T1W];
    {
        int a1[G::T1W];
        bool ok1[G::T1W];
#pragma unroll
        for (int t = 0; t < G::T1W; ++t) {
            int pr = (pg + t * G::NPG) * 32 + ql;
            ok1[t] = true;
            if (pr >= G::R1) pr = G::R1 - 1;                    // padding lanes of the last tile compute a duplicate
            const int ry = pr / G::R1W, rx = pr - ry * G::R1W;
            a1[t] = 2 * ry * G::PITCH1 + rx * G::PSTR1 + h * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc1[t][r] = 0.f;
        }
        bb_conv_chunk<T, G::T1W>(patch1, a1, ok1, G::PITCH1, G::PSTR1, w1src, 0, 9, bq, acc1, G::ODD);
    }
    // conv2's first fragments (this wave's half of the reduction: channels 64 pg ..) and the shortcut's while the intermediate is written
    load_b(bq[0], w2src, pg * 9, 18);
    load_b(bq[1], w2src, pg * 9 + 1, 18);
    uint4 wd[2];
#pragma unroll
    for (int g = 0; g < 2; ++g) wd[g] = p.wds[(size_t)(ct * 8 + 2 * pg + g) * 64 + lane];
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();                                            // every wave is done reading the input patch
    // ---- intermediate: ReLU(conv1 + b1) rounded to bf16 -> patch2 [region pixel][128] ; zeros outside the map (conv2's padding)
    {
        const int c0 = ct * 32 + 4 * h;
        float4 bias[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) bias[k] = *(const float4*)&sbias[c0 + 8 * k];
#pragma unroll
        for (int t = 0; t < G::T1W; ++t) {
            const int pr = (pg + t * G::NPG) * 32 + ql;
            if (pr >= G::R1) continue;
            const int ry = pr / G::R1W, rx = pr - ry * G::R1W;
            const int my = oy0 - 1 + ry, mx = ox0 - 1 + rx;
            const bool inside = my >= 0 && my < p.Ho && mx >= 0 && mx < p.Wo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                float v[4] = {acc1[t][4 * k] + bias[k].x, acc1[t][4 * k + 1] + bias[k].y, acc1[t][4 * k + 2] + bias[k].z,
                              acc1[t][4 * k + 3] + bias[k].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e], 0.f) : 0.f;
                *(uint2*)(patch2 + ry * G::PITCH2 + rx * G::PSTR2 + (c0 + 8 * k) * 2) = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
            }
        }
    }
    __syncthreads();                                            // patch2 complete

    // ---- conv2 on the 4 x 16 tile: both output pixel tiles (rows 2 t, 2 t + 1), input channels 64 pg .. 64 pg + 63 (one B fragment
    //      feeds two MFMAs); the two halves of the reduction meet through LDS below
    f32x16 acc[2];
    const int tx = ql & 15;
    {
        int a2[2];
        bool ok2[2] = {true, true};
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            a2[t] = (2 * t + (ql >> 4)) * G::PITCH2 + tx * G::PSTR2 + h * 16;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        }
        bb_conv_chunk<T, 2>(patch2 + pg * 128, a2, ok2, G::PITCH2, G::PSTR2, w2src, pg * 9, 18, bq, acc);
    }
    // ---- projection shortcut: k-groups 2 pg, 2 pg + 1 of x[2 oy][2 ox] into the same accumulators (never rounded on its own)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const unsigned char* ap = smem + G::SC + (32 * t + ql) * G::PSTR1 + h * 16 + 2 * pg * 32;
#pragma unroll
        for (int g = 0; g < 2; ++g) mfma_kgroup<T>(wd[g], *(const uint4*)(ap + g * 32), acc[t]);
    }
    // ---- wave (ct, pg) finishes pixel tile pg: hand the other tile's partial sums to wave (ct, 1 - pg)
    {
        float4* xw = (float4*)(smem + G::XCH + (ct * 2 + pg) * 4096) + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k)
            xw[k * 64] = pg ? make_float4(acc[0][4 * k], acc[0][4 * k + 1], acc[0][4 * k + 2], acc[0][4 * k + 3])
                            : make_float4(acc[1][4 * k], acc[1][4 * k + 1], acc[1][4 * k + 2], acc[1][4 * k + 3]);
    }
    __syncthreads();
    float fin[16];
    {
        const float4* xr = (const float4*)(smem + G::XCH + (ct * 2 + (1 - pg)) * 4096) + lane;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 o = xr[k * 64];
            fin[4 * k] = (pg ? acc[1][4 * k] : acc[0][4 * k]) + o.x;
            fin[4 * k + 1] = (pg ? acc[1][4 * k + 1] : acc[0][4 * k + 1]) + o.y;
            fin[4 * k + 2] = (pg ? acc[1][4 * k + 2] : acc[0][4 * k + 2]) + o.z;
            fin[4 * k + 3] = (pg ? acc[1][4 * k + 3] : acc[0][4 * k + 3]) + o.w;
        }
    }
    // ---- epilogue from the registers: + (b2 + b_ds), ReLU, pair the 8-byte cout runs of the two half-waves into 16-byte stores
    {
        const int oy = oy0 + 2 * pg + (ql >> 4), ox = ox0 + tx;
        const bool live = oy < p.Ho && ox < p.Wo;
        T* orow = (T*)p.out + ((size_t)(img * p.Ho + oy) * p.Wo + ox) * 128 + ct * 32;
#pragma unroll
        for (int m = 0; m < 2; ++m) {
            const float4 b0 = *(const float4*)&sbias[128 + ct * 32 + 16 * m + 4 * h];
            const float4 b1v = *(const float4*)&sbias[128 + ct * 32 + 16 * m + 8 + 4 * h];
            uint32_t r0x = pack_bf2(fmaxf(fin[8 * m] + b0.x, 0.f), fmaxf(fin[8 * m + 1] + b0.y, 0.f));
            uint32_t r0y = pack_bf2(fmaxf(fin[8 * m + 2] + b0.z, 0.f), fmaxf(fin[8 * m + 3] + b0.w, 0.f));
            uint32_t r1x = pack_bf2(fmaxf(fin[8 * m + 4] + b1v.x, 0.f), fmaxf(fin[8 * m + 5] + b1v.y, 0.f));
            uint32_t r1y = pack_bf2(fmaxf(fin[8 * m + 6] + b1v.z, 0.f), fmaxf(fin[8 * m + 7] + b1v.w, 0.f));
            auto sx = __builtin_amdgcn_permlane32_swap(r0x, r1x, false, false);
            auto sy = __builtin_amdgcn_permlane32_swap(r0y, r1y, false, false);
            if (live) *(uint4*)(orow + 16 * m + 8 * h) = make_uint4(sx[0], sy[0], sx[1], sy[1]);
        }
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_basicblock_nhwc(const void* in, const void* wfrag1, const float* bias1, const void* wfrag2,
                                      const float* bias2, void* out, const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W, C, tile_rows]
    if (!in || !wfrag1 || !wfrag2 || !out || !dims) return COBEVT_ERR_ARG;
    BasicBlockParams p;
    p.in = in; p.w1 = (const uint4*)wfrag1; p.b1 = bias1; p.w2 = (const uint4*)wfrag2; p.b2 = bias2; p.out = out;
    const int dtype = dims[0];
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3];
    const int c = dims[4];
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 1 || p.W < 1) return COBEVT_ERR_SHAPE;
    if ((long)p.N * p.H * p.W * c >= 0x7fffffffL) return COBEVT_ERR_UNSUPPORTED;        // 32-bit element offsets
    // 64 channels: 16 x 16 tiles (27 % halo recompute, 180 MFMAs per wave) or 8 x 16 tiles with two workgroups per CU
    // (dims[5] = 8 | 16, 0 = default); 128 channels: 8 x 16 (LDS, and 640 tiles on the 64 x 64 maps)
    const int th = dims[5];
    if (th != 0 && th != 8 && th != 12 && th != 16) return COBEVT_ERR_ARG;
    // 128 channels, 12-row tiles (round 6 A/B): 20 x 64 x 64 maps make 480 tiles of 14 MFMA pixel-tile units instead of 640 of 10 -
    // 1.9 rounds on 256 CUs instead of 2.5 - at 121 KB of LDS
    if (c == 128 && dtype == 0 && th == 12) return launch_basicblock<bf16_t, 128, 12>(p, stream);
    if (c == 64 && dtype == 0 && th == 8) return launch_basicblock<bf16_t, 64, 8>(p, stream);
    if (c == 64) return dtype == 0 ? launch_basicblock<bf16_t, 64, 16>(p, stream) : launch_basicblock<float, 64, 16>(p, stream);
    if (c == 128) return dtype == 0 ? launch_basicblock<bf16_t, 128, 8>(p, stream) : launch_basicblock<float, 128, 8>(p, stream);
    return COBEVT_ERR_UNSUPPORTED;
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_dsblock_nhwc(const void* in, const void* wfrag1, const float* bias1, const void* wfrag2, const float* bias2,
                                   const void* wfrag_ds, const float* bias_ds, void* out, const int* dims, hipStream_t stream) {
    // dims: [dtype (0), N, H, W, Cin (64), Cout (128)]
    if (!in || !wfrag1 || !wfrag2 || !wfrag_ds || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0 || dims[4] != 64 || dims[5] != 128) return COBEVT_ERR_UNSUPPORTED;
    DsBlockParams p;
    p.in = in; p.w1 = (const uint4*)wfrag1; p.b1 = bias1; p.w2 = (const uint4*)wfrag2; p.b2 = bias2;
    p.wds = (const uint4*)wfrag_ds; p.bds = bias_ds; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3];
    if (p.N < 1 || p.H < 2 || p.W < 2 || (p.H & 1) || (p.W & 1)) return COBEVT_ERR_SHAPE;
    if ((long)p.N * p.H * p.W * 64 >= 0x7fffffffL) return COBEVT_ERR_UNSUPPORTED;          // 32-bit element offsets
    p.Ho = p.H / 2; p.Wo = p.W / 2;
    p.tiles_y = (p.Ho + DsCfg::TH - 1) / DsCfg::TH;
    p.tiles_x = (p.Wo + DsCfg::TW - 1) / DsCfg::TW;
    const long blocks = (long)p.N * p.tiles_y * p.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)dsblock_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, DsCfg::LDS);
    }
    hipLaunchKernelGGL(dsblock_kernel, dim3((unsigned)blocks), dim3(512), DsCfg::LDS, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

#ifdef COBEVT_BB_TRACE
extern "C" int cobevt_bb_read_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_bb_trace), sizeof(unsigned long long) * 32) == hipSuccess ? 0 : 1;
}
#endif
