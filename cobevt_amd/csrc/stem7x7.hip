// ResNet stem: 7x7 / stride 2 / pad 3 convolution on the 3-channel fp32 channels-last image (+ folded BN + ReLU),
// torchvision resnet conv1/bn1/relu reached from opv2v/opencood/models/backbones/resnet_ms.py:67-69 (gfx950).
//
// Reformulated as a 4x4 stride-1 convolution over the 2x2 space-to-depth image:
//     S[Y][X][dy][dx][c] = in[2Y+dy][2X+dx][c]                      (12 channels, padded to 16)
//     out[oy][ox][n] = sum_{a,b in 0..3} sum_{dy,dx,c} S[oy+a-2][ox+b-2][dy][dx][c] * W'[n][a][b][dy][dx][c]
//     W'[n][a][b][dy][dx][c] = w[n][c][2a+dy-1][2b+dx-1]   (0 where an index is -1)
// so every tap is exactly one 32-byte k-group (bf16) whose A fragment is read straight out of an LDS patch at the
// tap's pixel offset, as in conv3x3.hip — no im2col, no per-element tap decoding (the generic igemm "small Cin" path
// spent 269 us here).  A workgroup (4 waves, 8x16 output pixels x 64 channels) is persistent: the [64][16][16]
// weight tile stays in LDS, the next tile's patch is prefetched into registers (fp32 image read once, converted on
// the fly), and the epilogue is staged through LDS for 16-byte coalesced stores.
#include "common.hpp"

namespace cobevt {

struct StemParams {
    const float* in;     // (N, H, W, 3) fp32
    const void* wgt;     // [Cout][16 taps][16 ch]
    const float* bias;   // folded BN shift
    void* out;           // (N, Ho, Wo, Cout)
    int N, H, W, Ho, Wo, Cout;
    int act;
    int tiles_y, tiles_x, tiles_n, ntiles;
};

template <typename T> struct StemCfg {
    static constexpr int KG = 16 * Elem<T>::kBytes / 32;      // k-groups per tap (1 bf16, 2 fp32)
    static constexpr int PIX = 16 * Elem<T>::kBytes + 16;     // patch pixel stride (odd multiple of 16 bytes)
    static constexpr int PH = 11, PW = 19;                    // (8 + 3) x (16 + 3) space-to-depth pixels
    static constexpr int PROW = (PW * PIX + 255) / 256 * 256;
    static constexpr int PATCH = PH * PROW;
    static constexpr int WROW = 16 * 16 * Elem<T>::kBytes + 16;
    static constexpr int WBYTES = 64 * WROW;
    static constexpr int SROW = 64 * 4 + 16;                  // fp32 staging row (64 channels)
    static constexpr int STAGE = 128 * SROW;
    static constexpr int LDS = PATCH + WBYTES + STAGE;
};

template <typename T>
__global__ __launch_bounds__(256) void stem7x7_kernel(StemParams p) {
    using C = StemCfg<T>;
    constexpr int KG = C::KG;
    constexpr int NPIECE = C::PH * C::PW * 2 * 3;             // (pixel, dy, 2-float piece): 8-byte global loads
    constexpr int P_IT = (NPIECE + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wl = smem + C::PATCH;
    float* stage = (float*)(smem + C::PATCH + C::WBYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    // zero the patch once: channels 12..15 of every pixel (and the pad bytes) stay zero for the whole kernel
    for (int i = tid; i < C::PATCH / 16; i += 256) ((uint4*)patch)[i] = make_uint4(0, 0, 0, 0);

    float2 preg[P_IT];
    auto decode = [&](int tile, int& img, int& oy0, int& ox0, int& n0) {
        const int tn = tile % p.tiles_n;
        int rest = tile / p.tiles_n;
        const int tx = rest % p.tiles_x; rest /= p.tiles_x;
        const int ty = rest % p.tiles_y;
        img = rest / p.tiles_y;
        oy0 = ty * 8; ox0 = tx * 16; n0 = tn * 64;
    };
    auto load_patch = [&](int tile) {
        int img, oy0, ox0, n0;
        decode(tile, img, oy0, ox0, n0);
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int item = tid + it * 256;
            float2 v = make_float2(0.f, 0.f);
            if (item < NPIECE) {
                const int pc = item % 3, rest = item / 3;     // piece = 2 consecutive floats of the 6 (dx, c) values
                const int dy = rest & 1, pix = rest >> 1;
                const int py = pix / C::PW, px = pix - py * C::PW;
                const int iy = 2 * (oy0 - 2 + py) + dy, ix = 2 * (ox0 - 2 + px);     // image row / first image column
                // the 6 floats in[iy][ix..ix+1][0..2] are contiguous; columns are valid in pairs (W is even)
                if (iy >= 0 && iy < p.H && ix >= 0 && ix + 1 < p.W)
                    v = *(const float2*)(p.in + (((size_t)img * p.H + iy) * p.W + ix) * 3 + pc * 2);
            }
            preg[it] = v;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            const int item = tid + it * 256;
            if (item < NPIECE) {
                const int pc = item % 3, rest = item / 3;
                const int dy = rest & 1, pix = rest >> 1;
                const int py = pix / C::PW, px = pix - py * C::PW;
                unsigned char* dst = patch + py * C::PROW + px * C::PIX + (dy * 6 + pc * 2) * Elem<T>::kBytes;
                if constexpr (Elem<T>::kIsBf16) *(uint32_t*)dst = pack_bf2(preg[it].x, preg[it].y);
                else *(float2*)dst = preg[it];
            }
        }
    };

    // ---- weights resident in LDS: [64][16 taps][16 ch]
    int img, oy0, ox0, n0;
    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    decode(tile, img, oy0, ox0, n0);
    int n0_loaded = -1;
    auto load_weights = [&](int nb) {
        constexpr int PIECES = 16 * 16 * Elem<T>::kBytes / 16;        // 16-byte pieces per weight row
        for (int i = tid; i < 64 * PIECES; i += 256) {
            const int row = i / PIECES, j = i - row * PIECES;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (nb + row < p.Cout) v = *(const uint4*)((const unsigned char*)p.wgt + ((size_t)(nb + row) * PIECES + j) * 16);
            *(uint4*)(wl + row * C::WROW + j * 16) = v;
        }
        n0_loaded = nb;
    };
    __syncthreads();          // patch zeroed
    load_patch(tile);
    load_weights(n0);

    const int abase = ((2 * wave + (ql >> 4)) * C::PROW) + (ql & 15) * C::PIX + h * 16;   // wave owns output rows 2w, 2w+1
    const int bbase = ql * C::WROW + h * 16;

    for (; tile < p.ntiles; tile += gridDim.x) {
        decode(tile, img, oy0, ox0, n0);
        store_patch();
        if (n0 != n0_loaded) { __syncthreads(); load_weights(n0); }
        __syncthreads();
        const int next = tile + gridDim.x;
        if (next < p.ntiles) load_patch(next);

        f32x16 acc[2];
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
#pragma unroll
            for (int bb = 0; bb < 4; ++bb) {
                const unsigned char* pa = patch + abase + a * C::PROW + bb * C::PIX;
                const unsigned char* pw = wl + bbase + (a * 4 + bb) * 16 * Elem<T>::kBytes;
#pragma unroll
                for (int g = 0; g < KG; ++g) {
                    const uint4 af = *(const uint4*)(pa + g * 32);
                    const uint4 b0 = *(const uint4*)(pw + g * 32);
                    const uint4 b1 = *(const uint4*)(pw + 32 * C::WROW + g * 32);
                    mfma_kgroup<T>(af, b0, acc[0]);
                    mfma_kgroup<T>(af, b1, acc[1]);
                }
            }
        }
        // ---- epilogue: bias + activation staged as fp32, then coalesced 16-byte stores
        constexpr int SR = C::SROW / 4;
#pragma unroll
        for (int b = 0; b < 2; ++b) {
            const int cl = b * 32 + ql;
            const float bias = (p.bias && n0 + cl < p.Cout) ? p.bias[n0 + cl] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, lane);
                float v = acc[b][r] + bias;
                v = p.act == 1 ? fmaxf(v, 0.f) : v;
                stage[((2 * wave + (row >> 4)) * 16 + (row & 15)) * SR + cl] = v;
            }
        }
        __syncthreads();      // staging complete; every wave is done with the patch -> next store_patch may overwrite it
        constexpr int CH = Elem<T>::kChunk;
        constexpr int CPP = 64 / CH;
        T* out = (T*)p.out;
        for (int item = tid; item < 128 * CPP; item += 256) {
            const int px = item / CPP, cj = item - px * CPP;
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15), col = n0 + cj * CH;
            if (oy >= p.Ho || ox >= p.Wo || col >= p.Cout) continue;
            float v[8];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = stage[px * SR + cj * CH + e];
            const size_t o = (((size_t)img * p.Ho + oy) * p.Wo + ox) * p.Cout + col;
            if (col + CH <= p.Cout) *(uint4*)(out + o) = f32_to_chunk<T>(v);
            else for (int e = 0; e < CH && col + e < p.Cout; ++e) store_elem<T>(out, o + e, v[e]);
        }
        // the next iteration's store_patch / staging writes are ordered by the barriers above and below
        __syncthreads();
    }
}


// ---------------------------------------------------------------------------------------------------------------------
// Stem + MaxPool2d(3, 2, 1) in one kernel (cobevt_stem_conv7x7s2_pool): torchvision resnet conv1/bn1/relu/maxpool,
// resnet_ms.py:67-71.  Separately the stem writes its 168 MB map (20 images) and the pool reads it back: 123 us for
// 63 MB of input and 42 MB of pooled output.  Here a workgroup owns a 4 x 8 tile of POOLED pixels: it evaluates the
// conv on the 9 x 17 region those windows cover (origin (2 py0 - 1, 2 px0 - 1); its 153 pixels linearised into five
// 32-row MFMA tiles), stages bias + ReLU as the storage type (rounding commutes with max) with zeros for conv pixels
// outside the map (post-ReLU values are >= 0 and every window holds a valid pixel, so 0 is the max-pool identity),
// and writes the 3 x 3 / stride 2 maxima.  20 % of the conv is recomputed on tile borders; the kernel stays HBM-bound.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T> struct StemPoolCfg {
    static constexpr int KG = 16 * Elem<T>::kBytes / 32;
    static constexpr int PIX = 16 * Elem<T>::kBytes + 16;
    static constexpr int RH = 9, RW = 17, RPIX = RH * RW;     // conv region
    static constexpr int NPT = (RPIX + 31) / 32;              // 5 MFMA pixel tiles
    static constexpr int PH = RH + 3, PW = RW + 3;            // 12 x 20 space-to-depth pixels
    static constexpr int PROW = (PW * PIX + 255) / 256 * 256;
    static constexpr int PATCH = PH * PROW;
    static constexpr int WROW = 16 * 16 * Elem<T>::kBytes + 16;
    static constexpr int WBYTES = 64 * WROW;
    static constexpr int SROW = 64 * Elem<T>::kBytes + 16;    // staging row: 64 channels in the storage type
    static constexpr int STAGE = (RPIX * SROW + 255) / 256 * 256;
    static constexpr int LDS = PATCH + WBYTES + STAGE + 256;      // + the 64 bias values
};

#ifdef COBEVT_STEM_TRACE      // tools/stem_trace.py builds a copy with s_memtime marks (never the product .so)
__device__ unsigned long long cobevt_stem_trace[16];
#define COBEVT_ST_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (int)blockIdx.x + 2 * (int)gridDim.x) cobevt_stem_trace[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define COBEVT_ST_MARK(i) do {} while (0)
#endif

constexpr int kStemPoolThreads = 512;   // 8 waves share the ten (pixel tile, cout half) units of a region: 2,2,1,1,1,1,1,1

template <typename T>
__global__ __launch_bounds__(kStemPoolThreads, 4) void stem_pool_kernel(StemParams p) {
    using C = StemPoolCfg<T>;
    constexpr int KG = C::KG, RW = C::RW, RPIX = C::RPIX, NT = kStemPoolThreads;
    constexpr int NPIECE = C::PH * C::PW * 2 * 3;
    constexpr int P_IT = (NPIECE + NT - 1) / NT;
    constexpr int NUNIT = C::NPT * 2;                         // (pixel tile, 32-cout half)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wl = smem + C::PATCH;
    unsigned char* stage = smem + C::PATCH + C::WBYTES;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int Hp = p.Ho / 2, Wp = p.Wo / 2;                   // pooled map (Ho, Wo even)

    for (int i = tid; i < C::PATCH / 16; i += NT) ((uint4*)patch)[i] = make_uint4(0, 0, 0, 0);

    // patch pieces of this thread: the (pixel, dy, float-pair) decomposition does not depend on the tile
    int pdst[P_IT], prow[P_IT], pxo[P_IT], pcol[P_IT];        // LDS offset; image row, image column, float offset relative to the region origin
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int item = tid + it * NT;
        pdst[it] = -1; prow[it] = 0; pxo[it] = 0; pcol[it] = 0;
        if (item < NPIECE) {
            const int pc = item % 3, rest = item / 3;
            const int dy = rest & 1, pix = rest >> 1;
            const int py = pix / C::PW, px = pix - py * C::PW;
            pdst[it] = py * C::PROW + px * C::PIX + (dy * 6 + pc * 2) * Elem<T>::kBytes;
            prow[it] = 2 * (py - 2) + dy;                     // image row = 2*oy0 + prow
            pxo[it] = 2 * (px - 2);                           // first image column of the pixel pair = 2*ox0 + pxo
            pcol[it] = pxo[it] * 3 + pc * 2;                  // float offset inside the image row = 6*ox0 + pcol
        }
    }
    float2 preg[P_IT];
    auto decode = [&](int tile, int& img, int& py0, int& px0) {
        const int tx = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y;
        img = rest / p.tiles_y;
        py0 = ty * 4; px0 = tx * 8;
    };
    auto load_patch = [&](int tile) {
        int img, py0, px0;
        decode(tile, img, py0, px0);
        const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;       // conv region origin
        const float* base = p.in + (size_t)img * p.H * p.W * 3;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            float2 v = make_float2(0.f, 0.f);
            const int iy = 2 * oy0 + prow[it], ix = 2 * ox0 + pxo[it];
            // the 6 floats in[iy][ix..ix+1][0..2] are contiguous; columns are valid in pairs (W is even)
            if (pdst[it] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix + 1 < p.W)
                v = *(const float2*)(base + (size_t)iy * p.W * 3 + 6 * ox0 + pcol[it]);
            preg[it] = v;
        }
    };
    auto store_patch = [&]() {
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (pdst[it] >= 0) {
                unsigned char* dst = patch + pdst[it];
                if constexpr (Elem<T>::kIsBf16) *(uint32_t*)dst = pack_bf2(preg[it].x, preg[it].y);
                else *(float2*)dst = preg[it];
            }
    };

    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    {
        // the first image patch is requested before the weights, and the weight pieces of a thread are all in flight at
        // once (unconditional, row clamped): the rolled, branch-guarded loop paid one L2 round trip per iteration before the
        // patch load could even be issued
        load_patch(tile);
        constexpr int PIECES = 16 * 16 * Elem<T>::kBytes / 16;
        constexpr int W_IT = (64 * PIECES + NT - 1) / NT;
        uint4 wv[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * NT;
            const int row = i / PIECES, j = i - row * PIECES;
            wv[it] = *(const uint4*)((const unsigned char*)p.wgt + ((size_t)(row < p.Cout ? row : 0) * PIECES + j) * 16);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * NT;
            const int row = i / PIECES, j = i - row * PIECES;
            if (i < 64 * PIECES) *(uint4*)(wl + row * C::WROW + j * 16) = row < p.Cout ? wv[it] : make_uint4(0, 0, 0, 0);
        }
    }
    __syncthreads();

    // this wave's units: u = wave and u = wave + 8 (waves 0, 1): pixel tile u >> 1, cout half u & 1
    int abase[2], bbase[2], rpix[2], chalf[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int u = wave + 8 * t;
        const int pt = (u < NUNIT ? u : 0) >> 1;
        chalf[t] = u & 1;
        int pr = pt * 32 + ql;
        rpix[t] = pr;
        if (pr >= RPIX) pr = RPIX - 1;
        const int ry = pr / RW, rx = pr - ry * RW;
        abase[t] = ry * C::PROW + rx * C::PIX + h * 16;
        bbase[t] = (chalf[t] * 32 + ql) * C::WROW + h * 16;
    }
    const bool two = wave + 8 < NUNIT;                        // wave-uniform
    float* bias_l = (float*)(smem + C::PATCH + C::WBYTES + C::STAGE);     // bias in LDS: 32 VGPRs less than keeping it
    if (tid < 64) bias_l[tid] = p.bias ? p.bias[tid] : 0.f;
    __syncthreads();

    for (; tile < p.ntiles; tile += gridDim.x) {
        int img, py0, px0;
        decode(tile, img, py0, px0);
        const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;
        COBEVT_ST_MARK(0);
        store_patch();
        __syncthreads();
        COBEVT_ST_MARK(1);
        const int next = tile + gridDim.x;
        if (next < p.ntiles) load_patch(next);
        COBEVT_ST_MARK(2);

        f32x16 acc[2];
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
        // 16 taps x KG k-groups, operands one step ahead of the MFMAs in a two-slot register ring with the issue order
        // pinned (each MFMA had its two ds_read_b128 right in front of it: 3.3k cycles for 32 MFMAs in the s_memtime trace)
        constexpr int NS = 16 * KG;
        constexpr int RS = 2;                                 // ring slots (one step ahead keeps the kernel at 128 VGPRs = 2 workgroups / CU)
        uint4 ra[RS][2], rb[RS][2];
        auto fetch = [&](int slot, int i) {                  // i = tap * KG + g (compile-time after unrolling)
            const int tap = i / KG, g = i - tap * KG;
            const int toff = (tap >> 2) * C::PROW + (tap & 3) * C::PIX + g * 32;
            const int woff = tap * 16 * Elem<T>::kBytes + g * 32;
            rb[slot][0] = *(const uint4*)(wl + bbase[0] + woff);
            ra[slot][0] = *(const uint4*)(patch + abase[0] + toff);
            if (two) {
                rb[slot][1] = *(const uint4*)(wl + bbase[1] + woff);
                ra[slot][1] = *(const uint4*)(patch + abase[1] + toff);
            }
        };
        fetch(0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i + 1 < NS) fetch((i + 1) % RS, i + 1);
            mfma_kgroup<T>(rb[i % RS][0], ra[i % RS][0], acc[0]);      // D = W . X^T: lane <-> pixel, registers <-> couts
            if (two) mfma_kgroup<T>(rb[i % RS][1], ra[i % RS][1], acc[1]);
            __builtin_amdgcn_sched_barrier(0);
        }
        COBEVT_ST_MARK(3);
        // ---- bias + ReLU, rounded to T, staged [region pixel][64]; conv pixels outside the map -> 0
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            if (t == 1 && !two) break;
            const int pr = rpix[t];
            if (pr >= RPIX) continue;
            const int ry = pr / RW, rx = pr - ry * RW;
            const int oy = oy0 + ry, ox = ox0 + rx;
            const bool inside = oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = chalf[t] * 32 + 8 * k + 4 * h;
                const float4 b = *(const float4*)(bias_l + c0);
                float v[4] = {acc[t][4 * k] + b.x, acc[t][4 * k + 1] + b.y, acc[t][4 * k + 2] + b.z, acc[t][4 * k + 3] + b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e], 0.f) : 0.f;
                unsigned char* d = stage + pr * C::SROW + c0 * Elem<T>::kBytes;
                if constexpr (Elem<T>::kIsBf16) *(uint2*)d = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        COBEVT_ST_MARK(4);
        __syncthreads();      // staging complete; every wave is done with the patch
        COBEVT_ST_MARK(5);
        // ---- 3 x 3 / stride 2 maxima: pooled pixel (qy, qx) of the 4 x 8 tile covers region rows 2qy..2qy+2, cols 2qx..2qx+2
        constexpr int CH = Elem<T>::kChunk;
        constexpr int CPP = 64 / CH;
        T* out = (T*)p.out;
        for (int item = tid; item < 32 * CPP; item += NT) {
            const int q = item / CPP, cj = item - q * CPP;
            const int qy = q >> 3, qx = q & 7;
            const int py = py0 + qy, px = px0 + qx;
            if (py >= Hp || px >= Wp) continue;
            float m[8];
#pragma unroll
            for (int e = 0; e < CH; ++e) m[e] = 0.f;
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    float v[8];
                    chunk_to_f32<T>(*(const uint4*)(stage + ((2 * qy + dy) * RW + 2 * qx + dx) * C::SROW + cj * 16), v);
#pragma unroll
                    for (int e = 0; e < CH; ++e) m[e] = fmaxf(m[e], v[e]);
                }
            *(uint4*)(out + (((size_t)img * Hp + py) * Wp + px) * 64 + cj * CH) = f32_to_chunk<T>(m);
        }
        COBEVT_ST_MARK(6);
        __syncthreads();
        COBEVT_ST_MARK(7);
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_stem_conv7x7s2(const float* in, const void* wgt, const float* bias, void* out, const int* dims,
                                     hipStream_t stream) {
    // dims: [dtype, N, H, W, Cout, act]
    if (!in || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    StemParams p;
    const int dtype = dims[0];
    p.in = in; p.wgt = wgt; p.bias = bias; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cout = dims[4]; p.act = dims[5];
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 2 || p.W < 2 || (p.H & 1) || (p.W & 1) || p.Cout < 1) return COBEVT_ERR_SHAPE;
    if (p.Cout % (dtype == 0 ? 8 : 4)) return COBEVT_ERR_SHAPE;
    p.Ho = p.H / 2; p.Wo = p.W / 2;            // (H + 6 - 7) / 2 + 1 for even H
    p.tiles_y = (p.Ho + 7) / 8; p.tiles_x = (p.Wo + 15) / 16; p.tiles_n = (p.Cout + 63) / 64;
    const long nt = (long)p.N * p.tiles_y * p.tiles_x * p.tiles_n;
    if (nt > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    p.ntiles = (int)nt;
    const size_t lds = dtype == 0 ? StemCfg<bf16_t>::LDS : StemCfg<float>::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem7x7_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemCfg<bf16_t>::LDS);
        (void)hipFuncSetAttribute((const void*)stem7x7_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemCfg<float>::LDS);
        attr_set = true;
    }
    const int per_cu = dtype == 0 ? 2 : 1;
    const unsigned blocks = (unsigned)(nt < 256L * per_cu ? nt : 256L * per_cu);
    if (dtype == 0) hipLaunchKernelGGL(stem7x7_kernel<bf16_t>, dim3(blocks), dim3(256), lds, stream, p);
    else hipLaunchKernelGGL(stem7x7_kernel<float>, dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_stem_conv7x7s2_pool(const float* in, const void* wgt, const float* bias, void* out, const int* dims,
                                          hipStream_t stream) {
    // dims: [dtype, N, H, W]   (Cout = 64, ReLU, H and W multiples of 4)
    if (!in || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    StemParams p;
    const int dtype = dims[0];
    p.in = in; p.wgt = wgt; p.bias = bias; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cout = 64; p.act = 1;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 4 || p.W < 4 || (p.H & 3) || (p.W & 3)) return COBEVT_ERR_SHAPE;
    p.Ho = p.H / 2; p.Wo = p.W / 2;
    p.tiles_y = (p.Ho / 2 + 3) / 4; p.tiles_x = (p.Wo / 2 + 7) / 8; p.tiles_n = 1;
    const long nt = (long)p.N * p.tiles_y * p.tiles_x;
    if (nt > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    p.ntiles = (int)nt;
    const size_t lds = dtype == 0 ? StemPoolCfg<bf16_t>::LDS : StemPoolCfg<float>::LDS;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<bf16_t>::LDS);
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<float>::LDS);
        attr_set = true;
    }
    const int per_cu = dtype == 0 ? 2 : 1;
    const unsigned blocks = (unsigned)(nt < 256L * per_cu ? nt : 256L * per_cu);
    if (dtype == 0) hipLaunchKernelGGL(stem_pool_kernel<bf16_t>, dim3(blocks), dim3(kStemPoolThreads), lds, stream, p);
    else hipLaunchKernelGGL(stem_pool_kernel<float>, dim3(blocks), dim3(kStemPoolThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

#ifdef COBEVT_STEM_TRACE
extern "C" int cobevt_stem_read_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_stem_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif
