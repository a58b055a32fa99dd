// ResNet stem: 7x7 / stride 2 / pad 3 convolution on the 3-channel fp32 channels-last image (+ folded BN + ReLU),
// torchvision resnet conv1/bn1/relu reached from opv2v/opencood/models/backbones/resnet_ms.py:67-69 (gfx950).
//
// Reformulated as a 4x4 stride-1 convolution over the 2x2 space-to-depth image:
//     S[Y][X][dy][dx][c] = in[2Y+dy][2X+dx][c]                      (12 channels, padded to 16)
//     out[oy][ox][n] = sum_{a,b in 0..3} sum_{dy,dx,c} S[oy+a-2][ox+b-2][dy][dx][c] * W'[n][a][b][dy][dx][c]
//     W'[n][a][b][dy][dx][c] = w[n][c][2a+dy-1][2b+dx-1]   (0 where an index is -1)
// The A fragments are read straight out of an LDS patch of S, as in conv3x3.hip — no im2col, no per-element tap decoding
// (the generic igemm "small Cin" path spent 269 us here).  In LDS a patch pixel holds its 12 real channels only, so a tap
// row a is 48 CONTIGUOUS values of the patch row (b, dy, dx, c) = three 16-value k-groups: K = 4 x 48 = 192, 12 MFMA steps.
// (The weight image in global memory keeps one 16-channel group per tap, [Cout][16 taps][16 ch] with 4 zero channels; the
// kernels drop them while staging the weights into LDS.)  A workgroup (4 waves, 8x16 output pixels x 64 channels) is
// persistent: the weight tile stays in LDS, the next tile's patch is prefetched into registers (fp32 image read once,
// converted on the fly), and the epilogue is staged through LDS for 16-byte coalesced stores.
#include "common.hpp"

namespace cobevt {

struct StemParams {
    const float* in;     // (N, H, W, 3) fp32
    const unsigned char* in_u8;   // stem_pool_kernel<T, true>: (N, H, W, 3) uint8 camera frames instead of `in`
    const float* lut;    // ... and the [3][256] fp32 table byte -> normalised value of RgbPreProcessor (host/rgb_preprocessor.py)
    const void* wgt;     // [Cout][16 taps][16 ch] (12 real channels per tap)
    const float* bias;   // folded BN shift
    void* out;           // (N, Ho, Wo, Cout)
    int N, H, W, Ho, Wo, Cout;
    int act;
    int tiles_y, tiles_x, tiles_n, ntiles;
};

// 16 bytes from an 8-byte aligned LDS address as two ds_read_b64.  Volatile so that the compiler does not fuse them into one
// ds_read2_b64: that instruction is serviced as two 4 x 16-lane accesses on 32 banks (8 LDS cycles per wave, 128 B / clock)
// where two ds_read_b64 take 2 x 2 cycles on 64 banks (MI355X_MICROARCH.md, LDS table)
__device__ __forceinline__ uint4 lds_read_2x8(const unsigned char* a) {
    typedef const volatile __attribute__((address_space(3))) unsigned long long* lds_u64_ptr;   // keep it a DS access
    const unsigned long long lo = *(lds_u64_ptr)a, hi = *(lds_u64_ptr)(a + 8);
    return make_uint4((unsigned)lo, (unsigned)(lo >> 32), (unsigned)hi, (unsigned)(hi >> 32));
}

template <typename T> struct StemCfg {
    static constexpr int KB = Elem<T>::kBytes;
    static constexpr int STEPS_ROW = 48 * KB / 32;            // 32-byte operand steps per tap row (bf16 3, fp32 6)
    static constexpr int PIX = 12 * KB;                       // patch pixel pitch: 6 / 12 banks (bf16: 8-byte aligned only)
    static constexpr int PH = 11, PW = 19;                    // (8 + 3) x (16 + 3) space-to-depth pixels
    // a lane half = pixels (y, 0..15), (y + 1, 0..15): pitch == 16 PIX (mod 256 B) continues the bank sequence into the next row
    static constexpr int PROW = (PW * PIX - 16 * PIX % 256 + 255) / 256 * 256 + 16 * PIX % 256;
    static constexpr int PATCH = (PH * PROW + 255) / 256 * 256;
    static constexpr int WROW = 192 * KB + 16;
    static constexpr int WBYTES = 64 * WROW;
    static constexpr int SROW = 64 * 4 + 16;                  // fp32 staging row (64 channels)
    static constexpr int STAGE = 128 * SROW;
    static constexpr int LDS = PATCH + WBYTES + STAGE;
};

template <typename T>
__global__ __launch_bounds__(256) void stem7x7_kernel(StemParams p) {
    using C = StemCfg<T>;
    constexpr int KB = C::KB;
    constexpr int NPIECE = C::PH * C::PW * 2 * 3;             // (pixel, dy, 2-float piece): 8-byte global loads
    constexpr int P_IT = (NPIECE + 255) / 256;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* patch = smem;
    unsigned char* wl = smem + C::PATCH;
    float* stage = (float*)(smem + C::PATCH + C::WBYTES);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

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
                unsigned char* dst = patch + py * C::PROW + px * C::PIX + (dy * 6 + pc * 2) * KB;
                if constexpr (Elem<T>::kIsBf16) *(uint32_t*)dst = pack_bf2(preg[it].x, preg[it].y);
                else *(float2*)dst = preg[it];
            }
        }
    };

    // ---- weights resident in LDS: [64][4 tap rows][48] (the global image's 4 zero channels per tap dropped)
    int img, oy0, ox0, n0;
    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    decode(tile, img, oy0, ox0, n0);
    int n0_loaded = -1;
    auto load_weights = [&](int nb) {
        constexpr int PIECES = 16 * 16 * Elem<T>::kBytes / 16;        // 16-byte pieces per weight row
        constexpr int PPT = PIECES / 16;                              // pieces per tap: bf16 2 (8 channels), fp32 4 (4 channels)
        for (int i = tid; i < 64 * PIECES; i += 256) {
            const int row = i / PIECES, j = i - row * PIECES;
            const int tap = j / PPT, part = j - tap * PPT;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (nb + row < p.Cout) v = *(const uint4*)((const unsigned char*)p.wgt + ((size_t)(nb + row) * PIECES + j) * 16);
            unsigned char* d = wl + row * C::WROW + ((tap >> 2) * 48 + (tap & 3) * 12) * KB + part * 16;
            if constexpr (Elem<T>::kIsBf16) {                         // channels 0..7 (two 8-byte halves), then 8..11
                *(uint2*)d = make_uint2(v.x, v.y);
                if (part == 0) *(uint2*)(d + 8) = make_uint2(v.z, v.w);
            } else if (part < 3) {
                *(uint4*)d = v;
            }
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
            for (int sub = 0; sub < C::STEPS_ROW; ++sub) {
                const unsigned char* pa = patch + abase + a * C::PROW + sub * 32;
                const unsigned char* pw = wl + bbase + a * 48 * KB + sub * 32;
                uint4 af;
                if constexpr (Elem<T>::kIsBf16) {             // 8-byte aligned: two ds_read_b64 (see lds_read_2x8)
                    af = lds_read_2x8(pa);
                } else {
                    af = *(const uint4*)pa;
                }
                const uint4 b0 = *(const uint4*)pw;
                const uint4 b1 = *(const uint4*)(pw + 32 * C::WROW);
                mfma_kgroup<T, false>(af, b0, acc[0]);    // A = activation rows, B = weights
                mfma_kgroup<T, false>(af, b1, acc[1]);
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
// 63 MB of input and 42 MB of pooled output.  Here a workgroup owns a ROWS x 8 tile of POOLED pixels: it evaluates the
// conv on the (2 ROWS + 1) x 17 region those windows cover (origin (2 py0 - 1, 2 px0 - 1); its pixels linearised into
// 32-row MFMA tiles), stages bias + ReLU as the storage type (rounding commutes with max) with zeros for conv pixels
// outside the map (post-ReLU values are >= 0 and every window holds a valid pixel, so 0 is the max-pool identity),
// and writes the 3 x 3 / stride 2 maxima.  17 - 24 % of the conv is recomputed on tile borders.
// One wave per (pixel tile, 32-cout half) unit: bf16 = 5 rows = 187 region pixels = 6 pixel tiles (97 % of the MFMA rows
// used) = 12 waves, two workgroups per CU; fp32 = 3 rows = 119 pixels = 4 tiles = 8 waves, one workgroup per CU.
// What the knock-out builds (COBEVT_STEM_KNOCK, tools/stem_probe.py) showed: the kernel is bound by LDS throughput - taking
// out the operand reads, the pool reads, the staging or the image loads each shortens it, deeper operand rings lengthen it.
// So the work went into LDS bytes and bank conflicts: the K order without the zero channels (12 steps, not 16), the pooled
// pixel order of the window reads, the patch row pitch; and into waits that were on the critical path for no reason (the
// store of a tile's result, nine window reads one at a time).  87 -> 70 us bf16, 535 -> 350 us fp32 (same job).
// (Earlier: eight waves for ten units had two waves carry two units - 32 dependent MFMA steps where the others had 16 - and,
// `wave` being a VGPR value, the other six issued the second unit's MFMAs under an empty exec mask, which costs the same
// pipeline time: 2.6 M MFMAs per launch in the PMC pass against 1.6 M of work.)
// ---------------------------------------------------------------------------------------------------------------------
#ifndef COBEVT_STEM_POOL_ROWS_BF16        // tools/stem_probe.py builds other geometries side by side
#define COBEVT_STEM_POOL_ROWS_BF16 5
#endif
#ifndef COBEVT_STEM_POOL_ROWS_F32
#define COBEVT_STEM_POOL_ROWS_F32 3
#endif
#ifndef COBEVT_STEM_KNOCK                 // probe builds only: 1 no operand reads, 2 no MFMAs, 4 no pool reads, 8 no image loads, 16 no staging
#define COBEVT_STEM_KNOCK 0
#endif

typedef short short2_t __attribute__((ext_vector_type(2)));
// maximum of non-negative storage-type values on their bit patterns (see the pool phase): one fp32, or two packed bf16
template <typename T> __device__ __forceinline__ uint32_t word_max(uint32_t a, uint32_t b) {
    if constexpr (Elem<T>::kIsBf16) {
        const short2_t r = __builtin_elementwise_max(__builtin_bit_cast(short2_t, a), __builtin_bit_cast(short2_t, b));
        return __builtin_bit_cast(uint32_t, r);
    } else {
        return (uint32_t)max((int)a, (int)b);
    }
}

template <typename T> struct StemPoolCfg {
    // K order of the pool kernel: (tap row a, tap column b, the 12 real channels of a space-to-depth pixel) - the four zero
    // channels that pad a tap to one k-group in the weight image are dropped in LDS, so a tap row is 48 contiguous values of
    // the patch row (4 pixels x 12) = three 16-value k-groups instead of four: 12 MFMA steps and operand reads, not 16
    static constexpr int KB = Elem<T>::kBytes;
    static constexpr int STEPS_ROW = 48 * KB / 32;            // 32-byte operand steps per tap row (bf16 3, fp32 6)
    static constexpr int NS = 4 * STEPS_ROW;
    static constexpr int PIX = 12 * KB;                       // patch pixel pitch (bf16: 8-byte aligned only)
    static constexpr int ROWS = Elem<T>::kIsBf16 ? COBEVT_STEM_POOL_ROWS_BF16 : COBEVT_STEM_POOL_ROWS_F32;   // pooled rows of a tile
    static constexpr int RH = 2 * ROWS + 1, RW = 17, RPIX = RH * RW;     // conv region
    static constexpr int NPT = (RPIX + 31) / 32;              // MFMA pixel tiles
    static constexpr int NT = 64 * NPT * 2;                   // one wave per (pixel tile, cout half)
    static constexpr int PH = RH + 3, PW = RW + 3;            // space-to-depth pixels under the region's 4 x 4 taps
    // patch row pitch == RW * PIX (mod 256 B): the 32 pixels of an MFMA tile are consecutive in the LINEARISED region, and
    // with this pitch a tile that wraps to the next region row keeps the bank sequence of 32 consecutive pixels (6 or 12
    // banks apart: conflict-free for the 8-byte halves of bf16 and the 16-byte reads of fp32)
    static constexpr int PROW = (PW * PIX - RW * PIX % 256 + 255) / 256 * 256 + RW * PIX % 256;
    static constexpr int PATCH = (PH * PROW + 255) / 256 * 256;
    static constexpr int WROW = 192 * KB + 16;                // 100 / 196 banks: rows 36 / 4 banks apart
    static constexpr int WBYTES = 64 * WROW;
    static constexpr int SROW = 64 * Elem<T>::kBytes + 16;    // staging row: 64 channels in the storage type
    static constexpr int STAGE = (RPIX * SROW + 255) / 256 * 256;
    static constexpr int LDS = PATCH + WBYTES + STAGE + 256;      // + the 64 bias values
    static constexpr int LDS_U8 = LDS + 3 * 256 * 4;              // + the [3][256] fp32 normalisation table of the uint8 ingest
    static_assert(NT <= 1024, "stem_pool_kernel: region too large for one workgroup");
};

#ifdef COBEVT_STEM_TRACE      // tools/stem_probe.py builds copies with s_memtime marks (never the product .so)
__device__ unsigned long long cobevt_stem_trace[16];
#define COBEVT_ST_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0 && tile == (int)blockIdx.x + 2 * (int)gridDim.x) cobevt_stem_trace[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define COBEVT_ST_MARK(i) do {} while (0)
#endif

// U8 = true: the image is the uint8 camera frame itself (a quarter of the fp32 image's bytes over PCIe and out of HBM); a patch
// piece is the 2 bytes of two consecutive (dx, c) values and the value the patch receives is LUT[c][byte], the table holding
// fl32((double(u) / 255 - mean[c]) / std[c]) exactly as RgbPreProcessor + the collate cast produce it
// (opv2v/opencood/data_utils/pre_processor/rgb_preprocessor.py:14-31): bit-identical to the fp32-image path by construction.
template <typename T, bool U8 = false>
__global__ __launch_bounds__(StemPoolCfg<T>::NT) void stem_pool_kernel(StemParams p) {
    using C = StemPoolCfg<T>;
    constexpr int RW = C::RW, RPIX = C::RPIX, NT = C::NT, KB = C::KB;
    constexpr int NPIECE = C::PH * C::PW * 2 * 3;
    constexpr int P_IT = (NPIECE + NT - 1) / NT;
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
    int plut[P_IT];                                           // uint8 ingest: table rows (channels) of the piece's two values
#pragma unroll
    for (int it = 0; it < P_IT; ++it) {
        const int item = tid + it * NT;
        pdst[it] = -1; prow[it] = 0; pxo[it] = 0; pcol[it] = 0; plut[it] = 0;
        if (item < NPIECE) {
            const int pc = item % 3, rest = item / 3;
            const int dy = rest & 1, pix = rest >> 1;
            const int py = pix / C::PW, px = pix - py * C::PW;
            pdst[it] = py * C::PROW + px * C::PIX + (dy * 6 + pc * 2) * KB;
            prow[it] = 2 * (py - 2) + dy;                     // image row = 2*oy0 + prow
            pxo[it] = 2 * (px - 2);                           // first image column of the pixel pair = 2*ox0 + pxo
            pcol[it] = pxo[it] * 3 + pc * 2;                  // float offset inside the image row = 6*ox0 + pcol
            plut[it] = ((pc * 2) % 3) | (((pc * 2 + 1) % 3) << 8);
        }
    }
    // element offset of the piece relative to in[img][2 oy0][2 ox0][0] (fits 32 bits: |prow| <= 2 PH, the row pitch < 2^24)
    int poff[P_IT];
#pragma unroll
    for (int it = 0; it < P_IT; ++it) poff[it] = prow[it] * p.W * 3 + pcol[it];
    float2 preg[P_IT];
    unsigned int praw[P_IT];
    const float* lut_l = (const float*)(smem + C::LDS);       // [3][256], filled below (U8 only)
    auto decode = [&](int tile, int& img, int& py0, int& px0) {
        const int tx = tile % p.tiles_x;
        const int rest = tile / p.tiles_x;
        const int ty = rest % p.tiles_y;
        img = rest / p.tiles_y;
        py0 = ty * C::ROWS; px0 = tx * 8;
    };
    auto load_patch = [&](int tile) {
        int img, py0, px0;
        decode(tile, img, py0, px0);
        const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;       // conv region origin
        // uniform part of the address in scalar registers; per piece one 32-bit offset and four compares against scalars
        const size_t origin = ((size_t)img * p.H + 2 * oy0) * (size_t)p.W * 3 + 6 * ox0;
        const float* base = p.in + origin;
        const unsigned char* base8 = p.in_u8 + origin;
        const int ylo = -2 * oy0, yhi = p.H - 2 * oy0, xlo = -2 * ox0, xhi = p.W - 1 - 2 * ox0;
#pragma unroll
        for (int it = 0; it < P_IT; ++it) {
            // the 6 values in[iy][ix..ix+1][0..2] are contiguous; columns are valid in pairs (W is even)
            const bool ok = (pdst[it] >= 0) & (prow[it] >= ylo) & (prow[it] < yhi) & (pxo[it] >= xlo) & (pxo[it] < xhi);
            if constexpr (U8) {
                // an out-of-image piece is padding: zero in the NORMALISED image, not byte 0 -> bit 16 marks it for store_patch
                unsigned int r = 0x10000u;
                if (!(COBEVT_STEM_KNOCK & 8) && ok) r = *(const unsigned short*)(base8 + poff[it]);      // even offset: W is even
                praw[it] = r;
            } else {
                float2 v = make_float2(0.f, 0.f);
                if (!(COBEVT_STEM_KNOCK & 8) && ok) v = *(const float2*)(base + poff[it]);
                preg[it] = v;
            }
        }
    };
    auto store_patch = [&]() {
        if constexpr (U8) {
            // both table reads of every piece in flight before the first patch write (one LDS round trip, not P_IT)
#pragma unroll
            for (int it = 0; it < P_IT; ++it) {
                const unsigned int r = praw[it];
                const float a = lut_l[(plut[it] & 0xff) * 256 + (r & 0xff)], b = lut_l[(plut[it] >> 8) * 256 + ((r >> 8) & 0xff)];
                preg[it] = (r >> 16) ? make_float2(0.f, 0.f) : make_float2(a, b);
            }
        }
#pragma unroll
        for (int it = 0; it < P_IT; ++it)
            if (pdst[it] >= 0) {
                unsigned char* dst = patch + pdst[it];
                if constexpr (Elem<T>::kIsBf16) *(uint32_t*)dst = pack_bf2(preg[it].x, preg[it].y);
                else if constexpr (kXSplit<T>) {
                    // second / third library: the 16-byte operand piece {x0..x3} lives in LDS as {hi(x0,x1), hi(x2,x3), lo(x0,x1), lo(x2,x3)}
                    // (common.hpp stage_x_piece); this float pair is the first or the second half of its piece
                    uint32_t hi, lo;
                    split_pair_staged(preg[it].x, preg[it].y, hi, lo);
                    unsigned char* pb = patch + (pdst[it] & ~15) + ((pdst[it] >> 1) & 4);
                    *(uint32_t*)pb = hi;
                    *(uint32_t*)(pb + 8) = lo;
                } else *(float2*)dst = preg[it];
            }
    };

    int tile = blockIdx.x;
    if (tile >= p.ntiles) return;
    if constexpr (U8) {
        float* l = (float*)(smem + C::LDS);
        for (int i = tid; i < 768; i += NT) l[i] = p.lut[i];
    }
    {
        // the first image patch is requested before the weights, and the weight pieces of a thread are all in flight at
        // once (unconditional, row clamped): the rolled, branch-guarded loop paid one L2 round trip per iteration before the
        // patch load could even be issued
        load_patch(tile);
        // weight image in global memory: [64][16 taps][16 channels] (shared with stem7x7_kernel); compacted to [64][4][48] here
        constexpr int PIECES = 16 * 16 * KB / 16;             // 16-byte pieces of a cout row
        constexpr int PPT = PIECES / 16;                      // pieces per tap: bf16 2 (8 channels), fp32 4 (4 channels)
        constexpr int W_IT = (64 * PIECES + NT - 1) / NT;
        uint4 wv[W_IT];
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * NT;
            const int row = i / PIECES, j = i - row * PIECES;
            wv[it] = *(const uint4*)((const unsigned char*)p.wgt + ((size_t)(row < p.Cout ? row : 0) * PIECES + (i < 64 * PIECES ? j : 0)) * 16);
        }
#pragma unroll
        for (int it = 0; it < W_IT; ++it) {
            const int i = tid + it * NT;
            const int row = i / PIECES, j = i - row * PIECES;
            const int tap = j / PPT, part = j - tap * PPT;
            if (i < 64 * PIECES) {
                const uint4 v = row < p.Cout ? wv[it] : make_uint4(0, 0, 0, 0);
                unsigned char* d = wl + row * C::WROW + ((tap >> 2) * 48 + (tap & 3) * 12) * KB + part * 16;
                if constexpr (Elem<T>::kIsBf16) {             // channels 0..7 (two 8-byte halves), then 8..11
                    *(uint2*)d = make_uint2(v.x, v.y);
                    if (part == 0) *(uint2*)(d + 8) = make_uint2(v.z, v.w);
                } else if (part < 3) {
                    *(uint4*)d = stage_w_piece<T>(v);
                }
            }
        }
    }

    // this wave's unit: pixel tile wave >> 1, cout half wave & 1
    const int chalf = wave & 1;
    const int rpix = (wave >> 1) * 32 + ql;                   // region pixel of this lane's accumulator column
    const int rpc = rpix < RPIX ? rpix : RPIX - 1;
    const int ry = rpc / RW, rx = rpc - ry * RW;
    const int abase = ry * C::PROW + rx * C::PIX + h * 16;
    const int bbase = (chalf * 32 + ql) * C::WROW + h * 16;
    float* bias_l = (float*)(smem + C::PATCH + C::WBYTES + C::STAGE);     // bias in LDS: 32 VGPRs less than keeping it
    if (tid < 64) bias_l[tid] = p.bias ? p.bias[tid] : 0.f;
    __syncthreads();

    // the pooled chunk of a thread is stored one tile late, after the next patch has left its registers: vmcnt counts the
    // store behind the prefetch loads, so a store issued at the end of a tile made the patch wait at the top of the next one
    // wait for the write to complete
    constexpr int CH = Elem<T>::kChunk;
    constexpr int CPP = 64 / CH;
    static_assert((C::ROWS + 1) / 2 * 2 * 8 * CPP <= NT, "stem_pool_kernel: one pooled chunk per thread (rows in pairs)");
    uint4 pend = make_uint4(0, 0, 0, 0);
    T* pend_dst = nullptr;

    for (; tile < p.ntiles; tile += gridDim.x) {
        int img, py0, px0;
        decode(tile, img, py0, px0);
        const int oy0 = 2 * py0 - 1, ox0 = 2 * px0 - 1;
        COBEVT_ST_MARK(0);
        store_patch();
        if (pend_dst) *(uint4*)pend_dst = pend;
        __syncthreads();
        COBEVT_ST_MARK(1);
        const int next = tile + gridDim.x;
        if (next < p.ntiles) load_patch(next);
        COBEVT_ST_MARK(2);

        f32x16 acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        // 4 tap rows x STEPS_ROW 32-byte steps, operands one step ahead of the MFMAs in a register ring with the issue order
        // pinned (each MFMA had its two LDS reads right in front of it: 3.3k cycles for 32 MFMAs in the s_memtime trace; deeper
        // rings measured slower - the loop is bound by LDS throughput, not latency)
        constexpr int NS = C::NS;
        constexpr int RS = 2;
        uint4 ra[RS], rb[RS];
        auto fetch = [&](int slot, int i) {                  // compile-time i after unrolling
            const int a = i / C::STEPS_ROW, sub = i - a * C::STEPS_ROW;
            rb[slot] = *(const uint4*)(wl + bbase + a * 48 * KB + sub * 32);
            const unsigned char* pa = patch + abase + a * C::PROW + sub * 32;
            if constexpr (Elem<T>::kIsBf16) {                 // 8-byte aligned: two ds_read_b64 (see lds_read_2x8)
                ra[slot] = lds_read_2x8(pa);
            } else {
                ra[slot] = *(const uint4*)pa;
            }
        };
#pragma unroll
        for (int i = 0; i < RS - 1; ++i) fetch(i, i);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if (i + RS - 1 < NS && !(COBEVT_STEM_KNOCK & 1)) fetch((i + RS - 1) % RS, i + RS - 1);
            if constexpr (COBEVT_STEM_KNOCK & 2) asm volatile("" :: "v"(rb[i % RS].x), "v"(rb[i % RS].y), "v"(rb[i % RS].z), "v"(rb[i % RS].w), "v"(ra[i % RS].x), "v"(ra[i % RS].y), "v"(ra[i % RS].z), "v"(ra[i % RS].w));
            else
            mfma_kgroup_staged<T>(rb[i % RS], ra[i % RS], acc);      // D = W . X^T: lane <-> pixel, registers <-> couts
            __builtin_amdgcn_sched_barrier(0);
        }
        COBEVT_ST_MARK(3);
        // ---- bias + ReLU, rounded to T, staged [region pixel][64]; conv pixels outside the map -> 0
        if (rpix < RPIX && !(COBEVT_STEM_KNOCK & 16)) {
            const int oy = oy0 + ry, ox = ox0 + rx;
            const bool inside = oy >= 0 && oy < p.Ho && ox >= 0 && ox < p.Wo;
            float4 bq[4];                                     // the four reads in flight together (one LDS round trip, not four)
#pragma unroll
            for (int k = 0; k < 4; ++k) bq[k] = *(const float4*)(bias_l + chalf * 32 + 8 * k + 4 * h);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int c0 = chalf * 32 + 8 * k + 4 * h;
                const float4 b = bq[k];
                float v[4] = {acc[4 * k] + b.x, acc[4 * k + 1] + b.y, acc[4 * k + 2] + b.z, acc[4 * k + 3] + b.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e], 0.f) : 0.f;
                unsigned char* d = stage + rpix * C::SROW + c0 * Elem<T>::kBytes;
                if constexpr (Elem<T>::kIsBf16) *(uint2*)d = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        COBEVT_ST_MARK(4);
        __syncthreads();      // staging complete; every wave is done with the patch
        COBEVT_ST_MARK(5);
        // ---- 3 x 3 / stride 2 maxima: pooled pixel (qy, qx) of the ROWS x 8 tile covers region rows 2qy..2qy+2, cols 2qx..2qx+2
        pend_dst = nullptr;
        // lane -> (pooled pixel, 16-byte chunk).  bf16 (8 chunks per pixel): a ds_read_b128 is serviced in the 16-lane groups
        // {0-3, 12-15, 20-27} and {4-11, 16-19, 28-31} (+32), i.e. chunk halves of the four pixels of a half-wave in the pattern
        // (lo, hi, hi, lo) / (hi, lo, lo, hi); staging pixel (qy, qx) starts at bank 8 (qy + qx) mod 64, so the four pixels are
        // chosen with qy + qx = s, s, s + 4, s + 4 (mod 8) from a pair of pooled rows: banks [0,16) [16,32) [48,64) [32,48).
        // fp32 (16 chunks): a 16-lane group is one whole pixel = all 64 banks, any order works.
        int q_y, q_x, cj;
        if constexpr (Elem<T>::kIsBf16) {
            const int hw = tid >> 5, j = (tid >> 3) & 3, t = hw & 3;
            q_y = 2 * (hw >> 2) + (j & 1);
            q_x = j == 0 ? t : j == 1 ? ((t + 7) & 7) : j == 2 ? t + 4 : t + 3;
            cj = tid & 7;
        } else {
            const int q = tid / CPP;
            q_y = q >> 3; q_x = q & 7; cj = tid - q * CPP;
        }
        if (q_y < C::ROWS) {
            const int qy = q_y, qx = q_x;
            const int py = py0 + qy, px = px0 + qx;
            // the nine window reads are issued together (one LDS round trip; value by value the compiler waited after each)
            uint4 wv[9];
#pragma unroll
            for (int dy = 0; dy < 3; ++dy)
#pragma unroll
                for (int dx = 0; dx < 3; ++dx) {
                    wv[dy * 3 + dx] = *(const uint4*)(stage + ((2 * qy + dy) * RW + 2 * qx + dx) * C::SROW + cj * 16);
                    if constexpr (COBEVT_STEM_KNOCK & 4) wv[dy * 3 + dx] = make_uint4(dy, dx, dy, dx);
                }
            __builtin_amdgcn_sched_barrier(0);
            // staged values are >= +0 after the ReLU (or the -0 a ReLU of -0 may leave): as SIGNED integers their bit patterns
            // order like the values and -0 sorts below +0, so the maximum runs on the raw words - packed 16-bit for bf16 -
            // starting from +0 exactly like the float maximum it replaces
            uint4 m = make_uint4(0, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                m.x = word_max<T>(m.x, wv[i].x); m.y = word_max<T>(m.y, wv[i].y);
                m.z = word_max<T>(m.z, wv[i].z); m.w = word_max<T>(m.w, wv[i].w);
            }
            pend = m;
            if (py < Hp && px < Wp) pend_dst = (T*)p.out + (((size_t)img * Hp + py) * Wp + px) * 64 + cj * CH;
        }
        COBEVT_ST_MARK(6);
        __syncthreads();
        COBEVT_ST_MARK(7);
    }
    if (pend_dst) *(uint4*)pend_dst = pend;
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
    p.in = in; p.in_u8 = nullptr; p.lut = nullptr; p.wgt = wgt; p.bias = bias; p.out = out;
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
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)stem7x7_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemCfg<bf16_t>::LDS);
        (void)hipFuncSetAttribute((const void*)stem7x7_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemCfg<float>::LDS);
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
    p.in = in; p.in_u8 = nullptr; p.lut = nullptr; p.wgt = wgt; p.bias = bias; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cout = 64; p.act = 1;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 4 || p.W < 4 || (p.H & 3) || (p.W & 3)) return COBEVT_ERR_SHAPE;
    p.Ho = p.H / 2; p.Wo = p.W / 2;
    const int rows = dtype == 0 ? StemPoolCfg<bf16_t>::ROWS : StemPoolCfg<float>::ROWS;
    p.tiles_y = (p.Ho / 2 + rows - 1) / rows; p.tiles_x = (p.Wo / 2 + 7) / 8; p.tiles_n = 1;
    const long nt = (long)p.N * p.tiles_y * p.tiles_x;
    if (nt > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    p.ntiles = (int)nt;
    const size_t lds = dtype == 0 ? StemPoolCfg<bf16_t>::LDS : StemPoolCfg<float>::LDS;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<bf16_t>::LDS);
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<float>::LDS);
    }
    // persistent grid: as many workgroups per CU as LDS and the 32-wave limit admit (bf16 2, fp32 1)
    const int nthreads = dtype == 0 ? StemPoolCfg<bf16_t>::NT : StemPoolCfg<float>::NT;
    int per_cu = (int)(163840 / lds);
    if (per_cu > 2048 / nthreads) per_cu = 2048 / nthreads;
    if (per_cu < 1) per_cu = 1;
    const unsigned blocks = (unsigned)(nt < 256L * per_cu ? nt : 256L * per_cu);
    if (dtype == 0) hipLaunchKernelGGL(stem_pool_kernel<bf16_t>, dim3(blocks), dim3(StemPoolCfg<bf16_t>::NT), lds, stream, p);
    else hipLaunchKernelGGL(stem_pool_kernel<float>, dim3(blocks), dim3(StemPoolCfg<float>::NT), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h: the same launch on uint8 camera frames (N, H, W, 3) + the [3][256] normalisation table
extern "C" int cobevt_stem_conv7x7s2_pool_u8(const unsigned char* in, const float* lut, const void* wgt, const float* bias, void* out,
                                             const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W]   (Cout = 64, ReLU, H and W multiples of 4)
    if (!in || !lut || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    StemParams p;
    const int dtype = dims[0];
    p.in = nullptr; p.in_u8 = in; p.lut = lut; p.wgt = wgt; p.bias = bias; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.Cout = 64; p.act = 1;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.H < 4 || p.W < 4 || (p.H & 3) || (p.W & 3)) return COBEVT_ERR_SHAPE;
    p.Ho = p.H / 2; p.Wo = p.W / 2;
    const int rows = dtype == 0 ? StemPoolCfg<bf16_t>::ROWS : StemPoolCfg<float>::ROWS;
    p.tiles_y = (p.Ho / 2 + rows - 1) / rows; p.tiles_x = (p.Wo / 2 + 7) / 8; p.tiles_n = 1;
    const long nt = (long)p.N * p.tiles_y * p.tiles_x;
    if (nt > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    p.ntiles = (int)nt;
    const size_t lds = dtype == 0 ? StemPoolCfg<bf16_t>::LDS_U8 : StemPoolCfg<float>::LDS_U8;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<bf16_t>::LDS_U8);
        (void)hipFuncSetAttribute((const void*)stem_pool_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)StemPoolCfg<float>::LDS_U8);
    }
    const int nthreads = dtype == 0 ? StemPoolCfg<bf16_t>::NT : StemPoolCfg<float>::NT;
    int per_cu = (int)(163840 / lds);
    if (per_cu > 2048 / nthreads) per_cu = 2048 / nthreads;
    if (per_cu < 1) per_cu = 1;
    const unsigned blocks = (unsigned)(nt < 256L * per_cu ? nt : 256L * per_cu);
    if (dtype == 0) hipLaunchKernelGGL((stem_pool_kernel<bf16_t, true>), dim3(blocks), dim3(StemPoolCfg<bf16_t>::NT), lds, stream, p);
    else hipLaunchKernelGGL((stem_pool_kernel<float, true>), dim3(blocks), dim3(StemPoolCfg<float>::NT), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

#ifdef COBEVT_STEM_TRACE
extern "C" int cobevt_stem_read_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_stem_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif
