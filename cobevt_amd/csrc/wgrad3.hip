// Weight gradient of a 3x3 / stride-1 / pad-1 convolution (bf16 operands, fp32 result) straight from the channels-last maps (gfx950):
//     dW[o][c][ta][tb] = sum over (n, oy, ox) of dY[n][oy][ox][o] * X[n][oy + ta - 1][ox + tb - 1][c]
// - what cuDNN's backward-filter does for the ResNet / decoder convolutions under train_camera.py:143-179 (resnet_ms.py:67-74,
// naive_decoder.py:78-87).  The matrix instruction contracts 16 PIXELS and wants a lane's 8 pixels of ITS channel in one register quad,
// while memory holds a pixel's channels together.  cobevt_conv_wgrad_blocked reads a blocked (transposed) copy of both maps that a second
// kernel writes per convolution and step, one 32 x 32 (cout, cin) tile x tap row per workgroup with every operand fetched from L2 by
// every tile that needs it: 85 us + 2 x 10 us of copies per ResNet convolution at 5 agents, a fifth of the captured training step.
// Here the transposition is the LDS read: a wave copies its 32-channel slices of one dY row segment and of the three X rows around it
// into LDS as they lie in memory ([pixel][32 channels], 64-byte rows: four pixel rows fill the 64 banks) and ds_read_b64_tr_b16 hands
// each lane 4 consecutive pixels of its channel (lane i of a 16-lane group supplies the address of pixel i / 4, channels 4 (i % 4) ..+3 and
// receives channel i of the group's four pixels; tools/_probe/tr_read_probe.hip).  One wave = one 32 x 32 (cout, cin) tile with all NINE
// taps as accumulators: per 16 pixels 2 transposing reads for dY, 3 per X row (12 pixels: the three horizontal taps are the same
// registers shifted by one pixel, v_alignbit for the middle one) and 9 MFMAs.  The X rows roll through a three-slot ring (one new row
// per output row); everything is wave-private, so there is no barrier in the loop.  The four waves of a workgroup split the pixels; their
// accumulators are added through LDS and the workgroup's tile goes to a partial buffer [chunk][tile] that a second launch sums into dW
// (plain stores: deterministic, no zero fill of dW, no atomics).
#include "common.hpp"

namespace cobevt {
namespace {

typedef short v4s __attribute__((ext_vector_type(4)));

struct Wgrad3Params {
    const bf16_t* x;     // (N, H, W, Cin)
    const bf16_t* dy;    // (N, H, W, Cout)
    float* part;         // [nchunks][Cout/32][Cin/32][9][16][64]
    int N, H, W, Cin, Cout;
    int nseg;            // W / WS column segments
    int units;           // N * nseg * H  (image, segment, row) triples, rows fastest
    int nchunks;
};

__device__ __forceinline__ uint2 tr_read(const unsigned char* lds) {
    const v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s __attribute__((address_space(3)))*)lds);
    return __builtin_bit_cast(uint2, r);
}

template <int WS>
__global__ __launch_bounds__(256, 2) void conv_wgrad3_tr_kernel(Wgrad3Params p) {
    constexpr int XROW = (WS + 2) * 64;                 // bytes of one staged X row: pixels seg * WS - 1 .. seg * WS + WS, 32 channels
    constexpr int DROW = WS * 64;
    constexpr int WAVE_LDS = 3 * XROW + DROW;
    constexpr int XP = (WS + 2) * 4, DP = WS * 4;       // 16-byte pieces per row
    constexpr int XR = (XP + 63) / 64, DR = (DP + 63) / 64;
    constexpr int NSTEP = WS / 16;
    constexpr int RED = 3 * 16 * 64 * 4;                // the cross-wave reduction of one tap
    constexpr int LDS = 4 * WAVE_LDS > RED ? 4 * WAVE_LDS : RED;
    __shared__ __attribute__((aligned(16))) unsigned char smem[LDS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    unsigned char* xs = smem + wave * WAVE_LDS;
    unsigned char* ds = xs + 3 * XROW;
    const int to = blockIdx.x, tc = blockIdx.y, chunk = blockIdx.z;
    const int o0 = to * 32, c0 = tc * 32;
    // this wave's share of the (image, segment, row) list
    const int nsub = 4 * p.nchunks, sub = chunk * 4 + wave;
    // (wave-uniform by construction; readfirstlane tells the compiler, so the row loop branches on scalars)
    const int u0 = __builtin_amdgcn_readfirstlane((int)((long)p.units * sub / nsub));
    const int u1 = __builtin_amdgcn_readfirstlane((int)((long)p.units * (sub + 1) / nsub));
    // operand addressing: lane = 16 g + i; group g: channels 16 (g & 1) .., pixels 8 (g >> 1) ..; lane i -> pixel i / 4, channels 4 (i % 4) ..
    const int g = lane >> 4, i = lane & 15;
    const int lane_off = (8 * (g >> 1) + (i >> 2)) * 64 + (16 * (g & 1) + 4 * (i & 3)) * 2;

    f32x16 acc[9];
#pragma unroll
    for (int t = 0; t < 9; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    // piece m (lane + 64 m) of one X row / one dY row of this wave: global -> register (zeros outside the map) and register -> LDS.  Scalars,
    // not arrays: loop-carried register arrays filled by a lambda went through scratch, with a vmcnt(0) right behind the loads.
    // (the zero padding is applied when the piece is STORED: a select at fetch time makes the compiler wait for the load in the
    // middle of the matrix instructions that should cover it)
    auto fetch_x = [&](int m, int n, int seg, int iy, unsigned& okmask) __attribute__((always_inline)) -> uint4 {
        const int q = lane + 64 * m;
        const int j = q >> 2, pc = q & 3;
        const int ix = seg * WS - 1 + j;
        const bool ok = q < XP && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        okmask = ok ? (okmask | (1u << m)) : (okmask & ~(1u << m));
        const size_t off = (((size_t)n * p.H + (ok ? iy : 0)) * p.W + (ok ? ix : 0)) * p.Cin + c0 + pc * 8;
        return *(const uint4*)(p.x + off);                            // unconditional load from a clamped address
    };
    auto store_x = [&](int m, const uint4& t, int slot, unsigned okmask) __attribute__((always_inline)) {
        const int q = lane + 64 * m;
        const bool ok = (okmask >> m) & 1u;
        if (q < XP) *(uint4*)(xs + slot * XROW + q * 16) = make_uint4(ok ? t.x : 0u, ok ? t.y : 0u, ok ? t.z : 0u, ok ? t.w : 0u);
    };
    auto fetch_d = [&](int m, int n, int seg, int oy) __attribute__((always_inline)) -> uint4 {
        const int q = lane + 64 * m;
        const int j = q >> 2, pc = q & 3;
        const size_t off = (((size_t)n * p.H + oy) * p.W + seg * WS + (q < DP ? j : 0)) * p.Cout + o0 + pc * 8;
        return *(const uint4*)(p.dy + off);
    };
    auto store_d = [&](int m, const uint4& v) __attribute__((always_inline)) {
        const int q = lane + 64 * m;
        if (q < DP) *(uint4*)(ds + q * 16) = v;
    };
    static_assert(XR <= 3 && DR <= 2, "pieces per lane");
#define WG3_FETCH_X(iy) { nx0 = fetch_x(0, n, seg, iy, okm); if (XR > 1) nx1 = fetch_x(1, n, seg, iy, okm); if (XR > 2) nx2 = fetch_x(2, n, seg, iy, okm); }
#define WG3_STORE_X(slot) { store_x(0, nx0, slot, okm); if (XR > 1) store_x(1, nx1, slot, okm); if (XR > 2) store_x(2, nx2, slot, okm); }
#define WG3_FETCH_D(oy_) { nd0 = fetch_d(0, n, seg, oy_); if (DR > 1) nd1 = fetch_d(1, n, seg, oy_); }
#define WG3_STORE_D() { store_d(0, nd0); if (DR > 1) store_d(1, nd1); }

    uint4 nx0 = make_uint4(0, 0, 0, 0), nx1 = nx0, nx2 = nx0, nd0 = nx0, nd1 = nx0;
    unsigned okm = 0;
    bool have_next = false;
    for (int u = u0; u < u1; ++u) {
        __builtin_amdgcn_sched_barrier(0);              // the stores below (and their selects) stay behind the previous row's MFMAs

        const int oy = u % p.H, strip = u / p.H;
        const int seg = strip % p.nseg, n = strip / p.nseg;
        // X row iy lives in ring slot (iy + 1) % 3; this row needs oy - 1, oy, oy + 1
        if (!have_next) {                               // first row of this wave's share, or of an (image, segment) strip
            WG3_FETCH_X(oy - 1);
            WG3_STORE_X(oy % 3);
            WG3_FETCH_X(oy);
            WG3_STORE_X((oy + 1) % 3);
            WG3_FETCH_X(oy + 1);
            WG3_FETCH_D(oy);
        }
        WG3_STORE_X((oy + 2) % 3);                      // row oy + 1 (fetched during the previous row)
        WG3_STORE_D();
        // the next row of the same strip: its new X row and its dY row travel while this row is computed (fetched unconditionally; at the
        // end of a strip the data is dropped)
        have_next = u + 1 < u1 && oy + 1 < p.H;
        WG3_FETCH_X(oy + 2);
        WG3_FETCH_D(oy + 1 < p.H ? oy + 1 : oy);
        __builtin_amdgcn_sched_barrier(0);              // ... and the prefetch is issued before the first of this row's
        const unsigned char* xrow[3] = {xs + (oy % 3) * XROW + lane_off, xs + ((oy + 1) % 3) * XROW + lane_off,
                                        xs + ((oy + 2) % 3) * XROW + lane_off};
        const unsigned char* drow = ds + lane_off;
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            const uint2 a0 = tr_read(drow + (16 * s) * 64), a1 = tr_read(drow + (16 * s + 4) * 64);
            const bf16x8 af = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
#pragma unroll
            for (int ta = 0; ta < 3; ++ta) {
                // staged pixel j = map pixel seg * WS - 1 + j: the 12 pixels p - 1 .. p + 10 around this lane's 8 (p = 16 s + 8 (g >> 1))
                const uint2 q0 = tr_read(xrow[ta] + (16 * s) * 64), q1 = tr_read(xrow[ta] + (16 * s + 4) * 64),
                            q2 = tr_read(xrow[ta] + (16 * s + 8) * 64);
                const uint4 b0 = make_uint4(q0.x, q0.y, q1.x, q1.y);                                   // tap column 0: pixels p - 1 .. p + 6
                const uint4 b1 = make_uint4(__builtin_amdgcn_alignbit(q0.y, q0.x, 16), __builtin_amdgcn_alignbit(q1.x, q0.y, 16),
                                            __builtin_amdgcn_alignbit(q1.y, q1.x, 16), __builtin_amdgcn_alignbit(q2.x, q1.y, 16));   // p .. p + 7
                const uint4 b2 = make_uint4(q0.y, q1.x, q1.y, q2.x);                                   // p + 1 .. p + 8
                acc[ta * 3 + 0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b0), acc[ta * 3 + 0], 0, 0, 0);
                acc[ta * 3 + 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b1), acc[ta * 3 + 1], 0, 0, 0);
                acc[ta * 3 + 2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, b2), acc[ta * 3 + 2], 0, 0, 0);
            }
        }
    }
    // the four waves' accumulators -> one tile per workgroup
    float* red = (float*)smem;
    float* dst = p.part + (((size_t)chunk * gridDim.x + to) * gridDim.y + tc) * (9 * 16 * 64);
    __syncthreads();                                    // every wave is done with its staging area
#pragma unroll
    for (int t = 0; t < 9; ++t) {
        if (t) __syncthreads();
        if (wave > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r)
                dst[(t * 16 + r) * 64 + lane] = acc[t][r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
        }
    }
}

// dw[(o * Cin + c) * 9 + t] = sum over the chunks of part[chunk][o / 32][c / 32][t][r][lane], (o % 32, c % 32) = (acc_row(r, lane), lane % 32).
// A workgroup sums 64 consecutive partial words: its four waves take every fourth chunk (four loads in flight each - up to 128 chunks
// for the 64-channel layers; one thread walking them serially cost 18 us per launch) and meet through LDS.
__global__ __launch_bounds__(256) void conv_wgrad3_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nchunks, long tile_elems,
                                                                 int TC, int Cin) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;               // tile_elems is a multiple of 1024
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = wave;
    for (; c + 12 < nchunks; c += 16) {
        s0 += part[(size_t)c * tile_elems + e];
        s1 += part[(size_t)(c + 4) * tile_elems + e];
        s2 += part[(size_t)(c + 8) * tile_elems + e];
        s3 += part[(size_t)(c + 12) * tile_elems + e];
    }
    for (; c < nchunks; c += 4) s0 += part[(size_t)c * tile_elems + e];
    float s = (s0 + s1) + (s2 + s3);
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave > 0) return;
    s += red[0][lane] + red[1][lane] + red[2][lane];
    const int r = (int)((e >> 6) & 15);
    long t = e >> 10;
    const int tap = (int)(t % 9);
    t /= 9;
    const int tc = (int)(t % TC), to = (int)(t / TC);
    const int o = to * 32 + acc_row(r, lane), ci = tc * 32 + (lane & 31);
    dw[((size_t)o * Cin + ci) * 9 + tap] = s;
}

// ---------------------------------------------------------------------------------------------------------------------
// Weight gradient of a dense projection / 1x1 convolution, dW[o][c] = sum over the rows of dY[row][o] X[row][c] (bf16 rows, 128 | Cin,
// 128 | Cout): nn.Linear's and the 1x1 convolutions' backward-filter under train_camera.py:143-179 (the FAX / fusion projections and
// MLPs, base_transformer.py:102-124, fax_modules.py:189-237).  A handful of FLOPs per byte: the kernel is a stream over X and dY, so a
// workgroup owns a whole 128 x 128 output tile and every row slice is read once per tile.  16 waves = 4 row groups x (2 x 2) waves of
// 64 x 64: a row group copies its 16 rows of both slices into LDS as they lie in memory ([row][128 channels], the 32-byte blocks of a row
// XOR-ed with 2 (row % 4) so that the four rows of a transposing read cover the 64 banks) and ds_read_b64_tr_b16 hands out the pixel-major
// operands; 8 reads and 4 MFMAs per wave and step, the next step's rows in flight meanwhile (two LDS buffers, one barrier per 64 rows).
// The four row groups meet through LDS, the workgroup's tile goes to the partial buffer and the reduce launch above's sibling sums it.
struct Wgrad1Params {
    const bf16_t* x;     // (R, Cin)
    const bf16_t* dy;    // (R, Cout)
    float* part;         // [nchunks][tiles][4 waves][4 accumulators][16][64]
    long R;
    int Cin, Cout, TCn;  // TCn = Cin / 128
    int nchunks;
};

__global__ __launch_bounds__(1024, 1) void linear_wgrad_tr_kernel(Wgrad1Params p) {
    constexpr int GRP = 2 * 16 * 256;                   // one row group's buffer: 16 rows of dY's slice, 16 of X's
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * 4 * GRP];          // 64 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int kg = wave >> 2, wo = (wave >> 1) & 1, wc = wave & 1;
    const int gt = tid & 255;                           // thread within the row group
    const int tile = blockIdx.x, chunk = blockIdx.y;
    const int o_base = (tile / p.TCn) * 128, c_base = (tile % p.TCn) * 128;
    const long steps = (p.R + 63) / 64;
    const long s0 = steps * chunk / p.nchunks, s1 = steps * (chunk + 1) / p.nchunks;
    // staging: thread gt copies piece (gt & 15) of row (gt >> 4) of each slice
    const int srow = gt >> 4, sj = gt & 15;
    const int st_off = srow * 256 + ((((sj >> 1) ^ (2 * (srow & 3))) << 5) | ((sj & 1) << 4));
    // operands: lane = 16 g + i -> row 8 (g >> 1) + i / 4 (+ 4 for the second read), 16-channel block cblk ^ 2 (i / 4), channels 4 (i % 4) ..
    const int g = lane >> 4, i = lane & 15;
    const int rowoff = (8 * (g >> 1) + (i >> 2)) * 256 + 8 * (i & 3);
    int aoff[2], boff[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        aoff[t] = rowoff + (((4 * wo + 2 * t + (g & 1)) ^ (2 * (i >> 2))) << 5);
        boff[t] = 16 * 256 + rowoff + (((4 * wc + 2 * t + (g & 1)) ^ (2 * (i >> 2))) << 5);
    }
    f32x16 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    uint4 nd = make_uint4(0, 0, 0, 0), nx = nd;
    bool nok = false;
    auto fetch = [&](long step) __attribute__((always_inline)) {
        const long row = step * 64 + kg * 16 + srow;
        nok = row < p.R;
        const long rr = nok ? row : 0;
        nd = *(const uint4*)(p.dy + rr * p.Cout + o_base + sj * 8);
        nx = *(const uint4*)(p.x + rr * p.Cin + c_base + sj * 8);
    };
    if (s0 < s1) fetch(s0);
    for (long s = s0; s < s1; ++s) {
        unsigned char* buf = smem + ((int)((s - s0) & 1) * 4 + kg) * GRP;
        *(uint4*)(buf + st_off) = make_uint4(nok ? nd.x : 0u, nok ? nd.y : 0u, nok ? nd.z : 0u, nok ? nd.w : 0u);
        *(uint4*)(buf + 16 * 256 + st_off) = make_uint4(nok ? nx.x : 0u, nok ? nx.y : 0u, nok ? nx.z : 0u, nok ? nx.w : 0u);
        __syncthreads();                                // this step's rows are in LDS; the other buffer is free again
        if (s + 1 < s1) fetch(s + 1);
        __builtin_amdgcn_sched_barrier(0);
        bf16x8 af[2], bfr[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const uint2 a0 = tr_read(buf + aoff[t]), a1 = tr_read(buf + aoff[t] + 4 * 256);
            const uint2 b0 = tr_read(buf + boff[t]), b1 = tr_read(buf + boff[t] + 4 * 256);
            af[t] = __builtin_bit_cast(bf16x8, make_uint4(a0.x, a0.y, a1.x, a1.y));
            bfr[t] = __builtin_bit_cast(bf16x8, make_uint4(b0.x, b0.y, b1.x, b1.y));
        }
#pragma unroll
        for (int to = 0; to < 2; ++to)
#pragma unroll
            for (int tc = 0; tc < 2; ++tc)
                acc[to * 2 + tc] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[to], bfr[tc], acc[to * 2 + tc], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
    }
    // the four row groups -> one tile: per accumulator, groups 1..3 park theirs in LDS (3 x 4 waves x 4 KiB), group 0 adds and stores
    float* red = (float*)smem;
    float* dst = p.part + (((size_t)chunk * gridDim.x + tile) * 4 + (wave & 3)) * (4 * 16 * 64);
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        __syncthreads();
        if (kg > 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(((kg - 1) * 4 + (wave & 3)) * 16 + r) * 64 + lane] = acc[t][r];
        }
        __syncthreads();
        if (kg == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int k = ((wave & 3) * 16 + r) * 64 + lane;
                dst[(t * 16 + r) * 64 + lane] = acc[t][r] + red[k] + red[4 * 16 * 64 + k] + red[2 * 4 * 16 * 64 + k];
            }
        }
    }
}

// dw[o * Cin + c] = sum over the chunks of part[chunk][tile][wave][acc][r][lane]
__global__ __launch_bounds__(256) void linear_wgrad_reduce_kernel(const float* __restrict__ part, float* __restrict__ dw, int nchunks, long elems,
                                                                  int TCn, int Cin) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long e = (long)blockIdx.x * 64 + lane;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    int c = wave;
    for (; c + 12 < nchunks; c += 16) {
        s0 += part[(size_t)c * elems + e];
        s1 += part[(size_t)(c + 4) * elems + e];
        s2 += part[(size_t)(c + 8) * elems + e];
        s3 += part[(size_t)(c + 12) * elems + e];
    }
    for (; c < nchunks; c += 4) s0 += part[(size_t)c * elems + e];
    float s = (s0 + s1) + (s2 + s3);
    if (wave > 0) red[wave - 1][lane] = s;
    __syncthreads();
    if (wave > 0) return;
    s += red[0][lane] + red[1][lane] + red[2][lane];
    const int r = (int)((e >> 6) & 15), a = (int)((e >> 10) & 3), w = (int)((e >> 12) & 3);
    const int tile = (int)(e >> 14);
    const int o = (tile / TCn) * 128 + ((w >> 1) & 1) * 64 + (a >> 1) * 32 + acc_row(r, lane);
    const int ci = (tile % TCn) * 128 + (w & 1) * 64 + (a & 1) * 32 + (lane & 31);
    dw[(size_t)o * Cin + ci] = s;
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_conv_wgrad3_chunks(const int* dims) {
    // dims: [N, H, W, Cin, Cout]: the number of partial tiles per weight element cobevt_conv_wgrad3 will write (scratch = that many
    // copies of the (Cout, Cin, 3, 3) fp32 gradient), or a negative error code when the shape is not served
    if (!dims) return -COBEVT_ERR_ARG;
    const int N = dims[0], H = dims[1], W = dims[2], Cin = dims[3], Cout = dims[4];
    if (N < 1 || H < 1 || W < 16 || W % 16 || Cin < 32 || Cin % 32 || Cout < 32 || Cout % 32) return -COBEVT_ERR_UNSUPPORTED;
    const int WS = W % 32 == 0 ? 32 : 16;
    const long units = (long)N * (W / WS) * H;
    if (units >= 0x7fffffffL || (long)N * H * W * (Cin > Cout ? Cin : Cout) >= 0x7fffffffL * 4) return -COBEVT_ERR_UNSUPPORTED;
    const long tiles = (long)(Cout / 32) * (Cin / 32);
    long chunks = (512 + tiles - 1) / tiles;                // about 512 workgroups (two per CU)
    if (chunks > units / 8) chunks = units / 8;             // at least two rows per wave
    if (chunks < 1) chunks = 1;
    if (chunks > 65535 || Cin / 32 > 65535) return -COBEVT_ERR_UNSUPPORTED;
    return (int)chunks;
}

extern "C" int cobevt_conv_wgrad3(const void* x, const void* dy, float* dw, float* scratch, const int* dims, hipStream_t stream) {
    // dims: [N, H, W, Cin, Cout, nchunks]
    if (!x || !dy || !dw || !scratch || !dims) return COBEVT_ERR_ARG;
    const int chunks = cobevt_conv_wgrad3_chunks(dims);
    if (chunks < 0) return -chunks;
    if (dims[5] != chunks) return COBEVT_ERR_ARG;
    Wgrad3Params p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.part = scratch;
    p.N = dims[0]; p.H = dims[1]; p.W = dims[2]; p.Cin = dims[3]; p.Cout = dims[4];
    const int WS = p.W % 32 == 0 ? 32 : 16;
    p.nseg = p.W / WS;
    p.units = p.N * p.nseg * p.H;
    p.nchunks = chunks;
    const dim3 grid(p.Cout / 32, p.Cin / 32, chunks);
    if (WS == 32) hipLaunchKernelGGL(conv_wgrad3_tr_kernel<32>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv_wgrad3_tr_kernel<16>, grid, dim3(256), 0, stream, p);
    const long tile_elems = (long)p.Cout * p.Cin * 9;
    hipLaunchKernelGGL(conv_wgrad3_reduce_kernel, dim3((unsigned)(tile_elems / 64)), dim3(256), 0, stream, scratch, dw, chunks, tile_elems,
                       p.Cin / 32, p.Cin);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_linear_wgrad_chunks(const long* dims) {
    // dims: [R, Cin, Cout]: partial copies of dw (Cout x Cin fp32) cobevt_linear_wgrad writes, or minus an error code (shape not served)
    if (!dims) return -COBEVT_ERR_ARG;
    const long R = dims[0], Cin = dims[1], Cout = dims[2];
    if (R < 1 || Cin < 128 || Cin % 128 || Cout < 128 || Cout % 128 || Cin > 65536 || Cout > 65536) return -COBEVT_ERR_UNSUPPORTED;
    if (R * (Cin > Cout ? Cin : Cout) >= (1L << 40)) return -COBEVT_ERR_UNSUPPORTED;
    const long tiles = (Cout / 128) * (Cin / 128), steps = (R + 63) / 64;
    long chunks = (256 + tiles - 1) / tiles;                 // one 16-wave workgroup per CU
    if (chunks > steps / 4) chunks = steps / 4;              // at least four steps each
    if (chunks < 1) chunks = 1;
    if (chunks > 65535 || tiles > 0x7fffffffL) return -COBEVT_ERR_UNSUPPORTED;
    return (int)chunks;
}

extern "C" int cobevt_linear_wgrad(const void* x, const void* dy, float* dw, float* scratch, const long* dims, hipStream_t stream) {
    // dims: [R, Cin, Cout, nchunks]
    if (!x || !dy || !dw || !scratch || !dims) return COBEVT_ERR_ARG;
    const int chunks = cobevt_linear_wgrad_chunks(dims);
    if (chunks < 0) return -chunks;
    if (dims[3] != chunks) return COBEVT_ERR_ARG;
    Wgrad1Params p;
    p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.part = scratch;
    p.R = dims[0]; p.Cin = (int)dims[1]; p.Cout = (int)dims[2]; p.TCn = p.Cin / 128;
    p.nchunks = chunks;
    const int tiles = (p.Cout / 128) * p.TCn;
    hipLaunchKernelGGL(linear_wgrad_tr_kernel, dim3(tiles, chunks), dim3(1024), 0, stream, p);
    const long elems = (long)p.Cout * p.Cin;
    hipLaunchKernelGGL(linear_wgrad_reduce_kernel, dim3((unsigned)(elems / 64)), dim3(256), 0, stream, scratch, dw, chunks, elems, p.TCn, p.Cin);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
