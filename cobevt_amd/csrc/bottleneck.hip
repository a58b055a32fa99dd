// Fused torchvision Bottleneck (inplanes 128, planes 32, stride 1, no downsample) on MFMA (gfx950, bf16):
//
//     out = ReLU( conv1x1_3( ReLU( conv3x3_2( ReLU( conv1x1_1(x) + b1 ) ) + b2 ) ) + b3 + x )     (eval BatchNorms folded)
//
// = ResNetBottleNeck(dim) of the FAX pyramid: reference opv2v/opencood/models/sub_modules/fax_modules.py:10,472,512
// (`self.layers[i]`, two per level on the (b, 128, H, W) BEV map) with torchvision.models.resnet.Bottleneck.forward.
// As three launches (dense-row GEMM, 3x3 patch kernel, dense-row GEMM + residual) a block costs 30-40 us on the level-0 map
// although it moves 42 MB (10 us at the HBM rate) and 2.9 GFLOP: each launch is a dependent ~10-16 us step and the two
// 32-channel intermediates make a round trip through HBM.  Here a workgroup owns a TH x 16 output tile:
//   1. the three weight matrices (34 KB in MFMA fragment order) and biases are copied to LDS once;
//   2. conv1 runs on the (TH + 2) x 18 region conv2 needs - the B operand (pixel rows of x) straight from global memory, 8 x 16-byte
//      loads per lane in flight - + b1, ReLU, rounded to bf16 as the unfused path stores it, into an LDS patch [pixel][32 ch]
//      (zeros outside the image = conv2's padding);
//   3. each wave owns one 32-pixel output tile (2 rows x 16 columns): conv2 = 9 taps x 2 k-groups out of that patch; its result
//      (+ b2, ReLU, bf16) stays in REGISTERS and is the B operand of conv3 - the accumulator layout of D = W . X^T hands a lane
//      its pixel's channels in the order (r & 3) + 8 (r >> 2) + 4 half, so W3's fragments are stored with the contraction index
//      permuted the same way (host: ops.BottleneckPlan) and no shuffle is needed;
//   4. conv3 per 32-cout tile: + b3 (C operand) + residual + ReLU; a v_permlane32_swap per register pair turns the lane's 8-byte
//      channel runs into 16-byte ones, so residual loads and stores are 16 bytes per lane.
// No intermediate reaches HBM, one launch instead of three, one barrier.
#include "common.hpp"

namespace cobevt {

struct BottleneckParams {
    const void* in;
    const uint4* w1;        // [8 k-groups][64 lanes]            A fragments of W1 (32 x 128)
    const uint4* w2;        // [9 taps][2 k-groups][64 lanes]    A fragments of W2 (32 x 32 per tap)
    const uint4* w3;        // [4 cout tiles][2 k-groups][64]    A fragments of W3 (128 x 32), k permuted
    const float* b1;        // [32]
    const float* b2;        // [32]
    const float* b3;        // [128]
    void* out;
    int N, H, W;
    int tiles_y, tiles_x;
};

namespace {

constexpr int kC = 128, kMid = 32;
constexpr int kW1 = 8 * 1024, kW2 = 18 * 1024, kW3 = 8 * 1024;      // bytes of the fragment tables
constexpr int kP1Pitch = 80;                                          // bytes per region pixel in the conv1 patch: 64 + 16 pad

template <int TH> struct BnCfg {
    static constexpr int NW = TH / 2;                     // one wave per 32-pixel output tile
    static constexpr int NTHR = NW * 64;
    static constexpr int R1H = TH + 2, R1W = 18, R1 = R1H * R1W;
    static constexpr int N1 = (R1 + 31) / 32;             // conv1 pixel tiles of the region
    static constexpr int T1W = (N1 + NW - 1) / NW;
    static constexpr int P1 = R1 * kP1Pitch;
    static constexpr int LDS = kW1 + kW2 + kW3 + P1 + (kMid + kMid + kC) * 4;
};

__device__ __forceinline__ uint32_t relu_pack(float a, float b) { return pack_bf2(fmaxf(a, 0.f), fmaxf(b, 0.f)); }

template <int TH>
__global__ __launch_bounds__(BnCfg<TH>::NTHR, 2) void bottleneck_kernel(BottleneckParams p) {
    using G = BnCfg<TH>;
    constexpr int NTHR = G::NTHR, NW = G::NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* sw1 = smem;
    unsigned char* sw2 = smem + kW1;
    unsigned char* sw3 = smem + kW1 + kW2;
    unsigned char* p1 = smem + kW1 + kW2 + kW3;
    float* sb = (float*)(p1 + G::P1);                     // b1[32] | b2[32] | b3[128]

    int logical;
    {   // consecutive tiles on one XCD (shared halo rows in its L2)
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tx = logical % p.tiles_x;
    const int ty = (logical / p.tiles_x) % p.tiles_y;
    const int img = logical / (p.tiles_x * p.tiles_y);
    const int oy0 = ty * TH, ox0 = tx * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const bf16_t* in = (const bf16_t*)p.in;

    // ---- conv1 operands of this wave's first region tile: issued before anything else (the longest latency of the kernel)
    int pr[G::T1W];
    bool inside[G::T1W];
    uint4 xb[G::T1W][8];
#pragma unroll
    for (int t = 0; t < G::T1W; ++t) {
        const int tile = wave + t * NW;
        int r = tile * 32 + ql;
        if (r >= G::R1) r = G::R1 - 1;                    // padding lanes of the last tile compute a duplicate
        pr[t] = r;
        const int ry = r / G::R1W, rx = r - ry * G::R1W;
        const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
        inside[t] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
        const size_t pix = inside[t] ? ((size_t)img * p.H + iy) * p.W + ix : 0;       // clamped: unconditional loads
        const bf16_t* row = in + pix * kC + h * 8;
        if (tile < G::N1) {                               // wave-uniform
#pragma unroll
            for (int g = 0; g < 8; ++g) xb[t][g] = *(const uint4*)(row + g * 16);
        }
    }
    // ---- weights + biases -> LDS (fragment order: a wave's A operand is one contiguous 1-KB read)
    {
        // the three tables are consecutive in LDS; every thread requests ALL its pieces (and its bias value) before it stores the first:
        // as rolled copy loops (load, wait, store per iteration) the 34 KB cost nine serialised L2 round trips per workgroup - ~3 us of
        // the ~10 us a launch on a small map lasts (round 6)
        constexpr int N1_ = kW1 / 16, N2_ = kW2 / 16, TOT = (kW1 + kW2 + kW3) / 16, PER = (TOT + NTHR - 1) / NTHR;
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NTHR, ic = i < TOT ? i : TOT - 1;
            const uint4* s_ = ic < N1_ ? p.w1 + ic : (ic < N1_ + N2_ ? p.w2 + (ic - N1_) : p.w3 + (ic - N1_ - N2_));
            tmp[u] = *s_;
        }
        constexpr int NB = kMid + kMid + kC, PB = (NB + NTHR - 1) / NTHR;
        float bt[PB];
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int i = tid + u * NTHR, ic = i < NB ? i : NB - 1;
            bt[u] = ic < kMid ? p.b1[ic] : (ic < 2 * kMid ? p.b2[ic - kMid] : p.b3[ic - 2 * kMid]);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NTHR;
            if (i < TOT) *(uint4*)(smem + i * 16) = tmp[u];
        }
#pragma unroll
        for (int u = 0; u < PB; ++u) {
            const int i = tid + u * NTHR;
            if (i < NB) sb[i] = bt[u];
        }
    }
    __syncthreads();

    // bias fragments in accumulator layout: register r of a lane <-> channel (r & 3) + 8 (r >> 2) + 4 h of the 32-row tile
    auto bias_frag = [&](const float* b) {
        f32x16 f;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f32x4 v = *(const f32x4*)(b + 8 * k + 4 * h);
            f[4 * k] = v.x; f[4 * k + 1] = v.y; f[4 * k + 2] = v.z; f[4 * k + 3] = v.w;
        }
        return f;
    };

    // ---- conv1 (1x1, 128 -> 32) on the region, ReLU, bf16 -> patch [region pixel][32 channels]
    {
        const f32x16 b1f = bias_frag(sb);
#pragma unroll
        for (int t = 0; t < G::T1W; ++t) {
            if (wave + t * NW >= G::N1) continue;         // wave-uniform
            f32x16 acc = b1f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                const uint4 a = *(const uint4*)(sw1 + g * 1024 + lane * 16);
                mfma_kgroup<bf16_t>(a, xb[t][g], acc);    // D = W1 . X^T: lane <-> pixel, registers <-> mid channels
            }
            unsigned char* d = p1 + pr[t] * kP1Pitch + 8 * h;
#pragma unroll
            for (int k = 0; k < 4; ++k) {                 // channels 8k + 4h .. + 3
                uint2 w;
                w.x = inside[t] ? relu_pack(acc[4 * k], acc[4 * k + 1]) : 0u;      // outside the image: conv2's zero padding
                w.y = inside[t] ? relu_pack(acc[4 * k + 2], acc[4 * k + 3]) : 0u;
                *(uint2*)(d + 16 * k) = w;
            }
        }
    }
    // residual rows of this wave's output pixels: 16-byte pieces in the post-swap channel order (see the epilogue), in flight
    // under conv2.  Lane (ql, h): pixel (2 wave + (ql >> 4), ql & 15); piece j covers channels 16 j + 8 h' .. + 7 of cout tile
    // j >> 1 where h' = lane half after the swap = h.
    const int oy = oy0 + 2 * wave + (ql >> 4), ox = ox0 + (ql & 15);
    const bool o_ok = oy < p.H && ox < p.W;
    const size_t opix = o_ok ? ((size_t)img * p.H + oy) * p.W + ox : 0;
    uint4 res[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) res[j] = *(const uint4*)(in + opix * kC + 16 * j + 8 * h);
    __syncthreads();

    // ---- conv2 (3x3, 32 -> 32) for this wave's 2 x 16 pixels
    f32x16 acc2 = bias_frag(sb + kMid);
    {
        const unsigned char* pbase = p1 + ((2 * wave + (ql >> 4)) * G::R1W + (ql & 15)) * kP1Pitch + h * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int off = ((tap / 3) * G::R1W + (tap % 3)) * kP1Pitch;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const uint4 a = *(const uint4*)(sw2 + (tap * 2 + u) * 1024 + lane * 16);
                const uint4 b = *(const uint4*)(pbase + off + u * 32);
                mfma_kgroup<bf16_t>(a, b, acc2);
            }
        }
    }
    // y2 = ReLU(conv2 + b2) in bf16, as conv3's B operand: k-slot 8 h + j of k-block u <-> register 8 u + j
    uint4 y2[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        y2[u].x = relu_pack(acc2[8 * u + 0], acc2[8 * u + 1]);
        y2[u].y = relu_pack(acc2[8 * u + 2], acc2[8 * u + 3]);
        y2[u].z = relu_pack(acc2[8 * u + 4], acc2[8 * u + 5]);
        y2[u].w = relu_pack(acc2[8 * u + 6], acc2[8 * u + 7]);
    }
    // ---- conv3 (1x1, 32 -> 128) + b3 + residual, ReLU, 16-byte stores
    bf16_t* orow = (bf16_t*)p.out + opix * kC;
#pragma unroll
    for (int ct = 0; ct < 4; ++ct) {
        f32x16 acc = bias_frag(sb + 2 * kMid + ct * 32);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const uint4 a = *(const uint4*)(sw3 + (ct * 2 + u) * 1024 + lane * 16);
            mfma_kgroup<bf16_t>(a, y2[u], acc);
        }
        // lane (pixel, h) holds couts 32 ct + 8 k + 4 h + {0..3}, k = 0..3.  Swap so that half 0 holds 8 k' + {0..7} for even
        // runs and half 1 for odd ones: after v_permlane32_swap(run k (even), run k + 1) the lower half owns [k | upper's k] =
        // couts 8k .. 8k + 7 and the upper half [lower's k + 1 | k + 1] = couts 8 (k + 1) .. + 7  (cdna_hip_programming.md T21)
#pragma unroll
        for (int kk = 0; kk < 2; ++kk) {
            // fp32 pairs of run 2kk (a0, a1) and run 2kk + 1 (c0, c1)
            const int k0 = 2 * kk, k1 = 2 * kk + 1;
            uint32_t a0 = __float_as_uint(acc[4 * k0]), a1 = __float_as_uint(acc[4 * k0 + 1]);
            uint32_t a2 = __float_as_uint(acc[4 * k0 + 2]), a3 = __float_as_uint(acc[4 * k0 + 3]);
            uint32_t c0 = __float_as_uint(acc[4 * k1]), c1 = __float_as_uint(acc[4 * k1 + 1]);
            uint32_t c2 = __float_as_uint(acc[4 * k1 + 2]), c3 = __float_as_uint(acc[4 * k1 + 3]);
            // lower half: keeps run k0 (own couts 8k0 + 0..3), receives the UPPER half's run k0 (couts 8k0 + 4..7)
            // upper half: keeps run k1 (own couts 8k1 + 4..7), receives the LOWER half's run k1 (couts 8k1 + 0..3)
            auto s0 = __builtin_amdgcn_permlane32_swap(a0, c0, false, false);
            auto s1 = __builtin_amdgcn_permlane32_swap(a1, c1, false, false);
            auto s2 = __builtin_amdgcn_permlane32_swap(a2, c2, false, false);
            auto s3 = __builtin_amdgcn_permlane32_swap(a3, c3, false, false);
            // after the swap: r[0] = {lower: own a, upper: lower's c}; r[1] = {lower: upper's a, upper: own c}
            float v[8];
            if (h == 0) {
                v[0] = __uint_as_float(s0[0]); v[1] = __uint_as_float(s1[0]); v[2] = __uint_as_float(s2[0]); v[3] = __uint_as_float(s3[0]);
                v[4] = __uint_as_float(s0[1]); v[5] = __uint_as_float(s1[1]); v[6] = __uint_as_float(s2[1]); v[7] = __uint_as_float(s3[1]);
            } else {
                v[0] = __uint_as_float(s0[0]); v[1] = __uint_as_float(s1[0]); v[2] = __uint_as_float(s2[0]); v[3] = __uint_as_float(s3[0]);
                v[4] = __uint_as_float(s0[1]); v[5] = __uint_as_float(s1[1]); v[6] = __uint_as_float(s2[1]); v[7] = __uint_as_float(s3[1]);
            }
            // both halves now hold 8 consecutive couts: 32 ct + 16 kk + 8 h + {0..7}
            float rv[8];
            chunk_to_f32<bf16_t>(res[ct * 2 + kk], rv);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaxf(v[e] + rv[e], 0.f);
            if (o_ok) *(uint4*)(orow + ct * 32 + 16 * kk + 8 * h) = f32_to_chunk<bf16_t>(v);
        }
    }
}

template <int TH>
int launch_bn(BottleneckParams p, hipStream_t stream) {
    using G = BnCfg<TH>;
    p.tiles_y = (p.H + TH - 1) / TH;
    p.tiles_x = (p.W + 15) / 16;
    const long blocks = (long)p.N * p.tiles_y * p.tiles_x;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)bottleneck_kernel<TH>, hipFuncAttributeMaxDynamicSharedMemorySize, G::LDS);
    }
    hipLaunchKernelGGL((bottleneck_kernel<TH>), dim3((unsigned)blocks), dim3(G::NTHR), G::LDS, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_bottleneck_nhwc(const void* in, const void* w1, const void* w2, const void* w3, const float* b1,
                                      const float* b2, const float* b3, void* out, const int* dims, hipStream_t stream) {
    // dims: [dtype (0 = bf16), N, H, W, C (128), mid (32), tile_rows (0 = automatic, 8, 16)]
    if (!in || !w1 || !w2 || !w3 || !b1 || !b2 || !b3 || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0 || dims[4] != kC || dims[5] != kMid) return COBEVT_ERR_UNSUPPORTED;
    BottleneckParams p;
    p.in = in; p.w1 = (const uint4*)w1; p.w2 = (const uint4*)w2; p.w3 = (const uint4*)w3;
    p.b1 = b1; p.b2 = b2; p.b3 = b3; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3];
    if (p.N < 1 || p.H < 1 || p.W < 1) return COBEVT_ERR_SHAPE;
    if ((long)p.N * p.H * p.W * kC >= 0x7fffffffL * 8L) return COBEVT_ERR_UNSUPPORTED;
    int th = dims[6];
    if (th != 0 && th != 8 && th != 16) return COBEVT_ERR_ARG;
    if (th == 0) {      // 16-row tiles (8 waves, 27 % halo) once they fill the chip twice over, else 8-row tiles (4 waves, 41 % halo)
        const long t16 = (long)p.N * ((p.H + 15) / 16) * ((p.W + 15) / 16);
        th = t16 >= 512 ? 16 : 8;
    }
    return th == 16 ? launch_bn<16>(p, stream) : launch_bn<8>(p, stream);
}
