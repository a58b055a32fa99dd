// The key / value side of a 128-channel FAX level (row_chain.hip's "projection chain": pre-activation BatchNorm -> ReLU -> 1x1 conv
// (+ ray embedding) -> LayerNorm -> stacked to_k | to_v of both cross attentions; fax_modules.py:281-292,377-396,201-205) with the
// rows in registers and the weights in LDS (gfx950, bf16) - the wave-level form of row_chain64.hip at 128 channels:
//
//     y    = ReLU?(a * pre_scale + pre_shift) . Wp^T + bp + skip          (rounded to bf16: what the unfused path stores)
//     next = act( LN?(y) . Wn'^T + bn' )                                   (Nn = 256: to_k or to_v of the two attentions)
//
// The barrier-phased kernel runs 39 us per operand on the 81,920-row level-0 maps (84 MB / 63 MB: 12-16 us of HBM time), its
// 2,560 32-row workgroups re-streaming 96 KB of weight fragments each.  Here one workgroup per CU copies the fragments into LDS
// once (Wn with its contraction index in accumulator-register order, see row_chain64.hip) and its waves walk 32-row blocks on
// their own: a in natural B-operand order straight from global memory, y / LN(y) in registers, 16-byte stores after a half swap.
#include "row_chain.hpp"

namespace cobevt {

namespace {

constexpr int kFp = 0, kFn = 32, kMaxNnt = 8;               // fragments: Wp 4 x 8 (natural), Wn up to 8 tiles x 8 (accumulator order)
constexpr int kTab = (kFn + kMaxNnt * 8) * 1024;             // fp32 tables behind the 96 KB of fragments:
constexpr int kSc = 0, kSh = 128, kBp = 256, kBn = 384;      // pre_scale[128] pre_shift[128] bp[128] bn[256]
constexpr int kTabFloats = 640;
constexpr int kLdsBytes = kTab + kTabFloats * 4;             // 100,864 B: one workgroup per CU

__device__ __forceinline__ float xhalf_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void half_swap(uint2& a, uint2& b) {      // see row_chain64.hip
    auto r = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    a.x = r[0]; b.x = r[1];
    r = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.y = r[0]; b.y = r[1];
}
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }

// NNT: 32-column tiles of the next projection (Nn = 32 NNT).  NW waves per workgroup.
template <int NNT, int NW>
__global__ __launch_bounds__(NW * 64, 1) void proj_chain128_kernel(RowChainParams p, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* wl = (const uint4*)smem;
    float* tab = (float*)(smem + kTab);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    {
        // (all of a thread's loads first, then its LDS stores: a rolled load -> store loop is one exposed L2 round trip per
        //  iteration - twelve of them took ~11 us of a 45-us launch)
        uint4* dst = (uint4*)smem;
        constexpr int TOTAL = (kFn + NNT * 8) * 64, PER = (TOTAL + NW * 64 - 1) / (NW * 64);
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64, ic = i < TOTAL ? i : TOTAL - 1;
            const int f = ic >> 6, ln = ic & 63, hh = ln >> 5, q = ln & 31;
            // natural fragment: the lane's own 16 bytes; accumulator k order: bytes [8 hh, +8) of the half-0 and of the half-1 piece
            const uint4* base = f < kFn ? p.wp + (size_t)f * 64 : p.wn + (size_t)(f - kFn) * 64;
            const unsigned char* a0 = (const unsigned char*)(base + (f < kFn ? ln : q)) + (f < kFn ? 0 : 8 * hh);
            const unsigned char* a1 = f < kFn ? a0 + 8 : (const unsigned char*)(base + 32 + q) + 8 * hh;
            const uint2 lo = *(const uint2*)a0, hi = *(const uint2*)a1;
            tmp[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64;
            if (i < TOTAL) dst[i] = tmp[u];
        }
        for (int i = tid; i < kTabFloats; i += NW * 64) {
            float v;
            if (i < kSh) v = p.pre_scale ? p.pre_scale[i] : 1.f;            // (no BatchNorm: identity)
            else if (i < kBp) v = p.pre_shift ? p.pre_shift[i - kSh] : 0.f;
            else if (i < kBn) v = p.bp ? p.bp[i - kBp] : 0.f;
            else v = (p.bn && i - kBn < p.Nn) ? p.bn[i - kBn] : 0.f;
            tab[i] = v;
        }
    }
    __syncthreads();                                  // the only barrier

    int opq = 0;                                      // opaque zero: keeps the loop-invariant LDS reads inside the loop
    const float relu_floor = (p.pre_scale && p.pre_relu) ? 0.f : -INFINITY;
    // weight fragments of one 32-column tile: all eight k-groups requested together, one tile ahead of the MFMAs that use them
    // (a fragment read issued right in front of its MFMA costs the whole LDS round trip with two waves per SIMD)
    auto load_tile = [&](uint4 (&fr)[8], int first) {
#pragma unroll
        for (int g = 0; g < 8; ++g) fr[g] = wl[(first + g) * 64 + lane + opq];
    };
    const int nwaves = gridDim.x * NW;
    for (int blk = blockIdx.x * NW + wave; blk < nblk; blk += nwaves) {
        asm volatile("" : "+v"(opq));
        const int m0 = blk * 32;
        const bool live = m0 + ql < p.M;
        const size_t grow = live ? m0 + ql : p.M - 1;                 // tail block: clamped row (finite data, never stored)
        const bf16_t* arow = p.a + grow * 128;
        uint4 af[8], sk[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) af[g] = *(const uint4*)(arow + 16 * g + 8 * h);
        if (p.skip) {
            const bf16_t* srow = p.skip + (size_t)(grow % p.skip_rows) * 128;
#pragma unroll
            for (int g = 0; g < 8; ++g) sk[g] = *(const uint4*)(srow + 16 * g + 8 * h);
        } else {
#pragma unroll
            for (int g = 0; g < 8; ++g) sk[g] = make_uint4(0, 0, 0, 0);
        }
        // pre-activation BatchNorm -> ReLU on the rows as they arrive (channels 16 g + 8 h ..); branch-free: without a BatchNorm
        // the table holds scale 1 / shift 0 (exact on bf16 values), without a ReLU the floor is -inf
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            float v[8];
            chunk_to_f32<bf16_t>(af[g], v);
            const float* sc = tab + kSc + 16 * g + 8 * h + opq;
            const float4 s0 = *(const float4*)sc, s1 = *(const float4*)(sc + 4);
            const float4 t0 = *(const float4*)(sc + (kSh - kSc)), t1 = *(const float4*)(sc + (kSh - kSc) + 4);
            v[0] = fmaxf(fmaf(v[0], s0.x, t0.x), relu_floor); v[1] = fmaxf(fmaf(v[1], s0.y, t0.y), relu_floor);
            v[2] = fmaxf(fmaf(v[2], s0.z, t0.z), relu_floor); v[3] = fmaxf(fmaf(v[3], s0.w, t0.w), relu_floor);
            v[4] = fmaxf(fmaf(v[4], s1.x, t1.x), relu_floor); v[5] = fmaxf(fmaf(v[5], s1.y, t1.y), relu_floor);
            v[6] = fmaxf(fmaf(v[6], s1.z, t1.z), relu_floor); v[7] = fmaxf(fmaf(v[7], s1.w, t1.w), relu_floor);
            af[g] = f32_to_chunk<bf16_t>(v);
        }

        // ---- phase A: y = a . Wp^T + bp + skip (bf16-rounded), 4 column tiles
        float y[4][16];
        uint4 fr[2][8];
        load_tile(fr[0], kFp);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            load_tile(fr[(t + 1) & 1], t < 3 ? kFp + (t + 1) * 8 : kFn);        // the next tile (after tile 3: the next projection's first)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) mfma_kgroup<bf16_t>(fr[t & 1][g], af[g], acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {             // piece 2 t + m = channels 32 t + 16 m + 8 h .. -> runs 2 m, 2 m + 1
                uint2 s0 = make_uint2(sk[2 * t + m].x, sk[2 * t + m].y), s1 = make_uint2(sk[2 * t + m].z, sk[2 * t + m].w);
                half_swap(s0, s1);
                const float4 b0 = *(const float4*)(tab + kBp + 32 * t + 16 * m + 4 * h + opq);
                const float4 b1 = *(const float4*)(tab + kBp + 32 * t + 16 * m + 8 + 4 * h + opq);
                float* d = &y[t][8 * m];
                d[0] = rbf(acc[8 * m + 0] + b0.x + bf2f(s0.x & 0xffff)); d[1] = rbf(acc[8 * m + 1] + b0.y + bf2f(s0.x >> 16));
                d[2] = rbf(acc[8 * m + 2] + b0.z + bf2f(s0.y & 0xffff)); d[3] = rbf(acc[8 * m + 3] + b0.w + bf2f(s0.y >> 16));
                d[4] = rbf(acc[8 * m + 4] + b1.x + bf2f(s1.x & 0xffff)); d[5] = rbf(acc[8 * m + 5] + b1.y + bf2f(s1.x >> 16));
                d[6] = rbf(acc[8 * m + 6] + b1.z + bf2f(s1.y & 0xffff)); d[7] = rbf(acc[8 * m + 7] + b1.w + bf2f(s1.y >> 16));
            }
        }
        if (p.out) {                                   // the map itself (not stored on the product path)
            bf16_t* orow = p.out + grow * 128;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const uint4 pk = pack8(&y[i >> 1][8 * (i & 1)]);
                uint2 r0 = make_uint2(pk.x, pk.y), r1 = make_uint2(pk.z, pk.w);
                half_swap(r0, r1);
                if (live) *(uint4*)(orow + 16 * i + 8 * h) = make_uint4(r0.x, r0.y, r1.x, r1.y);
            }
        }
        // ---- LayerNorm over the row's 128 channels (this lane's 64 + the partner's), B operands in accumulator k order
        if (p.next_ln) {
            float s = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) s += y[t][r];
            const float mean = xhalf_sum(s) * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) { const float d = y[t][r] - mean; q += d * d; }
            const float rstd = rsqrtf(xhalf_sum(q) * (1.0f / 128.0f) + p.eps_next);
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) y[t][r] = (y[t][r] - mean) * rstd;
        }
        uint4 yh[8];
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) yh[2 * t + u] = pack8(&y[t][8 * u]);

        // ---- next projection: NNT column tiles, 16-byte stores after the half swap
        bf16_t* nrow = p.out_next + grow * p.Nn;
#pragma unroll
        for (int n = 0; n < NNT; ++n) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            if (n + 1 < NNT) load_tile(fr[(n + 1) & 1], kFn + (n + 1) * 8);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g) mfma_kgroup<bf16_t>(fr[n & 1][g], yh[g], acc);
            float nv[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 b = *(const float4*)(tab + kBn + 32 * n + 8 * k + 4 * h + opq);
                nv[4 * k] = acc[4 * k] + b.x; nv[4 * k + 1] = acc[4 * k + 1] + b.y;
                nv[4 * k + 2] = acc[4 * k + 2] + b.z; nv[4 * k + 3] = acc[4 * k + 3] + b.w;
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                const uint4 pk = pack8(&nv[8 * m]);
                uint2 r0 = make_uint2(pk.x, pk.y), r1 = make_uint2(pk.z, pk.w);
                half_swap(r0, r1);
                if (live) *(uint4*)(nrow + 32 * n + 16 * m + 8 * h) = make_uint4(r0.x, r0.y, r1.x, r1.y);
            }
        }
    }
}

template <int NNT> int launch128(const RowChainParams& p, hipStream_t stream) {
    constexpr int NW = 8;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)proj_chain128_kernel<NNT, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    }
    const int nblk = (p.M + 31) / 32;
    int blocks = (nblk + NW - 1) / NW;
    if (blocks > 256) blocks = 256;                        // persistent: one workgroup per CU
    hipLaunchKernelGGL((proj_chain128_kernel<NNT, NW>), dim3((unsigned)blocks), dim3(NW * 64), kLdsBytes, stream, p, nblk);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace

// cobevt_proj_chain (row_chain.hip) hands the big maps to this launcher; -1 = shape does not qualify
int launch_proj_chain128(const RowChainParams& p, hipStream_t stream) {
    if (p.C != 128 || !p.wn || !p.out_next || p.Nn % 32 != 0 || p.Nn > 32 * kMaxNnt || p.next_act != 0) return -1;
    if (p.M < 32768) return -1;                            // smaller maps: the 32-row workgroups of the generic kernel fill the chip better
    switch (p.Nn / 32) {
        case 4: return launch128<4>(p, stream);
        case 8: return launch128<8>(p, stream);
        default: return -1;
    }
}

}  // namespace cobevt
