// The fused row-local chain of row_chain.hip for 64-channel rows with a 128-wide hidden layer - the chain behind every attention of
// the LiDAR FuseBEVT encoder (swap_fusion_modules.py:87-128,126,172-190 with input_dim 64, mlp_dim 128; base_transformer.py:102-124)
// - with NO workgroup barrier and no activation tile in LDS (gfx950, bf16):
//
//     y = a . Wp^T + bp + skip ;  z = y + ( GELU( LN(y) . W1'^T + b1' ) . W2^T + b2 ) ;  out = post-LN?(z) ;
//     next = act( LN?(out) . Wn'^T + bn' )                                              (optional: the next half's to_qkv)
//
// The generic kernel walks 128-column panels with four waves: at C = 64 two of them compute zero columns in the out-projection
// and fc2 phases, every k-group sits behind a uniform branch, each of the 16,384 32-row workgroups of a (8, 256, 256, 64) map
// re-streams the chain's 74 KB of weight fragments from L2 (1.2 GB per launch) and every phase is a barrier-separated LDS round
// trip - 209 us per launch against ~75 us for the 400 MB the chain has to move (a, skip, out, next).  Here:
//   * the 64 KB of weight fragments live in LDS for the life of a (persistent) workgroup, one ds_read_b128 per MFMA;
//   * a WAVE owns 32 rows through the whole chain.  MFMAs are issued as D = W . X^T, so a lane holds, of its own row, the columns
//     {32 t + 8 k + 4 h + j} - and the next GEMM takes exactly those registers, packed to bf16, as its B operand: the contraction
//     index of W1 / W2 / Wn is stored in accumulator-register order (a half-swap of 8-byte runs inside each fragment, done while
//     the fragments are copied into LDS), so y, LN(y), the hidden activations and `out` never leave the registers.  The two lanes
//     that share a row (lane, lane ^ 32) complete each other's LayerNorm sums with one cross-half exchange;
//   * global I/O is 16 bytes per lane at natural addresses; one v_permlane32_swap per register turns a loaded 16-byte piece into
//     the two 8-byte column runs of the accumulator order and back;
//   * waves never synchronise, so 12-16 independent waves per CU hide each other's load latency (the barrier-phased form ran
//     157 us per launch, a 64-row tile costing ~20k cycles of mostly exposed round trips).
#include <stdlib.h>
#include "row_chain.hpp"

namespace cobevt {

namespace {

// LDS: fragment-ordered weights [tile][k-group][64 lanes][16 B]: Wp 2 x 4 (natural k order), W1 4 x 4, W2 2 x 8, Wn 6 x 4
// (accumulator k order), then the fp32 bias / affine table
constexpr int kFp = 0, kF1 = kFp + 8, kF2 = kF1 + 16, kFn = kF2 + 16, kFrags = kFn + 24;      // in 1-KB fragments
constexpr int kBp = 0, kB1 = 64, kB2 = 192, kPg = 256, kPb = 320, kBn = 384, kBiasFloats = 576;
constexpr int kLdsBytes = kFrags * 1024 + kBiasFloats * 4;                                    // 67,840 B: two workgroups per CU

__device__ __forceinline__ float xhalf_sum(float v) {          // v + the value of lane ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
// {lo, hi} (16 bytes at natural channel order, channels 16 m + 8 h .. + 7) <-> the two 4-channel runs this lane holds in
// accumulator order (run 2 m: channels 16 m + 4 h .. + 3; run 2 m + 1: channels 16 m + 8 + 4 h .. + 3).  The same exchange in
// both directions: it swaps the upper half-wave's `a` with the lower half-wave's `b`.
__device__ __forceinline__ void half_swap(uint2& a, uint2& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    a.x = r[0]; b.x = r[1];
    r = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.y = r[0]; b.y = r[1];
}

// v[t][r]: the lane's 32 values of its row (2 column tiles x 16 accumulator registers) -> normalised over the row's 64 channels
__device__ __forceinline__ void row_normalise(float (&v)[2][16], float eps) {
    float s = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) s += v[t][r];
    const float mean = xhalf_sum(s) * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { const float d = v[t][r] - mean; q += d * d; }
    const float rstd = rsqrtf(xhalf_sum(q) * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) v[t][r] = (v[t][r] - mean) * rstd;
}

// 8 accumulator registers [8 u, 8 u + 8) of a tile -> one bf16 B operand (k-group 2 t + u of the accumulator k order)
__device__ __forceinline__ uint4 pack8(const float* v) {
    return make_uint4(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]), pack_bf2(v[4], v[5]), pack_bf2(v[6], v[7]));
}
__device__ __forceinline__ float rbf(float x) { return bf2f(f2bf(x)); }          // the value a bf16 store would keep

// NNT2: 32-column tiles of the next projection (Nn = 32 NNT2 <= 192; 0 = none).  NW waves per workgroup.
// PF (round 6): the NEXT block's a / skip rows are requested before the current block's chain starts (32 more registers: four waves per
// workgroup instead of six) - the shipped form, launch64 below.
template <int NNT2, int NW, bool PF = false>
__global__ __launch_bounds__(NW * 64, 2) void row_chain64_kernel(RowChainParams p, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* wl = (const uint4*)smem;
    float* sb = (float*)(smem + kFrags * 1024);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    // ---- weights -> LDS.  Natural fragments are [tile][k-group][half][32 lanes][16 B]; accumulator k order: lane (h, q) of
    // k-group G takes bytes [8 h, +8) of the half-0 piece and bytes [8 h, +8) of the half-1 piece of lane q
    {
        // (all of a thread's loads first, then its LDS stores: a rolled load -> store loop is one exposed L2 round trip per iteration)
        uint4* dst = (uint4*)smem;
        constexpr int TOTAL = (kFn + 4 * NNT2) * 64, PER = (TOTAL + NW * 64 - 1) / (NW * 64);
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64, ic = i < TOTAL ? i : TOTAL - 1;
            const int f = ic >> 6, ln = ic & 63, hh = ln >> 5, q = ln & 31;
            const uint4* src;
            int sf;                                   // source fragment index (tile * k-groups-per-tile + k-group)
            if (f < kF1) { src = p.wp; sf = (f >> 2) * 8 + (f & 3); }
            else if (f < kF2) { src = p.w1; sf = ((f - kF1) >> 2) * 8 + ((f - kF1) & 3); }
            else if (f < kFn) { src = p.w2; sf = f - kF2; }                       // [2 tiles][8 k-groups] = Hdp / 16 per tile
            else { src = p.wn ? p.wn : p.w1; sf = ((f - kFn) >> 2) * 8 + ((f - kFn) & 3); }
            const bool nat = f < kF1;                 // Wp keeps the natural k order
            const uint4* base = src + (size_t)sf * 64;
            const unsigned char* a0 = (const unsigned char*)(base + (nat ? ln : q)) + (nat ? 0 : 8 * hh);
            const unsigned char* a1 = nat ? a0 + 8 : (const unsigned char*)(base + 32 + q) + 8 * hh;
            const uint2 lo = *(const uint2*)a0, hi = *(const uint2*)a1;
            tmp[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64;
            if (i < TOTAL) dst[i] = tmp[u];
        }
        for (int i = tid; i < kBiasFloats; i += NW * 64) {
            const float* src = i < kB1 ? p.bp : i < kB2 ? p.b1 : i < kPg ? p.b2 : i < kPb ? p.post_g : i < kBn ? p.post_b : p.bn;
            const int j = i < kB1 ? i : i < kB2 ? i - kB1 : i < kPg ? i - kB2 : i < kPb ? i - kPg : i < kBn ? i - kPb : i - kBn;
            const int n = i < kBn ? (i >= kB1 && i < kB2 ? 128 : 64) : p.Nn;
            const bool keep = (src != nullptr) & (j < n);
            const float val = (src ? src : p.b1)[keep ? j : 0];
            sb[i] = keep ? val : 0.f;
        }
    }
    __syncthreads();                                  // the only barrier: from here on every wave is on its own

    // (`opq` is an opaque zero refreshed every block: the fragments are loop-invariant, and hoisted out of the block loop they
    // would occupy 256 VGPRs)
    int opq = 0;
    auto frag = [&](int f) { return wl[f * 64 + lane + opq]; };
    // bias of this lane's column run k of tile t: columns 32 t + 8 k + 4 h ..
    auto bias4 = [&](int table, int t, int k) { return *(const float4*)(sb + table + 32 * t + 8 * k + 4 * h); };

    const int nwaves = gridDim.x * NW;
    uint4 afn[PF ? 4 : 1], skn[PF ? 4 : 1];
    auto load_rows = [&](int blk, uint4 (&a4)[4], uint4 (&s4)[4]) {
        const int m0 = blk * 32;
        const int grow = m0 + ql < p.M ? m0 + ql : p.M - 1;          // tail block: clamped row (finite data, never stored)
        const bf16_t* arow = p.a + (size_t)grow * 64;
        const bf16_t* srow = p.skip + (size_t)grow * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) a4[g] = *(const uint4*)(arow + 16 * g + 8 * h);
#pragma unroll
        for (int g = 0; g < 4; ++g) s4[g] = *(const uint4*)(srow + 16 * g + 8 * h);
    };
    if constexpr (PF) {
        const int b0 = blockIdx.x * NW + wave;
        load_rows(b0 < nblk ? b0 : nblk - 1, (uint4(&)[4])afn, (uint4(&)[4])skn);
    }
    for (int blk = blockIdx.x * NW + wave; blk < nblk; blk += nwaves) {
        asm volatile("" : "+v"(opq));
        const int m0 = blk * 32;
        const bool live = m0 + ql < p.M;
        const int grow = live ? m0 + ql : p.M - 1;                   // tail block: clamped row (finite data, never stored)

        // ---- loads: a as natural B operands (k-group g: channels 16 g + 8 h ..), skip as 16-byte pieces
        uint4 af[4], sk[4];
        if constexpr (PF) {
#pragma unroll
            for (int g = 0; g < 4; ++g) { af[g] = afn[g]; sk[g] = skn[g]; }
            const int nb = blk + nwaves;
            load_rows(nb < nblk ? nb : nblk - 1, (uint4(&)[4])afn, (uint4(&)[4])skn);      // unconditional (clamped): in flight under this block's chain
            __builtin_amdgcn_sched_barrier(0);
        } else {
            load_rows(blk, af, sk);
        }

        // ---- phase A: y = a . Wp^T + bp + skip, rounded to bf16 (what the unfused path stores)
        float y[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(frag(kFp + t * 4 + g), af[g], acc);
#pragma unroll
            for (int m = 0; m < 2; ++m) {             // 16-byte piece m of this tile = channels 32 t + 16 m + 8 h .. -> runs 2 m, 2 m + 1
                uint2 s0 = make_uint2(sk[2 * t + m].x, sk[2 * t + m].y), s1 = make_uint2(sk[2 * t + m].z, sk[2 * t + m].w);
                half_swap(s0, s1);
                const float4 b0 = bias4(kBp, t, 2 * m), b1 = bias4(kBp, t, 2 * m + 1);
                float* d = &y[t][8 * m];
                d[0] = rbf(acc[8 * m + 0] + b0.x + bf2f(s0.x & 0xffff)); d[1] = rbf(acc[8 * m + 1] + b0.y + bf2f(s0.x >> 16));
                d[2] = rbf(acc[8 * m + 2] + b0.z + bf2f(s0.y & 0xffff)); d[3] = rbf(acc[8 * m + 3] + b0.w + bf2f(s0.y >> 16));
                d[4] = rbf(acc[8 * m + 4] + b1.x + bf2f(s1.x & 0xffff)); d[5] = rbf(acc[8 * m + 5] + b1.y + bf2f(s1.x >> 16));
                d[6] = rbf(acc[8 * m + 6] + b1.z + bf2f(s1.y & 0xffff)); d[7] = rbf(acc[8 * m + 7] + b1.w + bf2f(s1.y >> 16));
            }
        }

        // ---- phase B: x_hat = normalise(y) -> B operands of fc1 (k-group 2 t + u = registers [8 u, +8) of tile t)
        uint4 xh[4];
        {
            float v[2][16];
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) v[t][r] = y[t][r];
            row_normalise(v, p.eps1);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int u = 0; u < 2; ++u) xh[2 * t + u] = pack8(&v[t][8 * u]);
        }

        // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> B operands of fc2
        uint4 hid[8];
#pragma unroll
        for (int n = 0; n < 4; ++n) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(frag(kF1 + n * 4 + g), xh[g], acc);
            float hv[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 b = bias4(kB1, n, k);
                hv[4 * k] = gelu_bf16(acc[4 * k] + b.x); hv[4 * k + 1] = gelu_bf16(acc[4 * k + 1] + b.y);
                hv[4 * k + 2] = gelu_bf16(acc[4 * k + 2] + b.z); hv[4 * k + 3] = gelu_bf16(acc[4 * k + 3] + b.w);
            }
            hid[2 * n] = pack8(hv);
            hid[2 * n + 1] = pack8(hv + 8);
        }

        // ---- phase D: z = hidden . W2^T + b2 + y ; phase E: optional post-LayerNorm ; `out` rows (bf16)
        float z[2][16];
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) mfma_kgroup<bf16_t>(frag(kF2 + t * 8 + g), hid[g], acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 b = bias4(kB2, t, k);
                z[t][4 * k] = acc[4 * k] + b.x + y[t][4 * k]; z[t][4 * k + 1] = acc[4 * k + 1] + b.y + y[t][4 * k + 1];
                z[t][4 * k + 2] = acc[4 * k + 2] + b.z + y[t][4 * k + 2]; z[t][4 * k + 3] = acc[4 * k + 3] + b.w + y[t][4 * k + 3];
            }
        }
        if (p.post_g) {
            row_normalise(z, p.eps_post);
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 g = bias4(kPg, t, k), b = bias4(kPb, t, k);
                    z[t][4 * k] = z[t][4 * k] * g.x + b.x; z[t][4 * k + 1] = z[t][4 * k + 1] * g.y + b.y;
                    z[t][4 * k + 2] = z[t][4 * k + 2] * g.z + b.z; z[t][4 * k + 3] = z[t][4 * k + 3] * g.w + b.w;
                }
        }
        uint4 oh[4];                                   // `out` in accumulator order, bf16: the next projection's B operands
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int u = 0; u < 2; ++u) oh[2 * t + u] = pack8(&z[t][8 * u]);
        {
            bf16_t* orow = p.out + (size_t)grow * 64;
#pragma unroll
            for (int i = 0; i < 4; ++i) {             // piece i = tile i >> 1, m = i & 1: runs 2 m, 2 m + 1 -> channels 32 t + 16 m + 8 h ..
                uint2 r0 = make_uint2(oh[i].x, oh[i].y), r1 = make_uint2(oh[i].z, oh[i].w);
                half_swap(r0, r1);
                if (live) *(uint4*)(orow + 16 * i + 8 * h) = make_uint4(r0.x, r0.y, r1.x, r1.y);
            }
        }

        if (NNT2 > 0) {
            if (p.next_ln) {                           // the rows exactly as stored (bf16), normalised
                float v[2][16];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 16; ++r) v[t][r] = rbf(z[t][r]);
                row_normalise(v, p.eps_next);
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int u = 0; u < 2; ++u) oh[2 * t + u] = pack8(&v[t][8 * u]);
            }
            bf16_t* nrow = p.out_next + (size_t)grow * p.Nn;
#pragma unroll
            for (int n = 0; n < NNT2; ++n) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(frag(kFn + n * 4 + g), oh[g], acc);
                float nv[16];
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 b = bias4(kBn, n, k);
                    nv[4 * k] = acc[4 * k] + b.x; nv[4 * k + 1] = acc[4 * k + 1] + b.y;
                    nv[4 * k + 2] = acc[4 * k + 2] + b.z; nv[4 * k + 3] = acc[4 * k + 3] + b.w;
                }
                if (p.next_act == 1) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) nv[r] = fmaxf(nv[r], 0.f);
                } else if (p.next_act == 2) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) nv[r] = gelu_bf16(nv[r]);
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
}

template <int NNT2, int NW, bool PF> int launch64v(const RowChainParams& p, hipStream_t stream) {
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)row_chain64_kernel<NNT2, NW, PF>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    }
    const int nblk = (p.M + 31) / 32;
    int blocks = (nblk + NW - 1) / NW;
    if (blocks > 512) blocks = 512;                        // persistent: two workgroups per CU walk the 32-row blocks
    hipLaunchKernelGGL((row_chain64_kernel<NNT2, NW, PF>), dim3((unsigned)blocks), dim3(NW * 64), kLdsBytes, stream, p, nblk);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
template <int NNT2> int launch64(const RowChainParams& p, hipStream_t stream) {
    // Round 6, same-job A/B on the LiDAR FuseBEVT workload (profiles/r06_rc64_prefetch_ab.txt): the next block's rows prefetched under the
    // current block's chain, 2 x 4 waves per CU at 192 VGPRs - 102 -> 87 us per launch (370 MB: 3.6 -> 4.25 TB/s), 525 -> 538 frames/s; the
    // same prefetch with six waves per workgroup (one workgroup per CU fits) 103 us.  COBEVT_RC64_PREFETCH=0 selects the round-5 form.
    static const int mode = [] { const char* e = getenv("COBEVT_RC64_PREFETCH"); return e ? atoi(e) : 1; }();
    if (mode == 1) return launch64v<NNT2, 4, true>(p, stream);
    if (mode == 2) return launch64v<NNT2, 6, true>(p, stream);
    return launch64v<NNT2, 6, false>(p, stream);                    // 2 workgroups x 6 waves per CU = 3 waves per SIMD, no prefetch
}

}  // namespace

int launch_row_chain64(const RowChainParams& p, hipStream_t stream) {
    if (p.C != 64 || p.Hd != 128 || p.Hdp != 128 || !p.skip || p.skip_rows != p.M) return -1;
    if (p.wn && (p.Nn % 32 != 0 || p.Nn > 192)) return -1;
    if (p.M < 16384) return -1;                            // small maps: the generic kernel's finer workgroups fill the chip better
    switch (p.wn ? p.Nn / 32 : 0) {
        case 0: return launch64<0>(p, stream);
        case 2: return launch64<2>(p, stream);
        case 4: return launch64<4>(p, stream);
        case 6: return launch64<6>(p, stream);
        default: return -1;
    }
}

}  // namespace cobevt
