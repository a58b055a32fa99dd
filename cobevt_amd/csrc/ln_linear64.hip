// LayerNorm + Linear on 64-channel rows of a BIG map as independent waves (gfx950, bf16; round 6):
//     out[m][0 .. N) = act( LN(x[m][0 .. 64)) . W'^T + b' ),   N = 32 .. 192 in steps of 32   (LayerNorm affine folded into W', b')
// = the first half block's to_qkv behind PreNormResidual.norm of the LiDAR FuseBEVT encoder (swap_fusion_modules.py:93 with input_dim 64:
// 524,288 rows -> 192 columns); every later to_qkv rides at the end of the 64-channel row chain (row_chain64.hip), this one has no chain
// in front of it and ran on the generic 32-row dense-row kernel (gemm_rows3.hip): 126 us for 268 MB = 2.1 TB/s - barrier-phased
// workgroups, K padded from 64 to 128, half of the second 128-column pass empty.  Here, as in row_chain64.hip: the weight fragments
// (24 KB) live in LDS for the life of a persistent workgroup, a WAVE owns 32 rows - a lane loads 64 bytes of its row at natural
// addresses, the two lanes that share a row complete each other's LayerNorm sums with one cross-half exchange, D = W . X^T hands the lane
// four-column runs that a v_permlane32_swap per register turns back into 16-byte stores - waves never synchronise, and the NEXT block's
// rows are requested before the current block's arithmetic starts.
#include <stdlib.h>
#include "common.hpp"

namespace cobevt {

namespace {

__device__ __forceinline__ float xhalf_sum(float v) {          // v + the value of lane ^ 32
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}
__device__ __forceinline__ void half_swap(uint2& a, uint2& b) {
    auto r = __builtin_amdgcn_permlane32_swap(a.x, b.x, false, false);
    a.x = r[0]; b.x = r[1];
    r = __builtin_amdgcn_permlane32_swap(a.y, b.y, false, false);
    a.y = r[0]; b.y = r[1];
}

struct LnLin64Params {
    const bf16_t* in;       // [M][64]
    const uint4* wfrag;     // fragment-ordered [N_p/32][8 k-groups (K padded to 128)][64 lanes][16 B]; k-groups 0..3 are used
    const float* bias;      // [N] or null
    bf16_t* out;            // [M][N]
    int M, N, act;
    float eps;
};

// NNT: 32-column tiles (N = 32 NNT).  NW waves per workgroup.
template <int NNT, int NW>
__global__ __launch_bounds__(NW * 64, 6) void ln_linear64_kernel(LnLin64Params p, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    uint4* wl = (uint4*)smem;                              // [NNT * 4 fragments][64 lanes]
    float* sb = (float*)(smem + NNT * 4 * 1024);           // [N] bias
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    {
        constexpr int TOTAL = NNT * 4 * 64, PER = (TOTAL + NW * 64 - 1) / (NW * 64);
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64, ic = i < TOTAL ? i : TOTAL - 1;
            const int f = ic >> 6, ln = ic & 63;
            tmp[u] = p.wfrag[(size_t)((f >> 2) * 8 + (f & 3)) * 64 + ln];
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int i = tid + u * NW * 64;
            if (i < TOTAL) wl[i] = tmp[u];
        }
        for (int i = tid; i < NNT * 32; i += NW * 64) sb[i] = p.bias ? p.bias[i] : 0.f;
    }
    __syncthreads();                                       // the only barrier

    int opq = 0;                                           // opaque zero: keeps the fragment reads inside the block loop (row_chain64.hip)
    const int nwaves = gridDim.x * NW;
    auto load_rows = [&](int blk, uint4 (&x4)[4]) {
        const int m0 = blk * 32;
        const int grow = m0 + ql < p.M ? m0 + ql : p.M - 1;           // tail block: clamped row (finite data, never stored)
        const bf16_t* row = p.in + (size_t)grow * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) x4[g] = *(const uint4*)(row + 16 * g + 8 * h);
    };
    uint4 xn[4];
    {
        const int b0 = blockIdx.x * NW + wave;
        load_rows(b0 < nblk ? b0 : nblk - 1, xn);
    }
    for (int blk = blockIdx.x * NW + wave; blk < nblk; blk += nwaves) {
        asm volatile("" : "+v"(opq));
        const int m0 = blk * 32;
        const bool live = m0 + ql < p.M;
        const int grow = live ? m0 + ql : p.M - 1;
        uint4 xr[4];
#pragma unroll
        for (int g = 0; g < 4; ++g) xr[g] = xn[g];
        {
            const int nb = blk + nwaves;
            load_rows(nb < nblk ? nb : nblk - 1, xn);      // unconditional (clamped): in flight under this block's arithmetic
            __builtin_amdgcn_sched_barrier(0);
        }
        // ---- LayerNorm of the row: this lane holds channels 16 g + 8 h .. + 7 (g = 0 .. 3), lane ^ 32 the other 32
        float v[32];
#pragma unroll
        for (int g = 0; g < 4; ++g) chunk_to_f32<bf16_t>(xr[g], v + 8 * g);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) s += v[e];
        const float mean = xhalf_sum(s) * (1.0f / 64.0f);
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 32; ++e) { const float d = v[e] - mean; q += d * d; }
        const float rstd = rsqrtf(xhalf_sum(q) * (1.0f / 64.0f) + p.eps);
#pragma unroll
        for (int e = 0; e < 32; ++e) v[e] = (v[e] - mean) * rstd;
        uint4 xh[4];                                       // the normalised row, bf16 (what the dense-row kernel stages), as B operands
#pragma unroll
        for (int g = 0; g < 4; ++g) xh[g] = f32_to_chunk<bf16_t>(v + 8 * g);

        bf16_t* orow = p.out + (size_t)grow * p.N;
#pragma unroll
        for (int n = 0; n < NNT; ++n) {
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(wl[(n * 4 + g) * 64 + lane + opq], xh[g], acc);   // D = W . X^T: lane <-> row
            float nv[16];
#pragma unroll
            for (int k = 0; k < 4; ++k) {                  // accumulator registers 4 k .. 4 k + 3 <-> columns 32 n + 8 k + 4 h ..
                const float4 b = *(const float4*)(sb + 32 * n + 8 * k + 4 * h);
                nv[4 * k] = apply_act<bf16_t>(acc[4 * k] + b.x, p.act); nv[4 * k + 1] = apply_act<bf16_t>(acc[4 * k + 1] + b.y, p.act);
                nv[4 * k + 2] = apply_act<bf16_t>(acc[4 * k + 2] + b.z, p.act); nv[4 * k + 3] = apply_act<bf16_t>(acc[4 * k + 3] + b.w, p.act);
            }
#pragma unroll
            for (int m = 0; m < 2; ++m) {                  // runs 2 m, 2 m + 1 -> the 16-byte piece of channels 32 n + 16 m + 8 h ..
                const uint4 pk = f32_to_chunk<bf16_t>(nv + 8 * m);
                uint2 r0 = make_uint2(pk.x, pk.y), r1 = make_uint2(pk.z, pk.w);
                half_swap(r0, r1);
                if (live) *(uint4*)(orow + 32 * n + 16 * m + 8 * h) = make_uint4(r0.x, r0.y, r1.x, r1.y);
            }
        }
    }
}

template <int NNT> int launch_nnt(const LnLin64Params& p, hipStream_t stream) {
    constexpr int NW = 4;
    constexpr int lds = NNT * 4 * 1024 + NNT * 32 * 4;
    const int nblk = (p.M + 31) / 32;
    int blocks = (nblk + NW - 1) / NW;
    // persistent: SIX co-resident workgroups per CU (24.8 KB of LDS and 72 VGPRs each) walk the 32-row blocks.  Measured on the LiDAR
    // shape (M = 524,288, N = 192): 512 workgroups 90.8 us, 1024 84.2, 1536 64.8 (4.1 TB/s), 2048 (not co-resident: a tail) 79.8.
    static const int cap = [] { const char* e = getenv("COBEVT_LN64_BLOCKS"); return e && atoi(e) > 0 ? atoi(e) : 1536; }();
    if (blocks > cap) blocks = cap;
    hipLaunchKernelGGL((ln_linear64_kernel<NNT, NW>), dim3((unsigned)blocks), dim3(NW * 64), lds, stream, p, nblk);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace

// the big-map LayerNorm + Linear fast path of cobevt_linear_rows_small_k (gemm_rows3.hip); -1 when the shape does not qualify
int launch_ln_linear64(const void* in, const void* wfrag, const float* bias, void* out, int M, int N, int K, long lda, int act, float eps,
                       hipStream_t stream) {
    static const bool on = [] { const char* e = getenv("COBEVT_LN_LINEAR64"); return !(e && e[0] == '0'); }();
    if (!on || K != 64 || lda != 64 || N < 32 || N > 192 || N % 32 || M < 65536) return -1;
    LnLin64Params p;
    p.in = (const bf16_t*)in; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.out = (bf16_t*)out;
    p.M = M; p.N = N; p.act = act; p.eps = eps;
    switch (N / 32) {
        case 1: return launch_nnt<1>(p, stream);
        case 2: return launch_nnt<2>(p, stream);
        case 3: return launch_nnt<3>(p, stream);
        case 4: return launch_nnt<4>(p, stream);
        case 5: return launch_nnt<5>(p, stream);
        default: return launch_nnt<6>(p, stream);
    }
}

}  // namespace cobevt
