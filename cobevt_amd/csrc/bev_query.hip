// FAX cross attention #1, query side in one launch (gfx950, bf16):
//
//     query[b, cam, pix, :] = L2norm_c( bev_embed(world[pix]) - cam_embed(cam) ) + x[b, pix, :]      fax_modules.py:370-375,387-388
//     q[b, cam, pix, :]     = to_q( LayerNorm(query) )                                                 fax_modules.py:201,211-216
//
// The (b, n, HW, 128) query (84 MB on the 5-agent level-0 map) used to be written by the embedding kernel (34 us) and read back
// by the to_q GEMM (63 us, whose 10,240 32-row workgroups also re-stream the 32-KB weight from L2).  Here a WAVE owns 32 BEV
// positions: it loads their x rows once (16 bytes per lane and k-group at natural addresses: lane (q, h) holds channels
// 16 g + 8 h .. + 7 of row q - exactly an MFMA B operand), and for every camera recomputes the embedding in registers, adds x,
// rounds to bf16 (what the embedding kernel would have stored), normalises (the two lanes of a row exchange their partial sums
// with one v_permlane32_swap) and feeds the packed rows straight to the to_q MFMAs, whose weight fragments sit in LDS for the
// life of the (persistent) workgroup.  No barrier after the prologue; HBM sees x once and q once.
#include "bev_query.hpp"

namespace cobevt {

namespace {

constexpr int kMaxCams = 32;                       // B * n cameras whose offsets fit the LDS table
constexpr int kWBytes = 32 * 1024;                 // 32 fragments
// LDS: weights | coefxy[128] float2 (w_bev) | coefz[B n][128] (b_bev - w_cam . c) | bias[128]
constexpr int kLdsBytes = kWBytes + 128 * 8 + kMaxCams * 128 * 4 + 128 * 4;

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

template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void bev_query_kernel(BevQueryParams p, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* wl = (const uint4*)smem;
    float2* cxy = (float2*)(smem + kWBytes);
    float* cz = (float*)(smem + kWBytes + 128 * 8);
    float* sbias = cz + kMaxCams * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    {
        // (every thread's loads first, then its LDS stores: rolled load -> store loops cost one exposed round trip per iteration)
        uint4* dst = (uint4*)smem;
        constexpr int PER = 32 * 64 / (NW * 64);
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) tmp[u] = p.wfrag[tid + u * NW * 64];
        const int ncam = p.B * p.n;
        constexpr int CPER = (kMaxCams * 128 + NW * 64 - 1) / (NW * 64);
        float4 wc[CPER];
        float e3[CPER], e7[CPER], e11[CPER], e15[CPER], bb[CPER];
#pragma unroll
        for (int u = 0; u < CPER; ++u) {
            const int i = tid + u * NW * 64, ic = i < ncam * 128 ? i : 0;
            const int bn = ic >> 7, ch = ic & 127;
            const float* E = p.E_inv + (size_t)bn * 16;
            wc[u] = *(const float4*)(p.w_cam + ch * 4);
            e3[u] = E[3]; e7[u] = E[7]; e11[u] = E[11]; e15[u] = E[15];
            bb[u] = p.b_bev[ch];
        }
        float2 cc = make_float2(0.f, 0.f);
        float bq = 0.f;
        if (tid < 128) {
            cc = make_float2(p.w_bev[2 * tid], p.w_bev[2 * tid + 1]);
            bq = p.bias ? p.bias[tid] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) dst[tid + u * NW * 64] = tmp[u];
#pragma unroll
        for (int u = 0; u < CPER; ++u) {
            const int i = tid + u * NW * 64;
            if (i < ncam * 128) cz[i] = bb[u] - (wc[u].x * e3[u] + wc[u].y * e7[u] + wc[u].z * e11[u] + wc[u].w * e15[u]);
        }
        if (tid < 128) {
            cxy[tid] = cc;
            sbias[tid] = bq;
        }
    }
    __syncthreads();

    int opq = 0;                                       // opaque zero: keeps the loop-invariant LDS reads inside the loop
    // block = (batch b, 32-pixel block, camera), camera fastest: the cameras of a pixel block run on neighbouring waves at the
    // same time and share the x rows in L2 / the vector L1
    const int pblk = p.hw >> 5;                        // 32-pixel blocks per map
    const int nwaves = gridDim.x * NW;
    for (int blk = blockIdx.x * NW + wave; blk < nblk; blk += nwaves) {
        asm volatile("" : "+v"(opq));
        const int pb = blk / p.n, cam = blk - pb * p.n;
        const int b = pb / pblk, pix = (pb - b * pblk) * 32 + ql;
        const bf16_t* xrow = p.x + ((size_t)(p.x_bcast ? 0 : b) * p.hw + pix) * 128;
        uint4 xr[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) xr[g] = *(const uint4*)(xrow + 16 * g + 8 * h);
        const float wx = p.world[pix], wy = p.world[p.hw + pix];

        {
            const int bn = b * p.n + cam;
            const float* czc = cz + bn * 128 + opq;
            // pass 1: squared norm of the embedding over the row's 128 channels (this lane's 64 + the partner's)
            float ss = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int ch = 16 * g + 8 * h + e;
                    const float2 c = cxy[ch + opq];
                    const float em = c.x * wx + c.y * wy + czc[ch];
                    ss += em * em;
                }
            }
            const float inv = 1.0f / (sqrtf(xhalf_sum(ss)) + 1e-7f);
            int opq2 = 0;
            asm volatile("" : "+v"(opq2));
            // pass 2: query = embedding * inv + x, rounded to bf16 (the stored query of the unfused path); row sum
            uint4 qv[8];
            float s = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float v[8], xf[8];
                chunk_to_f32<bf16_t>(xr[g], xf);
#pragma unroll
                for (int e = 0; e < 8; ++e) {                // (re-read, not kept from pass 1: 192 coefficients per lane)
                    const int ch = 16 * g + 8 * h + e;
                    const float2 c = cxy[ch + opq2];
                    v[e] = (c.x * wx + c.y * wy + czc[ch + opq2]) * inv + xf[e];
                }
                qv[g] = f32_to_chunk<bf16_t>(v);
                chunk_to_f32<bf16_t>(qv[g], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s += v[e];
            }
            const float mean = xhalf_sum(s) * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float v[8];
                chunk_to_f32<bf16_t>(qv[g], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
            }
            const float rstd = rsqrtf(xhalf_sum(q) * (1.0f / 128.0f) + p.ln_eps);
#pragma unroll
            for (int g = 0; g < 8; ++g) {
                float v[8];
                chunk_to_f32<bf16_t>(qv[g], v);
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = (v[e] - mean) * rstd;
                qv[g] = f32_to_chunk<bf16_t>(v);
            }
            // to_q: 4 column tiles x 8 k-groups; 16-byte stores after the half swap
            bf16_t* orow = p.out + ((size_t)bn * p.hw + pix) * 128;
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                f32x16 acc;
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int g = 0; g < 8; ++g) mfma_kgroup<bf16_t>(wl[(t * 8 + g) * 64 + lane + opq], qv[g], acc);
#pragma unroll
                for (int m = 0; m < 2; ++m) {
                    const float4 b0 = *(const float4*)(sbias + 32 * t + 16 * m + 4 * h + opq);
                    const float4 b1 = *(const float4*)(sbias + 32 * t + 16 * m + 8 + 4 * h + opq);
                    uint2 r0 = make_uint2(pack_bf2(acc[8 * m] + b0.x, acc[8 * m + 1] + b0.y), pack_bf2(acc[8 * m + 2] + b0.z, acc[8 * m + 3] + b0.w));
                    uint2 r1 = make_uint2(pack_bf2(acc[8 * m + 4] + b1.x, acc[8 * m + 5] + b1.y), pack_bf2(acc[8 * m + 6] + b1.z, acc[8 * m + 7] + b1.w));
                    half_swap(r0, r1);
                    *(uint4*)(orow + 32 * t + 16 * m + 8 * h) = make_uint4(r0.x, r0.y, r1.x, r1.y);
                }
            }
        }
    }
}

}  // namespace

// gemm_rows3.hip's entry point cobevt_bev_embed_linear_rows_small_k hands the 128 -> 128 LayerNorm case to this launcher;
// returns -1 when the shape does not qualify
int launch_bev_query(const BevQueryParams& p, hipStream_t stream) {
    if (p.hw % 32 != 0 || (long)p.B * p.n > kMaxCams || p.B < 1 || p.n < 1) return -1;
    constexpr int NW = 4;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)bev_query_kernel<NW>, hipFuncAttributeMaxDynamicSharedMemorySize, kLdsBytes);
    }
    const int nblk = p.B * (p.hw >> 5) * p.n;
    int blocks = (nblk + NW - 1) / NW;
    if (blocks > 256 * 3) blocks = 256 * 3;                // persistent: three workgroups (12 waves) per CU
    hipLaunchKernelGGL((bev_query_kernel<NW>), dim3((unsigned)blocks), dim3(NW * 64), kLdsBytes, stream, p, nblk);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt
