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
// LDS: to_q fragments (accumulator k order) | embedding operand base[128] uint4 {cx_hi cx_hi | cx_lo cy_hi | cy_hi cy_lo | -}
//      | czw[B n][128] bf16 pair {cz_hi, cz_lo} of (b_bev - w_cam . c) | bias[128]
constexpr int kLdsBytes = kWBytes + 128 * 16 + kMaxCams * 128 * 4 + 128 * 4;

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
// x = hi + lo with both halves bf16 values (|x - hi - lo| <= 2^-17 |x|)
__device__ __forceinline__ void split_bf(float x, float& hi, float& lo) {
    hi = bf2f(f2bf(x));
    lo = bf2f(f2bf(x - hi));
}

// The embedding  em[row][ch] = w_bev[ch] . world[row] + (b_bev[ch] - w_cam[ch] . c_cam)  is a K = 3 matrix product; on the VALU it
// cost 128 LDS coefficient reads and ~400 instructions per 32-row block and pass.  Here it is ONE MFMA per 32-channel tile with both
// operands split into bf16 halves (hi, lo): k-slots {cx_hi wx_hi, cx_hi wx_lo, cx_lo wx_hi, cy_hi wy_hi, cy_hi wy_lo, cy_lo wy_hi,
// cz_hi, cz_lo} - fp32 accumulation, dropped terms <= 2^-16 relative - and the result lands in accumulator-register order, which is
// the order the rest of the chain runs in anyway (x is brought there with a half swap, to_q's contraction index is stored that way).
template <int NW>
__global__ __launch_bounds__(NW * 64, 3) void bev_query_kernel(BevQueryParams p, int nblk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const uint4* wl = (const uint4*)smem;
    const uint4* ebase = (const uint4*)(smem + kWBytes);
    const uint32_t* czw = (const uint32_t*)(smem + kWBytes + 128 * 16);
    const float* sbias = (const float*)(czw + kMaxCams * 128);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    {
        // (every thread's loads first, then its LDS stores: rolled load -> store loops cost one exposed round trip per iteration)
        uint4* dst = (uint4*)smem;
        constexpr int PER = 32 * 64 / (NW * 64);
        uint4 tmp[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {                // to_q fragments -> accumulator k order (half swap of the 8-byte runs)
            const int i = tid + u * NW * 64, ln = i & 63, hh = ln >> 5, q = ln & 31;
            const uint4* base = p.wfrag + (size_t)(i >> 6) * 64;
            const uint2 lo = *(const uint2*)((const unsigned char*)(base + q) + 8 * hh);
            const uint2 hi = *(const uint2*)((const unsigned char*)(base + 32 + q) + 8 * hh);
            tmp[u] = make_uint4(lo.x, lo.y, hi.x, hi.y);
        }
        const int ncam = p.B * p.n;
        float2 cc = make_float2(0.f, 0.f);
        float bq = 0.f;
        if (tid < 128) {
            cc = make_float2(p.w_bev[2 * tid], p.w_bev[2 * tid + 1]);
            bq = p.bias ? p.bias[tid] : 0.f;
        }
#pragma unroll
        for (int u = 0; u < PER; ++u) dst[tid + u * NW * 64] = tmp[u];
        uint32_t* czw_w = (uint32_t*)(smem + kWBytes + 128 * 16);
        constexpr int CB = 8;                          // camera offsets: batches of eight entries per thread (two round trips at most)
        for (int base = 0; base < ncam * 128; base += CB * NW * 64) {
            float4 wc[CB];
            float e3[CB], e7[CB], e11[CB], e15[CB], bb[CB];
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int i = base + tid + u * NW * 64, ic = i < ncam * 128 ? i : 0;
                const int bn = ic >> 7, ch = ic & 127;
                const float* E = p.E_inv + (size_t)bn * 16;
                wc[u] = *(const float4*)(p.w_cam + ch * 4);
                e3[u] = E[3]; e7[u] = E[7]; e11[u] = E[11]; e15[u] = E[15];
                bb[u] = p.b_bev[ch];
            }
#pragma unroll
            for (int u = 0; u < CB; ++u) {
                const int i = base + tid + u * NW * 64;
                if (i < ncam * 128) {
                    const float cz = bb[u] - (wc[u].x * e3[u] + wc[u].y * e7[u] + wc[u].z * e11[u] + wc[u].w * e15[u]);
                    float hi, lo;
                    split_bf(cz, hi, lo);
                    czw_w[i] = pack_bf2(hi, lo);
                }
            }
        }
        if (tid < 128) {
            float xh, xl, yh, yl;
            split_bf(cc.x, xh, xl);
            split_bf(cc.y, yh, yl);
            ((uint4*)(smem + kWBytes))[tid] = make_uint4(pack_bf2(xh, xh), pack_bf2(xl, yh), pack_bf2(yh, yl), 0u);
            ((float*)(czw_w + kMaxCams * 128))[tid] = bq;
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
        const int bn = b * p.n + cam;
        const bf16_t* xrow = p.x + ((size_t)(p.x_bcast ? 0 : b) * p.hw + pix) * 128;
        uint4 xr[8];
#pragma unroll
        for (int g = 0; g < 8; ++g) xr[g] = *(const uint4*)(xrow + 16 * g + 8 * h);
        const float wx = p.world[pix], wy = p.world[p.hw + pix];
        // B operand of the embedding product: this row's {wx_hi wx_lo | wx_hi wy_hi | wy_lo wy_hi | 1 1} in the k-slice of lanes 0-31
        uint4 wb = make_uint4(0u, 0u, 0u, 0u);
        {
            float xh, xl, yh, yl;
            split_bf(wx, xh, xl);
            split_bf(wy, yh, yl);
            if (h == 0) wb = make_uint4(pack_bf2(xh, xl), pack_bf2(xh, yh), pack_bf2(yl, yh), 0x3f803f80u);
        }
        // ---- embedding, 4 column tiles in accumulator order; squared norm over the row's 128 channels
        float em[4][16];
        float ss = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            uint4 ea = ebase[32 * t + ql + opq];
            ea.w = czw[bn * 128 + 32 * t + ql + opq];
            if (h) ea = make_uint4(0u, 0u, 0u, 0u);
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
            mfma_kgroup<bf16_t>(ea, wb, acc);
#pragma unroll
            for (int r = 0; r < 16; ++r) { em[t][r] = acc[r]; ss += acc[r] * acc[r]; }
        }
        const float inv = 1.0f / (sqrtf(xhalf_sum(ss)) + 1e-7f);
        // ---- query = embedding * inv + x, rounded to bf16 (the stored query of the unfused path); x by half swap: 16-byte piece
        // 2 t + m (channels 32 t + 16 m + 8 h ..) -> the lane's runs 2 m, 2 m + 1 of tile t
        float s = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int m = 0; m < 2; ++m) {
                uint2 s0 = make_uint2(xr[2 * t + m].x, xr[2 * t + m].y), s1 = make_uint2(xr[2 * t + m].z, xr[2 * t + m].w);
                half_swap(s0, s1);
                float* d = &em[t][8 * m];
                d[0] = bf2f(f2bf(d[0] * inv + bf2f(s0.x & 0xffff))); d[1] = bf2f(f2bf(d[1] * inv + bf2f(s0.x >> 16)));
                d[2] = bf2f(f2bf(d[2] * inv + bf2f(s0.y & 0xffff))); d[3] = bf2f(f2bf(d[3] * inv + bf2f(s0.y >> 16)));
                d[4] = bf2f(f2bf(d[4] * inv + bf2f(s1.x & 0xffff))); d[5] = bf2f(f2bf(d[5] * inv + bf2f(s1.x >> 16)));
                d[6] = bf2f(f2bf(d[6] * inv + bf2f(s1.y & 0xffff))); d[7] = bf2f(f2bf(d[7] * inv + bf2f(s1.y >> 16)));
#pragma unroll
                for (int e = 0; e < 8; ++e) s += d[e];
            }
        const float mean = xhalf_sum(s) * (1.0f / 128.0f);
        float q = 0.f;
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) { const float dd = em[t][r] - mean; q += dd * dd; }
        const float rstd = rsqrtf(xhalf_sum(q) * (1.0f / 128.0f) + p.ln_eps);
        uint4 qv[8];                                   // B operands of to_q, k-group 2 t + u = registers [8 u, +8) of tile t
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) em[t][r] = (em[t][r] - mean) * rstd;
            qv[2 * t] = pack8(&em[t][0]);
            qv[2 * t + 1] = pack8(&em[t][8]);
        }
        // ---- to_q: 4 column tiles x 8 k-groups; 16-byte stores after the half swap
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
