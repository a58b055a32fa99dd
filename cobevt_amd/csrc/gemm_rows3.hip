// Dense-row GEMM for K <= 512, the row chain's way (gfx950, bf16):
//   out[m][n] = act( f(A[m][:]) . W[n][:] + bias[n] + residual[m][n] ),   f = LayerNorm | per-channel affine (+ReLU) | id
// reference: the same Linear / 1x1-conv sites as gemm_rows.hip (to_q / to_k / to_v / to_qkv behind a LayerNorm, feature_proj /
// feature_linear behind BN + ReLU, the Bottleneck 1x1 convs; fax_modules.py:193-203,283-300,472, swap_fusion_modules.py:93).
//
// gemm_rows.hip stages a 128 x 128 weight tile through LDS per workgroup and runs load -> LDS -> MFMA -> staging -> store as
// serial phases of two co-resident 8-wave workgroups; its s_memtime trace is dominated by the per-CU load / store pipes and
// the phases do not overlap.  The fused row chain (row_chain.hip) moves the same kind of GEMM 2-3x faster per phase with
// 32-row workgroups of 4 waves whose weights arrive as MFMA fragments straight from L2 (one coalesced 1-KB load per wave and
// k-group, host-side re-layout, no weight LDS) and whose many small workgroups per CU overlap each other's phases.  This
// kernel is that structure for a single GEMM: 32 rows per workgroup (8 threads per row stage + normalise the A rows into
// LDS), wave w owns columns [32w, 32w+32) of every 128-column pass with the next pass's fragments in flight (two register
// sets), D = W.X^T so a lane owns one row and runs of four consecutive columns (bias / residual / activation in registers),
// results leave through a bf16 LDS tile as 16-byte coalesced stores.
#include "common.hpp"
#include "bev_query.hpp"

namespace cobevt {

struct Gr3Params {
    const bf16_t* in;       // [M][lda]
    const uint4* wfrag;     // fragment-ordered [N_p/32][8][64 lanes][16 B]
    const float* bias;      // [N] or null
    const bf16_t* residual; // [M][N] or null
    const float* pre_scale; // [K] or null (with pre_shift)
    const float* pre_shift;
    bf16_t* out;            // [M][N]
    int M, N, K, Kp;        // Kp = K rounded up to 128 (<= 512): the fragment array has Kp / 16 k-groups per 32-column tile
    long lda;
    int pre_relu, act, ln;
    float ln_eps;
    int in_stride, src_H, src_W, in_H, in_W;   // in_stride > 1: row m = (n, oy, ox) of an (src_H, src_W) map reads input pixel
                                               // (oy * in_stride, ox * in_stride) of an (in_H, in_W) map (1x1 / stride-2 conv)
    // EMB: the A rows are the BEV query  x[b][pix] + L2norm_c(w_bev . world[pix] + b_bev - w_cam . E_inv[b, cam][:, 3])
    // (fax_modules.py:370-375,387-388) of row m = ((b * n + cam) * hw + pix), produced while staging; `in` is x (B | 1, hw, K)
    const float* emb_E; const float* emb_world; const float* emb_wbev; const float* emb_bbev; const float* emb_wcam;
    int emb_n, emb_hw, emb_xbcast;
    // navg > 1: A row m = (b, r) is the MEAN of the navg rows in[b * avg_batch_stride + j * avg_stride + r * lda], j < navg
    // (SwapFusionEncoder.mlp_head: Reduce('b m d h w -> b d h w', 'mean') -> LayerNorm -> Linear, swap_fusion_modules.py:275-281)
    int navg, avg_rows_per_batch;
    long avg_stride, avg_batch_stride;
};

constexpr int kG3Row = 256 + 16;            // staged output row: 128 bf16 + pad
constexpr int kG3Rows = 32;
// LDS: A rows [32][Kp * 2 + 16] | output tile [32][272] | N_p floats of bias

// ROWS = 32 (4 waves) or 64 (8 waves = 2 row halves x 4 column tiles: every weight fragment serves twice the rows)
template <bool EMB, int ROWS = 32>
__global__ __launch_bounds__(ROWS * 8, 4) void gemm_rows3_kernel(Gr3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int arow = p.Kp * 2 + 16;
    unsigned char* As = smem;
    unsigned char* Ys = smem + ROWS * arow;
    float* sb = (float*)(Ys + ROWS * kG3Row);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave & 3, wm = wave >> 2;
    const int h = lane >> 5, ql = lane & 31;
    const int m0 = blockIdx.x * ROWS;
    const int row = wm * 32 + ql;
    const bool row_ok = m0 + row < p.M;
    const int npn = (p.N + 127) / 128, nkt = p.Kp >> 7, nkg = p.Kp >> 4;
    const int nsteps = npn * nkt;
    const int cbase = wn * 32 + 4 * h;

    // step s = (pass, k-tile): the eight fragments of columns [128 pass + 32 wn, +32) x k-groups [8 kt, +8)
    auto load_frags = [&](uint4 (&b)[8], int step) {
        const int pass = step / nkt, kt = step - pass * nkt;
        const uint4* src = p.wfrag + ((size_t)(pass * 4 + wn) * nkg + kt * 8) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    uint4 fa[8], fb[8];
    load_frags(fa, 0);

    // bias of all passes into LDS (zero padded), unconditional clamped loads
    for (int i = tid; i < npn * 128; i += ROWS * 8) {
        const float b = p.bias ? p.bias[i < p.N ? i : 0] : 0.f;
        sb[i] = i < p.N ? b : 0.f;
    }
    // EMB: per-channel (w_bev0, w_bev1, b_bev - w_cam . c) of this workgroup's camera (hw % 32 == 0: one camera per workgroup)
    float4* coef = (float4*)(sb + npn * 128);
    if (EMB) {
        const int bn = m0 / p.emb_hw;
        if (tid < 128) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            const int k = tid < p.K ? tid : 0;
            const float* E = p.emb_E + (size_t)bn * 16;
            const float4 wc = *(const float4*)(p.emb_wcam + k * 4);
            c.x = p.emb_wbev[k * 2];
            c.y = p.emb_wbev[k * 2 + 1];
            c.z = p.emb_bbev[k] - (wc.x * E[3] + wc.y * E[7] + wc.z * E[11] + wc.w * E[15]);
            coef[tid] = tid < p.K ? c : make_float4(0.f, 0.f, 0.f, 0.f);
        }
        __syncthreads();
    }
    // ---- stage the 32 A rows: 8 threads per row, 16 channels per 128-channel tile each; LayerNorm (single tile) or the
    // per-channel pre-activation in flight
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool ok = m0 + r < p.M;
        size_t arow_idx = ok ? m0 + r : 0;
        if (p.in_stride > 1) {
            const int hw = p.src_H * p.src_W;
            const int n = (int)(arow_idx / hw), rem = (int)(arow_idx - (size_t)n * hw);
            const int oy = rem / p.src_W, ox = rem - oy * p.src_W;
            arow_idx = ((size_t)n * p.in_H + (size_t)oy * p.in_stride) * p.in_W + (size_t)ox * p.in_stride;
        }
        float wx = 0.f, wy = 0.f;
        if (EMB) {                                            // x row of (b, pix); world coordinates of pix
            const int m = (int)arow_idx, bn = m / p.emb_hw, pix = m - bn * p.emb_hw;
            wx = p.emb_world[pix];
            wy = p.emb_world[p.emb_hw + pix];
            arow_idx = (size_t)(p.emb_xbcast ? 0 : bn / p.emb_n) * p.emb_hw + pix;
        }
        const bf16_t* src = p.in + arow_idx * p.lda;
        if (p.navg > 1) {
            const long bb = (long)arow_idx / p.avg_rows_per_batch, rr = (long)arow_idx - bb * p.avg_rows_per_batch;
            src = p.in + bb * p.avg_batch_stride + rr * p.lda;
        }
        for (int kt = 0; kt < nkt; ++kt) {
            const int kb = kt * 128 + sub * 16;
            uint4 raw[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) raw[j] = *(const uint4*)(src + (kb + j * 8 < p.K ? kb + j * 8 : 0));
            float v[16];
            chunk_to_f32<bf16_t>(raw[0], v);
            chunk_to_f32<bf16_t>(raw[1], v + 8);
            if (p.navg > 1) {
                for (int a = 1; a < p.navg; ++a) {
                    float u[16];
                    const bf16_t* sa = src + a * p.avg_stride;
                    chunk_to_f32<bf16_t>(*(const uint4*)(sa + (kb < p.K ? kb : 0)), u);
                    chunk_to_f32<bf16_t>(*(const uint4*)(sa + (kb + 8 < p.K ? kb + 8 : 0)), u + 8);
#pragma unroll
                    for (int e = 0; e < 16; ++e) v[e] += u[e];
                }
                const float inv = 1.0f / (float)p.navg;
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] *= inv;
            }
#pragma unroll
            for (int e = 0; e < 16; ++e)
                if (!ok || kb + e >= p.K) v[e] = 0.f;
            if (EMB) {                                        // single tile (K <= 128): v holds x, add the normalised embedding
                float em[16], ss = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const float4 c = coef[kb + e];
                    em[e] = (kb + e) < p.K ? (c.x * wx + c.y * wy + c.z) : 0.f;
                    ss += em[e] * em[e];
                }
                ss += __shfl_xor(ss, 1, 64); ss += __shfl_xor(ss, 2, 64); ss += __shfl_xor(ss, 4, 64);
                const float inv = 1.0f / (sqrtf(ss) + 1e-7f);
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = ok ? em[e] * inv + v[e] : 0.f;
                // rounded exactly as cobevt_fax_bev_embed would have stored the query, then normalised
                chunk_to_f32<bf16_t>(f32_to_chunk<bf16_t>(v), v);
                chunk_to_f32<bf16_t>(f32_to_chunk<bf16_t>(v + 8), v + 8);
            }
            if (p.ln) {                                   // Kp == 128 (checked by the entry point): the row is here
                float s = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) s += v[e];
                s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
                const float mean = s / (float)p.K;
                float q = 0.f;
#pragma unroll
                for (int e = 0; e < 16; ++e) { const float d = (kb + e) < p.K ? v[e] - mean : 0.f; q += d * d; }
                q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
                const float rstd = rsqrtf(q / (float)p.K + p.ln_eps);
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = (kb + e) < p.K ? (v[e] - mean) * rstd : 0.f;
            } else if (p.pre_scale) {
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const int k = kb + e;
                    const float4 sc = *(const float4*)(p.pre_scale + (k < p.K ? k : 0)), sh = *(const float4*)(p.pre_shift + (k < p.K ? k : 0));
                    const float x0 = v[e] * sc.x + sh.x, x1 = v[e + 1] * sc.y + sh.y, x2 = v[e + 2] * sc.z + sh.z, x3 = v[e + 3] * sc.w + sh.w;
                    const bool in = ok && k < p.K;                  // K % 8 == 0: a run of four is inside or outside
                    v[e] = in ? (p.pre_relu ? fmaxf(x0, 0.f) : x0) : 0.f;
                    v[e + 1] = in ? (p.pre_relu ? fmaxf(x1, 0.f) : x1) : 0.f;
                    v[e + 2] = in ? (p.pre_relu ? fmaxf(x2, 0.f) : x2) : 0.f;
                    v[e + 3] = in ? (p.pre_relu ? fmaxf(x3, 0.f) : x3) : 0.f;
                }
            }
            *(uint4*)(As + r * arow + kb * 2) = f32_to_chunk<bf16_t>(v);
            *(uint4*)(As + r * arow + kb * 2 + 16) = f32_to_chunk<bf16_t>(v + 8);
        }
    }
    __syncthreads();

    const int abase = row * arow + h * 16;
    f32x16 acc;
    uint2 rs[4];
    auto step_body = [&](int step, const uint4 (&cur)[8], uint4 (&nxt)[8]) {
        const int pass = step / nkt, kt = step - pass * nkt;
        if (step + 1 < nsteps) load_frags(nxt, step + 1);
        if (kt == 0) {
            // residual pieces of this lane's (row, column runs): in flight under the MFMAs
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int col0 = pass * 128 + cbase + 8 * k;
                const bool ok = p.residual && row_ok && col0 < p.N;
                rs[k] = make_uint2(0, 0);
                if (p.residual) rs[k] = *(const uint2*)(p.residual + (ok ? (size_t)(m0 + row) * p.N + col0 : 0));
                if (!ok) rs[k] = make_uint2(0, 0);
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        const int kleft = p.K - kt * 128;
        const int ng = kleft >= 128 ? 8 : (kleft * 2 + 31) / 32;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g < ng) {
                const uint4 af = *(const uint4*)(As + abase + (kt * 8 + g) * 32);
                mfma_kgroup<bf16_t>(cur[g], af, acc);
            }
        if (kt + 1 < nkt) return;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = *(const float4*)(sb + col0);
            float v0 = acc[4 * k] + b.x + bf2f(rs[k].x & 0xffff), v1 = acc[4 * k + 1] + b.y + bf2f(rs[k].x >> 16);
            float v2 = acc[4 * k + 2] + b.z + bf2f(rs[k].y & 0xffff), v3 = acc[4 * k + 3] + b.w + bf2f(rs[k].y >> 16);
            if (p.act) { v0 = apply_act<bf16_t>(v0, p.act); v1 = apply_act<bf16_t>(v1, p.act); v2 = apply_act<bf16_t>(v2, p.act); v3 = apply_act<bf16_t>(v3, p.act); }
            *(uint2*)(Ys + row * kG3Row + (cbase + 8 * k) * 2) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
        }
        __syncthreads();                              // ROWS x 128 result staged in Ys
        {
            const int r = tid >> 3, sub = tid & 7;
            if (m0 + r < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c0 = pass * 128 + sub * 16 + j * 8;
                    if (c0 < p.N) *(uint4*)(p.out + (size_t)(m0 + r) * p.N + c0) = *(const uint4*)(Ys + r * kG3Row + sub * 32 + j * 16);
                }
            }
        }
        if (pass + 1 < npn) __syncthreads();          // Ys is rewritten by the next pass
    };
    for (int step = 0; step < nsteps; step += 2) {
        step_body(step, fa, fb);
        if (step + 1 < nsteps) step_body(step + 1, fb, fa);
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
namespace cobevt {
int launch_linear_rows_f32(const void* in, const void* wfrag, const float* bias, const void* residual, const float* pre_scale,
                           const float* pre_shift, void* out, const long* dims, float ln_eps, hipStream_t stream);   // gemm_rows3_f32.hip
int launch_ln_linear64(const void* in, const void* wfrag, const float* bias, void* out, int M, int N, int K, long lda, int act, float eps,
                       hipStream_t stream);                                                                          // ln_linear64.hip
}

extern "C" int cobevt_linear_rows_small_k(const void* in, const void* wfrag, const float* bias, const void* residual,
                                          const float* pre_scale, const float* pre_shift, void* out, const long* dims, float ln_eps,
                                          hipStream_t stream) {
    // dims: [dtype(0), M, N, K, lda, pre_relu, act, ln, in_stride, src_H, src_W, in_H, in_W, rows_per_workgroup]
    if (!in || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] == 1) {                                  // fp32 storage: gemm_rows3_f32.hip (round 6)
        const int rc = cobevt::launch_linear_rows_f32(in, wfrag, bias, residual, pre_scale, pre_shift, out, dims, ln_eps, stream);
        return rc < 0 ? COBEVT_ERR_UNSUPPORTED : rc;
    }
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    Gr3Params p;
    p.in = (const bf16_t*)in; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.residual = (const bf16_t*)residual;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.out = (bf16_t*)out;
    p.M = (int)dims[1]; p.N = (int)dims[2]; p.K = (int)dims[3]; p.lda = dims[4];
    p.pre_relu = (int)dims[5]; p.act = (int)dims[6]; p.ln = (int)dims[7]; p.ln_eps = ln_eps;
    p.in_stride = (int)dims[8]; p.src_H = (int)dims[9]; p.src_W = (int)dims[10]; p.in_H = (int)dims[11]; p.in_W = (int)dims[12];
    p.Kp = (p.K + 127) / 128 * 128;
    if (p.M < 1 || p.N < 8 || p.N % 8 || p.N > 4096 || p.K < 8 || p.K > 512 || p.K % 8 || p.lda < p.K || p.lda % 8) return COBEVT_ERR_SHAPE;
    if (p.ln && p.K > 128) return COBEVT_ERR_UNSUPPORTED;          // LayerNorm fusion needs the row in one 128-channel tile
    if (p.in_stride < 1 || (p.in_stride > 1 && (p.src_H < 1 || p.src_W < 1 || p.in_H < 1 || p.in_W < 1 || p.M % (p.src_H * p.src_W))))
        return COBEVT_ERR_SHAPE;
    if (p.in_stride > 1 && residual) return COBEVT_ERR_UNSUPPORTED;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return COBEVT_ERR_ARG;
    if (p.ln && pre_scale) return COBEVT_ERR_UNSUPPORTED;
    if (p.act < 0 || p.act > 4) return COBEVT_ERR_ARG;
    if (p.ln && !residual && !pre_scale && !p.pre_relu && p.in_stride == 1) {   // LayerNorm + Linear of a big 64-channel map: independent waves
        const int rc = cobevt::launch_ln_linear64(in, wfrag, bias, out, p.M, p.N, p.K, p.lda, p.act, ln_eps, stream);
        if (rc >= 0) return rc;
    }
    const int rows = dims[13] == 64 ? 64 : 32;                     // dims[13]: rows per workgroup (0 = 32)
    const size_t lds = (size_t)rows * (p.Kp * 2 + 16) + (size_t)rows * kG3Row + (size_t)((p.N + 127) / 128) * 128 * 4;
    const unsigned blocks = (unsigned)((p.M + rows - 1) / rows);
    p.emb_E = p.emb_world = p.emb_wbev = p.emb_bbev = p.emb_wcam = nullptr;
    p.emb_n = p.emb_hw = 1; p.emb_xbcast = 0;
    p.navg = 1; p.avg_rows_per_batch = 1; p.avg_stride = p.avg_batch_stride = 0;
    if (rows == 64) {
        static cobevt::PerDeviceOnce attr_once;
        if (attr_once.first()) {
            (void)hipFuncSetAttribute((const void*)gemm_rows3_kernel<false, 64>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1040 + 64 * kG3Row + 4096 * 4);
        }
        hipLaunchKernelGGL((gemm_rows3_kernel<false, 64>), dim3(blocks), dim3(512), lds, stream, p);
    } else {
        hipLaunchKernelGGL((gemm_rows3_kernel<false, 32>), dim3(blocks), dim3(256), lds, stream, p);
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_bev_embed_linear_rows_small_k(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                                                    const float* w_cam, const void* x, const void* wfrag, const float* bias, void* out,
                                                    const long* dims, float ln_eps, hipStream_t stream) {
    // dims: [dtype(0), B, n, hw, D (= K <= 128), N, ln, x_bcast]
    if (!E_inv || !world || !w_bev || !b_bev || !w_cam || !x || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    const long B = dims[1], n = dims[2], hw = dims[3];
    if (dims[4] == 128 && dims[5] == 128 && dims[6] && hw % 32 == 0 && B >= 1 && n >= 1 && B * n * hw <= 0x7fffffffL) {
        // the FAX level-0 shape: the wave-level kernel (bev_query.hip) - rows never leave the registers, weights in LDS
        BevQueryParams q;
        q.E_inv = E_inv; q.world = world; q.w_bev = w_bev; q.b_bev = b_bev; q.w_cam = w_cam; q.x = (const bf16_t*)x;
        q.wfrag = (const uint4*)wfrag; q.bias = bias; q.out = (bf16_t*)out;
        q.B = (int)B; q.n = (int)n; q.hw = (int)hw; q.x_bcast = (int)dims[7]; q.ln_eps = ln_eps;
        const int rc = launch_bev_query(q, stream);
        if (rc >= 0) return rc;
    }
    Gr3Params p;
    p.in = (const bf16_t*)x; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.residual = nullptr;
    p.pre_scale = p.pre_shift = nullptr; p.out = (bf16_t*)out;
    p.K = (int)dims[4]; p.N = (int)dims[5]; p.ln = (int)dims[6]; p.ln_eps = ln_eps;
    p.Kp = 128; p.lda = p.K; p.pre_relu = 0; p.act = 0;
    p.in_stride = 1; p.src_H = p.src_W = p.in_H = p.in_W = 1;
    if (B < 1 || n < 1 || hw < 1 || hw % kG3Rows != 0 || B * n * hw > 0x7fffffffL) return COBEVT_ERR_SHAPE;   // one camera per workgroup
    if (p.N < 8 || p.N % 8 || p.N > 4096 || p.K < 8 || p.K > 128 || p.K % 8) return COBEVT_ERR_SHAPE;
    p.M = (int)(B * n * hw);
    p.emb_E = E_inv; p.emb_world = world; p.emb_wbev = w_bev; p.emb_bbev = b_bev; p.emb_wcam = w_cam;
    p.emb_n = (int)n; p.emb_hw = (int)hw; p.emb_xbcast = (int)dims[7];
    p.navg = 1; p.avg_rows_per_batch = 1; p.avg_stride = p.avg_batch_stride = 0;
    const size_t lds = (size_t)kG3Rows * (p.Kp * 2 + 16) + (size_t)kG3Rows * kG3Row + (size_t)((p.N + 127) / 128) * 128 * 4 + 128 * 16;
    const unsigned blocks = (unsigned)((p.M + kG3Rows - 1) / kG3Rows);
    hipLaunchKernelGGL((gemm_rows3_kernel<true, 32>), dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h: out (B * R, N) = act(LayerNorm?(mean_j in[b][j][r][:]) . W^T + bias), in (B, L, R, K)
extern "C" int cobevt_mean_linear_rows_small_k(const void* in, const void* wfrag, const float* bias, void* out, const long* dims,
                                               float ln_eps, hipStream_t stream) {
    // dims: [dtype(0), B, L, R, K, N, ln, act]
    if (!in || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    const long B = dims[1], Lr = dims[2], R = dims[3];
    Gr3Params p;
    p.in = (const bf16_t*)in; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.residual = nullptr;
    p.pre_scale = p.pre_shift = nullptr; p.out = (bf16_t*)out;
    p.K = (int)dims[4]; p.N = (int)dims[5]; p.ln = (int)dims[6]; p.act = (int)dims[7]; p.ln_eps = ln_eps;
    p.Kp = 128; p.lda = p.K; p.pre_relu = 0;
    p.in_stride = 1; p.src_H = p.src_W = p.in_H = p.in_W = 1;
    if (B < 1 || Lr < 1 || Lr > 64 || R < 1 || B * R > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    if (p.N < 8 || p.N % 8 || p.N > 4096 || p.K < 8 || p.K > 128 || p.K % 8 || p.act < 0 || p.act > 4) return COBEVT_ERR_SHAPE;
    p.M = (int)(B * R);
    p.emb_E = p.emb_world = p.emb_wbev = p.emb_bbev = p.emb_wcam = nullptr;
    p.emb_n = p.emb_hw = 1; p.emb_xbcast = 0;
    p.navg = (int)Lr; p.avg_rows_per_batch = (int)R; p.avg_stride = R * p.K; p.avg_batch_stride = Lr * R * p.K;
    const size_t lds = (size_t)kG3Rows * (p.Kp * 2 + 16) + (size_t)kG3Rows * kG3Row + (size_t)((p.N + 127) / 128) * 128 * 4;
    const unsigned blocks = (unsigned)((p.M + kG3Rows - 1) / kG3Rows);
    hipLaunchKernelGGL((gemm_rows3_kernel<false, 32>), dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
