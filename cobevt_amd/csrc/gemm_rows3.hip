// Dense-row GEMM for K <= 128, the row chain's way (gfx950, bf16):
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

namespace cobevt {

struct Gr3Params {
    const bf16_t* in;       // [M][lda]
    const uint4* wfrag;     // fragment-ordered [N_p/32][8][64 lanes][16 B]
    const float* bias;      // [N] or null
    const bf16_t* residual; // [M][N] or null
    const float* pre_scale; // [K] or null (with pre_shift)
    const float* pre_shift;
    bf16_t* out;            // [M][N]
    int M, N, K;
    long lda;
    int pre_relu, act, ln;
    float ln_eps;
};

constexpr int kG3Row = 256 + 16;            // 128 bf16 + pad
constexpr int kG3Rows = 32;
constexpr int kG3A = 0, kG3Y = kG3Rows * kG3Row, kG3Bias = 2 * kG3Rows * kG3Row;   // + N_p floats of bias

__global__ __launch_bounds__(256, 4) void gemm_rows3_kernel(Gr3Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + kG3A;
    unsigned char* Ys = smem + kG3Y;
    float* sb = (float*)(smem + kG3Bias);

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int m0 = blockIdx.x * kG3Rows;
    const int row = ql;
    const bool row_ok = m0 + row < p.M;
    const int npn = (p.N + 127) / 128;
    const int cbase = wn * 32 + 4 * h;

    auto load_frags = [&](uint4 (&b)[8], int tile) {
        const uint4* src = p.wfrag + (size_t)tile * 8 * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    uint4 fa[8], fb[8];
    load_frags(fa, wn);

    // bias of all passes into LDS (zero padded), unconditional clamped loads
    for (int i = tid; i < npn * 128; i += 256) {
        const float b = p.bias ? p.bias[i < p.N ? i : 0] : 0.f;
        sb[i] = i < p.N ? b : 0.f;
    }
    // ---- stage the 32 A rows: 8 threads per row, 16 channels each; LayerNorm / pre-activation in flight
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool ok = m0 + r < p.M;
        const bf16_t* src = p.in + (size_t)(ok ? m0 + r : 0) * p.lda;
        uint4 raw[2];
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = sub * 16 + j * 8;
            raw[j] = *(const uint4*)(src + (k < p.K ? k : 0));
        }
        float v[16];
        chunk_to_f32<bf16_t>(raw[0], v);
        chunk_to_f32<bf16_t>(raw[1], v + 8);
#pragma unroll
        for (int e = 0; e < 16; ++e)
            if (!ok || sub * 16 + e >= p.K) v[e] = 0.f;
        if (p.ln) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = (sub * 16 + e) < p.K ? v[e] - mean : 0.f; q += d * d; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            const float rstd = rsqrtf(q / (float)p.K + p.ln_eps);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = (sub * 16 + e) < p.K ? (v[e] - mean) * rstd : 0.f;
        } else if (p.pre_scale) {
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const int k = sub * 16 + e;
                const float4 sc = *(const float4*)(p.pre_scale + (k < p.K ? k : 0)), sh = *(const float4*)(p.pre_shift + (k < p.K ? k : 0));
                const float x0 = v[e] * sc.x + sh.x, x1 = v[e + 1] * sc.y + sh.y, x2 = v[e + 2] * sc.z + sh.z, x3 = v[e + 3] * sc.w + sh.w;
                const bool in = ok && k < p.K;                      // K % 8 == 0: a run of four is inside or outside
                v[e] = in ? (p.pre_relu ? fmaxf(x0, 0.f) : x0) : 0.f;
                v[e + 1] = in ? (p.pre_relu ? fmaxf(x1, 0.f) : x1) : 0.f;
                v[e + 2] = in ? (p.pre_relu ? fmaxf(x2, 0.f) : x2) : 0.f;
                v[e + 3] = in ? (p.pre_relu ? fmaxf(x3, 0.f) : x3) : 0.f;
            }
        }
        *(uint4*)(As + r * kG3Row + sub * 32) = f32_to_chunk<bf16_t>(v);
        *(uint4*)(As + r * kG3Row + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
    }
    __syncthreads();

    const int abase = row * kG3Row + h * 16;
    const int ng = (p.K * 2 + 31) / 32;
    f32x16 acc;
    auto pass_body = [&](int pass, const uint4 (&cur)[8], uint4 (&nxt)[8]) {
        if (pass + 1 < npn) load_frags(nxt, (pass + 1) * 4 + wn);
        // residual pieces of this lane's (row, column runs): in flight under the MFMAs
        uint2 rs[4];
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
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g < ng) {
                const uint4 af = *(const uint4*)(As + abase + g * 32);
                mfma_kgroup<bf16_t>(cur[g], af, acc);
            }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = *(const float4*)(sb + col0);
            float v0 = acc[4 * k] + b.x + bf2f(rs[k].x & 0xffff), v1 = acc[4 * k + 1] + b.y + bf2f(rs[k].x >> 16);
            float v2 = acc[4 * k + 2] + b.z + bf2f(rs[k].y & 0xffff), v3 = acc[4 * k + 3] + b.w + bf2f(rs[k].y >> 16);
            if (p.act) { v0 = apply_act(v0, p.act); v1 = apply_act(v1, p.act); v2 = apply_act(v2, p.act); v3 = apply_act(v3, p.act); }
            *(uint2*)(Ys + row * kG3Row + (cbase + 8 * k) * 2) = make_uint2(pack_bf2(v0, v1), pack_bf2(v2, v3));
        }
        __syncthreads();                              // 32 x 128 result staged in Ys
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
    for (int pass = 0; pass < npn; pass += 2) {
        pass_body(pass, fa, fb);
        if (pass + 1 < npn) pass_body(pass + 1, fb, fa);
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_linear_rows_small_k(const void* in, const void* wfrag, const float* bias, const void* residual,
                                          const float* pre_scale, const float* pre_shift, void* out, const long* dims, float ln_eps,
                                          hipStream_t stream) {
    // dims: [dtype(0), M, N, K, lda, pre_relu, act, ln]
    if (!in || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    Gr3Params p;
    p.in = (const bf16_t*)in; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.residual = (const bf16_t*)residual;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.out = (bf16_t*)out;
    p.M = (int)dims[1]; p.N = (int)dims[2]; p.K = (int)dims[3]; p.lda = dims[4];
    p.pre_relu = (int)dims[5]; p.act = (int)dims[6]; p.ln = (int)dims[7]; p.ln_eps = ln_eps;
    if (p.M < 1 || p.N < 8 || p.N % 8 || p.N > 4096 || p.K < 8 || p.K > 128 || p.K % 8 || p.lda < p.K || p.lda % 8) return COBEVT_ERR_SHAPE;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return COBEVT_ERR_ARG;
    if (p.ln && pre_scale) return COBEVT_ERR_UNSUPPORTED;
    if (p.act < 0 || p.act > 4) return COBEVT_ERR_ARG;
    const size_t lds = (size_t)kG3Bias + (size_t)((p.N + 127) / 128) * 128 * 4;
    const unsigned blocks = (unsigned)((p.M + kG3Rows - 1) / kG3Rows);
    hipLaunchKernelGGL(gemm_rows3_kernel, dim3(blocks), dim3(256), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
