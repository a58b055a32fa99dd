// Dense-row GEMM for K <= 512 in fp32 STORAGE, the row chain's way (gfx950; round 6) - the fp32 counterpart of gemm_rows3.hip:
//   out[m][n] = act( f(A[m][:]) . W[n][:] + bias[n] + residual[m][n] ),   f = LayerNorm (K = 128) | per-channel affine (+ReLU) | id
// reference: the Linear / 1x1-conv sites of gemm_rows.hip (to_q / to_k / to_v behind a LayerNorm, feature_proj / feature_linear behind
// BN + ReLU, the Bottleneck 1x1 convs, the stride-2 projection shortcuts; fax_modules.py:193-203,283-300,472, resnet_ms.py:67-74).
// The fp32 modes ran these through the 128 x 128-tile kernels, whose 5,120-row launches are 40-120 workgroups of serial
// load -> LDS -> MFMA -> staging -> store phases (17-25 us each, 33 launches per frame after the row chain went to one launch).  Same
// structure as row_chain_f32.hip: 32 rows per 4-wave workgroup (160 workgroups for 5,120 rows, several per CU), the A rows staged once
// through stage_x_piece (already split in the split-bf16 / fp16 libraries), wave w owns columns [32 w, 32 w + 32) of every 128-column
// pass, weights as MFMA fragments straight from L2 in two ping-pong sets of eight k-groups with the next set in flight, D = W . X^T so
// bias / residual / activation happen in registers, results leave through an fp32 LDS tile as 16-byte coalesced stores.
#include "common.hpp"

namespace cobevt {

namespace {

constexpr int kRows = 32, kThreads = 256;
constexpr int kYRow = 512 + 16;             // staged output row: 128 fp32 + pad

struct Gr3fParams {
    const float* in;        // [M][lda]
    const uint4* wfrag;     // fragment-ordered [N_p/32][Kp/8][64 lanes][16 B]
    const float* bias;      // [N] or null
    const float* residual;  // [M][N] or null
    const float* pre_scale; // [K] or null (with pre_shift)
    const float* pre_shift;
    float* out;             // [M][N]
    int M, N, K, Kp;        // Kp = K rounded up to 64: the fragment array has Kp / 8 k-groups per 32-column tile
    long lda;
    int pre_relu, act, ln;
    float ln_eps;
    int in_stride, src_H, src_W, in_H, in_W;   // in_stride > 1: row m = (n, oy, ox) of an (src_H, src_W) map reads input pixel
                                               // (oy * in_stride, ox * in_stride) of an (in_H, in_W) map (1x1 / stride-2 conv)
};

__global__ __launch_bounds__(kThreads, 2) void gemm_rows3_f32_kernel(Gr3fParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int arow = p.Kp * 4 + 16;                       // bytes per staged A row (Kp / 4 sixteen-byte slots + 1: odd -> conflict-free)
    unsigned char* As = smem;
    unsigned char* Ys = smem + kRows * arow;
    float* sb = (float*)(Ys + kRows * kYRow);

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int m0 = blockIdx.x * kRows;
    const bool row_ok = m0 + ql < p.M;
    const int npn = (p.N + 127) / 128, nset = p.Kp >> 6, nkg = p.Kp >> 3;
    const int nsteps = npn * nset;
    const int cbase = wn * 32 + 4 * h;

    // step s = (pass, set): the eight fragments of columns [128 pass + 32 wn, +32) x k-groups [8 set, +8)
    auto load_set = [&](uint4 (&b)[8], int step) {
        const int pass = step / nset, set = step - pass * nset;
        const uint4* src = p.wfrag + ((size_t)(pass * 4 + wn) * nkg + set * 8) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    uint4 fa[8], fb[8];
    load_set(fa, 0);

    for (int i = tid; i < npn * 128; i += kThreads) {     // bias of all passes into LDS (zero padded), unconditional clamped loads
        const float b = p.bias ? p.bias[i < p.N ? i : 0] : 0.f;
        sb[i] = i < p.N ? b : 0.f;
    }
    // ---- stage the 32 A rows: 8 threads per row, 16-byte pieces sub, sub + 8, ... ; LayerNorm (K = 128) or the per-channel pre-activation in flight
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
        const float* src = p.in + arow_idx * p.lda;
        if (p.ln) {                                       // K == 128: this thread's 16 channels = pieces 4 sub .. 4 sub + 3
            float v[16];
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (ok) t = *(const float4*)(src + sub * 16 + j * 4);
                v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
            }
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s * (1.0f / 128.0f);
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; q += d * d; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            const float rstd = rsqrtf(q * (1.0f / 128.0f) + p.ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const uint4 o = make_uint4(__float_as_uint((v[4 * j] - mean) * rstd), __float_as_uint((v[4 * j + 1] - mean) * rstd),
                                           __float_as_uint((v[4 * j + 2] - mean) * rstd), __float_as_uint((v[4 * j + 3] - mean) * rstd));
                *(uint4*)(As + r * arow + (sub * 4 + j) * 16) = ok ? stage_x_piece<float>(o) : make_uint4(0, 0, 0, 0);
            }
        } else {
            const int npiece = p.Kp >> 2;                 // 16-byte pieces per row (a multiple of 16)
            for (int base = 0; base < npiece; base += 32) {           // four pieces of the thread in flight at a time
                uint4 v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = base + u * 8 + sub;
                    v[u] = make_uint4(0, 0, 0, 0);
                    if (ok && pc < npiece && pc * 4 < p.K) v[u] = *(const uint4*)(src + pc * 4);
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int pc = base + u * 8 + sub;
                    if (pc >= npiece) continue;
                    uint4 o = v[u];
                    if (p.pre_scale && pc * 4 < p.K) {
                        const float4 sc = *(const float4*)(p.pre_scale + pc * 4), sh = *(const float4*)(p.pre_shift + pc * 4);
                        float f[4] = {fmaf(__uint_as_float(o.x), sc.x, sh.x), fmaf(__uint_as_float(o.y), sc.y, sh.y),
                                      fmaf(__uint_as_float(o.z), sc.z, sh.z), fmaf(__uint_as_float(o.w), sc.w, sh.w)};
                        if (p.pre_relu) { f[0] = fmaxf(f[0], 0.f); f[1] = fmaxf(f[1], 0.f); f[2] = fmaxf(f[2], 0.f); f[3] = fmaxf(f[3], 0.f); }
                        o = ok ? make_uint4(__float_as_uint(f[0]), __float_as_uint(f[1]), __float_as_uint(f[2]), __float_as_uint(f[3])) : make_uint4(0, 0, 0, 0);
                    }
                    *(uint4*)(As + r * arow + pc * 16) = stage_x_piece<float>(o);
                }
            }
        }
    }
    __syncthreads();

    const int abase = ql * arow + h * 16;
    f32x16 acc;
    auto do_step = [&](int step, const uint4 (&b)[8]) {
        const int pass = step / nset, set = step - pass * nset;
        if (set == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) mfma_kgroup_xs<float>(b[g], *(const uint4*)(As + abase + set * 256 + g * 32), acc);   // D = W . X^T
        if (set != nset - 1) return;
        // ---- epilogue of this 128-column pass: bias, residual, activation in registers; the tile leaves through Ys
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 bv = *(const float4*)(sb + col0);
            float4 rs = make_float4(0.f, 0.f, 0.f, 0.f);
            if (p.residual && row_ok && col0 < p.N) rs = *(const float4*)(p.residual + (size_t)(m0 + ql) * p.N + col0);
            float v[4] = {acc[4 * k] + bv.x + rs.x, acc[4 * k + 1] + bv.y + rs.y, acc[4 * k + 2] + bv.z + rs.z, acc[4 * k + 3] + bv.w + rs.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = apply_act<float>(v[e], p.act);
            *(float4*)(Ys + ql * kYRow + (cbase + 8 * k) * 4) = make_float4(v[0], v[1], v[2], v[3]);
        }
        __syncthreads();
        {
            const int r = tid >> 3, sub = tid & 7;
            if (m0 + r < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = pass * 128 + sub * 16 + j * 4;
                    if (c0 < p.N) *(float4*)(p.out + (size_t)(m0 + r) * p.N + c0) = *(const float4*)(Ys + r * kYRow + sub * 64 + j * 16);
                }
            }
        }
        if (pass + 1 < npn) __syncthreads();              // Ys is rewritten by the next pass
    };
    for (int s = 0; s < nsteps; s += 2) {
        if (s + 1 < nsteps) load_set(fb, s + 1);
        do_step(s, fa);
        if (s + 1 < nsteps) {
            if (s + 2 < nsteps) load_set(fa, s + 2);
            do_step(s + 1, fb);
        }
    }
}

}  // namespace

// the fp32-storage form of cobevt_linear_rows_small_k (gemm_rows3.hip); -1 when the shape does not qualify
int launch_linear_rows_f32(const void* in, const void* wfrag, const float* bias, const void* residual, const float* pre_scale,
                           const float* pre_shift, void* out, const long* dims, float ln_eps, hipStream_t stream) {
    // dims: [dtype(1), M, N, K, lda, pre_relu, act, ln, in_stride, src_H, src_W, in_H, in_W, rows_per_workgroup (ignored)]
    Gr3fParams p;
    p.in = (const float*)in; p.wfrag = (const uint4*)wfrag; p.bias = bias; p.residual = (const float*)residual;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.out = (float*)out;
    p.M = (int)dims[1]; p.N = (int)dims[2]; p.K = (int)dims[3]; p.lda = dims[4];
    p.pre_relu = (int)dims[5]; p.act = (int)dims[6]; p.ln = (int)dims[7]; p.ln_eps = ln_eps;
    p.in_stride = (int)dims[8]; p.src_H = (int)dims[9]; p.src_W = (int)dims[10]; p.in_H = (int)dims[11]; p.in_W = (int)dims[12];
    p.Kp = (p.K + 63) / 64 * 64;
    if (p.M < 1 || p.N < 4 || p.N % 4 || p.N > 4096 || p.K < 4 || p.K > 512 || p.K % 4 || p.lda < p.K || p.lda % 4) return -1;
    if (p.ln && p.K != 128) return -1;                    // the fused LayerNorm is the 128-channel one of the FAX / fusion blocks
    if (p.in_stride < 1 || (p.in_stride > 1 && (p.src_H < 1 || p.src_W < 1 || p.in_H < 1 || p.in_W < 1 || p.M % (p.src_H * p.src_W)))) return -1;
    if (p.in_stride > 1 && residual) return -1;
    if ((pre_scale == nullptr) != (pre_shift == nullptr) || (p.ln && pre_scale)) return -1;
    if (p.act < 0 || p.act > 4) return -1;
    const size_t lds = (size_t)kRows * (p.Kp * 4 + 16) + (size_t)kRows * kYRow + (size_t)((p.N + 127) / 128) * 128 * 4;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)gemm_rows3_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kRows * (512 * 4 + 16) + kRows * kYRow + 4096 * 4);
    const unsigned blocks = (unsigned)((p.M + kRows - 1) / kRows);
    hipLaunchKernelGGL(gemm_rows3_f32_kernel, dim3(blocks), dim3(kThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt
