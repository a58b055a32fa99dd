// Dense-row GEMM for every nn.Linear / 1x1 stride-1 convolution of the FAX hot path (gfx950), with the
// LayerNorm that precedes most of them fused into the A-operand staging.
//
//   out[m][n] = act( f(A[m][:]) . W[n][:] + bias[n] + residual[m][n] ),   f = LayerNorm | BN-ReLU pre-activation | id
//
// Replaces (together with the epilogue fusions) the LayerNorm -> Linear pairs of
// opv2v/opencood/models/sub_modules/fax_modules.py:189-193 (to_q/to_k/to_v), :309-313,411,435 (prenorm + mlp),
// opv2v/opencood/models/fusion_modules/swap_fusion_modules.py:45-53 + base_transformer.py:102-124 (PreNormResidual +
// to_qkv / FeedForward), the pre-activation 1x1 convs fax_modules.py:281-292, Bottleneck 1x1 convs :472.
//
// Tile 128 x 128, K-tile = 256 bytes per row (128 bf16 / 64 fp32), 512 threads = 8 waves (4 x 2, 32 x 64 each,
// 16 MFMAs per wave per K-tile).  Thread t stages 64 contiguous bytes of row t>>2 of the A tile and of the W tile,
// so a 256-byte row is owned by 4 adjacent lanes: LayerNorm statistics are two xor-shuffles in registers before
// the tile is written to LDS (fused only when the whole row fits one K-tile: K <= 128 bf16 / 64 fp32; otherwise the
// caller runs cobevt_layernorm first).  One LDS buffer + register prefetch of the next K-tile; fp32-staged,
// 16-byte coalesced epilogue (bias, residual, ReLU / exact GELU) as in conv3x3.hip.
#include "common.hpp"

namespace cobevt {

struct GemmRowsParams {
    const void* in;
    const void* wgt;        // [N][Kp], Kp = K rounded up to a whole K-tile, zero padded
    const float* bias;
    const void* residual;   // [M][N] or null
    const float* ln_gamma;  // optional LayerNorm affine (normally folded into wgt / bias on the host -> null)
    const float* ln_beta;
    int ln;                 // 1: normalise every A row over K (mean / biased variance, eps inside the sqrt)
    const float* pre_scale; // per-channel affine (+ReLU) on A (null = off)
    const float* pre_shift;
    void* out;
    long lda;               // elements between consecutive A rows
    int M, N, K, Kp;
    float ln_eps;
    int pre_relu;
    int act;
    // optional spatial remap of output rows into a zero-padded (src_n, out_H, out_W) map
    int src_H, src_W, out_H, out_W;
    // optional strided row gather (1x1 convolution with stride s): output row (n, oy, ox) of the (src_H, src_W) map reads
    // input pixel (n, s*oy, s*ox) of an (in_H, in_W) map
    int in_stride, in_H, in_W;
    // optional A-row producer: BEV query embedding (fax_modules.py:370-375,387-388) computed on the fly instead of
    // being read:  A[(bn, pix)] = x[bn / n][pix] + L2norm_c( w_bev . world[pix] + b_bev - w_cam . E_inv[bn][:, 3] )
    const float* emb_world;  // (2, hw) or null = rows are read from `in`
    const float* emb_wbev;   // (K, 2)
    const float* emb_bbev;   // (K)
    const float* emb_wcam;   // (K, 4)
    const float* emb_E;      // (B*n, 4, 4)
    int emb_n, emb_hw;
};

constexpr int kGrThreads = 512;
constexpr int kGrRow = 256 + 16;          // LDS row stride (bytes)
constexpr int kGrTile = 128;              // rows of A and of W per tile
constexpr int kGrStageRow = 128 * 4 + 16; // fp32 staging row stride
constexpr int kGrLds = 2 * kGrTile * kGrRow + 128 * 16;   // 69,632 B of tiles + the embedding producer's coefficient table

#ifdef COBEVT_GEMM_TRACE      // tools/gemm_trace.py builds a copy of this file with s_memtime marks (never the product .so)
__device__ unsigned long long cobevt_gemm_trace[16];
#define COBEVT_GT_MARK(i) do { if (blockIdx.x == 0 && threadIdx.x == 0) cobevt_gemm_trace[(i)] = __builtin_readcyclecounter(); } while (0)
#else
#define COBEVT_GT_MARK(i) do {} while (0)
#endif

template <typename T, bool EMB>
__global__ __launch_bounds__(kGrThreads, 4) void gemm_rows_kernel(GemmRowsParams p) {
    constexpr int CH = Elem<T>::kChunk;
    constexpr int TK = 256 / Elem<T>::kBytes;        // elements per K-tile
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Ws = smem + kGrTile * kGrRow;

    const int ntn = (p.N + 127) / 128;
    int logical;
    {
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m0 = (logical / ntn) * 128, n0 = (logical % ntn) * 128;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    float ebias[2];                                  // the epilogue's two bias values per lane, fetched up front (the
#pragma unroll                                       // epilogue would otherwise start with a global round trip)
    for (int b = 0; b < 2; ++b) {
        const int c = n0 + wn * 64 + b * 32 + ql;
        ebias[b] = p.bias ? p.bias[c < p.N ? c : 0] : 0.f;
        if (c >= p.N) ebias[b] = 0.f;
    }
    const int srow = tid >> 2, sub = tid & 3;        // staging: row of the tile, 64-byte quarter of the 256-byte row
    const T* in = (const T*)p.in;
    const T* wg = (const T*)p.wgt;
    const bool a_ok = m0 + srow < p.M;
    const bool w_ok = n0 + srow < p.N;
    size_t arow_idx = a_ok ? m0 + srow : 0;
    if (p.in_stride > 1) {
        const int hw = p.src_H * p.src_W;
        const int n = (int)(arow_idx / hw), rem = (int)(arow_idx - (size_t)n * hw);
        const int oy = rem / p.src_W, ox = rem - oy * p.src_W;
        arow_idx = ((size_t)n * p.in_H + (size_t)oy * p.in_stride) * p.in_W + (size_t)ox * p.in_stride;
    }
    const T* arow = in + arow_idx * p.lda;
    const T* wrow = wg + (size_t)(w_ok ? n0 + srow : 0) * p.Kp;
    const int nkt = p.Kp / TK;

    uint4 areg[4], wreg[4];
    // embedding producer: per-channel (w_bev0, w_bev1, b_bev - w_cam . c) of this tile's camera in LDS (one camera per
    // tile: hw % 128 == 0, checked by the entry point)
    float4* coef = (float4*)(smem + 2 * kGrTile * kGrRow);
    if (EMB) {
        const int bn = m0 / p.emb_hw;
        if (tid < TK) {
            float4 c = make_float4(0.f, 0.f, 0.f, 0.f);
            if (tid < p.K) {
                const float* E = p.emb_E + (size_t)bn * 16;
                const float4 wc = *(const float4*)(p.emb_wcam + tid * 4);
                c.x = p.emb_wbev[tid * 2];
                c.y = p.emb_wbev[tid * 2 + 1];
                c.z = p.emb_bbev[tid] - (wc.x * E[3] + wc.y * E[7] + wc.z * E[11] + wc.w * E[15]);
            }
            coef[tid] = c;
        }
        __syncthreads();
    }
    auto load_tile = [&](int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kt * TK + (sub * 4 + j) * CH;
            if (!EMB) {
                areg[j] = (a_ok && k < p.K) ? *(const uint4*)(arow + k) : make_uint4(0, 0, 0, 0);
                wreg[j] = w_ok ? *(const uint4*)(wrow + k) : make_uint4(0, 0, 0, 0);
            }
        }
        if (EMB) {                                       // single K-tile (K <= TK): the whole row is here
            const int m = a_ok ? m0 + srow : m0;
            const int bn = m / p.emb_hw, pix = m - bn * p.emb_hw;
            const float wx = p.emb_world[pix], wy = p.emb_world[p.emb_hw + pix];
            const T* xrow = (const T*)p.in + ((size_t)(bn / p.emb_n) * p.emb_hw + pix) * p.K;
            // two passes over the embedding (sum of squares, then scale + add) instead of holding all 32 values:
            // keeps the kernel at 128 VGPRs = two workgroups per CU like the plain GEMM
            float ss = 0.f;
#pragma unroll 1
            for (int j = 0; j < 4; ++j)                   // rolled: the compiler would hoist all 32 coefficient reads
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = (sub * 4 + j) * CH + e;
                    const float4 c = coef[k < TK ? k : 0];
                    const float val = k < p.K ? (c.x * wx + c.y * wy + c.z) : 0.f;
                    ss += val * val;
                }
            ss += __shfl_xor(ss, 1, 64);
            ss += __shfl_xor(ss, 2, 64);
            const float inv = 1.0f / (sqrtf(ss) + 1e-7f);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = (sub * 4 + j) * CH;
                float xv[8], v[8];
                chunk_to_f32<T>(k0 < p.K ? *(const uint4*)(xrow + k0) : make_uint4(0, 0, 0, 0), xv);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = k0 + e;
                    const float4 c = coef[k < TK ? k : 0];
                    v[e] = (k < p.K ? (c.x * wx + c.y * wy + c.z) : 0.f) * inv + xv[e];
                }
                areg[j] = a_ok ? f32_to_chunk<T>(v) : make_uint4(0, 0, 0, 0);   // rounded exactly as the stored query
            }
            __builtin_amdgcn_sched_barrier(0);           // the weight tile is requested after the register-hungry part
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k = kt * TK + (sub * 4 + j) * CH;
                wreg[j] = w_ok ? *(const uint4*)(wrow + k) : make_uint4(0, 0, 0, 0);
            }
        }
    };
    // LayerNorm / pre-activation on this thread's 64 bytes (and its 3 neighbours' for the row statistics)
    auto transform_a = [&](int kt) {
        if (p.ln) {
            float v[4][8];
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                chunk_to_f32<T>(areg[j], v[j]);
#pragma unroll
                for (int e = 0; e < CH; ++e) s += v[j][e];      // columns >= K are zero
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float mean = s / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = (sub * 4 + j) * CH + e;
                    const float d = k < p.K ? v[j][e] - mean : 0.f;
                    q += d * d;
                }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            const float rstd = rsqrtf(q / (float)p.K + p.ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = (sub * 4 + j) * CH + e;
                    float y = (v[j][e] - mean) * rstd;
                    if (p.ln_gamma) y = y * p.ln_gamma[k] + p.ln_beta[k];
                    v[j][e] = k < p.K ? y : 0.f;
                }
                areg[j] = f32_to_chunk<T>(v[j]);
            }
        } else if (p.pre_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = kt * TK + (sub * 4 + j) * CH;     // K % CH == 0: a chunk is entirely inside or outside K
                if (k0 >= p.K) { areg[j] = make_uint4(0, 0, 0, 0); continue; }
                float v[8], sc[8], sh[8];
                chunk_to_f32<T>(areg[j], v);
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) {               // 16-byte loads of the per-channel affine
                    const float4 a = *(const float4*)(p.pre_scale + k0 + 4 * q), c = *(const float4*)(p.pre_shift + k0 + 4 * q);
                    sc[4 * q] = a.x; sc[4 * q + 1] = a.y; sc[4 * q + 2] = a.z; sc[4 * q + 3] = a.w;
                    sh[4 * q] = c.x; sh[4 * q + 1] = c.y; sh[4 * q + 2] = c.z; sh[4 * q + 3] = c.w;
                }
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const float x = v[e] * sc[e] + sh[e];
                    v[e] = p.pre_relu ? fmaxf(x, 0.f) : x;
                }
                areg[j] = f32_to_chunk<T>(v);
            }
        }
    };
    auto store_tile = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            *(uint4*)(As + srow * kGrRow + (sub * 4 + j) * 16) = stage_x_piece<T>(areg[j]);
            *(uint4*)(Ws + srow * kGrRow + (sub * 4 + j) * 16) = stage_ws_piece<T>(wreg[j]);
        }
    };

    f32x16 acc[2];
#pragma unroll
    for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[b][r] = 0.f;
    const int abase = (wm * 32 + ql) * kGrRow + h * 16;
    const int bbase = (wn * 64 + ql) * kGrRow + h * 16;

    COBEVT_GT_MARK(0);
    load_tile(0);
    for (int kt = 0; kt < nkt; ++kt) {
        transform_a(kt);
        COBEVT_GT_MARK(1);
        if (kt > 0) __syncthreads();          // previous tile fully consumed
        store_tile();
        __syncthreads();
        COBEVT_GT_MARK(2);
        if (kt + 1 < nkt) load_tile(kt + 1);
        const int kleft = p.K - kt * TK;
        const int ng = kleft >= TK ? 8 : (kleft * Elem<T>::kBytes + 31) / 32;
        for (int g = 0; g < ng; ++g) {
            const uint4 af = *(const uint4*)(As + abase + g * 32);
            const uint4 b0 = *(const uint4*)(Ws + bbase + g * 32);
            const uint4 b1 = *(const uint4*)(Ws + bbase + 32 * kGrRow + g * 32);
            mfma_kgroup_ss<T>(af, b0, acc[0]);    // A = activation rows, B = weights (both staged: common.hpp)
            mfma_kgroup_ss<T>(af, b1, acc[1]);
        }
    }
    COBEVT_GT_MARK(3);
    __syncthreads();
    COBEVT_GT_MARK(4);

    // ---- epilogue: fp32 staging [128][128] then coalesced 16-byte passes
    float* stage = (float*)smem;
    constexpr int SROW = kGrStageRow / 4;
#pragma unroll
    for (int b = 0; b < 2; ++b) {
        const int cl = wn * 64 + b * 32 + ql;
        const float bias = ebias[b];
#pragma unroll
        for (int r = 0; r < 16; ++r) stage[(wm * 32 + acc_row(r, lane)) * SROW + cl] = acc[b][r] + bias;
    }
    __syncthreads();
    COBEVT_GT_MARK(5);
    T* out = (T*)p.out;
    constexpr int CPR = 128 / CH;                 // 16-byte output chunks per tile row
    const bool remap = p.out_H != p.src_H || p.out_W != p.src_W;
    const bool vec_ok = (p.N % CH) == 0;
    if (vec_ok) {
        // unrolled: every residual load and staging read of the thread is in flight before the first store (the rolled loop
        // serialised one LDS + global round trip per item: 5.3k of a workgroup's 19k cycles in the s_memtime trace)
        constexpr int NIT = 128 * CPR / kGrThreads;
        long orow[NIT];
        uint4 rres[NIT];
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = tid + i * kGrThreads;
            const int row = item / CPR, cj = item - row * CPR;
            const int m = m0 + row, col = n0 + cj * CH;
            const bool ok = (m < p.M) & (col < p.N);
            orow[i] = ok ? m : -1;
            rres[i] = make_uint4(0, 0, 0, 0);
            if (remap && ok) {
                const int hw = p.src_H * p.src_W;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / p.src_W, ow = rem - oh * p.src_W;
                orow[i] = ((long)n * p.out_H + oh) * p.out_W + ow;
            }
        }
        if (p.residual) {                     // unconditional, clamped: all of the thread's residual loads in flight at once
#pragma unroll
            for (int i = 0; i < NIT; ++i) {
                const int item = tid + i * kGrThreads;
                const int row = item / CPR, cj = item - row * CPR;
                const int m = m0 + row, col = n0 + cj * CH;
                const bool ok = (m < p.M) & (col < p.N);
                rres[i] = *(const uint4*)((const T*)p.residual + (ok ? (size_t)m * p.N + col : 0));
            }
        }
        float sv[NIT][CH];                    // and all staging reads
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = tid + i * kGrThreads;
            const int row = item / CPR, cj = item - row * CPR;
#pragma unroll
            for (int e = 0; e < CH; ++e) sv[i][e] = stage[row * SROW + cj * CH + e];
        }
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            if (orow[i] < 0) continue;
            const int cj = (tid + i * kGrThreads) % CPR;
            float v[8], rv[8];
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = sv[i][e];
            if (p.residual) {
                chunk_to_f32<T>(rres[i], rv);
#pragma unroll
                for (int e = 0; e < CH; ++e) v[e] += rv[e];
            }
#pragma unroll
            for (int e = 0; e < CH; ++e) v[e] = p.act == 1 ? fmaxf(v[e], 0.f) : (p.act == 2 ? gelu_t<T>(v[e]) : (p.act >= 3 ? apply_act<T>(v[e], p.act) : v[e]));
            *(uint4*)(out + (size_t)orow[i] * p.N + n0 + cj * CH) = f32_to_chunk<T>(v);
        }
    } else {                                          // ragged N (not a multiple of the 16-byte chunk): scalar
        for (int item = tid; item < 128 * CPR; item += kGrThreads) {
            const int row = item / CPR, cj = item - row * CPR;
            const int m = m0 + row, col = n0 + cj * CH;
            if (m >= p.M || col >= p.N) continue;
            size_t orow = (size_t)m;
            if (remap) {
                const int hw = p.src_H * p.src_W;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / p.src_W, ow = rem - oh * p.src_W;
                orow = ((size_t)n * p.out_H + oh) * p.out_W + ow;
            }
            for (int e = 0; e < CH && col + e < p.N; ++e) {
                float x = stage[row * SROW + cj * CH + e];
                if (p.residual) x += load_elem<T>((const T*)p.residual, (size_t)m * p.N + col + e);
                x = p.act == 1 ? fmaxf(x, 0.f) : (p.act == 2 ? gelu_t<T>(x) : (p.act >= 3 ? apply_act<T>(x, p.act) : x));
                store_elem<T>(out, orow * p.N + col + e, x);
            }
        }
    }
    COBEVT_GT_MARK(6);
}


// ---------------------------------------------------------------------------------------------------------------------
// Version 2 (cobevt_linear_rows_wfrag): persistent workgroups, fragment-ordered weights.
//
// Every Linear / 1x1 conv of the hot path is HBM-bound (K <= 512, arithmetic intensity 64-128 FLOP/B) and the kernel above
// reached 1.6-2.7 TB/s of the ~6.3 achievable: a workgroup requests its 128 x K tile, waits, computes, stages, stores
// - the memory pipe only carries requests during the first of those phases - and re-reads its 32-KB weight tile from L2
// into LDS every time.  Here
//   * a workgroup is persistent over row tiles of ONE 128-column tile: with K <= one K-tile its weight fragments
//     (MFMA fragment order, one coalesced 1-KB wave load per k-group, as in row_chain.hip) are loaded once and stay in
//     registers; no weight bytes pass through LDS at all;
//   * the A rows of the NEXT tile are requested right after this tile's rows are handed to LDS, so loads are in flight
//     under the MFMAs, the epilogue and the stores of the current tile;
//   * D = W . X^T: a lane owns one row and runs of four output columns -> bias / residual / activation in registers, the
//     result is staged in the STORAGE type (8-byte LDS writes, half the staging bytes) for 16-byte coalesced stores.
// Same options as above (fused LayerNorm with folded affine, pre-activation affine + ReLU, residual, padded-map row
// remap, strided row gather).  8 waves = 2 row halves (two 32-row MFMA tiles each) x 4 column tiles of 32.
// ---------------------------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kGrThreads, 2) void gemm_rows2_kernel(GemmRowsParams p) {
    constexpr int CH = Elem<T>::kChunk;
    constexpr int EB = Elem<T>::kBytes;
    constexpr int TK = 256 / EB;                      // elements per K-tile
    constexpr int CROW = 128 * EB + 16;               // staging row (storage type)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem;
    unsigned char* Cs = smem + kGrTile * kGrRow;

    const int ntn = (p.N + 127) / 128, ntm = (p.M + 127) / 128;
    const int tn = blockIdx.x % ntn, tm_first = blockIdx.x / ntn, tm_step = gridDim.x / ntn;
    const int n0 = tn * 128;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;
    const int srow = tid >> 2, sub = tid & 3;         // staging: row of the tile, 64-byte quarter of the 256-byte row
    const T* in = (const T*)p.in;
    const int nkt = p.Kp / TK;
    const int nkg = p.Kp * EB / 32;                   // k-groups per weight row
    const uint4* wq = (const uint4*)p.wgt + (size_t)(n0 / 32 + wn) * nkg * 64 + lane;

    uint4 bfrag[8];
    auto load_b = [&](int kt) {
        // two base addresses + immediate offsets (a global_load immediate reaches +-4 KB): eight 64-bit address pairs
        // would cost 16 VGPRs in a kernel that has to fit 128
        const uint4* b0 = wq + (size_t)(kt * 8) * 64;
        const uint4* b1 = b0 + 4 * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) bfrag[g] = b0[g * 64];
#pragma unroll
        for (int g = 0; g < 4; ++g) bfrag[4 + g] = b1[g * 64];
    };
    uint4 areg[4];
    auto row_ptr = [&](int tm) {                      // this thread's A row of tile tm (row 0 when past M)
        size_t idx = (size_t)tm * 128 + srow;
        if (idx >= (size_t)p.M) idx = 0;
        if (p.in_stride > 1) {
            const int hw = p.src_H * p.src_W;
            const int n = (int)(idx / hw), rem = (int)(idx - (size_t)n * hw);
            const int oy = rem / p.src_W, ox = rem - oy * p.src_W;
            idx = ((size_t)n * p.in_H + (size_t)oy * p.in_stride) * p.in_W + (size_t)ox * p.in_stride;
        }
        return in + idx * p.lda;
    };
    auto load_a = [&](const T* arow, bool ok, int kt) {
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int k = kt * TK + (sub * 4 + j) * CH;
            areg[j] = (ok && k < p.K) ? *(const uint4*)(arow + k) : make_uint4(0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);            // keep the prefetch here (LLVM sinks loads to their first use)
    };
    auto transform_a = [&](int kt) {
        if (p.ln) {
            // three passes that unpack one 16-byte chunk at a time (8 live floats instead of 32): this kernel keeps its weight
            // fragments, the accumulators and the next tile's rows in registers and has to fit 128 VGPRs
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
                chunk_to_f32<T>(areg[j], v);
#pragma unroll
                for (int e = 0; e < CH; ++e) s += v[e];
            }
            s += __shfl_xor(s, 1, 64);
            s += __shfl_xor(s, 2, 64);
            const float mean = s / (float)p.K;
            float q = 0.f;
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
                chunk_to_f32<T>(areg[j], v);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = (sub * 4 + j) * CH + e;
                    const float d = k < p.K ? v[e] - mean : 0.f;
                    q += d * d;
                }
            }
            q += __shfl_xor(q, 1, 64);
            q += __shfl_xor(q, 2, 64);
            const float rstd = rsqrtf(q / (float)p.K + p.ln_eps);
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float v[8];
                chunk_to_f32<T>(areg[j], v);
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const int k = (sub * 4 + j) * CH + e;
                    v[e] = k < p.K ? (v[e] - mean) * rstd : 0.f;
                }
                areg[j] = f32_to_chunk<T>(v);
            }
        } else if (p.pre_scale) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int k0 = kt * TK + (sub * 4 + j) * CH;
                if (k0 >= p.K) { areg[j] = make_uint4(0, 0, 0, 0); continue; }
                float v[8], sc[8], sh[8];
                chunk_to_f32<T>(areg[j], v);
#pragma unroll
                for (int q = 0; q < CH / 4; ++q) {
                    const float4 a = *(const float4*)(p.pre_scale + k0 + 4 * q), c = *(const float4*)(p.pre_shift + k0 + 4 * q);
                    sc[4 * q] = a.x; sc[4 * q + 1] = a.y; sc[4 * q + 2] = a.z; sc[4 * q + 3] = a.w;
                    sh[4 * q] = c.x; sh[4 * q + 1] = c.y; sh[4 * q + 2] = c.z; sh[4 * q + 3] = c.w;
                }
#pragma unroll
                for (int e = 0; e < CH; ++e) {
                    const float x = v[e] * sc[e] + sh[e];
                    v[e] = p.pre_relu ? fmaxf(x, 0.f) : x;
                }
                areg[j] = f32_to_chunk<T>(v);
            }
        }
    };

    if (tm_first >= ntm) return;
    const bool resident_b = nkt == 1;
    if (resident_b) load_b(0);
    const T* arow = row_ptr(tm_first);
    bool a_ok = (size_t)tm_first * 128 + srow < (size_t)p.M;
    load_a(arow, a_ok, 0);
    const int c0w = wn * 32 + 4 * h;                  // this lane's first column of the 128-column tile
    const bool remap = p.out_H != p.src_H || p.out_W != p.src_W;
    T* out = (T*)p.out;

    for (int tm = tm_first; tm < ntm; tm += tm_step) {
        const int m0 = tm * 128;
        f32x16 acc[2];
#pragma unroll
        for (int rt = 0; rt < 2; ++rt)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[rt][r] = 0.f;
        for (int kt = 0; kt < nkt; ++kt) {
            transform_a(kt);
            if (kt > 0) __syncthreads();              // the previous K-tile is consumed
#pragma unroll
            for (int j = 0; j < 4; ++j) *(uint4*)(As + srow * kGrRow + (sub * 4 + j) * 16) = stage_x_piece<T>(areg[j]);
            if (!resident_b) load_b(kt);
            __syncthreads();
            // next A rows: the following K-tile of this tile, or the first K-tile of the workgroup's next row tile
            if (kt + 1 < nkt) load_a(arow, a_ok, kt + 1);
            else if (tm + tm_step < ntm) {
                arow = row_ptr(tm + tm_step);
                a_ok = (size_t)(tm + tm_step) * 128 + srow < (size_t)p.M;
                load_a(arow, a_ok, 0);
            }
            const int kleft = p.K - kt * TK;
            const int ng = kleft >= TK ? 8 : (kleft * EB + 31) / 32;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 8; ++g)
                if (g < ng) {
#pragma unroll
                    for (int rt = 0; rt < 2; ++rt) {
                        const uint4 af = *(const uint4*)(As + (wm * 64 + rt * 32 + ql) * kGrRow + h * 16 + g * 32);
                        mfma_kgroup_xs<T>(bfrag[g], af, acc[rt]);     // D = W . X^T : lane <-> row, registers <-> columns (A rows staged: common.hpp)
                    }
                    if (Elem<T>::kIsBf16) {                        // pin "read, MFMA": unpinned, LLVM hoists all 16 fragment
#pragma unroll                                                      // reads (64 VGPRs) above the MFMAs
                        for (int rt = 0; rt < 2; ++rt) {
                            __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        }
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
        }
        // ---- epilogue in registers, staged in the storage type
#pragma unroll
        for (int rt = 0; rt < 2; ++rt) {
            const int row = wm * 64 + rt * 32 + ql;
            const int m = m0 + row;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int col = n0 + c0w + 8 * k;
                float v[4] = {acc[rt][4 * k], acc[rt][4 * k + 1], acc[rt][4 * k + 2], acc[rt][4 * k + 3]};
                if (col < p.N) {
                    if (p.bias) {
                        const float4 b = *(const float4*)(p.bias + col);
                        v[0] += b.x; v[1] += b.y; v[2] += b.z; v[3] += b.w;
                    }
                    if (p.residual && m < p.M) {
                        const T* r = (const T*)p.residual + (size_t)m * p.N + col;
                        if constexpr (Elem<T>::kIsBf16) {
                            const uint2 u = *(const uint2*)r;
                            v[0] += bf2f(u.x & 0xffff); v[1] += bf2f(u.x >> 16); v[2] += bf2f(u.y & 0xffff); v[3] += bf2f(u.y >> 16);
                        } else {
                            const float4 u = *(const float4*)r;
                            v[0] += u.x; v[1] += u.y; v[2] += u.z; v[3] += u.w;
                        }
                    }
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = p.act == 1 ? fmaxf(v[e], 0.f) : (p.act == 2 ? gelu_t<T>(v[e]) : (p.act >= 3 ? apply_act<T>(v[e], p.act) : v[e]));
                }
                unsigned char* d = Cs + row * CROW + (c0w + 8 * k) * EB;
                if constexpr (Elem<T>::kIsBf16) *(uint2*)d = make_uint2(pack_bf2(v[0], v[1]), pack_bf2(v[2], v[3]));
                else *(float4*)d = make_float4(v[0], v[1], v[2], v[3]);
            }
        }
        __syncthreads();
        constexpr int CPR = 128 / CH;                 // 16-byte chunks per tile row
        constexpr int NIT = 128 * CPR / kGrThreads;
#pragma unroll
        for (int i = 0; i < NIT; ++i) {
            const int item = tid + i * kGrThreads;
            const int row = item / CPR, cj = item - row * CPR;
            const int m = m0 + row, col = n0 + cj * CH;
            if (m >= p.M || col >= p.N) continue;
            size_t orow = (size_t)m;
            if (remap) {
                const int hw = p.src_H * p.src_W;
                const int n = m / hw, rem = m - n * hw;
                const int oh = rem / p.src_W, ow = rem - oh * p.src_W;
                orow = ((size_t)n * p.out_H + oh) * p.out_W + ow;
            }
            *(uint4*)(out + orow * p.N + col) = *(const uint4*)(Cs + row * CROW + cj * 16);
            __builtin_amdgcn_sched_barrier(0);        // one item at a time: keeps four 64-bit store addresses from living at once
        }
        // the next iteration's Cs writes come after its own barrier; its As writes after this tile's MFMA reads (barrier above)
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_linear_rows(const void* in, const void* wgt, const float* bias, const void* residual,
                                  const float* ln_gamma, const float* ln_beta, const float* pre_scale,
                                  const float* pre_shift, void* out, const long* dims, float ln_eps, hipStream_t stream) {
    // dims: [dtype, M, N, K, Kp, lda, pre_relu, act, src_H, src_W, out_H, out_W, ln, in_stride, in_H, in_W]
    if (!in || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    GemmRowsParams p;
    const int dtype = (int)dims[0];
    p.in = in; p.wgt = wgt; p.bias = bias; p.residual = residual;
    p.ln_gamma = ln_gamma; p.ln_beta = ln_beta; p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.out = out;
    p.M = (int)dims[1]; p.N = (int)dims[2]; p.K = (int)dims[3]; p.Kp = (int)dims[4]; p.lda = dims[5];
    p.pre_relu = (int)dims[6]; p.act = (int)dims[7];
    p.src_H = (int)dims[8]; p.src_W = (int)dims[9]; p.out_H = (int)dims[10]; p.out_W = (int)dims[11];
    p.ln_eps = ln_eps;
    p.ln = (int)dims[12] || ln_gamma != nullptr;
    p.in_stride = (int)dims[13]; p.in_H = (int)dims[14]; p.in_W = (int)dims[15];
    p.emb_world = p.emb_wbev = p.emb_bbev = p.emb_wcam = p.emb_E = nullptr;
    p.emb_n = p.emb_hw = 1;
    if (p.in_stride < 1) return COBEVT_ERR_ARG;
    if (p.in_stride > 1 && ((p.src_H - 1) * p.in_stride >= p.in_H || (p.src_W - 1) * p.in_stride >= p.in_W ||
                            p.M % ((long)p.src_H * p.src_W) != 0)) return COBEVT_ERR_SHAPE;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    const int tk = dtype == 0 ? 128 : 64, ch = dtype == 0 ? 8 : 4;
    if (p.M < 1 || p.N < 1 || p.K < 1 || p.Kp % tk != 0 || p.Kp < p.K) return COBEVT_ERR_SHAPE;
    if (p.K % ch != 0 || p.lda % ch != 0 || p.lda < p.K) return COBEVT_ERR_SHAPE;
    if ((ln_gamma == nullptr) != (ln_beta == nullptr)) return COBEVT_ERR_ARG;
    if (p.ln && p.K > tk) return COBEVT_ERR_UNSUPPORTED;           // LayerNorm fusion needs the row in one K-tile
    if (p.ln && pre_scale) return COBEVT_ERR_UNSUPPORTED;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return COBEVT_ERR_ARG;
    if (p.residual && (p.out_H != p.src_H || p.out_W != p.src_W)) return COBEVT_ERR_UNSUPPORTED;
    const long blocks = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    constexpr size_t lds = kGrLds;                 // A + W tiles (the fp32 staging (128 x 528) fits inside) + coefficients
    static_assert(128 * kGrStageRow <= 2 * kGrTile * kGrRow, "staging must fit");
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<bf16_t, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<float, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    }
    if (dtype == 0) hipLaunchKernelGGL((gemm_rows_kernel<bf16_t, false>), dim3((unsigned)blocks), dim3(kGrThreads), lds, stream, p);
    else hipLaunchKernelGGL((gemm_rows_kernel<float, false>), dim3((unsigned)blocks), dim3(kGrThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_bev_embed_linear_rows(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                                            const float* w_cam, const void* x, const void* wgt, const float* bias, void* out,
                                            const long* dims, float ln_eps, hipStream_t stream) {
    // dims: [dtype, B, n, hw, D (= K), N, Kp, ln]
    if (!E_inv || !world || !w_bev || !b_bev || !w_cam || !x || !wgt || !out || !dims) return COBEVT_ERR_ARG;
    GemmRowsParams p;
    const int dtype = (int)dims[0];
    const long B = dims[1], n = dims[2], hw = dims[3];
    p.in = x; p.wgt = wgt; p.bias = bias; p.residual = nullptr;
    p.ln_gamma = p.ln_beta = p.pre_scale = p.pre_shift = nullptr; p.out = out;
    p.K = (int)dims[4]; p.N = (int)dims[5]; p.Kp = (int)dims[6]; p.ln = (int)dims[7];
    p.lda = p.K; p.pre_relu = 0; p.act = 0;
    p.ln_eps = ln_eps;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    const int tk = dtype == 0 ? 128 : 64, ch = dtype == 0 ? 8 : 4;
    if (B < 1 || n < 1 || hw < 1 || hw % 128 != 0 || B * n * hw > 0x7fffffffL) return COBEVT_ERR_SHAPE;   // one camera per tile
    if (p.N < 1 || p.K < 1 || p.K > tk || p.K % ch != 0 || p.Kp != tk) return COBEVT_ERR_SHAPE;          // whole row in one K-tile
    p.M = (int)(B * n * hw);
    p.src_H = p.out_H = 1; p.src_W = p.out_W = p.M;
    p.in_stride = 1; p.in_H = p.in_W = 1;
    p.emb_world = world; p.emb_wbev = w_bev; p.emb_bbev = b_bev; p.emb_wcam = w_cam; p.emb_E = E_inv;
    p.emb_n = (int)n; p.emb_hw = (int)hw;
    const long blocks = (long)((p.M + 127) / 128) * ((p.N + 127) / 128);
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<bf16_t, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrLds);
        (void)hipFuncSetAttribute((const void*)gemm_rows_kernel<float, true>, hipFuncAttributeMaxDynamicSharedMemorySize, kGrLds);
    }
    if (dtype == 0) hipLaunchKernelGGL((gemm_rows_kernel<bf16_t, true>), dim3((unsigned)blocks), dim3(kGrThreads), kGrLds, stream, p);
    else hipLaunchKernelGGL((gemm_rows_kernel<float, true>), dim3((unsigned)blocks), dim3(kGrThreads), kGrLds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_linear_rows_wfrag(const void* in, const void* wfrag, const float* bias, const void* residual,
                                        const float* pre_scale, const float* pre_shift, void* out, const long* dims,
                                        float ln_eps, hipStream_t stream) {
    // dims: as cobevt_linear_rows [dtype, M, N, K, Kp, lda, pre_relu, act, src_H, src_W, out_H, out_W, ln, in_stride, in_H, in_W]
    if (!in || !wfrag || !out || !dims) return COBEVT_ERR_ARG;
    GemmRowsParams p;
    const int dtype = (int)dims[0];
    p.in = in; p.wgt = wfrag; p.bias = bias; p.residual = residual;
    p.ln_gamma = p.ln_beta = nullptr; p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.out = out;
    p.M = (int)dims[1]; p.N = (int)dims[2]; p.K = (int)dims[3]; p.Kp = (int)dims[4]; p.lda = dims[5];
    p.pre_relu = (int)dims[6]; p.act = (int)dims[7];
    p.src_H = (int)dims[8]; p.src_W = (int)dims[9]; p.out_H = (int)dims[10]; p.out_W = (int)dims[11];
    p.ln_eps = ln_eps;
    p.ln = (int)dims[12];
    p.in_stride = (int)dims[13]; p.in_H = (int)dims[14]; p.in_W = (int)dims[15];
    p.emb_world = p.emb_wbev = p.emb_bbev = p.emb_wcam = p.emb_E = nullptr;
    p.emb_n = p.emb_hw = 1;
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    const int tk = dtype == 0 ? 128 : 64, ch = dtype == 0 ? 8 : 4;
    if (p.M < 1 || p.N < 1 || p.K < 1 || p.Kp % tk != 0 || p.Kp < p.K) return COBEVT_ERR_SHAPE;
    if (p.K % ch != 0 || p.N % ch != 0 || p.lda % ch != 0 || p.lda < p.K) return COBEVT_ERR_SHAPE;
    if (p.ln && p.K > tk) return COBEVT_ERR_UNSUPPORTED;
    if (p.ln && pre_scale) return COBEVT_ERR_UNSUPPORTED;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return COBEVT_ERR_ARG;
    if (p.residual && (p.out_H != p.src_H || p.out_W != p.src_W)) return COBEVT_ERR_UNSUPPORTED;
    if (p.in_stride < 1) return COBEVT_ERR_ARG;
    if (p.in_stride > 1 && ((p.src_H - 1) * p.in_stride >= p.in_H || (p.src_W - 1) * p.in_stride >= p.in_W ||
                            p.M % ((long)p.src_H * p.src_W) != 0)) return COBEVT_ERR_SHAPE;
    const long ntn = (p.N + 127) / 128, ntm = (p.M + 127) / 128;
    long per_col = 256 / ntn;                                   // one 8-wave workgroup per CU (the kernel wants ~190 VGPRs; capped at 128 it spills 50 and is slower still), persistent over row tiles
    if (per_col < 1) per_col = 1;
    if (per_col > ntm) per_col = ntm;
    const long blocks = ntn * per_col;
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    const size_t lds = (size_t)kGrTile * kGrRow + (size_t)128 * (128 * (dtype == 0 ? 2 : 4) + 16);
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)gemm_rows2_kernel<bf16_t>, hipFuncAttributeMaxDynamicSharedMemorySize, 34816 + 34816);
        (void)hipFuncSetAttribute((const void*)gemm_rows2_kernel<float>, hipFuncAttributeMaxDynamicSharedMemorySize, 34816 + 67584);
    }
    if (dtype == 0) hipLaunchKernelGGL(gemm_rows2_kernel<bf16_t>, dim3((unsigned)blocks), dim3(kGrThreads), lds, stream, p);
    else hipLaunchKernelGGL(gemm_rows2_kernel<float>, dim3((unsigned)blocks), dim3(kGrThreads), lds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

#ifdef COBEVT_GEMM_TRACE
extern "C" int cobevt_gemm_read_trace(unsigned long long* dst) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt_gemm_trace), sizeof(unsigned long long) * 16) == hipSuccess ? 0 : 1;
}
#endif
