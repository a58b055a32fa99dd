// Fused row-local chain that follows every attention of the FAX hot path (gfx950, bf16 mode):
//
//     y = a . Wp^T (+ bp) + skip                       out-projection + skip connection
//     z = y + ( GELU( LN(y) . W1'^T + b1' ) . W2^T + b2 )     pre-norm MLP with residual (LN affine folded into W1', b1')
//     out = post-LayerNorm(z)  (optional)
//     next = act( normalise(out) . Wn'^T + bn' )     (optional) the LayerNorm + Linear / 1x1 conv that consumes `out` next
//
// i.e. fax_modules.py:240,246-247 (proj + skip) followed by :411 / :435-437 (mlp_1 / mlp_2 + postnorm), and
// swap_fusion_modules.py:126,177 (to_out + PreNormResidual residual) followed by base_transformer.py:102-124
// (PreNormResidual(FeedForward)).  Rows are independent, so a workgroup carries a 64-row tile through all three GEMMs
// with y, LN(y) and the 2C-wide hidden activations resident in LDS; only `a`, `skip` and `out` touch HBM
// (the unfused path writes and re-reads y and the hidden tensor and costs three launches).  The optional `next` phase
// is the row-local GEMM that reads `out` in the reference graph - the to_qkv of the following swap-fusion attention
// (swap_fusion_modules.py:93 behind PreNormResidual.norm), the to_q of the second cross attention (fax_modules.py:201,
// 420-428) or the first 1x1 conv + BN + ReLU of the ResNetBottleNeck that follows (fax_modules.py:472) - done while the
// rows are still in LDS: in the latency-bound tail of the frame every separate launch costs 7-12 us for ~2 us of work.
//
// 512 threads = 8 waves arranged 2 (rows) x 4 (columns): every wave owns one 32x32 MFMA tile of a 64 x 128 output
// panel.  Weight panels ([128 rows][256 bytes], gemm_rows layout) stream through one LDS buffer with register
// prefetch of the next panel.  C <= 128 channels, hidden <= 256.
#include "common.hpp"

namespace cobevt {

struct RowChainParams {
    const bf16_t* a;        // [M][C] attention output
    const bf16_t* skip;     // [M][C] or null
    bf16_t* out;            // [M][C]
    const bf16_t* wp;       // [C][128]   out-projection
    const float* bp;        // [C] or null
    const bf16_t* w1;       // [Hd][128]  fc1 with the LayerNorm affine folded in
    const float* b1;        // [Hd]
    const bf16_t* w2;       // [C][Hdp]   fc2, Hdp = Hd rounded up to 128
    const float* b2;        // [C]
    const float* post_g;    // post-LayerNorm affine or null
    const float* post_b;
    const bf16_t* wn;       // [Nn][128]  next projection (LayerNorm affine / BN folded in) or null
    const float* bn;        // [Nn] or null
    bf16_t* out_next;       // [M][Nn]
    int M, C, Hd, Hdp;
    int Nn, next_ln, next_act;
    float eps1, eps_post, eps_next;
};

constexpr int kRcThreads = 512;
constexpr int kRcRows = 64;
constexpr int kRcRow = 256 + 16;            // 128 bf16 + pad
constexpr int kRcHRow = 512 + 16;           // 256 bf16 + pad ; also the fp32 staging row of 128 floats
constexpr int kRcA = 0;                                  // a tile, later LN(y)
constexpr int kRcY = kRcA + kRcRows * kRcRow;            // y tile (bf16)
constexpr int kRcW = kRcY + kRcRows * kRcRow;            // weight panel [128][272]
constexpr int kRcH = kRcW + 128 * kRcRow;                // hidden tile [64][528] ; fp32 staging of z
constexpr int kRcLds = kRcH + kRcRows * kRcHRow;         // 103,424 bytes

__global__ __launch_bounds__(kRcThreads) void row_chain_kernel(RowChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + kRcA;
    unsigned char* Ys = smem + kRcY;
    unsigned char* Ws = smem + kRcW;
    unsigned char* Hs = smem + kRcH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;         // 2 x 4 waves, 32 x 32 each
    const int m0 = blockIdx.x * kRcRows;
    const int ngc = (p.C * 2 + 31) / 32;             // 32-byte k-groups covering C channels

    // weight panel staging: thread t -> row t>>2, 64-byte quarter t&3 (4 x 16 bytes)
    const int wrow = tid >> 2, wsub = tid & 3;
    uint4 wreg[4];
    auto load_w = [&](const bf16_t* w, int ld, int row0, int nrows, int k0) {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            wreg[j] = (row0 + wrow < nrows) ? *(const uint4*)(w + (size_t)(row0 + wrow) * ld + k0 + (wsub * 4 + j) * 8)
                                            : make_uint4(0, 0, 0, 0);
    };
    auto store_w = [&]() {
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint4*)(Ws + wrow * kRcRow + (wsub * 4 + j) * 16) = wreg[j];
    };

    // ---- phase A: stage a (64 x C) and Wp
    {
        const int r = tid >> 3, sub = tid & 7;       // 8 threads per row, 32 bytes each
        const bool ok = m0 + r < p.M;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (sub * 2 + j) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok && k < p.C) v = *(const uint4*)(p.a + (size_t)(m0 + r) * p.C + k);
            *(uint4*)(As + r * kRcRow + (sub * 2 + j) * 16) = v;
        }
    }
    load_w(p.wp, 128, 0, p.C, 0);
    store_w();
    load_w(p.w1, 128, 0, p.Hd, 0);                   // prefetch the first fc1 panel
    __syncthreads();

    const int abase = (wm * 32 + ql) * kRcRow + h * 16;
    const int bbase = (wn * 32 + ql) * kRcRow + h * 16;
    f32x16 acc;
    auto zero_acc = [&]() {
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    };
    auto mma_panel = [&](const unsigned char* A, int a_off, int ng) {
        for (int g = 0; g < ng; ++g) {
            const uint4 af = *(const uint4*)(A + a_off + g * 32);
            const uint4 bf = *(const uint4*)(Ws + bbase + g * 32);
            mfma_kgroup<bf16_t>(af, bf, acc);
        }
    };

    // y = a . Wp^T + bp + skip  -> Ys (bf16, exactly what the unfused path would have stored)
    zero_acc();
    mma_panel(As, abase, ngc);
    {
        const int col = wn * 32 + ql;
        const float bias = (p.bp && col < p.C) ? p.bp[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + acc_row(r, lane);
            float v = acc[r] + bias;
            if (p.skip && col < p.C && m0 + row < p.M) v += bf2f(p.skip[(size_t)(m0 + row) * p.C + col].bits);
            if (col >= p.C) v = 0.f;
            *(uint16_t*)(Ys + row * kRcRow + col * 2) = f2bf(v);
        }
    }
    __syncthreads();                                  // Ys complete; As and Ws free

    // ---- phase B: x_hat = normalise(y) -> As ; 8 threads per row, 32 bytes (16 channels) each
    {
        const int r = tid >> 3, sub = tid & 7;
        float v[16];
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRcRow + sub * 32), v);
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRcRow + sub * 32 + 16), v + 8);
        float s = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) s += v[e];       // columns >= C are zero
        s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
        const float mean = s / (float)p.C;
        float q = 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) { const float d = (sub * 16 + e) < p.C ? v[e] - mean : 0.f; q += d * d; }
        q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
        const float rstd = rsqrtf(q / (float)p.C + p.eps1);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = (sub * 16 + e) < p.C ? (v[e] - mean) * rstd : 0.f;
        *(uint4*)(As + r * kRcRow + sub * 32) = f32_to_chunk<bf16_t>(v);
        *(uint4*)(As + r * kRcRow + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
    }
    store_w();                                        // fc1 panel 0 (prefetched) -> Ws
    __syncthreads();

    // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> Hs, 128 hidden columns per pass
    const int npass = (p.Hd + 127) / 128;
    for (int pass = 0; pass < npass; ++pass) {
        if (pass + 1 < npass) load_w(p.w1, 128, (pass + 1) * 128, p.Hd, 0);
        else load_w(p.w2, p.Hdp, 0, p.C, 0);          // prefetch the first fc2 panel
        zero_acc();
        mma_panel(As, abase, ngc);
        const int col = pass * 128 + wn * 32 + ql;
        const float bias = col < p.Hd ? p.b1[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + acc_row(r, lane);
            const float v = col < p.Hd ? gelu_erf(acc[r] + bias) : 0.f;
            *(uint16_t*)(Hs + row * kRcHRow + col * 2) = f2bf(v);
        }
        __syncthreads();                              // every wave done with Ws (and Hs columns written)
        store_w();
        __syncthreads();
    }

    // ---- phase D: z = hidden . W2^T + b2 + y
    zero_acc();
    for (int kt = 0; kt < npass; ++kt) {
        if (kt + 1 < npass) load_w(p.w2, p.Hdp, 0, p.C, (kt + 1) * 128);
        else if (p.wn) load_w(p.wn, 128, 0, p.Nn, 0);   // prefetch panel 0 of the next projection
        const int kleft = p.Hd - kt * 128;
        mma_panel(Hs, (wm * 32 + ql) * kRcHRow + kt * 256 + h * 16, kleft >= 128 ? 8 : (kleft * 2 + 31) / 32);
        if (kt + 1 < npass) {
            __syncthreads();
            store_w();
            __syncthreads();
        }
    }
    __syncthreads();                                  // Hs no longer read: reuse it as the fp32 staging of z
    float* stage = (float*)Hs;
    constexpr int SROW = kRcHRow / 4;
    {
        const int col = wn * 32 + ql;
        const float bias = col < p.C ? p.b2[col] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wm * 32 + acc_row(r, lane);
            const float y = bf2f(*(const uint16_t*)(Ys + row * kRcRow + col * 2));
            stage[row * SROW + col] = col < p.C ? acc[r] + bias + y : 0.f;
        }
    }
    const int npn = p.wn ? (p.Nn + 127) / 128 : 0;    // 128-column passes of the next projection
    if (npn) {
        store_w();                                    // every wave is past its last read of Ws (barrier above)
        load_w(p.wn, 128, 128, p.Nn, 0);              // prefetch panel 1 (zeros past Nn)
    }
    __syncthreads();

    // ---- phase E: optional post-LayerNorm, coalesced 16-byte stores ; 8 threads per row, 16 channels each
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool live = m0 + r < p.M;               // a row group (8 lanes) is entirely inside or outside M
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = stage[r * SROW + sub * 16 + e];
        if (p.post_g) {
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s / (float)p.C;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = (sub * 16 + e) < p.C ? v[e] - mean : 0.f; q += d * d; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            const float rstd = rsqrtf(q / (float)p.C + p.eps_post);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int c = sub * 16 + e;
                v[e] = c < p.C ? (v[e] - mean) * rstd * p.post_g[c] + p.post_b[c] : 0.f;
            }
        }
        uint4 o[2];
        o[0] = f32_to_chunk<bf16_t>(v);
        o[1] = f32_to_chunk<bf16_t>(v + 8);
        if (live) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c0 = sub * 16 + j * 8;
                if (c0 < p.C) *(uint4*)(p.out + (size_t)(m0 + r) * p.C + c0) = o[j];
            }
        }
        if (!npn) return;

        // ---- phase F: A operand of the next projection = (normalised) `out` rows exactly as stored (bf16) -> As
        if (p.next_ln) {
            chunk_to_f32<bf16_t>(o[0], v);
            chunk_to_f32<bf16_t>(o[1], v + 8);
            float s = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) s += v[e];
            s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
            const float mean = s / (float)p.C;
            float q = 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) { const float d = (sub * 16 + e) < p.C ? v[e] - mean : 0.f; q += d * d; }
            q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
            const float rstd = rsqrtf(q / (float)p.C + p.eps_next);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = (sub * 16 + e) < p.C ? (v[e] - mean) * rstd : 0.f;
            o[0] = f32_to_chunk<bf16_t>(v);
            o[1] = f32_to_chunk<bf16_t>(v + 8);
        }
        *(uint4*)(As + r * kRcRow + sub * 32) = o[0];
        *(uint4*)(As + r * kRcRow + sub * 32 + 16) = o[1];
    }
    __syncthreads();
    for (int pass = 0; pass < npn; ++pass) {
        zero_acc();
        mma_panel(As, abase, ngc);
        {
            const int col = pass * 128 + wn * 32 + ql;
            const float bias = (p.bn && col < p.Nn) ? p.bn[col] : 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wm * 32 + acc_row(r, lane);
                float v = acc[r] + bias;
                v = p.next_act == 1 ? fmaxf(v, 0.f) : (p.next_act == 2 ? gelu_erf(v) : v);
                *(uint16_t*)(Ys + row * kRcRow + (wn * 32 + ql) * 2) = f2bf(v);
            }
        }
        __syncthreads();                              // panel consumed, 64 x 128 result staged in Ys
        if (pass + 1 < npn) {
            store_w();
            if (pass + 2 < npn) load_w(p.wn, 128, (pass + 2) * 128, p.Nn, 0);
        }
        {
            const int r = tid >> 3, sub = tid & 7;
            if (m0 + r < p.M) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c0 = pass * 128 + sub * 16 + j * 8;
                    if (c0 < p.Nn) *(uint4*)(p.out_next + (size_t)(m0 + r) * p.Nn + c0) = *(const uint4*)(Ys + r * kRcRow + sub * 32 + j * 16);
                }
            }
        }
        if (pass + 1 < npn) __syncthreads();
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_attn_mlp_chain(const void* a, const void* skip, void* out, const void* wp, const float* bp,
                                     const void* w1, const float* b1, const void* w2, const float* b2,
                                     const float* post_gamma, const float* post_beta, const void* wnext, const float* bnext,
                                     void* out_next, const int* dims, float eps1, float eps_post, float eps_next,
                                     hipStream_t stream) {
    // dims: [dtype, M, C, Hd, Hdp, Nn, next_ln, next_act]
    if (!a || !out || !wp || !w1 || !b1 || !w2 || !b2 || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;           // bf16 mode only; fp32 runs the GEMMs separately
    RowChainParams p;
    p.a = (const bf16_t*)a; p.skip = (const bf16_t*)skip; p.out = (bf16_t*)out;
    p.wp = (const bf16_t*)wp; p.bp = bp; p.w1 = (const bf16_t*)w1; p.b1 = b1; p.w2 = (const bf16_t*)w2; p.b2 = b2;
    p.post_g = post_gamma; p.post_b = post_beta;
    p.wn = (const bf16_t*)wnext; p.bn = bnext; p.out_next = (bf16_t*)out_next;
    p.M = dims[1]; p.C = dims[2]; p.Hd = dims[3]; p.Hdp = dims[4];
    p.Nn = dims[5]; p.next_ln = dims[6]; p.next_act = dims[7];
    p.eps1 = eps1; p.eps_post = eps_post; p.eps_next = eps_next;
    if (p.M < 1 || p.C < 8 || p.C > 128 || p.C % 8 || p.Hd < 8 || p.Hd > 256 || p.Hd % 8) return COBEVT_ERR_SHAPE;
    if (p.Hdp % 128 || p.Hdp < p.Hd) return COBEVT_ERR_SHAPE;
    if ((post_gamma == nullptr) != (post_beta == nullptr)) return COBEVT_ERR_ARG;
    if ((wnext == nullptr) != (out_next == nullptr)) return COBEVT_ERR_ARG;
    if (wnext && (p.Nn < 8 || p.Nn % 8 || p.Nn > 1024 || p.next_act < 0 || p.next_act > 2)) return COBEVT_ERR_SHAPE;
    static bool attr_set = false;
    if (!attr_set) {
        (void)hipFuncSetAttribute((const void*)row_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kRcLds);
        attr_set = true;
    }
    const unsigned blocks = (unsigned)((p.M + kRcRows - 1) / kRcRows);
    hipLaunchKernelGGL(row_chain_kernel, dim3(blocks), dim3(kRcThreads), kRcLds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
