// Fused row-local chain that follows every attention of the FAX hot path (gfx950, bf16 mode):
//
//     y = a . Wp^T (+ bp) + skip                       out-projection + skip connection
//     z = y + ( GELU( LN(y) . W1'^T + b1' ) . W2^T + b2 )     pre-norm MLP with residual (LN affine folded into W1', b1')
//     out = post-LayerNorm(z)  (optional)
//     next = act( normalise(out) . Wn'^T + bn' )     (optional) the LayerNorm + Linear / 1x1 conv that consumes `out` next
//
// i.e. fax_modules.py:240,246-247 (proj + skip) followed by :411 / :435-437 (mlp_1 / mlp_2 + postnorm), and
// swap_fusion_modules.py:126,177 (to_out + PreNormResidual residual) followed by base_transformer.py:102-124
// (PreNormResidual(FeedForward)).  Rows are independent, so a workgroup carries a 64-row tile through all the GEMMs
// with y, LN(y) and the 2C-wide hidden activations resident in LDS; only `a`, `skip`, `out` (and `next`) touch HBM
// (the unfused path writes and re-reads y and the hidden tensor and costs three launches).  The optional `next` phase
// is the row-local GEMM that reads `out` in the reference graph - the to_qkv of the following swap-fusion attention
// (swap_fusion_modules.py:93 behind PreNormResidual.norm), the to_q of the second cross attention (fax_modules.py:201,
// 420-428) or the first 1x1 conv + BN + ReLU of the ResNetBottleNeck that follows (fax_modules.py:472) - done while the
// rows are still in LDS: in the latency-bound tail of the frame every separate launch costs 7-12 us for ~2 us of work.
//
// Weights arrive in MFMA fragment order [N/32 tiles][Kp/16 k-groups][64 lanes][16 bytes] (lane = 32*half + n%32 holds
// bytes [32*kgroup + 16*half, +16) of weight row n), so a wave's operand for one (32-column tile, k-group) is ONE coalesced
// 1-KB load from L2 straight into registers, prefetched one GEMM ahead in two ping-pong register sets: no weight panel in
// LDS, no barrier pair per panel, and 67 KB of LDS per workgroup instead of 101 KB -> two workgroups per CU, so one
// tile's barriers and epilogues hide under the other's MFMAs (the first version ran 95-104 us on the 81,920-row level-0
// maps with the MFMA pipe ~6 % busy).  The MFMAs are issued as D = W . X^T, so a lane ends up with one ROW of the tile
// and four runs of four consecutive output columns: every epilogue is 8-byte LDS traffic instead of sixteen 2-byte writes.
//
// 512 threads = 8 waves arranged 2 (row halves) x 4 (32-column tiles of a 128-column panel).  C <= 128, hidden <= 256.
#include "row_chain.hpp"

namespace cobevt {

constexpr int kRcRow = 256 + 16;            // 128 bf16 + pad
constexpr int kRcHRow = 512 + 16;           // 256 bf16 + pad ; also the fp32 staging row of 128 floats
// LDS regions for a ROWS-row tile (ROWS = 64: 8 waves, 68,608 B; ROWS = 32: 4 waves, 34,304 B -> four workgroups per CU)
template <int ROWS> struct RcLds {
    static constexpr int A = 0;                          // a tile, later LN(y), later the next projection's A operand
    static constexpr int Y = A + ROWS * kRcRow;          // y tile (bf16) ; staging of the next projection's output
    static constexpr int H = Y + ROWS * kRcRow;          // hidden tile [ROWS][528] ; fp32 staging of z
    static constexpr int TOTAL = H + ROWS * kRcHRow;
    static constexpr int BIAS = TOTAL;                   // fp32 table of every bias / affine vector of the chain (below)
    static constexpr int BYTES = TOTAL + 4 * 1536;       // 40,448 B at ROWS = 32: still four workgroups per CU
};
// bias table (floats): [0,128) bp, [128,384) b1, [384,512) b2, [512,640) post gamma, [640,768) post beta, [768,1536) bnext.
// Filled once per workgroup, zero where a vector is absent or shorter: the phase epilogues read it with unconditional
// ds_read_b128 instead of starting with a (branch-guarded -> vmcnt(0)) global round trip that also waited for the next
// phase's weight-fragment prefetch.
constexpr int kRcBp = 0, kRcB1 = 128, kRcB2 = 384, kRcPg = 512, kRcPb = 640, kRcBn = 768, kRcBnMax = 768, kRcBiasFloats = 1536;

// normalise one row held by 8 lanes (16 channels each; channels >= C are zero on entry and on exit)
__device__ __forceinline__ void rc_normalise(float (&v)[16], int sub, int C, float eps) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = (sub * 16 + e) < C ? v[e] - mean : 0.f; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (sub * 16 + e) < C ? (v[e] - mean) * rstd : 0.f;
}

// NPASS: 128-column passes over the hidden layer (Hd <= 128 -> 1, else 2).  ROWS: rows per workgroup (8 threads per row).
// FULL: C == 128, i.e. every operand row holds all eight 32-byte k-groups - the per-k-group `g < n` tests fold away and the
// fragment loads / MFMAs become straight-line code (with runtime n each group sat behind its own uniform branch).
template <int NPASS, int ROWS, bool FULL, bool MLP = true>
__global__ __launch_bounds__(ROWS * 8, 4) void row_chain_kernel(RowChainParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + RcLds<ROWS>::A;
    unsigned char* Ys = smem + RcLds<ROWS>::Y;
    unsigned char* Hs = smem + RcLds<ROWS>::H;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave >> 2, wn = wave & 3;         // (ROWS / 32) x 4 waves, 32 rows x 32 columns each
    const int m0 = blockIdx.x * ROWS;
    const int ngc = FULL ? 8 : (p.C * 2 + 31) / 32;  // 32-byte k-groups covering C channels
    const int row = wm * 32 + ql;                    // this lane's row of the tile in every MFMA result
    const bool row_ok = m0 + row < p.M;

    // fragment loads: up to 8 k-groups of one 32-column tile; fully unrolled predicated code so both register sets stay
    // in VGPRs.  sched_barrier keeps each prefetch where it is written (LLVM otherwise sinks the loads to their first use).
    auto load_frags = [&](uint4 (&b)[8], const uint4* w, int tile, int nkg, int kg0, int n) {
        const uint4* src = w + ((size_t)tile * nkg + kg0) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g < n) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc;
    auto mma = [&](const unsigned char* A, int a_off, const uint4 (&b)[8], int n, bool zero) {
        if (zero) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g)
            if (g < n) {
                const uint4 af = *(const uint4*)(A + a_off + g * 32);
                mfma_kgroup<bf16_t>(b[g], af, acc);  // D = W . X^T : acc register r <-> column acc_row(r), lane <-> row
            }
    };
    // this lane's four column runs: run k covers columns cbase + 8k .. +3 of the wave's 128-column panel
    const int cbase = wn * 32 + 4 * h;
    const float* sb = (const float*)(smem + RcLds<ROWS>::BIAS);
    {
        // branch-free and unconditional (absent vectors read bp / b1, which are never null; out-of-range entries read
        // element 0; both are zeroed by the select): all of a thread's loads are in flight together
        float* w = (float*)(smem + RcLds<ROWS>::BIAS);
        constexpr int NT = ROWS * 8, NIT = kRcBiasFloats / NT;
        static_assert(kRcBiasFloats % NT == 0, "bias table fill");
        float val[NIT];
        bool keep[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * NT;
            const float* src = i < kRcB1 ? p.bp : i < kRcB2 ? p.b1 : i < kRcPg ? p.b2 : i < kRcPb ? p.post_g : i < kRcBn ? p.post_b : p.bn;
            const int j = i < kRcB1 ? i : i < kRcB2 ? i - kRcB1 : i < kRcPg ? i - kRcB2 : i < kRcPb ? i - kRcPg : i < kRcBn ? i - kRcPb : i - kRcBn;
            const int n = i < kRcB1 ? p.C : i < kRcB2 ? p.Hd : i < kRcBn ? p.C : p.Nn;
            keep[it] = (src != nullptr) & (j < n);
            val[it] = (src ? src : p.b1)[keep[it] ? j : 0];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) w[tid + it * NT] = keep[it] ? val[it] : 0.f;
    }
    auto bias4 = [&](int table, int col0) { return *(const float4*)(sb + table + col0); };   // col0 % 4 == 0, inside the table
    auto pack4 = [&](float x, float y, float z, float w) { return make_uint2(pack_bf2(x, y), pack_bf2(z, w)); };

    uint4 fa[8], fb[8];
    const int abase = row * kRcRow + h * 16;

    // ---- stage a (64 x C) ; first weight fragments in flight meanwhile
    load_frags(fa, p.wp, wn, 8, 0, ngc);
    {
        const int r = tid >> 3, sub = tid & 7;       // 8 threads per row, 32 bytes each
        const bool ok = m0 + r < p.M;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int k = (sub * 2 + j) * 8;
            uint4 v = make_uint4(0, 0, 0, 0);
            if (ok && k < p.C) v = *(const uint4*)(p.a + (size_t)(m0 + r) * p.C + k);
            if (!MLP && p.pre_scale) {
                float f[8];
                chunk_to_f32<bf16_t>(v, f);
                const float4 s0 = *(const float4*)(p.pre_scale + k), s1 = *(const float4*)(p.pre_scale + k + 4);
                const float4 t0 = *(const float4*)(p.pre_shift + k), t1 = *(const float4*)(p.pre_shift + k + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = fmaf(f[e], sc[e], sh[e]);
                    if (p.pre_relu) f[e] = fmaxf(f[e], 0.f);
                }
                v = ok ? f32_to_chunk<bf16_t>(f) : make_uint4(0, 0, 0, 0);
            }
            *(uint4*)(As + r * kRcRow + (sub * 2 + j) * 16) = v;
        }
    }
    // skip values of this lane's (row, column runs), straight from global while the tile lands
    uint2 skp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        skp[k] = make_uint2(0, 0);
        if (p.skip && row_ok && col0 < p.C) skp[k] = *(const uint2*)(p.skip + (size_t)((m0 + row) % p.skip_rows) * p.C + col0);
    }
    __syncthreads();

    // ---- phase A: y = a . Wp^T + bp + skip -> Ys (bf16, exactly what the unfused path would have stored)
    mma(As, abase, fa, ngc, true);
    if (MLP) load_frags(fb, p.w1, wn, 8, 0, ngc);    // fc1 columns [32 wn, +32) ; needed after phase B
    else load_frags(fb, p.wn, wn, 8, 0, ngc);        // projection chain: the next projection's first pass
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 b = bias4(kRcBp, col0);
        float v0 = acc[4 * k] + b.x + bf2f(skp[k].x & 0xffff), v1 = acc[4 * k + 1] + b.y + bf2f(skp[k].x >> 16);
        float v2 = acc[4 * k + 2] + b.z + bf2f(skp[k].y & 0xffff), v3 = acc[4 * k + 3] + b.w + bf2f(skp[k].y >> 16);
        if (col0 >= p.C) v0 = v1 = v2 = v3 = 0.f;
        *(uint2*)(Ys + row * kRcRow + col0 * 2) = pack4(v0, v1, v2, v3);
    }
    __syncthreads();                                  // Ys complete; As free
    float* stage = (float*)Hs;
    constexpr int SROW = kRcHRow / 4;
    if constexpr (MLP) {

    // ---- phase B: x_hat = normalise(y) -> As ; 8 threads per row, 32 bytes (16 channels) each
    if (NPASS == 2) load_frags(fa, p.w1, 4 + wn, 8, 0, ngc);   // fc1 columns [128 + 32 wn, +32)
    else load_frags(fa, p.w2, wn, p.Hdp / 16, 0, 8);
    {
        const int r = tid >> 3, sub = tid & 7;
        float v[16];
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRcRow + sub * 32), v);
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRcRow + sub * 32 + 16), v + 8);
        rc_normalise(v, sub, p.C, p.eps1);
        *(uint4*)(As + r * kRcRow + sub * 32) = f32_to_chunk<bf16_t>(v);
        *(uint4*)(As + r * kRcRow + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
    }
    __syncthreads();

    // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> Hs, 128 hidden columns per pass
    auto hidden_epilogue = [&](int pass) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = bias4(kRcB1, col0);
            uint2 o = make_uint2(0, 0);
            if (col0 < p.Hd) o = pack4(gelu_bf16(acc[4 * k] + b.x), gelu_bf16(acc[4 * k + 1] + b.y),
                                       gelu_bf16(acc[4 * k + 2] + b.z), gelu_bf16(acc[4 * k + 3] + b.w));
            *(uint2*)(Hs + row * kRcHRow + col0 * 2) = o;
        }
    };
    const int hbase = row * kRcHRow + h * 16;
    if (NPASS == 2) {
        mma(As, abase, fb, ngc, true);
        load_frags(fb, p.w2, wn, p.Hdp / 16, 0, 8);  // fc2 k-groups 0..7 (hidden columns 0..127)
        hidden_epilogue(0);
        mma(As, abase, fa, ngc, true);
        load_frags(fa, p.w2, wn, p.Hdp / 16, 8, 8);  // fc2 k-groups 8..15
        hidden_epilogue(1);
        __syncthreads();
        // ---- phase D: z = hidden . W2^T
        mma(Hs, hbase, fb, 8, true);
        load_frags(fb, p.wn ? p.wn : p.w1, wn, 8, 0, ngc);   // unconditional (w1 stands in when there is no next projection)
        mma(Hs, hbase + 256, fa, 8, false);
    } else {
        mma(As, abase, fb, ngc, true);
        load_frags(fb, p.wn ? p.wn : p.w1, wn, 8, 0, ngc);   // unconditional (w1 stands in when there is no next projection)
        hidden_epilogue(0);
        __syncthreads();
        mma(Hs, hbase, fa, (p.Hd * 2 + 31) / 32, true);
    }
    __syncthreads();                                  // Hs no longer read: reuse it as the fp32 staging of z
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 b = bias4(kRcB2, col0);
        const uint2 y = *(const uint2*)(Ys + row * kRcRow + col0 * 2);
        float4 z = make_float4(0.f, 0.f, 0.f, 0.f);
        if (col0 < p.C)
            z = make_float4(acc[4 * k] + b.x + bf2f(y.x & 0xffff), acc[4 * k + 1] + b.y + bf2f(y.x >> 16),
                            acc[4 * k + 2] + b.z + bf2f(y.y & 0xffff), acc[4 * k + 3] + b.w + bf2f(y.y >> 16));
        *(float4*)(stage + row * SROW + col0) = z;
    }
    } else {
        // projection chain: "z" = y exactly as the unfused path would have stored it (bf16), widened into the staging tile
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = cbase + 8 * k;
            const uint2 y = *(const uint2*)(Ys + row * kRcRow + col0 * 2);
            *(float4*)(stage + row * SROW + col0) = make_float4(bf2f(y.x & 0xffff), bf2f(y.x >> 16), bf2f(y.y & 0xffff), bf2f(y.y >> 16));
        }
    }
    const int npn = p.wn ? (p.Nn + 127) / 128 : 0;    // 128-column passes of the next projection
    __syncthreads();

    // ---- phase E: optional post-LayerNorm, coalesced 16-byte stores ; 8 threads per row, 16 channels each
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool live = m0 + r < p.M;               // a row group (8 lanes) is entirely inside or outside M
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const float4 t = *(const float4*)(stage + r * SROW + sub * 16 + e);
            v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        if (p.post_g) {
            rc_normalise(v, sub, p.C, p.eps_post);
#pragma unroll
            for (int e = 0; e < 16; e += 4) {          // channels >= C: gamma = beta = 0 in the table -> 0
                const float4 g = *(const float4*)(sb + kRcPg + sub * 16 + e), b = *(const float4*)(sb + kRcPb + sub * 16 + e);
                v[e] = v[e] * g.x + b.x; v[e + 1] = v[e + 1] * g.y + b.y;
                v[e + 2] = v[e + 2] * g.z + b.z; v[e + 3] = v[e + 3] * g.w + b.w;
            }
        }
        uint4 o[2];
        o[0] = f32_to_chunk<bf16_t>(v);
        o[1] = f32_to_chunk<bf16_t>(v + 8);
        if (live && (MLP || p.out)) {
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
            rc_normalise(v, sub, p.C, p.eps_next);
            o[0] = f32_to_chunk<bf16_t>(v);
            o[1] = f32_to_chunk<bf16_t>(v + 8);
        }
        *(uint4*)(As + r * kRcRow + sub * 32) = o[0];
        *(uint4*)(As + r * kRcRow + sub * 32 + 16) = o[1];
    }
    __syncthreads();
    // one 128-column pass of the next projection from fragment set `cur`; the following pass's fragments go to `nxt`
    auto next_pass = [&](int pass, const uint4 (&cur)[8], uint4 (&nxt)[8]) {
        if (pass + 1 < npn) load_frags(nxt, p.wn, (pass + 1) * 4 + wn, 8, 0, ngc);
        mma(As, abase, cur, ngc, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = bias4(kRcBn, col0);
            float v0 = acc[4 * k] + b.x, v1 = acc[4 * k + 1] + b.y, v2 = acc[4 * k + 2] + b.z, v3 = acc[4 * k + 3] + b.w;
            if (p.next_act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            else if (p.next_act == 2) { v0 = gelu_bf16(v0); v1 = gelu_bf16(v1); v2 = gelu_bf16(v2); v3 = gelu_bf16(v3); }
            *(uint2*)(Ys + row * kRcRow + (cbase + 8 * k) * 2) = pack4(v0, v1, v2, v3);
        }
        __syncthreads();                              // 64 x 128 result staged in Ys
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
        if (pass + 1 < npn) __syncthreads();          // Ys is rewritten by the next pass
    };
    for (int pass = 0; pass < npn; pass += 2) {
        next_pass(pass, fb, fa);
        if (pass + 1 < npn) next_pass(pass + 1, fa, fb);
    }
}

template <int NPASS, int ROWS, bool FULL, bool MLP = true>
static void launch_chain(const RowChainParams& p, hipStream_t stream) {
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)row_chain_kernel<NPASS, ROWS, FULL, MLP>, hipFuncAttributeMaxDynamicSharedMemorySize, RcLds<ROWS>::BYTES);
    }
    const unsigned blocks = (unsigned)((p.M + ROWS - 1) / ROWS);
    hipLaunchKernelGGL((row_chain_kernel<NPASS, ROWS, FULL, MLP>), dim3(blocks), dim3(ROWS * 8), RcLds<ROWS>::BYTES, stream, p);
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_attn_mlp_chain(const void* a, const void* skip, void* out, const void* wp, const float* bp,
                                     const void* w1, const float* b1, const void* w2, const float* b2,
                                     const float* post_gamma, const float* post_beta, const void* wnext, const float* bnext,
                                     void* out_next, const int* dims, float eps1, float eps_post, float eps_next,
                                     hipStream_t stream) {
    // dims: [dtype, M, C, Hd, Hdp, Nn, next_ln, next_act, rows_per_workgroup, skip_rows]
    if (!a || !out || !wp || !w1 || !b1 || !w2 || !b2 || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0 && dims[0] != 1) return COBEVT_ERR_ARG;
    if (dims[0] == 1) {                                        // fp32 storage: the C = 128 / hidden 256 chain (row_chain_f32.hip); other widths run the GEMMs separately
        if ((post_gamma == nullptr) != (post_beta == nullptr) || (wnext == nullptr) != (out_next == nullptr)) return COBEVT_ERR_ARG;
        const int M = dims[1], sr = dims[9] > 0 ? dims[9] : dims[1];
        if (M < 1 || sr > M || M % sr || (wnext && (dims[7] < 0 || dims[7] > 2))) return COBEVT_ERR_SHAPE;
        const int rc = launch_row_chain_f32(a, skip, out, wp, bp, w1, b1, w2, b2, post_gamma, post_beta, wnext, bnext, out_next, M, dims[2], dims[3],
                                            dims[4], dims[5], dims[6], dims[7], sr, eps1, eps_post, eps_next, stream);
        return rc < 0 ? COBEVT_ERR_UNSUPPORTED : rc;
    }
    RowChainParams p;
    p.a = (const bf16_t*)a; p.skip = (const bf16_t*)skip; p.out = (bf16_t*)out;
    p.wp = (const uint4*)wp; p.bp = bp; p.w1 = (const uint4*)w1; p.b1 = b1; p.w2 = (const uint4*)w2; p.b2 = b2;
    p.post_g = post_gamma; p.post_b = post_beta;
    p.wn = (const uint4*)wnext; p.bn = bnext; p.out_next = (bf16_t*)out_next;
    p.M = dims[1]; p.C = dims[2]; p.Hd = dims[3]; p.Hdp = dims[4];
    p.Nn = dims[5]; p.next_ln = dims[6]; p.next_act = dims[7];
    p.skip_rows = dims[9] > 0 ? dims[9] : p.M;                 // dims[9]: rows of `skip` (0 = M), M % skip_rows == 0
    p.eps1 = eps1; p.eps_post = eps_post; p.eps_next = eps_next;
    p.pre_scale = p.pre_shift = nullptr; p.pre_relu = 0;
    if (p.M < 1 || p.C < 8 || p.C > 128 || p.C % 8 || p.Hd < 8 || p.Hd > 256 || p.Hd % 8) return COBEVT_ERR_SHAPE;
    if (p.Hdp % 128 || p.Hdp < p.Hd || p.Hdp > 256) return COBEVT_ERR_SHAPE;
    if (p.skip_rows > p.M || p.M % p.skip_rows) return COBEVT_ERR_SHAPE;
    if ((post_gamma == nullptr) != (post_beta == nullptr)) return COBEVT_ERR_ARG;
    if ((wnext == nullptr) != (out_next == nullptr)) return COBEVT_ERR_ARG;
    if (wnext && (p.Nn < 8 || p.Nn % 8 || p.Nn > kRcBnMax || p.next_act < 0 || p.next_act > 2)) return COBEVT_ERR_SHAPE;
    if (dims[8] == 0) {                                        // automatic: 64-channel chains on big maps take the persistent form
        const int rc = launch_row_chain64(p, stream);
        if (rc >= 0) return rc;
    }
    const int rows = dims[8] == 64 ? 64 : 32;                  // dims[8]: rows per workgroup (0 = default 32; 32 / 64 pin the generic kernel)
    const bool two = p.Hd > 128, full = p.C == 128;
    if (rows == 64) {
        if (two) { if (full) launch_chain<2, 64, true>(p, stream); else launch_chain<2, 64, false>(p, stream); }
        else { if (full) launch_chain<1, 64, true>(p, stream); else launch_chain<1, 64, false>(p, stream); }
    } else {
        if (two) { if (full) launch_chain<2, 32, true>(p, stream); else launch_chain<2, 32, false>(p, stream); }
        else { if (full) launch_chain<1, 32, true>(p, stream); else launch_chain<1, 32, false>(p, stream); }
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// Projection chain, see include/cobevt_hip.h: y = ReLU?(a * pre_scale + pre_shift) . Wp^T + bp + skip ; next = act(LN?(y) . Wn'^T + bn')
extern "C" int cobevt_proj_chain(const void* a, const float* pre_scale, const float* pre_shift, const void* skip, const void* wp,
                                 const float* bp, void* out, const void* wnext, const float* bnext, void* out_next, const int* dims,
                                 float eps_next, hipStream_t stream) {
    // dims: [dtype, M, C (= 128), Nn, next_ln, next_act, skip_rows, pre_relu, variant (0 = automatic, 1 = the barrier-phased kernel)]
    if (!a || !wp || !wnext || !bnext || !out_next || !dims) return COBEVT_ERR_ARG;
    if ((pre_scale == nullptr) != (pre_shift == nullptr)) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    RowChainParams p;
    p.a = (const bf16_t*)a; p.skip = (const bf16_t*)skip; p.out = (bf16_t*)out;
    p.wp = (const uint4*)wp; p.bp = bp;
    p.w1 = (const uint4*)wnext; p.b1 = bnext; p.w2 = (const uint4*)wnext; p.b2 = bnext;      // unused by the projection chain (valid stand-ins)
    p.post_g = p.post_b = nullptr;
    p.wn = (const uint4*)wnext; p.bn = bnext; p.out_next = (bf16_t*)out_next;
    p.M = dims[1]; p.C = dims[2]; p.Hd = 8; p.Hdp = 128;
    p.Nn = dims[3]; p.next_ln = dims[4]; p.next_act = dims[5];
    p.skip_rows = dims[6] > 0 ? dims[6] : p.M;
    p.pre_scale = pre_scale; p.pre_shift = pre_shift; p.pre_relu = dims[7];
    p.eps1 = 0.f; p.eps_post = 0.f; p.eps_next = eps_next;
    if (p.M < 1 || p.C != 128) return COBEVT_ERR_SHAPE;             // K = C = 128 (the level-0 feature / embedding width)
    if (p.Nn < 8 || p.Nn % 8 || p.Nn > kRcBnMax || p.next_act < 0 || p.next_act > 2) return COBEVT_ERR_SHAPE;
    if (p.skip_rows > p.M || p.M % p.skip_rows) return COBEVT_ERR_SHAPE;
    if (dims[8] == 0) {
        const int rc = launch_proj_chain128(p, stream);         // big maps: rows in registers, weights in LDS (proj_chain128.hip)
        if (rc >= 0) return rc;
    }
    launch_chain<1, 32, true, false>(p, stream);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
