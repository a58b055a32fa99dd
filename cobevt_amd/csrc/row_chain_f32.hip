// The fused row-local chain of row_chain.hip for fp32 STORAGE (gfx950; round 6): the three fp32 compute modes - exact fp32 MFMA,
// "fp32_split", "fp32_fast" - ran the chain that follows every attention as four or five dense-row launches (gemm_rows.hip:
// out-projection + skip, LayerNorm + fc1 + GELU, fc2 + residual, [post-LayerNorm], next projection), 54 of the 80 dense-row launches
// of a 5-agent frame and most of their 2.2 ms: on the 5,120-row maps of pyramid level 2 and of the fusion stage each of those launches
// is 17-25 us of latency for 40-120 workgroups.  Same arithmetic, one launch:
//
//     y = a . Wp^T (+ bp) + skip                                  fax_modules.py:240,246-247 / swap_fusion_modules.py:126,177
//     z = y + ( GELU( LN(y) . W1'^T + b1' ) . W2^T + b2 )         fax_modules.py:411,435-437 / base_transformer.py:102-124
//     out = post-LayerNorm(z)  (optional)
//     next = act( LN?(out) . Wn'^T + bn' )  (optional)            the row-local GEMM that reads `out` next (row_chain.hip header)
//
// for C = 128 channels and a hidden width of 256 (every FAX level and the camera fusion stage; other widths keep the separate launches).
// Structure as the bf16 kernel's 32-row form: a 4-wave workgroup carries 32 rows through all GEMMs, wave w owns columns [32 w, 32 w + 32)
// of every 128-column pass, D = W . X^T so a lane holds one row and four runs of four consecutive columns, weights arrive as MFMA
// fragments straight from L2 in two ping-pong register sets of eight k-groups (an fp32 row of 128 channels is SIXTEEN 32-byte k-groups,
// so every K = 128 product is two sets and fc2's K = 256 four), the next set always in flight under the current one's MFMAs.
// LDS: y stays plain fp32 (the LayerNorm and the residual read it); the tiles that only feed MFMA operand reads - a, LN(y), the hidden
// activations, the next projection's input - are written through stage_x_piece, i.e. already split into (hi, lo) halves in the
// split-bf16 / fp16 libraries (common.hpp): the main loops are ds_read_b128 + MFMA, the weight fragment's split is shared by nothing
// (one 32-row tile per wave) but costs 12 VALU instructions per MFMA pair against 24.  73 KB per workgroup, two per CU.
#include "row_chain.hpp"

namespace cobevt {

namespace {

constexpr int kFRow = 512 + 16;             // 128 fp32 + pad (33 sixteen-byte slots: odd -> conflict-free ds_read_b128 down a column of rows)
constexpr int kFHRow = 1024 + 16;           // 256 fp32 + pad
constexpr int kFRows = 32;
constexpr int kFThreads = 256;
struct F32Lds {
    static constexpr int A = 0;                              // a, later LN(y), later the next projection's A operand (staged form)
    static constexpr int Y = A + kFRows * kFRow;             // y / z (plain fp32); staging of a next-projection pass
    static constexpr int H = Y + kFRows * kFRow;             // hidden tile [32][1040] (staged form)
    static constexpr int BIAS = H + kFRows * kFHRow;         // fp32 table: [0,128) bp, [128,384) b1, [384,512) b2, [512,640) post gamma, [640,768) post beta, [768,1536) bnext
    static constexpr int BYTES = BIAS + 4 * 1536;
};
constexpr int kBp = 0, kB1 = 128, kB2 = 384, kPg = 512, kPb = 640, kBn = 768, kBnMax = 768, kBiasFloats = 1536;

struct RowChainF32Params {
    const float* a; const float* skip; float* out;
    const uint4* wp; const float* bp; const uint4* w1; const float* b1; const uint4* w2; const float* b2;
    const float* post_g; const float* post_b;
    const uint4* wn; const float* bn; float* out_next;
    int M, Nn, next_ln, next_act, skip_rows;
    float eps1, eps_post, eps_next;
};

// normalise one 128-channel row held by 8 lanes (16 channels each)
__device__ __forceinline__ void normalise128(float (&v)[16], float eps) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64); s += __shfl_xor(s, 4, 64);
    const float mean = s * (1.0f / 128.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64); q += __shfl_xor(q, 4, 64);
    const float rstd = rsqrtf(q * (1.0f / 128.0f) + eps);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (v[e] - mean) * rstd;
}

__global__ __launch_bounds__(kFThreads, 2) void row_chain_f32_kernel(RowChainF32Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + F32Lds::A;
    unsigned char* Ys = smem + F32Lds::Y;
    unsigned char* Hs = smem + F32Lds::H;
    float* sb = (float*)(smem + F32Lds::BIAS);

    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int m0 = blockIdx.x * kFRows;
    const int row = ql;                               // this lane's row of the tile in every MFMA result
    const bool row_ok = m0 + row < p.M;
    const int cbase = wn * 32 + 4 * h;                // run k covers columns cbase + 8k .. +3 of the wave's 128-column panel

    // one set = eight consecutive k-groups of one 32-column tile (nkg k-groups per tile in the fragment array)
    auto load_set = [&](uint4 (&b)[8], const uint4* w, int tile, int nkg, int kg0) {
        const uint4* src = w + ((size_t)tile * nkg + kg0) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc;
    auto mma_set = [&](const unsigned char* A, int a_off, const uint4 (&b)[8], bool zero) {
        if (zero) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) mfma_kgroup_xs<float>(b[g], *(const uint4*)(A + a_off + g * 32), acc);   // D = W . X^T
    };

    uint4 fa[8], fb[8];
    load_set(fa, p.wp, wn, 16, 0);
    {   // bias table, branch-free (absent vectors read b1, which is never null; the select zeroes them)
        constexpr int NIT = kBiasFloats / kFThreads;
        float val[NIT];
        bool keep[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * kFThreads;
            const float* src = i < kB1 ? p.bp : i < kB2 ? p.b1 : i < kPg ? p.b2 : i < kPb ? p.post_g : i < kBn ? p.post_b : p.bn;
            const int j = i < kB1 ? i : i < kB2 ? i - kB1 : i < kPg ? i - kB2 : i < kPb ? i - kPg : i < kBn ? i - kPb : i - kBn;
            const int n = i < kB1 ? 128 : i < kB2 ? 256 : i < kBn ? 128 : p.Nn;
            keep[it] = (src != nullptr) & (j < n);
            val[it] = (src ? src : p.b1)[keep[it] ? j : 0];
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) sb[tid + it * kFThreads] = keep[it] ? val[it] : 0.f;
    }
    // ---- stage a: 8 threads per row, 64 bytes (16 channels = four 16-byte pieces) each
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool ok = m0 + r < p.M;
        uint4 v[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            v[j] = make_uint4(0, 0, 0, 0);
            if (ok) v[j] = *(const uint4*)(p.a + (size_t)(m0 + r) * 128 + sub * 16 + j * 4);
        }
#pragma unroll
        for (int j = 0; j < 4; ++j) *(uint4*)(As + r * kFRow + sub * 64 + j * 16) = stage_x_piece<float>(v[j]);
    }
    // skip values of this lane's (row, column runs), straight from global while the tile lands
    float4 skp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        skp[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        if (p.skip && row_ok) skp[k] = *(const float4*)(p.skip + (size_t)((m0 + row) % p.skip_rows) * 128 + cbase + 8 * k);
    }
    __syncthreads();

    const int abase = row * kFRow + h * 16;
    const int hbase = row * kFHRow + h * 16;

    // ---- phase A: y = a . Wp^T + bp + skip -> Ys
    load_set(fb, p.wp, wn, 16, 8);
    mma_set(As, abase, fa, true);
    load_set(fa, p.w1, wn, 16, 0);                    // fc1 columns [32 wn, +32), first half of K
    mma_set(As, abase + 256, fb, false);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 b = *(const float4*)(sb + kBp + col0);
        *(float4*)(Ys + row * kFRow + col0 * 4) =
            make_float4(acc[4 * k] + b.x + skp[k].x, acc[4 * k + 1] + b.y + skp[k].y, acc[4 * k + 2] + b.z + skp[k].z, acc[4 * k + 3] + b.w + skp[k].w);
    }
    __syncthreads();                                  // Ys complete; As free

    // ---- phase B: x_hat = normalise(y) -> As (staged form)
    load_set(fb, p.w1, wn, 16, 8);
    {
        const int r = tid >> 3, sub = tid & 7;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *(const float4*)(Ys + r * kFRow + sub * 64 + j * 16);
            v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
        normalise128(v, p.eps1);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(uint4*)(As + r * kFRow + sub * 64 + j * 16) =
                stage_x_piece<float>(make_uint4(__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])));
    }
    __syncthreads();

    // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> Hs (staged form), two 128-column passes
    auto hidden_epilogue = [&](int pass) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = *(const float4*)(sb + kB1 + col0);
            const float g0 = gelu_erf(acc[4 * k] + b.x), g1 = gelu_erf(acc[4 * k + 1] + b.y), g2 = gelu_erf(acc[4 * k + 2] + b.z), g3 = gelu_erf(acc[4 * k + 3] + b.w);
            *(uint4*)(Hs + row * kFHRow + col0 * 4) =
                stage_x_piece<float>(make_uint4(__float_as_uint(g0), __float_as_uint(g1), __float_as_uint(g2), __float_as_uint(g3)));
        }
    };
    mma_set(As, abase, fa, true);
    load_set(fa, p.w1, 4 + wn, 16, 0);                // fc1 columns [128 + 32 wn, +32)
    mma_set(As, abase + 256, fb, false);
    load_set(fb, p.w1, 4 + wn, 16, 8);
    hidden_epilogue(0);
    mma_set(As, abase, fa, true);
    load_set(fa, p.w2, wn, 32, 0);                    // fc2, k-groups 0..7 of 32 (hidden columns 0..63)
    mma_set(As, abase + 256, fb, false);
    load_set(fb, p.w2, wn, 32, 8);
    hidden_epilogue(1);
    __syncthreads();                                  // Hs complete

    // ---- phase D: z = hidden . W2^T + b2 + y -> Ys in place (a lane rewrites exactly the elements it reads)
    mma_set(Hs, hbase, fa, true);
    load_set(fa, p.w2, wn, 32, 16);
    mma_set(Hs, hbase + 256, fb, false);
    load_set(fb, p.w2, wn, 32, 24);
    mma_set(Hs, hbase + 512, fa, false);
    load_set(fa, p.wn ? p.wn : p.w1, wn, 16, 0);      // unconditional (w1 stands in when there is no next projection)
    mma_set(Hs, hbase + 768, fb, false);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 b = *(const float4*)(sb + kB2 + col0);
        float4* d = (float4*)(Ys + row * kFRow + col0 * 4);
        const float4 y = *d;
        *d = make_float4(acc[4 * k] + b.x + y.x, acc[4 * k + 1] + b.y + y.y, acc[4 * k + 2] + b.z + y.z, acc[4 * k + 3] + b.w + y.w);
    }
    const int npn = p.wn ? (p.Nn + 127) / 128 : 0;    // 128-column passes of the next projection
    __syncthreads();

    // ---- phase E: optional post-LayerNorm, coalesced 16-byte stores ; 8 threads per row, 16 channels each
    {
        const int r = tid >> 3, sub = tid & 7;
        const bool live = m0 + r < p.M;
        float v[16];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float4 t = *(const float4*)(Ys + r * kFRow + sub * 64 + j * 16);
            v[4 * j] = t.x; v[4 * j + 1] = t.y; v[4 * j + 2] = t.z; v[4 * j + 3] = t.w;
        }
        if (p.post_g) {
            normalise128(v, p.eps_post);
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const float4 g = *(const float4*)(sb + kPg + sub * 16 + e), b = *(const float4*)(sb + kPb + sub * 16 + e);
                v[e] = v[e] * g.x + b.x; v[e + 1] = v[e + 1] * g.y + b.y;
                v[e + 2] = v[e + 2] * g.z + b.z; v[e + 3] = v[e + 3] * g.w + b.w;
            }
        }
        if (live) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
                *(float4*)(p.out + (size_t)(m0 + r) * 128 + sub * 16 + j * 4) = make_float4(v[4 * j], v[4 * j + 1], v[4 * j + 2], v[4 * j + 3]);
        }
        if (!npn) return;
        // ---- phase F: A operand of the next projection = (normalised) `out` rows -> As (staged form)
        if (p.next_ln) normalise128(v, p.eps_next);
#pragma unroll
        for (int j = 0; j < 4; ++j)
            *(uint4*)(As + r * kFRow + sub * 64 + j * 16) =
                stage_x_piece<float>(make_uint4(__float_as_uint(v[4 * j]), __float_as_uint(v[4 * j + 1]), __float_as_uint(v[4 * j + 2]), __float_as_uint(v[4 * j + 3])));
    }
    __syncthreads();
    // one 128-column pass of the next projection: first half of K from `fa` (in flight since phase D), second half through `fb`
    for (int pass = 0; pass < npn; ++pass) {
        load_set(fb, p.wn, pass * 4 + wn, 16, 8);
        mma_set(As, abase, fa, true);
        if (pass + 1 < npn) load_set(fa, p.wn, (pass + 1) * 4 + wn, 16, 0);
        mma_set(As, abase + 256, fb, false);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = *(const float4*)(sb + kBn + col0);
            float v0 = acc[4 * k] + b.x, v1 = acc[4 * k + 1] + b.y, v2 = acc[4 * k + 2] + b.z, v3 = acc[4 * k + 3] + b.w;
            if (p.next_act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
            else if (p.next_act == 2) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
            *(float4*)(Ys + row * kFRow + (cbase + 8 * k) * 4) = make_float4(v0, v1, v2, v3);
        }
        __syncthreads();                              // 32 x 128 result staged in Ys
        {
            const int r = tid >> 3, sub = tid & 7;
            if (m0 + r < p.M) {
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int c0 = pass * 128 + sub * 16 + j * 4;
                    if (c0 < p.Nn) *(float4*)(p.out_next + (size_t)(m0 + r) * p.Nn + c0) = *(const float4*)(Ys + r * kFRow + sub * 64 + j * 16);
                }
            }
        }
        if (pass + 1 < npn) __syncthreads();          // Ys is rewritten by the next pass
    }
}

}  // namespace

// the fp32-storage form of cobevt_attn_mlp_chain (row_chain.hip): C = 128, hidden 256; returns -1 when the shape does not qualify
int launch_row_chain_f32(const void* a, const void* skip, void* out, const void* wp, const float* bp, const void* w1, const float* b1,
                         const void* w2, const float* b2, const float* post_g, const float* post_b, const void* wnext, const float* bnext,
                         void* out_next, int M, int C, int Hd, int Hdp, int Nn, int next_ln, int next_act, int skip_rows, float eps1,
                         float eps_post, float eps_next, hipStream_t stream) {
    if (C != 128 || Hd != 256 || Hdp != 256) return -1;
    if (wnext && (Nn < 4 || Nn % 4 || Nn > kBnMax)) return -1;
    RowChainF32Params p;
    p.a = (const float*)a; p.skip = (const float*)skip; p.out = (float*)out;
    p.wp = (const uint4*)wp; p.bp = bp; p.w1 = (const uint4*)w1; p.b1 = b1; p.w2 = (const uint4*)w2; p.b2 = b2;
    p.post_g = post_g; p.post_b = post_b; p.wn = (const uint4*)wnext; p.bn = bnext; p.out_next = (float*)out_next;
    p.M = M; p.Nn = Nn; p.next_ln = next_ln; p.next_act = next_act; p.skip_rows = skip_rows;
    p.eps1 = eps1; p.eps_post = eps_post; p.eps_next = eps_next;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)row_chain_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, F32Lds::BYTES);
    const unsigned blocks = (unsigned)((M + kFRows - 1) / kFRows);
    hipLaunchKernelGGL(row_chain_f32_kernel, dim3(blocks), dim3(kFThreads), F32Lds::BYTES, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt
