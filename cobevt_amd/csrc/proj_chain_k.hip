// Key AND value side of a FAX pyramid level whose image features are wider than the level (gfx950, bf16 mode): ONE launch for
//
//     key = ReLU(BN_k(feature)) . Wk^T + ray embedding          feature_proj   fax_modules.py:281-292, 392-396
//     val = ReLU(BN_v(feature)) . Wv^T                          feature_linear fax_modules.py:281-292, 394
//     kk  = LN(key) . [Wk1' | Wk2']^T + [bk1' | bk2']           to_k of both cross attentions, fax_modules.py:201-205 (LayerNorm affine folded)
//     vv  = LN(val) . [Wv1' | Wv2']^T + [bv1' | bv2']           to_v of both cross attentions
//
// i.e. the projection chain of row_chain.hip (MLP = false) / proj_chain128.hip for feature widths K = 256 / 384 / 512 (pyramid levels
// 1 and 2 of corpbevt.yaml: ResNet-34 layer3 / layer4 maps): the four dense-row launches per level that existed only to move the
// `key` / `val` rows through HBM (VERDICT r04 missing #3, weak #4: 8 of the 18 gemm_rows3 launches of a frame, 160 / 640
// workgroups each).  blockIdx.y = side (0 key, 1 value): both read the same feature rows, which therefore come from L2 once.
//
// A workgroup (4 waves) owns 32 rows.  The first GEMM walks K in 128-channel chunks: the next chunk's rows are in flight in
// registers and the next chunk's weight fragments in the other register set while the MFMAs of the current one issue; `key` /
// `val` (bf16, exactly what the unfused path stores) stays in LDS, is normalised there and feeds the stacked second GEMM.
// Weights arrive in MFMA fragment order [N/32 tiles][K/16 k-groups][64 lanes][16 B] straight from L2 (ops.ConvPlan.wfrag_rows);
// D = W . X^T, so a lane owns one row and four runs of four consecutive columns.
#include "row_chain.hpp"

namespace cobevt {

namespace {

constexpr int kRow = 256 + 16;              // 128 bf16 + pad (conflict-free ds_read_b128 of the A fragments)
constexpr int kNnMax = 768;

struct ProjChainKSide {
    const float* pre_scale;  // [K] folded eval BatchNorm in front of the 1x1 conv (null: none)
    const float* pre_shift;
    const uint4* wp;         // fragment-ordered [4 tiles][K/16]
    const float* bp;         // [128] or null
    const bf16_t* skip;      // [skip_rows][128] or null (the ray embedding of the key side)
    const uint4* wn;         // fragment-ordered [ceil(Nn/128)*4 tiles][8]
    const float* bn;         // [Nn]
    bf16_t* out;             // [M][128] or null: the key / value map itself (not needed by the FAX path)
    bf16_t* out_next;        // [M][Nn]
    int pre_relu, skip_rows;
};

struct ProjChainKParams {
    const bf16_t* a;         // [M][K] feature rows
    ProjChainKSide side[2];
    int M, K, Nn, next_ln;
    float eps_next;
};

struct Lds {
    static constexpr int A = 0;                         // staged chunk of a ; later LN(y)
    static constexpr int Y = A + 32 * kRow;             // y tile (bf16) ; staging of the next projection's output
    static constexpr int BIAS = Y + 32 * kRow;          // fp32: [0,128) bp, [128, 128 + 768) bn
    static constexpr int BYTES = BIAS + 4 * (128 + kNnMax);
};

__device__ __forceinline__ void normalise128(float (&v)[16], float eps) {      // one row held by 8 lanes, 16 channels each
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

// NCH = K / 128 chunks of the first GEMM (2, 3 or 4)
template <int NCH>
__global__ __launch_bounds__(256, 4) void proj_chain_k_kernel(ProjChainKParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + Lds::A;
    unsigned char* Ys = smem + Lds::Y;
    float* sb = (float*)(smem + Lds::BIAS);

    // this workgroup's side, field by field through uniform selects (an indexed copy of the kernel-argument struct goes to scratch)
    ProjChainKSide sd;
    {
        const bool s1 = blockIdx.y != 0;
        const ProjChainKSide &u = p.side[0], &w = p.side[1];
        sd.pre_scale = s1 ? w.pre_scale : u.pre_scale; sd.pre_shift = s1 ? w.pre_shift : u.pre_shift;
        sd.wp = s1 ? w.wp : u.wp; sd.bp = s1 ? w.bp : u.bp; sd.skip = s1 ? w.skip : u.skip;
        sd.wn = s1 ? w.wn : u.wn; sd.bn = s1 ? w.bn : u.bn; sd.out = s1 ? w.out : u.out; sd.out_next = s1 ? w.out_next : u.out_next;
        sd.pre_relu = s1 ? w.pre_relu : u.pre_relu; sd.skip_rows = s1 ? w.skip_rows : u.skip_rows;
    }
    const int tid = threadIdx.x, lane = tid & 63, wn = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int m0 = blockIdx.x * 32;
    const int row = ql;                               // this lane's row of the tile in every MFMA result
    const bool row_ok = m0 + row < p.M;
    const int sr = tid >> 3, sub = tid & 7;           // staging: 8 threads per row, 16 channels (32 bytes) each
    const bool srow_ok = m0 + sr < p.M;
    constexpr int NKG = NCH * 8;                      // k-groups of a weight row of the first GEMM

    // bias table: every load of a thread in flight together, absent vectors -> 0
    {
        constexpr int NIT = (128 + kNnMax + 255) / 256;
        float val[NIT];
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            const bool isp = i < 128;
            const float* src = isp ? sd.bp : sd.bn;
            const int j = isp ? i : i - 128;
            const bool keep = (src != nullptr) & (i < 128 + kNnMax) & (isp | (j < p.Nn));
            val[it] = keep ? src[j] : 0.f;
        }
#pragma unroll
        for (int it = 0; it < NIT; ++it) {
            const int i = tid + it * 256;
            if (i < 128 + kNnMax) sb[i] = val[it];
        }
    }

    auto load_frags = [&](uint4 (&b)[8], const uint4* w, int tile, int nkg, int kg0) {
        const uint4* src = w + ((size_t)tile * nkg + kg0) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) b[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc;
    auto mma = [&](const uint4 (&b)[8], bool zero) {
        if (zero) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
        const int abase = row * kRow + h * 16;
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const uint4 af = *(const uint4*)(As + abase + g * 32);
            mfma_kgroup<bf16_t>(b[g], af, acc);       // D = W . X^T : register r <-> column acc_row(r), lane <-> row
        }
    };
    // rows of chunk c of this thread's row: 16 channels = two 16-byte pieces
    auto load_rows = [&](uint4 (&v)[2], int c) {
        const bf16_t* src = p.a + (size_t)(srow_ok ? m0 + sr : 0) * p.K + c * 128 + sub * 16;
        v[0] = *(const uint4*)src;
        v[1] = *(const uint4*)(src + 8);
    };
    // pre-activation (folded BatchNorm -> ReLU) of chunk c, then into the A tile
    auto stage_rows = [&](const uint4 (&v)[2], int c) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            uint4 o = v[j];
            if (sd.pre_scale) {
                const int k = c * 128 + sub * 16 + j * 8;
                float f[8];
                chunk_to_f32<bf16_t>(v[j], f);
                const float4 s0 = *(const float4*)(sd.pre_scale + k), s1 = *(const float4*)(sd.pre_scale + k + 4);
                const float4 t0 = *(const float4*)(sd.pre_shift + k), t1 = *(const float4*)(sd.pre_shift + k + 4);
                const float sc[8] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w}, sh[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    f[e] = fmaf(f[e], sc[e], sh[e]);
                    if (sd.pre_relu) f[e] = fmaxf(f[e], 0.f);
                }
                o = f32_to_chunk<bf16_t>(f);
            }
            if (!srow_ok) o = make_uint4(0, 0, 0, 0);
            *(uint4*)(As + sr * kRow + sub * 32 + j * 16) = o;
        }
    };

    uint4 fa[8], fb[8];
    uint4 ra[2], rb[2];
    load_rows(ra, 0);
    load_frags(fa, sd.wp, wn, NKG, 0);
    // the skip values (ray embedding) of this lane's (row, column runs), straight from global under the first GEMM
    const int cbase = wn * 32 + 4 * h;
    uint2 skp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        skp[k] = make_uint2(0, 0);
        if (sd.skip && row_ok) skp[k] = *(const uint2*)(sd.skip + (size_t)((m0 + row) % sd.skip_rows) * 128 + cbase + 8 * k);
    }

    // ---- phase A: y = act(a) . Wp^T over NCH chunks ; chunk c+1's rows and fragments in flight under chunk c's MFMAs
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c & 1) stage_rows(rb, c); else stage_rows(ra, c);
        if (c + 1 < NCH) {
            if (c & 1) { load_rows(ra, c + 1); load_frags(fa, sd.wp, wn, NKG, 8 * (c + 1)); }
            else { load_rows(rb, c + 1); load_frags(fb, sd.wp, wn, NKG, 8 * (c + 1)); }
        } else {                                      // the next projection's first pass
            if (c & 1) load_frags(fa, sd.wn, wn, 8, 0); else load_frags(fb, sd.wn, wn, 8, 0);
        }
        __syncthreads();                              // chunk staged (first iteration: bias table too)
        if (c & 1) mma(fb, false); else mma(fa, c == 0);
        __syncthreads();                              // every wave is done with the A tile
    }
    // after the loop the fragments of the next projection's pass 0 sit in: NCH even -> fa, NCH odd -> fb

    // ---- y = acc + bp + skip -> Ys as bf16 (exactly what the unfused path stores as the key / value map)
    auto pack4 = [&](float x, float y, float z, float w) { return make_uint2(pack_bf2(x, y), pack_bf2(z, w)); };
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 b = *(const float4*)(sb + col0);
        *(uint2*)(Ys + row * kRow + col0 * 2) = pack4(acc[4 * k] + b.x + bf2f(skp[k].x & 0xffff), acc[4 * k + 1] + b.y + bf2f(skp[k].x >> 16),
                                                      acc[4 * k + 2] + b.z + bf2f(skp[k].y & 0xffff), acc[4 * k + 3] + b.w + bf2f(skp[k].y >> 16));
    }
    __syncthreads();

    // ---- LayerNorm of the rows as stored (bf16) -> A tile of the next projection ; optional store of the map itself
    {
        const uint4 y0 = *(const uint4*)(Ys + sr * kRow + sub * 32), y1 = *(const uint4*)(Ys + sr * kRow + sub * 32 + 16);
        if (sd.out && srow_ok) {
            *(uint4*)(sd.out + (size_t)(m0 + sr) * 128 + sub * 16) = y0;
            *(uint4*)(sd.out + (size_t)(m0 + sr) * 128 + sub * 16 + 8) = y1;
        }
        uint4 o0 = y0, o1 = y1;
        if (p.next_ln) {
            float v[16];
            chunk_to_f32<bf16_t>(y0, v);
            chunk_to_f32<bf16_t>(y1, v + 8);
            normalise128(v, p.eps_next);
            o0 = f32_to_chunk<bf16_t>(v);
            o1 = f32_to_chunk<bf16_t>(v + 8);
        }
        *(uint4*)(As + sr * kRow + sub * 32) = o0;
        *(uint4*)(As + sr * kRow + sub * 32 + 16) = o1;
    }
    __syncthreads();

    // ---- next projection in 128-column passes from fragment set `cur`; the following pass's fragments go to `nxt`
    const int npn = (p.Nn + 127) / 128;
    auto next_pass = [&](int pass, const uint4 (&cur)[8], uint4 (&nxt)[8]) {
        if (pass + 1 < npn) load_frags(nxt, sd.wn, (pass + 1) * 4 + wn, 8, 0);
        mma(cur, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 b = *(const float4*)(sb + 128 + col0);
            *(uint2*)(Ys + row * kRow + (cbase + 8 * k) * 2) = pack4(acc[4 * k] + b.x, acc[4 * k + 1] + b.y, acc[4 * k + 2] + b.z, acc[4 * k + 3] + b.w);
        }
        __syncthreads();                              // 32 x 128 result staged in Ys
        if (srow_ok) {
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int c0 = pass * 128 + sub * 16 + j * 8;
                if (c0 < p.Nn) *(uint4*)(sd.out_next + (size_t)(m0 + sr) * p.Nn + c0) = *(const uint4*)(Ys + sr * kRow + sub * 32 + j * 16);
            }
        }
        if (pass + 1 < npn) __syncthreads();          // Ys is rewritten by the next pass
    };
    for (int pass = 0; pass < npn; pass += 2) {
        if (NCH & 1) {
            next_pass(pass, fb, fa);
            if (pass + 1 < npn) next_pass(pass + 1, fa, fb);
        } else {
            next_pass(pass, fa, fb);
            if (pass + 1 < npn) next_pass(pass + 1, fb, fa);
        }
    }
}

template <int NCH>
void launch(const ProjChainKParams& p, int nsides, hipStream_t stream) {
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first())
        (void)hipFuncSetAttribute((const void*)proj_chain_k_kernel<NCH>, hipFuncAttributeMaxDynamicSharedMemorySize, Lds::BYTES);
    hipLaunchKernelGGL((proj_chain_k_kernel<NCH>), dim3((unsigned)((p.M + 31) / 32), (unsigned)nsides), dim3(256), Lds::BYTES, stream, p);
}

}  // namespace

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_proj_chain_kv(const void* a, const void* const* ptrs, const int* dims, float eps_next, hipStream_t stream) {
    // ptrs: per side s (0 key, 1 value) at [9 s ..]: pre_scale, pre_shift, wp, bp, skip, wn, bn, out, out_next
    // dims: [dtype, M, K, Nn, next_ln, nsides, pre_relu_0, skip_rows_0, pre_relu_1, skip_rows_1]
    if (!a || !ptrs || !dims) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;
    ProjChainKParams p;
    p.a = (const bf16_t*)a;
    p.M = dims[1]; p.K = dims[2]; p.Nn = dims[3]; p.next_ln = dims[4];
    const int nsides = dims[5];
    p.eps_next = eps_next;
    if (nsides < 1 || nsides > 2) return COBEVT_ERR_ARG;
    if (p.M < 1 || p.K < 256 || p.K > 512 || p.K % 128) return COBEVT_ERR_SHAPE;
    if (p.Nn < 8 || p.Nn % 8 || p.Nn > kNnMax) return COBEVT_ERR_SHAPE;
    for (int s = 0; s < 2; ++s) {
        const void* const* q = ptrs + 9 * (s < nsides ? s : 0);
        ProjChainKSide& sd = p.side[s];
        sd.pre_scale = (const float*)q[0]; sd.pre_shift = (const float*)q[1];
        sd.wp = (const uint4*)q[2]; sd.bp = (const float*)q[3]; sd.skip = (const bf16_t*)q[4];
        sd.wn = (const uint4*)q[5]; sd.bn = (const float*)q[6];
        sd.out = (bf16_t*)q[7]; sd.out_next = (bf16_t*)q[8];
        sd.pre_relu = dims[6 + 2 * (s < nsides ? s : 0)];
        sd.skip_rows = dims[7 + 2 * (s < nsides ? s : 0)] > 0 ? dims[7 + 2 * (s < nsides ? s : 0)] : p.M;
        if (!sd.wp || !sd.wn || !sd.out_next) return COBEVT_ERR_ARG;
        if ((sd.pre_scale == nullptr) != (sd.pre_shift == nullptr)) return COBEVT_ERR_ARG;
        if (sd.skip_rows > p.M || p.M % sd.skip_rows) return COBEVT_ERR_SHAPE;
    }
    switch (p.K / 128) {
        case 2: launch<2>(p, nsides, stream); break;
        case 3: launch<3>(p, nsides, stream); break;
        default: launch<4>(p, nsides, stream); break;
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
