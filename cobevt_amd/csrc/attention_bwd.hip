// Backward of the gathered window / dilated-grid attention (fp32 storage, exact-fp32 MFMA): dQ, dK, dV (+ the gradient of the
// relative-position bias table) for cobevt_window_attention, with the probabilities recomputed from the saved log-sum-exp.
//
// First slice of the training path (the reference trains through torch autograd: the einsum -> softmax ->
// einsum of fax_modules.py:219-237 and swap_fusion_modules.py:100-121 under train_camera.py:143-179).  For one (window, head):
//     P = softmax(scale * Q K^T + bias (+ mask)),  O = P V
//     dV = P^T dO,   dP = dO V^T,   dZ = P o (dP - rowsum(dO o O)),   dQ = scale * dZ K,   dK = scale * dZ^T Q,   dbias = dZ
// Two kernels, neither of which adds into global memory per element (the first version did dQ with 16 fp32 atomics per lane
// and (query, key) tile - 5 x 10^8 of them on the LiDAR shape - and flushed the bias gradient once per workgroup: 9.1 ms there):
//
// attn_bwd_kv_kernel  (dK, dV, dbias): a workgroup owns one 32-key tile position of one head and walks a share of the WINDOWS;
//   per window its four waves walk the 32-query tiles.  With D = A.B on v_mfma_f32_32x32x2_f32 a lane owns one COLUMN of the
//   result: S = Q K^T and dP = dO V^T are computed with lane = key (the key's K / V rows are the B operands, in registers for the
//   whole window), P / dP / dZ live as [query row registers][key lane], and dV^T += dO^T P, dK^T += Q^T dZ contract over the query
//   rows with the P / dZ registers as B operands directly (step j pairs rows (j & 3) + 8 (j >> 2) + 4 half, on both operands).
//   dK / dV rows belong to this workgroup alone; the bias gradient is accumulated in an LDS copy of the table over ALL the
//   windows of the workgroup and flushed once.
// attn_bwd_q_kernel  (dQ): a workgroup owns one 32-query tile of one (window, head); its waves take the key tiles round robin.
//   Here lane = QUERY: S^T = K Q^T and dP^T = V dO^T take the query's Q / dO rows as register B operands, lse and
//   D = rowsum(dO o O) are per-lane scalars, and dQ^T += K^T dZ^T contracts over the keys = the register index, so dQ stays in
//   registers over the whole key loop and is stored once (the waves' partial sums meet in LDS).
// The softmax is recomputed in both (3 extra products over the minimal 5): 7 x 10^4 fp32 MFMAs are cheaper than the atomics.
// The partitions / reverses are the forward's index arithmetic (attn_common.hpp).
#include "attn_common.hpp"

namespace cobevt {

struct AttnBwdParams {
    AttnParams a;             // q, k, v, out (= O of the forward), maps, bias / mask, scale, lse (base-2, [B][L][heads][Nq])
    const float* dout;        // same layout as out
    float* dq;                // same layouts as q / k / v (ld and column offsets shared with the forward tensors)
    float* dk;
    float* dv;
    float* dbias;             // [bias_rows][heads], zero-initialised by the caller (nullable without bias)
    const float* dlse;        // nullable: gradient w.r.t. the NATURAL log-sum-exp of every query's logits, [B][L][heads][Nq] (an output of
                              // the forward that a caller combines further - the per-camera attentions of CVT's CrossAttention are merged
                              // by a softmax over their lse, cvt_modules.py:142-153).  d lse / d logit = P, so it enters as D - dlse.
    int nsplit;               // attn_bwd_kv_kernel: workgroups that share the windows of one (head, key tile position)
    int sb;                   // 1: q / k / v / out / dout / dq / dk / dv are bf16 in memory (attn_bwd_kv2_kernel / attn_bwd_q4_kernel only)
    float* dscr;              // nullable: [B][L][heads][Nq] scratch for D = rowsum(dO o O) - dlse: attn_bwd_q4_kernel (launched first) writes it,
                              // attn_bwd_kv2_kernel reads it instead of re-deriving it from the O tiles (a third of its staging loads)
};

namespace {

constexpr int kPad = 33;      // floats per row of the 32 x 32 LDS tiles
constexpr float kLog2eB = 1.4426950408889634f;
constexpr int kMaskedKey = (int)0x80000000;   // per-key info of a masked / out-of-range key

// ---- BF (bf16 matrix path, a bf16 autocast region: torch runs the two einsums and their gradients in bf16, the softmax in fp32 -
// train_camera.py:157-160): the tensors stay fp32 in memory, the tiles are rounded to bf16 as they are staged ([32 rows][32 d] with 80-byte
// rows) and every 32 x 32 x 32 product is two v_mfma_f32_32x32x16_bf16 instead of sixteen v_mfma_f32_32x32x2_f32.  Products whose
// contraction runs over the tile's ROWS take their A operand through ds_read_b64_tr_b16 (4 consecutive rows of a lane's column per read);
// the k-slots (half h, element e) of instruction m then stand for row 16 m + 8 (e / 4) + 4 h + e % 4 = acc_row(8 m + e, lane), i.e. the B
// operand is the lane's accumulator registers 8 m .. 8 m + 7 packed to bf16, unchanged.
constexpr int kRowB = 80;     // bytes per row of a bf16 tile
// four consecutive elements at element offset `off` of an fp32 (sb = 0) or bf16 (sb = 1) tensor
__device__ __forceinline__ float4 ld4q(const void* base, size_t off, int sb) {
    if (sb) {
        const uint2 u = *(const uint2*)((const uint16_t*)base + off);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    return *(const float4*)((const float*)base + off);
}
__device__ __forceinline__ void st4q(void* base, size_t off, int sb, const float4& v) {
    if (sb) *(uint2*)((uint16_t*)base + off) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    else *(float4*)((float*)base + off) = v;
}
typedef short v4s_b __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 tr_read_b(const unsigned char* lds) {
    const v4s_b r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((v4s_b __attribute__((address_space(3)))*)lds);
    return __builtin_bit_cast(uint2, r);
}
// accumulator registers 8 M .. 8 M + 7 as one bf16 operand (constant indices: an indexed copy of the vector went through scratch)
template <int M> __device__ __forceinline__ bf16x8 pack_acc8(const f32x16& v) {
    return __builtin_bit_cast(bf16x8, make_uint4(pack_bf2(v[8 * M], v[8 * M + 1]), pack_bf2(v[8 * M + 2], v[8 * M + 3]),
                                                  pack_bf2(v[8 * M + 4], v[8 * M + 5]), pack_bf2(v[8 * M + 6], v[8 * M + 7])));
}
__device__ __forceinline__ uint4 pack8f4(const float4& a, const float4& b) {
    return make_uint4(pack_bf2(a.x, a.y), pack_bf2(a.z, a.w), pack_bf2(b.x, b.y), pack_bf2(b.z, b.w));
}
// A operand of a product contracting over the rows of `tile` (bf16 [32][kRowB]): this lane's column = lane % 32, rows of instruction m
__device__ __forceinline__ bf16x8 tr_operand(const unsigned char* tile, int lane, int m) {
    const int g = lane >> 4, i = lane & 15;
    const unsigned char* a = tile + (16 * m + 4 * (g >> 1) + (i >> 2)) * kRowB + (16 * (g & 1) + 4 * (i & 3)) * 2;
    const uint2 lo = tr_read_b(a), hi = tr_read_b(a + 8 * kRowB);
    return __builtin_bit_cast(bf16x8, make_uint4(lo.x, lo.y, hi.x, hi.y));
}

__device__ __forceinline__ bool key_visible(const AttnParams& p, int b, int l, const TokCoord& kc) {
    if (p.kmap.mode == 2)
        return p.mask[((((size_t)b * p.L + l) * p.kmap.w1 + kc.i) * p.kmap.w2 + kc.j) * p.kmap.ncam + kc.cam] != 0.f;
    int ph, pw;
    tok_pixel(p.kmap, l, kc, ph, pw);
    return p.mask[(((size_t)b * p.kmap.HH + ph) * p.kmap.WW + pw) * p.kmap.ncam + kc.cam] != 0.f;
}

template <bool BIAS, bool MASK, bool BF>
__global__ __launch_bounds__(256) void attn_bwd_kv_kernel(AttnBwdParams bp) {
    const AttnParams& p = bp.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = (float*)smem_raw;
    constexpr int kWaveFloats = 2 * 32 * kPad + 3 * 32;
    float* red = smem + 4 * kWaveFloats;                // [3][16][64] cross-wave reduction of dK^T / dV^T
    float* bias_col = red + 3 * 16 * 64;                // [bias_rows] forward bias (x log2e)   (BIAS)
    float* dtab = bias_col + (BIAS ? p.bias_rows : 0);  // [bias_rows] gradient accumulator      (BIAS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, kt = blockIdx.y;
    const int head = blockIdx.x % p.heads, split = blockIdx.x / p.heads;
    float* Qs = smem + wave * kWaveFloats;              // per wave: Q and dO tiles [32][33], lse, D, bias query term [32]
    float* dOs = Qs + 32 * kPad;
    float* lse_s = dOs + 32 * kPad;
    float* D_s = lse_s + 32;
    int* qb_s = (int*)(D_s + 32);

    if (BIAS) {
        for (int i = tid; i < p.bias_rows; i += 256) {
            bias_col[i] = p.bias_table[(size_t)i * p.heads + head] * kLog2eB;
            dtab[i] = 0.f;
        }
    }
    const int tk = kt * 32 + ql;
    const bool k_in = tk < p.Nk;
    const float sl2 = p.scale * kLog2eB;
    const int nqt = (p.Nq + 31) / 32;
    const int nit = (nqt + 3) / 4;                      // uniform trip count: the loop contains workgroup barriers

    for (int l = split; l < p.L; l += bp.nsplit) {
        // ---- this lane's key: K / V rows as B operands (element 2 i + half of step i), mask, bias key term
        const TokCoord kc = tok_coord(p.kmap, k_in ? tk : 0);
        const size_t krow = tok_row(p.kmap, b, l, kc);
        bool k_ok = k_in;
        if (MASK && k_in) k_ok = key_visible(p, b, l, kc);
        const int kterm = BIAS ? rel_bias_key_term(p.kmap, kc) : 0;
        float kreg[16], vreg[16];
        uint4 kbf0 = make_uint4(0, 0, 0, 0), kbf1 = kbf0, vbf0 = kbf0, vbf1 = kbf0;   // BF: d = 16 m + 8 h .. + 7 of this key's K / V rows
        if constexpr (BF) {
            const float* kr = (const float*)p.k + krow * p.ldk + p.koff + head * 32 + 8 * h;
            const float* vr = (const float*)p.v + krow * p.ldv + p.voff + head * 32 + 8 * h;
            if (k_in) {
                kbf0 = pack8f4(*(const float4*)kr, *(const float4*)(kr + 4));
                kbf1 = pack8f4(*(const float4*)(kr + 16), *(const float4*)(kr + 20));
                vbf0 = pack8f4(*(const float4*)vr, *(const float4*)(vr + 4));
                vbf1 = pack8f4(*(const float4*)(vr + 16), *(const float4*)(vr + 20));
            }
        } else {
            const float* kr = (const float*)p.k + krow * p.ldk + p.koff + head * 32 + h;
            const float* vr = (const float*)p.v + krow * p.ldv + p.voff + head * 32 + h;
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                kreg[i] = k_in ? kr[2 * i] : 0.f;
                vreg[i] = k_in ? vr[2 * i] : 0.f;
            }
        }
        f32x16 dKT, dVT;
#pragma unroll
        for (int r = 0; r < 16; ++r) { dKT[r] = 0.f; dVT[r] = 0.f; }
        for (int it = 0; it < nit; ++it) {
            const int qt = it * 4 + wave;
            const bool t_ok = qt < nqt;
            __syncthreads();                            // the previous iteration's tile reads are done (and the tables are ready)
            // ---- stage the Q and dO tiles (two lanes per query row, 16 floats each), D = rowsum(dO o O), lse, bias query term
            {
                const int r = lane >> 1, half = lane & 1;
                const int tq = qt * 32 + r;
                const bool ok = t_ok && tq < p.Nq;
                const TokCoord qc = tok_coord(p.qmap, ok ? tq : 0);
                const size_t qrow = tok_row(p.qmap, b, l, qc);
                const size_t orow = tok_row(p.omap, b, l, qc);
                const float* qp = (const float*)p.q + qrow * p.ldq + p.qoff + head * 32 + half * 16;
                const float* op = (const float*)p.out + orow * p.ldo + p.ooff + head * 32 + half * 16;
                const float* dp = bp.dout + orow * p.ldo + p.ooff + head * 32 + half * 16;
                float dsum = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 qv = ok ? *(const float4*)(qp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 dv = ok ? *(const float4*)(dp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 ov = ok ? *(const float4*)(op + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                    if constexpr (BF) {
                        *(uint2*)((unsigned char*)Qs + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(qv.x, qv.y), pack_bf2(qv.z, qv.w));
                        *(uint2*)((unsigned char*)dOs + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(dv.x, dv.y), pack_bf2(dv.z, dv.w));
                    } else {
                        float* qd = Qs + r * kPad + half * 16 + 4 * c;
                        float* dd = dOs + r * kPad + half * 16 + 4 * c;
                        qd[0] = qv.x; qd[1] = qv.y; qd[2] = qv.z; qd[3] = qv.w;
                        dd[0] = dv.x; dd[1] = dv.y; dd[2] = dv.z; dd[3] = dv.w;
                    }
                    dsum += dv.x * ov.x + dv.y * ov.y + dv.z * ov.z + dv.w * ov.w;
                }
                dsum += __shfl_xor(dsum, 1, 64);
                if (half == 0) {
                    const size_t li = (((size_t)b * p.L + l) * p.heads + head) * p.Nq + (ok ? tq : 0);
                    D_s[r] = dsum - ((bp.dlse && ok) ? bp.dlse[li] : 0.f);
                    // invalid rows: lse = +inf -> P = exp2(-inf) = 0
                    lse_s[r] = ok ? p.lse[li] : INFINITY;
                    qb_s[r] = BIAS ? rel_bias_query_term(p.kmap, p.bias_L, qc) : 0;
                }
            }
            __syncthreads();
            // ---- S = Q K^T and dP = dO V^T  (lane = key column, register r <-> query row (r & 3) + 8 (r >> 2) + 4 h)
            f32x16 S, dP;
#pragma unroll
            for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
            if constexpr (BF) {
                const unsigned char* qa = (const unsigned char*)Qs + ql * kRowB + 16 * h;
                const unsigned char* da = (const unsigned char*)dOs + ql * kRowB + 16 * h;
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)qa), __builtin_bit_cast(bf16x8, kbf0), S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)da), __builtin_bit_cast(bf16x8, vbf0), dP, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(qa + 32)), __builtin_bit_cast(bf16x8, kbf1), S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(da + 32)), __builtin_bit_cast(bf16x8, vbf1), dP, 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    S = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[ql * kPad + 2 * i + h], kreg[i], S, 0, 0, 0);
                    dP = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[ql * kPad + 2 * i + h], vreg[i], dP, 0, 0, 0);
                }
            }
            // ---- P = exp2(z log2e - lse2),  dZ = P (dP - D)
            f32x16 P, dZ;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = acc_row(r, lane);
                float z = S[r] * sl2;
                int bidx = 0;
                if (BIAS) { bidx = qb_s[row] - kterm; z += bias_col[k_ok ? bidx : 0]; }
                const float pr = k_ok ? __builtin_amdgcn_exp2f(z - lse_s[row]) : 0.f;
                // dropout on the probabilities: O = (P o M') V, M' = keep / (1 - p)  ->  dV takes P o M', dZ = P o (M' o dP - D)
                float keep = 1.f;
                if (p.drop_p > 0.f) keep = attn_keep(p, b, l, head, (it * 4 + wave) * 32 + row, tk) ? 1.f / (1.f - p.drop_p) : 0.f;
                P[r] = pr * keep;
                dZ[r] = pr * (keep * dP[r] - D_s[row]);
                if (BIAS && k_ok && dZ[r] != 0.f) atomicAdd(&dtab[bidx], dZ[r]);
            }
            // ---- dV^T += dO^T P,  dK^T += Q^T dZ  (contraction over the query rows; step j pairs rows qj(0) and qj(1))
            if constexpr (BF) {
                dVT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)dOs, lane, 0), pack_acc8<0>(P), dVT, 0, 0, 0);
                dKT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)Qs, lane, 0), pack_acc8<0>(dZ), dKT, 0, 0, 0);
                dVT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)dOs, lane, 1), pack_acc8<1>(P), dVT, 0, 0, 0);
                dKT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)Qs, lane, 1), pack_acc8<1>(dZ), dKT, 0, 0, 0);
            } else {
#pragma unroll
                for (int j = 0; j < 16; ++j) {
                    const int qj = (j & 3) + 8 * (j >> 2) + 4 * h;
                    dVT = __builtin_amdgcn_mfma_f32_32x32x2f32(dOs[qj * kPad + ql], P[j], dVT, 0, 0, 0);
                    dKT = __builtin_amdgcn_mfma_f32_32x32x2f32(Qs[qj * kPad + ql], dZ[j], dKT, 0, 0, 0);
                }
            }
        }
        // ---- reduce dK^T / dV^T over the four waves, store (lane = key column, register r <-> dh row)
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = dKT[r];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dKT[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
        }
        __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = dVT[r];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) dVT[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
            if (k_in) {
                float* dkr = bp.dk + krow * p.ldk + p.koff + head * 32 + 4 * h;
                float* dvr = bp.dv + krow * p.ldv + p.voff + head * 32 + 4 * h;
#pragma unroll
                for (int g = 0; g < 4; ++g) {               // dh rows 8 g + 4 h .. + 3
                    *(float4*)(dkr + 8 * g) = make_float4(dKT[4 * g] * p.scale, dKT[4 * g + 1] * p.scale, dKT[4 * g + 2] * p.scale,
                                                          dKT[4 * g + 3] * p.scale);
                    *(float4*)(dvr + 8 * g) = make_float4(dVT[4 * g], dVT[4 * g + 1], dVT[4 * g + 2], dVT[4 * g + 3]);
                }
            }
        }
    }
    if (BIAS) {
        __syncthreads();
        for (int i = tid; i < p.bias_rows; i += 256) {
            const float g = dtab[i];
            if (g != 0.f) atomicAdd(bp.dbias + (size_t)i * p.heads + head, g);
        }
    }
}

// attn_bwd_kv_kernel<.., BF = true> with TWO 32-key tiles per workgroup.  On the bf16 matrix path the kernel is bound by its staging traffic:
// every key tile of a window reads all the window's Q / dO / O tiles (4 GB through L2 per level-0 launch, 350 us); with two key tiles
// sharing each staged query tile (and its four row-contracting A operands) that traffic halves.  Same arithmetic, same results.
struct KeyTileB {
    uint4 kb0, kb1, vb0, vb1;      // d = 16 m + 8 h .. + 7 of this lane's key, bf16
    size_t krow;
    int tk, kterm;
    bool k_in, k_ok;
    f32x16 dKT, dVT;
};

template <bool BIAS, bool MASK>
__global__ __launch_bounds__(256, 2) void attn_bwd_kv2_kernel(AttnBwdParams bp) {
    const AttnParams& p = bp.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = (float*)smem_raw;
    constexpr int kWaveFloats = 2 * 32 * kPad + 3 * 32;
    float* red = smem + 4 * kWaveFloats;                // [3][16][64] cross-wave reduction of dK^T / dV^T
    float* bias_col = red + 3 * 16 * 64;                // [bias_rows] forward bias (x log2e)   (BIAS)
    float* dtab = bias_col + (BIAS ? p.bias_rows : 0);  // [bias_rows] gradient accumulator      (BIAS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, kt2 = blockIdx.y;
    const int head = blockIdx.x % p.heads, split = blockIdx.x / p.heads;
    unsigned char* Qs = (unsigned char*)(smem + wave * kWaveFloats);        // per wave: Q and dO tiles bf16 [32][kRowB], lse, D, bias query term [32]
    unsigned char* dOs = Qs + 32 * kPad * 4;
    float* lse_s = (float*)(dOs + 32 * kPad * 4);
    float* D_s = lse_s + 32;
    int* qb_s = (int*)(D_s + 32);

    if (BIAS) {
        for (int i = tid; i < p.bias_rows; i += 256) {
            bias_col[i] = p.bias_table[(size_t)i * p.heads + head] * kLog2eB;
            dtab[i] = 0.f;
        }
    }
    const float sl2 = p.scale * kLog2eB;
    const int nqt = (p.Nq + 31) / 32;
    const int nit = (nqt + 3) / 4;                      // uniform trip count: the loop contains workgroup barriers
    KeyTileB A, Bt;
    A.tk = (kt2 * 2) * 32 + ql;
    Bt.tk = (kt2 * 2 + 1) * 32 + ql;
    A.k_in = A.tk < p.Nk;
    Bt.k_in = Bt.tk < p.Nk;

    for (int l = split; l < p.L; l += bp.nsplit) {
        auto load_key = [&](KeyTileB& T) __attribute__((always_inline)) {
            const TokCoord kc = tok_coord(p.kmap, T.k_in ? T.tk : 0);
            T.krow = tok_row(p.kmap, b, l, kc);
            T.k_ok = T.k_in;
            if (MASK && T.k_in) T.k_ok = key_visible(p, b, l, kc);
            T.kterm = BIAS ? rel_bias_key_term(p.kmap, kc) : 0;
            T.kb0 = T.kb1 = T.vb0 = T.vb1 = make_uint4(0, 0, 0, 0);
            if (T.k_in) {
                const size_t ko = T.krow * p.ldk + p.koff + head * 32 + 8 * h, vo = T.krow * p.ldv + p.voff + head * 32 + 8 * h;
                T.kb0 = pack8f4(ld4q(p.k, ko, bp.sb), ld4q(p.k, ko + 4, bp.sb));
                T.kb1 = pack8f4(ld4q(p.k, ko + 16, bp.sb), ld4q(p.k, ko + 20, bp.sb));
                T.vb0 = pack8f4(ld4q(p.v, vo, bp.sb), ld4q(p.v, vo + 4, bp.sb));
                T.vb1 = pack8f4(ld4q(p.v, vo + 16, bp.sb), ld4q(p.v, vo + 20, bp.sb));
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) { T.dKT[r] = 0.f; T.dVT[r] = 0.f; }
        };
        load_key(A);
        load_key(Bt);
        for (int it = 0; it < nit; ++it) {
            const int qt = it * 4 + wave;
            const bool t_ok = qt < nqt;
            __syncthreads();                            // the previous iteration's tile reads are done (and the tables are ready)
            // ---- stage the Q and dO tiles (two lanes per query row, 16 floats each), D = rowsum(dO o O), lse, bias query term
            {
                const int r = lane >> 1, half = lane & 1;
                const int tq = qt * 32 + r;
                const bool ok = t_ok && tq < p.Nq;
                const TokCoord qc = tok_coord(p.qmap, ok ? tq : 0);
                const size_t qrow = tok_row(p.qmap, b, l, qc);
                const size_t orow = tok_row(p.omap, b, l, qc);
                const size_t qo = qrow * p.ldq + p.qoff + head * 32 + half * 16, oo = orow * p.ldo + p.ooff + head * 32 + half * 16;
                float dsum = 0.f;
#pragma unroll
                for (int c = 0; c < 4; ++c) {
                    const float4 qv = ok ? ld4q(p.q, qo + 4 * c, bp.sb) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 dv = ok ? ld4q(bp.dout, oo + 4 * c, bp.sb) : make_float4(0.f, 0.f, 0.f, 0.f);
                    const float4 ov = (ok && !bp.dscr) ? ld4q(p.out, oo + 4 * c, bp.sb) : make_float4(0.f, 0.f, 0.f, 0.f);
                    *(uint2*)(Qs + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(qv.x, qv.y), pack_bf2(qv.z, qv.w));
                    *(uint2*)(dOs + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(dv.x, dv.y), pack_bf2(dv.z, dv.w));
                    dsum += dv.x * ov.x + dv.y * ov.y + dv.z * ov.z + dv.w * ov.w;
                }
                dsum += __shfl_xor(dsum, 1, 64);
                if (half == 0) {
                    const size_t li = (((size_t)b * p.L + l) * p.heads + head) * p.Nq + (ok ? tq : 0);
                    D_s[r] = bp.dscr ? (ok ? bp.dscr[li] : 0.f) : dsum - ((bp.dlse && ok) ? bp.dlse[li] : 0.f);
                    lse_s[r] = ok ? p.lse[li] : INFINITY;         // invalid rows: lse = +inf -> P = exp2(-inf) = 0
                    qb_s[r] = BIAS ? rel_bias_query_term(p.kmap, p.bias_L, qc) : 0;
                }
            }
            __syncthreads();
            // (the staged tile's LDS operands are read again per key tile: holding them across both cost a wave per SIMD in registers)
            auto tile_step = [&](KeyTileB& T) __attribute__((always_inline)) {
                // ---- S = Q K^T and dP = dO V^T  (lane = key column, register r <-> query row (r & 3) + 8 (r >> 2) + 4 h)
                f32x16 S, dP;
#pragma unroll
                for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
                const bf16x8 qa0 = __builtin_bit_cast(bf16x8, *(const uint4*)(Qs + ql * kRowB + 16 * h));
                const bf16x8 qa1 = __builtin_bit_cast(bf16x8, *(const uint4*)(Qs + ql * kRowB + 16 * h + 32));
                const bf16x8 da0 = __builtin_bit_cast(bf16x8, *(const uint4*)(dOs + ql * kRowB + 16 * h));
                const bf16x8 da1 = __builtin_bit_cast(bf16x8, *(const uint4*)(dOs + ql * kRowB + 16 * h + 32));
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa0, __builtin_bit_cast(bf16x8, T.kb0), S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da0, __builtin_bit_cast(bf16x8, T.vb0), dP, 0, 0, 0);
                S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(qa1, __builtin_bit_cast(bf16x8, T.kb1), S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(da1, __builtin_bit_cast(bf16x8, T.vb1), dP, 0, 0, 0);
                // ---- P = exp2(z log2e - lse2),  dZ = P (dP - D)
                f32x16 P, dZ;
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int row = acc_row(r, lane);
                    float z = S[r] * sl2;
                    int bidx = 0;
                    if (BIAS) { bidx = qb_s[row] - T.kterm; z += bias_col[T.k_ok ? bidx : 0]; }
                    const float pr = T.k_ok ? __builtin_amdgcn_exp2f(z - lse_s[row]) : 0.f;
                    float keep = 1.f;
                    if (p.drop_p > 0.f) keep = attn_keep(p, b, l, head, (it * 4 + wave) * 32 + row, T.tk) ? 1.f / (1.f - p.drop_p) : 0.f;
                    P[r] = pr * keep;
                    dZ[r] = pr * (keep * dP[r] - D_s[row]);
                    if (BIAS && T.k_ok && dZ[r] != 0.f) atomicAdd(&dtab[bidx], dZ[r]);
                }
                // ---- dV^T += dO^T P,  dK^T += Q^T dZ  (contraction over the query rows)
                T.dVT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(dOs, lane, 0), pack_acc8<0>(P), T.dVT, 0, 0, 0);
                T.dKT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(Qs, lane, 0), pack_acc8<0>(dZ), T.dKT, 0, 0, 0);
                T.dVT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(dOs, lane, 1), pack_acc8<1>(P), T.dVT, 0, 0, 0);
                T.dKT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(Qs, lane, 1), pack_acc8<1>(dZ), T.dKT, 0, 0, 0);
            };
            tile_step(A);
            tile_step(Bt);
        }
        // ---- reduce dK^T / dV^T over the four waves, store (lane = key column, register r <-> dh row)
        auto reduce_store = [&](KeyTileB& T) __attribute__((always_inline)) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = T.dKT[r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) T.dKT[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
            }
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = T.dVT[r];
            __syncthreads();
            if (wave == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) T.dVT[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
                if (T.k_in) {
                    const size_t ko = T.krow * p.ldk + p.koff + head * 32 + 4 * h, vo = T.krow * p.ldv + p.voff + head * 32 + 4 * h;
#pragma unroll
                    for (int g = 0; g < 4; ++g) {           // dh rows 8 g + 4 h .. + 3
                        st4q(bp.dk, ko + 8 * g, bp.sb, make_float4(T.dKT[4 * g] * p.scale, T.dKT[4 * g + 1] * p.scale, T.dKT[4 * g + 2] * p.scale,
                                                                    T.dKT[4 * g + 3] * p.scale));
                        st4q(bp.dv, vo + 8 * g, bp.sb, make_float4(T.dVT[4 * g], T.dVT[4 * g + 1], T.dVT[4 * g + 2], T.dVT[4 * g + 3]));
                    }
                }
            }
        };
        reduce_store(A);
        reduce_store(Bt);
    }
    if (BIAS) {
        __syncthreads();
        for (int i = tid; i < p.bias_rows; i += 256) {
            const float g = dtab[i];
            if (g != 0.f) atomicAdd(bp.dbias + (size_t)i * p.heads + head, g);
        }
    }
}

template <bool BIAS, bool MASK, bool BF>
__global__ __launch_bounds__(256) void attn_bwd_q_kernel(AttnBwdParams bp) {
    const AttnParams& p = bp.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    float* smem = (float*)smem_raw;
    constexpr int kWaveFloats = 2 * 32 * kPad + 32;
    float* red = smem + 4 * kWaveFloats;                // [3][16][64] cross-wave reduction of dQ^T
    float* bias_col = red + 3 * 16 * 64;                // [bias_rows] forward bias (x log2e)   (BIAS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, qt = blockIdx.y;
    const int l = blockIdx.x / p.heads, head = blockIdx.x - l * p.heads;
    float* Kt = smem + wave * kWaveFloats;              // per wave: K and V tiles [32][33], per-key info [32]
    float* Vt = Kt + 32 * kPad;
    int* kinfo_s = (int*)(Vt + 32 * kPad);              // bias key term, or kMaskedKey for a masked / out-of-range key

    if (BIAS) {
        for (int i = tid; i < p.bias_rows; i += 256) bias_col[i] = p.bias_table[(size_t)i * p.heads + head] * kLog2eB;
    }
    // ---- this lane's query: Q / dO rows as B operands, lse, D = rowsum(dO o O), bias query term
    const int tq = qt * 32 + ql;
    const bool q_ok = tq < p.Nq;
    const TokCoord qc = tok_coord(p.qmap, q_ok ? tq : 0);
    const size_t qrow = tok_row(p.qmap, b, l, qc);
    const size_t orow = tok_row(p.omap, b, l, qc);
    float qreg[16], doreg[16];
    uint4 qbf0 = make_uint4(0, 0, 0, 0), qbf1 = qbf0, dobf0 = qbf0, dobf1 = qbf0;   // BF: d = 16 m + 8 h .. + 7 of this query's Q / dO rows
    float Dq = 0.f;
    if constexpr (BF) {
        const float* qp = (const float*)p.q + qrow * p.ldq + p.qoff + head * 32 + 8 * h;
        const float* op = (const float*)p.out + orow * p.ldo + p.ooff + head * 32 + 8 * h;
        const float* dp = bp.dout + orow * p.ldo + p.ooff + head * 32 + 8 * h;
        if (q_ok) {
            const float4 d0 = *(const float4*)dp, d1 = *(const float4*)(dp + 4), d2 = *(const float4*)(dp + 16), d3 = *(const float4*)(dp + 20);
            const float4 o0 = *(const float4*)op, o1 = *(const float4*)(op + 4), o2 = *(const float4*)(op + 16), o3 = *(const float4*)(op + 20);
            qbf0 = pack8f4(*(const float4*)qp, *(const float4*)(qp + 4));
            qbf1 = pack8f4(*(const float4*)(qp + 16), *(const float4*)(qp + 20));
            dobf0 = pack8f4(d0, d1);
            dobf1 = pack8f4(d2, d3);
            Dq = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w
               + d2.x * o2.x + d2.y * o2.y + d2.z * o2.z + d2.w * o2.w + d3.x * o3.x + d3.y * o3.y + d3.z * o3.z + d3.w * o3.w;
        }
        Dq += __shfl_xor(Dq, 32, 64);
    } else {
        const float* qp = (const float*)p.q + qrow * p.ldq + p.qoff + head * 32 + h;
        const float* op = (const float*)p.out + orow * p.ldo + p.ooff + head * 32 + h;
        const float* dp = bp.dout + orow * p.ldo + p.ooff + head * 32 + h;
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            qreg[i] = q_ok ? qp[2 * i] : 0.f;
            doreg[i] = q_ok ? dp[2 * i] : 0.f;
            Dq += q_ok ? doreg[i] * op[2 * i] : 0.f;
        }
        Dq += __shfl_xor(Dq, 32, 64);
    }
    const size_t lse_i = (((size_t)b * p.L + l) * p.heads + head) * p.Nq + (q_ok ? tq : 0);
    if (bp.dlse && q_ok) Dq -= bp.dlse[lse_i];
    const float lse_q = q_ok ? p.lse[lse_i] : INFINITY;
    const int qterm = BIAS ? rel_bias_query_term(p.kmap, p.bias_L, qc) : 0;
    const float sl2 = p.scale * kLog2eB;

    f32x16 dQT;
#pragma unroll
    for (int r = 0; r < 16; ++r) dQT[r] = 0.f;
    const int nkt = (p.Nk + 31) / 32;
    const int nit = (nkt + 3) / 4;
    for (int it = 0; it < nit; ++it) {
        const int kt = it * 4 + wave;
        const bool t_ok = kt < nkt;
        __syncthreads();
        {   // stage this wave's key tile: two lanes per key row, 16 floats each
            const int r = lane >> 1, half = lane & 1;
            const int tk = kt * 32 + r;
            const bool ok = t_ok && tk < p.Nk;
            const TokCoord kc = tok_coord(p.kmap, ok ? tk : 0);
            const size_t krow = tok_row(p.kmap, b, l, kc);
            const float* kp = (const float*)p.k + krow * p.ldk + p.koff + head * 32 + half * 16;
            const float* vp = (const float*)p.v + krow * p.ldv + p.voff + head * 32 + half * 16;
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                const float4 kv = ok ? *(const float4*)(kp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                const float4 vv = ok ? *(const float4*)(vp + 4 * c) : make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (BF) {
                    *(uint2*)((unsigned char*)Kt + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(kv.x, kv.y), pack_bf2(kv.z, kv.w));
                    *(uint2*)((unsigned char*)Vt + r * kRowB + (half * 16 + 4 * c) * 2) = make_uint2(pack_bf2(vv.x, vv.y), pack_bf2(vv.z, vv.w));
                } else {
                    float* kd = Kt + r * kPad + half * 16 + 4 * c;
                    float* vd = Vt + r * kPad + half * 16 + 4 * c;
                    kd[0] = kv.x; kd[1] = kv.y; kd[2] = kv.z; kd[3] = kv.w;
                    vd[0] = vv.x; vd[1] = vv.y; vd[2] = vv.z; vd[3] = vv.w;
                }
            }
            if (half == 0) {
                bool vis = ok;
                if (MASK && ok) vis = key_visible(p, b, l, kc);
                kinfo_s[r] = vis ? (BIAS ? rel_bias_key_term(p.kmap, kc) : 0) : kMaskedKey;
            }
        }
        __syncthreads();
        // ---- S^T = K Q^T and dP^T = V dO^T  (lane = query column, register r <-> key row (r & 3) + 8 (r >> 2) + 4 h)
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
        if constexpr (BF) {
            const unsigned char* ka = (const unsigned char*)Kt + ql * kRowB + 16 * h;
            const unsigned char* va = (const unsigned char*)Vt + ql * kRowB + 16 * h;
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)ka), __builtin_bit_cast(bf16x8, qbf0), S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)va), __builtin_bit_cast(bf16x8, dobf0), dP, 0, 0, 0);
            S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(ka + 32)), __builtin_bit_cast(bf16x8, qbf1), S, 0, 0, 0);
            dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(va + 32)), __builtin_bit_cast(bf16x8, dobf1), dP, 0, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                S = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[ql * kPad + 2 * i + h], qreg[i], S, 0, 0, 0);
                dP = __builtin_amdgcn_mfma_f32_32x32x2f32(Vt[ql * kPad + 2 * i + h], doreg[i], dP, 0, 0, 0);
            }
        }
        f32x16 dZ;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kinfo = kinfo_s[acc_row(r, lane)];
            const bool vis = kinfo != kMaskedKey;
            float z = S[r] * sl2;
            if (BIAS) z += bias_col[vis ? qterm - kinfo : 0];
            const float pr = vis ? __builtin_amdgcn_exp2f(z - lse_q) : 0.f;
            float keep = 1.f;
            if (p.drop_p > 0.f) keep = attn_keep(p, b, l, head, tq, kt * 32 + acc_row(r, lane)) ? 1.f / (1.f - p.drop_p) : 0.f;
            dZ[r] = pr * (keep * dP[r] - Dq);
        }
        // ---- dQ^T += K^T dZ^T  (contraction over the keys = the register index; step j pairs key rows kj(0) and kj(1))
        if constexpr (BF) {
            dQT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)Kt, lane, 0), pack_acc8<0>(dZ), dQT, 0, 0, 0);
            dQT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand((const unsigned char*)Kt, lane, 1), pack_acc8<1>(dZ), dQT, 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int kj = (j & 3) + 8 * (j >> 2) + 4 * h;
                dQT = __builtin_amdgcn_mfma_f32_32x32x2f32(Kt[kj * kPad + ql], dZ[j], dQT, 0, 0, 0);
            }
        }
    }
    // ---- the waves' partial sums meet in LDS; lane = query column, register r <-> dh row
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = dQT[r];
    __syncthreads();
    if (wave == 0 && q_ok) {
#pragma unroll
        for (int r = 0; r < 16; ++r) dQT[r] += red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
        float* dqr = bp.dq + qrow * p.ldq + p.qoff + head * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *(float4*)(dqr + 8 * g) = make_float4(dQT[4 * g] * p.scale, dQT[4 * g + 1] * p.scale, dQT[4 * g + 2] * p.scale,
                                                  dQT[4 * g + 3] * p.scale);
    }
}

// attn_bwd_q_kernel<.., BF = true> with FOUR query tiles per workgroup, one per wave, all on the same key tiles: the K / V tile is staged
// once per workgroup (every thread copies one float4 of each, rounded to bf16; the next tile's loads are issued before this tile's products:
// two LDS buffers, one barrier per key tile) instead of once per wave and query tile, and a wave keeps its dQ^T to itself - no cross-wave
// reduction.  Level 0 of the 5-agent frame was 40,960 workgroups of 48 MFMAs each behind 64 KB of staging loads.
template <bool BIAS, bool MASK>
__global__ __launch_bounds__(256) void attn_bwd_q4_kernel(AttnBwdParams bp) {
    const AttnParams& p = bp.a;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int kTile = 32 * kRowB;                   // one bf16 tile
    unsigned char* tiles = smem_raw;                    // [2 buffers][K, V][kTile]
    int* kinfo_s = (int*)(smem_raw + 4 * kTile);        // [2][32]: bias key term, or kMaskedKey
    float* bias_col = (float*)(kinfo_s + 64);           // [bias_rows] forward bias (x log2e)   (BIAS)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.z, qt = blockIdx.y * 4 + wave;
    const int l = blockIdx.x / p.heads, head = blockIdx.x - l * p.heads;
    if (BIAS) {
        for (int i = tid; i < p.bias_rows; i += 256) bias_col[i] = p.bias_table[(size_t)i * p.heads + head] * kLog2eB;
    }
    // ---- this lane's query: Q / dO rows as B operands, lse, D = rowsum(dO o O), bias query term
    const int tq = qt * 32 + ql;
    const bool q_ok = tq < p.Nq;
    const TokCoord qc = tok_coord(p.qmap, q_ok ? tq : 0);
    const size_t qrow = tok_row(p.qmap, b, l, qc);
    const size_t orow = tok_row(p.omap, b, l, qc);
    uint4 qbf0 = make_uint4(0, 0, 0, 0), qbf1 = qbf0, dobf0 = qbf0, dobf1 = qbf0;
    float Dq = 0.f;
    if (q_ok) {
        const size_t qo = qrow * p.ldq + p.qoff + head * 32 + 8 * h, oo = orow * p.ldo + p.ooff + head * 32 + 8 * h;
        const float4 d0 = ld4q(bp.dout, oo, bp.sb), d1 = ld4q(bp.dout, oo + 4, bp.sb), d2 = ld4q(bp.dout, oo + 16, bp.sb), d3 = ld4q(bp.dout, oo + 20, bp.sb);
        const float4 o0 = ld4q(p.out, oo, bp.sb), o1 = ld4q(p.out, oo + 4, bp.sb), o2 = ld4q(p.out, oo + 16, bp.sb), o3 = ld4q(p.out, oo + 20, bp.sb);
        qbf0 = pack8f4(ld4q(p.q, qo, bp.sb), ld4q(p.q, qo + 4, bp.sb));
        qbf1 = pack8f4(ld4q(p.q, qo + 16, bp.sb), ld4q(p.q, qo + 20, bp.sb));
        dobf0 = pack8f4(d0, d1);
        dobf1 = pack8f4(d2, d3);
        Dq = d0.x * o0.x + d0.y * o0.y + d0.z * o0.z + d0.w * o0.w + d1.x * o1.x + d1.y * o1.y + d1.z * o1.z + d1.w * o1.w
           + d2.x * o2.x + d2.y * o2.y + d2.z * o2.z + d2.w * o2.w + d3.x * o3.x + d3.y * o3.y + d3.z * o3.z + d3.w * o3.w;
    }
    Dq += __shfl_xor(Dq, 32, 64);
    const size_t lse_i = (((size_t)b * p.L + l) * p.heads + head) * p.Nq + (q_ok ? tq : 0);
    if (bp.dlse && q_ok) Dq -= bp.dlse[lse_i];
    if (bp.dscr && q_ok && h == 0) bp.dscr[lse_i] = Dq;            // for attn_bwd_kv2_kernel, launched behind this kernel
    const float lse_q = q_ok ? p.lse[lse_i] : INFINITY;
    const int qterm = BIAS ? rel_bias_query_term(p.kmap, p.bias_L, qc) : 0;
    const float sl2 = p.scale * kLog2eB;

    // staging: thread -> key row tid / 8, float4 (tid % 8) of its K and V rows; the row's visibility / bias term by the first of the 8
    const int sr = tid >> 3, sc = tid & 7;
    float4 nk = make_float4(0.f, 0.f, 0.f, 0.f), nv = nk;
    int ninfo = kMaskedKey;
    auto fetch = [&](int kt) __attribute__((always_inline)) {
        const int tk = kt * 32 + sr;
        const bool ok = tk < p.Nk;
        const TokCoord kc = tok_coord(p.kmap, ok ? tk : 0);
        const size_t krow = tok_row(p.kmap, b, l, kc);
        const float4 kv = ld4q(p.k, krow * p.ldk + p.koff + head * 32 + 4 * sc, bp.sb);
        const float4 vv = ld4q(p.v, krow * p.ldv + p.voff + head * 32 + 4 * sc, bp.sb);
        nk = ok ? kv : make_float4(0.f, 0.f, 0.f, 0.f);
        nv = ok ? vv : make_float4(0.f, 0.f, 0.f, 0.f);
        bool vis = ok;
        if (MASK && ok && sc == 0) vis = key_visible(p, b, l, kc);
        ninfo = vis ? (BIAS ? rel_bias_key_term(p.kmap, kc) : 0) : kMaskedKey;
    };
    auto stage = [&](int buf) __attribute__((always_inline)) {
        unsigned char* kt_ = tiles + buf * 2 * kTile;
        *(uint2*)(kt_ + sr * kRowB + sc * 8) = make_uint2(pack_bf2(nk.x, nk.y), pack_bf2(nk.z, nk.w));
        *(uint2*)(kt_ + kTile + sr * kRowB + sc * 8) = make_uint2(pack_bf2(nv.x, nv.y), pack_bf2(nv.z, nv.w));
        if (sc == 0) kinfo_s[buf * 32 + sr] = ninfo;
    };
    f32x16 dQT;
#pragma unroll
    for (int r = 0; r < 16; ++r) dQT[r] = 0.f;
    const int nkt = (p.Nk + 31) / 32;
    fetch(0);
    for (int kt = 0; kt < nkt; ++kt) {
        const int buf = kt & 1;
        stage(buf);
        __syncthreads();                                // tile kt (and the bias column) is in LDS; the other buffer is free
        if (kt + 1 < nkt) fetch(kt + 1);
        const unsigned char* Kt = tiles + buf * 2 * kTile;
        const unsigned char* Vt = Kt + kTile;
        // ---- S^T = K Q^T and dP^T = V dO^T  (lane = query column, register r <-> key row (r & 3) + 8 (r >> 2) + 4 h)
        f32x16 S, dP;
#pragma unroll
        for (int r = 0; r < 16; ++r) { S[r] = 0.f; dP[r] = 0.f; }
        const unsigned char* ka = Kt + ql * kRowB + 16 * h;
        const unsigned char* va = Vt + ql * kRowB + 16 * h;
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)ka), __builtin_bit_cast(bf16x8, qbf0), S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)va), __builtin_bit_cast(bf16x8, dobf0), dP, 0, 0, 0);
        S = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(ka + 32)), __builtin_bit_cast(bf16x8, qbf1), S, 0, 0, 0);
        dP = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, *(const uint4*)(va + 32)), __builtin_bit_cast(bf16x8, dobf1), dP, 0, 0, 0);
        f32x16 dZ;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int kinfo = kinfo_s[buf * 32 + acc_row(r, lane)];
            const bool vis = kinfo != kMaskedKey;
            float z = S[r] * sl2;
            if (BIAS) z += bias_col[vis ? qterm - kinfo : 0];
            const float pr = vis ? __builtin_amdgcn_exp2f(z - lse_q) : 0.f;
            float keep = 1.f;
            if (p.drop_p > 0.f) keep = attn_keep(p, b, l, head, tq, kt * 32 + acc_row(r, lane)) ? 1.f / (1.f - p.drop_p) : 0.f;
            dZ[r] = pr * (keep * dP[r] - Dq);
        }
        // ---- dQ^T += K^T dZ^T  (contraction over the keys = the register index)
        dQT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(Kt, lane, 0), pack_acc8<0>(dZ), dQT, 0, 0, 0);
        dQT = __builtin_amdgcn_mfma_f32_32x32x16_bf16(tr_operand(Kt, lane, 1), pack_acc8<1>(dZ), dQT, 0, 0, 0);
    }
    if (q_ok) {                                         // lane = query column, register r <-> dh row
        const size_t qo = qrow * p.ldq + p.qoff + head * 32 + 4 * h;
#pragma unroll
        for (int g = 0; g < 4; ++g)
            st4q(bp.dq, qo + 8 * g, bp.sb, make_float4(dQT[4 * g] * p.scale, dQT[4 * g + 1] * p.scale, dQT[4 * g + 2] * p.scale, dQT[4 * g + 3] * p.scale));
    }
}

template <typename K>
static void set_max_lds(K kernel) {
    (void)hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_window_attention_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                           const void* dout, const float* dlse, void* dq, void* dk, void* dv, float* dbias,
                                           const float* bias_table, const float* mask, const int* dims, float scale,
                                           float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, hipStream_t stream) {
    if (!q || !k || !v || !out || !lse || !dout || !dq || !dk || !dv || !dims) return COBEVT_ERR_ARG;
    if (drop_p < 0.f || drop_p >= 1.f) return COBEVT_ERR_ARG;
    if (dlse && drop_p > 0.f) return COBEVT_ERR_UNSUPPORTED;
    AttnBwdParams bp;
    bp.dlse = dlse;
    AttnParams& p = bp.a;
    const int dtype = dims[0] & 0xff;
    const bool bfmm = (dims[0] & 0x100) != 0;               // + 0x100: the products on the bf16 matrix path (bf16 autocast regions)
    // fp32 storage (the parity / training mode); bf16 storage of q / k / v / out / dout / dq / dk / dv only on the bf16 matrix path
    if (dtype != 1 && !(dtype == 0 && bfmm && (dims[0] & 0x200) == 0)) return COBEVT_ERR_UNSUPPORTED;
    bp.sb = dtype == 0;
    // + 0x400: `lse` is [2][B][L][heads][Nq] - the second half is scratch for D (dQ kernel writes, dK / dV kernel reads; bf16 matrix path only)
    bp.dscr = nullptr;
    p.q = q; p.k = k; p.v = v; p.out = const_cast<void*>(out);
    p.B = dims[1]; p.L = dims[2]; p.heads = dims[3];
    p.ldq = dims[4]; p.ldk = dims[5]; p.ldv = dims[6]; p.ldo = dims[7];
    p.qoff = dims[8]; p.koff = dims[9]; p.voff = dims[10]; p.ooff = dims[11];
    p.bias_mode = dims[12]; p.bias_rows = dims[13]; p.bias_L = dims[14];
    p.mean_q = dims[15];
    p.qmap = read_map(dims + 16); p.kmap = read_map(dims + 24); p.omap = read_map(dims + 32);
    p.bias_table = bias_table; p.mask = mask; p.scale = scale; p.lse = const_cast<float*>(lse); p.klinear = 0;
    p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_seed_dev = drop_seed_dev;
    bp.dout = (const float*)dout; bp.dq = (float*)dq; bp.dk = (float*)dk; bp.dv = (float*)dv; bp.dbias = dbias;
    if (!map_ok(p.qmap) || !map_ok(p.kmap) || !map_ok(p.omap)) return COBEVT_ERR_SHAPE;
    if (p.B < 1 || p.heads < 1 || p.L != p.qmap.X * p.qmap.Y || p.L != p.kmap.X * p.kmap.Y) return COBEVT_ERR_SHAPE;
    if (p.mean_q || p.omap.ncam != p.qmap.ncam) return COBEVT_ERR_UNSUPPORTED;   // (camera mean: done outside the kernel when training)
    if (p.bias_mode && (!bias_table || !dbias || p.bias_rows < 1 || p.bias_L < 1)) return COBEVT_ERR_ARG;
    if ((p.ldq | p.ldk | p.ldv | p.ldo | p.qoff | p.koff | p.voff | p.ooff) % 4) return COBEVT_ERR_SHAPE;
    if (bp.sb && p.drop_p > 0.f) return COBEVT_ERR_UNSUPPORTED;        // (the dropout forward is fp32 storage)
    p.Nq = p.qmap.ncam * p.qmap.w1 * p.qmap.w2;
    p.Nk = p.kmap.ncam * p.kmap.w1 * p.kmap.w2;
    const int nkt = (p.Nk + 31) / 32, nqt = (p.Nq + 31) / 32;
    if (nkt > 65535 || nqt > 65535 || p.B > 65535) return COBEVT_ERR_SHAPE;
    // dK / dV / dbias: about 2048 workgroups in all; the windows of one (head, key tile position) are shared by nsplit of them
    long per_window = (long)p.heads * nkt * p.B;
    int nsplit = (int)((2048 + per_window - 1) / per_window);
    nsplit = nsplit < 1 ? 1 : (nsplit > p.L ? p.L : nsplit);
    bp.nsplit = nsplit;
    const size_t lds_kv = (size_t)(4 * (2 * 32 * kPad + 3 * 32) + 3 * 16 * 64) * 4 + (p.bias_mode ? (size_t)p.bias_rows * 8 : 0);
    const size_t lds_q = (size_t)(4 * (2 * 32 * kPad + 32) + 3 * 16 * 64) * 4 + (p.bias_mode ? (size_t)p.bias_rows * 4 : 0);
    if (lds_kv > 160 * 1024) return COBEVT_ERR_UNSUPPORTED;
    const dim3 grid_kv(p.heads * nsplit, nkt, p.B), grid_q(p.L * p.heads, nqt, p.B), block(256);
    // bf16 matrix path: two key tiles per workgroup (halves the staging traffic that bounds it), window shares re-balanced to ~2048 workgroups
    const bool kv2 = bfmm && (nkt >= 2 || bp.sb) && (dims[0] & 0x200) == 0;     // (+ 0x200: one key tile per workgroup, for A/B runs)
    const int nkt2 = (nkt + 1) / 2;
    AttnBwdParams bp2 = bp;
    if (kv2 && (dims[0] & 0x400)) bp2.dscr = const_cast<float*>(lse) + (size_t)p.B * p.L * p.heads * p.Nq;
    {
        const long per2 = (long)p.heads * nkt2 * p.B;
        int ns2 = (int)((2048 + per2 - 1) / per2);
        bp2.nsplit = ns2 < 1 ? 1 : (ns2 > p.L ? p.L : ns2);
    }
    const dim3 grid_kv2(p.heads * bp2.nsplit, nkt2, p.B);
    const dim3 grid_q4(p.L * p.heads, (nqt + 3) / 4, p.B);           // ... and four query tiles per workgroup in the dQ kernel
    const size_t lds_q4 = (size_t)4 * 32 * kRowB + 64 * 4 + (p.bias_mode ? (size_t)p.bias_rows * 4 : 0);
    const bool hb = p.bias_mode != 0, hm = mask != nullptr;
#define COBEVT_BWD_LAUNCH2(B_, M_, F_)                                                                \
    do {                                                                                              \
        static cobevt::PerDeviceOnce attr;                                                                   \
        if (attr.first()) { set_max_lds(attn_bwd_kv_kernel<B_, M_, F_>); set_max_lds(attn_bwd_q_kernel<B_, M_, F_>); set_max_lds(attn_bwd_kv2_kernel<B_, M_>); set_max_lds(attn_bwd_q4_kernel<B_, M_>); } \
        if (F_ && kv2) {                                                                              \
            hipLaunchKernelGGL((attn_bwd_q4_kernel<B_, M_>), grid_q4, block, lds_q4, stream, bp2);    \
            hipLaunchKernelGGL((attn_bwd_kv2_kernel<B_, M_>), grid_kv2, block, lds_kv, stream, bp2);  \
        } else {                                                                                      \
            hipLaunchKernelGGL((attn_bwd_kv_kernel<B_, M_, F_>), grid_kv, block, lds_kv, stream, bp); \
            hipLaunchKernelGGL((attn_bwd_q_kernel<B_, M_, F_>), grid_q, block, lds_q, stream, bp);    \
        }                                                                                             \
    } while (0)
#define COBEVT_BWD_LAUNCH(B_, M_) do { if (bfmm) COBEVT_BWD_LAUNCH2(B_, M_, true); else COBEVT_BWD_LAUNCH2(B_, M_, false); } while (0)
    if (hb && hm) COBEVT_BWD_LAUNCH(true, true);
    else if (hb) COBEVT_BWD_LAUNCH(true, false);
    else if (hm) COBEVT_BWD_LAUNCH(false, true);
    else COBEVT_BWD_LAUNCH(false, false);
#undef COBEVT_BWD_LAUNCH
#undef COBEVT_BWD_LAUNCH2
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
