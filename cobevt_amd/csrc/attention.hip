// Gathered window / dilated-grid attention on MFMA (gfx950): QK^T -> online softmax -> PV in one kernel,
// with the einops window<->grid partition / reverse done as index arithmetic on token rows (no permute
// round trips through HBM).
//
// Replaces, for already-projected q/k/v token matrices:
//   * CrossWinAttention core          reference: opv2v/opencood/models/sub_modules/fax_modules.py:211-237,243
//     (window partition :399-404, grid partition :417-424, window reverse :409,433, camera mean :243)
//   * swap-fusion Attention core      reference: opv2v/opencood/models/fusion_modules/swap_fusion_modules.py:93-123
//     (3-D relative position bias :55-85,106-107; key mask :110-115; window/grid partition :172-190)
//   * FAX global self-attention core  reference: fax_modules.py:137-171 (2-D relative position bias :121-130,157-158)
//
// Work decomposition: one workgroup = (batch b, window l, head m, query tile).  All waves of the
// workgroup share the K / V^T tiles of that window+head staged in LDS (64 keys per step).  A wave owns 32
// query tokens.  In "mean" mode (level-0 cross attention, per-camera queries) wave w handles camera w of
// the same 32 BEV positions and the workgroup averages the waves' outputs in LDS (z.mean(1), :243;
// the out-projection is linear so the mean commutes with it).
//
// MFMA mapping (32x32 tiles, dh = 32):  S^T = K.Q^T  (A = K rows from LDS, B = Q rows held in registers)
// so lane (q = lane&31, half h) holds 16 of the 32 key scores of its query -> softmax reductions are
// lane-local plus one xor-32 exchange.  O^T += V^T.P^T uses the score registers directly as the B
// operand; the key <-> k-slot permutation this implies is applied identically when reading V^T from LDS.
#include "attn_common.hpp"

namespace cobevt {

// KT = keys per staged tile: 64, or 128 (bf16, key counts >= 256).  The tile loop is one global round trip per iteration (the
// next tile is requested when the current one's MFMAs start), so on the small grids that take this kernel - the level-2 / global
// attentions with 1024 keys, the fusion windows with 320 - the iteration count, not the arithmetic, sets the time.
template <typename T, int KT> struct AttnLds {
    static constexpr int kKeysPerTile = KT;
    static constexpr int kKRow = 32 * Elem<T>::kBytes + 16;              // K tile row: 32 dh + pad
    static constexpr int kVRow = kKeysPerTile * Elem<T>::kBytes + (Elem<T>::kIsBf16 ? 8 : 16);  // V^T row: 64 keys + pad
    static constexpr int kKBytes = kKeysPerTile * kKRow;
    static constexpr int kVBytes = 32 * kVRow;
    static constexpr int kInfoBytes = kKeysPerTile * 4;
    static constexpr int kBuf = kKBytes + kVBytes + kInfoBytes;          // one K/V^T/info tile
    static constexpr int kFixed = 2 * kBuf;                              // double buffered
};

// BIAS / MASK are compile-time so the plain cross-attention path carries no per-element metadata work.
// DROP (fp32 training forward only): dropout on the probabilities - the softmax denominator sums every exponential, the PV
// product takes the kept ones scaled by 1 / (1 - p).
template <typename T, bool BIAS, bool MASK, int KT, bool DROP = false>
__global__ __launch_bounds__(512) void attn_gather_kernel(AttnParams p) {
    using L = AttnLds<T, KT>;
    constexpr int kKeysPerTile = KT;
    constexpr int NS = KT / 32;                    // 32-key sub-tiles per tile
    constexpr int CH = Elem<T>::kChunk;
    constexpr int NG = 32 * Elem<T>::kBytes / 32;  // 32-byte k-groups along dh (2 bf16, 4 fp32)
    constexpr int CPR = 32 / CH;                   // 16-byte chunks per K token row (head slice)
    constexpr int K_ITEMS = kKeysPerTile * CPR;    // 256 bf16 / 512 fp32
    constexpr int K_IT = K_ITEMS / 256;
    // V^T staging: bf16 pairs two consecutive keys so every LDS write is a full dword (key pair x 4 dh per item);
    // fp32 writes one dword per element (key x 4 dh per item)
    constexpr int V_ITEMS = Elem<T>::kIsBf16 ? (kKeysPerTile / 2) * 8 : kKeysPerTile * 8;
    constexpr int V_IT = V_ITEMS / 256;
    constexpr bool INFO = BIAS || MASK;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* bias_col = (float*)(smem + L::kFixed);
    // per-key (row index, cam << 16 | i << 8 | j) of this window, computed ONCE per workgroup: the token -> row arithmetic
    // is two runtime integer divisions (~50 VALU instructions each) and used to be redone by every staging item of every
    // key tile - twelve times per key, ~330 integer VALU instructions per tile against ~250 for the softmax itself
    int2* ktab = (int2*)(smem + L::kFixed + (BIAS ? ((p.bias_rows * 4 + 15) & ~15) : 0));
    float* red = (float*)smem;  // aliases the K/V tiles after the key loop (mean mode)

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, nthr = blockDim.x;
    const int h = lane >> 5, ql = lane & 31;
    // grid = (windows*heads, query tiles, batch): the query-tile blocks of one (window, head) are gridDim.x apart in
    // dispatch order, i.e. (for the usual multiple-of-8 window*head counts) on the SAME XCD, so its L2 serves their
    // shared K/V instead of 8 XCDs each fetching it from HBM (rocprofv3 FETCH_SIZE was 3x the algorithmic bytes).
    const int b = blockIdx.z;
    const int l = blockIdx.x / p.heads, head = blockIdx.x - l * p.heads;
    const int nqt = gridDim.y / p.ksplit;                 // (ksplit = 1: every launch but the key-split one)
    const int split = blockIdx.y / nqt, qtile = blockIdx.y - split * nqt;

    // ---- this lane's query token (mean mode: wave = camera; waves beyond ncam only help staging)
    const int P = p.qmap.w1 * p.qmap.w2;
    int tq;
    bool q_ok;
    const bool pair = p.mean_q == 2;      // CVT: the key tile's camera selects the query copy (one softmax over all cameras)
    if (p.mean_q == 1) { const int pos = qtile * 32 + ql; q_ok = pos < P && wave < p.qmap.ncam; tq = wave * P + pos; }
    else { tq = qtile * 32 * (nthr >> 6) + wave * 32 + ql; q_ok = tq < (pair ? P : p.Nq); }
    const TokCoord qc = tok_coord(p.qmap, q_ok ? tq : 0);          // pair mode: tq < P -> camera 0

    // Split-bf16 library, fp32 storage (round 6): the K tile and the V^T tile are written to LDS already split into (hi, lo) bf16 halves
    // (common.hpp stage_x_piece: a staged tile is read by every query tile of the workgroup) and the query fragments are split once per
    // query tile, so a score MFMA pair has no conversion in front of it and a PV pair only the split of its probabilities - the in-loop
    // form split all four operands per 16-byte piece: ~380 of the ~500 VALU instructions of a 64-key tile against 32 MFMAs.
    // Third library (round 6, "fp32_fast" routes the attention launches to it): K / V^T staged as fp16 (hi, lo) pairs, the query fragments
    // and the probabilities as ONE fp16 term - a single fp16 MFMA per piece, four conversions per probability piece.  tests/precision_emul.py
    // mode fp16_qp: rounding queries and probabilities to fp16 changes the 5-agent frame's logit error in the third digit (2.6-2.9e-4).
    constexpr bool kStage = (COBEVT_F32_SPLIT != 0) && !Elem<T>::kIsBf16;
    constexpr bool kStage16 = (COBEVT_F32_SPLIT == 2) && !Elem<T>::kIsBf16;
    auto dup_split = [](const uint4& x, uint4& hh, uint4& ll) {           // {x0..x3} -> {hi01, hi23, hi01, hi23}, {lo01, lo23, lo01, lo23}
        if constexpr (kStage16) { hh = dup_f16_piece(x); ll = hh; return; }
        uint32_t h01, h23, l01, l23;
        split_bf16_pair(__uint_as_float(x.x), __uint_as_float(x.y), h01, l01);
        split_bf16_pair(__uint_as_float(x.z), __uint_as_float(x.w), h23, l23);
        hh = make_uint4(h01, h23, h01, h23);
        ll = make_uint4(l01, l23, l01, l23);
    };
    auto mfma_staged_pair = [](const uint4& a_staged, const uint4& b_hh, const uint4& b_ll, f32x16& acc) {
        if constexpr (kStage16) {       // b_hh = the duplicated fp16 form {b, b / 2^s} (common.hpp dup_f16_piece); b_ll unused
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a_staged), __builtin_bit_cast(f16x8, b_hh), acc, 0, 0, 0);
        } else {
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_staged), __builtin_bit_cast(bf16x8, b_hh), acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a_staged), __builtin_bit_cast(bf16x8, b_ll), acc, 0, 0, 0);
        }
    };
    uint4 qf[NG];
    uint4 qhh[kStage ? NG : 1], qll[kStage ? NG : 1];
    auto load_q = [&](int cam) {
        TokCoord c = qc;
        c.cam += cam;
        const T* qrow = (const T*)p.q + tok_row(p.qmap, b, l, c) * p.ldq + p.qoff + head * 32;
#pragma unroll
        for (int g = 0; g < NG; ++g)
            qf[g] = q_ok ? *(const uint4*)(qrow + g * (2 * CH) + h * CH) : make_uint4(0, 0, 0, 0);
        if constexpr (kStage) {
#pragma unroll
            for (int g = 0; g < NG; ++g) dup_split(qf[g], qhh[g], qll[g]);
        }
    };
    load_q(0);
    int q_cam = 0;
    const int keys_per_cam = p.kmap.w1 * p.kmap.w2;
    if (BIAS) {
        // in the base-2 softmax domain already.  Eight strided loads per thread in flight at a time: the rolled loop paid one
        // global round trip per iteration (8 of them for the 2025-row 3-D table) before the first key tile could start
        constexpr int BI = 8;
        for (int base = 0; base < p.bias_rows; base += nthr * BI) {
            float tv[BI];
#pragma unroll
            for (int u = 0; u < BI; ++u) {
                const int i = base + u * nthr + tid;
                tv[u] = p.bias_table[(size_t)(i < p.bias_rows ? i : p.bias_rows - 1) * p.heads + head];
            }
#pragma unroll
            for (int u = 0; u < BI; ++u) {
                const int i = base + u * nthr + tid;
                if (i < p.bias_rows) bias_col[i] = tv[u] * 1.4426950408889634f;
            }
        }
    }
    const bool klin = p.klinear != 0;      // one window = the whole key map in row-major order: row = b * Nk + tk, no table
    if (!klin) {
        for (int tk = tid; tk < p.Nk; tk += nthr) {
            const TokCoord kc = tok_coord(p.kmap, tk);
            ktab[tk] = make_int2((int)tok_row(p.kmap, b, l, kc), (kc.cam << 16) | (kc.i << 8) | kc.j);
        }
    }
    const int klin_base = b * p.Nk;
    __syncthreads();

    // ---- staging registers (threads 0..255 stage; K_IT / V_IT items each)
    uint4 kreg[K_IT];
    uint4 vreg[V_IT];          // bf16: {row0 lo, row0 hi, row1 lo, row1 hi} (8 bytes of 2 key rows); fp32: 16 bytes of 1 row
    int ireg[K_IT];
    const bool stager = tid < 256;
    // key tiles: 64 consecutive key tokens; in pair mode every camera's keys start a new tile (a tile never mixes cameras), so
    // tile kt covers tokens [tile_base, tile_base + tile_valid) with tile_valid <= 64
    const int tiles_per_cam = (keys_per_cam + kKeysPerTile - 1) / kKeysPerTile;
    const int nkt = pair ? p.kmap.ncam * tiles_per_cam : (p.Nk + kKeysPerTile - 1) / kKeysPerTile;
    auto tile_base = [&](int kt) {
        if (!pair) return kt * kKeysPerTile;
        const int cam = kt / tiles_per_cam;
        return cam * keys_per_cam + (kt - cam * tiles_per_cam) * kKeysPerTile;
    };
    auto tile_valid = [&](int kt) {
        if (!pair) return p.Nk - kt * kKeysPerTile;
        return keys_per_cam - (kt % tiles_per_cam) * kKeysPerTile;
    };

    auto load_tile = [&](int kt) {
        if (!stager) return;
        const int tbase = tile_base(kt), tvalid = tile_valid(kt);
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int item = tid + it * 256;
            const int kk = item / CPR, cj = item - kk * CPR;
            const int tk = tbase + kk;
            const bool ok = kk < tvalid;
            const int2 ke = klin ? make_int2(klin_base + (ok ? tk : 0), 0) : ktab[ok ? tk : 0];
            const size_t row = (size_t)ke.x;
            TokCoord kc;
            kc.cam = ke.y >> 16; kc.i = (ke.y >> 8) & 0xff; kc.j = ke.y & 0xff;
            kreg[it] = ok ? *(const uint4*)((const T*)p.k + row * p.ldk + p.koff + head * 32 + cj * CH) : make_uint4(0, 0, 0, 0);
            if (INFO) {
                int info = -1;
                if (ok && cj == 0) {
                    bool valid = true;
                    if (MASK) {
                        if (p.kmap.mode == 2) {  // mask stored partitioned like the keys: (B, X*Y, w1, w2, ncam)
                            valid = p.mask[((((size_t)b * p.L + l) * p.kmap.w1 + kc.i) * p.kmap.w2 + kc.j) * p.kmap.ncam + kc.cam] != 0.f;
                        } else {
                            int ph, pw;
                            tok_pixel(p.kmap, l, kc, ph, pw);
                            valid = p.mask[(((size_t)b * p.kmap.HH + ph) * p.kmap.WW + pw) * p.kmap.ncam + kc.cam] != 0.f;
                        }
                    }
                    // the key's share of the relative-position index (the table index is linear in the coordinates:
                    // index = query term - key term), -1 = masked out
                    if (valid) info = BIAS ? rel_bias_key_term(p.kmap, kc) : 0;
                }
                ireg[it] = info;
            }
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int item = tid + it * 256;
            if constexpr (Elem<T>::kIsBf16) {
                const int kp = item >> 3, dq = item & 7;               // key pair, dh quad
                uint2 r0 = make_uint2(0, 0), r1 = make_uint2(0, 0);
                const int tk = tbase + 2 * kp;
                if (2 * kp < tvalid) {
                    const size_t row = (size_t)(klin ? klin_base + tk : ktab[tk].x);
                    r0 = *(const uint2*)((const T*)p.v + row * p.ldv + p.voff + head * 32 + dq * 4);
                }
                if (2 * kp + 1 < tvalid) {
                    const size_t row = (size_t)(klin ? klin_base + tk + 1 : ktab[tk + 1].x);
                    r1 = *(const uint2*)((const T*)p.v + row * p.ldv + p.voff + head * 32 + dq * 4);
                }
                vreg[it] = make_uint4(r0.x, r0.y, r1.x, r1.y);
            } else {
                const int kk = item >> 3, dq = item & 7;
                const int tk = tbase + kk;
                uint4 v = make_uint4(0, 0, 0, 0);
                if (kk < tvalid) {
                    const size_t row = (size_t)(klin ? klin_base + tk : ktab[tk].x);
                    v = *(const uint4*)((const T*)p.v + row * p.ldv + p.voff + head * 32 + dq * 4);
                }
                vreg[it] = v;
            }
        }
    };
    auto store_tile = [&](int buf) {
        if (!stager) return;
        unsigned char* Ks = smem + buf * L::kBuf;
        unsigned char* Vts = Ks + L::kKBytes;
        int* kinfo = (int*)(Vts + L::kVBytes);
#pragma unroll
        for (int it = 0; it < K_IT; ++it) {
            const int item = tid + it * 256;
            const int kk = item / CPR, cj = item - kk * CPR;
            if constexpr (kStage) *(uint4*)(Ks + kk * L::kKRow + cj * 16) = stage_x_piece<T>(kreg[it]);
            else *(uint4*)(Ks + kk * L::kKRow + cj * 16) = kreg[it];
            if (INFO && cj == 0) kinfo[kk] = ireg[it];
        }
#pragma unroll
        for (int it = 0; it < V_IT; ++it) {
            const int item = tid + it * 256;
            if constexpr (Elem<T>::kIsBf16) {
                const int kp = item >> 3, dq = item & 7;
                const uint32_t a[2] = {vreg[it].x, vreg[it].y}, c[2] = {vreg[it].z, vreg[it].w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {   // dh = dq*4 + e : {key 2kp, key 2kp+1} as one dword
                    const uint32_t lo = (a[e >> 1] >> ((e & 1) * 16)) & 0xffffu, hi = (c[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                    *(uint32_t*)(Vts + (dq * 4 + e) * L::kVRow + kp * 4) = lo | (hi << 16);
                }
            } else {
                const int kk = item >> 3, dq = item & 7;
                const uint32_t w[4] = {vreg[it].x, vreg[it].y, vreg[it].z, vreg[it].w};
                if constexpr (kStage) {
                    // the 16-byte operand piece of V^T row dh holds keys 4j .. 4j + 3 as {hi(k0,k1), hi(k2,k3), lo(k0,k1), lo(k2,k3)}:
                    // this item's key is slot kk & 3 of piece kk >> 2 -> one 16-bit write into the hi half, one into the lo half
                    uint32_t hi2[2], lo2[2];
                    split_pair_staged(__uint_as_float(w[0]), __uint_as_float(w[1]), hi2[0], lo2[0]);
                    split_pair_staged(__uint_as_float(w[2]), __uint_as_float(w[3]), hi2[1], lo2[1]);
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        unsigned char* d = Vts + (dq * 4 + e) * L::kVRow + (kk >> 2) * 16 + (kk & 3) * 2;
                        *(uint16_t*)d = (uint16_t)(hi2[e >> 1] >> ((e & 1) * 16));
                        *(uint16_t*)(d + 8) = (uint16_t)(lo2[e >> 1] >> ((e & 1) * 16));
                    }
                } else {
#pragma unroll
                for (int e = 0; e < 4; ++e) *(uint32_t*)(Vts + (dq * 4 + e) * L::kVRow + kk * 4) = w[e];
                }
            }
        }
    };

    f32x16 ot;
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] = 0.f;
    float m_run = -INFINITY, l_run = 0.f;
    const float sl2 = p.scale * 1.4426950408889634f;  // softmax in base 2
    // this lane's query term of the relative-position index (swap_fusion_modules.py:55-85, fax_modules.py:121-130):
    // ((dl + L-1)(2 w1 - 1) + (di + w1-1))(2 w2 - 1) + (dj + w2-1) with d = query - key coordinate
    const int bias_q = BIAS ? rel_bias_query_term(p.kmap, p.bias_L, qc) : 0;

    // key split: this workgroup walks tiles [kt0, kt1) of the window
    const int kt0 = (int)((long)split * nkt / p.ksplit), kt1 = (int)((long)(split + 1) * nkt / p.ksplit);
    load_tile(kt0);
    store_tile(0);
    __syncthreads();
    for (int kt = kt0; kt < kt1; ++kt) {
        const int buf = (kt - kt0) & 1;
        if (pair) {                                   // (keys per camera is a multiple of the tile: one camera per tile)
            const int cam = kt / tiles_per_cam;
            if (cam != q_cam) { q_cam = cam; load_q(cam); }
        }
        if (kt + 1 < kt1) load_tile(kt + 1);
        const unsigned char* Ks = smem + buf * L::kBuf;
        const unsigned char* Vts = Ks + L::kKBytes;
        const int* kinfo = (const int*)(Vts + L::kVBytes);

        // ---- S^T = K . Q^T for the two 32-key sub-tiles
        f32x16 st[NS];
#pragma unroll
        for (int s = 0; s < NS; ++s) {
#pragma unroll
            for (int r = 0; r < 16; ++r) st[s][r] = 0.f;
#pragma unroll
            for (int g = 0; g < NG; ++g) {
                const uint4 a = *(const uint4*)(Ks + (s * 32 + ql) * L::kKRow + g * 32 + h * 16);
                if constexpr (kStage) mfma_staged_pair(a, qhh[g], qll[g], st[s]);
                else mfma_kgroup<T>(a, qf[g], st[s]);
            }
        }
        // ---- softmax numerators in the base-2 domain: p = 2^(s*scale*log2e [+ bias*log2e] - m)
        float mloc = -INFINITY;
        const int nvalid = tile_valid(kt);               // >= 64 except in the last tile (of a camera, in pair mode)
        if (INFO) {
#pragma unroll
            for (int s = 0; s < NS; ++s) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int info = kinfo[s * 32 + acc_row(r, lane)];
                    float v;
                    if (BIAS) {
                        const int idx = info < 0 ? 0 : bias_q - info;
                        v = fmaf(st[s][r], sl2, bias_col[idx]);
                    } else {
                        v = st[s][r] * sl2;
                    }
                    v = info < 0 ? -INFINITY : v;
                    st[s][r] = v;
                    mloc = fmaxf(mloc, v);
                }
            }
        } else {
            if (nvalid < kKeysPerTile) {                 // ragged last tile (wave-uniform branch)
#pragma unroll
                for (int s = 0; s < NS; ++s)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        if (s * 32 + acc_row(r, lane) >= nvalid) st[s][r] = -INFINITY;
            }
#pragma unroll
            for (int s = 0; s < NS; ++s)
#pragma unroll
                for (int r = 0; r < 16; ++r) mloc = fmaxf(mloc, st[s][r]);
            mloc *= sl2;                                 // scale > 0: max commutes with the scaling
        }
        const float mtile = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float m_new = fmaxf(m_run, mtile);
        const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
        const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);
        m_run = m_new;
        float psum = 0.f;
#pragma unroll
        for (int s = 0; s < NS; ++s)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float e = INFO ? __builtin_amdgcn_exp2f(st[s][r] - m_safe)
                                     : __builtin_amdgcn_exp2f(fmaf(st[s][r], sl2, -m_safe));
                psum += e;
                if (DROP) {
                    const int tk = tile_base(kt) + s * 32 + acc_row(r, lane);
                    st[s][r] = attn_keep(p, b, l, head, tq, tk) ? e * (1.f / (1.f - p.drop_p)) : 0.f;
                } else {
                    st[s][r] = e;
                }
            }
        l_run = l_run * alpha + psum;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] *= alpha;

        // ---- O^T += V^T . P^T
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            if constexpr (Elem<T>::kIsBf16) {
#pragma unroll
                for (int u = 0; u < 2; ++u) {
                    uint4 pb;
                    pb.x = pack_bf2(st[s][8 * u + 0], st[s][8 * u + 1]);
                    pb.y = pack_bf2(st[s][8 * u + 2], st[s][8 * u + 3]);
                    pb.z = pack_bf2(st[s][8 * u + 4], st[s][8 * u + 5]);
                    pb.w = pack_bf2(st[s][8 * u + 6], st[s][8 * u + 7]);
                    const unsigned char* vr = Vts + ql * L::kVRow + (s * 32 + 16 * u + 4 * h) * 2;
                    const uint2 lo = *(const uint2*)vr, hi = *(const uint2*)(vr + 16);
                    mfma_kgroup<T>(make_uint4(lo.x, lo.y, hi.x, hi.y), pb, ot);
                }
            } else {
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    uint4 pb;
                    pb.x = __float_as_uint(st[s][4 * u + 0]); pb.y = __float_as_uint(st[s][4 * u + 1]);
                    pb.z = __float_as_uint(st[s][4 * u + 2]); pb.w = __float_as_uint(st[s][4 * u + 3]);
                    const uint4 a = *(const uint4*)(Vts + ql * L::kVRow + (s * 32 + 8 * u + 4 * h) * 4);
                    if constexpr (kStage) {
                        uint4 phh, pll;
                        dup_split(pb, phh, pll);
                        mfma_staged_pair(a, phh, pll, ot);
                    } else {
                        mfma_kgroup<T>(a, pb, ot);
                    }
                }
            }
        }
        if (kt + 1 < kt1) store_tile(buf ^ 1);
        __syncthreads();
    }

    const float l_tot = l_run + __shfl_xor(l_run, 32, 64);
    const float inv = 1.0f / l_tot;  // an all-masked row yields NaN like the reference softmax
    if (p.lse && q_ok && h == 0)     // training forward: what the backward kernel needs to recompute the probabilities
        p.lse[(((size_t)b * p.L + l) * p.heads + head) * p.Nq + tq] = m_run + __builtin_amdgcn_logf(l_tot);
#pragma unroll
    for (int r = 0; r < 16; ++r) ot[r] *= inv;

    if (p.ksplit > 1) {           // partial result of this key range: normalised rows + their log-sum-exp, merged afterwards
        if (!q_ok) return;
        const size_t orow_i = tok_row(p.omap, b, l, qc);
        const int d = p.heads * 32;
        if (h == 0) p.part_lse[((size_t)split * p.part_rows + orow_i) * p.heads + head] = m_run + __builtin_amdgcn_logf(l_tot);
        T* orow = (T*)p.part_out + ((size_t)split * p.part_rows + orow_i) * d + head * 32;
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {
            const int d0 = 8 * g4 + 4 * h;
            if constexpr (Elem<T>::kIsBf16) {
                uint2 w;
                w.x = pack_bf2(ot[4 * g4 + 0], ot[4 * g4 + 1]);
                w.y = pack_bf2(ot[4 * g4 + 2], ot[4 * g4 + 3]);
                *(uint2*)(orow + d0) = w;
            } else {
                *(float4*)(orow + d0) = make_float4(ot[4 * g4 + 0], ot[4 * g4 + 1], ot[4 * g4 + 2], ot[4 * g4 + 3]);
            }
        }
        return;
    }
    if (p.mean_q == 1) {
        const int nw = p.qmap.ncam;
        if (wave < nw) {
#pragma unroll
            for (int r = 0; r < 16; ++r) red[(wave * 16 + r) * 64 + lane] = ot[r];
        }
        __syncthreads();
        if (wave != 0) return;
        const float invn = 1.0f / (float)nw;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sacc = 0.f;
            for (int w = 0; w < nw; ++w) sacc += red[(w * 16 + r) * 64 + lane];
            ot[r] = sacc * invn;
        }
    }
    if (!q_ok) return;
    TokCoord oc = qc;
    if (p.mean_q) oc.cam = 0;
    T* orow = (T*)p.out + tok_row(p.omap, b, l, oc) * p.ldo + p.ooff + head * 32;
#pragma unroll
    for (int g4 = 0; g4 < 4; ++g4) {
        const int d0 = 8 * g4 + 4 * h;
        if constexpr (Elem<T>::kIsBf16) {
            uint2 w;
            w.x = pack_bf2(ot[4 * g4 + 0], ot[4 * g4 + 1]);
            w.y = pack_bf2(ot[4 * g4 + 2], ot[4 * g4 + 3]);
            *(uint2*)(orow + d0) = w;
        } else {
            *(float4*)(orow + d0) = make_float4(ot[4 * g4 + 0], ot[4 * g4 + 1], ot[4 * g4 + 2], ot[4 * g4 + 3]);
        }
    }
}

// out[row][head*32 + c] = sum_s w_s part_out[s][row][..] / sum_s w_s,  w_s = 2^(lse_s - max_s lse_s); a split whose keys were all
// masked (lse = -inf, rows NaN) carries weight 0 and is skipped.  One thread per (row, head, 8 channels).
template <typename T>
__global__ __launch_bounds__(256) void attn_ksplit_merge_kernel(AttnParams p) {
    const int d = p.heads * 32;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = p.part_rows * p.heads * 4;
    if (item >= total) return;
    const int c8 = (int)(item & 3), head = (int)((item >> 2) % p.heads);
    const long row = (item >> 2) / p.heads;
    float lse[8], mx = -INFINITY;
    for (int s = 0; s < p.ksplit; ++s) {
        lse[s] = p.part_lse[((size_t)s * p.part_rows + row) * p.heads + head];
        if (lse[s] > mx) mx = lse[s];
    }
    float acc[8], wsum = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    for (int s = 0; s < p.ksplit; ++s) {
        if (!(lse[s] > -INFINITY)) continue;
        const float w = __builtin_amdgcn_exp2f(lse[s] - mx);
        wsum += w;
        const T* src = (const T*)p.part_out + ((size_t)s * p.part_rows + row) * d + head * 32 + c8 * 8;
        float v[8];
        if constexpr (Elem<T>::kIsBf16) {
            chunk_to_f32<T>(*(const uint4*)src, v);
        } else {
            chunk_to_f32<T>(*(const uint4*)src, v);
            chunk_to_f32<T>(*(const uint4*)(src + 4), v + 4);
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += w * v[e];
    }
    const float inv = 1.0f / wsum;                    // every split masked: NaN, like the reference softmax
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] *= inv;
    T* dst = (T*)p.out + (size_t)row * p.ldo + p.ooff + head * 32 + c8 * 8;
    if constexpr (Elem<T>::kIsBf16) {
        *(uint4*)dst = f32_to_chunk<T>(acc);
    } else {
        *(uint4*)dst = f32_to_chunk<T>(acc);
        *(uint4*)(dst + 4) = f32_to_chunk<T>(acc + 4);
    }
}

// Debug / test entry points: the token -> row map and the relative-position index exactly as the attention kernels compute them
__global__ void attn_index_dump_kernel(TokMap m, int B, int L, int ntok, int* rows) {
    const long total = (long)B * L * ntok;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int t = (int)(i % ntok);
        const int l = (int)((i / ntok) % L);
        const int b = (int)(i / ((long)ntok * L));
        rows[i] = (int)tok_row(m, b, l, tok_coord(m, t));
    }
}

__global__ void attn_bias_index_dump_kernel(TokMap qm, TokMap km, int bias_L, int nq, int nk, int* idx) {
    const long total = (long)nq * nk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tq = (int)(i / nk), tk = (int)(i % nk);
        idx[i] = rel_bias_query_term(km, bias_L, tok_coord(qm, tq)) - rel_bias_key_term(km, tok_coord(km, tk));
    }
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
static int window_attention_impl(const void* q, const void* k, const void* v, void* out, float* lse,
                                 const float* bias_table, const float* mask, const int* dims, float scale,
                                 float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, hipStream_t stream,
                                 int ksplit = 1, void* part_out = nullptr, float* part_lse = nullptr, long part_rows = 0) {
    // dims: [dtype, B, L, heads, ldq, ldk, ldv, ldo, qoff, koff, voff, ooff, bias_mode, bias_rows, bias_L,
    //        mean_q, qmap[8], kmap[8], omap[8]]
    if (!q || !k || !v || !out || !dims) return COBEVT_ERR_ARG;
    AttnParams p;
    // dims[0] = dtype | variant << 8 | query split << 16: variant 0 = automatic (K/V-resident kernel where it applies), 1 = force
    // the streaming kernel (A/B runs, parity tests of both paths); query split 0 = automatic
    const int dtype = dims[0] & 0xff, variant = (dims[0] >> 8) & 0xff, qsplit_hint = (dims[0] >> 16) & 0xff;
    p.q = q; p.k = k; p.v = v; p.out = out;
    p.B = dims[1]; p.L = dims[2]; p.heads = dims[3];
    p.ldq = dims[4]; p.ldk = dims[5]; p.ldv = dims[6]; p.ldo = dims[7];
    p.qoff = dims[8]; p.koff = dims[9]; p.voff = dims[10]; p.ooff = dims[11];
    p.bias_mode = dims[12]; p.bias_rows = dims[13]; p.bias_L = dims[14];
    p.mean_q = dims[15];
    p.qmap = read_map(dims + 16); p.kmap = read_map(dims + 24); p.omap = read_map(dims + 32);
    p.bias_table = bias_table; p.mask = mask; p.scale = scale; p.lse = lse;
    p.drop_p = drop_p; p.drop_seed = drop_seed; p.drop_seed_dev = drop_seed_dev;
    p.ksplit = 1; p.part_out = nullptr; p.part_lse = nullptr; p.part_rows = 0;
    if (drop_p < 0.f || drop_p >= 1.f || (drop_p > 0.f && (!lse || dtype != 1))) return COBEVT_ERR_ARG;   // dropout: training forward only
    if (dtype != 0 && dtype != 1) return COBEVT_ERR_ARG;
    if (!map_ok(p.qmap) || !map_ok(p.kmap) || !map_ok(p.omap)) return COBEVT_ERR_SHAPE;
    if (p.B < 1 || p.heads < 1 || p.L != p.qmap.X * p.qmap.Y || p.L != p.kmap.X * p.kmap.Y) return COBEVT_ERR_SHAPE;
    if (p.bias_mode && (!bias_table || p.bias_rows < 1 || p.bias_L < 1)) return COBEVT_ERR_ARG;
    const int ch = dtype == 0 ? 8 : 4;
    if ((p.ldq | p.ldk | p.ldv | p.ldo | p.qoff | p.koff | p.voff | p.ooff) % ch) return COBEVT_ERR_SHAPE;
    p.Nq = p.qmap.ncam * p.qmap.w1 * p.qmap.w2;
    p.Nk = p.kmap.ncam * p.kmap.w1 * p.kmap.w2;
    if (p.mean_q < 0 || p.mean_q > 2) return COBEVT_ERR_ARG;
    if (p.mean_q && p.qmap.ncam == 1) p.mean_q = 0;
    if (p.mean_q == 1 && (p.qmap.ncam > 8 || p.omap.ncam != 1)) return COBEVT_ERR_UNSUPPORTED;
    if (p.mean_q == 2) {     // camera-paired queries: cameras on both sides, whole tiles per camera, no bias / mask
        if (p.omap.ncam != 1 || p.qmap.ncam != p.kmap.ncam || p.bias_mode || mask) return COBEVT_ERR_UNSUPPORTED;
    }
    // keys of a single window that covers the whole map are rows b * Nk + tk: no table (CVT attends to 4 x 64 x 64 keys)
    p.klinear = (p.kmap.mode != 2 && p.kmap.X == 1 && p.kmap.Y == 1 && !p.bias_mode && !mask) ? 1 : 0;
    if (lse && p.mean_q) return COBEVT_ERR_UNSUPPORTED;     // (the training path averages cameras outside the kernel)
    if (ksplit > 1) {       // key split: streaming kernel only, plain inference attention, every query of a window on its own
        if (ksplit > 8 || !part_out || !part_lse || part_rows < 1 || lse || p.mean_q || drop_p > 0.f) return COBEVT_ERR_ARG;
        if (p.omap.ncam != p.qmap.ncam) return COBEVT_ERR_UNSUPPORTED;
        p.ksplit = ksplit; p.part_out = part_out; p.part_lse = part_lse; p.part_rows = part_rows;
    }
    if (dtype == 0 && variant == 0 && p.mean_q != 2 && !lse && ksplit == 1) {
        const int rc = launch_attn_resident(p, qsplit_hint, stream);
        if (rc >= 0) return rc;
    }
    const int P = p.qmap.w1 * p.qmap.w2;
    dim3 grid, block;
    if (p.mean_q == 1) { block = dim3(64 * (p.qmap.ncam < 4 ? 4 : p.qmap.ncam)); grid = dim3(p.L * p.heads, (P + 31) / 32, p.B); }
    else { block = dim3(256); grid = dim3(p.L * p.heads, ((p.mean_q == 2 ? P : p.Nq) + 127) / 128 * p.ksplit, p.B); }
    if (grid.y > 65535 || grid.z > 65535) return COBEVT_ERR_SHAPE;
    // 128-key tiles: bf16, enough keys, not the camera-paired mode (its tiles never mix cameras)
    // ... and a grid that does not fill the chip anyway (there the iteration count sets the time; on a full grid the wider tile's
    // registers cost occupancy: 512-token LiDAR windows, bias + mask, 8192 workgroups: 357 us against 266 us with 64-key tiles)
    const bool wide = dtype == 0 && p.mean_q != 2 && p.Nk >= 256 && variant != 2 &&      // variant 2: 64-key tiles (A/B)
                      (long)grid.x * grid.y * grid.z <= 1024;
    if (p.ksplit > 1 && (p.Nk + (wide ? 127 : 63)) / (wide ? 128 : 64) < p.ksplit) return COBEVT_ERR_SHAPE;   // >= 1 tile per split
    size_t lds = dtype == 0 ? (wide ? AttnLds<bf16_t, 128>::kFixed : AttnLds<bf16_t, 64>::kFixed) : AttnLds<float, 64>::kFixed;
    if (p.bias_mode) lds += ((size_t)p.bias_rows * 4 + 15) & ~(size_t)15;
    if (!p.klinear) lds += (size_t)p.Nk * 8;        // per-key row / coordinate table
    if ((long)p.B * p.kmap.ncam * (p.kmap.mode == 2 ? (long)p.L * p.kmap.w1 * p.kmap.w2 : (long)p.kmap.HH * p.kmap.WW) >= 0x7fffffffL)
        return COBEVT_ERR_UNSUPPORTED;              // the table holds 32-bit row indices
    if (p.mean_q == 1) { const size_t need = (size_t)p.qmap.ncam * 16 * 64 * 4; if (need > lds) lds = need; }
    if (lds > 64 * 1024) return COBEVT_ERR_UNSUPPORTED;
    const bool hb = p.bias_mode != 0, hm = p.mask != nullptr;
#define COBEVT_ATTN_LAUNCH(TT, KT_)                                                                                  \
    do {                                                                                                              \
        if (hb && hm) hipLaunchKernelGGL((attn_gather_kernel<TT, true, true, KT_>), grid, block, lds, stream, p);     \
        else if (hb) hipLaunchKernelGGL((attn_gather_kernel<TT, true, false, KT_>), grid, block, lds, stream, p);     \
        else if (hm) hipLaunchKernelGGL((attn_gather_kernel<TT, false, true, KT_>), grid, block, lds, stream, p);     \
        else hipLaunchKernelGGL((attn_gather_kernel<TT, false, false, KT_>), grid, block, lds, stream, p);            \
    } while (0)
    if (p.drop_p > 0.f) {                          // training forward (fp32, lse): checked by the caller
        if (hb && hm) hipLaunchKernelGGL((attn_gather_kernel<float, true, true, 64, true>), grid, block, lds, stream, p);
        else if (hb) hipLaunchKernelGGL((attn_gather_kernel<float, true, false, 64, true>), grid, block, lds, stream, p);
        else if (hm) hipLaunchKernelGGL((attn_gather_kernel<float, false, true, 64, true>), grid, block, lds, stream, p);
        else hipLaunchKernelGGL((attn_gather_kernel<float, false, false, 64, true>), grid, block, lds, stream, p);
    } else if (dtype == 0 && wide) COBEVT_ATTN_LAUNCH(bf16_t, 128);
    else if (dtype == 0) COBEVT_ATTN_LAUNCH(bf16_t, 64);
    else COBEVT_ATTN_LAUNCH(float, 64);
#undef COBEVT_ATTN_LAUNCH
    if (p.ksplit > 1) {
        const long items = p.part_rows * p.heads * 4;
        const dim3 mg((unsigned)((items + 255) / 256));
        if (dtype == 0) hipLaunchKernelGGL(attn_ksplit_merge_kernel<bf16_t>, mg, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL(attn_ksplit_merge_kernel<float>, mg, dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// Key-split form of cobevt_window_attention, see include/cobevt_hip.h
extern "C" int cobevt_window_attention_ksplit(const void* q, const void* k, const void* v, void* out, const float* bias_table,
                                              const float* mask, void* part_out, float* part_lse, const int* dims, float scale,
                                              int ksplit, long out_rows, hipStream_t stream) {
    if (ksplit < 2) return COBEVT_ERR_ARG;
    return window_attention_impl(q, k, v, out, nullptr, bias_table, mask, dims, scale, 0.f, 0u, nullptr, stream, ksplit, part_out,
                                 part_lse, out_rows);
}

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_window_attention(const void* q, const void* k, const void* v, void* out,
                                       const float* bias_table, const float* mask, const int* dims, float scale,
                                       hipStream_t stream) {
    return window_attention_impl(q, k, v, out, nullptr, bias_table, mask, dims, scale, 0.f, 0u, nullptr, stream);
}

extern "C" int cobevt_window_attention_lse(const void* q, const void* k, const void* v, void* out, float* lse,
                                           const float* bias_table, const float* mask, const int* dims, float scale,
                                           float drop_p, unsigned drop_seed, const unsigned* drop_seed_dev, hipStream_t stream) {
    if (!lse) return COBEVT_ERR_ARG;
    return window_attention_impl(q, k, v, out, lse, bias_table, mask, dims, scale, drop_p, drop_seed, drop_seed_dev, stream);
}

// Test hook: the keep mask of the probability dropout exactly as the training kernels regenerate it
__global__ void attn_dropout_mask_kernel(AttnParams p, unsigned char* keep) {
    const long total = (long)p.B * p.L * p.heads * p.Nq * p.Nk;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
        const int tk = (int)(i % p.Nk);
        long r = i / p.Nk;
        const int tq = (int)(r % p.Nq); r /= p.Nq;
        const int head = (int)(r % p.heads); r /= p.heads;
        const int l = (int)(r % p.L), b = (int)(r / p.L);
        keep[i] = cobevt::attn_keep(p, b, l, head, tq, tk) ? 1 : 0;
    }
}

extern "C" int cobevt_attention_dropout_mask(int B, int L, int heads, int Nq, int Nk, float drop_p, unsigned drop_seed,
                                             unsigned char* keep, hipStream_t stream) {
    if (!keep || B < 1 || L < 1 || heads < 1 || Nq < 1 || Nk < 1) return COBEVT_ERR_ARG;
    AttnParams p = {};
    p.B = B; p.L = L; p.heads = heads; p.Nq = Nq; p.Nk = Nk; p.drop_p = drop_p; p.drop_seed = drop_seed;
    hipLaunchKernelGGL(attn_dropout_mask_kernel, dim3(2048), dim3(256), 0, stream, p, keep);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// Test hooks (tests/test_kernels_gpu.py: bit-exact against tests/golden/gv1_index_maps.npz), see include/cobevt_hip.h
extern "C" int cobevt_attention_index_map(const int* map8, int B, int* rows, hipStream_t stream) {
    if (!map8 || !rows || B < 1) return COBEVT_ERR_ARG;
    const TokMap m = read_map(map8);
    if (!map_ok(m)) return COBEVT_ERR_SHAPE;
    const int L = m.X * m.Y, ntok = m.ncam * m.w1 * m.w2;
    const long total = (long)B * L * ntok;
    hipLaunchKernelGGL(attn_index_dump_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256), 0,
                       stream, m, B, L, ntok, rows);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_attention_bias_index(const int* qmap8, const int* kmap8, int bias_L, int* idx, hipStream_t stream) {
    if (!qmap8 || !kmap8 || !idx || bias_L < 1) return COBEVT_ERR_ARG;
    const TokMap qm = read_map(qmap8), km = read_map(kmap8);
    if (!map_ok(qm) || !map_ok(km)) return COBEVT_ERR_SHAPE;
    const int nq = qm.ncam * qm.w1 * qm.w2, nk = km.ncam * km.w1 * km.w2;
    const long total = (long)nq * nk;
    hipLaunchKernelGGL(attn_bias_index_dump_kernel, dim3((unsigned)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256)), dim3(256),
                       0, stream, qm, km, bias_L, nq, nk, idx);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
