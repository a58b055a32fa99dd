// Gathered window / dilated-grid attention with the K / V of one (window, head) RESIDENT in LDS (gfx950, bf16).
//
// Same contract as attn_gather_kernel (attention.hip; reference fax_modules.py:211-237,243 / swap_fusion_modules.py:93-123 /
// fax_modules.py:137-171) for windows of at most 512 keys - every cross attention of FAX levels 0 and 1 and the swap-fusion
// attentions.  Differences that matter on MI355X:
//   * a workgroup loads its window's projected K rows and V^T ONCE (key table -> coalesced 16-byte gathers -> LDS), then every
//     wave walks its query tiles with NO further barrier: the streaming kernel re-staged the same K / V per 32-query tile
//     through a double buffer with a barrier per 64 keys (2.1x the algorithmic HBM bytes, 36 VALU per MFMA in rocprofv3);
//   * in "mean" mode (level-0 cross attention: one query copy per camera) a wave runs the cameras of its 32 BEV positions back
//     to back and keeps z.mean(1) (fax_modules.py:243) in registers - no LDS reduction, no barrier;
//   * LDS images are XOR-swizzled instead of padded, so K rows are 64 B and V^T rows Nk * 2 B (32 KB for 256 keys: 4 workgroups
//     per CU); every ds_read_b128 is conflict free and all its addressing is an immediate offset from a per-lane base;
//   * V^T keeps the 16 keys of an MFMA k-block in the order the score registers hold them (bits 2 and 3 of the key index
//     swapped), so one ds_read_b128 feeds one PV MFMA straight from the packed P registers;
//   * online softmax with a deferred rescale (the running maximum only moves when a tile exceeds it by more than 2^8), row
//     maxima by v_max3 + one v_permlane32_swap.
// MFMA mapping as in attention.hip: S^T = K.Q^T (lane = query), O^T += V^T.P^T.
#include <stdlib.h>
#include "attn_common.hpp"

namespace cobevt {

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr float kHeadroom = 1.099511627776e12f;   // 2^40: a tile is redone against its exact maximum when a row sum exceeds this
constexpr float kTiny = 8.271806125530277e-25f;   // 2^-80: a task is redone exactly when a row's total sum ends up below this

__device__ __forceinline__ int perm16(int k) {   // swap bits 2 and 3: key order inside a 16-key MFMA k-block
    return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1);
}

__device__ __forceinline__ float max3(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// 8 bf16 values times a scalar, rounded back to bf16 (v_pk_mul_f32 + v_cvt_pk_bf16_f32 per pair)
__device__ __forceinline__ uint4 scale_bf16x8(const uint4& v, float sc) {
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    uint32_t o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const f32x2 f = f32x2{__uint_as_float(w[i] << 16), __uint_as_float(w[i] & 0xffff0000u)} * sc;
        o[i] = pack_bf2(f.x, f.y);
    }
    return make_uint4(o[0], o[1], o[2], o[3]);
}

// max / sum across the two half-waves (lane ^ 32) without LDS: v_permlane32_swap exchanges the upper half of the first
// operand with the lower half of the second, so {r0, r1} = {own, partner} in one order or the other on every lane
__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

#ifdef COBEVT_RES_TRACE
// probe builds only (tools/attn_trace.py): s_memtime marks of wave 0 of the first workgroups: start, tables done, K / V staged, end
__device__ unsigned long long g_res_trace[4 * 8192];
#define RES_MARK(i)                                                                                   \
    do {                                                                                               \
        const unsigned wgid = item + nitems * blockIdx.y;          /* one record per (window, head) item */ \
        if (threadIdx.x == 0 && wgid < 8192) g_res_trace[4 * wgid + (i)] = __builtin_amdgcn_s_memtime(); \
    } while (0)
#else
#define RES_MARK(i) do { } while (0)
#endif

template <int NT> struct ResLds {
    static constexpr int kNkp = NT * 64;
    static constexpr int kVRow = kNkp * 2;
    static constexpr int kKBytes = kNkp * 64;
    static constexpr int kVBytes = 32 * kVRow;
};

// NT: 64-key tiles (keys padded to NT * 64; NT even so a V^T row is a whole number of 256-byte swizzle groups)
// NW: waves per workgroup.  MEAN: per-camera queries averaged (mean_q).  BIAS / MASK as in attn_gather_kernel.
// RAGGED: Nk is not a multiple of 128 and there is no per-key metadata (BIAS / MASK) to carry the padding: the padded keys are
// taken out of the softmax by a compare + select per score (a template parameter: as a run-time branch the compiler turns it
// into 64 always-executed compare / select pairs per tile).
// W8 (BIAS only): 8 x 8 windows and no padded keys - a 64-key tile is exactly one agent's window, so the padded key term of the
// quad (tile c, sub-tile s, accumulator quad g, half h) is the constant 960 c + 256 s + 64 g + 16 h bytes: the four bias reads
// of a sub-tile are immediate offsets from one per-lane pointer (no key-term lookup, no address arithmetic).
// PERSIST: a workgroup walks SEVERAL (window, head) items (grid.x < the item count) and issues the K / V / mask loads of its next
// item BEFORE the query loop of the current one, so the global round trip of a prologue hides under ~24k cycles of MFMA / VALU work.
// For the shapes with ONE workgroup per CU (the 512-key LiDAR FuseBEVT windows: 136 KB of LDS), where nothing else overlaps a
// prologue - 13.5k of a workgroup's 37k cycles in the s_memtime trace.  With grid.x a multiple of 8 x heads a
// workgroup keeps its head, so the four shifted copies of the bias column (58 KB) are built once per workgroup instead of once per
// window.  The prefetch registers (~50) live across the query loop: only instantiated where the VGPR budget is 256.
#ifndef COBEVT_ATTN_PLAIN_WAVES      // waves per SIMD the plain (no camera mean, no bias / mask) variants with >= 256 keys are compiled for
#define COBEVT_ATTN_PLAIN_WAVES 3
#endif
#ifndef COBEVT_ATTN_VTR              // V of the window as a row-major [key][32 dh] LDS image read through ds_read_b64_tr_b16 (below); 0 = the transposed image
#define COBEVT_ATTN_VTR 1
#endif
#ifndef COBEVT_ATTN_PIPE_SUMV
#define COBEVT_ATTN_PIPE_SUMV 1
#endif
#ifndef COBEVT_ATTN_MEAN_WAVES       // waves per SIMD the camera-mean variant is compiled for (2 = up to 256 VGPRs; it uses ~150 = three waves; 4 = 128 VGPRs spills 25
                                     // registers: one frame 2.003 -> 2.018 ms, measured and left)
#define COBEVT_ATTN_MEAN_WAVES 2
#endif
#ifndef COBEVT_ATTN_PREFETCH
#define COBEVT_ATTN_PREFETCH 0
#endif
#ifndef COBEVT_ATTN_PIPE             // software-pipelined key loop of the eight-wave bias / mask variants (below); 0 = the per-tile loop everywhere
#define COBEVT_ATTN_PIPE 1
#endif
template <int NT, int NW, bool MEAN, bool BIAS, bool MASK, bool RAGGED, bool W8 = false, bool PERSIST = false>
// Register budget: the plain variants keep 4 waves per SIMD (35 KB of LDS -> 4 workgroups per CU); with a bias table / mask the
// LDS footprint (>= 56 KB for the shipped windows) allows 2 waves per SIMD at most, so those variants may use 256 VGPRs.
__global__ __launch_bounds__(NW * 64, NT > 8 ? 1 : (BIAS || MASK || PERSIST) ? 2 : (MEAN ? COBEVT_ATTN_MEAN_WAVES : (NT >= 4 ? COBEVT_ATTN_PLAIN_WAVES : 3))) void attn_resident_kernel(AttnParams p, int qsplit) {
    using L = ResLds<NT>;
    constexpr int NKP = L::kNkp;
    constexpr int NTHR = NW * 64;
    constexpr bool INFO = BIAS || MASK;
    // two waves per SIMD, 256-VGPR budget; COBEVT_ATTN_PIPE = 2 (probe builds): the camera-mean variant (level-0 cross attention) as well
    constexpr bool PIPE = (COBEVT_ATTN_PIPE != 0 && INFO && NW == 8) || (COBEVT_ATTN_PIPE == 2 && MEAN && !RAGGED);
    constexpr int NITEM = NKP * 4 / NTHR;              // staging items per thread (K: 16-byte chunks; V: key pair x dh quad)
    static_assert(NKP * 4 % NTHR == 0 && NITEM >= 1, "tile / workgroup shape");
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* Ks = smem;
    unsigned char* Vts = smem + L::kKBytes;
    int* ktab = (int*)(smem + L::kKBytes + L::kVBytes);    // [NKP] row of key tk, -1 = padding
    // (INFO) [NKP][2] A operand of the "augmented" score MFMA: word 0 = bf16 {1, additive key mask (0, or -inf for masked / padded
    // keys)} for the k-slice of lanes 0-31, word 1 = 0 for lanes 32-63
    uint32_t* kaug = (uint32_t*)(ktab + NKP);
    int* kinfo4 = (int*)(kaug + (INFO ? 2 * NKP : 0));      // [NKP] (BIAS) 4 x the key term of the padded bias index (a byte offset)
    const int P = p.qmap.w1 * p.qmap.w2;
    const int NQ = MEAN ? P : p.Nq;                         // table entries: mean mode keeps camera 0 and strides over cameras
    int* qtab = kinfo4 + (BIAS ? NKP : 0);                  // [NQ] row of query token
    int* otab = qtab + NQ;                                  // [NQ] row of its output
    int* qbias = otab + NQ;                                 // [NQ] (BIAS) byte offset of the query's base inside bias4
    // (BIAS) the head's table column in the base-2 domain, REVERSED and with rows padded from 2 w2 - 1 to Wp = 2 w2 entries, in four
    // copies shifted by 0..3 entries: the four consecutive keys a lane holds per accumulator quad (same agent and window row,
    // columns j0..j0+3) then read four consecutive table entries - one aligned ds_read_b128 straight into the accumulator image
    // - and which copy is aligned depends on the query's column only, i.e. it is a per-lane constant (its base offset is qbias)
    const int Wp = 2 * p.kmap.w2;
    const int Rp = (2 * p.bias_L - 1) * (2 * p.kmap.w1 - 1) * Wp;
    const int CS = Rp + 4;                                  // floats per copy (a multiple of 4: every copy is 16-byte aligned)
    float* bias4 = (float*)(smem + ((((unsigned char*)(qbias + (BIAS ? NQ : 0)) - smem) + 15) & ~(size_t)15));

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.y;
    // grid.x = qsplit x (windows * heads), query split OUTER: the workgroups sharing a (window, head) are L * heads apart in
    // dispatch order = on the same XCD for the usual multiple-of-8 counts, so its L2 serves their common K / V
    const int LH = p.L * p.heads;
    const int nitems = LH * qsplit;
    // ... and the heads of one window too (they read the same 128-byte lines of the K / V / Q rows: a row holds all heads):
    // item lh -> (window, head) such that the heads of a window are 8 apart in dispatch order (8 XCDs, round-robin)
    auto decode_item = [&](int it, int& qs_, int& l_, int& head_) {
        qs_ = it / LH;
        const int lh = it - qs_ * LH;
        if ((p.L & 7) == 0) {
            const int grp = lh / (8 * p.heads), within = lh - grp * (8 * p.heads);
            head_ = within >> 3;
            l_ = grp * 8 + (within & 7);
        } else {
            l_ = lh / p.heads;
            head_ = lh - l_ * p.heads;
        }
    };
    int item = blockIdx.x;
    int qs, l, head;
    decode_item(item, qs, l, head);

    RES_MARK(0);
    // ---- prologue, ONE global round trip: with one workgroup per CU (the 512-key windows) nothing else runs on the CU while a
    // workgroup builds its tables and stages K / V - 19k of a LiDAR workgroup's 43k cycles in the s_memtime trace when the key
    // mask, the bias column and the K / V rows were three dependent batches of loads (tables -> barrier -> staging).  Every
    // staging item therefore derives its key's row itself (shifts for power-of-two windows) and all loads - the thread's mask
    // word first, it is needed first - are in flight together; the table arithmetic runs under them.
    const float sl2 = p.scale * kLog2e;
    // token -> (camera, i, j): shifts for power-of-two windows (every OPV2V shape; a runtime integer division is ~45 VALU
    // instructions, and the prologue evaluates a dozen of them per thread)
    auto fast_coord = [&](const TokMap& m, int t) {
        const int ws = m.w1 * m.w2;
        if (((ws & (ws - 1)) | (m.w2 & (m.w2 - 1))) != 0) return tok_coord(m, t);      // (uniform)
        TokCoord c;
        c.cam = t >> (31 - __builtin_clz(ws));
        const int rem = t & (ws - 1);
        c.i = rem >> (31 - __builtin_clz(m.w2));
        c.j = rem & (m.w2 - 1);
        return c;
    };
    auto key_coord = [&](int tk) { return fast_coord(p.kmap, tk); };

    // (a) this thread's keys of the tables: coordinates, row, mask word ; (b) K / V staging loads (rows, 16-byte chunks; key
    // pairs x dh quads), rows derived per staging item.  Both for ONE (window, head) item, into registers that live until its
    // staging: PERSIST issues them for the NEXT item in front of the current item's query loop
    constexpr int NKT = (NKP + NTHR - 1) / NTHR;
    TokCoord tkc[NKT];
    int trow[NKT];
    float tmask[NKT];
    uint4 kreg[NITEM];
    constexpr bool VTR = COBEVT_ATTN_VTR != 0;
    uint2 v0[VTR ? 1 : NITEM], v1[VTR ? 1 : NITEM];
    uint4 vreg[VTR ? NITEM : 1];                         // VTR: 16-byte pieces of the V rows, the K staging's item map
    int krow[NITEM], vr0[NITEM], vr1[VTR ? 1 : NITEM];
    // part A: the table keys' coordinates / rows / mask words and the K rows; part B: the V rows.  kPrefetch == 2 issues part A of the NEXT
    // item in front of the query loop (25 registers across it) and part B at the item's start, under the table arithmetic and the K staging
    auto issue_item_loads = [&](int l_, int head_, int tidp, bool part_a, bool part_b) {
        const RowAffine kaff = row_affine(p.kmap, b, l_);
        auto key_row = [&](int tk) { return tk < p.Nk ? row_of(kaff, key_coord(tk)) : -1; };
        if (part_a) {
#pragma unroll
        for (int u = 0; u < NKT; ++u) {
            const int tk = tidp + u * NTHR;
            const bool in = tk < p.Nk;
            tkc[u] = key_coord(in ? tk : 0);
            trow[u] = in ? row_of(kaff, tkc[u]) : -1;
            tmask[u] = 1.f;
            if (MASK) {                                   // unconditional load (address of token 0 for the padded keys)
                size_t mi;
                if (p.kmap.mode == 2) {
                    mi = ((((size_t)b * p.L + l_) * p.kmap.w1 + tkc[u].i) * p.kmap.w2 + tkc[u].j) * p.kmap.ncam + tkc[u].cam;
                } else {
                    const int ph = kaff.ph0 + tkc[u].i * kaff.pi, pw = kaff.pw0 + tkc[u].j * kaff.pj;
                    mi = (((size_t)b * p.kmap.HH + ph) * p.kmap.WW + pw) * p.kmap.ncam + tkc[u].cam;
                }
                tmask[u] = p.mask[mi];
            }
        }
        const bf16_t* kbase = (const bf16_t*)p.k + p.koff + head_ * 32;
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int si = tidp + it * NTHR;
            const int kk = si >> 2, cj = si & 3;
            krow[it] = key_row(kk);
            // unconditional loads from a clamped row, zeroed afterwards: a load under a branch makes hipcc wait for it inside the
            // branch (vmcnt(0) per load = one serialised HBM round trip per staging item)
            kreg[it] = *(const uint4*)(kbase + (size_t)(krow[it] < 0 ? 0 : krow[it]) * p.ldk + cj * 8);
        }
        }
        if (part_b) {
            const bf16_t* vbase = (const bf16_t*)p.v + p.voff + head_ * 32;
#pragma unroll
            for (int it = 0; it < NITEM; ++it) {
                const int si = tidp + it * NTHR;
                const int kp = si >> 3, dq = si & 7;
                if constexpr (VTR) {
                    vr0[it] = key_row(si >> 2);
                    vreg[it] = *(const uint4*)(vbase + (size_t)(vr0[it] < 0 ? 0 : vr0[it]) * p.ldv + (si & 3) * 8);
                } else {
                    vr0[it] = key_row(2 * kp);
                    vr1[it] = key_row(2 * kp + 1);
                    v0[it] = *(const uint2*)(vbase + (size_t)(vr0[it] < 0 ? 0 : vr0[it]) * p.ldv + dq * 4);
                    v1[it] = *(const uint2*)(vbase + (size_t)(vr1[it] < 0 ? 0 : vr1[it]) * p.ldv + dq * 4);
                }
            }
        }
    };
    constexpr int kPf = PERSIST ? COBEVT_ATTN_PREFETCH : 0;     // 0: none, 1: the whole next item, 2: its tables + K rows
    if (kPf != 0 || !PERSIST) issue_item_loads(l, head, tid, true, kPf != 2);
    int head_built = -1;                               // (BIAS) the head whose table column the four copies in LDS hold
    bool first_item = true;
    do {
    // the prologue's thread index, opaque per iteration: hoisted out of the item loop its address arithmetic (staging stores, table
    // slots, bias groups) stays live across the query loop - 125 spilled VGPRs in the first persistent build
    int tidp = tid;
    if (PERSIST) asm volatile("" : "+v"(tidp));
    if (!first_item) RES_MARK(0);
    if (PERSIST && kPf == 2) issue_item_loads(l, head, tidp, false, true);
    if (PERSIST && kPf == 0) issue_item_loads(l, head, tidp, true, true);     // (unconditional: behind `!first_item` the staging registers
    //                                                                    would be live around the back edge, i.e. across the query loop)
    const RowAffine qaff = row_affine(p.qmap, b, l), oaff = row_affine(p.omap, b, l);
    const bool build_bias = BIAS && head != head_built;
    // (c) the head's bias column.  A thread fills aligned 16-byte groups of the four shifted copies: group m of copy sh holds
    // rev[4 m + sh .. + 3], so the seven values rev[4 m .. 4 m + 6] make the group in all four copies (8 ds_write_b128 per thread
    // instead of 32 scattered ds_write_b32).  rev[x] = padded table entry Rp - 1 - x (0 in the padding column and past the end).
    constexpr int BG = 2;
    const int NG = Rp >> 2;
    const int w2m = 2 * p.kmap.w2 - 1;
    const bool wp_pow2 = (Wp & (Wp - 1)) == 0;
    const int wp_sh = 31 - __builtin_clz(Wp);
    float tv[BG][7];
    unsigned tvok[BG];
    auto bias_fetch = [&](int base) {
#pragma unroll
        for (int u = 0; u < BG; ++u) {
            const int m = base + u * NTHR + tidp;
            tvok[u] = 0u;
#pragma unroll
            for (int e = 0; e < 7; ++e) {
                const int i = Rp - 1 - (4 * m + e);
                const int ic = i < 0 ? 0 : (i >= Rp ? Rp - 1 : i);
                const int row = wp_pow2 ? (ic >> wp_sh) : ic / Wp, c = ic - row * Wp;
                const int src = row * w2m + (c < w2m ? c : w2m - 1);
                tv[u][e] = p.bias_table[(size_t)src * p.heads + head];          // (clamped, unconditional load)
                if (i >= 0 && i < Rp && c < w2m) tvok[u] |= 1u << e;
            }
        }
    };
    auto bias_store = [&](int base) {
#pragma unroll
        for (int u = 0; u < BG; ++u) {
            const int m = base + u * NTHR + tidp;
            if (m < NG) {
                float val[7];
#pragma unroll
                for (int e = 0; e < 7; ++e) val[e] = ((tvok[u] >> e) & 1u) ? tv[u][e] * kLog2e : 0.f;
#pragma unroll
                for (int sh = 0; sh < 4; ++sh)
                    *(f32x4*)(bias4 + sh * CS + 4 * m) = f32x4{val[sh], val[sh + 1], val[sh + 2], val[sh + 3]};
            }
        }
    };
    if (build_bias) bias_fetch(0);
    __builtin_amdgcn_sched_barrier(0);
    // (d) table arithmetic under the loads: query rows / output rows / bias bases, key terms
    for (int t = tidp; t < NQ; t += NTHR) {
        const TokCoord qc = fast_coord(p.qmap, t);          // mean mode: t < P -> camera 0
        qtab[t] = row_of(qaff, qc);
        otab[t] = row_of(oaff, qc);
        if (BIAS && first_item) {                           // (window- and head-independent)
            // padded index = query term - key term (linear in the coordinates, attn_common.hpp); reversed: a + key term
            const int qterm = ((qc.cam + p.bias_L - 1) * (2 * p.kmap.w1 - 1) + qc.i + p.kmap.w1 - 1) * Wp + qc.j + p.kmap.w2 - 1;
            const int a = Rp - 1 - qterm, sh = a & 3;
            qbias[t] = 4 * (sh * CS + a - sh);
        }
    }
#pragma unroll
    for (int u = 0; u < NKT; ++u) {
        const int tk = tidp + u * NTHR;
        if (tk < NKP) {
            ktab[tk] = trow[u];
            if (BIAS) kinfo4[tk] = trow[u] >= 0 ? 4 * ((tkc[u].cam * (2 * p.kmap.w1 - 1) + tkc[u].i) * Wp + tkc[u].j) : 0;
            if (INFO) {
                const bool valid = trow[u] >= 0 && tmask[u] != 0.f;
                kaug[2 * tk] = pack_bf2(1.0f, valid ? 0.f : -INFINITY);
                kaug[2 * tk + 1] = 0u;
            }
        }
    }
    RES_MARK(1);
    // (e) the loads land: bias copies, then K (rows, 16-byte chunks XOR-swizzled by (key >> 2) & 3) and V^T (dh rows, 16-byte
    // chunks XOR-swizzled by dh & 15)
    if (build_bias) {
        for (int base = 0; base < NG; base += NTHR * BG) {
            if (base > 0) bias_fetch(base);            // tables wider than one batch
            bias_store(base);
        }
        head_built = head;
    }
    {
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tidp + it * NTHR;
            const int kk = item >> 2, cj = item & 3;
            // K pre-scaled by scale * log2(e) (one extra bf16 rounding, once per workgroup): the score MFMA then delivers base-2
            // logits and, with C = -reference maximum, the exponent's argument itself - no per-score VALU before v_exp_f32
            const uint4 kv = krow[it] >= 0 ? scale_bf16x8(kreg[it], sl2) : make_uint4(0, 0, 0, 0);
            *(uint4*)(Ks + kk * 64 + ((cj ^ ((kk >> 2) & 3)) << 4)) = kv;
        }
        if constexpr (VTR) {
#pragma unroll
            for (int it = 0; it < NITEM; ++it) {         // V rows as they are: [key][32 dh], 16-byte pieces (padded / masked-out rows zero)
                const int item = tidp + it * NTHR;
                *(uint4*)(Vts + (item >> 2) * 64 + (item & 3) * 16) = vr0[it] >= 0 ? vreg[it] : make_uint4(0, 0, 0, 0);
            }
        } else
#pragma unroll
        for (int it = 0; it < NITEM; ++it) {
            const int item = tidp + it * NTHR;
            const int kp = item >> 3, dq = item & 7;
            const uint2 w0 = vr0[it] >= 0 ? v0[it] : make_uint2(0, 0), w1 = vr1[it] >= 0 ? v1[it] : make_uint2(0, 0);
            const int pos = ((2 * kp) & ~15) | perm16((2 * kp) & 15);     // even key of the pair; its partner sits at pos + 1
            const uint32_t a[2] = {w0.x, w0.y}, c[2] = {w1.x, w1.y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dh = dq * 4 + e;
                const uint32_t lo = (a[e >> 1] >> ((e & 1) * 16)) & 0xffffu, hi = (c[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                *(uint32_t*)(Vts + dh * L::kVRow + ((((pos >> 3) ^ (dh & 15))) << 4) + (pos & 7) * 2) = lo | (hi << 16);
            }
        }
    }
    // this wave's FIRST task's query rows, requested before the barrier (rows derived directly: the query table is not readable
    // yet): together with the next-task prefetch inside the tile loop no task starts with an exposed global round trip any more
    // (before: one per 32-query tile - two of a LiDAR wave's two tasks per item, two of a level-0 wave's eight)
    const int ntiles = (NQ + 31) >> 5;                  // 32-query tiles (mean mode: position tiles)
    const int tstep = qsplit * NW;
    const size_t qcam_stride = MEAN ? (p.qmap.mode == 2 ? (size_t)p.L * P : (size_t)p.qmap.HH * p.qmap.WW) : 0;
    const bf16_t* qbase = (const bf16_t*)p.q + p.qoff + head * 32 + h * 8;
    uint4 qf[2];
    {
        const int t0 = (qs + qsplit * wave) * 32 + ql;
        const bf16_t* qrow = qbase + (size_t)(unsigned)row_of(qaff, fast_coord(p.qmap, t0 < NQ ? t0 : 0)) * p.ldq;
        qf[0] = *(const uint4*)(qrow);
        qf[1] = *(const uint4*)(qrow + 16);
    }
    __syncthreads();
    RES_MARK(2);
    int nqs = qs, nl = l, nhead = head;
    if (PERSIST && item + (int)gridDim.x < nitems) {
        decode_item(item + gridDim.x, nqs, nl, nhead);
        // kPrefetch: the next item's global loads in flight under this item's query loop.  Measured with s_memtime marks
        // (tools/attn_trace.py --lidar, profiles/r05_attn_trace_lidar.txt): it takes the prologue from 12.8k to 3.9k cycles per item
        // but the ~50 registers it keeps live push the kernel from 171 to 251 VGPRs and the query loop from 24.2k to 31.3k cycles
        if (kPf != 0) issue_item_loads(nl, nhead, tidp, true, kPf != 2);
    }

    // ---- per-lane LDS read bases (everything else is an immediate offset)
    const uint32_t kx = (ql >> 2) & 3;
    const unsigned char* kptr0 = Ks + ql * 64 + ((h ^ kx) << 4);           // k-group 0: chunk h
    const unsigned char* kptr1 = Ks + ql * 64 + (((2 + h) ^ kx) << 4);     // k-group 1: chunk 2 + h
    const uint32_t vlow = (uint32_t)(ql & 15) << 4;                          // swizzle term, pre-shifted
    const unsigned char* vrow = Vts + ql * L::kVRow;                         // multiple of 256 B: the low byte is free for the XOR
    // VTR: this lane's address inside a 16-key block's first transpose read (see read_vt16): row 4 h + ((lane & 15) >> 2), dh half (lane >> 4) & 1
    const unsigned char* vtr = Vts + (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;

    const int ncam = MEAN ? p.qmap.ncam : 1;
    const float inv_ncam = 1.0f / (float)ncam;

    float m_run = 0.f;                             // softmax reference value, carried across this lane's tasks
    f32x16 mneg;                                   // splat(-m_run): the C operand of the score MFMAs (variants without bias / mask)
#pragma unroll
    for (int r = 0; r < 16; ++r) mneg[r] = 0.f;
    // Bias / mask variants: the accumulator image of a score MFMA is the BIAS tile itself (four aligned ds_read_b128 per 32 keys,
    // no VALU), so "- reference" and "+ key mask" ride on the matrix pipe instead: a third k-group whose A operand is {1, mask(key)}
    // (kaug) and whose B operand is {-reference(query), 1} - the reference therefore has to be a bf16 value (it is truncated to
    // one when it is set; any value works as a softmax reference as long as numerator and denominator use the same one).
    // qaug0 = {0, 1} is the B operand of the throw-away pass that measures a tile's exact maximum.
    const uint32_t qaug0 = h == 0 ? pack_bf2(0.f, 1.0f) : 0u;
    uint32_t qaugm = qaug0;
    for (int tile = qs + qsplit * wave; tile < ntiles; tile += tstep) {
        const int t = tile * 32 + ql;
        const bool q_ok = t < NQ;
        const int tq = q_ok ? t : 0;
        const size_t qrow0 = (size_t)(unsigned)qtab[tq];
        // first row of this lane's query in this wave's NEXT tile (this tile when there is none: loaded, never used)
        const int tnx = (tile + tstep < ntiles ? tile + tstep : tile) * 32 + ql;
        const size_t qrow_next_tile = (size_t)(unsigned)qtab[tnx < NQ ? tnx : 0];
        // LDS address of this query's base inside its (aligned) copy of the reversed bias column: a quad of keys reads 16 bytes
        // at this plus the quad's key term (one v_add)
        const unsigned char* bias_qp = (const unsigned char*)bias4 + (BIAS ? qbias[tq] : 0);
        f32x16 osum;
        if (MEAN) {
#pragma unroll
            for (int r = 0; r < 16; ++r) osum[r] = 0.f;
        }
        f32x16 ot;
        for (int cam = 0; cam < ncam; ++cam) {
            uint4 qn[2];
            {   // the NEXT task's query rows (unconditional, clamped: rows of lanes past the end are loaded but never stored), in
                // flight under this task's key tiles: the next camera of this tile, else camera 0 of this wave's next tile
                const bool more_cam = cam + 1 < ncam;
                const bf16_t* qrow = qbase + (more_cam ? qrow0 + (size_t)(cam + 1) * qcam_stride : qrow_next_tile) * p.ldq;
                qn[0] = *(const uint4*)(qrow);
                qn[1] = *(const uint4*)(qrow + 16);
            }
            const uint4 qs0 = qf[0], qs1 = qf[1];
            float l_run;
            // The softmax reference value m_run is carried over from the previous task of this lane (scores of neighbouring
            // queries live on the same scale) instead of being measured per tile: probabilities are 2^(s - m_run) in fp32 /
            // bf16, whose exponent range makes any reference within ~2^+-40 of the true maximum exact.  A tile whose row sum
            // leaves that range is redone against its exact maximum (`need_max`), a task whose total sum ends up tiny is redone
            // from an exact first tile (`exact`) - both wave-uniform and rare.
            for (int exact = 0; exact < 2; ++exact) {
#pragma unroll
            for (int r = 0; r < 16; ++r) ot[r] = 0.f;
            l_run = 0.f;
            if (PIPE && exact == 0) {
                // ---- the regular pass, software-pipelined over 32-key sub-tiles.  The per-tile loop below is a dependent chain per tile -
                // score MFMAs -> v_exp -> row-sum / PV MFMAs -> a wave-uniform branch on the tile's row sum - so a wave has either matrix
                // or VALU work to issue, never both, and with two waves per SIMD (one 8-wave workgroup per CU: the 512-key windows) the
                // two pipes were each ~60 % busy (747 cycles per 64-key tile and wave against 448 of MFMA issue / 456 of VALU).  Here the
                // scores of sub-tile u + 1 are issued BEFORE the exponentials of sub-tile u, their LDS operands (K rows, bias quads, key
                // mask word) a whole step earlier still, and nothing branches: the row sums accumulate on the matrix pipe over the whole task
                // and the range check moves to its end (a task whose sum left [2^-80, 2^40] - or is not finite - is redone by the exact
                // per-tile loop, exactly as a failed tile was).  Same arithmetic per score; the row sum is associated differently.
                f32x16 lt;
#pragma unroll
                for (int r = 0; r < 16; ++r) lt[r] = 0.f;
                // SUMV: the row sums as packed fp32 adds of the exponentials (this lane's 16 keys of a sub-tile, the two half-waves meet
                // once per task) instead of two ones(32 x 16) . P^T MFMAs per sub-tile: with the branches and the per-tile bookkeeping gone
                // the matrix pipe (7 MFMAs = 224 cycles per sub-tile and wave) is the longer side, the VALU (~180) has the room
                constexpr bool SUMV = COBEVT_ATTN_PIPE_SUMV != 0;
                f32x2 ls2[4] = {f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}, f32x2{0.f, 0.f}};
                const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
                const uint4 qa = make_uint4(qaugm, 0u, 0u, 0u);
                // LDS operands of sub-tile u (0..3) of key-tile pair kp2v; st = the accumulator image (the bias tile, or zero)
                auto sub_reads = [&](int kp2v, int u, uint4& a0, uint4& a1, uint32_t& ka, f32x16& st) {
                    const int off = kp2v * (128 * 64) + u * (32 * 64);
                    a0 = *(const uint4*)(kptr0 + off);
                    a1 = *(const uint4*)(kptr1 + off);
                    const int kt = kp2v * 128 + u * 32;
                    if (BIAS) {
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const f32x4 bb = W8 ? *(const f32x4*)(bias_qp + 16 * h + kp2v * 1920 + (u >> 1) * 960 + (u & 1) * 256 + g * 64)
                                                : *(const f32x4*)(bias_qp + kinfo4[kt + 8 * g + 4 * h]);
                            st[4 * g] = bb.x; st[4 * g + 1] = bb.y; st[4 * g + 2] = bb.z; st[4 * g + 3] = bb.w;
                        }
                    } else if (INFO) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) st[r] = 0.f;
                    }
                    ka = INFO ? kaug[2 * (kt + ql) + h] : 0u;
                };
                auto sub_scores = [&](const uint4& a0, const uint4& a1, uint32_t ka, f32x16& st) {
                    if (!INFO) st = mneg;                     // (the first MFMA's C operand)
                    mfma_kgroup<bf16_t>(a0, qs0, st);
                    mfma_kgroup<bf16_t>(a1, qs1, st);
                    if (INFO) mfma_kgroup<bf16_t>(make_uint4(ka, 0u, 0u, 0u), qa, st);
                };
                constexpr bool AHEAD2 = INFO;               // operand reads two sub-tiles ahead (the 256-VGPR variants), else one
                f32x16 sc, sn;
                uint4 n0, n1;                               // operands of the sub-tile whose scores are issued next
                uint32_t nk;
                {
                    uint4 a0, a1;
                    uint32_t ka;
                    sub_reads(0, 0, a0, a1, ka, sc);
                    if (AHEAD2) sub_reads(0, 1, n0, n1, nk, sn);
                    sub_scores(a0, a1, ka, sc);
                }
#pragma unroll 1
                for (int kp2 = 0; kp2 < NT / 2; ++kp2) {
                    const int kp2n = kp2 + 1 < NT / 2 ? kp2 + 1 : kp2;       // past the last pair: its first sub-tiles again, never used
                    const unsigned char* vp = vrow + kp2 * 256;
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        // LDS reads: sub-tile u + 2's score operands (a whole step ahead of their MFMAs), this sub-tile's V^T fragments
                        f32x16 sm;
                        uint4 m0, m1, va[2];
                        uint32_t mk;
                        if (AHEAD2) sub_reads(u < 2 ? kp2 : kp2n, (u + 2) & 3, m0, m1, mk, sm);
                        else sub_reads(u < 3 ? kp2 : kp2n, (u + 1) & 3, n0, n1, nk, sn);
#pragma unroll
                        for (int uu = 0; uu < 2; ++uu) {
                            const uint32_t lowc = (uint32_t)((u >> 1) * 8 + ((u & 1) * 2 + uu) * 2) << 4;
                            if constexpr (VTR) va[uu] = read_vt16(vtr + (kp2 * 128 + u * 32 + uu * 16) * 64);
                            else va[uu] = *(const uint4*)(vp + ((lowc | ((uint32_t)h << 4)) ^ vlow));
                        }
                        __builtin_amdgcn_sched_barrier(0);
                        f32x2 e2[8];                         // pairs: the row-sum adds below are v_pk_add_f32 on them
                        uint4 pA, pB;
#pragma unroll
                        for (int r = 0; r < 4; ++r) e2[r] = f32x2{__builtin_amdgcn_exp2f(sc[2 * r]), __builtin_amdgcn_exp2f(sc[2 * r + 1])};
                        pA = make_uint4(pack_bf2(e2[0].x, e2[0].y), pack_bf2(e2[1].x, e2[1].y), pack_bf2(e2[2].x, e2[2].y), pack_bf2(e2[3].x, e2[3].y));
                        __builtin_amdgcn_sched_barrier(0);
                        sub_scores(n0, n1, nk, sn);          // sub-tile u + 1
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int r = 4; r < 8; ++r) e2[r] = f32x2{__builtin_amdgcn_exp2f(sc[2 * r]), __builtin_amdgcn_exp2f(sc[2 * r + 1])};
                        pB = make_uint4(pack_bf2(e2[4].x, e2[4].y), pack_bf2(e2[5].x, e2[5].y), pack_bf2(e2[6].x, e2[6].y), pack_bf2(e2[7].x, e2[7].y));
                        __builtin_amdgcn_sched_barrier(0);
                        if (SUMV) {
#pragma unroll
                            for (int r = 0; r < 8; ++r)     // (as inline asm: the register allocator then keeps the exponentials in aligned pairs)
                                asm("v_pk_add_f32 %0, %1, %2" : "=v"(ls2[r & 3]) : "v"(ls2[r & 3]), "v"(e2[r]));
                        } else {
                            mfma_kgroup<bf16_t>(ones, pA, lt);
                            mfma_kgroup<bf16_t>(ones, pB, lt);
                        }
                        mfma_kgroup<bf16_t>(va[0], pA, ot);
                        mfma_kgroup<bf16_t>(va[1], pB, ot);
                        __builtin_amdgcn_sched_barrier(0);
                        sc = sn;
                        if (AHEAD2) { sn = sm; n0 = m0; n1 = m1; nk = mk; }
                    }
                }
                if (SUMV) {
                    const f32x2 t2 = (ls2[0] + ls2[1]) + (ls2[2] + ls2[3]);
                    l_run = xor32_sum(t2.x + t2.y);
                } else {
                    l_run = lt[0];
                }
                if (!__any(!(l_run <= kHeadroom) || !(l_run >= kTiny))) break;
                continue;                                   // redo this task exactly
            }
            bool have_m = exact == 0;
            if (exact) m_run = -INFINITY;
            // two 64-key tiles per iteration: inside a pair every LDS address is a per-lane base + an immediate (the V^T swizzle
            // flips with the tile parity); rolled over the pairs, so the register footprint does not grow with the window size
#pragma unroll 1
            for (int kp2 = 0; kp2 < NT / 2; ++kp2) {
                const unsigned char* kp0 = kptr0 + kp2 * (128 * 64);
                const unsigned char* kp1 = kptr1 + kp2 * (128 * 64);
                const unsigned char* vp = vrow + kp2 * 256;
                const unsigned char* bk = bias_qp + 16 * h + kp2 * 1920;       // W8: this lane's bias base of agent 2 kp2
#pragma unroll
                for (int par = 0; par < 2; ++par) {
                    const int key0 = kp2 * 128 + par * 64;      // first key of this tile
                    // scores of one 32-key sub-tile; with_ref: minus the running reference (the exponent's argument), otherwise
                    // the plain logits (+ bias, + additive key mask in both cases)
                    auto scores = [&](int s, bool with_ref) -> f32x16 {
                        f32x16 st;
                        const int off = (par * 64 + s * 32) * 64;
                        const uint4 a0 = *(const uint4*)(kp0 + off);
                        const uint4 a1 = *(const uint4*)(kp1 + off);
                        if (INFO) {
                            const int kt = key0 + s * 32;
                            if (BIAS) {    // this lane's 16 keys of the sub-tile: 4 x (4 consecutive keys 8g + 4h ..) = 4 aligned 16-byte reads
#pragma unroll
                                for (int g = 0; g < 4; ++g) {
                                    const f32x4 bb = W8 ? *(const f32x4*)(bk + par * 960 + s * 256 + g * 64)
                                                        : *(const f32x4*)(bias_qp + kinfo4[kt + 8 * g + 4 * h]);
                                    st[4 * g] = bb.x; st[4 * g + 1] = bb.y; st[4 * g + 2] = bb.z; st[4 * g + 3] = bb.w;
                                }
                            } else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) st[r] = 0.f;
                            }
                            const uint4 ka = make_uint4(kaug[2 * (kt + ql) + h], 0u, 0u, 0u);
                            const uint4 qa = make_uint4(with_ref ? qaugm : qaug0, 0u, 0u, 0u);
                            mfma_kgroup<bf16_t>(a0, qs0, st);
                            mfma_kgroup<bf16_t>(a1, qs1, st);
                            mfma_kgroup<bf16_t>(ka, qa, st);
                        } else {
                            if (with_ref) st = mneg;
                            else {
#pragma unroll
                                for (int r = 0; r < 16; ++r) st[r] = 0.f;
                            }
                            mfma_kgroup<bf16_t>(a0, qs0, st);
                            mfma_kgroup<bf16_t>(a1, qs1, st);
                            if (RAGGED) {                       // padded keys out of the softmax
                                const int nv = p.Nk - key0 - s * 32 - 4 * h;
#pragma unroll
                                for (int r = 0; r < 16; ++r)
                                    if ((r & 3) + 8 * (r >> 2) >= nv) st[r] = -INFINITY;
                            }
                        }
                        return st;
                    };
                    uint4 pb[4];
                    float psum;
                    bool need_max = !have_m;
                    for (;;) {
                        if (need_max) {
                            // exact maximum of this tile (first tile of a task, or a tile that outgrew the running maximum):
                            // throw-away score MFMAs, then the rescale; the regular pass below then runs against the new maximum
                            float mloc = -INFINITY;
#pragma unroll
                            for (int s = 0; s < 2; ++s) {
                                const f32x16 st = scores(s, false);
                                float m0 = max3(st[0], st[1], st[2]);
#pragma unroll
                                for (int r = 3; r < 15; r += 2) m0 = max3(m0, st[r], st[r + 1]);
                                mloc = max3(mloc, m0, st[15]);
                            }
                            float m_new = fmaxf(m_run, xor32_max(mloc));
                            if (INFO) m_new = __uint_as_float(__float_as_uint(m_new) & 0xffff0000u);     // a bf16 value (see qaugm)
                            const float m_safe = (m_new == -INFINITY) ? 0.f : m_new;
                            const float alpha = __builtin_amdgcn_exp2f(m_run - m_safe);       // first tile: exp2(-inf) = 0
                            m_run = m_new;
                            l_run *= alpha;
                            ot *= alpha;
#pragma unroll
                            for (int r = 0; r < 16; ++r) mneg[r] = -m_safe;
                            if (INFO) qaugm = h == 0 ? pack_bf2(-m_safe, 1.0f) : 0u;
                            have_m = true;
                        }
                        // ---- regular pass: P = exp2(S - m_run) straight from the MFMA result, packed to bf16.  Row sums on the
                        // matrix pipe too (it has the slack, the VALU does not): ones(32 x 16) . P^T sums the 16 keys of a k-block
                        // for every query into all 16 accumulator rows - 4 MFMAs replace 32 v_add_f32 and the half-wave exchange
                        // (the first one takes a literal zero accumulator - an inline constant of the instruction - instead of sixteen
                        //  v_mov per tile to clear `lt`: 16 of the ~70 VALU instructions of a tile in the ISA)
                        f32x16 lt;
                        const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
                        for (int s = 0; s < 2; ++s) {
                            const f32x16 st = scores(s, true);
                            float e[16];
#pragma unroll
                            for (int r = 0; r < 16; ++r) e[r] = __builtin_amdgcn_exp2f(st[r]);
#pragma unroll
                            for (int u = 0; u < 2; ++u) {
                                pb[s * 2 + u].x = pack_bf2(e[8 * u + 0], e[8 * u + 1]);
                                pb[s * 2 + u].y = pack_bf2(e[8 * u + 2], e[8 * u + 3]);
                                pb[s * 2 + u].z = pack_bf2(e[8 * u + 4], e[8 * u + 5]);
                                pb[s * 2 + u].w = pack_bf2(e[8 * u + 6], e[8 * u + 7]);
                                if (s == 0 && u == 0) {
                                    const f32x16 zero = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
                                    lt = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ones), __builtin_bit_cast(bf16x8, pb[0]),
                                                                                 zero, 0, 0, 0);
                                } else {
                                    mfma_kgroup<bf16_t>(ones, pb[s * 2 + u], lt);
                                }
                            }
                        }
                        psum = lt[0];                       // the whole tile's row sum of this lane's query (both key halves)
                        // no per-tile maximum: the running maximum is kept as long as no probability of the tile exceeds
                        // 2^kHeadroomLog2 (fp32 / bf16 share the exponent range, so nothing is lost below that); inf / NaN
                        // fail the comparison too
                        if (need_max || !__any(!(psum <= kHeadroom))) break;
                        need_max = true;
                    }
                    have_m = true;
                    l_run += psum;
                    // ---- O^T += V^T . P^T
                    if constexpr (PERSIST) {
                        // all four V^T fragments requested before the first PV MFMA, the order pinned: inside the item loop the
                        // scheduler otherwise issues them one at a time, each behind lgkmcnt(0) in front of its MFMA
                        uint4 va[4];
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            const uint32_t lowc = (uint32_t)(par * 8 + j * 2) << 4;
                            if constexpr (VTR) va[j] = read_vt16(vtr + (key0 + j * 16) * 64);
                            else va[j] = *(const uint4*)(vp + ((lowc | ((uint32_t)h << 4)) ^ vlow));
                        }
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int j = 0; j < 4; ++j) mfma_kgroup<bf16_t>(va[j], pb[j], ot);
                    } else {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {            // j = s * 2 + u: keys key0 + s*32 + u*16 .. +15, this half's 8 slots
                        const uint32_t lowc = (uint32_t)(par * 8 + j * 2) << 4;      // chunk within the 256-byte group (+ h)
                        const uint4 va = VTR ? read_vt16(vtr + (key0 + j * 16) * 64) : *(const uint4*)(vp + ((lowc | ((uint32_t)h << 4)) ^ vlow));
                        mfma_kgroup<bf16_t>(va, pb[j], ot);
                    }
                    }
                }
            }
            if (exact || !__any(!(l_run >= kTiny))) break;  // reference too far above this task's scores: redo it exactly
            }
            const float inv = 1.0f / l_run;                // an all-masked row yields NaN like the reference softmax
            if (MEAN) {
                osum += ot * inv;                              // v_pk_fma_f32 x 8
            } else {
                ot *= inv;
            }
            qf[0] = qn[0];
            qf[1] = qn[1];
        }
        if (MEAN) ot = osum * inv_ncam;
        if (q_ok) {
            bf16_t* orow = (bf16_t*)p.out + (size_t)(unsigned)otab[tq] * p.ldo + p.ooff + head * 32;
#pragma unroll
            for (int g4 = 0; g4 < 4; ++g4) {
                uint2 w;
                w.x = pack_bf2(ot[4 * g4 + 0], ot[4 * g4 + 1]);
                w.y = pack_bf2(ot[4 * g4 + 2], ot[4 * g4 + 3]);
                *(uint2*)(orow + 8 * g4 + 4 * h) = w;
            }
        }
    }
    RES_MARK(3);
    if (!PERSIST || item + (int)gridDim.x >= nitems) break;
    __syncthreads();                               // every wave is done with this item's K / V / tables: the next item overwrites them
    item += gridDim.x;
    qs = nqs; l = nl; head = nhead;
    first_item = false;
    } while (true);
}

template <int NT, int NW, bool P = false>
int launch_nt(const AttnParams& p, int qsplit, size_t lds, dim3 grid, hipStream_t stream) {
    const bool hb = p.bias_mode != 0, hm = p.mask != nullptr, mean = p.mean_q != 0;
    const bool ragged = p.Nk != NT * 64;
#define COBEVT_RES_LAUNCH(M, B, K, R) \
    hipLaunchKernelGGL((attn_resident_kernel<NT, NW, M, B, K, R, false, P>), grid, dim3(NW * 64), lds, stream, p, qsplit)
#define COBEVT_RES_LAUNCH_W8(K) \
    hipLaunchKernelGGL((attn_resident_kernel<NT, NW, false, true, K, false, true, P>), grid, dim3(NW * 64), lds, stream, p, qsplit)
    const bool w8 = hb && p.kmap.w1 == 8 && p.kmap.w2 == 8 && p.Nk == NT * 64;
    if (mean) {
        if (hb || hm) return -1;                       // the camera mean only occurs in the plain cross attention
        if (ragged) COBEVT_RES_LAUNCH(true, false, false, true);
        else COBEVT_RES_LAUNCH(true, false, false, false);
    } else if (w8 && hm) COBEVT_RES_LAUNCH_W8(true);
    else if (w8) COBEVT_RES_LAUNCH_W8(false);
    else if (hb && hm) COBEVT_RES_LAUNCH(false, true, true, false);
    else if (hb) COBEVT_RES_LAUNCH(false, true, false, false);
    else if (hm) COBEVT_RES_LAUNCH(false, false, true, false);
    else if (ragged) COBEVT_RES_LAUNCH(false, false, false, true);
    else COBEVT_RES_LAUNCH(false, false, false, false);
#undef COBEVT_RES_LAUNCH
#undef COBEVT_RES_LAUNCH_W8
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// windows of 513 .. 1024 keys: the plain variant only (four waves, one workgroup per CU)
template <int NT>
int launch_big(const AttnParams& p, int qsplit, size_t lds, dim3 grid, hipStream_t stream) {
    if (p.Nk != NT * 64) hipLaunchKernelGGL((attn_resident_kernel<NT, 4, false, false, false, true>), grid, dim3(256), lds, stream, p, qsplit);
    else hipLaunchKernelGGL((attn_resident_kernel<NT, 4, false, false, false, false>), grid, dim3(256), lds, stream, p, qsplit);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace

// A/B switch: COBEVT_ATTN_BIG=0 in the environment keeps windows of more than 512 keys on the streaming kernel
static bool attn_big_enabled() {
    static const bool on = [] { const char* e = getenv("COBEVT_ATTN_BIG"); return !(e && e[0] == '0'); }();
    return on;
}

// A/B switch: COBEVT_ATTN_PERSIST=0 in the environment keeps one (window, head) item per workgroup on every shape
static bool attn_persist_enabled() {
    static const bool on = [] { const char* e = getenv("COBEVT_ATTN_PERSIST"); return !(e && e[0] == '0'); }();
    return on;
}

int launch_attn_resident(const AttnParams& p, int qsplit_hint, hipStream_t stream) {
    // <= 64 keys: one streaming tile is already optimal; > 512: LDS - except the plain variant (no bias / mask / camera mean), whose K / V
    // of up to 1024 keys (the FAX level-2 and global attentions: one whole-map window per agent) fit as 128 KB + tables: four-wave
    // workgroups, one per CU, one 32-query tile per wave instead of the streaming kernel's key split + merge launch (round 6)
    const bool big = p.Nk > 512;
    if (p.Nk < 65 || p.Nk > 1024 || (big && (p.bias_mode != 0 || p.mask != nullptr || p.mean_q != 0 || !attn_big_enabled()))) return -1;
    const int nt = ((p.Nk + 127) / 128) * 2;           // 64-key tiles, even
    const int P = p.qmap.w1 * p.qmap.w2;
    const bool mean = p.mean_q != 0;
    if (!mean && p.omap.ncam != p.qmap.ncam) return -1;
    if (mean && p.omap.ncam != 1) return -1;
    const int NQ = mean ? P : p.Nq;
    const bool info = p.bias_mode != 0 || p.mask != nullptr;
    size_t lds = (size_t)nt * 64 * 128 + (size_t)nt * 64 * 4 * (1 + (info ? 2 : 0) + (p.bias_mode ? 1 : 0)) +
                 (size_t)NQ * 4 * (p.bias_mode ? 3 : 2);
    if (p.bias_mode) {
        // four shifted copies of the reversed, row-padded table column (see the kernel): key quads must share (agent, window row)
        if (p.kmap.w2 % 4 != 0 || p.bias_rows != (2 * p.bias_L - 1) * (2 * p.kmap.w1 - 1) * (2 * p.kmap.w2 - 1)) return -1;
        const size_t rp = (size_t)(2 * p.bias_L - 1) * (2 * p.kmap.w1 - 1) * (2 * p.kmap.w2);
        lds = ((lds + 15) & ~(size_t)15) + 4 * (rp + 4) * 4;
    }
    lds = (lds + 15) & ~(size_t)15;
    if (lds > 160 * 1024) return -1;
    if ((long)p.B * p.qmap.ncam * (p.qmap.mode == 2 ? (long)p.L * P : (long)p.qmap.HH * p.qmap.WW) >= 0x7fffffffL) return -1;
    const int ntiles = (NQ + 31) / 32;
    // waves per workgroup: 8 when the LDS footprint leaves room for one or two workgroups per CU only and the window has the
    // query tiles to feed them (LiDAR FuseBEVT: 512 tokens per window)
    // (16 waves = 4 per SIMD at one workgroup per CU was measured 2x slower for the 512-key bias + mask windows: 128 VGPRs spill)
    const int nw = big ? 4 : (lds > 40 * 1024 && ntiles >= 16) ? 8 : 4;
    // query split: enough workgroups to fill 256 CUs x (4 | 2 | 1 resident workgroups), every wave keeping >= 1 tile
    int qsplit = qsplit_hint;
    if (qsplit <= 0) {
        const long base = (long)p.B * p.L * p.heads;
        const int resident = lds > 80 * 1024 ? 1 : (lds > 40 * 1024 ? 2 : 4);
        qsplit = 1;
        while (base * qsplit < 256L * resident && ntiles >= 2 * qsplit * nw) qsplit *= 2;
        if (base * qsplit < 256L && ntiles >= 2 * qsplit * nw - nw) qsplit *= 2;     // fewer workgroups than CUs: one tile per wave
        // bias / mask windows on a grid that still does not reach the CU count (the 5-agent fusion: 64 (window, head) pairs of 10 query
        // tiles): keep splitting while a workgroup keeps two tiles - 13.2 us against the streaming kernel's 14.9 us in-graph
        if (info) while (base * qsplit < 256L && ntiles >= 4 * qsplit) qsplit *= 2;
    }
    if (big && qsplit_hint <= 0) {                      // one query tile per wave where the window has them
        qsplit = 1;
        while (qsplit * 2 * nw <= ntiles) qsplit *= 2;
    }
    if (qsplit > ntiles) qsplit = ntiles;
    if (qsplit < 1) qsplit = 1;
    // fewer workgroups than CUs (nuScenes: 100 windows x 1 head; the 5-agent fusion: 16 windows x 4 heads): the streaming kernel's
    // finer query split fills the chip better than one staging per (window, head) can (measured: 31 vs 52 us, 25 vs 29 us)
    if (qsplit_hint <= 0 && !big && (long)p.B * p.L * p.heads * qsplit < 256) return -1;
    dim3 grid(p.L * p.heads * qsplit, p.B);
    if (grid.y > 65535) return -1;
    // one workgroup per CU (> 80 KB of LDS) and several items per CU: persistent workgroups, a whole number of (8 windows x heads)
    // groups of them so that a workgroup keeps its head (and its bias copies) across its items
    bool persist = false;
    if (attn_persist_enabled() && nw == 8 && lds > 80 * 1024 && !mean) {
        static int cus = 0;
        if (cus == 0) {
            int dev = 0;
            hipDeviceProp_t prop;
            cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
        }
        const int per = 8 * p.heads;
        int gx = cus >= per && (p.L & 7) == 0 ? (cus / per) * per : cus;
        if ((long)gx * 2 <= (long)grid.x) {             // at least two items per workgroup, else the plain form
            grid.x = gx;
            persist = true;
        }
    }
    switch (nt * 10 + nw) {
        case 24: return launch_nt<2, 4>(p, qsplit, lds, grid, stream);
        case 44: return launch_nt<4, 4>(p, qsplit, lds, grid, stream);
        case 64: return launch_nt<6, 4>(p, qsplit, lds, grid, stream);
        case 84: return launch_nt<8, 4>(p, qsplit, lds, grid, stream);
        case 28: return persist ? launch_nt<2, 8, true>(p, qsplit, lds, grid, stream) : launch_nt<2, 8>(p, qsplit, lds, grid, stream);
        case 48: return persist ? launch_nt<4, 8, true>(p, qsplit, lds, grid, stream) : launch_nt<4, 8>(p, qsplit, lds, grid, stream);
        case 68: return persist ? launch_nt<6, 8, true>(p, qsplit, lds, grid, stream) : launch_nt<6, 8>(p, qsplit, lds, grid, stream);
        case 88: return persist ? launch_nt<8, 8, true>(p, qsplit, lds, grid, stream) : launch_nt<8, 8>(p, qsplit, lds, grid, stream);
        case 104: return launch_big<10>(p, qsplit, lds, grid, stream);
        case 124: return launch_big<12>(p, qsplit, lds, grid, stream);
        case 144: return launch_big<14>(p, qsplit, lds, grid, stream);
        case 164: return launch_big<16>(p, qsplit, lds, grid, stream);
        default: return -1;
    }
}

}  // namespace cobevt

#ifdef COBEVT_RES_TRACE
extern "C" int cobevt_res_trace_read(unsigned long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(cobevt::g_res_trace), sizeof(unsigned long long) * n);
}
#endif
