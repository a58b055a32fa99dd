// One half of a SwapFusionBlock as ONE launch (gfx950, bf16): PreNormResidual(Attention) + PreNormResidual(FeedForward) over
// the window (mode 0) or dilated-grid (mode 1) partition of the (b, l, h, w, d) agent maps -
// opv2v/opencood/models/fusion_modules/swap_fusion_modules.py:87-128 (Attention.forward: q * scale, QK^T, + 3-D relative position
// bias, key mask, softmax, PV), :126 / :172-190 (to_out, the residuals, the window / grid rearranges) and base_transformer.py:102-124
// (PreNormResidual, FeedForward) - plus, while the rows are still in LDS, the LayerNorm + to_qkv of the NEXT half (:93 behind
// PreNormResidual.norm).
//
// The 5-agent fusion stage is 5120 tokens x 128 channels = 1.3 MB: it lives in L2, and as separate launches (attention core, then
// the row chain of row_chain.hip) each half cost ~28 us for ~2 GFLOP - launch latency, a 256-workgroup attention grid at 4 % MFMA
// utilisation and the attention output's round trip through memory.  Here a workgroup owns 32 QUERY TOKENS of one window / grid
// group (16 groups x 10 query slices = 160 workgroups for the camera config) and carries them through the whole half:
//   attention   wave w = head w (dim_head 32, 4 heads): K rows of the group straight from L2 as MFMA A fragments (S^T = K.Q^T, a
//               lane owns one query), V^T of the head staged by the wave into LDS in the score registers' key order (O^T +=
//               V^T.P^T straight from the packed probabilities), online softmax per 32-key tile in the base-2 domain, the
//               3-D bias as (query term - key term) gathers from an LDS copy of the head's table column, additive key mask;
//   row chain   the 32 x 128 attention output never leaves LDS: out-projection + residual -> LayerNorm -> fc1 + GELU -> fc2 +
//               residual -> (LayerNorm -> next to_qkv), the phases of row_chain_kernel<2, 32, true> with the rows addressed through
//               the group's token -> row table (gather of the residual rows, scatter of the results; the maps stay (b, l, h, w, d)).
// LDS: the attention region (V of 4 heads 80 KB, row-major, read through the transpose read + bias columns 40 KB + key tables) is dead when the chain starts and is reused
// for its y / hidden / staging tiles; ~130 KB per workgroup, one workgroup per CU (the grid has 160).
#include "attn_common.hpp"

namespace cobevt {

namespace {

constexpr float kLog2e = 1.4426950408889634f;
constexpr int kRow = 256 + 16;              // 128 bf16 + pad  (row_chain.hip kRcRow)
constexpr int kHRow = 512 + 16;             // 256 bf16 + pad ; also the fp32 staging row of 128 floats
constexpr int kRows = 32, kThreads = 256;
// bias table of the chain (floats): [0,128) bp, [128,384) b1, [384,512) b2, [512,896) bnext (<= 384 columns)
constexpr int kBp = 0, kB1 = 128, kB2 = 384, kBn = 512, kBiasFloats = 1024;
// relative-position table image handed over by the host: [4 heads][brows] fp32 (x log2 e) zero-padded to this many bytes
constexpr int kBiasCopyIters = 10, kBiasCopyBytes = kBiasCopyIters * kThreads * 16;

struct SwapStageParams {
    const bf16_t* qkv;      // [rows][3C]  q | k | v = to_qkv(LayerNorm(x)) of this half
    const bf16_t* x;        // [rows][C]   the half's input (residual of the attention)
    bf16_t* out;            // [rows][C]
    bf16_t* qkv_next;       // [rows][Nn]  to_qkv(LayerNorm(out)) of the next half, or null
    TokMap map;             // window (0) / grid (1) partition: ncam = agents, HH x WW map, w1 x w2 windows, X x Y groups
    int B, NK, NKP, nsplit;
    float scale;
    const float* bias_table;    // [4 heads][brows] fp32 (brows = bias_rows rounded up to 4) x log2(e), zero-padded to 10240 floats (host)
    int bias_rows, bias_L;
    const float* mask;      // (B, HH, WW, ncam) fp32, 0 = key masked out; nullable
    const uint4* wp; const float* bp;       // to_out            fragment-ordered [4 tiles][8]
    const uint4* w1; const float* b1;       // fc1 (LN folded)   [8 tiles][8]
    const uint4* w2; const float* b2;       // fc2               [4 tiles][Hdp / 16]
    const uint4* wn; const float* bn;       // next to_qkv (LN folded) [4 * ceil(Nn / 128) tiles][8], nullable
    int Hd, Hdp, Nn;
    float eps1, eps_next;
};

#ifdef COBEVT_STAGE_TRACE
// probe builds only (tools/stage_trace.py): s_memtime marks of wave 0 of every workgroup at the phase boundaries
__device__ unsigned long long g_stage_trace[16 * 4096];
#define STAGE_MARK(i)                                                                                       \
    do {                                                                                                     \
        const unsigned wgid = blockIdx.x + gridDim.x * blockIdx.y;                                           \
        if (threadIdx.x == 0 && wgid < 4096) g_stage_trace[16 * wgid + (i)] = __builtin_amdgcn_s_memtime();   \
    } while (0)
#else
#define STAGE_MARK(i) do { } while (0)
#endif

__device__ __forceinline__ int perm16(int k) { return (k & ~12) | ((k & 4) << 1) | ((k & 8) >> 1); }

__device__ __forceinline__ float xor32_max(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return fmaxf(__uint_as_float(r[0]), __uint_as_float(r[1]));
}
__device__ __forceinline__ float xor32_sum(float v) {
    const auto r = __builtin_amdgcn_permlane32_swap(__float_as_uint(v), __float_as_uint(v), false, false);
    return __uint_as_float(r[0]) + __uint_as_float(r[1]);
}

// normalise one row held by 8 lanes (16 channels each), C = 128
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

// LDS layout (bytes).  Fixed part: the attention output / LN tile and the token -> row table of the workgroup's 32 queries.
// Region R is used twice: by the attention phase (V^T, bias columns, key tables) and then by the chain (y, hidden, biases).
#ifndef COBEVT_STAGE_VTR
#define COBEVT_STAGE_VTR 1
#endif
struct StageLds {
    int vstr, vt, bias, brows, ktab, kmadd, kterm, qterm, attn_bytes, total;
    __host__ __device__ StageLds(int nkp, int bias_rows) {
        vstr = nkp * 2 + 16;                        // V^T row: nkp keys + 16 B (row r starts 144 B mod 256 later: conflict-free b128)
        if ((vstr % 256) != 144 && (vstr % 256) != 112) vstr += ((144 - (vstr % 256)) + 256) % 256;
        brows = (bias_rows + 3) & ~3;
        vt = 0;
        // COBEVT_STAGE_VTR: V of a head as the row-major image [key][32 dh] (64 B per key) read through ds_read_b64_tr_b16
        bias = vt + (COBEVT_STAGE_VTR ? 4 * nkp * 64 : 4 * 32 * vstr);
        ktab = bias + kBiasCopyBytes;                   // the bias image is copied whole (unconditional 16-byte pieces)
        kmadd = ktab + nkp * 4;
        kterm = kmadd + nkp * 4;
        qterm = kterm + nkp * 4;
        attn_bytes = qterm + kRows * 4;
        const int chain_bytes = kRows * kRow + kRows * kHRow + kBiasFloats * 4;
        total = kRows * kRow + kRows * 4 + (attn_bytes > chain_bytes ? attn_bytes : chain_bytes);
    }
};

// NPASS: 128-column passes over the hidden layer.  NT: 32-key tiles (keys padded to NT * 32 = p.NKP).
template <int NPASS, int NT>
__global__ __launch_bounds__(kThreads, 1) void swap_stage_kernel(SwapStageParams p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const StageLds L(p.NKP, p.bias_rows);
    unsigned char* As = smem;                                   // [32][272]
    int* qrow = (int*)(smem + kRows * kRow);                    // [32] global row of query token i, -1 = past the group
    unsigned char* R = smem + kRows * kRow + kRows * 4;
    unsigned char* Vt = R + L.vt;
    float* biasl = (float*)(R + L.bias);                        // [4 heads][brows], base-2 domain
    int* ktab = (int*)(R + L.ktab);
    float* kmadd = (float*)(R + L.kmadd);
    int* kterm = (int*)(R + L.kterm);
    int* qterm = (int*)(R + L.qterm);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int b = blockIdx.y;
    const int ngroups = p.map.X * p.map.Y;
    // workgroups of one group are `ngroups` apart in dispatch order: for the usual multiple-of-8 group counts they share an XCD,
    // whose L2 then serves the group's K / V rows to all of its query slices
    const int grp = blockIdx.x % ngroups, split = blockIdx.x / ngroups;
    const int q0 = split * kRows;
    constexpr int C = 128;
    const int ld = 3 * C;

    STAGE_MARK(0);
    // ---- chain bias vectors: fetched now (registers), written to LDS when region R changes hands
    float cb[kBiasFloats / kThreads];
#pragma unroll
    for (int it = 0; it < kBiasFloats / kThreads; ++it) {
        const int i = tid + it * kThreads;
        const float* src = i < kB1 ? p.bp : i < kB2 ? p.b1 : i < kBn ? p.b2 : p.bn;
        const int j = i < kB1 ? i : i < kB2 ? i - kB1 : i < kBn ? i - kB2 : i - kBn;
        const int n = i < kB1 ? C : i < kB2 ? p.Hd : i < kBn ? C : p.Nn;
        const bool keep = (src != nullptr) & (j < n);
        const float v = (src ? src : p.b1)[keep ? j : 0];
        cb[it] = keep ? v : 0.f;
    }

    // ---- tables: key token -> row / additive mask / bias key term; query token -> row / bias query term; bias columns.
    // Everything is straight-line code with a compile-time trip count: a rolled loop with a global load inside makes the compiler
    // drain ALL outstanding loads at every iteration (vmcnt(0) at the back edge), which is what the first version's s_memtime
    // trace showed - tables 9.6k + V^T staging 15.3k of a workgroup's 47k cycles, spent in a dozen serialised round trips.
    {
        constexpr int KIT = (NT * 32 + kThreads - 1) / kThreads;
        int row[KIT], info[KIT];
        float mval[KIT];
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int tk = tid + it * kThreads;
            const TokCoord kc = tok_coord(p.map, tk < p.NK ? tk : 0);
            row[it] = tk < p.NK ? (int)tok_row(p.map, b, grp, kc) : -1;
            info[it] = tk < p.NK ? 4 * rel_bias_key_term(p.map, kc) : 0;
            int ph, pw;
            tok_pixel(p.map, grp, kc, ph, pw);
            // unconditional load from a valid address (the key mask, or - without one - the bias table as a stand-in)
            const float* mp = p.mask ? p.mask + (((size_t)b * p.map.HH + ph) * p.map.WW + pw) * p.map.ncam + kc.cam : p.bias_table;
            mval[it] = *mp;
        }
        // the head columns of the bias table, [4][brows] fp32 already scaled by log2(e) and zero-padded to 40 KB on the host
        // (>= 4 heads x 2475 rows: 6 agents, 8 x 8 windows): unconditional 16-byte copies, one batch - with a bounds test the
        // compiler fuses load and store into one branch per piece and waits for each load on its own
        constexpr int BIT = kBiasCopyIters;
        uint4 bv[BIT];
#pragma unroll
        for (int u = 0; u < BIT; ++u) bv[u] = ((const uint4*)p.bias_table)[u * kThreads + tid];
        if (tid < kRows) {
            const int t = q0 + tid;
            const bool ok = t < p.NK;
            const TokCoord qc = tok_coord(p.map, ok ? t : 0);
            qrow[tid] = ok ? (int)tok_row(p.map, b, grp, qc) : -1;
            qterm[tid] = 4 * rel_bias_query_term(p.map, p.bias_L, qc);
        }
#pragma unroll
        for (int it = 0; it < KIT; ++it) {
            const int tk = tid + it * kThreads;
            if (tk < NT * 32) {
                ktab[tk] = row[it];
                kmadd[tk] = (row[it] >= 0 && (!p.mask || mval[it] != 0.f)) ? 0.f : -INFINITY;
                kterm[tk] = info[it];
            }
        }
#pragma unroll
        for (int u = 0; u < BIT; ++u) ((uint4*)biasl)[u * kThreads + tid] = bv[u];
    }
    __syncthreads();
    STAGE_MARK(1);

    // ================================ attention: wave = head ================================
    const int head = wave;
    // Q and every K fragment of this wave's (head, 32 queries): requested now, in flight while V^T is transposed into LDS
    uint4 qf0, qf1, ka[NT][2];
    {
        const int qr = qrow[ql];
        const bf16_t* qp = p.qkv + (size_t)(qr < 0 ? 0 : qr) * ld + head * 32 + h * 8;
        qf0 = *(const uint4*)(qp);
        qf1 = *(const uint4*)(qp + 16);
        const bf16_t* kbase = p.qkv + C + head * 32 + h * 8;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int kr = ktab[t * 32 + ql];
            const bf16_t* kp = kbase + (size_t)(kr < 0 ? 0 : kr) * ld;
            ka[t][0] = *(const uint4*)(kp);
            ka[t][1] = *(const uint4*)(kp + 16);
        }
    }
    {
        const bf16_t* vbase = p.qkv + 2 * C + head * 32;
        constexpr int VIT = NT * 2;                         // NT * 32 keys x 4 dh quads / 2 keys per item / 64 lanes
#if COBEVT_STAGE_VTR
        // ---- V of this head -> LDS as it is: [key][32 dh], item = (key, 16-byte piece) - one 16-byte gather and one ds_write_b128 per
        // item where the transposed image took two 8-byte gathers and four 4-byte writes; the PV operand comes out of it through the
        // transpose read (attn_common.hpp read_vt16).  Rows of padded keys: row 0, finite, weight exactly 0 (as below).
        unsigned char* vth = Vt + head * p.NKP * 64;
        {
            int rk[VIT];
#pragma unroll
            for (int u = 0; u < VIT; ++u) rk[u] = ktab[(u * 64 + lane) >> 2];
            STAGE_MARK(8);
            uint4 vv[VIT];
#pragma unroll
            for (int u = 0; u < VIT; ++u) vv[u] = *(const uint4*)(vbase + (size_t)max(rk[u], 0) * ld + (lane & 3) * 8);
            STAGE_MARK(9);
#ifdef COBEVT_STAGE_TRACE
            __builtin_amdgcn_s_waitcnt(0x0f70);
            STAGE_MARK(10);
#endif
#pragma unroll
            for (int u = 0; u < VIT; ++u) *(uint4*)(vth + (u * 64 + lane) * 16) = vv[u];
        }
#else
        unsigned char* vth = Vt + head * 32 * L.vstr;
        // ---- V^T of this head -> LDS: item = (key pair, dh quad); 16-key blocks in the score registers' key order (perm16)
        // Rows of padded keys are read from row 0 and NOT zeroed: their probabilities are exactly 0 (additive -inf) and row 0 holds
        // finite values, so they add nothing - while a `row >= 0 ? load : 0` select makes hipcc put the load under a branch
        // (s_cbranch_execz + a full lgkmcnt / vmcnt drain per item: 15k of the first version's 47k cycles).  Two 8-byte gathers per
        // (key pair, dh quad) item; one 16-byte gather per key + a DPP exchange of halves between the pair's lanes was measured too:
        // the issue time of the gathers did not change (4.2k cycles either way: it goes by lanes, not by instructions) and the
        // exchange made the write phase 1.7k cycles longer
        int2 rr[VIT];
#pragma unroll
        for (int u = 0; u < VIT; ++u) rr[u] = *(const int2*)(ktab + 2 * ((u * 64 + lane) >> 3));
        STAGE_MARK(8);
        uint2 v0[VIT], v1[VIT];
#pragma unroll
        for (int u = 0; u < VIT; ++u) {
            const int dq = lane & 7;
            v0[u] = *(const uint2*)(vbase + (size_t)max(rr[u].x, 0) * ld + dq * 4);
            v1[u] = *(const uint2*)(vbase + (size_t)max(rr[u].y, 0) * ld + dq * 4);
        }
        STAGE_MARK(9);
#ifdef COBEVT_STAGE_TRACE
        __builtin_amdgcn_s_waitcnt(0x0f70);      // probe builds: vmcnt(0) - when did the last K / V load return?
        STAGE_MARK(10);
#endif
#pragma unroll
        for (int u = 0; u < VIT; ++u) {
            const int item = u * 64 + lane;
            const int kp = item >> 3, dq = item & 7;
            const int pos = ((2 * kp) & ~15) | perm16((2 * kp) & 15);        // even key of the pair; its partner sits at pos + 1
            const uint32_t a[2] = {v0[u].x, v0[u].y}, c[2] = {v1[u].x, v1[u].y};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int dh = dq * 4 + e;
                const uint32_t lo = (a[e >> 1] >> ((e & 1) * 16)) & 0xffffu, hi = (c[e >> 1] >> ((e & 1) * 16)) & 0xffffu;
                *(uint32_t*)(vth + dh * L.vstr + pos * 2) = lo | (hi << 16);
            }
        }
#endif
    }
    STAGE_MARK(11);
    __syncthreads();          // (a wave only reads its own head's V^T; the barrier just keeps the hand-over free of ordering assumptions)
    STAGE_MARK(2);
    {
#if COBEVT_STAGE_VTR
        const unsigned char* vtr = Vt + head * p.NKP * 64 + (4 * h + ((lane & 15) >> 2)) * 64 + ((lane >> 4) & 1) * 32 + (lane & 3) * 8;
#else
        unsigned char* vth = Vt + head * 32 * L.vstr;
#endif

        const unsigned char* bias_qp = (const unsigned char*)(biasl + head * L.brows) + qterm[ql];
        const float sl2 = p.scale * kLog2e;
#if !COBEVT_STAGE_VTR
        const unsigned char* vrow = vth + ql * L.vstr + h * 16;
#endif

        // Two-pass softmax over the WHOLE key set (NT * 32 <= 384 keys = NT * 16 score registers per lane; a workgroup owns its CU,
        // so a lane may use ~450 VGPRs): every K fragment load, every score MFMA and every bias gather is independent of the
        // others - the first version walked the key tiles with an online softmax, i.e. NT dependent
        // load -> MFMA -> gather -> max -> exp -> MFMA chains of ~2k cycles each (26 us per launch, no faster than the two
        // launches it replaced).
        f32x16 s[NT];
#pragma unroll
        for (int t = 0; t < NT; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) s[t][r] = 0.f;
            mfma_kgroup<bf16_t>(ka[t][0], qf0, s[t]);       // S^T tile: rows = keys t*32 + acc_row(r), column = this lane's query
            mfma_kgroup<bf16_t>(ka[t][1], qf1, s[t]);
        }
        // bias + mask: per tile 4 x (kmadd, kterm) 16-byte reads, then 16 dependent 4-byte gathers.  Software-pipelined by hand -
        // tile t's math runs under tile t+1's gathers and tile t+2's table reads (as written first, every tile paid two exposed
        // LDS round trips: "read, wait, gather, wait, add")
        float mloc = -INFINITY;
        f32x4 madd[2][4];
        uint4 kt[2][4];
        float bg[2][16];
        auto read_tables = [&](int t, int slot) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int kb = t * 32 + 8 * g + 4 * h;
                madd[slot][g] = *(const f32x4*)(kmadd + kb);
                kt[slot][g] = *(const uint4*)(kterm + kb);
            }
        };
        auto gather = [&](int slot) {
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                bg[slot][4 * g] = *(const float*)(bias_qp - kt[slot][g].x);
                bg[slot][4 * g + 1] = *(const float*)(bias_qp - kt[slot][g].y);
                bg[slot][4 * g + 2] = *(const float*)(bias_qp - kt[slot][g].z);
                bg[slot][4 * g + 3] = *(const float*)(bias_qp - kt[slot][g].w);
            }
        };
        read_tables(0, 0);
        if (NT > 1) read_tables(1, 1);
        gather(0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int cur = t & 1, nxt = cur ^ 1;
            float b_[16];
            f32x4 m_[4];
#pragma unroll
            for (int r = 0; r < 16; ++r) b_[r] = bg[cur][r];
#pragma unroll
            for (int g = 0; g < 4; ++g) m_[g] = madd[cur][g];
            if (t + 1 < NT) gather(nxt);                     // kt[nxt] was requested one step ago
            if (t + 2 < NT) read_tables(t + 2, cur);         // (kt[cur] / madd[cur] are consumed above)
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int g = 0; g < 4; ++g) {                    // registers 4g..4g+3 <-> keys kb..kb+3
                s[t][4 * g] = fmaf(s[t][4 * g], sl2, b_[4 * g] + m_[g].x);
                s[t][4 * g + 1] = fmaf(s[t][4 * g + 1], sl2, b_[4 * g + 1] + m_[g].y);
                s[t][4 * g + 2] = fmaf(s[t][4 * g + 2], sl2, b_[4 * g + 2] + m_[g].z);
                s[t][4 * g + 3] = fmaf(s[t][4 * g + 3], sl2, b_[4 * g + 3] + m_[g].w);
                mloc = fmaxf(fmaxf(mloc, fmaxf(s[t][4 * g], s[t][4 * g + 1])), fmaxf(s[t][4 * g + 2], s[t][4 * g + 3]));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        STAGE_MARK(12);
        const float m_all = xor32_max(mloc);
        const float m_safe = (m_all == -INFINITY) ? 0.f : m_all;
        float l_run = 0.f;
        f32x16 ot;
#pragma unroll
        for (int r = 0; r < 16; ++r) ot[r] = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            float e[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) { e[r] = __builtin_amdgcn_exp2f(s[t][r] - m_safe); l_run += e[r]; }
#pragma unroll
            for (int u = 0; u < 2; ++u) {                    // k-block u = keys t*32 + 16u .. +15 (registers 8u .. 8u+7 of both halves)
                uint4 pb;
                pb.x = pack_bf2(e[8 * u + 0], e[8 * u + 1]);
                pb.y = pack_bf2(e[8 * u + 2], e[8 * u + 3]);
                pb.z = pack_bf2(e[8 * u + 4], e[8 * u + 5]);
                pb.w = pack_bf2(e[8 * u + 6], e[8 * u + 7]);
#if COBEVT_STAGE_VTR
                const uint4 va = read_vt16(vtr + (t * 32 + u * 16) * 64);
#else
                const uint4 va = *(const uint4*)(vrow + (t * 2 + u) * 32);
#endif
                mfma_kgroup<bf16_t>(va, pb, ot);             // O^T += V^T . P^T : rows = dh, column = query
            }
        }
        const float inv = 1.0f / xor32_sum(l_run);           // an all-masked row yields NaN like the reference softmax
#pragma unroll
        for (int g4 = 0; g4 < 4; ++g4) {                     // registers 4g4.. <-> dh 8 g4 + 4h + 0..3 of head `head`
            uint2 w;
            w.x = pack_bf2(ot[4 * g4 + 0] * inv, ot[4 * g4 + 1] * inv);
            w.y = pack_bf2(ot[4 * g4 + 2] * inv, ot[4 * g4 + 3] * inv);
            *(uint2*)(As + ql * kRow + (head * 32 + 8 * g4 + 4 * h) * 2) = w;
        }
    }

    STAGE_MARK(3);
    // ================================ row chain on the 32 x 128 tile in As ================================
    unsigned char* Ys = R;
    unsigned char* Hs = R + kRows * kRow;
    float* sbw = (float*)(R + kRows * kRow + kRows * kHRow);
    const float* sb = sbw;
    const int wn = wave;                              // 32-column tile of a 128-column panel
    const int row = ql;                               // this lane's row of the tile in every MFMA result
    const int grow = qrow[row];                       // (written before the first barrier)
    const bool row_ok = grow >= 0;

    auto load_frags = [&](uint4 (&f)[8], const uint4* w, int tile, int nkg, int kg0) {
        const uint4* src = w + ((size_t)tile * nkg + kg0) * 64 + lane;
#pragma unroll
        for (int g = 0; g < 8; ++g) f[g] = src[g * 64];
        __builtin_amdgcn_sched_barrier(0);
    };
    f32x16 acc;
    auto mma = [&](const unsigned char* A, int a_off, const uint4 (&f)[8], bool zero) {
        if (zero) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const uint4 af = *(const uint4*)(A + a_off + g * 32);
            mfma_kgroup<bf16_t>(f[g], af, acc);        // D = W . X^T : acc register r <-> column acc_row(r), lane <-> row
        }
    };
    const int cbase = wn * 32 + 4 * h;
    auto bias4 = [&](int table, int col0) { return *(const float4*)(sb + table + col0); };
    auto pack4 = [&](float x, float y, float z, float w) { return make_uint2(pack_bf2(x, y), pack_bf2(z, w)); };

    uint4 fa[8], fb[8];
    const int abase = row * kRow + h * 16;
    load_frags(fa, p.wp, wn, 8, 0);
    uint2 skp[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        skp[k] = make_uint2(0, 0);
        if (row_ok) skp[k] = *(const uint2*)(p.x + (size_t)grow * C + cbase + 8 * k);
    }
    __syncthreads();                                  // As complete; region R changes hands (V^T / tables are dead)
#pragma unroll
    for (int it = 0; it < kBiasFloats / kThreads; ++it) sbw[tid + it * kThreads] = cb[it];

    // ---- phase A: y = a . Wp^T + bp + x -> Ys
    mma(As, abase, fa, true);
    load_frags(fb, p.w1, wn, 8, 0);
    __syncthreads();                                  // bias table visible
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 bv = bias4(kBp, col0);
        const float v0 = acc[4 * k] + bv.x + bf2f(skp[k].x & 0xffff), v1 = acc[4 * k + 1] + bv.y + bf2f(skp[k].x >> 16);
        const float v2 = acc[4 * k + 2] + bv.z + bf2f(skp[k].y & 0xffff), v3 = acc[4 * k + 3] + bv.w + bf2f(skp[k].y >> 16);
        *(uint2*)(Ys + row * kRow + col0 * 2) = pack4(v0, v1, v2, v3);
    }
    __syncthreads();                                  // Ys complete; As free
    STAGE_MARK(4);

    // ---- phase B: x_hat = normalise(y) -> As ; 8 threads per row, 16 channels each
    if (NPASS == 2) load_frags(fa, p.w1, 4 + wn, 8, 0);
    else load_frags(fa, p.w2, wn, p.Hdp / 16, 0);
    {
        const int r = tid >> 3, sub = tid & 7;
        float v[16];
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRow + sub * 32), v);
        chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r * kRow + sub * 32 + 16), v + 8);
        normalise128(v, p.eps1);
        *(uint4*)(As + r * kRow + sub * 32) = f32_to_chunk<bf16_t>(v);
        *(uint4*)(As + r * kRow + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
    }
    __syncthreads();

    // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> Hs
    auto hidden_epilogue = [&](int pass) {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 bv = bias4(kB1, col0);
            uint2 o = make_uint2(0, 0);
            if (col0 < p.Hd) o = pack4(gelu_bf16(acc[4 * k] + bv.x), gelu_bf16(acc[4 * k + 1] + bv.y),
                                       gelu_bf16(acc[4 * k + 2] + bv.z), gelu_bf16(acc[4 * k + 3] + bv.w));
            *(uint2*)(Hs + row * kHRow + col0 * 2) = o;
        }
    };
    const int hbase = row * kHRow + h * 16;
    if (NPASS == 2) {
        mma(As, abase, fb, true);
        load_frags(fb, p.w2, wn, p.Hdp / 16, 0);
        hidden_epilogue(0);
        mma(As, abase, fa, true);
        load_frags(fa, p.w2, wn, p.Hdp / 16, 8);
        hidden_epilogue(1);
        __syncthreads();
        // ---- phase D: z = hidden . W2^T
        mma(Hs, hbase, fb, true);
        load_frags(fb, p.wn ? p.wn : p.w1, wn, 8, 0);
        mma(Hs, hbase + 256, fa, false);
    } else {
        mma(As, abase, fb, true);
        load_frags(fb, p.wn ? p.wn : p.w1, wn, 8, 0);
        hidden_epilogue(0);
        __syncthreads();
        mma(Hs, hbase, fa, true);                     // Hdp = 128: the fragment table is zero past Hd
    }
    __syncthreads();                                  // Hs no longer read: reuse it as the fp32 staging of z
    float* stage = (float*)Hs;
    constexpr int SROW = kHRow / 4;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int col0 = cbase + 8 * k;
        const float4 bv = bias4(kB2, col0);
        const uint2 y = *(const uint2*)(Ys + row * kRow + col0 * 2);
        *(float4*)(stage + row * SROW + col0) =
            make_float4(acc[4 * k] + bv.x + bf2f(y.x & 0xffff), acc[4 * k + 1] + bv.y + bf2f(y.x >> 16),
                        acc[4 * k + 2] + bv.z + bf2f(y.y & 0xffff), acc[4 * k + 3] + bv.w + bf2f(y.y >> 16));
    }
    const int npn = p.wn ? (p.Nn + 127) / 128 : 0;
    __syncthreads();
    STAGE_MARK(5);

    // ---- phase E: coalesced 16-byte stores of the rows (scattered through the token -> row table)
    {
        const int r = tid >> 3, sub = tid & 7;
        const int gr = qrow[r];
        float v[16];
#pragma unroll
        for (int e = 0; e < 16; e += 4) {
            const float4 t = *(const float4*)(stage + r * SROW + sub * 16 + e);
            v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
        }
        uint4 o[2];
        o[0] = f32_to_chunk<bf16_t>(v);
        o[1] = f32_to_chunk<bf16_t>(v + 8);
        if (gr >= 0) {
            *(uint4*)(p.out + (size_t)gr * C + sub * 16) = o[0];
            *(uint4*)(p.out + (size_t)gr * C + sub * 16 + 8) = o[1];
        }
        if (!npn) { STAGE_MARK(6); STAGE_MARK(7); return; }
        // ---- phase F: A operand of the next to_qkv = LayerNorm(out rows as stored) -> As
        chunk_to_f32<bf16_t>(o[0], v);
        chunk_to_f32<bf16_t>(o[1], v + 8);
        normalise128(v, p.eps_next);
        *(uint4*)(As + r * kRow + sub * 32) = f32_to_chunk<bf16_t>(v);
        *(uint4*)(As + r * kRow + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
    }
    __syncthreads();
    STAGE_MARK(6);
    auto next_pass = [&](int pass, const uint4 (&cur)[8], uint4 (&nxt)[8]) {
        if (pass + 1 < npn) load_frags(nxt, p.wn, (pass + 1) * 4 + wn, 8, 0);
        mma(As, abase, cur, true);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = pass * 128 + cbase + 8 * k;
            const float4 bv = bias4(kBn, col0);
            *(uint2*)(Ys + row * kRow + (cbase + 8 * k) * 2) = pack4(acc[4 * k] + bv.x, acc[4 * k + 1] + bv.y, acc[4 * k + 2] + bv.z,
                                                                     acc[4 * k + 3] + bv.w);
        }
        __syncthreads();
        {
            const int r = tid >> 3, sub = tid & 7;
            const int gr = qrow[r];
            if (gr >= 0) {
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c0 = pass * 128 + sub * 16 + j * 8;
                    if (c0 < p.Nn) *(uint4*)(p.qkv_next + (size_t)gr * p.Nn + c0) = *(const uint4*)(Ys + r * kRow + sub * 32 + j * 16);
                }
            }
        }
        if (pass + 1 < npn) __syncthreads();
    };
    for (int pass = 0; pass < npn; pass += 2) {
        next_pass(pass, fb, fa);
        if (pass + 1 < npn) next_pass(pass + 1, fa, fb);
    }
    STAGE_MARK(7);
}

}  // namespace
}  // namespace cobevt

#ifdef COBEVT_STAGE_TRACE
extern "C" int cobevt_stage_trace_read(unsigned long long* dst, int n) {
    return hipMemcpyFromSymbol(dst, HIP_SYMBOL(cobevt::g_stage_trace), sizeof(unsigned long long) * (size_t)n, 0, hipMemcpyDeviceToHost) == hipSuccess ? 0 : 1;
}
#endif

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_swap_fusion_stage(const void* qkv, const void* x, void* out, void* qkv_next, const int* map,
                                        const float* bias_table, const float* mask, const void* wp, const float* bp,
                                        const void* w1, const float* b1, const void* w2, const float* b2, const void* wn,
                                        const float* bn, const int* dims, float scale, float eps1, float eps_next,
                                        hipStream_t stream) {
    // dims: [dtype, B, C, heads, Hd, Hdp, Nn, bias_rows, bias_L]
    if (!qkv || !x || !out || !map || !bias_table || !wp || !w1 || !b1 || !w2 || !b2 || !dims) return COBEVT_ERR_ARG;
    if ((wn == nullptr) != (qkv_next == nullptr)) return COBEVT_ERR_ARG;
    if (dims[0] != 0) return COBEVT_ERR_UNSUPPORTED;           // bf16 only; fp32 mode runs attention and the GEMMs separately
    SwapStageParams p;
    p.qkv = (const bf16_t*)qkv; p.x = (const bf16_t*)x; p.out = (bf16_t*)out; p.qkv_next = (bf16_t*)qkv_next;
    p.map = read_map(map);
    p.B = dims[1];
    p.scale = scale;
    p.bias_table = bias_table; p.bias_rows = dims[7]; p.bias_L = dims[8];
    p.mask = mask;
    p.wp = (const uint4*)wp; p.bp = bp; p.w1 = (const uint4*)w1; p.b1 = b1; p.w2 = (const uint4*)w2; p.b2 = b2;
    p.wn = (const uint4*)wn; p.bn = bn;
    p.Hd = dims[4]; p.Hdp = dims[5]; p.Nn = dims[6];
    p.eps1 = eps1; p.eps_next = eps_next;
    if (!map_ok(p.map) || p.map.mode > 1) return COBEVT_ERR_SHAPE;
    if (dims[2] != 128 || dims[3] != 4) return COBEVT_ERR_UNSUPPORTED;     // 128 channels = 4 heads of 32: one wave per head
    if (p.B < 1 || p.B > 65535 || p.Hd < 8 || p.Hd > 256 || p.Hd % 8 || p.Hdp % 128 || p.Hdp < p.Hd || p.Hdp > 256) return COBEVT_ERR_SHAPE;
    if (wn && (p.Nn < 8 || p.Nn % 8 || p.Nn > 384)) return COBEVT_ERR_SHAPE;
    p.NK = p.map.ncam * p.map.w1 * p.map.w2;
    const int nt_need = (p.NK + 31) / 32;
    const int nt = nt_need <= 4 ? 4 : nt_need <= 6 ? 6 : nt_need <= 8 ? 8 : nt_need <= 10 ? 10 : 12;     // built tile counts
    if (nt_need > 12) return COBEVT_ERR_UNSUPPORTED;
    p.NKP = nt * 32;
    const int want_rows = (2 * p.bias_L - 1) * (2 * p.map.w1 - 1) * (2 * p.map.w2 - 1);
    if (p.bias_L != p.map.ncam || p.bias_rows != want_rows) return COBEVT_ERR_SHAPE;
    if (4 * ((p.bias_rows + 3) & ~3) * 4 > kBiasCopyBytes) return COBEVT_ERR_UNSUPPORTED;
    if ((long)p.B * p.map.ncam * p.map.HH * p.map.WW >= 0x7fffffffL / 384) return COBEVT_ERR_SHAPE;
    const StageLds L(p.NKP, p.bias_rows);
    if (L.total > 160 * 1024) return COBEVT_ERR_UNSUPPORTED;
    p.nsplit = (p.NK + kRows - 1) / kRows;
    const dim3 grid((unsigned)(p.map.X * p.map.Y * p.nsplit), (unsigned)p.B);
#define COBEVT_STAGE_LAUNCH(NP_, NT_)                                                                                               \
    do {                                                                                                                             \
        static cobevt::PerDeviceOnce once;                                                                                           \
        if (once.first())                                                                                                            \
            (void)hipFuncSetAttribute((const void*)swap_stage_kernel<NP_, NT_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024); \
        hipLaunchKernelGGL((swap_stage_kernel<NP_, NT_>), grid, dim3(kThreads), (size_t)L.total, stream, p);                        \
    } while (0)
#define COBEVT_STAGE_NT(NP_)                                                                                                         \
    switch (nt) {                                                                                                                    \
        case 4: COBEVT_STAGE_LAUNCH(NP_, 4); break;                                                                                  \
        case 6: COBEVT_STAGE_LAUNCH(NP_, 6); break;                                                                                  \
        case 8: COBEVT_STAGE_LAUNCH(NP_, 8); break;                                                                                  \
        case 10: COBEVT_STAGE_LAUNCH(NP_, 10); break;                                                                                \
        default: COBEVT_STAGE_LAUNCH(NP_, 12); break;                                                                                \
    }
    if (p.Hd > 128) { COBEVT_STAGE_NT(2) } else { COBEVT_STAGE_NT(1) }
#undef COBEVT_STAGE_NT
#undef COBEVT_STAGE_LAUNCH
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
