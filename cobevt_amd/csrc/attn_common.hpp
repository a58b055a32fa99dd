// Shared declarations of the gathered-attention kernels (attention.hip: streaming K/V tiles; attention_resident.hip: the
// K / V of a window held in LDS for all of its query tiles): token maps = the einops window / grid partitions as index
// arithmetic, the relative-position index split, the kernel parameter block.
#pragma once
#include "common.hpp"

namespace cobevt {

struct TokMap {
    int mode;  // 0 window partition, 1 grid partition, 2 rows already stored window-partitioned
    int ncam;  // cameras / agents concatenated inside a window
    int HH, WW;
    int w1, w2;
    int X, Y;  // windows along H and W (HH == X*w1, WW == Y*w2)
};

struct TokCoord { int cam, i, j; };

__device__ __forceinline__ TokCoord tok_coord(const TokMap& m, int t) {
    const int ws = m.w1 * m.w2;
    TokCoord c;
    c.cam = t / ws;
    const int rem = t - c.cam * ws;
    c.i = rem / m.w2;
    c.j = rem - c.i * m.w2;
    return c;
}

// (ph, pw) pixel of the token in the un-partitioned map; reference fax_modules.py:399-404 (window),
// :420-424 (grid: 'b n (w1 x) (w2 y) d -> b n x y w1 w2 d')
__device__ __forceinline__ void tok_pixel(const TokMap& m, int l, const TokCoord& c, int& ph, int& pw) {
    const int x = l / m.Y, y = l - x * m.Y;
    if (m.mode == 1) { ph = c.i * m.X + x; pw = c.j * m.Y + y; }
    else { ph = x * m.w1 + c.i; pw = y * m.w2 + c.j; }
}

__device__ __forceinline__ size_t tok_row(const TokMap& m, int b, int l, const TokCoord& c) {
    if (m.mode == 2) {
        return (((size_t)(b * m.ncam + c.cam) * (m.X * m.Y) + l) * m.w1 + c.i) * m.w2 + c.j;
    }
    int ph, pw;
    tok_pixel(m, l, c, ph, pw);
    return ((size_t)(b * m.ncam + c.cam) * m.HH + ph) * m.WW + pw;
}

// tok_row as an affine function of (camera, i, j) for one (batch b, window l): row = cam * cs + i * si + j * sj + base, all int32
// (the launchers reject maps with 2^31 or more rows).  The window's origin - one integer division of l by m.Y - is paid once per
// workgroup instead of once per token; the resident kernel's prologue evaluates a dozen token rows per thread.
struct RowAffine { int cs, si, sj, base; int ph0, pw0, pi, pj; };   // (ph, pw) = (ph0 + i * pi, pw0 + j * pj): the token's pixel (modes 0 / 1)
__device__ __forceinline__ RowAffine row_affine(const TokMap& m, int b, int l) {
    RowAffine a;
    if (m.mode == 2) {
        const int ws = m.w1 * m.w2;
        a.cs = m.X * m.Y * ws; a.si = m.w2; a.sj = 1;
        a.base = (b * m.ncam * (m.X * m.Y) + l) * ws;
        a.ph0 = a.pw0 = 0; a.pi = a.pj = 1;
        return a;
    }
    const int x = l / m.Y, y = l - x * m.Y;
    a.cs = m.HH * m.WW;
    if (m.mode == 1) { a.si = m.X * m.WW; a.sj = m.Y; a.base = (b * m.ncam * m.HH + x) * m.WW + y; a.ph0 = x; a.pw0 = y; a.pi = m.X; a.pj = m.Y; }
    else { a.si = m.WW; a.sj = 1; a.base = (b * m.ncam * m.HH + x * m.w1) * m.WW + y * m.w2; a.ph0 = x * m.w1; a.pw0 = y * m.w2; a.pi = a.pj = 1; }
    return a;
}
__device__ __forceinline__ int row_of(const RowAffine& a, const TokCoord& c) { return c.cam * a.cs + c.i * a.si + c.j * a.sj + a.base; }

// Relative-position bias index split into a query term and a key term (the table index is linear in the coordinates):
//   index = ((dl + L-1)(2 w1 - 1) + (di + w1-1))(2 w2 - 1) + (dj + w2-1),  d = query - key coordinate
// swap_fusion_modules.py:55-85 (3-D, agent extent L) and fax_modules.py:121-130 (2-D: L = 1, cam = 0).  Integer arithmetic that
// must be bit-exact: cobevt_attention_bias_index dumps query_term - key_term through these same two functions.
__device__ __forceinline__ int rel_bias_query_term(const TokMap& km, int bias_L, const TokCoord& qc) {
    return ((qc.cam + bias_L - 1) * (2 * km.w1 - 1) + qc.i + km.w1 - 1) * (2 * km.w2 - 1) + qc.j + km.w2 - 1;
}
__device__ __forceinline__ int rel_bias_key_term(const TokMap& km, const TokCoord& kc) {
    return (kc.cam * (2 * km.w1 - 1) + kc.i) * (2 * km.w2 - 1) + kc.j;
}

struct AttnParams {
    const void* q; const void* k; const void* v; void* out;
    int ldq, ldk, ldv, ldo;
    int qoff, koff, voff, ooff;
    TokMap qmap, kmap, omap;
    int B, L, heads, Nq, Nk;
    float scale;
    int bias_mode;            // 0 none, 1 relative-position table lookup
    const float* bias_table;  // [rows][heads]
    int bias_rows;
    int bias_L;               // agent extent of the 3-D table (1 => 2-D table)
    const float* mask;        // key mask fp32, 0 => key masked out; (B,HH,WW,ncam), or (B,L,w1,w2,ncam) for mode 2; may be null
    int mean_q;               // 0: every query token on its own; 1: per-camera query copies, outputs averaged over the cameras
                              // (fax_modules.py:243); 2: per-camera query copies, camera c's query scores camera c's keys only,
                              // ONE softmax over all cameras' keys (CVT CrossAttention, cvt_modules.py:142-153)
    float* lse;               // training forward: base-2 log-sum-exp of every query's logits, [B][L][heads][Nq] (nullable)
    int klinear;              // streaming kernel: key token tk of the (single) window is row b * Nk + tk - no key table in LDS
    // training only: nn.Dropout on the attention probabilities (FAX global attention, fax_modules.py:114,161): element (query,
    // key) of a (batch, window, head) is kept with probability 1 - drop_p and scaled by 1 / (1 - drop_p); the keep decision is a
    // counter-based hash of (drop_seed, element index), so the backward kernels regenerate the forward's mask
    float drop_p;
    unsigned drop_seed;
    // nullable device word ADDED to drop_seed: a captured training step (tools/train_graph_probe.py) bumps it inside the graph, so every
    // replay draws a new mask although the kernel arguments are frozen in the graph's nodes
    const unsigned* drop_seed_dev;
    // key split (streaming kernel, inference): the keys of a window are shared out over `ksplit` workgroups per query tile; each
    // writes its normalised partial output rows to part_out[split] (same row indexing as `out`, row stride heads * 32) and the
    // base-2 log-sum-exp of its keys to part_lse[split][row][head]; attn_ksplit_merge_kernel combines them into `out`.
    // For the launches whose grid leaves the chip idle AND whose per-query key walk is long (level-2 / global FAX attention:
    // 1024 keys, 160 workgroups) - the tile loop is one dependent round trip per iteration.
    int ksplit;
    void* part_out;
    float* part_lse;
    long part_rows;           // rows of `out` (stride between the splits' partial buffers)
};

__host__ __device__ __forceinline__ unsigned attn_mix32(unsigned x) {       // murmur3 finaliser
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
// keep decision of probability element (tq, tk) of (b, l, head): uniform 32-bit hash >= p * 2^32
__device__ __forceinline__ bool attn_keep(const AttnParams& p, int b, int l, int head, int tq, int tk) {
    const unsigned long long row = ((unsigned long long)((b * p.L + l) * p.heads + head)) * (unsigned)p.Nq + (unsigned)tq;
    const unsigned seed = p.drop_seed + (p.drop_seed_dev ? *p.drop_seed_dev : 0u);       // uniform scalar load
    const unsigned hrow = attn_mix32((unsigned)row ^ attn_mix32((unsigned)(row >> 32) ^ seed));
    const unsigned u = attn_mix32(hrow ^ ((unsigned)tk * 0x9e3779b1u));
    return (float)u * 2.3283064365386963e-10f >= p.drop_p;
}

static inline bool map_ok(const TokMap& m) {
    if (m.mode < 0 || m.mode > 2 || m.ncam < 1 || m.w1 < 1 || m.w2 < 1 || m.X < 1 || m.Y < 1) return false;
    if (m.mode != 2 && (m.HH != m.X * m.w1 || m.WW != m.Y * m.w2)) return false;
    return m.w1 < 256 && m.w2 < 256 && m.ncam < 32768;
}

static inline TokMap read_map(const int* d) {
    TokMap m;
    m.mode = d[0]; m.ncam = d[1]; m.HH = d[2]; m.WW = d[3]; m.w1 = d[4]; m.w2 = d[5]; m.X = d[6]; m.Y = d[7];
    return m;
}

// attention_resident.hip: launches the K/V-resident kernel when the problem qualifies; returns COBEVT_OK, an error code, or
// -1 when it does not apply (the caller then uses the streaming kernel).  hint: 0 auto, >0 = query split to use
int launch_attn_resident(const AttnParams& p, int qsplit_hint, hipStream_t stream);

// The A operand V^T (rows = dh, k = 16 keys) of a PV MFMA out of a ROW-MAJOR V image [key][32 dh] (64 B per key, staged with the same
// coalesced 16-byte copies as K): gfx950's transpose read.  ds_read_b64_tr_b16: every lane reads 8 bytes at its own address, then inside
// each 16-lane group lane c receives element (c & 3) of what lanes 4 j + (c >> 2) read (j = 0 .. 3) - so with lane s of a group reading
// row (s >> 2), columns 4 (s & 3) .. + 3 of a 4-key x 16-dh block, lane c ends up with column c of the four keys (checked on the
// device: tools/tr_read_probe.hip).  Lane (h, dh) of the MFMA holds k-slots 8 h .. 8 h + 7 = the keys {4 h .. 4 h + 3, 8 + 4 h .. 8 + 4 h + 3}
// of the 16-key block in the score registers' order: two reads, 8 rows apart.  Lanes 0-31 of a read cover four whole 64-byte rows =
// 256 contiguous bytes: conflict-free without a swizzle.  p = this lane's address for the block's first read.
typedef short v4s16 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint4 read_vt16(const unsigned char* p) {
    const v4s16 a = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)p);
    const v4s16 b = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s16*)(p + 8 * 64));
    const uint2 x = __builtin_bit_cast(uint2, a), y = __builtin_bit_cast(uint2, b);
    return make_uint4(x.x, x.y, y.x, y.y);
}


}  // namespace cobevt
