// The fused row-local chain of row_chain.hip for 64-channel rows with a 128-wide hidden layer - the chain behind every attention of
// the LiDAR FuseBEVT encoder (swap_fusion_modules.py:87-128,126,172-190 with input_dim 64, mlp_dim 128; base_transformer.py:102-124)
// - as PERSISTENT workgroups (gfx950, bf16):
//
//     y = a . Wp^T + bp + skip ;  z = y + ( GELU( LN(y) . W1'^T + b1' ) . W2^T + b2 ) ;  out = post-LN?(z) ;
//     next = act( LN?(out) . Wn'^T + bn' )                                              (optional: the next half's to_qkv)
//
// The generic kernel walks 128-column panels with four waves: at C = 64 two of them compute zero columns in the out-projection
// and fc2 phases, every k-group sits behind a uniform branch, and each of the 16,384 32-row workgroups of a (8, 256, 256, 64) map
// re-streams the chain's 74 KB of weight fragments from L2 (1.2 GB per launch) - 209 us per launch against ~70 us for the
// 400 MB the chain has to move (a, skip, out, next).  Here:
//   * a workgroup = 4 waves = (2 row halves) x (2 column-tile parities) of a 64-row tile; wave (wm, wn) owns the 32-column tiles
//     wn, wn + 2, .. of every GEMM for rows 32 wm .. +31, so its share of ALL weights is 32 fragments = 128 VGPRs, loaded once
//     and kept in registers while the workgroup loops over row tiles (2 workgroups per CU, 512 workgroups);
//   * the next tile's `a` and `skip` rows (8 KB each, contiguous) are requested with coalesced 16-byte loads right after the
//     current tile has been written to LDS and stay in flight under the whole chain of the current tile;
//   * phases as in row_chain.hip (y, LN(y), hidden in LDS; D = W . X^T so epilogues are 8-byte LDS traffic), six barriers per tile.
#include "row_chain.hpp"

namespace cobevt {

namespace {

constexpr int kR = 64;                   // rows per tile
constexpr int kRow = 128 + 16;           // 64 bf16 + pad: conflict-free ds_read_b128 B fragments (36-dword row stride)
constexpr int kHRow = 256 + 16;          // 128 bf16 hidden columns + pad
constexpr int kNRow = 384 + 16;          // up to 192 bf16 next-projection columns + pad; also the fp32 staging row of z (64 floats)
struct Rc64Lds {
    static constexpr int A = 0;                       // a tile, later LN(y), later the next projection's A operand
    static constexpr int S = A + kR * kRow;           // skip tile
    static constexpr int Y = S + kR * kRow;           // y tile (bf16)
    static constexpr int H = Y + kR * kRow;           // hidden tile
    static constexpr int N = H + kR * kHRow;          // fp32 staging of z, later the next projection's output tile
    static constexpr int BIAS = N + kR * kNRow;       // fp32 bias / affine table (below)
    static constexpr int BYTES = BIAS + 4 * 576;      // 72,960 B: two workgroups per CU
};
// bias table (floats): bp [0,64) b1 [64,192) b2 [192,256) post gamma [256,320) post beta [320,384) bnext [384,576)
constexpr int kBp = 0, kB1 = 64, kB2 = 192, kPg = 256, kPb = 320, kBn = 384, kBiasFloats = 576;

// normalise one 64-channel row held by 4 lanes (16 channels each)
__device__ __forceinline__ void rc64_normalise(float (&v)[16], float eps) {
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) s += v[e];
    s += __shfl_xor(s, 1, 64); s += __shfl_xor(s, 2, 64);
    const float mean = s * (1.0f / 64.0f);
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float d = v[e] - mean; q += d * d; }
    q += __shfl_xor(q, 1, 64); q += __shfl_xor(q, 2, 64);
    const float rstd = rsqrtf(q * (1.0f / 64.0f) + eps);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = (v[e] - mean) * rstd;
}

// NNT: 32-column tiles of the next projection per wave (Nn = 64 NNT; 0 = no next projection)
template <int NNT>
__global__ __launch_bounds__(256, 2) void row_chain64_kernel(RowChainParams p, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* As = smem + Rc64Lds::A;
    unsigned char* Ss = smem + Rc64Lds::S;
    unsigned char* Ys = smem + Rc64Lds::Y;
    unsigned char* Hs = smem + Rc64Lds::H;
    unsigned char* Ns = smem + Rc64Lds::N;
    float* sb = (float*)(smem + Rc64Lds::BIAS);

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int wm = wave >> 1, wn = wave & 1;
    const int row = wm * 32 + ql;                    // this lane's row of the tile in every MFMA result

    // ---- this wave's share of every weight matrix, resident in registers for the life of the workgroup
    uint4 fp[4], f1[2][4], f2[8], fn[NNT > 0 ? NNT : 1][4];
#pragma unroll
    for (int g = 0; g < 4; ++g) fp[g] = p.wp[(size_t)(wn * 8 + g) * 64 + lane];
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) f1[t][g] = p.w1[(size_t)((wn + 2 * t) * 8 + g) * 64 + lane];
#pragma unroll
    for (int g = 0; g < 8; ++g) f2[g] = p.w2[(size_t)(wn * 8 + g) * 64 + lane];
#pragma unroll
    for (int t = 0; t < NNT; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g) fn[t][g] = p.wn[(size_t)((wn + 2 * t) * 8 + g) * 64 + lane];

    // ---- bias / affine table (absent vectors -> 0; unconditional clamped loads)
    for (int i = tid; i < kBiasFloats; i += 256) {
        const float* src = i < kB1 ? p.bp : i < kB2 ? p.b1 : i < kPg ? p.b2 : i < kPb ? p.post_g : i < kBn ? p.post_b : p.bn;
        const int j = i < kB1 ? i : i < kB2 ? i - kB1 : i < kPg ? i - kB2 : i < kPb ? i - kPg : i < kBn ? i - kPb : i - kBn;
        const int n = i < kBn ? (i >= kB1 && i < kB2 ? 128 : 64) : p.Nn;
        const bool keep = (src != nullptr) & (j < n);
        const float val = (src ? src : p.b1)[keep ? j : 0];
        sb[i] = keep ? val : 0.f;
    }

    // tile rows are contiguous in memory: 64 rows x 128 B = 512 16-byte pieces, two per thread, fully coalesced
    uint4 ra[2], rs[2];
    auto load_tile = [&](int tile) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * 256;
            int grow = tile * kR + (idx >> 3);
            grow = grow < p.M ? grow : p.M - 1;              // tail tile: clamped (finite data, never stored)
            const size_t off = (size_t)grow * 64 + (idx & 7) * 8;
            ra[j] = *(const uint4*)(p.a + off);
            rs[j] = *(const uint4*)(p.skip + off);
        }
    };
    auto bias4 = [&](int table, int col0) { return *(const float4*)(sb + table + col0); };
    auto pack4 = [&](float x, float y, float z, float w) { return make_uint2(pack_bf2(x, y), pack_bf2(z, w)); };

    int tile = blockIdx.x;
    if (tile < ntiles) load_tile(tile);
    const int abase = row * kRow + h * 16;
    const int hbase = row * kHRow + h * 16;
    const int r4 = tid >> 2, sub = tid & 3;              // row-wise phases: 4 threads per row, 16 channels each
    for (; tile < ntiles; tile += gridDim.x) {
        const int m0 = tile * kR;
        // ---- the prefetched rows -> LDS ; the next tile's rows go in flight
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int idx = tid + j * 256;
            *(uint4*)(As + (idx >> 3) * kRow + (idx & 7) * 16) = ra[j];
            *(uint4*)(Ss + (idx >> 3) * kRow + (idx & 7) * 16) = rs[j];
        }
        __syncthreads();
        {
            const int nt = tile + (int)gridDim.x;
            load_tile(nt < ntiles ? nt : tile);                 // unconditional (clamped): the waits stay counted
        }
        __builtin_amdgcn_sched_barrier(0);

        f32x16 acc;
        // ---- phase A: y = a . Wp^T + bp + skip -> Ys
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(fp[g], *(const uint4*)(As + abase + g * 32), acc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = wn * 32 + 4 * h + 8 * k;
            const float4 b = bias4(kBp, col0);
            const uint2 s = *(const uint2*)(Ss + row * kRow + col0 * 2);
            *(uint2*)(Ys + row * kRow + col0 * 2) =
                pack4(acc[4 * k] + b.x + bf2f(s.x & 0xffff), acc[4 * k + 1] + b.y + bf2f(s.x >> 16),
                      acc[4 * k + 2] + b.z + bf2f(s.y & 0xffff), acc[4 * k + 3] + b.w + bf2f(s.y >> 16));
        }
        __syncthreads();

        // ---- phase B: x_hat = normalise(y) -> As
        {
            float v[16];
            chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r4 * kRow + sub * 32), v);
            chunk_to_f32<bf16_t>(*(const uint4*)(Ys + r4 * kRow + sub * 32 + 16), v + 8);
            rc64_normalise(v, p.eps1);
            *(uint4*)(As + r4 * kRow + sub * 32) = f32_to_chunk<bf16_t>(v);
            *(uint4*)(As + r4 * kRow + sub * 32 + 16) = f32_to_chunk<bf16_t>(v + 8);
        }
        __syncthreads();

        // ---- phase C: hidden = GELU(x_hat . W1'^T + b1') -> Hs (column tiles wn, wn + 2)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(f1[t][g], *(const uint4*)(As + abase + g * 32), acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int col0 = (wn + 2 * t) * 32 + 4 * h + 8 * k;
                const float4 b = bias4(kB1, col0);
                *(uint2*)(Hs + row * kHRow + col0 * 2) = pack4(gelu_erf(acc[4 * k] + b.x), gelu_erf(acc[4 * k + 1] + b.y),
                                                                gelu_erf(acc[4 * k + 2] + b.z), gelu_erf(acc[4 * k + 3] + b.w));
            }
        }
        __syncthreads();

        // ---- phase D: z = hidden . W2^T + b2 + y -> fp32 staging
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
        for (int g = 0; g < 8; ++g) mfma_kgroup<bf16_t>(f2[g], *(const uint4*)(Hs + hbase + g * 32), acc);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int col0 = wn * 32 + 4 * h + 8 * k;
            const float4 b = bias4(kB2, col0);
            const uint2 y = *(const uint2*)(Ys + row * kRow + col0 * 2);
            *(float4*)(Ns + row * kNRow + col0 * 4) =
                make_float4(acc[4 * k] + b.x + bf2f(y.x & 0xffff), acc[4 * k + 1] + b.y + bf2f(y.x >> 16),
                            acc[4 * k + 2] + b.z + bf2f(y.y & 0xffff), acc[4 * k + 3] + b.w + bf2f(y.y >> 16));
        }
        __syncthreads();

        // ---- phase E: optional post-LayerNorm, coalesced 16-byte stores of `out`, A operand of the next projection
        {
            float v[16];
#pragma unroll
            for (int e = 0; e < 16; e += 4) {
                const float4 t = *(const float4*)(Ns + r4 * kNRow + (sub * 16 + e) * 4);
                v[e] = t.x; v[e + 1] = t.y; v[e + 2] = t.z; v[e + 3] = t.w;
            }
            if (p.post_g) {
                rc64_normalise(v, p.eps_post);
#pragma unroll
                for (int e = 0; e < 16; e += 4) {
                    const float4 g = *(const float4*)(sb + kPg + sub * 16 + e), b = *(const float4*)(sb + kPb + sub * 16 + e);
                    v[e] = v[e] * g.x + b.x; v[e + 1] = v[e + 1] * g.y + b.y;
                    v[e + 2] = v[e + 2] * g.z + b.z; v[e + 3] = v[e + 3] * g.w + b.w;
                }
            }
            uint4 o[2];
            o[0] = f32_to_chunk<bf16_t>(v);
            o[1] = f32_to_chunk<bf16_t>(v + 8);
            if (m0 + r4 < p.M) {
                bf16_t* dst = p.out + (size_t)(m0 + r4) * 64 + sub * 16;
                *(uint4*)dst = o[0];
                *(uint4*)(dst + 8) = o[1];
            }
            if (NNT > 0) {
                if (p.next_ln) {                       // the rows exactly as stored (bf16), normalised
                    chunk_to_f32<bf16_t>(o[0], v);
                    chunk_to_f32<bf16_t>(o[1], v + 8);
                    rc64_normalise(v, p.eps_next);
                    o[0] = f32_to_chunk<bf16_t>(v);
                    o[1] = f32_to_chunk<bf16_t>(v + 8);
                }
                *(uint4*)(As + r4 * kRow + sub * 32) = o[0];
                *(uint4*)(As + r4 * kRow + sub * 32 + 16) = o[1];
            }
        }
        if (NNT > 0) {
            __syncthreads();                               // As complete; the fp32 staging is free again (it becomes the output tile)
            // ---- next projection: column tiles wn, wn + 2, .. -> Ns (bf16) -> coalesced 16-byte stores
#pragma unroll
            for (int t = 0; t < NNT; ++t) {
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
                for (int g = 0; g < 4; ++g) mfma_kgroup<bf16_t>(fn[t][g], *(const uint4*)(As + abase + g * 32), acc);
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int col0 = (wn + 2 * t) * 32 + 4 * h + 8 * k;
                    const float4 b = bias4(kBn, col0);
                    float v0 = acc[4 * k] + b.x, v1 = acc[4 * k + 1] + b.y, v2 = acc[4 * k + 2] + b.z, v3 = acc[4 * k + 3] + b.w;
                    if (p.next_act == 1) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); v2 = fmaxf(v2, 0.f); v3 = fmaxf(v3, 0.f); }
                    else if (p.next_act == 2) { v0 = gelu_erf(v0); v1 = gelu_erf(v1); v2 = gelu_erf(v2); v3 = gelu_erf(v3); }
                    *(uint2*)(Ns + row * kNRow + col0 * 2) = pack4(v0, v1, v2, v3);
                }
            }
            __syncthreads();
            constexpr int PPR = NNT * 8;                   // 16-byte pieces per row of `next`
#pragma unroll
            for (int i = 0; i < 2 * NNT; ++i) {
                const int idx = tid + i * 256;
                const int r = idx / PPR, c = idx - r * PPR;
                if (m0 + r < p.M) *(uint4*)(p.out_next + (size_t)(m0 + r) * p.Nn + c * 8) = *(const uint4*)(Ns + r * kNRow + c * 16);
            }
        }
        // (the next iteration's first LDS writes go to As / Ss, last read before the barriers above; Ns is rewritten four barriers later)
    }
}

template <int NNT> int launch64(const RowChainParams& p, hipStream_t stream) {
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) {
        (void)hipFuncSetAttribute((const void*)row_chain64_kernel<NNT>, hipFuncAttributeMaxDynamicSharedMemorySize, Rc64Lds::BYTES);
    }
    const int ntiles = (p.M + kR - 1) / kR;
    const int blocks = ntiles < 512 ? ntiles : 512;        // two workgroups per CU, each walking ntiles / 512 row tiles
    hipLaunchKernelGGL((row_chain64_kernel<NNT>), dim3((unsigned)blocks), dim3(256), Rc64Lds::BYTES, stream, p, ntiles);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace

int launch_row_chain64(const RowChainParams& p, hipStream_t stream) {
    if (p.C != 64 || p.Hd != 128 || p.Hdp != 128 || !p.skip || p.skip_rows != p.M) return -1;
    if (p.wn && (p.Nn % 64 != 0 || p.Nn > 192)) return -1;
    if (p.M < 16384) return -1;                            // small maps: the generic kernel's finer workgroups fill the chip better
    switch (p.wn ? p.Nn / 64 : 0) {
        case 0: return launch64<0>(p, stream);
        case 1: return launch64<1>(p, stream);
        case 2: return launch64<2>(p, stream);
        case 3: return launch64<3>(p, stream);
        default: return -1;
    }
}

}  // namespace cobevt
