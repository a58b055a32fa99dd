// Row-local backward kernels of the training slice (fp32): LayerNorm backward and exact-erf GELU forward / backward.
// The reference gets these from torch autograd (nn.LayerNorm / nn.GELU of base_transformer.py:102-124,
// swap_fusion_modules.py:275-279, fax_modules.py:189-191,309-313) under train_camera.py:143-179.  HBM-bound elementwise work:
// one wave per row, 16-byte accesses, the per-channel sums (dgamma, dbeta) accumulated in registers over a workgroup's rows
// and added to the global fp32 vectors once per workgroup.
#include "common.hpp"

namespace cobevt {
namespace {

constexpr int kLnMaxPerLane = 4;                         // float4 groups per lane: C <= 64 * 4 * 4 = 1024

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// y = LN(x) gamma + beta.  dx = rstd (g - mean(g) - xhat mean(g xhat)), g = dy gamma; dgamma += dy xhat; dbeta += dy.
// x / dy / dx are fp32 or bf16 each (xb / db / ob: 1 = bf16) - inside a bf16 autocast region the residual stream is fp32 while the
// gradient arriving from a projection's input gradient is bf16 (train_camera.py:157-160); the arithmetic is fp32 either way.
__device__ __forceinline__ float4 ld4(const void* p, int bf, size_t i) {
    if (bf) {
        const uint2 u = *(const uint2*)((const uint16_t*)p + i);
        return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xffff0000u), __uint_as_float(u.y << 16), __uint_as_float(u.y & 0xffff0000u));
    }
    return *(const float4*)((const float*)p + i);
}
__device__ __forceinline__ void st4(void* p, int bf, size_t i, const float4& v) {
    if (bf) *(uint2*)((uint16_t*)p + i) = make_uint2(pack_bf2(v.x, v.y), pack_bf2(v.z, v.w));
    else *(float4*)((float*)p + i) = v;
}

__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                            const float* __restrict__ gamma, void* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                            float eps, int rows_per_block, int xb, int db_, int ob) {
    __shared__ float red[2][4][1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int groups = C >> 2;                           // float4 groups per row
    float4 gam[kLnMaxPerLane], dg[kLnMaxPerLane], db[kLnMaxPerLane];
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int g = lane + 64 * i;
        gam[i] = (gamma && g < groups) ? *(const float4*)(gamma + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);
        dg[i] = make_float4(0.f, 0.f, 0.f, 0.f);
        db[i] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(rows, row0 + rows_per_block);
    const float invC = 1.f / (float)C;
    for (int row = row0 + wave; row < row1; row += 4) {
        float4 xv[kLnMaxPerLane], dv[kLnMaxPerLane];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPerLane; ++i) {
            const int g = lane + 64 * i;
            const bool ok = g < groups;
            xv[i] = ok ? ld4(x, xb, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            dv[i] = ok ? ld4(dy, db_, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
        }
        const float mean = wave_sum(s) * invC;
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPerLane; ++i) {
            if (lane + 64 * i < groups) {
                const float a = xv[i].x - mean, b = xv[i].y - mean, c = xv[i].z - mean, d = xv[i].w - mean;
                v += a * a + b * b + c * c + d * d;
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(v) * invC + eps);
        float sg = 0.f, sgx = 0.f;
#pragma unroll
        for (int i = 0; i < kLnMaxPerLane; ++i) {
            if (lane + 64 * i < groups) {
                // xv becomes xhat, dv becomes g = dy gamma
                xv[i].x = (xv[i].x - mean) * rstd; xv[i].y = (xv[i].y - mean) * rstd;
                xv[i].z = (xv[i].z - mean) * rstd; xv[i].w = (xv[i].w - mean) * rstd;
                dg[i].x += dv[i].x * xv[i].x; dg[i].y += dv[i].y * xv[i].y; dg[i].z += dv[i].z * xv[i].z; dg[i].w += dv[i].w * xv[i].w;
                db[i].x += dv[i].x; db[i].y += dv[i].y; db[i].z += dv[i].z; db[i].w += dv[i].w;
                dv[i].x *= gam[i].x; dv[i].y *= gam[i].y; dv[i].z *= gam[i].z; dv[i].w *= gam[i].w;
                sg += dv[i].x + dv[i].y + dv[i].z + dv[i].w;
                sgx += dv[i].x * xv[i].x + dv[i].y * xv[i].y + dv[i].z * xv[i].z + dv[i].w * xv[i].w;
            }
        }
        const float mg = wave_sum(sg) * invC, mgx = wave_sum(sgx) * invC;
#pragma unroll
        for (int i = 0; i < kLnMaxPerLane; ++i) {
            const int g = lane + 64 * i;
            if (g < groups) {
                float4 o;
                o.x = rstd * (dv[i].x - mg - xv[i].x * mgx); o.y = rstd * (dv[i].y - mg - xv[i].y * mgx);
                o.z = rstd * (dv[i].z - mg - xv[i].z * mgx); o.w = rstd * (dv[i].w - mg - xv[i].w * mgx);
                st4(dx, ob, (size_t)row * C + 4 * g, o);
            }
        }
    }
    if (!dgamma) return;
    // workgroup reduction of the per-channel sums, then one atomic per channel
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int g = lane + 64 * i;
        if (g < groups) {
            *(float4*)&red[0][wave][4 * g] = dg[i];
            *(float4*)&red[1][wave][4 * g] = db[i];
        }
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
        atomicAdd(dgamma + c, red[0][0][c] + red[0][1][c] + red[0][2][c] + red[0][3][c]);
        atomicAdd(dbeta + c, red[1][0][c] + red[1][1][c] + red[1][2][c] + red[1][3][c]);
    }
}

// C <= 128 (the FAX / fusion width): a row is 32 float4 groups, so a whole wave per row leaves half of it idle and a row per HALF-wave
// doubles the rows in flight (the 81,920 x 128 level-0 rows took 57 us per launch, 1.9 TB/s).  One float4 per lane, reductions inside 32 lanes.
__device__ __forceinline__ float half_wave_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// NW waves per workgroup: 16 on big inputs - every workgroup ends in 2 C atomics on the SAME words, and 1,024 four-wave workgroups made the
// 81,920-row launches 49 us for 105 MB (a serialised tail of ~1,000 atomics per word); the same waves in a quarter of the workgroups
template <int NW>
__global__ __launch_bounds__(NW * 64) void layernorm_bwd_narrow_kernel(const void* __restrict__ x, const void* __restrict__ dy,
                                                                   const float* __restrict__ gamma, void* __restrict__ dx,
                                                                   float* __restrict__ dgamma, float* __restrict__ dbeta, int rows, int C,
                                                                   float eps, int rows_per_block, int xb, int db_, int ob) {
    constexpr int HW = NW * 2;                        // half-waves = rows in flight per round
    __shared__ float red[2][HW][128];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 31, sub = lane >> 5;
    const int groups = C >> 2;
    const bool ok = g < groups;
    const float4 gam = (gamma && ok) ? *(const float4*)(gamma + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);
    float4 dg = make_float4(0.f, 0.f, 0.f, 0.f), db = dg;
    const int row0 = blockIdx.x * rows_per_block;
    const int row1 = min(rows, row0 + rows_per_block);
    const float invC = 1.f / (float)C;
    // two rows in flight per half-wave
    for (int row = row0 + wave * 2 + sub; row < row1; row += 2 * HW) {
        const int rowb = row + HW;
        const bool okb = rowb < row1;
        float4 xv[2], dv[2];
        xv[0] = ok ? ld4(x, xb, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[0] = ok ? ld4(dy, db_, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        xv[1] = (ok && okb) ? ld4(x, xb, (size_t)rowb * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        dv[1] = (ok && okb) ? ld4(dy, db_, (size_t)rowb * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float mean = half_wave_sum(xv[u].x + xv[u].y + xv[u].z + xv[u].w) * invC;
            const float a = xv[u].x - mean, b = xv[u].y - mean, c = xv[u].z - mean, d = xv[u].w - mean;
            const float rstd = 1.f / sqrtf(half_wave_sum(ok ? a * a + b * b + c * c + d * d : 0.f) * invC + eps);
            const float4 xh = make_float4(a * rstd, b * rstd, c * rstd, d * rstd);
            float4 gv = dv[u];
            if (ok && (u == 0 || okb)) {
                dg.x += gv.x * xh.x; dg.y += gv.y * xh.y; dg.z += gv.z * xh.z; dg.w += gv.w * xh.w;
                db.x += gv.x; db.y += gv.y; db.z += gv.z; db.w += gv.w;
            }
            gv.x *= gam.x; gv.y *= gam.y; gv.z *= gam.z; gv.w *= gam.w;
            const float mg = half_wave_sum(ok ? gv.x + gv.y + gv.z + gv.w : 0.f) * invC;
            const float mgx = half_wave_sum(ok ? gv.x * xh.x + gv.y * xh.y + gv.z * xh.z + gv.w * xh.w : 0.f) * invC;
            if (ok && (u == 0 || okb)) {
                const float4 o = make_float4(rstd * (gv.x - mg - xh.x * mgx), rstd * (gv.y - mg - xh.y * mgx), rstd * (gv.z - mg - xh.z * mgx),
                                             rstd * (gv.w - mg - xh.w * mgx));
                st4(dx, ob, (size_t)(u ? rowb : row) * C + 4 * g, o);
            }
        }
    }
    if (!dgamma) return;
    if (ok) {
        *(float4*)&red[0][wave * 2 + sub][4 * g] = dg;
        *(float4*)&red[1][wave * 2 + sub][4 * g] = db;
    }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NW * 64) {
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int k = 0; k < HW; ++k) { a += red[0][k][c]; b += red[1][k][c]; }
        atomicAdd(dgamma + c, a);
        atomicAdd(dbeta + c, b);
    }
}

__device__ __forceinline__ float gelu_f(float x) { return 0.5f * x * (1.f + erff(x * 0.70710678118654752f)); }
__device__ __forceinline__ float gelu_grad(float x) {
    return 0.5f * (1.f + erff(x * 0.70710678118654752f)) + x * 0.3989422804014327f * __expf(-0.5f * x * x);
}

__global__ __launch_bounds__(256) void gelu_kernel(const float* __restrict__ x, const float* __restrict__ dy, float* __restrict__ out,
                                                   long n4) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n4) return;
    const float4 v = ((const float4*)x)[i];
    float4 o;
    if (dy) {
        const float4 d = ((const float4*)dy)[i];
        o = make_float4(d.x * gelu_grad(v.x), d.y * gelu_grad(v.y), d.z * gelu_grad(v.z), d.w * gelu_grad(v.w));
    } else {
        o = make_float4(gelu_f(v.x), gelu_f(v.y), gelu_f(v.z), gelu_f(v.w));
    }
    ((float4*)out)[i] = o;
}

// bf16 storage (a bf16 autocast region: nn.GELU runs in the dtype of its input, the projection's bf16 output): 8 values per lane,
// fp32 arithmetic, one rounding of the result
__global__ __launch_bounds__(256) void gelu_bf16_kernel(const uint4* __restrict__ x, const uint4* __restrict__ dy, uint4* __restrict__ out, long n8) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n8) return;
    float v[8], d[8], o[8];
    chunk_to_f32<bf16_t>(x[i], v);
    if (dy) {
        chunk_to_f32<bf16_t>(dy[i], d);
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = d[e] * gelu_grad(v[e]);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) o[e] = gelu_f(v[e]);
    }
    out[i] = make_uint4(pack_bf2(o[0], o[1]), pack_bf2(o[2], o[3]), pack_bf2(o[4], o[5]), pack_bf2(o[6], o[7]));
}

// the C <= 128 form of the kernel below: a row per half-wave
__global__ __launch_bounds__(256) void layernorm_fwd_mixed_narrow_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                                         const float* __restrict__ beta, void* __restrict__ y, int rows, int C,
                                                                         float eps, int xb, int yb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 31;
    const int row = blockIdx.x * 8 + wave * 2 + (lane >> 5);
    const bool ok = g < (C >> 2) && row < rows;
    const float4 xv = ok ? ld4(x, xb, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    const float invC = 1.f / (float)C;
    const float mean = half_wave_sum(xv.x + xv.y + xv.z + xv.w) * invC;
    const float a = xv.x - mean, b = xv.y - mean, c = xv.z - mean, d = xv.w - mean;
    const float rstd = 1.f / sqrtf(half_wave_sum(ok ? a * a + b * b + c * c + d * d : 0.f) * invC + eps);
    if (!ok) return;
    const float4 ga = gamma ? *(const float4*)(gamma + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);
    const float4 be = beta ? *(const float4*)(beta + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    st4(y, yb, (size_t)row * C + 4 * g, make_float4(a * rstd * ga.x + be.x, b * rstd * ga.y + be.y, c * rstd * ga.z + be.z, d * rstd * ga.w + be.w));
}

// y = LN(x) gamma + beta with x fp32 | bf16 and y fp32 | bf16 (one wave per row, the statistics in fp32 from the row in registers):
// the training forward whose consumer is a bf16 projection writes the projection's operand directly - torch's autocast runs
// layer_norm in fp32 and casts its result for the linear that follows (train_camera.py:157-160), the same rounding in one pass
__global__ __launch_bounds__(256) void layernorm_fwd_mixed_kernel(const void* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, void* __restrict__ y, int rows, int C, float eps,
                                                                  int xb, int yb) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int row = blockIdx.x * 4 + wave;
    if (row >= rows) return;
    const int groups = C >> 2;
    float4 xv[kLnMaxPerLane];
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int g = lane + 64 * i;
        xv[i] = g < groups ? ld4(x, xb, (size_t)row * C + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        s += xv[i].x + xv[i].y + xv[i].z + xv[i].w;
    }
    const float invC = 1.f / (float)C;
    const float mean = wave_sum(s) * invC;
    float v = 0.f;
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        if (lane + 64 * i < groups) {
            const float a = xv[i].x - mean, b = xv[i].y - mean, c = xv[i].z - mean, d = xv[i].w - mean;
            v += a * a + b * b + c * c + d * d;
        }
    }
    const float rstd = 1.f / sqrtf(wave_sum(v) * invC + eps);
#pragma unroll
    for (int i = 0; i < kLnMaxPerLane; ++i) {
        const int g = lane + 64 * i;
        if (g < groups) {
            const float4 ga = gamma ? *(const float4*)(gamma + 4 * g) : make_float4(1.f, 1.f, 1.f, 1.f);
            const float4 be = beta ? *(const float4*)(beta + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
            float4 o;
            o.x = (xv[i].x - mean) * rstd * ga.x + be.x; o.y = (xv[i].y - mean) * rstd * ga.y + be.y;
            o.z = (xv[i].z - mean) * rstd * ga.z + be.z; o.w = (xv[i].w - mean) * rstd * ga.w + be.w;
            st4(y, yb, (size_t)row * C + 4 * g, o);
        }
    }
}

// Weight gradient of a k x k convolution (fp32, channels-last): dW[o][c][a][b] = sum over output pixels m of
// dY[m][o] * X[pixel of m shifted by tap (a, b)][c]  - a GEMM whose contraction runs over the PIXELS.  One workgroup = (32-cout
// tile, 32-cin tile, tap, share of the output rows); v_mfma_f32_32x32x2_f32 consumes two pixels per instruction and both operands
// are one coalesced 128-byte row piece per half-wave straight from global memory (A[cout][pixel] = dY, B[pixel][cin] = X), so there
// is no LDS staging; the four waves split the rows, meet in LDS and add the tile into dW with fp32 atomics (dW zero-initialised).
// The reference gets this from cuDNN under autograd (every nn.Conv2d of resnet_ms.py / fax_modules.py / naive_decoder.py).
// T = storage type of X and dY (bf16 under autocast: half the operand bytes); the products and dW are fp32 either way (a bf16
// matrix instruction would want the contraction - the PIXELS - contiguous per lane, i.e. both operands transposed).
struct WgradParams {
    const void* x;       // (N, H, W, Cin)
    const void* dy;      // (N, Ho, Wo, Cout)
    float* dw;           // (Cout, Cin, k, k) fp32
    int N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, nchunks;
};

// (Tried: one workgroup per tap ROW / per all k x k taps with one accumulator per tap, so that dY is loaded once for several
// matrix instructions: 82 -> 96 / 111 ms per CorpBEVT step (5 agents, fp32).  The kernel lives on having eight waves per SIMD hide its scalar
// loads; more accumulators per wave means fewer waves.  The structural fix is a bf16 matrix path over pixel-contiguous operands.)
template <typename T>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradParams p) {
    constexpr int PU = 4;              // pixel pairs per batch of loads (8: 82 -> 89 ms per step, a wave less per SIMD and idle lanes on narrow maps)
    __shared__ float red[3 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int co0 = blockIdx.x * 32;
    const int ntap = p.k * p.k;
    const int ci0 = (blockIdx.y / ntap) * 32, tap = blockIdx.y % ntap;
    const int ta = tap / p.k, tb = tap - ta * p.k;
    const int rows = p.N * p.Ho;                                   // output rows (image, oy)
    const int per = (rows + p.nchunks - 1) / p.nchunks;
    const int r0 = blockIdx.z * per, r1 = min(rows, r0 + per);
    const bool co_ok = co0 + ql < p.Cout, ci_ok = ci0 + ql < p.Cin;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    for (int row = r0 + wave; row < r1; row += 4) {
        const int n = row / p.Ho, oy = row - n * p.Ho;
        const int iy = oy * p.stride - p.pad + ta;
        const bool y_ok = iy >= 0 && iy < p.H;
        // unconditional loads from clamped addresses + select: a load under a branch makes the compiler drain vmcnt right after it
        const T* dyr = (const T*)p.dy + (size_t)row * p.Wo * p.Cout + min(co0 + ql, p.Cout - 1);
        const T* xr = (const T*)p.x + ((size_t)n * p.H + (y_ok ? iy : 0)) * p.W * p.Cin + min(ci0 + ql, p.Cin - 1);
        for (int ox0 = 0; ox0 < p.Wo; ox0 += 2 * PU) {             // PU MFMAs (2 PU pixels) with their loads in flight together
            float a[PU], bv[PU];
#pragma unroll
            for (int u = 0; u < PU; ++u) {
                const int ox = ox0 + 2 * u + h;
                const int ix = ox * p.stride - p.pad + tb;
                const bool o_ok = ox < p.Wo;
                const bool i_ok = o_ok && y_ok && ix >= 0 && ix < p.W;
                const float av = load_elem<T>(dyr, (size_t)min(ox, p.Wo - 1) * p.Cout);
                const float xv = load_elem<T>(xr, (size_t)min(max(ix, 0), p.W - 1) * p.Cin);
                a[u] = (o_ok && co_ok) ? av : 0.f;
                bv[u] = (i_ok && ci_ok) ? xv : 0.f;
            }
#pragma unroll
            for (int u = 0; u < PU; ++u) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[u], bv[u], acc, 0, 0, 0);
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r)
        if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const float v = acc[r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
            const int co = co0 + acc_row(r, lane), ci = ci0 + ql;   // D[cout rows][cin cols]
            if (co < p.Cout && ci < p.Cin && v != 0.f) atomicAdd(p.dw + ((size_t)co * p.Cin + ci) * ntap + tap, v);
        }
    }
}

// Weight gradient on the bf16 matrix path (stride 1, k = 1 or 3): v_mfma_f32_32x32x16_bf16 contracts 16 PIXELS per instruction
// and wants each lane's 8 pixels of its channel in one register quad, while consecutive lanes (channels) should read consecutive
// addresses.  Both hold for a BLOCKED copy of the operands, made by the caller with one permute:
//     Xb[n][iy + pad][blk][c][8]   X zero-padded by `pad` on every side (and on the right up to the block count), 8 pixels per block
//     Db[n][oy][blk][o][8]         dY, the row zero-padded to a multiple of 16 pixels
// A lane's operand is one 16-byte load and a half-wave reads 512 contiguous bytes; no bounds tests in the loop (the zero padding
// does it).  The k taps of a tap ROW differ only by a shift of 0 .. k-1 pixels inside the same two blocks: they share their loads
// and are separated by a funnel shift in registers (one accumulator per tap); tap rows, channel tiles and shares of the output
// rows are separate workgroups as in conv_wgrad_kernel.
struct Wgrad16Params {
    const uint4* xb;     // [N][Hp][XB][Cin] blocks of 8 bf16
    const uint4* db;     // [N][Ho][DB][Cout] blocks of 8 bf16
    float* dw;           // (Cout, Cin, k, k) fp32, zero-initialised
    int N, Hp, XB, Cin, Ho, DB, Cout, k, nchunks;
    int sy;              // padded X row of output row oy, tap row ta: oy * sy + ta (the convolution's stride)
    int P;               // planes per block of X: [N][Hp][XB][P][Cin] (1 for stride 1)
};

// S2 (stride 2; the first convolution and the 1 x 1 shortcut of a down-sampling BasicBlock, resnet_ms.py:67-74): the caller de-interleaves
// the columns - plane q of block b holds the (zero-padded) input pixels 2 (8 b + j) + q, j = 0..7 (cobevt_wgrad_block_operand with
// sx = 2) - so tap 0 / 1 / 2 of output pixel ox are pixel ox of plane 0, pixel ox of plane 1 and pixel ox + 1 of plane 0: again
// 16-byte loads and one funnel shift, no bounds tests.  k = 1: one plane (the even pixels).
template <int KW, bool S2>
__global__ __launch_bounds__(256) void conv_wgrad16_kernel(Wgrad16Params p) {
    __shared__ float red[3 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int co0 = blockIdx.x * 32;
    const int ci0 = (blockIdx.y / p.k) * 32, ta = blockIdx.y % p.k;          // one tap row per workgroup
    const int rows = p.N * p.Ho;
    const int per = (rows + p.nchunks - 1) / p.nchunks;
    const int r0 = blockIdx.z * per, r1 = min(rows, r0 + per);
    const bool co_ok = co0 + ql < p.Cout, ci_ok = ci0 + ql < p.Cin;
    const int co = min(co0 + ql, p.Cout - 1), ci = min(ci0 + ql, p.Cin - 1);
    f32x16 acc[KW];
#pragma unroll
    for (int t = 0; t < KW; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    // this wave's work = (row, 16-pixel step) pairs, rows r0 + wave, + 4, ...; U of them per batch so that 3 U loads are in flight
    // before the first matrix instruction (one step per batch left the wave waiting out a global round trip per 3 MFMAs)
    constexpr int U = 4;
    const int nsteps = p.DB >> 1;
    const int nrows_w = r1 > r0 + wave ? (r1 - r0 - wave + 3) >> 2 : 0;
    const int total = nrows_w * nsteps;
    for (int base = 0; base < total; base += U) {
        uint4 a[U], w0[U], w1[U];
        uint32_t w4[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = min(base + u, total - 1);              // the tail repeats the last step; its products are dropped below
            const int ri = idx / nsteps, b0 = (idx - ri * nsteps) * 2;
            const int row = r0 + wave + 4 * ri;
            const int n = row / p.Ho, oy = row - n * p.Ho;
            const uint4* dr = p.db + ((size_t)row * p.DB + b0 + h) * p.Cout + co;                          // block b0 + h of dY's row
            const uint4* xr = p.xb + ((((size_t)n * p.Hp + oy * p.sy + ta) * p.XB + b0 + h) * p.P) * p.Cin + ci;   // padded row oy * sy + ta
            a[u] = dr[0];
            w0[u] = xr[0];
            w4[u] = 0;
            if constexpr (KW > 1) w4[u] = ((const uint32_t*)(xr + p.P * p.Cin))[0];                        // pixels 8, 9 of the window
            if constexpr (KW > 1 && S2) w1[u] = xr[p.Cin];                                                  // plane 1: the odd columns
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = base + u < total;
            uint4 av = (co_ok && live) ? a[u] : zero;
            uint4 wv = ci_ok ? w0[u] : zero;
            const uint32_t w4v = ci_ok ? w4[u] : 0u;
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, wv), acc[0], 0, 0, 0);
            if constexpr (KW > 1) {
                // tap 1: pixels 1..8, tap 2: pixels 2..9 of the ten loaded
                const uint4 s1 = make_uint4(__builtin_amdgcn_alignbit(wv.y, wv.x, 16), __builtin_amdgcn_alignbit(wv.z, wv.y, 16),
                                            __builtin_amdgcn_alignbit(wv.w, wv.z, 16), __builtin_amdgcn_alignbit(w4v, wv.w, 16));
                const uint4 s2 = make_uint4(wv.y, wv.z, wv.w, w4v);
                if constexpr (S2) {                   // tap 1: plane 1 as loaded; tap 2: pixels 1..8 of plane 0
                    const uint4 odd = ci_ok ? w1[u] : zero;
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, odd), acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, s1), acc[2], 0, 0, 0);
                } else {
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, s1), acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, s2), acc[2], 0, 0, 0);
                }
            }
        }
    }
    const int ntap = p.k * p.k;
#pragma unroll
    for (int t = 0; t < KW; ++t) {
        if (t) __syncthreads();                                    // red is reused per tap
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        if (wave == 0) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[t][r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
                const int o = co0 + acc_row(r, lane), c = ci0 + ql;     // D[cout rows][cin cols]
                if (o < p.Cout && c < p.Cin && v != 0.f) atomicAdd(p.dw + ((size_t)o * p.Cin + c) * ntap + ta * p.k + t, v);
            }
        }
    }
}


// "Columns as channels" (the 7 x 7 / stride 2 stem on 3 input channels, resnet_ms.py:67-69: a 32-channel tile per tap would be 90 % empty):
// the caller lays the k taps of a tap row out as k planes per block - plane b of block q holds input pixels s (8 q + j) + b - pad -
// so that (tap column b, channel c) = pseudo-channel b Cin + c <= 32 is ONE lane of the B operand and a tap ROW is one matrix
// instruction per 16 pixels; a wave carries the KH tap rows as KH accumulators over one load of dY.  Workgroup = (32-cout tile, share of
// the output rows).
template <int KH>
__global__ __launch_bounds__(256) void conv_wgrad16_cols_kernel(Wgrad16Params p) {
    __shared__ float red[3 * 16 * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;
    const int co0 = blockIdx.x * 32;
    const int CP = p.P * p.Cin;                                     // pseudo-channels (<= 32)
    const int rows = p.N * p.Ho;
    const int per = (rows + p.nchunks - 1) / p.nchunks;
    const int r0 = blockIdx.z * per, r1 = min(rows, r0 + per);
    const bool co_ok = co0 + ql < p.Cout, cp_ok = ql < CP;
    const int co = min(co0 + ql, p.Cout - 1), cp = min(ql, CP - 1);
    f32x16 acc[KH];
#pragma unroll
    for (int t = 0; t < KH; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    const uint4 zero = make_uint4(0, 0, 0, 0);
    constexpr int U = 2;
    const int nsteps = p.DB >> 1;
    const int nrows_w = r1 > r0 + wave ? (r1 - r0 - wave + 3) >> 2 : 0;
    const int total = nrows_w * nsteps;
    for (int base = 0; base < total; base += U) {
        uint4 a[U], w[U][KH];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int idx = min(base + u, total - 1);
            const int ri = idx / nsteps, b0 = (idx - ri * nsteps) * 2;
            const int row = r0 + wave + 4 * ri;
            const int n = row / p.Ho, oy = row - n * p.Ho;
            a[u] = p.db[((size_t)row * p.DB + b0 + h) * p.Cout + co];
            const uint4* xr = p.xb + (((size_t)n * p.Hp + oy * p.sy) * p.XB + b0 + h) * CP + cp;
#pragma unroll
            for (int t = 0; t < KH; ++t) w[u][t] = xr[(size_t)t * p.XB * CP];            // padded row oy * sy + t
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const bool live = base + u < total;
            const uint4 av = (co_ok && live) ? a[u] : zero;
#pragma unroll
            for (int t = 0; t < KH; ++t) {
                const uint4 wv = cp_ok ? w[u][t] : zero;
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, wv), acc[t], 0, 0, 0);
            }
        }
    }
    const int ntap = p.k * p.k;
#pragma unroll
    for (int t = 0; t < KH; ++t) {
        if (t) __syncthreads();
#pragma unroll
        for (int r = 0; r < 16; ++r)
            if (wave > 0) red[((wave - 1) * 16 + r) * 64 + lane] = acc[t][r];
        __syncthreads();
        if (wave == 0) {
            const int tb = ql / p.Cin, c = ql - tb * p.Cin;             // pseudo-channel -> (tap column, channel)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const float v = acc[t][r] + red[r * 64 + lane] + red[(16 + r) * 64 + lane] + red[(32 + r) * 64 + lane];
                const int o = co0 + acc_row(r, lane);
                if (o < p.Cout && ql < CP && v != 0.f) atomicAdd(p.dw + ((size_t)o * p.Cin + c) * ntap + t * p.k + tb, v);
            }
        }
    }
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta,
                                    int rows, int C, float eps, hipStream_t stream) {
    if (!x || !dy || !dx || ((dgamma == nullptr) != (dbeta == nullptr))) return COBEVT_ERR_ARG;
    if (rows < 1 || C < 4 || C % 4 || C > 1024) return COBEVT_ERR_SHAPE;
    // enough workgroups to fill the chip, few enough that the per-channel atomics stay cheap
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const int blocks = (rows + rpb - 1) / rpb;
    if (C <= 128) {
        if (rows >= 16384) {
            rpb = ((rows + 255) / 256 + 63) / 64 * 64;
            hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<16>, dim3((rows + rpb - 1) / rpb), dim3(1024), 0, stream, (const void*)x, (const void*)dy,
                               gamma, (void*)dx, dgamma, dbeta, rows, C, eps, rpb, 0, 0, 0);
        } else {
            rpb = ((rpb + 15) / 16) * 16;
            hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<4>, dim3((rows + rpb - 1) / rpb), dim3(256), 0, stream, (const void*)x, (const void*)dy,
                               gamma, (void*)dx, dgamma, dbeta, rows, C, eps, rpb, 0, 0, 0);
        }
        return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, (const void*)x, (const void*)dy, gamma, (void*)dx, dgamma, dbeta, rows, C,
                       eps, rpb, 0, 0, 0);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dtypes: [x, dy, dx], 0 = bf16, 1 = fp32 each
extern "C" int cobevt_layernorm_bwd_t(const void* x, const void* dy, const float* gamma, void* dx, float* dgamma, float* dbeta,
                                      int rows, int C, float eps, const int* dtypes, hipStream_t stream) {
    if (!x || !dy || !dx || !dtypes || ((dgamma == nullptr) != (dbeta == nullptr))) return COBEVT_ERR_ARG;
    for (int i = 0; i < 3; ++i) if (dtypes[i] != 0 && dtypes[i] != 1) return COBEVT_ERR_ARG;
    if (rows < 1 || C < 4 || C % 4 || C > 1024) return COBEVT_ERR_SHAPE;
    int rpb = (rows + 1023) / 1024;
    rpb = ((rpb + 3) / 4) * 4;
    const int blocks = (rows + rpb - 1) / rpb;
    if (C <= 128) {
        if (rows >= 16384) {
            rpb = ((rows + 255) / 256 + 63) / 64 * 64;
            hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<16>, dim3((rows + rpb - 1) / rpb), dim3(1024), 0, stream, x, dy, gamma, dx, dgamma, dbeta, rows,
                               C, eps, rpb, dtypes[0] == 0, dtypes[1] == 0, dtypes[2] == 0);
        } else {
            rpb = ((rpb + 15) / 16) * 16;
            hipLaunchKernelGGL(layernorm_bwd_narrow_kernel<4>, dim3((rows + rpb - 1) / rpb), dim3(256), 0, stream, x, dy, gamma, dx, dgamma, dbeta, rows,
                               C, eps, rpb, dtypes[0] == 0, dtypes[1] == 0, dtypes[2] == 0);
        }
        return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
    }
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(blocks), dim3(256), 0, stream, x, dy, gamma, dx, dgamma, dbeta, rows, C, eps, rpb,
                       dtypes[0] == 0, dtypes[1] == 0, dtypes[2] == 0);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dtypes: [x, y], 0 = bf16, 1 = fp32 each
extern "C" int cobevt_layernorm_fwd_t(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps,
                                      const int* dtypes, hipStream_t stream) {
    if (!x || !y || !dtypes || ((gamma == nullptr) != (beta == nullptr))) return COBEVT_ERR_ARG;
    for (int i = 0; i < 2; ++i) if (dtypes[i] != 0 && dtypes[i] != 1) return COBEVT_ERR_ARG;
    if (rows < 1 || C < 4 || C % 4 || C > 1024) return COBEVT_ERR_SHAPE;
    if (C <= 128) hipLaunchKernelGGL(layernorm_fwd_mixed_narrow_kernel, dim3((rows + 7) / 8), dim3(256), 0, stream, x, gamma, beta, y, rows, C, eps,
                                     dtypes[0] == 0, dtypes[1] == 0);
    else hipLaunchKernelGGL(layernorm_fwd_mixed_kernel, dim3((rows + 3) / 4), dim3(256), 0, stream, x, gamma, beta, y, rows, C, eps, dtypes[0] == 0,
                            dtypes[1] == 0);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// bf16 x / dy / out: out = GELU(x) (dy null) or dy * GELU'(x); n % 8 == 0
extern "C" int cobevt_gelu_bf16(const void* x, const void* dy, void* out, long n, hipStream_t stream) {
    if (!x || !out) return COBEVT_ERR_ARG;
    if (n < 8 || n % 8) return COBEVT_ERR_SHAPE;
    const long n8 = n / 8;
    hipLaunchKernelGGL(gelu_bf16_kernel, dim3((unsigned)((n8 + 255) / 256)), dim3(256), 0, stream, (const uint4*)x, (const uint4*)dy, (uint4*)out, n8);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_gelu(const float* x, const float* dy, float* out, long n, hipStream_t stream) {
    if (!x || !out) return COBEVT_ERR_ARG;
    if (n < 4 || n % 4) return COBEVT_ERR_SHAPE;
    const long n4 = n / 4;
    hipLaunchKernelGGL(gelu_kernel, dim3((unsigned)((n4 + 255) / 256)), dim3(256), 0, stream, x, dy, out, n4);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_conv_wgrad(const void* x, const void* dy, float* dw, const int* dims, hipStream_t stream) {
    // dims: [N, H, W, Cin, Ho, Wo, Cout, k, stride, pad, dtype of x / dy (0 bf16, 1 fp32)]
    if (!x || !dy || !dw || !dims) return COBEVT_ERR_ARG;
    if (dims[10] != 0 && dims[10] != 1) return COBEVT_ERR_ARG;
    WgradParams p;
    p.x = x; p.dy = dy; p.dw = dw;
    p.N = dims[0]; p.H = dims[1]; p.W = dims[2]; p.Cin = dims[3]; p.Ho = dims[4]; p.Wo = dims[5]; p.Cout = dims[6];
    p.k = dims[7]; p.stride = dims[8]; p.pad = dims[9];
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.Cin < 1 || p.Ho < 1 || p.Wo < 1 || p.Cout < 1 || p.k < 1 || p.k > 7 || p.stride < 1 || p.pad < 0)
        return COBEVT_ERR_SHAPE;
    const int tiles_o = (p.Cout + 31) / 32, tiles_i = (p.Cin + 31) / 32, ntap = p.k * p.k;
    const long per = (long)tiles_o * tiles_i * ntap;
    const int rows = p.N * p.Ho;
    long chunks = (2048 + per - 1) / per;                          // about 2048 workgroups, at least four rows each
    if (chunks > rows / 4) chunks = rows / 4;
    if (chunks < 1) chunks = 1;
    p.nchunks = (int)chunks;
    if (tiles_i * ntap > 65535 || chunks > 65535) return COBEVT_ERR_SHAPE;
    const dim3 grid(tiles_o, tiles_i * ntap, (unsigned)chunks);
    if (dims[10] == 0) hipLaunchKernelGGL(conv_wgrad_kernel<bf16_t>, grid, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL(conv_wgrad_kernel<float>, grid, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_conv_wgrad_blocked(const void* xb, const void* db, float* dw, const int* dims, hipStream_t stream) {
    // dims: [N, Hp, XB, Cin, Ho, DB, Cout, k, sy, mode]; mode 0: stride 1 (k = 1 / 3), 1: stride 2 column planes (k = 1: one plane, k = 3: two),
    // 2: the k tap columns as planes = pseudo-channels (k * Cin <= 32, k <= 7)
    if (!xb || !db || !dw || !dims) return COBEVT_ERR_ARG;
    Wgrad16Params p;
    p.xb = (const uint4*)xb; p.db = (const uint4*)db; p.dw = dw;
    p.N = dims[0]; p.Hp = dims[1]; p.XB = dims[2]; p.Cin = dims[3]; p.Ho = dims[4]; p.DB = dims[5]; p.Cout = dims[6]; p.k = dims[7];
    p.sy = dims[8];
    const int mode = dims[9];
    if (mode < 0 || mode > 2) return COBEVT_ERR_ARG;
    if (p.N < 1 || p.Cin < 1 || p.Ho < 1 || p.Cout < 1 || p.sy < 1 || p.k < 1) return COBEVT_ERR_SHAPE;
    if (mode < 2 && p.k != 1 && p.k != 3) return COBEVT_ERR_SHAPE;
    if (mode == 0 && p.sy != 1) return COBEVT_ERR_SHAPE;
    if (mode == 2 && (p.k > 7 || p.k * p.Cin > 32)) return COBEVT_ERR_SHAPE;
    p.P = mode == 0 ? 1 : (mode == 1 ? (p.k == 3 ? 2 : 1) : p.k);
    // the blocked rows must hold what the loop reads: rows (Ho - 1) sy + k - 1, blocks b0 + h (+ 1 where a tap is a pixel shift) with b0 + 1 < DB + 1
    const int shift_blk = (mode < 2 && p.k > 1) ? 1 : 0;
    if (p.DB < 2 || (p.DB & 1) || p.Hp < (p.Ho - 1) * p.sy + p.k || p.XB < p.DB + shift_blk) return COBEVT_ERR_SHAPE;
    const int tiles_o = (p.Cout + 31) / 32, tiles_i = mode == 2 ? 1 : (p.Cin + 31) / 32;
    const int ky = mode == 2 ? 1 : p.k;                             // tap rows as separate workgroups (mode 2: as accumulators)
    const long per = (long)tiles_o * tiles_i * ky;
    const int rows = p.N * p.Ho;
    // about COBEVT_WGRAD16_WGS workgroups, at least four rows each: every workgroup ends in 1024 x k fp32 atomics
#ifndef COBEVT_WGRAD16_WGS
#define COBEVT_WGRAD16_WGS 512
#endif
    long chunks = (COBEVT_WGRAD16_WGS + per - 1) / per;
    if (chunks > rows / 4) chunks = rows / 4;
    if (chunks < 1) chunks = 1;
    p.nchunks = (int)chunks;
    if (tiles_i * ky > 65535 || chunks > 65535) return COBEVT_ERR_SHAPE;
    const dim3 grid(tiles_o, tiles_i * ky, (unsigned)chunks);
    if (mode == 2) {
        switch (p.k) {
            case 1: hipLaunchKernelGGL(conv_wgrad16_cols_kernel<1>, grid, dim3(256), 0, stream, p); break;
            case 3: hipLaunchKernelGGL(conv_wgrad16_cols_kernel<3>, grid, dim3(256), 0, stream, p); break;
            case 5: hipLaunchKernelGGL(conv_wgrad16_cols_kernel<5>, grid, dim3(256), 0, stream, p); break;
            case 7: hipLaunchKernelGGL(conv_wgrad16_cols_kernel<7>, grid, dim3(256), 0, stream, p); break;
            default: return COBEVT_ERR_UNSUPPORTED;
        }
    } else if (mode == 1) {
        if (p.k == 3) hipLaunchKernelGGL((conv_wgrad16_kernel<3, true>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad16_kernel<1, true>), grid, dim3(256), 0, stream, p);
    } else {
        if (p.k == 3) hipLaunchKernelGGL((conv_wgrad16_kernel<3, false>), grid, dim3(256), 0, stream, p);
        else hipLaunchKernelGGL((conv_wgrad16_kernel<1, false>), grid, dim3(256), 0, stream, p);
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
