// Training forms of the FAX query embedding (gfx950): CrossViewSwapAttention's BEV query
//     v = bev_embed(grid) - cam_embed(camera centre),   query = v / (||v|| + 1e-7) + x        (fax_modules.py:344-372)
// for every (batch, camera, BEV pixel), channels-last, forward and backward - what torch autograd runs as a K = 2 convolution, a broadcast
// subtraction, a norm, a division, an addition and a permute + copy over a (B, n, d, H, W) tensor (42 M elements at level 0 of the 5-agent
// frame: eight 76-us elementwise launches, two 100-us reductions and their copies per step, profiles/r04_train_amp_kernel_trace.txt) under
// train_camera.py:143-179.  The inference path has this fused since round 1 (cobevt_fax_bev_embed); here the parameters need gradients:
//     dv = dq / s - v (v . dq) / (r s^2),  r = ||v||, s = r + 1e-7;   dx = sum over cameras of dq;
//     dW[:, 0] = sum dv gx, dW[:, 1] = sum dv gy, dbias = sum dv, dc[b, cam] = - sum over pixels of dv.
// d = 128: a row is 32 lanes x 4 channels, one row per half-wave; a workgroup walks the pixels of one batch element, every camera of a pixel
// in turn (dx is written once, no atomics), keeps the parameter-gradient partial sums in registers and adds them to the global fp32
// accumulators once (LDS reduction over its 8 half-waves, then one atomic per word).
// round_bf16 = 1 (inside a bf16 autocast region): the 1x1 convolution's operands and result and the difference are rounded to bf16 as
// torch's autocast does (conv in bf16, the subtraction of two bf16 tensors in bf16; norm / division / addition in fp32).
#include "common.hpp"

namespace cobevt {
namespace {

constexpr int kD = 128;
constexpr int kMaxCam = 8;

struct BevQueryTrainParams {
    const float* grid;   // (KD, H, W), or (B, KD, H, W) with grid_stride = KD * H * W (the per-camera ray directions of the image embedding)
    const float* w;      // (d, KD)
    const float* bias;   // (d) | null
    const float* c;      // (B * n, d)
    const float* x;      // (B, H, W, d) | null     forward
    float* out;          // (B, n, H, W, d)         forward
    const float* dq;     // (B, n, H, W, d)         backward
    float* dx;           // (B, H, W, d) | null     backward
    float* dw;           // (d, KD) accumulated
    float* dbias;        // (d)     accumulated | null
    float* dc;           // (B * n, d) accumulated
    long grid_stride;    // floats between the batch elements' grids (0: one grid for all)
    int B, n, HW, W, round_bf16;
};

__device__ __forceinline__ float rbf(float v) { return bf2f(f2bf(v)); }
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
    for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// KD = input channels of the 1x1 embedding convolution: 2 (the BEV grid) or 4 (homogeneous ray directions, fax_modules.py:330-343)
template <bool BWD, int KD>
__global__ __launch_bounds__(256) void bev_query_train_kernel(BevQueryTrainParams p) {
    __shared__ float red[8][kD];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int g = lane & 31, hw8 = wave * 2 + (lane >> 5);        // channel group, half-wave of the workgroup
    const int b = blockIdx.y;
    // this lane's weight columns: w is (d, KD), channels 4 g .. 4 g + 3 are 4 KD consecutive floats
    float4 wk[KD], awk[KD];
    {
        float wr[4 * KD];
#pragma unroll
        for (int i = 0; i < KD; ++i) *(float4*)&wr[4 * i] = *(const float4*)(p.w + 4 * KD * g + 4 * i);
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            wk[k] = make_float4(wr[k], wr[KD + k], wr[2 * KD + k], wr[3 * KD + k]);
            if (p.round_bf16) wk[k] = make_float4(rbf(wk[k].x), rbf(wk[k].y), rbf(wk[k].z), rbf(wk[k].w));
            awk[k] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
    }
    const float4 bi = p.bias ? *(const float4*)(p.bias + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
    float4 ab = make_float4(0.f, 0.f, 0.f, 0.f);                   // backward: bias-gradient partial sum
    float4 cc[kMaxCam], ac[kMaxCam];
#pragma unroll
    for (int cam = 0; cam < kMaxCam; ++cam) {
        cc[cam] = cam < p.n ? *(const float4*)(p.c + ((size_t)b * p.n + cam) * kD + 4 * g) : make_float4(0.f, 0.f, 0.f, 0.f);
        ac[cam] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
    const float* grid = p.grid + (size_t)b * p.grid_stride;
    for (int pix = blockIdx.x * 8 + hw8; pix < p.HW; pix += gridDim.x * 8) {
        float gk[KD];
#pragma unroll
        for (int k = 0; k < KD; ++k) {
            gk[k] = grid[(size_t)k * p.HW + pix];
            if (p.round_bf16) gk[k] = rbf(gk[k]);
        }
        // the convolution's result for this pixel (shared by the cameras)
        float4 e = bi;
#pragma unroll
        for (int k = 0; k < KD; ++k) { e.x += wk[k].x * gk[k]; e.y += wk[k].y * gk[k]; e.z += wk[k].z * gk[k]; e.w += wk[k].w * gk[k]; }
        if (p.round_bf16) e = make_float4(rbf(e.x), rbf(e.y), rbf(e.z), rbf(e.w));
        float4 xv = make_float4(0.f, 0.f, 0.f, 0.f);               // forward: x of the pixel; backward: dx = sum over the cameras of dq
        if constexpr (!BWD) { if (p.x) xv = *(const float4*)(p.x + ((size_t)b * p.HW + pix) * kD + 4 * g); }
#pragma unroll
        for (int cam = 0; cam < kMaxCam; ++cam) {
            if (cam >= p.n) break;
            float4 v = make_float4(e.x - cc[cam].x, e.y - cc[cam].y, e.z - cc[cam].z, e.w - cc[cam].w);
            if (p.round_bf16) v = make_float4(rbf(v.x), rbf(v.y), rbf(v.z), rbf(v.w));
            const float r = sqrtf(half_sum(v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w));
            const float s = r + 1e-7f;
            const size_t row = (((size_t)b * p.n + cam) * p.HW + pix) * kD + 4 * g;
            if constexpr (!BWD) {
                const float inv = 1.f / s;
                *(float4*)(p.out + row) = make_float4(v.x * inv + xv.x, v.y * inv + xv.y, v.z * inv + xv.z, v.w * inv + xv.w);
            } else {
                const float4 dq = *(const float4*)(p.dq + row);
                const float vd = half_sum(v.x * dq.x + v.y * dq.y + v.z * dq.z + v.w * dq.w);
                const float a = 1.f / s, kk = r > 0.f ? vd / (r * s * s) : 0.f;
                const float4 dv = make_float4(dq.x * a - v.x * kk, dq.y * a - v.y * kk, dq.z * a - v.z * kk, dq.w * a - v.w * kk);
#pragma unroll
                for (int k = 0; k < KD; ++k) {
                    awk[k].x += dv.x * gk[k]; awk[k].y += dv.y * gk[k]; awk[k].z += dv.z * gk[k]; awk[k].w += dv.w * gk[k];
                }
                ab.x += dv.x; ab.y += dv.y; ab.z += dv.z; ab.w += dv.w;
                ac[cam].x -= dv.x; ac[cam].y -= dv.y; ac[cam].z -= dv.z; ac[cam].w -= dv.w;
                xv.x += dq.x; xv.y += dq.y; xv.z += dq.z; xv.w += dq.w;
            }
        }
        if constexpr (BWD) { if (p.dx) *(float4*)(p.dx + ((size_t)b * p.HW + pix) * kD + 4 * g) = xv; }
    }
    if constexpr (BWD) {
        // the workgroup's partial sums: LDS reduction over its 8 half-waves, one atomic per word
        auto flush = [&](const float4& v, float* dst, int stride) __attribute__((always_inline)) {
            __syncthreads();
            *(float4*)&red[hw8][4 * g] = v;
            __syncthreads();
            if (threadIdx.x < kD) {
                float t = 0.f;
#pragma unroll
                for (int k = 0; k < 8; ++k) t += red[k][threadIdx.x];
                if (dst && t != 0.f) atomicAdd(dst + threadIdx.x * stride, t);
            }
        };
#pragma unroll
        for (int cam = 0; cam < kMaxCam; ++cam)
            if (cam < p.n) flush(ac[cam], p.dc + ((size_t)b * p.n + cam) * kD, 1);
#pragma unroll
        for (int k = 0; k < KD; ++k) flush(awk[k], p.dw + k, KD);
        flush(ab, p.dbias, 1);
    }
}

inline int pixel_blocks(int HW, int B, int total = 2048) {
    int blocks = (HW + 7) / 8;                       // one row per half-wave and round
    const int cap = (total + B - 1) / B;             // about `total` workgroups in all (2048 = eight per CU)
    return blocks > cap ? cap : (blocks < 1 ? 1 : blocks);
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_fax_bev_query_train(const float* grid, const float* w, const float* bias, const float* c, const float* x, float* out,
                                          const int* dims, hipStream_t stream) {
    // dims: [B, n, H, W, d, round_bf16, KD (2 | 4), per_batch_grid (0 | 1)]; x nullable
    if (!grid || !w || !c || !out || !dims) return COBEVT_ERR_ARG;
    if (dims[4] != kD || (dims[6] != 2 && dims[6] != 4)) return COBEVT_ERR_UNSUPPORTED;
    BevQueryTrainParams p = {};
    p.grid = grid; p.w = w; p.bias = bias; p.c = c; p.x = x; p.out = out;
    p.B = dims[0]; p.n = dims[1]; p.HW = dims[2] * dims[3]; p.W = dims[3]; p.round_bf16 = dims[5] != 0;
    p.grid_stride = dims[7] ? (long)dims[6] * p.HW : 0;
    if (p.B < 1 || p.n < 1 || p.n > kMaxCam || dims[2] < 1 || dims[3] < 1 || p.B > 65535) return COBEVT_ERR_SHAPE;
    const dim3 grid_dim(pixel_blocks(p.HW, p.B), p.B);
    if (dims[6] == 2) hipLaunchKernelGGL((bev_query_train_kernel<false, 2>), grid_dim, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bev_query_train_kernel<false, 4>), grid_dim, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_fax_bev_query_train_bwd(const float* grid, const float* w, const float* bias, const float* c, const float* dq, float* dx,
                                              float* dw, float* dbias, float* dc, const int* dims, hipStream_t stream) {
    // dims as above; dw (d, KD), dbias (d) | null, dc (B * n, d): zero-initialised by the caller, accumulated; dx (nullable) written
    if (!grid || !w || !c || !dq || !dw || !dc || !dims) return COBEVT_ERR_ARG;
    if (dims[4] != kD || (dims[6] != 2 && dims[6] != 4)) return COBEVT_ERR_UNSUPPORTED;
    BevQueryTrainParams p = {};
    p.grid = grid; p.w = w; p.bias = bias; p.c = c; p.dq = dq; p.dx = dx; p.dw = dw; p.dbias = dbias; p.dc = dc;
    p.B = dims[0]; p.n = dims[1]; p.HW = dims[2] * dims[3]; p.W = dims[3]; p.round_bf16 = dims[5] != 0;
    p.grid_stride = dims[7] ? (long)dims[6] * p.HW : 0;
    if (p.B < 1 || p.n < 1 || p.n > kMaxCam || dims[2] < 1 || dims[3] < 1 || p.B > 65535) return COBEVT_ERR_SHAPE;
    // every workgroup ends in atomics on the SAME d x K (+ d) parameter-gradient words: 2,060 workgroups made the image embedding's
    // backward 159 us for 84 MB (profiles/r04_train_amp_kernel_trace.txt); two per CU keep the stream fed and the contention low
    const dim3 grid_dim(pixel_blocks(p.HW, p.B, 512), p.B);
    if (dims[6] == 2) hipLaunchKernelGGL((bev_query_train_kernel<true, 2>), grid_dim, dim3(256), 0, stream, p);
    else hipLaunchKernelGGL((bev_query_train_kernel<true, 4>), grid_dim, dim3(256), 0, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
