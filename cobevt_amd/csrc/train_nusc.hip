// Backward kernels of the nuScenes SinBEVT training path (gfx950) - what torch autograd + cuDNN do for the reference under
// nuscenes/cross_view_transformer/model/model_module.py:35-60 (Lightning training_step: loss.backward() through
// backbones/efficientnet.py:85-96, decoder.py:27-36 and losses.py:27-84) for the pieces the OPV2V path does not have:
//   swish (x sigmoid(x)) of efficientnet-pytorch's MBConvBlock, both directions;
//   the weight gradient of the depthwise k x k convolution (the input gradient is the forward kernel on flipped taps);
//   the adjoint of the align_corners bilinear resize of the decoder (nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True));
//   the gradient of the masked-mean sigmoid focal loss (BinarySegmentationLoss / CenterLoss around fvcore's sigmoid_focal_loss).
// All HBM-bound elementwise / reduction work on channels-last maps, 8 channels (one 16-byte bf16 piece) per lane.
#include "warp_common.hpp"

namespace cobevt {
namespace {

constexpr int kThreads = 256;

// ---- swish: dy == nullptr: out = x sigmoid(x); else out = dy (s + x s (1 - s)), s = sigmoid(x) ----
template <typename T>
__global__ __launch_bounds__(kThreads) void swish_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ out, long groups) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;
    if (i >= groups) return;
    float v[8], g[8], o[8];
    load8<T>(x + i * 8, v);
    if (dy) load8<T>(dy + i * 8, g);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const float s = 1.0f / (1.0f + __expf(-v[e]));
        o[e] = dy ? g[e] * (s + v[e] * s * (1.f - s)) : v[e] * s;
    }
    store8<T>(out + i * 8, o);
}

// ---- depthwise weight gradient: dw[(a * k + b)][c] = sum over (n, oy, ox) of dy[n][oy][ox][c] * x[n][oy s - pt + a][ox s - pl + b][c] ----
// grid = (pixel blocks, k tap rows); a thread owns one 8-channel group and walks its share of the block's output pixels with the k taps
// of the row as k x 8 accumulators; the threads of a group meet in LDS, one fp32 atomic per (workgroup, tap, channel) (dw zero-initialised)
template <typename T, int K>
__global__ __launch_bounds__(kThreads) void depthwise_wgrad_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dw,
                                                                   int N, int H, int W, int C, int stride, int pt, int pl, int Ho, int Wo,
                                                                   long pixels, int per_block) {
    const int G = C >> 3;                                          // <= 256 (entry point)
    const int a = blockIdx.y;
    __shared__ float red[kThreads][9];
    const int gl = threadIdx.x % G, r0 = threadIdx.x / G, rstep = kThreads / G;       // threads with r0 >= rstep idle
    float acc[K][8];
#pragma unroll
    for (int b = 0; b < K; ++b)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[b][e] = 0.f;
    const long lo = (long)blockIdx.x * per_block, hi = lo + per_block < pixels ? lo + per_block : pixels;
    if (r0 < rstep) {
        for (long p = lo + r0; p < hi; p += rstep) {
            const int ox = (int)(p % Wo);
            const long q = p / Wo;
            const int oy = (int)(q % Ho);
            const long n = q / Ho;
            const int iy = oy * stride - pt + a;
            if (iy < 0 || iy >= H) continue;
            float g[8];
            load8<T>(dy + p * C + gl * 8, g);
#pragma unroll
            for (int b = 0; b < K; ++b) {
                const int ix = ox * stride - pl + b;
                if (ix < 0 || ix >= W) continue;
                float v[8];
                load8<T>(x + ((n * H + iy) * W + ix) * C + gl * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[b][e] += g[e] * v[e];
            }
        }
    }
#pragma unroll
    for (int b = 0; b < K; ++b) {
        __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = acc[b][e];
        __syncthreads();
        if (threadIdx.x < G) {
            float s[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) s[e] = 0.f;
            for (int r = 0; r < rstep; ++r)
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += red[r * G + threadIdx.x][e];
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (s[e] != 0.f) atomicAdd(dw + (size_t)(a * K + b) * C + threadIdx.x * 8 + e, s[e]);
        }
    }
}

// ---- adjoint of the align_corners bilinear resize (elementwise.hip resize_kernel mode 1): dx (N, H, W, C) fp32 zero-initialised ----
template <typename T>
__global__ __launch_bounds__(kThreads) void resize_bilinear_bwd_kernel(const T* __restrict__ dy, float* __restrict__ dx, long items, int H, int W,
                                                                       int C, int Ho, int Wo) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int G = C >> 3;
    const int gl = (int)(gid % G);
    const long pix = gid / G;
    const int ow = (int)(pix % Wo), oh = (int)((pix / Wo) % Ho);
    const long n = pix / ((long)Wo * Ho);
    const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
    const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
    const float fy = oh * sh, fx = ow * sw;
    const int y0 = (int)fy, x0 = (int)fx;
    const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
    const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
    float g[8];
    load8<T>(dy + pix * C + gl * 8, g);
    float* base = dx + n * H * W * C + gl * 8;
    const float w00 = hy * hx, w01 = hy * lx, w10 = ly * hx, w11 = ly * lx;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        atomicAdd(base + ((size_t)y0 * W + x0) * C + e, w00 * g[e]);
        if (w01 != 0.f) atomicAdd(base + ((size_t)y0 * W + x1) * C + e, w01 * g[e]);
        if (w10 != 0.f) atomicAdd(base + ((size_t)y1 * W + x0) * C + e, w10 * g[e]);
        if (w11 != 0.f) atomicAdd(base + ((size_t)y1 * W + x1) * C + e, w11 * g[e]);
    }
}

// ---- gradient of the masked-mean sigmoid focal loss (postprocess.hip focal_partial_kernel) w.r.t. the logits ----
// stats: the forward's out[3] = {mean, sum, count}; gscale: the upstream gradient of the mean (a device scalar)
__global__ __launch_bounds__(kThreads) void focal_bwd_kernel(const float* __restrict__ pred, const float* __restrict__ label,
                                                             const unsigned char* __restrict__ visibility, const unsigned int* __restrict__ label_mask,
                                                             const float* __restrict__ stats, const float* __restrict__ gscale, float* __restrict__ dpred,
                                                             int C, int NL, int hw, int min_visibility, float alpha, float gamma, int soft_label) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * kThreads + threadIdx.x;
    if (pix >= hw) return;
    const bool kept = !(min_visibility >= 0 && (int)visibility[(size_t)n * hw + pix] < min_visibility);
    const float cnt = stats[2];
    const float k = (kept && cnt > 0.f) ? gscale[0] / cnt : 0.f;
    for (int c = 0; c < C; ++c) {
        float t;
        if (soft_label) {
            t = label[((size_t)n * NL + c) * hw + pix];
        } else {
            t = 0.f;
            for (int l = 0; l < NL; ++l)
                if ((label_mask[c] >> l) & 1u) t = fmaxf(t, label[((size_t)n * NL + l) * hw + pix]);
        }
        const size_t i = ((size_t)n * C + c) * hw + pix;
        const float x = pred[i];
        const float p = 1.0f / (1.0f + expf(-x));
        const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));
        const float pt = p * t + (1.f - p) * (1.f - t);
        const float om = 1.f - pt;
        // d/dx [ce (1 - pt)^gamma] = (p - t)(1 - pt)^gamma - ce gamma (1 - pt)^(gamma - 1) p (1 - p)(2 t - 1)
        float d = (p - t) * powf(om, gamma);
        if (gamma != 0.f && om > 0.f) d -= ce * gamma * powf(om, gamma - 1.f) * p * (1.f - p) * (2.f * t - 1.f);
        if (alpha >= 0.f) d *= alpha * t + (1.f - alpha) * (1.f - t);
        dpred[i] = k * d;
    }
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_swish(const void* x, const void* dy, void* out, int dtype, long n, hipStream_t stream) {
    if (!x || !out) return COBEVT_ERR_ARG;
    if (n < 8 || n % 8) return COBEVT_ERR_SHAPE;
    const long groups = n / 8;
    const dim3 grid((unsigned)((groups + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(swish_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)out, groups);
    else if (dtype == 1) hipLaunchKernelGGL(swish_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, (const float*)dy, (float*)out, groups);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_depthwise_wgrad(const void* x, const void* dy, float* dw, const int* dims, hipStream_t stream) {
    // dims: [dtype, N, H, W, C, k, stride, pad_top, pad_left, Ho, Wo]; dw fp32 [k * k][C], zero-initialised
    if (!x || !dy || !dw || !dims) return COBEVT_ERR_ARG;
    const int dtype = dims[0], N = dims[1], H = dims[2], W = dims[3], C = dims[4], k = dims[5], stride = dims[6], pt = dims[7], pl = dims[8],
              Ho = dims[9], Wo = dims[10];
    if (N < 1 || H < 1 || W < 1 || C < 8 || C % 8 || C > 8 * kThreads || stride < 1 || pt < 0 || pl < 0 || Ho < 1 || Wo < 1) return COBEVT_ERR_SHAPE;
    if (k != 3 && k != 5) return COBEVT_ERR_UNSUPPORTED;
    const long pixels = (long)N * Ho * Wo;
    int blocks = (int)((pixels + 1023) / 1024);
    if (blocks > 256) blocks = 256;
    if (blocks < 1) blocks = 1;
    const int per_block = (int)((pixels + blocks - 1) / blocks);
    const dim3 grid(blocks, k);
#define COBEVT_DWG(T_, K_) hipLaunchKernelGGL((depthwise_wgrad_kernel<T_, K_>), grid, dim3(kThreads), 0, stream, (const T_*)x, (const T_*)dy, dw, N, H, W, C, \
                                              stride, pt, pl, Ho, Wo, pixels, per_block)
    if (dtype == 0) { if (k == 3) COBEVT_DWG(bf16_t, 3); else COBEVT_DWG(bf16_t, 5); }
    else if (dtype == 1) { if (k == 3) COBEVT_DWG(float, 3); else COBEVT_DWG(float, 5); }
    else return COBEVT_ERR_ARG;
#undef COBEVT_DWG
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_resize_bilinear_bwd(const void* dy, float* dx, int dtype, int N, int H, int W, int C, int Ho, int Wo, hipStream_t stream) {
    // dy (N, Ho, Wo, C) -> dx fp32 (N, H, W, C), zero-initialised: adjoint of cobevt_resize_nhwc mode 1 (align_corners bilinear)
    if (!dy || !dx) return COBEVT_ERR_ARG;
    if (C < 8 || C % 8 || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1) return COBEVT_ERR_SHAPE;
    const long items = (long)N * Ho * Wo * (C >> 3);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(resize_bilinear_bwd_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)dy, dx, items, H, W, C, Ho, Wo);
    else if (dtype == 1) hipLaunchKernelGGL(resize_bilinear_bwd_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)dy, dx, items, H, W, C, Ho, Wo);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_sigmoid_focal_loss_bwd(const float* pred, const float* label, const unsigned char* visibility, const unsigned int* label_mask,
                                             const float* stats, const float* gscale, float* dpred, int N, int C, int NL, int hw,
                                             int min_visibility, float alpha, float gamma, int soft_label, hipStream_t stream) {
    if (!pred || !label || !stats || !gscale || !dpred) return COBEVT_ERR_ARG;
    if (!soft_label && !label_mask) return COBEVT_ERR_ARG;
    if (min_visibility >= 0 && !visibility) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || C < 1 || NL < 1 || NL > 32 || hw < 1 || (soft_label && NL != C)) return COBEVT_ERR_SHAPE;
    const dim3 grid((hw + kThreads - 1) / kThreads, N);
    hipLaunchKernelGGL(focal_bwd_kernel, grid, dim3(kThreads), 0, stream, pred, label, visibility, label_mask, stats, gscale, dpred, C, NL, hw,
                       min_visibility, alpha, gamma, soft_label);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
