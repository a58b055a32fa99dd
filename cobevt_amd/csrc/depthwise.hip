// MBConv pieces of the nuScenes image backbone (EfficientNet) that are not GEMM-shaped (gfx950):
//   depthwise_conv_kernel  k x k depthwise convolution (groups == channels), TensorFlow-"same" static padding given as
//                          (top, left) with the bottom / right handled by the bounds checks, folded BatchNorm, swish
//   spatial_mean_kernel    squeeze: per-(image, channel) mean over the pixels, fixed summation order (deterministic)
//   se_gate_kernel         excitation: sigmoid(W2 . swish(W1 . mean + b1) + b2)  -> (image, channel) gate, fp32
//   channel_gate_kernel    x * gate
// All HBM-bound streaming passes over channels-last maps: a lane owns 8 consecutive channels (16 bytes of bf16) of one
// pixel, the C/8 lanes of a pixel are adjacent so every access is coalesced; the depthwise taps re-read their input
// through L1 / L2 (k^2 / stride^2 reads per element, arithmetic intensity k^2 MACs per 2 bytes).
// reference: efficientnet-pytorch 0.7.1 model.py MBConvBlock.forward (third-party, restated in oracle/efficientnet.py),
// wrapped by nuscenes/cross_view_transformer/model/backbones/efficientnet.py:24-96.
#include "common.hpp"

namespace cobevt {

template <typename T> __device__ __forceinline__ void dw_load8(const T* p, float* v) {
    if constexpr (Elem<T>::kIsBf16) {
        chunk_to_f32<T>(*(const uint4*)p, v);
    } else {
        chunk_to_f32<T>(*(const uint4*)p, v);
        chunk_to_f32<T>(*(const uint4*)(p + 4), v + 4);
    }
}
template <typename T> __device__ __forceinline__ void dw_store8(T* p, const float* v) {
    if constexpr (Elem<T>::kIsBf16) {
        *(uint4*)p = f32_to_chunk<T>(v);
    } else {
        *(uint4*)p = f32_to_chunk<T>(v);
        *(uint4*)(p + 4) = f32_to_chunk<T>(v + 4);
    }
}

struct DwParams {
    const void* in; const float* wgt; const float* bias; void* out;
    int N, H, W, C, k, stride, pad_t, pad_l, Ho, Wo, act;
};

// wgt: [k*k][C] fp32 (BatchNorm scale folded in), bias [C] fp32
template <typename T>
__global__ __launch_bounds__(256) void depthwise_conv_kernel(DwParams p) {
    const int G = p.C >> 3;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long pix = item / G;
    const int g = (int)(item - pix * G);
    if (pix >= (long)p.N * p.Ho * p.Wo) return;
    const int n = (int)(pix / (p.Ho * p.Wo)), rem = (int)(pix - (long)n * p.Ho * p.Wo);
    const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
    const T* in = (const T*)p.in + (size_t)n * p.H * p.W * p.C + g * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = p.bias[g * 8 + e];
    const int iy0 = oy * p.stride - p.pad_t, ix0 = ox * p.stride - p.pad_l;
    for (int kh = 0; kh < p.k; ++kh) {
        const int iy = iy0 + kh;
        if (iy < 0 || iy >= p.H) continue;
        for (int kw = 0; kw < p.k; ++kw) {
            const int ix = ix0 + kw;
            if (ix < 0 || ix >= p.W) continue;
            float x[8];
            dw_load8<T>(in + ((size_t)iy * p.W + ix) * p.C, x);
            const float4 w0 = *(const float4*)(p.wgt + (size_t)(kh * p.k + kw) * p.C + g * 8);
            const float4 w1 = *(const float4*)(p.wgt + (size_t)(kh * p.k + kw) * p.C + g * 8 + 4);
            acc[0] = fmaf(x[0], w0.x, acc[0]); acc[1] = fmaf(x[1], w0.y, acc[1]);
            acc[2] = fmaf(x[2], w0.z, acc[2]); acc[3] = fmaf(x[3], w0.w, acc[3]);
            acc[4] = fmaf(x[4], w1.x, acc[4]); acc[5] = fmaf(x[5], w1.y, acc[5]);
            acc[6] = fmaf(x[6], w1.z, acc[6]); acc[7] = fmaf(x[7], w1.w, acc[7]);
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = apply_act(acc[e], p.act);
    dw_store8<T>((T*)p.out + (size_t)pix * p.C + g * 8, acc);
}

// out[n][c] = mean over the hw pixels of in[n][.][c].  grid (ceil(C / 64), N), 256 threads = 32 pixel lanes x 8 chunks of
// 8 channels; every lane walks pixels lane, lane + 32, ... in order and the 32 partial sums meet in a fixed-order LDS tree.
template <typename T>
__global__ __launch_bounds__(256) void spatial_mean_kernel(const T* in, float* out, int hw, int C) {
    __shared__ float part[32][8][8];
    const int n = blockIdx.y, tp = threadIdx.x >> 3, tc = threadIdx.x & 7;
    const int c0 = (blockIdx.x * 8 + tc) * 8;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    if (c0 < C) {
        const T* src = in + (size_t)n * hw * C + c0;
        for (int px = tp; px < hw; px += 32) {
            float x[8];
            dw_load8<T>(src + (size_t)px * C, x);
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] += x[e];
        }
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) part[tp][tc][e] = acc[e];
    __syncthreads();
    for (int s = 16; s > 0; s >>= 1) {
        if (tp < s) {
#pragma unroll
            for (int e = 0; e < 8; ++e) part[tp][tc][e] += part[tp + s][tc][e];
        }
        __syncthreads();
    }
    if (tp == 0 && c0 < C) {
        const float inv = 1.0f / (float)hw;
#pragma unroll
        for (int e = 0; e < 8; ++e) out[(size_t)n * C + c0 + e] = part[0][tc][e] * inv;
    }
}

// one workgroup per image: r = swish(W1 . mean + b1) (Cs values), gate = sigmoid(W2 . r + b2) (C values)
__global__ __launch_bounds__(256) void se_gate_kernel(const float* mean, const float* w1, const float* b1, const float* w2,
                                                      const float* b2, float* gate, int C, int Cs) {
    extern __shared__ float sq[];                    // Cs squeezed activations
    const int n = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const float* m = mean + (size_t)n * C;
    for (int j = wave; j < Cs; j += 4) {
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s = fmaf(m[c], w1[(size_t)j * C + c], s);
        s = wave_sum_xor(s, 64);
        if (lane == 0) sq[j] = apply_act(s + b1[j], 3);
    }
    __syncthreads();
    for (int c = tid; c < C; c += 256) {
        float s = b2[c];
        for (int j = 0; j < Cs; ++j) s = fmaf(sq[j], w2[(size_t)c * Cs + j], s);
        gate[(size_t)n * C + c] = apply_act(s, 4);
    }
}

template <typename T>
__global__ __launch_bounds__(256) void channel_gate_kernel(const T* in, const float* gate, T* out, long npix, int hw, int C) {
    const int G = C >> 3;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const long pix = item / G;
    const int g = (int)(item - pix * G);
    if (pix >= npix) return;
    const int n = (int)(pix / hw);
    float x[8];
    dw_load8<T>(in + (size_t)pix * C + g * 8, x);
    const float4 a = *(const float4*)(gate + (size_t)n * C + g * 8), b = *(const float4*)(gate + (size_t)n * C + g * 8 + 4);
    x[0] *= a.x; x[1] *= a.y; x[2] *= a.z; x[3] *= a.w; x[4] *= b.x; x[5] *= b.y; x[6] *= b.z; x[7] *= b.w;
    dw_store8<T>(out + (size_t)pix * C + g * 8, x);
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_depthwise_conv_nhwc(const void* in, const float* wgt, const float* bias, void* out, const int* dims,
                                          hipStream_t stream) {
    // dims: [dtype, N, H, W, C, k, stride, pad_top, pad_left, Ho, Wo, act]
    if (!in || !wgt || !bias || !out || !dims) return COBEVT_ERR_ARG;
    DwParams p;
    p.in = in; p.wgt = wgt; p.bias = bias; p.out = out;
    p.N = dims[1]; p.H = dims[2]; p.W = dims[3]; p.C = dims[4]; p.k = dims[5]; p.stride = dims[6];
    p.pad_t = dims[7]; p.pad_l = dims[8]; p.Ho = dims[9]; p.Wo = dims[10]; p.act = dims[11];
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.C < 8 || p.C % 8 || p.k < 1 || p.k > 7 || p.stride < 1 || p.Ho < 1 || p.Wo < 1)
        return COBEVT_ERR_SHAPE;
    if (p.pad_t < 0 || p.pad_l < 0 || p.act < 0 || p.act > 4) return COBEVT_ERR_ARG;
    if ((long)(p.Ho - 1) * p.stride - p.pad_t >= p.H || (long)(p.Wo - 1) * p.stride - p.pad_l >= p.W) return COBEVT_ERR_SHAPE;
    const long items = (long)p.N * p.Ho * p.Wo * (p.C / 8);
    const long blocks = (items + 255) / 256;
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    if (dims[0] == 0) hipLaunchKernelGGL(depthwise_conv_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else if (dims[0] == 1) hipLaunchKernelGGL(depthwise_conv_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, p);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_spatial_mean_nhwc(const void* in, float* out, int dtype, int N, int hw, int C, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || hw < 1 || C < 8 || C % 8) return COBEVT_ERR_SHAPE;
    const dim3 grid((C + 63) / 64, N), block(256);
    if (dtype == 0) hipLaunchKernelGGL(spatial_mean_kernel<bf16_t>, grid, block, 0, stream, (const bf16_t*)in, out, hw, C);
    else if (dtype == 1) hipLaunchKernelGGL(spatial_mean_kernel<float>, grid, block, 0, stream, (const float*)in, out, hw, C);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_se_gate(const float* mean, const float* w_reduce, const float* b_reduce, const float* w_expand,
                              const float* b_expand, float* gate, int N, int C, int Cs, hipStream_t stream) {
    if (!mean || !w_reduce || !b_reduce || !w_expand || !b_expand || !gate) return COBEVT_ERR_ARG;
    if (N < 1 || C < 1 || Cs < 1 || Cs > 4096) return COBEVT_ERR_SHAPE;
    hipLaunchKernelGGL(se_gate_kernel, dim3(N), dim3(256), (size_t)Cs * 4, stream, mean, w_reduce, b_reduce, w_expand, b_expand,
                       gate, C, Cs);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_channel_gate_nhwc(const void* in, const float* gate, void* out, int dtype, int N, int hw, int C,
                                        hipStream_t stream) {
    if (!in || !gate || !out) return COBEVT_ERR_ARG;
    if (N < 1 || hw < 1 || C < 8 || C % 8) return COBEVT_ERR_SHAPE;
    const long npix = (long)N * hw, items = npix * (C / 8), blocks = (items + 255) / 256;
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    if (dtype == 0) hipLaunchKernelGGL(channel_gate_kernel<bf16_t>, dim3((unsigned)blocks), dim3(256), 0, stream, (const bf16_t*)in, gate, (bf16_t*)out, npix, hw, C);
    else if (dtype == 1) hipLaunchKernelGGL(channel_gate_kernel<float>, dim3((unsigned)blocks), dim3(256), 0, stream, (const float*)in, gate, (float*)out, npix, hw, C);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
