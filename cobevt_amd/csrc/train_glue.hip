// Training glue between the convolution / attention kernels, both directions, channels-last (N, H, W, C) maps (gfx950):
// what torch autograd + cuDNN do for the reference under opv2v/opencood/tools/train_camera.py:143-179 -
//   BatchNorm2d with batch statistics (+ running-stat update) or frozen statistics, fused with the residual add and the ReLU that
//     follow it in torchvision's BasicBlock / Bottleneck (resnet_ms.py:67-74, fax_modules.py:10,472-489), NaiveDecoder
//     (naive_decoder.py:78-87) and the pre-activation BN -> ReLU of fax_modules.py:281-292;
//   MaxPool2d(3, 2, 1) of the ResNet stem (resnet_ms.py:70) backward; nn.PixelUnshuffle(2) (fax_modules.py:479) and nearest x2
//     up-sampling (naive_decoder.py:84) in both directions; the STTF bilinear warp's input gradient (corpbevt.py:28-64,
//     torch_transformation_utils.py:317-355); bias gradients (column sums).
// All of it is HBM-bound streaming work: one 16-byte piece (8 channels) per lane, per-channel reductions in fp64 partial sums
// per workgroup + one fp64 atomic per (workgroup, channel) (E[x^2] - E[x]^2 in fp32 loses the variance of activations whose
// mean dominates; torch's batch_norm uses Welford), scatter-type gradients as fp32 atomics on the input-gradient map.
#include "warp_common.hpp"

namespace cobevt {
namespace {

constexpr int kThreads = 256;

// ---- per-channel sums over rows: sum[c] += x, sumsq[c] += x^2 (BatchNorm statistics; column sum for bias gradients) ----
// grid.x = row blocks; a thread owns one 8-channel group (gl) and walks rows gid, gid + stride, ...
template <typename T, bool SQ>
__global__ __launch_bounds__(kThreads) void channel_sums_kernel(const T* __restrict__ x, const float* __restrict__ shift,
                                                                double* __restrict__ partial, long rows, int C, int rows_per_block) {
    // partial: [gridDim.x][2][C] fp64 - this workgroup's sums; combined by reduce_partials_kernel.  (Atomics on the C result
    // words serialise: with 1024 workgroups a launch took ~50 us whatever the map size.)
    // sums of (x - shift[c]) and (x - shift[c])^2 (shift nullable = 0): with the shift near the mean (BatchNorm passes its running
    // mean) the variance no longer comes out of a cancellation, so a thread accumulates its <= ~64 rows in fp32 (v_pk_add rate
    // instead of fp64 conversions per element); threads and workgroups are combined in fp64
    const int G = C >> 3;
    const int gl = threadIdx.x % G, r0 = threadIdx.x / G, rstep = kThreads / G;      // threads with r0 >= rstep idle (256 % G != 0)
    const long row_lo = (long)blockIdx.x * rows_per_block;
    const long row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
    float s[8], q[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { s[e] = 0.f; q[e] = 0.f; sh[e] = shift ? shift[gl * 8 + e] : 0.f; }
    if (r0 < rstep) {
        long r = row_lo + r0;
        for (; r + rstep < row_hi; r += 2 * rstep) {          // two rows in flight
            float v[8], u[8];
            load8<T>(x + r * C + gl * 8, v);
            load8<T>(x + (r + rstep) * C + gl * 8, u);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float a = v[e] - sh[e], b = u[e] - sh[e];
                s[e] += a + b;
                if (SQ) q[e] += a * a + b * b;
            }
        }
        if (r < row_hi) {
            float v[8];
            load8<T>(x + r * C + gl * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) { const float a = v[e] - sh[e]; s[e] += a; if (SQ) q[e] += a * a; }
        }
    }
    __shared__ float red[kThreads][9];
    for (int pass = 0; pass < (SQ ? 2 : 1); ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = pass ? q[e] : s[e];
        __syncthreads();
        if (threadIdx.x < G) {
            double t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = 0.0;
            for (int k = 0; k < rstep; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += (double)red[k * G + threadIdx.x][e];
#pragma unroll
            for (int e = 0; e < 8; ++e) partial[((size_t)blockIdx.x * 2 + pass) * C + threadIdx.x * 8 + e] = t[e];
        }
    }
}

// out[j] = sum over the nblk workgroups of partial[b * stride + j], j < n: one wave per output word (64 lanes share out the
// workgroups, then a shuffle reduction) - a thread per word walking 512 partials serially cost 45-60 us per launch
__global__ __launch_bounds__(256) void reduce_partials_strided_kernel(const double* __restrict__ partial, int nblk, int n, int stride,
                                                                      double* __restrict__ out_d, float* __restrict__ out_f) {
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (j >= n) return;
    double t = 0.0;
    for (int b = lane; b < nblk; b += 64) t += partial[(size_t)b * stride + j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);
    if (lane == 0) {
        if (out_d) out_d[j] = t;
        if (out_f) out_f[j] = (float)t;
    }
}

// ---- BatchNorm finalize: stats (training: from the sums; eval: the running statistics) -> per-channel scale / shift, mean / rstd
// saved for backward, running statistics updated in place (momentum, unbiased variance) as nn.BatchNorm2d does
// reduce_partials_strided_kernel + bn_finalize_kernel in one launch (training statistics): a wave per channel sums the workgroups' partial
// pairs [blk][2][C] and its first lane finishes the channel (statistics of x - running_mean when `shifted`, read before the update)
__global__ __launch_bounds__(256) void bn_reduce_finalize_kernel(const double* __restrict__ partial, int nblk, const float* gamma, const float* beta,
                                                                 float* running_mean, float* running_var, float* scale, float* shift, float* mean_out,
                                                                 float* rstd_out, int C, double rows, float eps, float momentum, int shifted,
                                                                 long long* batches_tracked) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (c >= C) return;
    double s = 0.0, q = 0.0;
    for (int b = lane; b < nblk; b += 64) {
        s += partial[((size_t)b * 2) * C + c];
        q += partial[((size_t)b * 2 + 1) * C + c];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { s += __shfl_xor(s, o, 64); q += __shfl_xor(q, o, 64); }
    if (lane) return;
    if (c == 0 && batches_tracked) *batches_tracked += 1;
    const double d = s / rows;
    double var = q / rows - d * d;
    const double mean = d + (shifted ? (double)running_mean[c] : 0.0);
    if (var < 0.0) var = 0.0;
    if (running_mean) {
        const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
        running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
        running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, bb = beta ? beta[c] : 0.f;
    scale[c] = g * rstd;
    shift[c] = bb - (float)mean * g * rstd;
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
}

__global__ void bn_finalize_kernel(const double* sum, const double* sumsq, const float* gamma, const float* beta, float* running_mean,
                                   float* running_var, float* scale, float* shift, float* mean_out, float* rstd_out, int C,
                                   double rows, float eps, float momentum, int training, int shifted, long long* batches_tracked) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    if (c >= C) return;
    if (c == 0 && batches_tracked) *batches_tracked += 1;          // nn.BatchNorm2d.num_batches_tracked (one launch less per BatchNorm)
    double mean, var;
    if (training) {
        // the sums were taken of x - running_mean (shifted != 0: channel_sums with shift = running_mean, read BEFORE its update below)
        const double d = sum[c] / rows;
        var = sumsq[c] / rows - d * d;
        mean = d + (shifted ? (double)running_mean[c] : 0.0);
        if (var < 0.0) var = 0.0;
        if (running_mean) {
            const double unbiased = rows > 1.0 ? var * rows / (rows - 1.0) : var;
            running_mean[c] = (float)((1.0 - momentum) * (double)running_mean[c] + momentum * mean);
            running_var[c] = (float)((1.0 - momentum) * (double)running_var[c] + momentum * unbiased);
        }
    } else {
        mean = (double)running_mean[c];
        var = (double)running_var[c];
    }
    const float rstd = (float)(1.0 / sqrt(var + (double)eps));
    const float g = gamma ? gamma[c] : 1.f, b = beta ? beta[c] : 0.f;
    scale[c] = g * rstd;
    shift[c] = b - (float)mean * g * rstd;
    mean_out[c] = (float)mean;
    rstd_out[c] = rstd;
}

// ---- y = act(x * scale[c] + shift[c] (+ residual)) ; act: 0 none, 1 ReLU
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_apply_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ scale,
                                                            const float* __restrict__ shift, T* __restrict__ y, long items, int C, int act) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int G = C >> 3;
    const int gl = (int)(gid % G);
    float v[8], r[8];
    load8<T>(x + gid * 8, v);
    if (res) load8<T>(res + gid * 8, r);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float t = fmaf(v[e], scale[gl * 8 + e], shift[gl * 8 + e]);
        if (res) t += r[e];
        v[e] = act == 1 ? fmaxf(t, 0.f) : t;
    }
    store8<T>(y + gid * 8, v);
}

// The same with the workgroup count fixed and every thread walking pieces gid, gid + T, ... (T = all threads, a multiple of the channel
// groups G, so a thread stays on ONE 8-channel group and holds its 16 coefficients in registers).  The piece-per-thread form above
// issues 16 four-byte coefficient loads beside its two or three 16-byte data loads: the texture-address path, not HBM, set its time
// (5 x the streaming bound on the 5-agent encoder maps).  Four pieces in flight per thread.
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_apply_walk_kernel(const T* __restrict__ x, const T* __restrict__ res, const float* __restrict__ scale,
                                                                 const float* __restrict__ shift, T* __restrict__ y, long items, int C, int act) {
    const int G = C >> 3;
    // a workgroup walks ONE contiguous span of pieces (a multiple of 256, hence of G): neighbouring waves stay in the same DRAM pages /
    // TLB entries (a grid-wide stride put every wave of the chip on its own 2 MiB page each round)
    constexpr long step = kThreads;
    const long span = ((items + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;
    long gid = (long)blockIdx.x * span + threadIdx.x;
    items = items < (long)(blockIdx.x + 1) * span ? items : (long)(blockIdx.x + 1) * span;
    const int gl = (int)(gid % G);
    float sc[8], sh[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) { sc[e] = scale[gl * 8 + e]; sh[e] = shift[gl * 8 + e]; }
    auto finish = [&](float* v, const float* r) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float t = fmaf(v[e], sc[e], sh[e]);
            if (res) t += r[e];
            v[e] = act == 1 ? fmaxf(t, 0.f) : t;
        }
    };
    for (; gid + 3 * step < items; gid += 4 * step) {
        float v[4][8], r[4][8];
#pragma unroll
        for (int u = 0; u < 4; ++u) load8<T>(x + (gid + u * step) * 8, v[u]);
        if (res) {
#pragma unroll
            for (int u = 0; u < 4; ++u) load8<T>(res + (gid + u * step) * 8, r[u]);
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) { finish(v[u], r[u]); store8<T>(y + (gid + u * step) * 8, v[u]); }
    }
    for (; gid < items; gid += step) {
        float v[8], r[8];
        load8<T>(x + gid * 8, v);
        if (res) load8<T>(res + gid * 8, r);
        finish(v, r);
        store8<T>(y + gid * 8, v);
    }
}

// ---- backward, pass 1: g = dy * [y > 0] ; dbeta[c] += g ; dgamma[c] += g * xhat, xhat = (x - mean) * rstd
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_reduce_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                                                 const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                 double* __restrict__ partial, long rows, int C, int act, int rows_per_block) {
    const int G = C >> 3;
    const int gl = threadIdx.x % G, r0 = threadIdx.x / G, rstep = kThreads / G;
    const long row_lo = (long)blockIdx.x * rows_per_block;
    const long row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
    float sg[8], sb[8];                              // per-thread partials in fp32 (<= ~64 rows), combined in fp64
#pragma unroll
    for (int e = 0; e < 8; ++e) { sg[e] = 0.f; sb[e] = 0.f; }
    if (r0 < rstep) {
        float mu[8], rs[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { mu[e] = mean[gl * 8 + e]; rs[e] = rstd[gl * 8 + e]; }
        for (long r = row_lo + r0; r < row_hi; r += rstep) {
            float xv[8], gv[8], yv[8];
            load8<T>(x + r * C + gl * 8, xv);
            load8<T>(dy + r * C + gl * 8, gv);
            if (act == 1) load8<T>(y + r * C + gl * 8, yv);
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float g = (act == 1 && !(yv[e] > 0.f)) ? 0.f : gv[e];
                sb[e] += g;
                sg[e] += g * ((xv[e] - mu[e]) * rs[e]);
            }
        }
    }
    __shared__ float red[kThreads][9];
    for (int pass = 0; pass < 2; ++pass) {
        if (pass) __syncthreads();
#pragma unroll
        for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = pass ? sb[e] : sg[e];
        __syncthreads();
        if (threadIdx.x < G) {
            double t[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) t[e] = 0.0;
            for (int k = 0; k < rstep; ++k)
#pragma unroll
                for (int e = 0; e < 8; ++e) t[e] += (double)red[k * G + threadIdx.x][e];
#pragma unroll
            for (int e = 0; e < 8; ++e) partial[((size_t)blockIdx.x * 2 + pass) * C + threadIdx.x * 8 + e] = t[e];
        }
    }
}

// ---- backward, pass 2: dx = gamma rstd (g - (dbeta + xhat dgamma) / rows)   (training)   |   dx = gamma rstd g   (frozen statistics)
//      dres = g (the gradient of the residual branch), when wanted
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                                                const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                const float* __restrict__ gamma, const double* __restrict__ dgamma,
                                                                const double* __restrict__ dbeta, T* __restrict__ dx, T* __restrict__ dres,
                                                                long items, int C, float inv_rows, int act, int training) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int G = C >> 3;
    const int gl = (int)(gid % G);
    float xv[8], gv[8], yv[8], o[8];
    load8<T>(x + gid * 8, xv);
    load8<T>(dy + gid * 8, gv);
    if (act == 1) load8<T>(y + gid * 8, yv);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = gl * 8 + e;
        const float g = (act == 1 && !(yv[e] > 0.f)) ? 0.f : gv[e];
        gv[e] = g;
        const float k = (gamma ? gamma[c] : 1.f) * rstd[c];
        if (training) {
            const float xhat = (xv[e] - mean[c]) * rstd[c];
            o[e] = k * (g - ((float)dbeta[c] + xhat * (float)dgamma[c]) * inv_rows);
        } else {
            o[e] = k * g;
        }
    }
    store8<T>(dx + gid * 8, o);
    if (dres) store8<T>(dres + gid * 8, gv);
}

// bn_bwd_apply_kernel as a walk (see bn_apply_walk_kernel): the 8 channels' k = gamma rstd, mean, rstd, dbeta, dgamma in registers; the
// arithmetic per element is unchanged.  Two pieces in flight (three or four 16-byte streams each).
template <typename T>
__global__ __launch_bounds__(kThreads) void bn_bwd_apply_walk_kernel(const T* __restrict__ x, const T* __restrict__ y, const T* __restrict__ dy,
                                                                     const float* __restrict__ mean, const float* __restrict__ rstd,
                                                                     const float* __restrict__ gamma, const double* __restrict__ dgamma,
                                                                     const double* __restrict__ dbeta, T* __restrict__ dx, T* __restrict__ dres,
                                                                     long items, int C, float inv_rows, int act, int training) {
    const int G = C >> 3;
    constexpr long step = kThreads;                  // one contiguous span per workgroup, see bn_apply_walk_kernel
    const long span = ((items + gridDim.x - 1) / gridDim.x + kThreads - 1) / kThreads * kThreads;
    long gid = (long)blockIdx.x * span + threadIdx.x;
    items = items < (long)(blockIdx.x + 1) * span ? items : (long)(blockIdx.x + 1) * span;
    const int gl = (int)(gid % G);
    float k[8], mu[8], rs[8], db[8], dg[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int c = gl * 8 + e;
        rs[e] = rstd[c];
        mu[e] = mean[c];
        k[e] = (gamma ? gamma[c] : 1.f) * rs[e];
        db[e] = training ? (float)dbeta[c] : 0.f;
        dg[e] = training ? (float)dgamma[c] : 0.f;
    }
    auto finish = [&](const float* xv, float* gv, const float* yv, float* o) {
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float g = (act == 1 && !(yv[e] > 0.f)) ? 0.f : gv[e];
            gv[e] = g;
            if (training) {
                const float xhat = (xv[e] - mu[e]) * rs[e];
                o[e] = k[e] * (g - (db[e] + xhat * dg[e]) * inv_rows);
            } else {
                o[e] = k[e] * g;
            }
        }
    };
    for (; gid + step < items; gid += 2 * step) {
        float xv[2][8], gv[2][8], yv[2][8], o[2][8];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            load8<T>(x + (gid + u * step) * 8, xv[u]);
            load8<T>(dy + (gid + u * step) * 8, gv[u]);
            if (act == 1) load8<T>(y + (gid + u * step) * 8, yv[u]);
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            finish(xv[u], gv[u], yv[u], o[u]);
            store8<T>(dx + (gid + u * step) * 8, o[u]);
            if (dres) store8<T>(dres + (gid + u * step) * 8, gv[u]);
        }
    }
    for (; gid < items; gid += step) {
        float xv[8], gv[8], yv[8], o[8];
        load8<T>(x + gid * 8, xv);
        load8<T>(dy + gid * 8, gv);
        if (act == 1) load8<T>(y + gid * 8, yv);
        finish(xv, gv, yv, o);
        store8<T>(dx + gid * 8, o);
        if (dres) store8<T>(dres + gid * 8, gv);
    }
}

// ---- MaxPool2d(3, stride 2, padding 1) backward as a GATHER: an input pixel lies in at most 2 x 2 pooling windows; for each of
// them the window's first maximum (row-major scan, ties -> the first, as torch) is recomputed and the pixel takes that window's
// gradient if it IS that maximum.  No atomics (the scatter form spent 965 us on the 5-agent stem map, 4 x torch's kernel), the
// summation order per pixel is fixed, dx is written once (no zero fill).
template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, float* __restrict__ dx,
                                                               int N, int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const long item = (long)blockIdx.x * kThreads + threadIdx.x;
    const long gid = item / G;
    const int gl = (int)(item % G);
    if (gid >= (long)N * H * W) return;
    const int iw = (int)(gid % W), ih = (int)((gid / W) % H), n = (int)(gid / ((long)W * H));
    float mine[8], acc[8];
    load8<T>(x + gid * C + gl * 8, mine);
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    // windows (oh, ow) containing (ih, iw): oh * 2 - 1 <= ih <= oh * 2 + 1
    const int oh_lo = ih >> 1, oh_hi = (ih + 1) >> 1, ow_lo = iw >> 1, ow_hi = (iw + 1) >> 1;
    for (int oh = oh_lo; oh <= oh_hi; ++oh) {
        if (oh >= Ho) continue;
        for (int ow = ow_lo; ow <= ow_hi; ++ow) {
            if (ow >= Wo) continue;
            float best[8];
            int arg[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = -1; }
            for (int kh = 0; kh < 3; ++kh) {
                const int yy = oh * 2 - 1 + kh;
                if (yy < 0 || yy >= H) continue;
                for (int kw = 0; kw < 3; ++kw) {
                    const int xx = ow * 2 - 1 + kw;
                    if (xx < 0 || xx >= W) continue;
                    float v[8];
                    load8<T>(x + (((size_t)n * H + yy) * W + xx) * C + gl * 8, v);
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (v[e] > best[e] || arg[e] < 0) { best[e] = v[e]; arg[e] = yy * W + xx; }
                }
            }
            float g[8];
            load8<T>(dy + (((size_t)n * Ho + oh) * Wo + ow) * C + gl * 8, g);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (arg[e] == ih * W + iw) acc[e] += g[e];
        }
    }
    float* o = dx + gid * C + gl * 8;
    *(float4*)o = make_float4(acc[0], acc[1], acc[2], acc[3]);
    *(float4*)(o + 4) = make_float4(acc[4], acc[5], acc[6], acc[7]);
}

// The same per 2 x 2 block of input pixels, dx in the map's own type: the block (2a.., 2b..) lies in the four windows {a, a + 1} x {b, b + 1},
// whose maxima are found once (36 loads for four pixels; the pixel-per-thread form above re-derives every window for each of its up to nine
// pixels: 21 loads per pixel) and a pixel takes the gradients of the windows whose first maximum it is.  The fp32 sum of its <= 4 gradients is
// rounded once - what the fp32 map + cast of the old entry point produced, without the 336 MB fp32 map on the 5-agent stem.
template <typename T>
__global__ __launch_bounds__(kThreads) void maxpool_bwd_block_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx,
                                                                     int N, int H, int W, int C, int Ho, int Wo) {
    const int G = C >> 3;
    const int Hb = (H + 1) >> 1, Wb = (W + 1) >> 1;
    const long item = (long)blockIdx.x * kThreads + threadIdx.x;
    const long bid = item / G;
    const int gl = (int)(item % G);
    if (bid >= (long)N * Hb * Wb) return;
    const int b = (int)(bid % Wb), a = (int)((bid / Wb) % Hb), n = (int)(bid / ((long)Wb * Hb));
    float acc[4][8];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[q][e] = 0.f;
#pragma unroll
    for (int wq = 0; wq < 4; ++wq) {
        const int oh = a + (wq >> 1), ow = b + (wq & 1);
        if (oh >= Ho || ow >= Wo) continue;
        float best[8];
        int arg[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) { best[e] = -INFINITY; arg[e] = -1; }
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int yy = oh * 2 - 1 + kh;
            if (yy < 0 || yy >= H) continue;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int xx = ow * 2 - 1 + kw;
                if (xx < 0 || xx >= W) continue;
                float v[8];
                load8<T>(x + (((size_t)n * H + yy) * W + xx) * C + gl * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (v[e] > best[e] || arg[e] < 0) { best[e] = v[e]; arg[e] = yy * W + xx; }
            }
        }
        float g[8];
        load8<T>(dy + (((size_t)n * Ho + oh) * Wo + ow) * C + gl * 8, g);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int pid = (2 * a + (q >> 1)) * W + 2 * b + (q & 1);
#pragma unroll
            for (int e = 0; e < 8; ++e)
                if (arg[e] == pid) acc[q][e] += g[e];
        }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ih = 2 * a + (q >> 1), iw = 2 * b + (q & 1);
        if (ih < H && iw < W) store8<T>(dx + (((size_t)n * H + ih) * W + iw) * C + gl * 8, acc[q]);
    }
}

// ---- mean over the n slabs of a (B, n, inner) tensor (the camera mean of CrossWinAttention, fax_modules.py:243) and its backward
// (every slab gets dout / n): one 16-byte piece per thread, fp32 arithmetic.  backward = 0: in (B, n, inner) -> out (B, inner);
// backward = 1: in (B, inner) -> out (B, n, inner).
template <typename T>
__global__ __launch_bounds__(kThreads) void group_mean_kernel(const T* __restrict__ in, T* __restrict__ out, long pieces, long inner8, int n, int backward) {
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;          // piece of the (B, inner) side
    if (gid >= pieces) return;
    constexpr int E = 16 / sizeof(T);
    const long b = gid / inner8, r = gid - b * inner8;
    const float inv = 1.f / (float)n;
    float v[8], a[8];
    if (backward) {
        load8<T>(in + gid * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= inv;
        for (int k = 0; k < n; ++k) store8<T>(out + ((b * n + k) * inner8 + r) * 8, v);
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] = 0.f;
        for (int k = 0; k < n; ++k) {
            load8<T>(in + ((b * n + k) * inner8 + r) * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) a[e] += v[e];
        }
#pragma unroll
        for (int e = 0; e < 8; ++e) a[e] *= inv;
        store8<T>(out + gid * 8, a);
    }
    (void)E;
}

// ---- nn.PixelUnshuffle(2) on channels-last maps: out[n][h][w][c*4 + 2i + j] = in[n][2h + i][2w + j][c]; inverse = 1: the other way
template <typename T>
__global__ __launch_bounds__(kThreads) void pixel_unshuffle_kernel(const T* __restrict__ in, T* __restrict__ out, long total, int Ho, int Wo,
                                                                   int C, int inverse) {
    const long i = (long)blockIdx.x * kThreads + threadIdx.x;       // index into the UNSHUFFLED tensor (N, Ho, Wo, 4C)
    if (i >= total) return;
    const int c4 = (int)(i % (4 * C));
    const long pix = i / (4 * C);
    const int w = (int)(pix % Wo), h = (int)((pix / Wo) % Ho);
    const long n = pix / ((long)Wo * Ho);
    const int c = c4 >> 2, di = (c4 >> 1) & 1, dj = c4 & 1;
    const long j = (((n * (2 * Ho) + 2 * h + di) * (2 * Wo)) + 2 * w + dj) * C + c;     // index into the (N, 2Ho, 2Wo, C) tensor
    if (inverse) out[j] = in[i];
    else out[i] = in[j];
}

// ---- nearest x2 up-sampling: forward out[n][h][w] = in[n][h/2][w/2]; backward dx[n][h][w] = sum of the 2 x 2 block of dy
template <typename T>
__global__ __launch_bounds__(kThreads) void upsample2_kernel(const T* __restrict__ in, T* __restrict__ out, long items, int H, int W, int C,
                                                             int backward) {
    // items = 8-channel groups of the SMALL map (N, H, W, C) for backward, of the LARGE map (N, 2H, 2W, C) for forward
    const long gid = (long)blockIdx.x * kThreads + threadIdx.x;
    if (gid >= items) return;
    const int G = C >> 3;
    const int gl = (int)(gid % G);
    const long pix = gid / G;
    if (!backward) {
        const int w = (int)(pix % (2 * W)), h = (int)((pix / (2 * W)) % (2 * H));
        const long n = pix / ((long)4 * W * H);
        float v[8];
        load8<T>(in + (((n * H + (h >> 1)) * W) + (w >> 1)) * C + gl * 8, v);
        store8<T>(out + gid * 8, v);
    } else {
        const int w = (int)(pix % W), h = (int)((pix / W) % H);
        const long n = pix / ((long)W * H);
        float s[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] = 0.f;
        for (int di = 0; di < 2; ++di)
            for (int dj = 0; dj < 2; ++dj) {
                float v[8];
                load8<T>(in + (((n * (2 * H) + 2 * h + di) * (2 * W)) + 2 * w + dj) * C + gl * 8, v);
#pragma unroll
                for (int e = 0; e < 8; ++e) s[e] += v[e];
            }
        store8<T>(out + gid * 8, s);
    }
}

// ---- STTF warp, gradient with respect to the feature maps: the adjoint of sttf_warp_kernel's bilinear gather (same sample
// positions, computed by the same device functions) as fp32 atomic adds.  dout (B, L, H, W, C) fp32; dx (agents, H, W, C) fp32, zeroed
__global__ __launch_bounds__(kThreads) void sttf_warp_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ tmat,
                                                                 const int* __restrict__ record_len, float* __restrict__ dx, int B, int Lc,
                                                                 int H, int W, int C, float discrete_ratio, float downsample_rate) {
    const int G = C >> 3;
    const int bl = blockIdx.y;
    const int b = bl / Lc, l = bl - b * Lc;
    __shared__ Affine th_feat;
    __shared__ int src_agent;
    if (threadIdx.x == 0) {
        th_feat = sttf_theta(tmat + (size_t)bl * 16, discrete_ratio, downsample_rate, /*Hd=*/W, /*Wd=*/H);
        int src = bl;
        if (record_len) {
            int off = 0;
            for (int bb = 0; bb < b; ++bb) off += record_len[bb];
            src = l < record_len[b] ? off + l : -1;
        }
        src_agent = src;
    }
    __syncthreads();
    const int srcb = src_agent;
    if (srcb < 0) return;
    const int item = blockIdx.x * kThreads + threadIdx.x;
    const int gid = item / G;
    const int gl = item % G;
    if (gid >= H * W) return;
    const int h = gid / W, w = gid - h * W;
    float ix, iy;
    affine_sample_xy(th_feat, w, H - 1 - h, W, H, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float g[8];
    load8<float>(dout + ((size_t)bl * H * W + gid) * C + gl * 8, g);
    float* dst = dx + (size_t)srcb * H * W * C + gl * 8;
    auto tap = [&](int xx, int yy, float wgt) {
        if (xx < 0 || xx >= H || yy < 0 || yy >= W) return;
        float* p = dst + ((size_t)(H - 1 - xx) * W + yy) * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(p + e, g[e] * wgt);
    };
    tap(x0, y0, wx0 * wy0);
    tap(x1, y0, wx1 * wy0);
    tap(x0, y1, wx0 * wy1);
    tap(x1, y1, wx1 * wy1);
}

__global__ void f64_to_f32_kernel(const double* in, float* out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = (float)in[i];
}

inline bool groups_ok(int C) { return C >= 8 && (C % 8) == 0 && (C >> 3) <= kThreads; }

// any channel count (bias gradients of 2- / 3-class heads, odd test shapes): one lane per channel, rows strided over blockIdx.y
template <typename T>
__global__ __launch_bounds__(kThreads) void channel_sums_generic_kernel(const T* __restrict__ x, double* __restrict__ sum,
                                                                        double* __restrict__ sumsq, long rows, int C, int rows_per_block) {
    const int c = blockIdx.x * kThreads + threadIdx.x;
    if (c >= C) return;
    const long row_lo = (long)blockIdx.y * rows_per_block;
    const long row_hi = row_lo + rows_per_block < rows ? row_lo + rows_per_block : rows;
    double s = 0.0, q = 0.0;
    for (long r = row_lo; r < row_hi; ++r) {
        const double v = (double)load_elem<T>(x, (size_t)(r * C + c));
        s += v;
        q += v * v;
    }
    atomicAdd(sum + c, s);
    if (sumsq) atomicAdd(sumsq + c, q);
}

inline int row_blocks(long rows, int C, int* rows_per_block, int cap = 1024) {
    const int rstep = kThreads / (C >> 3);
    long want = (rows + (long)rstep * 8 - 1) / ((long)rstep * 8);            // >= 8 rows per thread
    if (want > cap) want = cap;
    if (want < 1) want = 1;
    *rows_per_block = (int)((rows + want - 1) / want);
    return (int)((rows + *rows_per_block - 1) / *rows_per_block);
}

// workgroups of a walking kernel: about `per_thread` pieces per thread per round, at most 8 workgroups of 256 per CU
inline unsigned walk_blocks(long items, int per_thread) {
    long b = (items + (long)kThreads * per_thread - 1) / ((long)kThreads * per_thread);
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// sum[c] = sum over the rows of (x - shift[c]) and, when sumsq != null, sumsq[c] = sum of its squares, as fp64 (out_f nullable:
// the same 2 C values as fp32, sums then sums of squares).  scratch: fp64 [scratch_blocks][2][C] for the per-workgroup partials.
extern "C" int cobevt_channel_sums(const void* x, const float* shift, double* sum, double* sumsq, float* out_f, double* scratch,
                                   int scratch_blocks, int dtype, long rows, int C, hipStream_t stream) {
    if (!x || !sum) return COBEVT_ERR_ARG;
    if (C < 1 || rows < 1 || (dtype != 0 && dtype != 1)) return COBEVT_ERR_SHAPE;
    if (!groups_ok(C)) {
        if (shift || out_f) return COBEVT_ERR_UNSUPPORTED;
        if (hipMemsetAsync(sum, 0, sizeof(double) * C, stream) != hipSuccess) return COBEVT_ERR_LAUNCH;
        if (sumsq && hipMemsetAsync(sumsq, 0, sizeof(double) * C, stream) != hipSuccess) return COBEVT_ERR_LAUNCH;
        long nb = (rows + 255) / 256;
        if (nb > 512) nb = 512;
        const int rpb = (int)((rows + nb - 1) / nb);
        const dim3 grid((unsigned)((C + kThreads - 1) / kThreads), (unsigned)((rows + rpb - 1) / rpb));
        if (dtype == 0) hipLaunchKernelGGL(channel_sums_generic_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, sum, sumsq, rows, C, rpb);
        else hipLaunchKernelGGL(channel_sums_generic_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, sum, sumsq, rows, C, rpb);
        return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
    }
    if (!scratch || scratch_blocks < 1) return COBEVT_ERR_ARG;
    if (sumsq && sumsq != sum + C) return COBEVT_ERR_ARG;                  // the pair is reduced as one 2 C vector
    int rpb;
    const int blocks = row_blocks(rows, C, &rpb, scratch_blocks);
    if (dtype == 0) {
        if (sumsq) hipLaunchKernelGGL((channel_sums_kernel<bf16_t, true>), dim3(blocks), dim3(kThreads), 0, stream, (const bf16_t*)x, shift, scratch, rows, C, rpb);
        else hipLaunchKernelGGL((channel_sums_kernel<bf16_t, false>), dim3(blocks), dim3(kThreads), 0, stream, (const bf16_t*)x, shift, scratch, rows, C, rpb);
    } else {
        if (sumsq) hipLaunchKernelGGL((channel_sums_kernel<float, true>), dim3(blocks), dim3(kThreads), 0, stream, (const float*)x, shift, scratch, rows, C, rpb);
        else hipLaunchKernelGGL((channel_sums_kernel<float, false>), dim3(blocks), dim3(kThreads), 0, stream, (const float*)x, shift, scratch, rows, C, rpb);
    }
    // without sumsq only the first C words of every partial pair were written: reduce them with stride 2 C
    const int n = sumsq ? 2 * C : C;
    if (sumsq) {
        hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, scratch, blocks, n, n, sum, out_f);
    } else {
        hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((n + 3) / 4), dim3(256), 0, stream, scratch, blocks, n, 2 * C, sum, out_f);
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_f64_to_f32(const double* in, float* out, int n, hipStream_t stream) {
    if (!in || !out || n < 1) return COBEVT_ERR_ARG;
    hipLaunchKernelGGL(f64_to_f32_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, in, out, n);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_bn_finalize(const double* sum, const double* sumsq, const float* gamma, const float* beta, float* running_mean,
                                  float* running_var, float* scale, float* shift, float* mean, float* rstd, int C, long rows,
                                  float eps, float momentum, int training, int shifted, long long* batches_tracked,
                                  hipStream_t stream) {
    if (!scale || !shift || !mean || !rstd || C < 1 || rows < 1) return COBEVT_ERR_ARG;
    if (training ? (!sum || !sumsq) : (!running_mean || !running_var)) return COBEVT_ERR_ARG;
    if (shifted && !running_mean) return COBEVT_ERR_ARG;
    hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, stream, sum, sumsq, gamma, beta, running_mean, running_var,
                       scale, shift, mean, rstd, C, (double)rows, eps, momentum, training, shifted, batches_tracked);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// cobevt_channel_sums (with the square sums, shifted by running_mean when it is given) + cobevt_bn_finalize (training = 1) in two launches
// instead of three; running_mean / running_var nullable together (no tracking: plain batch statistics); 8 | C
extern "C" int cobevt_bn_batch_stats(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                                     float* shift, float* mean, float* rstd, double* scratch, int scratch_blocks, int dtype, long rows, int C,
                                     float eps, float momentum, long long* batches_tracked, hipStream_t stream) {
    if (!x || !scale || !shift || !mean || !rstd || !scratch || scratch_blocks < 1) return COBEVT_ERR_ARG;
    if ((running_mean == nullptr) != (running_var == nullptr)) return COBEVT_ERR_ARG;
    if (!groups_ok(C) || rows < 1 || (dtype != 0 && dtype != 1)) return COBEVT_ERR_SHAPE;
    int rpb;
    const int blocks = row_blocks(rows, C, &rpb, scratch_blocks);
    if (dtype == 0) hipLaunchKernelGGL((channel_sums_kernel<bf16_t, true>), dim3(blocks), dim3(kThreads), 0, stream, (const bf16_t*)x, running_mean, scratch, rows, C, rpb);
    else hipLaunchKernelGGL((channel_sums_kernel<float, true>), dim3(blocks), dim3(kThreads), 0, stream, (const float*)x, running_mean, scratch, rows, C, rpb);
    hipLaunchKernelGGL(bn_reduce_finalize_kernel, dim3((C + 3) / 4), dim3(256), 0, stream, scratch, blocks, gamma, beta, running_mean, running_var, scale,
                       shift, mean, rstd, C, (double)rows, eps, momentum, running_mean != nullptr, batches_tracked);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_bn_apply(const void* x, const void* residual, const float* scale, const float* shift, void* y, int dtype,
                               long rows, int C, int act, hipStream_t stream) {
    if (!x || !scale || !shift || !y) return COBEVT_ERR_ARG;
    if (C < 8 || C % 8 || rows < 1 || act < 0 || act > 1) return COBEVT_ERR_SHAPE;
    const long items = rows * (C >> 3);
    if (kThreads % (C >> 3) == 0 && items >= 4L * kThreads) {          // a thread keeps one channel group: coefficients in registers
        const dim3 wgrid(walk_blocks(items, 4));
        if (dtype == 0) hipLaunchKernelGGL(bn_apply_walk_kernel<bf16_t>, wgrid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)residual, scale, shift, (bf16_t*)y, items, C, act);
        else if (dtype == 1) hipLaunchKernelGGL(bn_apply_walk_kernel<float>, wgrid, dim3(kThreads), 0, stream, (const float*)x, (const float*)residual, scale, shift, (float*)y, items, C, act);
        else return COBEVT_ERR_ARG;
        return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
    }
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(bn_apply_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)residual, scale, shift, (bf16_t*)y, items, C, act);
    else if (dtype == 1) hipLaunchKernelGGL(bn_apply_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, (const float*)residual, scale, shift, (float*)y, items, C, act);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dgamma / dbeta: fp64 [2][C] back to back (dgamma, then dbeta), written; grads_f (nullable): the same 2 C values as fp32; scratch: fp64
// [scratch_blocks][2][C]; dres nullable
extern "C" int cobevt_bn_backward(const void* x, const void* y, const void* dy, const float* mean, const float* rstd, const float* gamma,
                                  double* dgamma_dbeta, float* grads_f, double* scratch, int scratch_blocks, void* dx, void* dres,
                                  int dtype, long rows, int C, int act, int training, hipStream_t stream) {
    if (!x || !dy || !mean || !rstd || !dgamma_dbeta || !scratch || !dx || (act == 1 && !y)) return COBEVT_ERR_ARG;
    if (!groups_ok(C) || rows < 1 || act < 0 || act > 1 || scratch_blocks < 1) return COBEVT_ERR_SHAPE;
    int rpb;
    const int blocks = row_blocks(rows, C, &rpb, scratch_blocks);
    const long items = rows * (C >> 3);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    const float inv_rows = 1.0f / (float)rows;
    double* dgamma = dgamma_dbeta;
    double* dbeta = dgamma_dbeta + C;
    const bool walk = kThreads % (C >> 3) == 0 && items >= 4L * kThreads;
    const dim3 wgrid(walk_blocks(items, 2));
    if (walk && (dtype == 0 || dtype == 1)) {
        if (dtype == 0) {
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, scratch, rows, C, act, rpb);
            hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, scratch, blocks, 2 * C, 2 * C, dgamma_dbeta, grads_f);
            hipLaunchKernelGGL(bn_bwd_apply_walk_kernel<bf16_t>, wgrid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, gamma, dgamma, dbeta, (bf16_t*)dx, (bf16_t*)dres, items, C, inv_rows, act, training);
        } else {
            hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, scratch, rows, C, act, rpb);
            hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, scratch, blocks, 2 * C, 2 * C, dgamma_dbeta, grads_f);
            hipLaunchKernelGGL(bn_bwd_apply_walk_kernel<float>, wgrid, dim3(kThreads), 0, stream, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, gamma, dgamma, dbeta, (float*)dx, (float*)dres, items, C, inv_rows, act, training);
        }
        return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
    }
    if (dtype == 0) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<bf16_t>, dim3(blocks), dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, scratch, rows, C, act, rpb);
        hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, scratch, blocks, 2 * C, 2 * C, dgamma_dbeta, grads_f);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)y, (const bf16_t*)dy, mean, rstd, gamma, dgamma, dbeta, (bf16_t*)dx, (bf16_t*)dres, items, C, inv_rows, act, training);
    } else if (dtype == 1) {
        hipLaunchKernelGGL(bn_bwd_reduce_kernel<float>, dim3(blocks), dim3(kThreads), 0, stream, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, scratch, rows, C, act, rpb);
        hipLaunchKernelGGL(reduce_partials_strided_kernel, dim3((2 * C + 3) / 4), dim3(256), 0, stream, scratch, blocks, 2 * C, 2 * C, dgamma_dbeta, grads_f);
        hipLaunchKernelGGL(bn_bwd_apply_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, (const float*)y, (const float*)dy, mean, rstd, gamma, dgamma, dbeta, (float*)dx, (float*)dres, items, C, inv_rows, act, training);
    } else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dx fp32 (N, H, W, C), every element written
extern "C" int cobevt_maxpool3x3s2_bwd(const void* x, const void* dy, float* dx, int dtype, int N, int H, int W, int C, hipStream_t stream) {
    if (!x || !dy || !dx) return COBEVT_ERR_ARG;
    if (!groups_ok(C) || N < 1 || H < 1 || W < 1) return COBEVT_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long items = (long)N * H * W * (C >> 3);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(maxpool_bwd_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, dx, N, H, W, C, Ho, Wo);
    else if (dtype == 1) hipLaunchKernelGGL(maxpool_bwd_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, (const float*)dy, dx, N, H, W, C, Ho, Wo);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dx (N, H, W, C) in the map's type, every element written
extern "C" int cobevt_maxpool3x3s2_bwd_t(const void* x, const void* dy, void* dx, int dtype, int N, int H, int W, int C, hipStream_t stream) {
    if (!x || !dy || !dx) return COBEVT_ERR_ARG;
    if (!groups_ok(C) || N < 1 || H < 1 || W < 1) return COBEVT_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long items = (long)N * ((H + 1) / 2) * ((W + 1) / 2) * (C >> 3);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(maxpool_bwd_block_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, N, H, W, C, Ho, Wo);
    else if (dtype == 1) hipLaunchKernelGGL(maxpool_bwd_block_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)x, (const float*)dy, (float*)dx, N, H, W, C, Ho, Wo);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_group_mean(const void* in, void* out, int dtype, long B, int n, long inner, int backward, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (B < 1 || n < 1 || inner < 8 || inner % 8 || (backward != 0 && backward != 1)) return COBEVT_ERR_SHAPE;
    const long pieces = B * (inner / 8);
    const dim3 grid((unsigned)((pieces + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(group_mean_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)in, (bf16_t*)out, pieces, inner / 8, n, backward);
    else if (dtype == 1) hipLaunchKernelGGL(group_mean_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)in, (float*)out, pieces, inner / 8, n, backward);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// inverse = 0: in (N, 2Ho, 2Wo, C) -> out (N, Ho, Wo, 4C) [nn.PixelUnshuffle(2)] ; inverse = 1: in (N, Ho, Wo, 4C) -> out (N, 2Ho, 2Wo, C)
extern "C" int cobevt_pixel_unshuffle2_nhwc(const void* in, void* out, int dtype, int N, int Ho, int Wo, int C, int inverse, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (N < 1 || Ho < 1 || Wo < 1 || C < 1) return COBEVT_ERR_SHAPE;
    const long total = (long)N * Ho * Wo * 4 * C;
    const dim3 grid((unsigned)((total + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(pixel_unshuffle_kernel<uint16_t>, grid, dim3(kThreads), 0, stream, (const uint16_t*)in, (uint16_t*)out, total, Ho, Wo, C, inverse);
    else if (dtype == 1) hipLaunchKernelGGL(pixel_unshuffle_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)in, (float*)out, total, Ho, Wo, C, inverse);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// backward = 0: in (N, H, W, C) -> out (N, 2H, 2W, C) nearest ; backward = 1: in = dy (N, 2H, 2W, C) -> out = dx (N, H, W, C)
extern "C" int cobevt_upsample_nearest2_nhwc(const void* in, void* out, int dtype, int N, int H, int W, int C, int backward, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (C < 8 || C % 8 || N < 1 || H < 1 || W < 1) return COBEVT_ERR_SHAPE;
    const long items = (long)N * H * W * (C >> 3) * (backward ? 1 : 4);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads));
    if (dtype == 0) hipLaunchKernelGGL(upsample2_kernel<bf16_t>, grid, dim3(kThreads), 0, stream, (const bf16_t*)in, (bf16_t*)out, items, H, W, C, backward);
    else if (dtype == 1) hipLaunchKernelGGL(upsample2_kernel<float>, grid, dim3(kThreads), 0, stream, (const float*)in, (float*)out, items, H, W, C, backward);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// dout (B, L, H, W, C) fp32 -> dx (agents, H, W, C) fp32 (zeroed by the caller); record_len as for cobevt_sttf_warp (nullable: agent = b * L + l)
extern "C" int cobevt_sttf_warp_bwd(const float* dout, const float* tmat, const int* record_len, float* dx, int B, int L, int H, int W, int C,
                                    float discrete_ratio, float downsample_rate, hipStream_t stream) {
    if (!dout || !tmat || !dx) return COBEVT_ERR_ARG;
    if (!groups_ok(C) || B < 1 || L < 1 || H < 1 || W < 1 || B * L > 65535) return COBEVT_ERR_SHAPE;
    const long items = (long)H * W * (C >> 3);
    const dim3 grid((unsigned)((items + kThreads - 1) / kThreads), (unsigned)(B * L));
    hipLaunchKernelGGL(sttf_warp_bwd_kernel, grid, dim3(kThreads), 0, stream, dout, tmat, record_len, dx, B, L, H, W, C, discrete_ratio, downsample_rate);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
