// Glue kernels of the pairwise-warp fusion baselines (V2VNet, DiscoNet) on channels-last maps (gfx950).  HBM / latency-bound
// element-wise work on 32 x 32 x 128 maps; the convolutions between them are the package's conv / GEMM kernels.
//
// The reference (fusion_modules/v2v_fuse.py:72-144, disconet_fuse.py:106-168) loops over samples b, target agents i and source
// agents j in Python, works on a TRANSPOSED + FLIPPED copy of every map ('b c h w -> b c w h', flip) and multiplies by a ROI mask
// that is NOT transposed / flipped.  Here every (b, i, j) is one workgroup row of one launch and all maps stay in their original
// orientation: the warp kernel undoes the transposition / flip like the STTF kernel does, the mask is read at the transposed /
// flipped position, and the 3x3 convolutions that the reference applies in the flipped domain run with their taps re-indexed on
// the host (conv(flipT(x), W) = flipT(conv(x, W~)), W~[u][v] = W[v][2 - u]; cobevt_amd/host/v2v_fuse.py).
#include "warp_common.hpp"

namespace cobevt {
namespace {

// first agent row of sample b and its agent count
__device__ __forceinline__ void sample_rows(const int* record_len, int b, int& off, int& n) {
    off = 0;
    for (int bb = 0; bb < b; ++bb) off += record_len[bb];
    n = record_len[b];
}

// nb[b][i][j] = agent j's map warped into agent i's frame (j, i < record_len[b]; zeros otherwise);
// roi[b][i][j][h][w] = the reference's mask value that multiplies out pixel (h, w)
template <typename T>
__global__ __launch_bounds__(256) void pairwise_warp_kernel(const T* x, const float* pairwise, const int* record_len, T* nb, float* roi,
                                                            int B, int L, int H, int W, int C, float discrete_ratio,
                                                            float downsample_rate) {
    const int G = C >> 3;
    const int v = blockIdx.y;                                   // (b, i, j)
    const int j = v % L, i = (v / L) % L, b = v / (L * L);
    __shared__ Affine th_feat, th_mask;
    __shared__ int src_agent;
    if (threadIdx.x == 0) {
        // v2v_fuse.py:95-103: t_matrix[:, i] = pairwise[b, j, i] for neighbour j
        const float* m = pairwise + (((size_t)b * L + j) * L + i) * 16;
        th_feat = sttf_theta(m, discrete_ratio, downsample_rate, /*Hd=*/W, /*Wd=*/H);
        th_mask = sttf_theta(m, discrete_ratio, downsample_rate, /*Hd=*/H, /*Wd=*/W, /*centred=*/false);
        int off, n;
        sample_rows(record_len, b, off, n);
        src_agent = (i < n && j < n) ? off + j : -1;
    }
    __syncthreads();
    const int srcb = src_agent;
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= H * W) return;
    const int h = gid / W, w = gid - h * W;
    // as in sttf_warp_kernel: the warped map y has dims (Hd = W, Wd = H); out[h][w] = y[w][H-1-h]; y samples x2[iy][ix] = x0[H-1-ix][iy]
    float ix, iy;
    affine_sample_xy(th_feat, w, H - 1 - h, W, H, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const T* src = x + (size_t)(srcb >= 0 ? srcb : 0) * H * W * C + gl * 8;
    auto tap = [&](int xx, int yy, float wgt) {
        if (srcb < 0 || xx < 0 || xx >= H || yy < 0 || yy >= W) return;
        float t[8];
        load8<T>(src + ((size_t)(H - 1 - xx) * W + yy) * C, t);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += t[e] * wgt;
    };
    tap(x0, y0, wx0 * wy0);
    tap(x1, y0, wx1 * wy0);
    tap(x0, y1, wx0 * wy1);
    tap(x1, y1, wx1 * wy1);
    store8<T>(nb + ((size_t)v * H * W + gid) * C + gl * 8, acc);
    if (gl == 0) {
        // the reference multiplies the flipped-domain pixel (a, c) = (w, H-1-h) by mask[a][c] (square maps)
        float mx, my;
        affine_sample_xy(th_mask, w, H - 1 - h, H, W, mx, my);
        const float rx = nearbyintf(mx), ry = nearbyintf(my);
        const bool inb = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
        roi[(size_t)v * H * W + gid] = (inb && srcb >= 0) ? 1.f : 0.f;
    }
}

// Adjoint of pairwise_warp_kernel w.r.t. x (training: torch autograd through F.affine_grid + F.grid_sample in the reference's
// warp_affine, v2v_fuse.py:95-103): dx[agent j] += the bilinear weights x dnb[b][i][j]; dx fp32 (N, H, W, C), zero-initialised.
template <typename T>
__global__ __launch_bounds__(256) void pairwise_warp_bwd_kernel(const T* dnb, const float* pairwise, const int* record_len, float* dx,
                                                                int B, int L, int H, int W, int C, float discrete_ratio,
                                                                float downsample_rate) {
    const int G = C >> 3;
    const int v = blockIdx.y;
    const int j = v % L, i = (v / L) % L, b = v / (L * L);
    __shared__ Affine th_feat;
    __shared__ int src_agent;
    if (threadIdx.x == 0) {
        const float* m = pairwise + (((size_t)b * L + j) * L + i) * 16;
        th_feat = sttf_theta(m, discrete_ratio, downsample_rate, /*Hd=*/W, /*Wd=*/H);
        int off, n;
        sample_rows(record_len, b, off, n);
        src_agent = (i < n && j < n) ? off + j : -1;
    }
    __syncthreads();
    const int srcb = src_agent;
    if (srcb < 0) return;
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= H * W) return;
    const int h = gid / W, w = gid - h * W;
    float ix, iy;
    affine_sample_xy(th_feat, w, H - 1 - h, W, H, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float g[8];
    load8<T>(dnb + ((size_t)v * H * W + gid) * C + gl * 8, g);
    float* dst = dx + (size_t)srcb * H * W * C + gl * 8;
    auto tap = [&](int xx, int yy, float wgt) {
        if (xx < 0 || xx >= H || yy < 0 || yy >= W || wgt == 0.f) return;
        float* q = dst + ((size_t)(H - 1 - xx) * W + yy) * C;
#pragma unroll
        for (int e = 0; e < 8; ++e) atomicAdd(q + e, g[e] * wgt);
    };
    tap(x0, y0, wx0 * wy0);
    tap(x1, y0, wx1 * wy0);
    tap(x0, y1, wx0 * wy1);
    tap(x1, y1, wx1 * wy1);
}

// V2VNet message aggregation (v2v_fuse.py:108-119): out[(b, i)] = reduce over j < N_b of (msg[b][i][j] + ego[(b, i)]) * roi[b][i][j]
// with reduce = mean (mode 0) or max (mode 1); written at agent row off_b + i of the un-grouped layout.
template <typename T>
__global__ __launch_bounds__(256) void agent_message_reduce_kernel(const T* msg, const T* ego, const float* roi, const int* record_len,
                                                                   T* out, int B, int L, int HW, int C, int mode) {
    const int G = C >> 3;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const int gl = (int)(item % G);
    const long pix = item / G;
    if (pix >= (long)B * L * HW) return;
    const int p = (int)(pix % HW), i = (int)((pix / HW) % L), b = (int)(pix / ((long)HW * L));
    int off, n;
    sample_rows(record_len, b, off, n);
    if (i >= n) return;
    float e[8], acc[8];
    load8<T>(ego + ((size_t)(off + i) * HW + p) * C + gl * 8, e);
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = mode ? -INFINITY : 0.f;
    for (int j = 0; j < n; ++j) {
        const size_t v = ((size_t)b * L + i) * L + j;
        const float m = roi[v * HW + p];
        float t[8];
        load8<T>(msg + (v * HW + p) * C + gl * 8, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const float val = (t[k] + e[k]) * m;
            acc[k] = mode ? fmaxf(acc[k], val) : acc[k] + val;
        }
    }
    if (!mode) {
        const float inv = 1.f / (float)n;
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] *= inv;
    }
    store8<T>(out + ((size_t)(off + i) * HW + p) * C + gl * 8, acc);
}

// ConvGRU cell with a zero hidden state (convgru.py:57-78 as v2v_fuse.py:125-130 calls it): h = sigmoid(update) * tanh(candidate);
// in[row] = [update pre-activations (C) | candidate pre-activations (C)]
template <typename T>
__global__ __launch_bounds__(256) void gru_zero_state_kernel(const T* in, T* out, long rows, int C) {
    const int G = C >> 3;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const int gl = (int)(item % G);
    const long row = item / G;
    if (row >= rows) return;
    float u[8], c[8], o[8];
    load8<T>(in + (size_t)row * 2 * C + gl * 8, u);
    load8<T>(in + (size_t)row * 2 * C + C + gl * 8, c);
#pragma unroll
    for (int k = 0; k < 8; ++k) o[k] = (1.f / (1.f + __expf(-u[k]))) * tanhf(c[k]);
    store8<T>(out + (size_t)row * C + gl * 8, o);
}

// DiscoNet (disconet_fuse.py:35-42,141-150): w_j = softmax over j < N_b of (score_j, or -inf where roi_j == 0);
// out[(b, i)] = sum_j w_j * nb[b][i][j] * roi_j.  score: column 0 of a (B*L*L*HW, lds) matrix (already ReLU'ed).
template <typename T>
__global__ __launch_bounds__(256) void agent_softmax_sum_kernel(const T* score, int lds, const T* nb, const float* roi, const int* record_len,
                                                                T* out, int B, int L, int HW, int C, int use_mask) {
    const int G = C >> 3;
    const long item = (long)blockIdx.x * 256 + threadIdx.x;
    const int gl = (int)(item % G);
    const long pix = item / G;
    if (pix >= (long)B * L * HW) return;
    const int p = (int)(pix % HW), i = (int)((pix / HW) % L), b = (int)(pix / ((long)HW * L));
    int off, n;
    sample_rows(record_len, b, off, n);
    if (i >= n) return;
    const size_t v0 = ((size_t)b * L + i) * L;
    float mx = -INFINITY;
    for (int j = 0; j < n; ++j) {
        const float m = roi[(v0 + j) * HW + p];
        const float s = (!use_mask || m != 0.f) ? load_elem<T>(score, ((v0 + j) * HW + p) * (size_t)lds) : -INFINITY;
        mx = fmaxf(mx, s);
    }
    float den = 0.f, acc[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] = 0.f;
    for (int j = 0; j < n; ++j) {
        const float m = roi[(v0 + j) * HW + p];
        const float s = (!use_mask || m != 0.f) ? load_elem<T>(score, ((v0 + j) * HW + p) * (size_t)lds) : -INFINITY;
        const float e = __expf(s - mx);
        den += e;
        float t[8];
        load8<T>(nb + ((v0 + j) * HW + p) * C + gl * 8, t);
#pragma unroll
        for (int k = 0; k < 8; ++k) acc[k] += e * m * t[k];
    }
    const float inv = 1.f / den;
#pragma unroll
    for (int k = 0; k < 8; ++k) acc[k] *= inv;
    store8<T>(out + ((size_t)(off + i) * HW + p) * C + gl * 8, acc);
}

template <typename K, typename... Args>
int launch1d(K kern, long work_items, hipStream_t stream, Args... args) {
    const long blocks = (work_items + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, args...);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

bool group_ok(int C) { const int G = C >> 3; return C % 8 == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0; }

}  // namespace
}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_pairwise_warp(const void* x, const float* pairwise, const int* record_len, void* nb, float* roi, int dtype,
                                    int B, int L, int H, int W, int C, float discrete_ratio, float downsample_rate,
                                    hipStream_t stream) {
    if (!x || !pairwise || !record_len || !nb || !roi) return COBEVT_ERR_ARG;
    if (!group_ok(C) || B < 1 || L < 1 || H < 1 || W < 1 || H != W || (long)B * L * L > 65535) return COBEVT_ERR_SHAPE;
    const long items = (long)H * W * (C >> 3);
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)(B * L * L));
    if (dtype == 0) hipLaunchKernelGGL(pairwise_warp_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, pairwise, record_len, (bf16_t*)nb, roi, B, L, H, W, C, discrete_ratio, downsample_rate);
    else if (dtype == 1) hipLaunchKernelGGL(pairwise_warp_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, pairwise, record_len, (float*)nb, roi, B, L, H, W, C, discrete_ratio, downsample_rate);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_pairwise_warp_bwd(const void* dnb, const float* pairwise, const int* record_len, float* dx, int dtype, int B, int L,
                                        int H, int W, int C, float discrete_ratio, float downsample_rate, hipStream_t stream) {
    if (!dnb || !pairwise || !record_len || !dx) return COBEVT_ERR_ARG;
    if (!group_ok(C) || B < 1 || L < 1 || H < 1 || W < 1 || H != W || (long)B * L * L > 65535) return COBEVT_ERR_SHAPE;
    const long items = (long)H * W * (C >> 3);
    const dim3 grid((unsigned)((items + 255) / 256), (unsigned)(B * L * L));
    if (dtype == 0) hipLaunchKernelGGL(pairwise_warp_bwd_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)dnb, pairwise, record_len, dx, B, L, H, W, C, discrete_ratio, downsample_rate);
    else if (dtype == 1) hipLaunchKernelGGL(pairwise_warp_bwd_kernel<float>, grid, dim3(256), 0, stream, (const float*)dnb, pairwise, record_len, dx, B, L, H, W, C, discrete_ratio, downsample_rate);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_agent_message_reduce(const void* msg, const void* ego, const float* roi, const int* record_len, void* out,
                                           int dtype, int B, int L, int HW, int C, int mode, hipStream_t stream) {
    if (!msg || !ego || !roi || !record_len || !out) return COBEVT_ERR_ARG;
    if (!group_ok(C) || B < 1 || L < 1 || HW < 1 || (mode != 0 && mode != 1)) return COBEVT_ERR_SHAPE;
    const long items = (long)B * L * HW * (C >> 3);
    if (dtype == 0) return launch1d(agent_message_reduce_kernel<bf16_t>, items, stream, (const bf16_t*)msg, (const bf16_t*)ego, roi, record_len, (bf16_t*)out, B, L, HW, C, mode);
    if (dtype == 1) return launch1d(agent_message_reduce_kernel<float>, items, stream, (const float*)msg, (const float*)ego, roi, record_len, (float*)out, B, L, HW, C, mode);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_gru_zero_state(const void* in, void* out, int dtype, long rows, int C, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (!group_ok(C) || rows < 1) return COBEVT_ERR_SHAPE;
    const long items = rows * (C >> 3);
    if (dtype == 0) return launch1d(gru_zero_state_kernel<bf16_t>, items, stream, (const bf16_t*)in, (bf16_t*)out, rows, C);
    if (dtype == 1) return launch1d(gru_zero_state_kernel<float>, items, stream, (const float*)in, (float*)out, rows, C);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_agent_softmax_sum(const void* score, int lds, const void* nb, const float* roi, const int* record_len, void* out,
                                        int dtype, int B, int L, int HW, int C, int use_mask, hipStream_t stream) {
    if (!score || !nb || !roi || !record_len || !out) return COBEVT_ERR_ARG;
    if (!group_ok(C) || B < 1 || L < 1 || HW < 1 || lds < 1) return COBEVT_ERR_SHAPE;
    const long items = (long)B * L * HW * (C >> 3);
    if (dtype == 0) return launch1d(agent_softmax_sum_kernel<bf16_t>, items, stream, (const bf16_t*)score, lds, (const bf16_t*)nb, roi, record_len, (bf16_t*)out, B, L, HW, C, use_mask);
    if (dtype == 1) return launch1d(agent_softmax_sum_kernel<float>, items, stream, (const float*)score, lds, (const float*)nb, roi, record_len, (float*)out, B, L, HW, C, use_mask);
    return COBEVT_ERR_ARG;
}
