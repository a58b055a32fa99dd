// The data formats downstream of the hot path: what the reference does to the model's logits
// before they are scored (gfx950).  Both kernels are HBM-bound streaming passes over planar (N, C, H*W) logits / (N, H*W)
// label maps: a lane owns consecutive pixels, class planes are read coalesced.
//   softmax_argmax_kernel : CameraBevPostprocessor.softmax_argmax (camera_bev_postprocessor.py:55-59): nn.Softmax(dim=1)
//                           then torch.argmax(dim=1) of the PROBABILITIES (first maximum wins on ties).
//   seg_counts_kernel     : the per-class pixel counts mean_IU / mean_precision reduce their masks to
//                           (seg_utils.py:25-50: n_ii = |pred==c & gt==c|, t_i = |gt==c|, n_ij = |pred==c|) - integer exact.
#include "common.hpp"

namespace cobevt {

constexpr int kPostMaxClasses = 8;

template <typename T> __device__ __forceinline__ float post_load(const T* p, size_t i);
template <> __device__ __forceinline__ float post_load<float>(const float* p, size_t i) { return p[i]; }
template <> __device__ __forceinline__ float post_load<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i].bits); }

template <typename T, int C>
__global__ __launch_bounds__(256) void softmax_argmax_kernel(const T* logits, float* prob, long long* map, int hw) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    const size_t base = (size_t)n * C * hw + pix;
    float x[C];
#pragma unroll
    for (int c = 0; c < C; ++c) x[c] = post_load<T>(logits, base + (size_t)c * hw);
    float m = x[0];
#pragma unroll
    for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
    float e[C], s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        e[c] = expf(x[c] - m);
        s += e[c];
    }
    int best = 0;
    float pb = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const float pc = e[c] / s;
        prob[base + (size_t)c * hw] = pc;
        if (c == 0 || pc > pb) {            // strict: the first maximum wins, as torch.argmax
            pb = pc;
            best = c;
        }
    }
    map[(size_t)n * hw + pix] = best;
}

// counts[n][c][3] += (n_ii, t_i, n_ij); pixels whose label falls outside [0, K) are counted in counts[n][K][0] (pred) and
// counts[n][K][1] (gt) so the host can refuse them.
__global__ __launch_bounds__(256) void seg_counts_kernel(const long long* pred, const long long* gt,
                                                         unsigned long long* counts, int hw, int K, int per_thread) {
    __shared__ unsigned int sh[(kPostMaxClasses + 1) * 3];
    const int n = blockIdx.y, tid = threadIdx.x;
    if (tid < (kPostMaxClasses + 1) * 3) sh[tid] = 0;
    __syncthreads();
    unsigned int loc[(kPostMaxClasses + 1) * 3];
#pragma unroll
    for (int i = 0; i < (kPostMaxClasses + 1) * 3; ++i) loc[i] = 0;
    const size_t base = (size_t)n * hw;
    const int p0 = blockIdx.x * 256 * per_thread;
    for (int j = 0; j < per_thread; ++j) {
        const int pix = p0 + j * 256 + tid;
        if (pix >= hw) break;
        const long long a = pred[base + pix], b = gt[base + pix];
        const bool a_ok = a >= 0 && a < K, b_ok = b >= 0 && b < K;
#pragma unroll
        for (int c = 0; c < kPostMaxClasses; ++c) {       // static indexing keeps `loc` in registers
            loc[c * 3 + 0] += (a_ok && b_ok && a == c && b == c) ? 1u : 0u;
            loc[c * 3 + 1] += (b_ok && b == c) ? 1u : 0u;
            loc[c * 3 + 2] += (a_ok && a == c) ? 1u : 0u;
        }
        loc[kPostMaxClasses * 3 + 0] += a_ok ? 0u : 1u;
        loc[kPostMaxClasses * 3 + 1] += b_ok ? 0u : 1u;
    }
#pragma unroll
    for (int i = 0; i < (kPostMaxClasses + 1) * 3; ++i) {
        unsigned int v = loc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0 && v) atomicAdd(&sh[i], v);
    }
    __syncthreads();
    if (tid < (kPostMaxClasses + 1) * 3 && sh[tid]) {
        const int c = tid / 3, f = tid - c * 3;
        const int dst = c == kPostMaxClasses ? K : c;        // out-of-range bucket is row K of the (K + 1)-row table
        if (c == kPostMaxClasses || c < K) atomicAdd(&counts[((size_t)n * (K + 1) + dst) * 3 + f], (unsigned long long)sh[tid]);
    }
}

template <typename T>
static int launch_softmax_argmax(const void* logits, float* prob, long long* map, int N, int C, int hw, hipStream_t stream) {
    const dim3 grid((hw + 255) / 256, N), block(256);
    switch (C) {
#define COBEVT_SA_CASE(c) case c: hipLaunchKernelGGL((softmax_argmax_kernel<T, c>), grid, block, 0, stream, (const T*)logits, prob, map, hw); break;
        COBEVT_SA_CASE(1) COBEVT_SA_CASE(2) COBEVT_SA_CASE(3) COBEVT_SA_CASE(4) COBEVT_SA_CASE(5) COBEVT_SA_CASE(6) COBEVT_SA_CASE(7) COBEVT_SA_CASE(8)
#undef COBEVT_SA_CASE
        default: return COBEVT_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry points, see include/cobevt_hip.h
extern "C" int cobevt_softmax_argmax(const void* logits, float* prob, long long* map, int dtype, int N, int C, int hw,
                                     hipStream_t stream) {
    if (!logits || !prob || !map) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || hw < 1 || C < 1 || C > kPostMaxClasses) return COBEVT_ERR_SHAPE;
    if (dtype == 0) return launch_softmax_argmax<bf16_t>(logits, prob, map, N, C, hw, stream);
    if (dtype == 1) return launch_softmax_argmax<float>(logits, prob, map, N, C, hw, stream);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_seg_class_counts(const long long* pred, const long long* gt, unsigned long long* counts, int N, int hw,
                                       int K, hipStream_t stream) {
    if (!pred || !gt || !counts) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || hw < 1 || K < 1 || K > kPostMaxClasses) return COBEVT_ERR_SHAPE;
    if (hipMemsetAsync(counts, 0, sizeof(unsigned long long) * (size_t)N * (K + 1) * 3, stream) != hipSuccess) return COBEVT_ERR_LAUNCH;
    const int per_thread = 16;
    const dim3 grid((hw + 256 * per_thread - 1) / (256 * per_thread), N), block(256);
    hipLaunchKernelGGL(seg_counts_kernel, grid, block, 0, stream, pred, gt, counts, hw, K, per_thread);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
