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

// ---------------------------------------------------------------------------------------------------------------------
// Class-weighted cross entropy of planar logits against an integer label map, the reduction nn.CrossEntropyLoss(weight=w)
// applies in VanillaSegLoss (opv2v/opencood/loss/vanilla_seg_loss.py:18-23,58-70):
//     loss = sum_i w[y_i] * (logsumexp_c x_i[c] - x_i[y_i]) / sum_i w[y_i]
// Two launches with a fixed summation order (bit-reproducible): per-workgroup partial (numerator, denominator) pairs, then
// one workgroup adds them in order and divides.
namespace cobevt {

template <typename T, int C>
__global__ __launch_bounds__(256) void wce_partial_kernel(const T* logits, const long long* target, const float* weight,
                                                          float* partial, int hw, int per_thread) {
    __shared__ float sn[256], sd[256], sb[256];
    const int n = blockIdx.y, tid = threadIdx.x;
    const size_t base = (size_t)n * C * hw;
    float num = 0.f, den = 0.f, bad = 0.f;
    const int p0 = blockIdx.x * 256 * per_thread;
    for (int j = 0; j < per_thread; ++j) {
        const int pix = p0 + j * 256 + tid;
        if (pix >= hw) break;
        float x[C];
#pragma unroll
        for (int c = 0; c < C; ++c) x[c] = post_load<T>(logits, base + (size_t)c * hw + pix);
        float m = x[0];
#pragma unroll
        for (int c = 1; c < C; ++c) m = fmaxf(m, x[c]);
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c) s += expf(x[c] - m);
        const long long y = target[(size_t)n * hw + pix];
        float xy = 0.f, wy = 0.f;
#pragma unroll
        for (int c = 0; c < C; ++c)
            if (y == c) { xy = x[c]; wy = weight[c]; }
        num += wy * (m + logf(s) - xy);
        den += wy;
        // -100 = nn.CrossEntropyLoss's ignore_index: out of numerator and denominator.  Any other label outside [0, C) is an
        // error in the reference (it raises); counted here and refused by the host (ops.weighted_cross_entropy)
        if ((y < 0 || y >= C) && y != -100) bad += 1.f;
    }
    sn[tid] = num; sd[tid] = den; sb[tid] = bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sn[tid] += sn[tid + s]; sd[tid] += sd[tid + s]; sb[tid] += sb[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[3 * b] = sn[0];
        partial[3 * b + 1] = sd[0];
        partial[3 * b + 2] = sb[0];
    }
}

// d loss / d logits of the class-weighted mean cross entropy: w[y] (softmax_c - [c == y]) / sum w[y], times the upstream
// gradient (a device scalar); pixels with the ignore label -100 get zeros.  stats = the forward's out[] (stats[2] = sum w[y]).
template <int C>
__global__ __launch_bounds__(256) void wce_backward_kernel(const float* logits, const long long* target, const float* weight,
                                                           const float* stats, const float* upstream, float* dlogits, int hw) {
    const int n = blockIdx.y;
    const int pix = blockIdx.x * 256 + threadIdx.x;
    if (pix >= hw) return;
    float x[C];
    float m = -INFINITY;
#pragma unroll
    for (int c = 0; c < C; ++c) { x[c] = logits[((size_t)n * C + c) * hw + pix]; m = fmaxf(m, x[c]); }
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c) { x[c] = expf(x[c] - m); s += x[c]; }
    const long long y = target[(size_t)n * hw + pix];
    float wy = 0.f;
#pragma unroll
    for (int c = 0; c < C; ++c)
        if (y == c) wy = weight[c];
    const float g = upstream[0] * wy / stats[2] / s;
#pragma unroll
    for (int c = 0; c < C; ++c) dlogits[((size_t)n * C + c) * hw + pix] = g * x[c] - (y == c ? upstream[0] * wy / stats[2] : 0.f);
}

// `stride` floats per partial record: 3 = {numerator, denominator, bad-label count} (cross entropy), 2 = {numerator, denominator}
__global__ __launch_bounds__(256) void wce_final_kernel(const float* partial, float* out, int nparts, int stride) {
    __shared__ float sn[256], sd[256], sb[256];
    const int tid = threadIdx.x;
    float num = 0.f, den = 0.f, bad = 0.f;
    for (int i = tid; i < nparts; i += 256) {
        num += partial[stride * i];
        den += partial[stride * i + 1];
        if (stride > 2) bad += partial[stride * i + 2];
    }
    sn[tid] = num; sd[tid] = den; sb[tid] = bad;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sn[tid] += sn[tid + s]; sd[tid] += sd[tid + s]; sb[tid] += sb[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        out[0] = sn[0] / sd[0]; out[1] = sn[0]; out[2] = sd[0];
        if (stride > 2) out[3] = sb[0];
    }
}

template <typename T>
static int launch_wce(const void* logits, const long long* target, const float* weight, float* partial, int N, int C, int hw,
                      int per_thread, hipStream_t stream) {
    const dim3 grid((hw + 256 * per_thread - 1) / (256 * per_thread), N), block(256);
    switch (C) {
#define COBEVT_WCE_CASE(c) case c: hipLaunchKernelGGL((wce_partial_kernel<T, c>), grid, block, 0, stream, (const T*)logits, target, weight, partial, hw, per_thread); break;
        COBEVT_WCE_CASE(2) COBEVT_WCE_CASE(3) COBEVT_WCE_CASE(4) COBEVT_WCE_CASE(5) COBEVT_WCE_CASE(6) COBEVT_WCE_CASE(7) COBEVT_WCE_CASE(8)
#undef COBEVT_WCE_CASE
        default: return COBEVT_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt

// out[0] = loss, out[1] = weighted numerator, out[2] = sum of weights, out[3] = number of labels outside [0, C) other than the
// ignore index -100 (the caller must refuse the result when it is non-zero); scratch >= 3 * ceil(hw / 4096) * N floats
extern "C" int cobevt_weighted_cross_entropy(const void* logits, const long long* target, const float* weight, float* scratch,
                                             float* out, int dtype, int N, int C, int hw, hipStream_t stream) {
    if (!logits || !target || !weight || !scratch || !out) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || hw < 1 || C < 2 || C > kPostMaxClasses) return COBEVT_ERR_SHAPE;
    const int per_thread = 16;
    const int nparts = ((hw + 256 * per_thread - 1) / (256 * per_thread)) * N;
    int rc;
    if (dtype == 0) rc = launch_wce<bf16_t>(logits, target, weight, scratch, N, C, hw, per_thread, stream);
    else if (dtype == 1) rc = launch_wce<float>(logits, target, weight, scratch, N, C, hw, per_thread, stream);
    else return COBEVT_ERR_ARG;
    if (rc != COBEVT_OK) return rc;
    hipLaunchKernelGGL(wce_final_kernel, dim3(1), dim3(256), 0, stream, scratch, out, nparts, 3);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// Backward of cobevt_weighted_cross_entropy (fp32 logits (N, C, hw)): see wce_backward_kernel.  stats = the forward's out[4].
extern "C" int cobevt_weighted_cross_entropy_bwd(const float* logits, const long long* target, const float* weight, const float* stats,
                                                 const float* upstream, float* dlogits, int N, int C, int hw, hipStream_t stream) {
    if (!logits || !target || !weight || !stats || !upstream || !dlogits) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || hw < 1 || C < 2 || C > kPostMaxClasses) return COBEVT_ERR_SHAPE;
    const dim3 grid((hw + 255) / 256, N), block(256);
    switch (C) {
#define COBEVT_WCEB_CASE(c) case c: hipLaunchKernelGGL((wce_backward_kernel<c>), grid, block, 0, stream, logits, target, weight, stats, upstream, dlogits, hw); break;
        COBEVT_WCEB_CASE(2) COBEVT_WCEB_CASE(3) COBEVT_WCEB_CASE(4) COBEVT_WCEB_CASE(5) COBEVT_WCEB_CASE(6) COBEVT_WCEB_CASE(7) COBEVT_WCEB_CASE(8)
#undef COBEVT_WCEB_CASE
        default: return COBEVT_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// nuScenes IoU metric counts (nuscenes/cross_view_transformer/metrics.py:22-31,56-72): per threshold t,
//   tp += |sigmoid(pred) >= t & label|, fp += |sigmoid(pred) >= t & !label|, fn += |sigmoid(pred) < t & label|
// over the pixels whose visibility is at least min_visibility (all pixels when min_visibility < 0).  The label of an
// output channel is the maximum (any non-zero) of a set of ground-truth channels given as a bit mask per output channel.
// Integer counts, exact; the comparison is done on the fp32 sigmoid as in the reference.
namespace cobevt {

constexpr int kIouMaxThr = 8;

__global__ __launch_bounds__(256) void iou_counts_kernel(const float* pred, const float* label, const unsigned char* visibility,
                                                         const unsigned int* label_mask, const float* thresholds,
                                                         unsigned long long* counts, int C, int NL, int hw, int T,
                                                         int min_visibility, int per_thread) {
    __shared__ unsigned int sh[kIouMaxThr * 3];
    const int n = blockIdx.y, tid = threadIdx.x;
    if (tid < kIouMaxThr * 3) sh[tid] = 0;
    __syncthreads();
    unsigned int loc[kIouMaxThr * 3];
#pragma unroll
    for (int i = 0; i < kIouMaxThr * 3; ++i) loc[i] = 0;
    const int p0 = blockIdx.x * 256 * per_thread;
    for (int j = 0; j < per_thread; ++j) {
        const int pix = p0 + j * 256 + tid;
        if (pix >= hw) break;
        if (min_visibility >= 0 && (int)visibility[(size_t)n * hw + pix] < min_visibility) continue;
        unsigned int present = 0;                              // bit l: ground-truth channel l is non-zero here
        for (int l = 0; l < NL; ++l) present |= (label[((size_t)n * NL + l) * hw + pix] != 0.f ? 1u : 0u) << l;
        for (int c = 0; c < C; ++c) {
            const float x = pred[((size_t)n * C + c) * hw + pix];
            const float s = 1.0f / (1.0f + expf(-x));
            const bool lab = (present & label_mask[c]) != 0;
#pragma unroll
            for (int t = 0; t < kIouMaxThr; ++t) {
                if (t < T) {
                    const bool pr = s >= thresholds[t];
                    loc[t * 3 + 0] += (pr && lab) ? 1u : 0u;
                    loc[t * 3 + 1] += (pr && !lab) ? 1u : 0u;
                    loc[t * 3 + 2] += (!pr && lab) ? 1u : 0u;
                }
            }
        }
    }
#pragma unroll
    for (int i = 0; i < kIouMaxThr * 3; ++i) {
        unsigned int v = loc[i];
        for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
        if ((tid & 63) == 0 && v) atomicAdd(&sh[i], v);
    }
    __syncthreads();
    if (tid < T * 3 && sh[tid]) atomicAdd(&counts[tid], (unsigned long long)sh[tid]);
}

}  // namespace cobevt

// counts[T][3] += (tp, fp, fn) (NOT zeroed here: the metric accumulates over batches).  pred (N, C, hw) fp32 logits, label
// (N, NL, hw) fp32 (NL <= 32), visibility (N, hw) uint8 or null, label_mask [C] uint32, thresholds [T <= 8] fp32.
extern "C" int cobevt_iou_counts(const float* pred, const float* label, const unsigned char* visibility,
                                 const unsigned int* label_mask, const float* thresholds, unsigned long long* counts, int N,
                                 int C, int NL, int hw, int T, int min_visibility, hipStream_t stream) {
    if (!pred || !label || !label_mask || !thresholds || !counts) return COBEVT_ERR_ARG;
    if (min_visibility >= 0 && !visibility) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || C < 1 || NL < 1 || NL > 32 || hw < 1 || T < 1 || T > kIouMaxThr) return COBEVT_ERR_SHAPE;
    const int per_thread = 8;
    const dim3 grid((hw + 256 * per_thread - 1) / (256 * per_thread), N), block(256);
    hipLaunchKernelGGL(iou_counts_kernel, grid, block, 0, stream, pred, label, visibility, label_mask, thresholds, counts, C, NL, hw,
                       T, min_visibility, per_thread);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// Sigmoid focal loss with label grouping and visibility masking, mean over the kept elements: the forward of
// BinarySegmentationLoss / CenterLoss (nuscenes/cross_view_transformer/losses.py:27-84) around fvcore's sigmoid_focal_loss
// (third-party, absent here; published definition: ce = BCE-with-logits(x, t), p_t = p t + (1 - p)(1 - t),
// loss = ce (1 - p_t)^gamma, times alpha t + (1 - alpha)(1 - t) when alpha >= 0).  Two launches, fixed summation order.
namespace cobevt {

__global__ __launch_bounds__(256) void focal_partial_kernel(const float* pred, const float* label, const unsigned char* visibility,
                                                            const unsigned int* label_mask, float* partial, int C, int NL, int hw,
                                                            int min_visibility, float alpha, float gamma, int soft_label,
                                                            int per_thread) {
    __shared__ float sn[256], sd[256];
    const int n = blockIdx.y, tid = threadIdx.x;
    float num = 0.f, den = 0.f;
    const int p0 = blockIdx.x * 256 * per_thread;
    for (int j = 0; j < per_thread; ++j) {
        const int pix = p0 + j * 256 + tid;
        if (pix >= hw) break;
        if (min_visibility >= 0 && (int)visibility[(size_t)n * hw + pix] < min_visibility) continue;
        for (int c = 0; c < C; ++c) {
            float t;
            if (soft_label) {                         // the label channel itself (CenterLoss: a heat map in [0, 1])
                t = label[((size_t)n * NL + c) * hw + pix];
            } else {                                  // max over the grouped binary channels
                t = 0.f;
                for (int l = 0; l < NL; ++l)
                    if ((label_mask[c] >> l) & 1u) t = fmaxf(t, label[((size_t)n * NL + l) * hw + pix]);
            }
            const float x = pred[((size_t)n * C + c) * hw + pix];
            const float p = 1.0f / (1.0f + expf(-x));
            const float ce = fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));       // binary_cross_entropy_with_logits
            const float pt = p * t + (1.f - p) * (1.f - t);
            float l = ce * powf(1.f - pt, gamma);
            if (alpha >= 0.f) l *= alpha * t + (1.f - alpha) * (1.f - t);
            num += l;
            den += 1.f;
        }
    }
    sn[tid] = num; sd[tid] = den;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) { sn[tid] += sn[tid + s]; sd[tid] += sd[tid + s]; }
        __syncthreads();
    }
    if (tid == 0) {
        const size_t b = (size_t)blockIdx.y * gridDim.x + blockIdx.x;
        partial[2 * b] = sn[0];
        partial[2 * b + 1] = sd[0];
    }
}

}  // namespace cobevt

// out[0] = mean loss over the kept elements, out[1] = sum, out[2] = count.  pred (N, C, hw) fp32 logits; label (N, NL, hw) fp32;
// label_mask [C] (bit l: label channel l belongs to output channel c) unless soft_label (then NL == C and label is used as is);
// visibility (N, hw) uint8 or null with min_visibility < 0; scratch >= 2 * N * ceil(hw / 2048) floats.
extern "C" int cobevt_sigmoid_focal_loss(const float* pred, const float* label, const unsigned char* visibility,
                                         const unsigned int* label_mask, float* scratch, float* out, int N, int C, int NL, int hw,
                                         int min_visibility, float alpha, float gamma, int soft_label, hipStream_t stream) {
    if (!pred || !label || !scratch || !out) return COBEVT_ERR_ARG;
    if (!soft_label && !label_mask) return COBEVT_ERR_ARG;
    if (min_visibility >= 0 && !visibility) return COBEVT_ERR_ARG;
    if (N < 1 || N > 65535 || C < 1 || NL < 1 || NL > 32 || hw < 1 || (soft_label && NL != C)) return COBEVT_ERR_SHAPE;
    const int per_thread = 8;
    const int gx = (hw + 256 * per_thread - 1) / (256 * per_thread);
    hipLaunchKernelGGL(focal_partial_kernel, dim3(gx, N), dim3(256), 0, stream, pred, label, visibility, label_mask, scratch, C, NL, hw,
                       min_visibility, alpha, gamma, soft_label, per_thread);
    hipLaunchKernelGGL(wce_final_kernel, dim3(1), dim3(256), 0, stream, scratch, out, gx * N, 2);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

// ---------------------------------------------------------------------------------------------------------------------
// 3x3 / stride 1 / pad 1 convolution with a handful of output channels, channels-last input -> fp32 NCHW logits: BevSegHead's
// dynamic / static heads (bev_seg_head.py:20-33,44-58: Conv2d(32, 2 | 3, 3, padding=1)) on the 256 x 256 BEV map.  75 MFLOP and
// 4.7 MB: on the implicit-GEMM kernel the 2 output channels occupy one lane column of a 32-wide MFMA tile (21 us); here a thread
// owns one pixel and all its output channels, the 18 x 18 x Cin input patch of a 16 x 16 pixel tile sits in LDS (16-byte reads),
// the weights are wave-uniform scalar loads.
namespace cobevt {

// CPP: 16-byte pieces per pixel when known at compile time (4 = the shipped heads' 32 bf16 channels; 0 = run-time): with the tap / piece
// loops unrolled the 72 wave-uniform weight loads of a thread are all requested up front instead of one exposed scalar round trip per
// (tap, piece) iteration; same summation order
template <typename T, int COUT, int CPP = 0>
__global__ __launch_bounds__(256) void conv3x3_head_kernel(const T* in, const float* wgt, const float* bias, float* out, int N, int H,
                                                           int W, int Cin) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int CH = Elem<T>::kChunk;
    const int pstr = Cin * (int)sizeof(T) + 16;                 // patch pixel stride in bytes (odd multiple of 16: conflict-free)
    const int tid = threadIdx.x;
    const int tx = blockIdx.x * 16, ty = blockIdx.y * 16, n = blockIdx.z;
    const int cpp = CPP ? CPP : Cin / CH;                       // 16-byte pieces per pixel
    if constexpr (CPP != 0) {
        // every piece of the thread requested before the first is written (the rolled form below pays one exposed round trip per iteration)
        constexpr int NIT = (18 * 18 * CPP + 255) / 256;
        uint4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int item = tid + u * 256, ic = item < 18 * 18 * CPP ? item : 18 * 18 * CPP - 1;
            const int pix = ic / CPP, j = ic - pix * CPP;
            const int py = pix / 18, px = pix - py * 18;
            const int y = ty - 1 + py, x = tx - 1 + px;
            ok[u] = y >= 0 && y < H && x >= 0 && x < W;
            v[u] = *(const uint4*)(in + (((size_t)n * H + (ok[u] ? y : 0)) * W + (ok[u] ? x : 0)) * Cin + j * CH);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int item = tid + u * 256;
            if (item < 18 * 18 * CPP) *(uint4*)(smem + (item / CPP) * pstr + (item % CPP) * 16) = ok[u] ? v[u] : make_uint4(0, 0, 0, 0);
        }
    } else
    for (int item = tid; item < 18 * 18 * cpp; item += 256) {
        const int pix = item / cpp, j = item - pix * cpp;
        const int py = pix / 18, px = pix - py * 18;
        const int y = ty - 1 + py, x = tx - 1 + px;
        const bool ok = y >= 0 && y < H && x >= 0 && x < W;
        const uint4 v = *(const uint4*)(in + (((size_t)n * H + (ok ? y : 0)) * W + (ok ? x : 0)) * Cin + j * CH);
        *(uint4*)(smem + pix * pstr + j * 16) = ok ? v : make_uint4(0, 0, 0, 0);
    }
    __syncthreads();
    const int ly = tid >> 4, lx = tid & 15;
    float acc[COUT];
#pragma unroll
    for (int o = 0; o < COUT; ++o) acc[o] = bias ? bias[o] : 0.f;
#pragma unroll(CPP ? 9 : 1)
    for (int tap = 0; tap < 9; ++tap) {
        const unsigned char* pp = smem + ((ly + tap / 3) * 18 + lx + tap % 3) * pstr;
#pragma unroll(CPP ? CPP : 1)
        for (int j = 0; j < cpp; ++j) {
            float v[8];
            chunk_to_f32<T>(*(const uint4*)(pp + j * 16), v);
#pragma unroll
            for (int o = 0; o < COUT; ++o) {
                const float* wp = wgt + ((size_t)o * 9 + tap) * Cin + j * CH;      // wave-uniform: scalar loads
#pragma unroll
                for (int e = 0; e < CH; ++e) acc[o] = fmaf(v[e], wp[e], acc[o]);
            }
        }
    }
    const int y = ty + ly, x = tx + lx;
    if (y < H && x < W) {
#pragma unroll
        for (int o = 0; o < COUT; ++o) out[(((size_t)n * COUT + o) * H + y) * W + x] = acc[o];
    }
}

template <typename T>
static int launch_head(const void* in, const float* wgt, const float* bias, float* out, int N, int H, int W, int Cin, int Cout,
                       hipStream_t stream) {
    const dim3 grid((W + 15) / 16, (H + 15) / 16, N), block(256);
    const size_t lds = (size_t)18 * 18 * (Cin * sizeof(T) + 16);
    if (lds > 96 * 1024) return COBEVT_ERR_UNSUPPORTED;            // a pixel row of at most 256 bytes
    switch (Cout) {
#define COBEVT_HEAD_CASE(c) case c: \
        if (Cin * (int)sizeof(T) == 64) { hipLaunchKernelGGL((conv3x3_head_kernel<T, c, 4>), grid, block, lds, stream, (const T*)in, wgt, bias, out, N, H, W, Cin); break; } \
        (void)hipFuncSetAttribute((const void*)conv3x3_head_kernel<T, c>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024); hipLaunchKernelGGL((conv3x3_head_kernel<T, c>), grid, block, lds, stream, (const T*)in, wgt, bias, out, N, H, W, Cin); break;
        COBEVT_HEAD_CASE(1) COBEVT_HEAD_CASE(2) COBEVT_HEAD_CASE(3) COBEVT_HEAD_CASE(4)
#undef COBEVT_HEAD_CASE
        default: return COBEVT_ERR_SHAPE;
    }
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

}  // namespace cobevt

// wgt fp32 [Cout][9 taps][Cin]; out fp32 (N, Cout, H, W); Cout 1..4, Cin a multiple of the 16-byte chunk, a pixel row <= 256 bytes
extern "C" int cobevt_conv3x3_head_nchw(const void* in, const float* wgt, const float* bias, float* out, int dtype, int N, int H, int W,
                                        int Cin, int Cout, hipStream_t stream) {
    if (!in || !wgt || !out) return COBEVT_ERR_ARG;
    const int ch = dtype == 0 ? 8 : 4;
    if (N < 1 || N > 65535 || H < 1 || W < 1 || Cin < ch || Cin % ch || Cin > 128 || Cout < 1 || Cout > 4) return COBEVT_ERR_SHAPE;
    if (dtype == 0) return launch_head<bf16_t>(in, wgt, bias, out, N, H, W, Cin, Cout, stream);
    if (dtype == 1) return launch_head<float>(in, wgt, bias, out, N, H, W, Cin, Cout, stream);
    return COBEVT_ERR_ARG;
}
