// Box calibration for bench.py: what THIS box's matrix pipes, HBM and shader clock deliver, measured in
// ~100 ms inside the bench process, so that a roofline fraction can be normalised by the box it was measured on and a 5 %
// frames/s difference between two driver runs can be attributed (boxes of the pool differ by +-6..10 %).  Not on the product
// path: nothing in cobevt_amd/host calls these.
#include <hip/hip_runtime.h>

#include "common.hpp"

namespace {

using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8v = __attribute__((ext_vector_type(8))) __bf16;

// One wave per SIMD x 4 independent accumulators: the issue-bound dense rate of v_mfma_f32_32x32x16_bf16 (no operand
// traffic).  clk[0] / clk[1]: shader-clock and 100-MHz wall-clock ticks spent by workgroup 0 inside the loop.
__global__ __launch_bounds__(256) void calib_mfma_kernel(float* out, long long* clk, int iters) {
    bf16x8v a, b;
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        a[i] = (__bf16)(float)((threadIdx.x + i) & 7);
        b[i] = (__bf16)(float)(i + 1);
    }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long long w0 = wall_clock64(), s0 = clock64();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    long long s1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        clk[0] = s1 - s0;
        clk[1] = w1 - w0;
    }
}

// Streaming copy, 16 bytes per lane, grid-stride: read n16 x 16 B, write n16 x 16 B.
using u32x4 = __attribute__((ext_vector_type(4))) unsigned int;
__global__ __launch_bounds__(256) void calib_copy_kernel(const u32x4* __restrict__ src, u32x4* __restrict__ dst, long n16) {
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long step = (long)gridDim.x * 256;
    for (; i < n16; i += step) __builtin_nontemporal_store(__builtin_nontemporal_load(src + i), dst + i);
}

}  // namespace

extern "C" int cobevt_calibrate_mfma(float* out, long long* clk, int blocks, int iters, hipStream_t stream) {
    if (!out || !clk || blocks < 1 || iters < 1) return COBEVT_ERR_ARG;
    hipLaunchKernelGGL(calib_mfma_kernel, dim3(blocks), dim3(256), 0, stream, out, clk, iters);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_calibrate_clock_khz(int* wall_khz, int* sclk_max_khz) {
    int dev = 0, w = 0, c = 0;
    if (!wall_khz || !sclk_max_khz) return COBEVT_ERR_ARG;
    if (hipGetDevice(&dev) != hipSuccess) return COBEVT_ERR_LAUNCH;
    if (hipDeviceGetAttribute(&w, hipDeviceAttributeWallClockRate, dev) != hipSuccess) return COBEVT_ERR_LAUNCH;
    if (hipDeviceGetAttribute(&c, hipDeviceAttributeClockRate, dev) != hipSuccess) return COBEVT_ERR_LAUNCH;
    *wall_khz = w;
    *sclk_max_khz = c;
    return COBEVT_OK;
}

extern "C" int cobevt_calibrate_copy(const void* src, void* dst, long bytes, hipStream_t stream) {
    if (!src || !dst || bytes < 16 || (bytes & 15)) return COBEVT_ERR_ARG;
    hipLaunchKernelGGL(calib_copy_kernel, dim3(256 * 16), dim3(256), 0, stream, (const u32x4*)src, (u32x4*)dst, bytes / 16);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
