// Sustained shader clock and MFMA issue rate on this box: a wave-per-SIMD (or more) loop of independent
// v_mfma_f32_32x32x16_bf16, timed with s_memtime (shader clock) and s_memrealtime (100 MHz) from inside the kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/_probe/libclock_probe.so
#include <hip/hip_runtime.h>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

__global__ __launch_bounds__(256) void mfma_loop(unsigned long long* out, int iters, int nacc) {
    f32x16 acc[4];
    for (int a = 0; a < 4; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    bf16x8 x, y;
    for (int e = 0; e < 8; ++e) { x[e] = (__bf16)(float)(threadIdx.x & 3); y[e] = (__bf16)1.0f; }
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int a = 0; a < 4; ++a) acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(x, y, acc[a], 0, 0, 0);
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < 4; ++a) s += acc[a][0];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = c1 - c0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
        out[blockIdx.x * 4 + 2] = (unsigned long long)s;
    }
}

extern "C" int clock_probe(unsigned long long* out, int blocks, int threads, int iters, hipStream_t stream) {
    hipLaunchKernelGGL(mfma_loop, dim3(blocks), dim3(threads), 0, stream, out, iters, 4);
    return (int)hipGetLastError();
}
