// Sanity microbenchmarks: MFMA peak (clock check), dependent global-load latency, barrier+load step cost.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__global__ __launch_bounds__(256) void mfma_peak(float* out, int iters) {
    bf16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(float)(threadIdx.x + i); b[i] = (__bf16)(float)(i + 1); }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * 256 + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

__global__ void chase(const int* next, int* out, long long* cyc, int steps) {
    int p = 0;
    long long t0 = wall_clock64();
    long long c0 = clock64();
    for (int i = 0; i < steps; ++i) p = next[p];
    long long c1 = clock64();
    long long t1 = wall_clock64();
    out[0] = p; cyc[0] = c1 - c0; cyc[1] = t1 - t0;
}

// one "step" of the staged-GEMM pattern: global load -> lds store -> barrier -> lds read
__global__ __launch_bounds__(256) void step_kernel(const uint4* src, uint4* dst, int steps, int stride) {
    __shared__ uint4 buf[512];
    uint4 acc = make_uint4(0, 0, 0, 0);
    const uint4* p = src + blockIdx.x * 4096 + threadIdx.x;
    for (int i = 0; i < steps; ++i) {
        uint4 v = p[i * stride];
        buf[threadIdx.x] = v;
        __syncthreads();
        uint4 w = buf[(threadIdx.x + 17) & 255];
        acc.x += w.x; acc.y ^= w.y;
        __syncthreads();
    }
    dst[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main() {
    hipDeviceProp_t prop; hipGetDeviceProperties(&prop, 0);
    int wall_khz = 0; hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    printf("device %s CUs %d clock %d kHz wallclock %d kHz\n", prop.name, prop.multiProcessorCount, prop.clockRate, wall_khz);
    float* out; hipMalloc(&out, 1 << 24);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) {
        int iters = 20000, blocks = 256 * 4;
        hipEventRecord(e0);
        mfma_peak<<<blocks, 256>>>(out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        double flops = (double)blocks * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
        printf("mfma_peak: %.3f ms  %.1f TFLOP/s\n", ms, flops / ms / 1e9);
    }
    // pointer chase over 64 MB (stride 4 KB+) to defeat caches
    int n = 1 << 24;
    std::vector<int> h(n, 0);
    int cur = 0; const int hop = 1031 * 16;
    for (int i = 0; i < 4096; ++i) { int nx = (cur + hop) % n; h[cur] = nx; cur = nx; }
    int *dn, *dout; long long* dc;
    hipMalloc(&dn, n * 4); hipMalloc(&dout, 64); hipMalloc(&dc, 64);
    hipMemcpy(dn, h.data(), n * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        chase<<<1, 1>>>(dn, dout, dc, 2000);
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf("chase (HBM/TLB-unfriendly): %.1f shader cycles/load, %.1f wallclock ticks/load (%.1f ns)\n", c[0] / 2000.0, c[1] / 2000.0,
               c[1] / 2000.0 * 1e6 / wall_khz);
    }
    // small footprint chase (L2 resident)
    for (int i = 0; i < 4096; ++i) h[i * 16] = ((i + 1) % 4096) * 16;
    hipMemcpy(dn, h.data(), 4096 * 16 * 4, hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) {
        chase<<<1, 1>>>(dn, dout, dc, 2000);
        hipDeviceSynchronize();
        long long c[2]; hipMemcpy(c, dc, 16, hipMemcpyDeviceToHost);
        printf("chase (256 KB footprint): %.1f shader cycles/load, %.1f ns\n", c[0] / 2000.0, c[1] / 2000.0 * 1e6 / wall_khz);
    }
    uint4 *src, *dst; hipMalloc(&src, 256ull << 20); hipMalloc(&dst, 1 << 24);
    hipMemset(src, 1, 256ull << 20);
    for (int blocks : {1, 80, 512}) for (int steps : {4, 16}) {
        step_kernel<<<blocks, 256>>>(src, dst, steps, 256);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 20; ++r) step_kernel<<<blocks, 256>>>(src, dst, steps, 256);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("step_kernel blocks %4d steps %2d: %.2f us per launch -> %.2f us per step\n", blocks, steps, ms * 1e3 / 20, ms * 1e3 / 20 / steps);
    }
    return 0;
}
