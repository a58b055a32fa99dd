// Which dynamic-LDS size still fits on a CU BESIDE a resident workgroup of the 3x3 strip kernel (113,280 B), of the fused BasicBlocks
// (89,472 / 113,280 B) ...?  Kernel A: 256 workgroups x 512 threads with `a_bytes` of LDS, spinning ~300 us.  Kernel B (another
// stream, launched while A runs): 256 workgroups x 256 threads with `b_bytes`, returns at once.  B's completion time says whether its
// workgroups found room next to A's (tens of us) or waited for A to retire (> 300 us).
// build: hipcc --offload-arch=gfx950 -O2 tools/lds_coreside.hip -o tools/_probe/lds_coreside ; run: tools/_probe/lds_coreside [a_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void spin_kernel(long long cycles, int* sink) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < cycles) __builtin_amdgcn_s_sleep(8);
    if (smem[(threadIdx.x + 1) & 255] == 77 && cycles < 0) *sink = 1;
}

__global__ void touch_kernel(int* sink) {
    extern __shared__ unsigned char smem[];
    smem[threadIdx.x] = (unsigned char)threadIdx.x;
    __syncthreads();
    if (smem[(threadIdx.x + 1) & 255] == 77 && sink == nullptr) *sink = 1;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main(int argc, char** argv) {
    std::vector<int> a_sizes = {113280, 89472, 80336, 62464};
    if (argc > 1) { a_sizes.clear(); for (int i = 1; i < argc; i++) a_sizes.push_back(atoi(argv[i])); }
    int* sink;
    CK(hipMalloc(&sink, 4));
    hipStream_t sa, sb;
    CK(hipStreamCreateWithFlags(&sa, hipStreamNonBlocking));
    CK(hipStreamCreateWithFlags(&sb, hipStreamNonBlocking));
    CK(hipFuncSetAttribute((const void*)spin_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute((const void*)touch_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const long long spin = 30000;      // wall_clock64 ticks at 100 MHz: 300 us
    for (int a : a_sizes) {
        int last_fit = -1, first_wait = -1;
        for (int b = 163840 - a - 8192; b <= 163840 - a + 2048; b += 128) {
            if (b <= 0) continue;
            hipLaunchKernelGGL(spin_kernel, dim3(256), dim3(512), a, sa, spin, sink);
            // let A's workgroups land first
            hipLaunchKernelGGL(spin_kernel, dim3(1), dim3(64), 0, sb, 2000LL, sink);
            CK(hipEventRecord(e0, sb));
            hipLaunchKernelGGL(touch_kernel, dim3(256), dim3(256), b, sb, sink);
            CK(hipEventRecord(e1, sb));
            CK(hipDeviceSynchronize());
            float ms = 0;
            CK(hipEventElapsedTime(&ms, e0, e1));
            const bool fit = ms < 0.15f;
            if (fit) last_fit = b; else if (first_wait < 0) first_wait = b;
            if (b % 1024 == 0 || (!fit && first_wait == b)) printf("  A %6d B  B %6d B: %.3f ms %s\n", a, b, ms, fit ? "beside" : "waited");
        }
        printf("A = %d B: largest B beside it %d B (sum %d), first B that waited %d B (sum %d)\n", a, last_fit, a + last_fit, first_wait, a + first_wait);
    }
    return 0;
}
