// Sustained shader clock and MFMA issue rate on this box, and what each ingredient of the conv main loop costs:
//   variant 0: independent v_mfma_f32_32x32x16_bf16 on 4 accumulators, constant operands
//   variant 1: 5 accumulators, 5 distinct A registers, one B register (the strip kernel's k-group)
//   variant 2: variant 1 + one ds_read_b128 per MFMA feeding a 3-slot A ring (read two k-groups ahead)
//   variant 3: variant 2 + one 1-KB global (L2-resident) B fragment load per k-group into a 3-slot ring
// timed with s_memtime (shader clock) and s_memrealtime (100 MHz) from inside the kernel.
// Build: hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/clock_probe.hip -o tools/_probe/libclock_probe.so
#include <hip/hip_runtime.h>
typedef __attribute__((__vector_size__(16 * sizeof(float)))) float f32x16;
typedef __attribute__((__vector_size__(8 * sizeof(__bf16)))) __bf16 bf16x8;

__device__ __forceinline__ bf16x8 as_bf(uint4 v) { return __builtin_bit_cast(bf16x8, v); }

template <int V>
__global__ __launch_bounds__(512) void mfma_loop(unsigned long long* out, const uint4* wsrc, int iters) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    constexpr int NACC = V == 0 ? 4 : 5;
    f32x16 acc[NACC];
    for (int a = 0; a < NACC; ++a) for (int r = 0; r < 16; ++r) acc[a][r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63;
    for (int i = tid; i < 65536 / 16; i += 512) ((uint4*)smem)[i] = make_uint4(i, i * 3, i * 5, 0x3f803f80u);
    __syncthreads();
    uint4 af[3][5], bq[3];
    for (int s = 0; s < 3; ++s) {
        bq[s] = make_uint4(0x3f803f80u, s, lane, 0x3f803f80u);
        for (int a = 0; a < 5; ++a) af[s][a] = make_uint4(0x3f803f80u, a, lane + s, 0x3f803f80u);
    }
    // conflict-free 16-byte reads at the strip kernel's 144-byte pixel stride
    const unsigned char* abase = smem + (lane & 31) * 144 + (lane >> 5) * 16;
    const uint4* wq = wsrc + lane;
    unsigned long long c0 = __builtin_readcyclecounter(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int n = 0; n < 3; ++n) {                        // three k-groups per iteration: ring slots static
            if (V >= 3) bq[(n + 2) % 3] = wq[((i * 3 + n) & 255) * 64];
            if (V >= 2) {
#pragma unroll
                for (int a = 0; a < 5; ++a) af[(n + 2) % 3][a] = *(const uint4*)(abase + a * 5120 + ((i * 3 + n) & 7) * 32);
            }
#pragma unroll
            for (int a = 0; a < NACC; ++a)
                acc[a] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as_bf(V == 0 ? bq[0] : bq[n]), as_bf(V == 0 ? af[0][0] : af[n][a]), acc[a], 0, 0, 0);
            if (V >= 2) {
#pragma unroll
                for (int a = 0; a < 5; ++a) {
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    unsigned long long c1 = __builtin_readcyclecounter(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0.f;
    for (int a = 0; a < NACC; ++a) s += acc[a][0];
    if (threadIdx.x == 0) {
        out[blockIdx.x * 4 + 0] = c1 - c0;
        out[blockIdx.x * 4 + 1] = r1 - r0;
        out[blockIdx.x * 4 + 2] = (unsigned long long)s;
    }
}

extern "C" int clock_probe(unsigned long long* out, const void* wsrc, int variant, int blocks, int threads, int iters, hipStream_t stream) {
    const size_t lds = 65536;
#define LAUNCH(V) hipLaunchKernelGGL(mfma_loop<V>, dim3(blocks), dim3(threads), lds, stream, out, (const uint4*)wsrc, iters)
    switch (variant) {
        case 0: LAUNCH(0); break;
        case 1: LAUNCH(1); break;
        case 2: LAUNCH(2); break;
        default: LAUNCH(3); break;
    }
    return (int)hipGetLastError();
}
