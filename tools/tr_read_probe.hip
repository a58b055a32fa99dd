// probe: semantics of ds_read_b64_tr_b16 (__builtin_amdgcn_ds_read_tr16_b64_v4i16) on gfx950: hipcc --offload-arch=gfx950 -O3 tools/tr_read_probe.hip -o /tmp/tr && /tmp/tr
// -> lane l of a 16-lane group receives column (l & 15) of the 4 x 16 block of 16-bit elements the group's lanes read 8 bytes each of ("bad 0")
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short v4s __attribute__((ext_vector_type(4)));
__global__ void k(const unsigned short* in, unsigned short* out) {
    __shared__ unsigned short lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) lds[i] = in[i];
    __syncthreads();
    int l = threadIdx.x;
    const int s = l & 15, g = l >> 4;
    const unsigned short* p = lds + (g * 4 + (s >> 2)) * 16 + 4 * (s & 3);
    v4s r = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)p);
    for (int j = 0; j < 4; ++j) out[l * 4 + j] = (unsigned short)r[j];
}
int main() {
    unsigned short h[4096], o[256];
    for (int i = 0; i < 4096; ++i) h[i] = i;
    unsigned short *d, *e;
    hipMalloc(&d, sizeof(h)); hipMalloc(&e, sizeof(o));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, e);
    hipMemcpy(o, e, sizeof(o), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) for (int j = 0; j < 4; ++j) { int ex = (4 * (l >> 4) + j) * 16 + (l & 15); if (o[l * 4 + j] != ex) ++bad; }
    printf("bad %d\n", bad);
    for (int l = 0; l < 20; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, o[l*4], o[l*4+1], o[l*4+2], o[l*4+3]);
    return 0;
}
