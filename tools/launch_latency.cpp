// Times repeated launches of a tiny GEMM / conv3x3 / LN through the C ABI (back-to-back, same kernel => warm code).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include "../include/cobevt_hip.h"

static float time_launches(int reps, auto&& fn) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0);
    for (int i = 0; i < reps; ++i) fn();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return ms * 1e3f / reps;
}

int main() {
    void *a, *w, *o, *r; float* bias;
    hipMalloc(&a, 256 << 20); hipMalloc(&w, 64 << 20); hipMalloc(&o, 256 << 20); hipMalloc(&r, 256 << 20); hipMalloc(&bias, 1 << 20);
    hipMemset(a, 0, 256 << 20); hipMemset(w, 0, 64 << 20); hipMemset(r, 0, 256 << 20); hipMemset(bias, 0, 1 << 20);
    for (int M : {5120, 20480, 81920, 327680}) {
        // dims: dtype, N, H, W, Cin, Ho, Wo, Cout, Kh, Kw, stride, pad, K, Kpad, upsample, pre_relu, act, store_mode, out_H, out_W, smallc
        int dims[21] = {0, 1, 1, M, 128, 1, M, 128, 1, 1, 1, 0, 128, 128, 0, 0, 0, 0, 1, M, 0};
        float us = time_launches(50, [&] { cobevt_conv2d_nhwc(a, w, bias, nullptr, nullptr, nullptr, nullptr, o, dims, 0); });
        printf("igemm 1x1 128->128 M=%6d : %.2f us per launch (back-to-back)\n", M, us);
    }
    {
        int d3[10] = {0, 5, 32, 32, 32, 32, 0, 1, 0, 32};
        float us = time_launches(50, [&] { cobevt_conv3x3_nhwc(a, w, bias, nullptr, o, d3, 0); });
        printf("conv3x3 32->32 5x32x32 : %.2f us\n", us);
        int d4[10] = {0, 20, 128, 128, 64, 64, 0, 1, 0, 64};
        us = time_launches(20, [&] { cobevt_conv3x3_nhwc(a, w, bias, r, o, d4, 0); });
        printf("conv3x3 64->64 20x128x128 : %.2f us\n", us);
        int d5[10] = {0, 20, 32, 32, 256, 256, 0, 1, 0, 64};
        us = time_launches(20, [&] { cobevt_conv3x3_nhwc(a, w, bias, r, o, d5, 0); });
        printf("conv3x3 256->256 20x32x32 : %.2f us\n", us);
    }
    {
        float* gam; hipMalloc(&gam, 4096); hipMemset(gam, 0, 4096);
        for (int M : {5120, 81920}) for (int ln = 0; ln < 2; ++ln) for (int N : {128, 384}) {
            long d[12] = {0, M, N, 128, 128, 128, 0, 0, 1, M, 1, M};
            float us = time_launches(30, [&] { cobevt_linear_rows(a, w, bias, nullptr, ln ? gam : nullptr, ln ? gam : nullptr, nullptr, nullptr, o, d, 1e-5f, 0); });
            printf("gemm_rows 128->%d M=%6d ln=%d : %.2f us\n", N, M, ln, us);
        }
    }
    // alternate two different kernels (code of each evicted?) 
    {
        int dims[21] = {0, 1, 1, 5120, 128, 1, 5120, 128, 1, 1, 1, 0, 128, 128, 0, 0, 0, 0, 1, 5120, 0};
        int d3[10] = {0, 5, 32, 32, 32, 32, 0, 1, 0, 32};
        float us = time_launches(25, [&] {
            cobevt_conv2d_nhwc(a, w, bias, nullptr, nullptr, nullptr, nullptr, o, dims, 0);
            cobevt_conv3x3_nhwc(a, w, bias, nullptr, o, d3, 0);
        });
        printf("alternating igemm(M=5120) + conv3x3(small): %.2f us per pair\n", us);
    }
    return 0;
}
