// Shared device helpers of the channels-last glue kernels: 8-channel row pieces as fp32, and the affine-warp arithmetic of the
// reference's STTF / ROI masks (torch_transformation_utils.py:108-134,160-191,254-355).
#pragma once
#include "common.hpp"

namespace cobevt {

// 8 consecutive channels of one row, as fp32
template <typename T> __device__ __forceinline__ void load8(const T* p, float* v) {
    if constexpr (Elem<T>::kIsBf16) {
        chunk_to_f32<T>(*(const uint4*)p, v);
    } else {
        chunk_to_f32<T>(*(const uint4*)p, v);
        chunk_to_f32<T>(*(const uint4*)(p + 4), v + 4);
    }
}
template <typename T> __device__ __forceinline__ void store8(T* p, const float* v) {
    if constexpr (Elem<T>::kIsBf16) {
        *(uint4*)p = f32_to_chunk<T>(v);
    } else {
        *(uint4*)p = f32_to_chunk<T>(v);
        *(uint4*)(p + 4) = f32_to_chunk<T>(v + 4);
    }
}

struct Affine { float t[6]; };

// theta = rows 0..1 of inverse( Nrm(Hd,Wd) @ [T;0 0 1] @ Nrm(Hd,Wd)^-1 ),  T = rot about (Wd/2,Hd/2) + translation
// centred = false: the discretised matrix as it is (get_rotated_roi, torch_transformation_utils.py:77-105, hands it to warp_affine
// without the re-centring of :254-297)
__device__ inline Affine sttf_theta(const float* m44, float discrete_ratio, float downsample_rate, int Hd, int Wd, bool centred = true) {
    const float r00 = m44[0], r01 = m44[1], r10 = m44[4], r11 = m44[5];
    const float div = discrete_ratio * downsample_rate;
    const float tx = m44[3] / div, ty = m44[7] / div;
    const float cx = (float)Wd / 2.f, cy = (float)Hd / 2.f;
    // shift(c) @ rot @ shift(-c), then + translation
    const float T02 = centred ? (r00 * -cx + r01 * -cy + cx) + tx : tx;
    const float T12 = centred ? (r10 * -cx + r11 * -cy + cy) + ty : ty;
    // M @ Ninv, Ninv = [[(W-1)/2,0,(W-1)/2],[0,(H-1)/2,(H-1)/2],[0,0,1]]
    const float sx = ((float)Wd - 1.f) / 2.f, sy = ((float)Hd - 1.f) / 2.f;
    const float a00 = r00 * sx, a01 = r01 * sy, a02 = r00 * sx + r01 * sy + T02;
    const float a10 = r10 * sx, a11 = r11 * sy, a12 = r10 * sx + r11 * sy + T12;
    // N @ (.), N = [[2/(W-1),0,-1],[0,2/(H-1),-1],[0,0,1]]
    const float nx = 2.f / ((float)Wd - 1.f), ny = 2.f / ((float)Hd - 1.f);
    const float d00 = nx * a00, d01 = nx * a01, d02 = nx * a02 - 1.f;
    const float d10 = ny * a10, d11 = ny * a11, d12 = ny * a12 - 1.f;
    const float det = d00 * d11 - d01 * d10;
    Affine A;
    A.t[0] = d11 / det;  A.t[1] = -d01 / det; A.t[2] = (d01 * d12 - d02 * d11) / det;
    A.t[3] = -d10 / det; A.t[4] = d00 / det;  A.t[5] = (d02 * d10 - d00 * d12) / det;
    return A;
}

__device__ __forceinline__ void affine_sample_xy(const Affine& A, int i, int j, int Hd, int Wd, float& ix, float& iy) {
    // affine_grid(align_corners=True) base coordinates, then grid_sample un-normalisation
    const float xn = Wd > 1 ? (2.f * (float)j / (float)(Wd - 1) - 1.f) : 0.f;
    const float yn = Hd > 1 ? (2.f * (float)i / (float)(Hd - 1) - 1.f) : 0.f;
    const float xs = xn * A.t[0] + yn * A.t[1] + A.t[2];
    const float ys = xn * A.t[3] + yn * A.t[4] + A.t[5];
    ix = (xs + 1.f) * 0.5f * (float)(Wd - 1);
    iy = (ys + 1.f) * 0.5f * (float)(Hd - 1);
}

}  // namespace cobevt
