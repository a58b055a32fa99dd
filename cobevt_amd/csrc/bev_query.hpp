// Parameter block of the fused BEV-query kernel (bev_query.hip); launched from gemm_rows3.hip's embedding entry point.
#pragma once
#include "common.hpp"

namespace cobevt {

struct BevQueryParams {
    const float* E_inv;     // [B * n][16] camera matrices (row-major 4 x 4)
    const float* world;     // [2][hw] BEV grid coordinates
    const float* w_bev;     // [128][2]
    const float* b_bev;     // [128]
    const float* w_cam;     // [128][4]
    const bf16_t* x;        // [B or 1][hw][128]
    const uint4* wfrag;     // to_q weight (LayerNorm affine folded) in fragment order [4 tiles][8 k-groups][64 lanes]
    const float* bias;      // [128] or null
    bf16_t* out;            // [B][n][hw][128]
    int B, n, hw, x_bcast;
    float ln_eps;
};

int launch_bev_query(const BevQueryParams& p, hipStream_t stream);

}  // namespace cobevt
