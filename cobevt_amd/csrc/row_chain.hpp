// Parameter block of the fused row-local chain (row_chain.hip) and of its 64-channel persistent form (row_chain64.hip).
#pragma once
#include "common.hpp"

namespace cobevt {

struct RowChainParams {
    const bf16_t* a;        // [M][C] attention output
    const bf16_t* skip;     // [skip_rows][C] or null; row m adds skip[m % skip_rows] (skip_rows = M: plain; < M: broadcast)
    bf16_t* out;            // [M][C]
    const uint4* wp;        // fragment-ordered [4 tiles][8]      out-projection
    const float* bp;        // [C] or null
    const uint4* w1;        // fragment-ordered [8 tiles][8]      fc1 with the LayerNorm affine folded in
    const float* b1;        // [Hd]
    const uint4* w2;        // fragment-ordered [4 tiles][Hdp/16] fc2
    const float* b2;        // [C]
    const float* post_g;    // post-LayerNorm affine or null
    const float* post_b;
    const uint4* wn;        // fragment-ordered [4*ceil(Nn/128) tiles][8]  next projection or null
    const float* bn;        // [Nn] or null
    bf16_t* out_next;       // [M][Nn]
    int M, C, Hd, Hdp;
    int Nn, next_ln, next_act, skip_rows;
    float eps1, eps_post, eps_next;
    // MLP = false ("projection chain"): a <- ReLU?(a * pre_scale[c] + pre_shift[c]) while it is staged (pre-activation
    // BatchNorm -> ReLU -> 1x1 conv, fax_modules.py:281-292), y = a . Wp^T + bp + skip, `out` is not stored unless non-null, and the
    // next projection (LayerNorm + Linear: to_k / to_v of both cross attentions, fax_modules.py:201-205) reads y from LDS
    const float* pre_scale;
    const float* pre_shift;
    int pre_relu;
};

// row_chain64.hip: the chain for C = 64 / hidden 128 (LiDAR FuseBEVT, swap_fusion_modules.py:87-128 with input_dim 64) as persistent
// workgroups with every weight fragment resident in registers; returns COBEVT_OK, an error code, or -1 when the shape does not
// qualify (the caller then launches the generic kernel)
int launch_row_chain64(const RowChainParams& p, hipStream_t stream);
// proj_chain128.hip: the projection chain (MLP = false form: pre-activation -> Wp + skip -> LN -> Wn) on big 128-channel maps, same idea
int launch_proj_chain128(const RowChainParams& p, hipStream_t stream);

// row_chain_f32.hip: the same chain for fp32 storage (C = 128, hidden 256; exact / split-bf16 matrix path by library); -1 = shape does not qualify
int launch_row_chain_f32(const void* a, const void* skip, void* out, const void* wp, const float* bp, const void* w1, const float* b1,
                         const void* w2, const float* b2, const float* post_g, const float* post_b, const void* wnext, const float* bnext,
                         void* out_next, int M, int C, int Hd, int Hdp, int Nn, int next_ln, int next_act, int skip_rows, float eps1,
                         float eps_post, float eps_next, hipStream_t stream);

}  // namespace cobevt
