// HBM-bound glue kernels of the CoBEVT hot path (gfx950).  Every kernel moves 16-byte chunks per lane over
// channels-last rows; a row of C channels is owned by a group of C/8 adjacent lanes so loads/stores stay
// coalesced and the per-row reductions are xor-shuffles inside the group.
#include "warp_common.hpp"

namespace cobevt {

// ---------------------------------------------------------------------------------------------
// LayerNorm over the channel axis (eps inside sqrt, biased variance), optionally preceded by a mean over
// `navg` slices `avg_stride` elements apart (SwapFusionEncoder.mlp_head: mean over agents then LayerNorm).
// reference: fax_modules.py:189-191,309-313,435-437 ; swap_fusion_modules.py:275-279 ; base_transformer.py:102-109
template <typename T>
__global__ __launch_bounds__(256) void layernorm_kernel(const T* in, const float* gamma, const float* beta, T* out,
                                                        int rows, int C, float eps, int navg, long avg_stride,
                                                        long in_batch_stride, int rows_per_batch) {
    const int G = C >> 3;  // lanes per row
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= rows) return;
    // with averaging the input row r of batch bb lives at bb*in_batch_stride + r*C (+ s*avg_stride)
    const int bb = gid / rows_per_batch, rr = gid - bb * rows_per_batch;
    const T* src = in + (size_t)bb * in_batch_stride + (size_t)rr * C + gl * 8;
    float v[8];
    load8<T>(src, v);
    if (navg > 1) {
        for (int s = 1; s < navg; ++s) {
            float w[8];
            load8<T>(src + (size_t)s * avg_stride, w);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += w[e];
        }
        const float inv = 1.0f / (float)navg;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= inv;
    }
    float s = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) s += v[e];
    s = wave_sum_xor(s, G);
    const float mean = s / (float)C;
    float q = 0.f;
#pragma unroll
    for (int e = 0; e < 8; ++e) { const float d = v[e] - mean; q += d * d; }
    q = wave_sum_xor(q, G);
    const float rstd = rsqrtf(q / (float)C + eps);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        v[e] = (v[e] - mean) * rstd;
        if (gamma) v[e] = v[e] * gamma[gl * 8 + e] + beta[gl * 8 + e];     // null affine: plain normalisation
    }
    store8<T>(out + (size_t)gid * C + gl * 8, v);
}

// ---------------------------------------------------------------------------------------------
// Camera-ray positional embedding:  img_embed = L2norm_c( img_embed_conv(E_inv @ pad1(I_inv @ pixel)) - cam_embed(E_inv[:, 3]) )
// reference: fax_modules.py:346-358.  Output (BN, h, w, D) channels-last.
// One workgroup = one camera (blockIdx.y) x a strip of pixels; a lane keeps the weights of its 8 channels and the
// camera embedding c_embed in registers and walks the strip (the weights used to be re-read per pixel).
constexpr int kEmbedPixPerBlock = 256;
// the ray embedding's maps are small (20 cameras x 4096 / 1024 / 256 pixels): 64 pixels per workgroup = ONE batch of global
// round trips per lane group instead of four dependent ones (17-21 us per launch in the frame's graph with 256)
constexpr int kRayPixPerBlock = 64;

template <typename T>
__global__ __launch_bounds__(256) void ray_embed_kernel(const float* I_inv, const float* E_inv, const float* plane,
                                                        const float* w_img, const float* w_cam, T* out, int BN,
                                                        int hw, int D) {
    const int G = D >> 3;                       // lanes per pixel
    const int gl = threadIdx.x & (G - 1), gp = threadIdx.x / G, ngroups = 256 / G;
    const int bn = blockIdx.y;
    const float* I = I_inv + bn * 9;
    const float* E = E_inv + bn * 16;
    float wi[8][4], ce[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = gl * 8 + e;
        const float4 a = *(const float4*)(w_img + ch * 4), c = *(const float4*)(w_cam + ch * 4);
        wi[e][0] = a.x; wi[e][1] = a.y; wi[e][2] = a.z; wi[e][3] = a.w;
        ce[e] = c.x * E[3] + c.y * E[7] + c.z * E[11] + c.w * E[15];
    }
    const int p0 = blockIdx.x * kRayPixPerBlock;
    constexpr int U = 4;                        // pixels of a lane group in flight together (the rolled loop paid one global
    for (int pb = p0 + gp; pb < min(p0 + kRayPixPerBlock, hw); pb += ngroups * U) {   // round trip per pixel)
        float pxs[U], pys[U], pzs[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(pb + u * ngroups, hw - 1);
            pxs[u] = plane[q]; pys[u] = plane[hw + q]; pzs[u] = plane[2 * hw + q];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
        const int pix = pb + u * ngroups;
        if (pix >= min(p0 + kRayPixPerBlock, hw)) break;
        const float px = pxs[u], py = pys[u], pz = pzs[u];
        float cam[4];
#pragma unroll
        for (int r = 0; r < 3; ++r) cam[r] = I[r * 3 + 0] * px + I[r * 3 + 1] * py + I[r * 3 + 2] * pz;
        cam[3] = 1.f;
        float d4[4];
#pragma unroll
        for (int r = 0; r < 4; ++r)
            d4[r] = E[r * 4 + 0] * cam[0] + E[r * 4 + 1] * cam[1] + E[r * 4 + 2] * cam[2] + E[r * 4 + 3] * cam[3];
        float v[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float de = wi[e][0] * d4[0] + wi[e][1] * d4[1] + wi[e][2] * d4[2] + wi[e][3] * d4[3];
            v[e] = de - ce[e];
            ss += v[e] * v[e];
        }
        ss = wave_sum_xor(ss, G);
        const float inv = 1.0f / (sqrtf(ss) + 1e-7f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= inv;
        store8<T>(out + ((size_t)bn * hw + pix) * D + gl * 8, v);
        }
    }
}

// BEV query positional embedding + prior: query[b,n] = L2norm_c( bev_embed(world) - cam_embed(c) ) + x[b]
// reference: fax_modules.py:370-375,387-388.  world: (2, HW) fp32 ; x: (B, HW, D) ; out: (B, n, HW, D)
template <typename T>
__global__ __launch_bounds__(256) void bev_embed_kernel(const float* E_inv, const float* world, const float* w_bev,
                                                        const float* b_bev, const float* w_cam, const T* x, T* out,
                                                        int B, int n, int hw, int D, int x_bcast) {
    const int G = D >> 3;
    const int gl = threadIdx.x & (G - 1), gp = threadIdx.x / G, ngroups = 256 / G;
    const int bn = blockIdx.y, b = x_bcast ? 0 : bn / n;     // x_bcast: one (HW, D) prior shared by every b
    const float* E = E_inv + bn * 16;
    float wb[8][2], off[8];                     // off = bias - c_embed
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int ch = gl * 8 + e;
        const float4 c = *(const float4*)(w_cam + ch * 4);
        wb[e][0] = w_bev[ch * 2 + 0]; wb[e][1] = w_bev[ch * 2 + 1];
        off[e] = c.x * E[3] + c.y * E[7] + c.z * E[11] + c.w * E[15];
    }
    float bb[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) bb[e] = b_bev[gl * 8 + e];
    const int p0 = blockIdx.x * kEmbedPixPerBlock;
    constexpr int U = 4;                        // pixels of a lane group in flight together
    for (int pb = p0 + gp; pb < min(p0 + kEmbedPixPerBlock, hw); pb += ngroups * U) {
        float wxs[U], wys[U], xvs[U][8];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int q = min(pb + u * ngroups, hw - 1);
            wxs[u] = world[q]; wys[u] = world[hw + q];
            load8<T>(x + ((size_t)b * hw + q) * D + gl * 8, xvs[u]);
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
        const int pix = pb + u * ngroups;
        if (pix >= min(p0 + kEmbedPixPerBlock, hw)) break;
        const float wx = wxs[u], wy = wys[u];
        float v[8], ss = 0.f;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float we = wb[e][0] * wx + wb[e][1] * wy + bb[e];
            v[e] = we - off[e];
            ss += v[e] * v[e];
        }
        ss = wave_sum_xor(ss, G);
        const float inv = 1.0f / (sqrtf(ss) + 1e-7f);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = v[e] * inv + xvs[u][e];
        store8<T>(out + ((size_t)bn * hw + pix) * D + gl * 8, v);
        }
    }
}

// ---------------------------------------------------------------------------------------------
// F-Cooper max-out over the agent slots: out[b][i] = max_l in[b][l][i], 8 elements per thread.
// reference: opv2v/opencood/models/fusion_modules/f_cooper_fuse.py:30-36 (SpatialFusionMask: torch.max(x, dim=1)[0])
template <typename T>
__global__ __launch_bounds__(256) void agent_max_kernel(const T* in, T* out, int B, int L, long per8) {
    const long gid = (long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= (long)B * per8) return;
    const long b = gid / per8, i = gid - b * per8;
    float m[8];
    load8<T>(in + ((size_t)b * L * per8 + i) * 8, m);
    for (int l = 1; l < L; ++l) {
        float v[8];
        load8<T>(in + (((size_t)b * L + l) * per8 + i) * 8, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
    }
    store8<T>(out + (size_t)gid * 8, m);
}

// ---------------------------------------------------------------------------------------------
// MaxPool2d(kernel 3, stride 2, padding 1) channels-last.  reference: resnet_ms.py:71 (torchvision resnet maxpool)
template <typename T>
__global__ __launch_bounds__(256) void maxpool3x3s2_kernel(const T* in, T* out, int N, int H, int W, int C, int Ho,
                                                           int Wo) {
    const int G = C >> 3;
    const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= (long)N * Ho * Wo) return;
    const int n = (int)(gid / (Ho * Wo)), rem = (int)(gid - (long)n * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    float m[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) m[e] = -INFINITY;
    for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh * 2 - 1 + kh;
        if (ih < 0 || ih >= H) continue;
        for (int kw = 0; kw < 3; ++kw) {
            const int iw = ow * 2 - 1 + kw;
            if (iw < 0 || iw >= W) continue;
            float v[8];
            load8<T>(in + (((size_t)n * H + ih) * W + iw) * C + gl * 8, v);
#pragma unroll
            for (int e = 0; e < 8; ++e) m[e] = fmaxf(m[e], v[e]);
        }
    }
    store8<T>(out + gid * C + gl * 8, m);
}

// ---------------------------------------------------------------------------------------------
// Strided (n,c,h,w) <-> channels-last conversion with dtype cast (module-boundary plumbing only).
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void to_nhwc_kernel(const TI* in, TO* out, int N, int C, int H, int W, long sN, long sC,
                                                      long sH, long sW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * C * H * W;
    if (i >= total) return;
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    store_elem<TO>(out, i, load_elem<TI>(in, n * sN + c * sC + h * sH + w * sW));
}
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void from_nhwc_kernel(const TI* in, TO* out, int N, int C, int H, int W, long sN,
                                                        long sC, long sH, long sW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    const long total = (long)N * C * H * W;
    if (i >= total) return;
    const int c = (int)(i % C);
    long r = i / C;
    const int w = (int)(r % W); r /= W;
    const int h = (int)(r % H);
    const int n = (int)(r / H);
    store_elem<TO>(out, n * sN + c * sC + h * sH + w * sW, load_elem<TI>(in, i));
}

// ---------------------------------------------------------------------------------------------
// regroup: split the agent batch by record_len, zero-pad every sample to max_cav agents, emit the agent mask.
// reference: fuse_utils.py:8-61 (which syncs the host at :26; this kernel reads record_len on the device).
template <typename T>
__global__ __launch_bounds__(256) void regroup_kernel(const T* in, const int* record_len, T* out, float* mask, int B,
                                                      int max_cav, long chunks_per_agent) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;  // 16-byte chunk index of the output
    const long total = (long)B * max_cav * chunks_per_agent;
    if (i >= total) return;
    const long agent_slot = i / chunks_per_agent, within = i - agent_slot * chunks_per_agent;
    const int b = (int)(agent_slot / max_cav), l = (int)(agent_slot - (long)b * max_cav);
    int off = 0;
    for (int bb = 0; bb < b; ++bb) off += record_len[bb];
    const bool present = l < record_len[b];
    uint4 v = make_uint4(0, 0, 0, 0);
    if (present) v = ((const uint4*)in)[(long)(off + l) * chunks_per_agent + within];
    ((uint4*)out)[i] = v;
    if (within == 0) mask[agent_slot] = present ? 1.f : 0.f;
}

// ---------------------------------------------------------------------------------------------
// STTF: warp every agent's BEV map into the ego frame + ROI/agent mask.
// reference: corpbevt.py:28-64 (transpose + flip, affine warp, flip + transpose back, channels last),
//            torch_transformation_utils.py:108-134 (discretise), :254-297 (rotation about (W/2,H/2) + translation),
//            :160-191 (normalise with (W-1),(H-1)), :317-355 (inverse, affine_grid + grid_sample align_corners=True),
//            :11-105 (ROI mask: nearest warp of ones, NOT transposed/flipped, times agent mask).
template <typename T>
__global__ __launch_bounds__(256) void sttf_warp_kernel(const T* x, const float* tmat, const float* cav_mask, T* out,
                                                        float* com_mask, const int* record_len, float* cav_out,
                                                        int B, int Lc, int H, int W, int C,
                                                        float discrete_ratio, float downsample_rate) {
    // x: (B*L, H, W, C), or with record_len the un-grouped agent batch (sum(record_len), H, W, C) (regroup,
    // fuse_utils.py:8-61, folded in: sample b's agent l is row sum(record_len[:b]) + l, absent agents warp to zeros);
    // out: (B, L, H, W, C) ; com_mask: (B, H, W, 1, L) ; cav_out: (B, L) agent mask when regrouping
    const int G = C >> 3;
    const int bl = blockIdx.y;
    const int b = bl / Lc, l = bl - b * Lc;
    __shared__ Affine th_feat, th_mask;
    __shared__ int src_agent;
    if (threadIdx.x == 0) {
        th_feat = sttf_theta(tmat + (size_t)bl * 16, discrete_ratio, downsample_rate, /*Hd=*/W, /*Wd=*/H);
        th_mask = sttf_theta(tmat + (size_t)bl * 16, discrete_ratio, downsample_rate, /*Hd=*/H, /*Wd=*/W);
        int src = bl;
        if (record_len) {
            int off = 0;
            for (int bb = 0; bb < b; ++bb) off += record_len[bb];
            src = l < record_len[b] ? off + l : -1;
            if (blockIdx.x == 0 && cav_out) cav_out[bl] = src >= 0 ? 1.f : 0.f;
        }
        src_agent = src;
    }
    __syncthreads();
    const int srcb = src_agent;
    const float present = record_len ? (srcb >= 0 ? 1.f : 0.f) : (cav_mask ? cav_mask[bl] : 1.f);
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= H * W) return;
    const int h = gid / W, w = gid - h * W;
    // feature: warped map y has dims (Hd=W, Wd=H); out[h][w] = y[i=w][j=H-1-h]; y samples x2[iy][ix] = x0[H-1-ix][iy]
    float ix, iy;
    affine_sample_xy(th_feat, w, H - 1 - h, W, H, ix, iy);
    const float fx = floorf(ix), fy = floorf(iy);
    const int x0 = (int)fx, y0 = (int)fy, x1 = x0 + 1, y1 = y0 + 1;
    const float wx1 = ix - fx, wy1 = iy - fy, wx0 = 1.f - wx1, wy0 = 1.f - wy1;
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.f;
    const T* src = x + (size_t)(srcb >= 0 ? srcb : 0) * H * W * C + gl * 8;
    auto tap = [&](int xx, int yy, float wgt) {
        // (xx, yy) indexes x2 with dims (rows W, cols H)
        if (srcb < 0 || xx < 0 || xx >= H || yy < 0 || yy >= W) return;
        float v[8];
        load8<T>(src + ((size_t)(H - 1 - xx) * W + yy) * C, v);
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] += v[e] * wgt;
    };
    tap(x0, y0, wx0 * wy0);
    tap(x1, y0, wx1 * wy0);
    tap(x0, y1, wx0 * wy1);
    tap(x1, y1, wx1 * wy1);
    store8<T>(out + ((size_t)bl * H * W + gid) * C + gl * 8, acc);
    if (gl == 0 && com_mask) {
        float mx, my;
        affine_sample_xy(th_mask, h, w, H, W, mx, my);
        const float rx = nearbyintf(mx), ry = nearbyintf(my);
        const bool inb = rx >= 0.f && rx <= (float)(W - 1) && ry >= 0.f && ry <= (float)(H - 1);
        com_mask[((size_t)b * H * W + gid) * Lc + l] = inb ? present : 0.f;
    }
}


// ---------------------------------------------------------------------------------------------
// Batched inverse of tiny (3x3 / 4x4) camera matrices: Gauss-Jordan with partial pivoting in fp64, one matrix
// per thread.  reference: fax_modules.py:500-501 (intrinsic.inverse()), encoder_pyramid_axial.py:538-539.
template <int D>
__global__ __launch_bounds__(64) void invert_small_kernel(const float* in, float* out, int n) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n) return;
    double a[D][2 * D];
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) { a[r][c] = (double)in[(size_t)i * D * D + r * D + c]; a[r][D + c] = r == c ? 1.0 : 0.0; }
#pragma unroll
    for (int col = 0; col < D; ++col) {
        int piv = col;
        double best = fabs(a[col][col]);
#pragma unroll
        for (int r = 0; r < D; ++r) if (r > col && fabs(a[r][col]) > best) { best = fabs(a[r][col]); piv = r; }
#pragma unroll
        for (int r = 0; r < D; ++r) if (r == piv && piv != col) {
#pragma unroll
            for (int c = 0; c < 2 * D; ++c) { const double t = a[col][c]; a[col][c] = a[r][c]; a[r][c] = t; }
        }
        const double inv = 1.0 / a[col][col];
#pragma unroll
        for (int c = 0; c < 2 * D; ++c) a[col][c] *= inv;
#pragma unroll
        for (int r = 0; r < D; ++r) if (r != col) {
            const double f = a[r][col];
#pragma unroll
            for (int c = 0; c < 2 * D; ++c) a[r][c] -= f * a[col][c];
        }
    }
#pragma unroll
    for (int r = 0; r < D; ++r)
#pragma unroll
        for (int c = 0; c < D; ++c) out[(size_t)i * D * D + r * D + c] = (float)a[r][D + c];
}


// ---------------------------------------------------------------------------------------------
// Channels-last spatial resize: mode 0 = nearest (F.interpolate default: src = floor(dst * in/out)),
// mode 1 = bilinear with align_corners=True (nn.Upsample(scale_factor=2, mode='bilinear', align_corners=True)).
// reference: nuscenes/cross_view_transformer/model/decoder.py:12,31.
template <typename T>
__global__ __launch_bounds__(256) void resize_kernel(const T* in, T* out, int N, int H, int W, int C, int Ho, int Wo,
                                                     int mode) {
    const int G = C >> 3;
    const long gid = ((long)blockIdx.x * 256 + threadIdx.x) / G;
    const int gl = threadIdx.x & (G - 1);
    if (gid >= (long)N * Ho * Wo) return;
    const int n = (int)(gid / (Ho * Wo)), rem = (int)(gid - (long)n * Ho * Wo);
    const int oh = rem / Wo, ow = rem - oh * Wo;
    const T* src = in + (size_t)n * H * W * C + gl * 8;
    float v[8];
    if (mode == 0) {
        const float sh = (float)H / (float)Ho, sw = (float)W / (float)Wo;
        const int ih = min((int)floorf(oh * sh), H - 1), iw = min((int)floorf(ow * sw), W - 1);
        load8<T>(src + ((size_t)ih * W + iw) * C, v);
    } else {
        const float sh = Ho > 1 ? (float)(H - 1) / (float)(Ho - 1) : 0.f;
        const float sw = Wo > 1 ? (float)(W - 1) / (float)(Wo - 1) : 0.f;
        const float fy = oh * sh, fx = ow * sw;
        const int y0 = (int)fy, x0 = (int)fx;
        const int y1 = min(y0 + 1, H - 1), x1 = min(x0 + 1, W - 1);
        const float ly = fy - (float)y0, lx = fx - (float)x0, hy = 1.f - ly, hx = 1.f - lx;
        float a[8], b[8], c[8], d[8];
        load8<T>(src + ((size_t)y0 * W + x0) * C, a);
        load8<T>(src + ((size_t)y0 * W + x1) * C, b);
        load8<T>(src + ((size_t)y1 * W + x0) * C, c);
        load8<T>(src + ((size_t)y1 * W + x1) * C, d);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = hy * (hx * a[e] + lx * b[e]) + ly * (hx * c[e] + lx * d[e]);
    }
    store8<T>(out + gid * C + gl * 8, v);
}

// Per-channel affine on a contiguous (N, C, HW) fp32 tensor: y = x * scale[c] + shift[c]
// (Normalize, nuscenes/cross_view_transformer/model/encoder_pyramid_axial.py:41-49).
__global__ __launch_bounds__(256) void channel_affine_kernel(const float* in, const float* scale, const float* shift,
                                                             float* out, long total, int C, long HW) {
    const long i = (long)blockIdx.x * 256 + threadIdx.x;
    if (i >= total) return;
    const int c = (int)((i / HW) % C);
    out[i] = in[i] * scale[c] + shift[c];
}

template <typename K, typename... Args>
static int launch1d(K kern, long work_items, hipStream_t stream, Args... args) {
    const long blocks = (work_items + 255) / 256;
    if (blocks <= 0 || blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), 0, stream, args...);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

static bool group_ok(int C) { const int G = C >> 3; return C % 8 == 0 && G >= 1 && G <= 64 && (G & (G - 1)) == 0; }

}  // namespace cobevt

using namespace cobevt;

extern "C" int cobevt_layernorm(const void* in, const float* gamma, const float* beta, void* out, int dtype, int rows,
                                int C, float eps, int navg, long avg_stride, long in_batch_stride, int rows_per_batch,
                                hipStream_t stream) {
    if (!in || !out || ((gamma == nullptr) != (beta == nullptr))) return COBEVT_ERR_ARG;
    if (!group_ok(C) || rows < 1 || navg < 1) return COBEVT_ERR_SHAPE;
    if (rows_per_batch <= 0) { rows_per_batch = rows; in_batch_stride = 0; }
    const long items = (long)rows * (C >> 3);
    if (dtype == 0) return launch1d(layernorm_kernel<bf16_t>, items, stream, (const bf16_t*)in, gamma, beta, (bf16_t*)out, rows, C, eps, navg, avg_stride, in_batch_stride, rows_per_batch);
    if (dtype == 1) return launch1d(layernorm_kernel<float>, items, stream, (const float*)in, gamma, beta, (float*)out, rows, C, eps, navg, avg_stride, in_batch_stride, rows_per_batch);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_fax_ray_embed(const float* I_inv, const float* E_inv, const float* image_plane, const float* w_img,
                                    const float* w_cam, void* out, int dtype, int BN, int hw, int D, hipStream_t stream) {
    if (!I_inv || !E_inv || !image_plane || !w_img || !w_cam || !out) return COBEVT_ERR_ARG;
    if (!group_ok(D) || BN < 1 || hw < 1) return COBEVT_ERR_SHAPE;
    if (BN > 65535) return COBEVT_ERR_SHAPE;
    const dim3 grid((hw + kRayPixPerBlock - 1) / kRayPixPerBlock, BN), block(256);
    if (dtype == 0) hipLaunchKernelGGL(ray_embed_kernel<bf16_t>, grid, block, 0, stream, I_inv, E_inv, image_plane, w_img, w_cam, (bf16_t*)out, BN, hw, D);
    else if (dtype == 1) hipLaunchKernelGGL(ray_embed_kernel<float>, grid, block, 0, stream, I_inv, E_inv, image_plane, w_img, w_cam, (float*)out, BN, hw, D);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_fax_bev_embed(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                                    const float* w_cam, const void* x, void* out, int dtype, int B, int n, int hw, int D,
                                    int x_bcast, hipStream_t stream) {
    if (!E_inv || !world || !w_bev || !b_bev || !w_cam || !x || !out) return COBEVT_ERR_ARG;
    if (!group_ok(D) || B < 1 || n < 1 || hw < 1) return COBEVT_ERR_SHAPE;
    if ((long)B * n > 65535) return COBEVT_ERR_SHAPE;
    const dim3 grid((hw + kEmbedPixPerBlock - 1) / kEmbedPixPerBlock, B * n), block(256);
    if (dtype == 0) hipLaunchKernelGGL(bev_embed_kernel<bf16_t>, grid, block, 0, stream, E_inv, world, w_bev, b_bev, w_cam, (const bf16_t*)x, (bf16_t*)out, B, n, hw, D, x_bcast);
    else if (dtype == 1) hipLaunchKernelGGL(bev_embed_kernel<float>, grid, block, 0, stream, E_inv, world, w_bev, b_bev, w_cam, (const float*)x, (float*)out, B, n, hw, D, x_bcast);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_agent_max(const void* in, void* out, int dtype, int B, int L, long per, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (B < 1 || L < 1 || per < 8 || per % 8) return COBEVT_ERR_SHAPE;
    const long items = (long)B * (per / 8);
    if (dtype == 0) return launch1d(agent_max_kernel<bf16_t>, items, stream, (const bf16_t*)in, (bf16_t*)out, B, L, per / 8);
    if (dtype == 1) return launch1d(agent_max_kernel<float>, items, stream, (const float*)in, (float*)out, B, L, per / 8);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_maxpool3x3s2(const void* in, void* out, int dtype, int N, int H, int W, int C, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (!group_ok(C) || N < 1 || H < 1 || W < 1) return COBEVT_ERR_SHAPE;
    const int Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
    const long items = (long)N * Ho * Wo * (C >> 3);
    if (dtype == 0) return launch1d(maxpool3x3s2_kernel<bf16_t>, items, stream, (const bf16_t*)in, (bf16_t*)out, N, H, W, C, Ho, Wo);
    if (dtype == 1) return launch1d(maxpool3x3s2_kernel<float>, items, stream, (const float*)in, (float*)out, N, H, W, C, Ho, Wo);
    return COBEVT_ERR_ARG;
}

// dtype codes: 0 bf16, 1 fp32
extern "C" int cobevt_to_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W,
                              const long* strides, hipStream_t stream) {
    if (!in || !out || !strides) return COBEVT_ERR_ARG;
    const long total = (long)N * C * H * W;
    if (total < 1) return COBEVT_ERR_SHAPE;
    const long sN = strides[0], sC = strides[1], sH = strides[2], sW = strides[3];
    if (in_dtype == 1 && out_dtype == 1) return launch1d(to_nhwc_kernel<float, float>, total, stream, (const float*)in, (float*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 1 && out_dtype == 0) return launch1d(to_nhwc_kernel<float, bf16_t>, total, stream, (const float*)in, (bf16_t*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 0 && out_dtype == 1) return launch1d(to_nhwc_kernel<bf16_t, float>, total, stream, (const bf16_t*)in, (float*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 0 && out_dtype == 0) return launch1d(to_nhwc_kernel<bf16_t, bf16_t>, total, stream, (const bf16_t*)in, (bf16_t*)out, N, C, H, W, sN, sC, sH, sW);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_from_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W,
                                const long* strides, hipStream_t stream) {
    if (!in || !out || !strides) return COBEVT_ERR_ARG;
    const long total = (long)N * C * H * W;
    if (total < 1) return COBEVT_ERR_SHAPE;
    const long sN = strides[0], sC = strides[1], sH = strides[2], sW = strides[3];
    if (in_dtype == 1 && out_dtype == 1) return launch1d(from_nhwc_kernel<float, float>, total, stream, (const float*)in, (float*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 1 && out_dtype == 0) return launch1d(from_nhwc_kernel<float, bf16_t>, total, stream, (const float*)in, (bf16_t*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 0 && out_dtype == 1) return launch1d(from_nhwc_kernel<bf16_t, float>, total, stream, (const bf16_t*)in, (float*)out, N, C, H, W, sN, sC, sH, sW);
    if (in_dtype == 0 && out_dtype == 0) return launch1d(from_nhwc_kernel<bf16_t, bf16_t>, total, stream, (const bf16_t*)in, (bf16_t*)out, N, C, H, W, sN, sC, sH, sW);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_regroup(const void* in, const int* record_len, void* out, float* mask, int dtype, int B, int max_cav,
                              long elems_per_agent, hipStream_t stream) {
    if (!in || !record_len || !out || !mask) return COBEVT_ERR_ARG;
    const int ch = dtype == 0 ? 8 : 4;
    if (B < 1 || max_cav < 1 || elems_per_agent < 1 || elems_per_agent % ch) return COBEVT_ERR_SHAPE;
    const long cpa = elems_per_agent / ch;
    const long items = (long)B * max_cav * cpa;
    if (dtype == 0) return launch1d(regroup_kernel<bf16_t>, items, stream, (const bf16_t*)in, record_len, (bf16_t*)out, mask, B, max_cav, cpa);
    if (dtype == 1) return launch1d(regroup_kernel<float>, items, stream, (const float*)in, record_len, (float*)out, mask, B, max_cav, cpa);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_sttf_warp(const void* x, const float* tmat, const float* cav_mask, void* out, float* com_mask,
                                const int* record_len, float* cav_out,
                                int dtype, int B, int L, int H, int W, int C, float discrete_ratio, float downsample_rate,
                                hipStream_t stream) {
    if (!x || !tmat || !out) return COBEVT_ERR_ARG;
    if (com_mask && !cav_mask && !record_len) return COBEVT_ERR_ARG;
    if (!group_ok(C) || B < 1 || L < 1 || H < 1 || W < 1 || B * L > 65535) return COBEVT_ERR_SHAPE;
    const long items = (long)H * W * (C >> 3);
    dim3 grid((unsigned)((items + 255) / 256), (unsigned)(B * L));
    if (dtype == 0) hipLaunchKernelGGL(sttf_warp_kernel<bf16_t>, grid, dim3(256), 0, stream, (const bf16_t*)x, tmat, cav_mask, (bf16_t*)out, com_mask, record_len, cav_out, B, L, H, W, C, discrete_ratio, downsample_rate);
    else if (dtype == 1) hipLaunchKernelGGL(sttf_warp_kernel<float>, grid, dim3(256), 0, stream, (const float*)x, tmat, cav_mask, (float*)out, com_mask, record_len, cav_out, B, L, H, W, C, discrete_ratio, downsample_rate);
    else return COBEVT_ERR_ARG;
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_invert_small(const float* in, float* out, int n, int dim, hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (n < 1 || (dim != 3 && dim != 4)) return COBEVT_ERR_SHAPE;
    const dim3 grid((n + 63) / 64), block(64);
    if (dim == 3) hipLaunchKernelGGL(invert_small_kernel<3>, grid, block, 0, stream, in, out, n);
    else hipLaunchKernelGGL(invert_small_kernel<4>, grid, block, 0, stream, in, out, n);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" int cobevt_resize_nhwc(const void* in, void* out, int dtype, int N, int H, int W, int C, int Ho, int Wo, int mode,
                                  hipStream_t stream) {
    if (!in || !out) return COBEVT_ERR_ARG;
    if (!group_ok(C) || N < 1 || H < 1 || W < 1 || Ho < 1 || Wo < 1 || (mode != 0 && mode != 1)) return COBEVT_ERR_SHAPE;
    const long items = (long)N * Ho * Wo * (C >> 3);
    if (dtype == 0) return launch1d(resize_kernel<bf16_t>, items, stream, (const bf16_t*)in, (bf16_t*)out, N, H, W, C, Ho, Wo, mode);
    if (dtype == 1) return launch1d(resize_kernel<float>, items, stream, (const float*)in, (float*)out, N, H, W, C, Ho, Wo, mode);
    return COBEVT_ERR_ARG;
}

extern "C" int cobevt_channel_affine(const float* in, const float* scale, const float* shift, float* out, long N, int C,
                                     long HW, hipStream_t stream) {
    if (!in || !scale || !shift || !out) return COBEVT_ERR_ARG;
    if (N < 1 || C < 1 || HW < 1) return COBEVT_ERR_SHAPE;
    return launch1d(channel_affine_kernel, N * C * HW, stream, in, scale, shift, out, N * C * HW, C, HW);
}

// Ingest inside the captured graph: a few workgroups pull the NEXT frame out of pinned (device-visible, fine-grained) host memory
// over PCIe while the step's kernels run.  The fetch waves need no LDS and 52 VGPRs, so they sit beside the convolution
// workgroups that own every CU's LDS; eight 8-byte loads per lane in flight (the captured step launches 16 workgroups x 256 lanes x
// 64 B = 256 KB in flight; 8 workgroups no longer fill the link, 32+ only add interference: profiles/r06_ingest_split_ab.txt).  No copy engine, no extra stream, no event between replays: see
// host.pipeline.HostFrameFeeder for why that matters (ROCm shares 4 hardware queues among a process's streams).
// Loads at SYSTEM scope (sc0 sc1: past the GPU's caches): the host rewrites a ring slot between two pulls of it, and a line of the
// previous frame must not be served from L2 - pinned memory from hipHostMalloc(default flags) is not guaranteed fine-grained.
__global__ __launch_bounds__(256) void host_fetch_kernel(const unsigned long long* __restrict__ src, unsigned long long* __restrict__ dst, long n8) {
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    const long stride = (long)gridDim.x * 256;
    long i = (long)blockIdx.x * 256 + threadIdx.x;
    constexpr int U = 8;
    for (; i + (U - 1) * stride < n8; i += U * stride) {
        unsigned long long v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) v[u] = __hip_atomic_load(src + i + u * stride, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
#pragma unroll
        for (int u = 0; u < U; ++u) dst[i + u * stride] = v[u];
    }
    for (; i < n8; i += stride) dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_host_fetch(const void* host_src, void* dst, long bytes, int blocks, hipStream_t stream) {
    if (!host_src || !dst) return COBEVT_ERR_ARG;
    if (bytes < 16 || bytes % 16 || ((size_t)host_src & 15) || ((size_t)dst & 15)) return COBEVT_ERR_SHAPE;
    if (blocks < 1) blocks = 128;
    if (blocks > 1024) blocks = 1024;
    hipLaunchKernelGGL(host_fetch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, (const unsigned long long*)host_src, (unsigned long long*)dst, bytes / 8);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}

extern "C" const char* cobevt_strerror(int code) {
    switch (code) {
        case COBEVT_OK: return "ok";
        case COBEVT_ERR_ARG: return "invalid argument (null pointer or bad enum)";
        case COBEVT_ERR_SHAPE: return "unsupported shape / alignment";
        case COBEVT_ERR_LAUNCH: return "HIP kernel launch failed";
        case COBEVT_ERR_UNSUPPORTED: return "combination not supported by this kernel";
        default: return "unknown cobevt error";
    }
}

extern "C" int cobevt_abi_version(void) { return 1; }
