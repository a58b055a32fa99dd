// Fused torchvision Bottleneck (inplanes 128, planes 32, stride 1, no downsample) for fp32 STORAGE (gfx950; round 6):
//
//     out = ReLU( conv1x1_3( ReLU( conv3x3_2( ReLU( conv1x1_1(x) + b1 ) ) + b2 ) ) + b3 + x )     (eval BatchNorms folded)
//
// = ResNetBottleNeck(dim) of the FAX pyramid (reference opv2v/opencood/models/sub_modules/fax_modules.py:10,472,512, two per level on the
// (b, 128, H, W) BEV map, with torchvision.models.resnet.Bottleneck.forward) - the fp32 counterpart of bottleneck.hip.  The fp32 modes ran a
// block as three launches (dense rows 128 -> 32, the 3x3 kernel 32 -> 32, dense rows 32 -> 128 + residual): ~0.3 ms per frame over the six
// blocks, every launch latency-bound on the small maps and the two 32-channel intermediates round-tripping through memory.  One launch:
// a 4-wave workgroup owns an 8 x 16 output tile;
//   1. conv1 on the 10 x 18 region conv2 needs (six 32-pixel MFMA tiles over the four waves; the B operand - a pixel's 128 channels -
//      straight from global memory), + b1, ReLU, into an LDS patch [region pixel][32 channels] (zeros outside the image = conv2's padding);
//      when the producer of x already computed conv1 (the row chain's `next` projection, ops.attn_mlp_chain) the patch is filled from y1;
//   2. conv2: a wave owns one 32-pixel output tile (2 rows x 16 columns): 9 taps x 4 k-groups out of that patch, + b2, ReLU, into a
//      second LDS tile;
//   3. conv3 per 32-cout tile out of that tile; + b3 staged through LDS 64 channels at a time, then + residual + ReLU and 16-byte
//      coalesced stores (the residual read coalesced as well).
// Weights arrive as MFMA fragments straight from L2 (the tables of the separate launches: cobevt_linear_rows_small_k's for the 1x1s,
// cobevt_conv3x3_wfrag_nhwc's for the 3x3).  LDS tiles that only feed MFMA operands are written through stage_x_piece (common.hpp): the
// matrix path is the library's - exact fp32 MFMA or split-bf16.  53 KB per workgroup, three per CU.
#include "common.hpp"

namespace cobevt {

namespace {

constexpr int kTH = 8, kTW = 16, kR1H = kTH + 2, kR1W = kTW + 2, kR1 = kR1H * kR1W;      // 180 region pixels
constexpr int kN1 = (kR1 + 31) / 32;                                                     // 6 conv1 pixel tiles
constexpr int kPix = 128 + 16;                    // bytes per patch pixel: 32 fp32 + pad (9 sixteen-byte slots: odd -> conflict-free)
constexpr int kP1 = kR1 * kPix;                   // 25,920
constexpr int kP2 = kTH * kTW * kPix;             // 18,432
constexpr int kSRow = 256 + 16;                   // staging row: 64 fp32 + pad
constexpr int kStage = kTH * kTW * kSRow;         // 34,816
constexpr int kLds = kP2 + (kP1 > kStage ? kP1 : kStage) + 192 * 4;
constexpr int kThreads = 256;

struct BnF32Params {
    const float* in;        // (N, H, W, 128)
    const float* y1;        // (N, H, W, 32) = ReLU(conv1(x) + b1) computed by the producer, or null
    const uint4* w1;        // [16 k-groups][64 lanes]             A fragments of W1 (32 x 128)
    const uint4* w2;        // [9 taps][4 k-groups][64 lanes]      A fragments of W2 (32 x 32 per tap)
    const uint4* w3;        // [4 cout tiles][4 k-groups][64]      A fragments of W3 (128 x 32); tile stride w3_tile uint4
    const float* b1; const float* b2; const float* b3;
    float* out;
    int N, H, W, tiles_y, tiles_x;
    long w3_tile;
};

__global__ __launch_bounds__(kThreads, 2) void bottleneck_f32_kernel(BnF32Params p) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned char* P2 = smem;
    unsigned char* P1 = smem + kP2;                       // later the staging tile
    float* sb = (float*)(smem + kP2 + (kP1 > kStage ? kP1 : kStage));   // b1[32] | b2[32] | b3[128]

    int logical;
    {   // consecutive tiles on one XCD (shared halo rows in its L2)
        const int nblk = gridDim.x, bid = blockIdx.x, xcd = bid & 7, slot = bid >> 3;
        const int q = nblk >> 3, r = nblk & 7;
        logical = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int tx = logical % p.tiles_x;
    const int ty = (logical / p.tiles_x) % p.tiles_y;
    const int img = logical / (p.tiles_x * p.tiles_y);
    const int oy0 = ty * kTH, ox0 = tx * kTW;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int h = lane >> 5, ql = lane & 31;

    if (tid < 192) sb[tid] = tid < 32 ? p.b1[tid] : (tid < 64 ? p.b2[tid - 32] : p.b3[tid - 64]);

    // ---- phase 1: the conv1 patch [180 region pixels][32 channels]
    if (p.y1) {
        // the producer's conv1 output: 8 sixteen-byte pieces per region pixel, zeros outside the image
        // (every piece of the thread requested before the first is written: as a rolled load / store loop the six iterations were six
        //  serialised round trips; unconditional loads from a clamped pixel, zeroed afterwards)
        constexpr int NIT = (kR1 * 8 + kThreads - 1) / kThreads;
        uint4 v[NIT];
        bool ok[NIT];
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int item = tid + u * kThreads, ic = item < kR1 * 8 ? item : kR1 * 8 - 1;
            const int r = ic >> 3, j = ic & 7;
            const int ry = r / kR1W, rx = r - ry * kR1W;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            ok[u] = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            v[u] = *(const uint4*)(p.y1 + (((size_t)img * p.H + (ok[u] ? iy : 0)) * p.W + (ok[u] ? ix : 0)) * 32 + j * 4);
        }
#pragma unroll
        for (int u = 0; u < NIT; ++u) {
            const int item = tid + u * kThreads;
            if (item < kR1 * 8) *(uint4*)(P1 + (item >> 3) * kPix + (item & 7) * 16) = ok[u] ? stage_x_piece<float>(v[u]) : make_uint4(0, 0, 0, 0);
        }
        __syncthreads();
    } else {
        __syncthreads();                                  // the bias table
#pragma unroll 1
        for (int tile = wave; tile < kN1; tile += 4) {
            int r = tile * 32 + ql;
            const bool live = r < kR1;
            if (!live) r = kR1 - 1;                       // padding lanes of the last tile compute a duplicate
            const int ry = r / kR1W, rx = r - ry * kR1W;
            const int iy = oy0 - 1 + ry, ix = ox0 - 1 + rx;
            const bool inside = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            const float* row = p.in + (inside ? (((size_t)img * p.H + iy) * p.W + ix) * 128 : 0) + h * 4;    // clamped: unconditional loads
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
#pragma unroll
            for (int half = 0; half < 2; ++half) {        // eight k-groups of operands in flight at a time
                uint4 xb[8], wf[8];
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    xb[g] = *(const uint4*)(row + (half * 8 + g) * 8);
                    wf[g] = p.w1[(half * 8 + g) * 64 + lane];
                }
#pragma unroll
                for (int g = 0; g < 8; ++g) mfma_kgroup<float>(wf[g], xb[g], acc);     // D = W1 . X^T: lane <-> pixel, registers <-> mid channels
            }
            if (live) {
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const float4 b = *(const float4*)(sb + 8 * k + 4 * h);
                    float v[4] = {acc[4 * k] + b.x, acc[4 * k + 1] + b.y, acc[4 * k + 2] + b.z, acc[4 * k + 3] + b.w};
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = inside ? fmaxf(v[e], 0.f) : 0.f;
                    *(uint4*)(P1 + r * kPix + (2 * k + h) * 16) =
                        stage_x_piece<float>(make_uint4(__float_as_uint(v[0]), __float_as_uint(v[1]), __float_as_uint(v[2]), __float_as_uint(v[3])));
                }
            }
        }
        __syncthreads();
    }

    // ---- phase 2: conv2 on this wave's 32-pixel output tile (rows 2 wave, 2 wave + 1)
    const int opx = wave * 32 + ql;                       // output pixel of this lane inside the 8 x 16 tile
    {
        f32x16 acc;
#pragma unroll
        for (int q = 0; q < 16; ++q) acc[q] = 0.f;
        const unsigned char* base = P1 + ((2 * wave + (ql >> 4)) * kR1W + (ql & 15)) * kPix + h * 16;
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            uint4 wf[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wf[g] = p.w2[(tap * 4 + g) * 64 + lane];
            const unsigned char* pt = base + ((tap / 3) * kR1W + (tap % 3)) * kPix;
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup_xs<float>(wf[g], *(const uint4*)(pt + g * 32), acc);
        }
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float4 b = *(const float4*)(sb + 32 + 8 * k + 4 * h);
            const float v0 = fmaxf(acc[4 * k] + b.x, 0.f), v1 = fmaxf(acc[4 * k + 1] + b.y, 0.f);
            const float v2 = fmaxf(acc[4 * k + 2] + b.z, 0.f), v3 = fmaxf(acc[4 * k + 3] + b.w, 0.f);
            *(uint4*)(P2 + opx * kPix + (2 * k + h) * 16) =
                stage_x_piece<float>(make_uint4(__float_as_uint(v0), __float_as_uint(v1), __float_as_uint(v2), __float_as_uint(v3)));
        }
    }
    __syncthreads();                                      // P2 complete; P1 is dead: the staging tile takes its place

    // ---- phase 3: conv3, 64 output channels at a time through the staging tile, + b3 + residual, ReLU, coalesced stores
    float* stage = (float*)P1;
    constexpr int SROW = kSRow / 4;
    uint4 xf[4];
#pragma unroll
    for (int g = 0; g < 4; ++g) xf[g] = *(const uint4*)(P2 + opx * kPix + h * 16 + g * 32);
#pragma unroll 1
    for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
        for (int c = 0; c < 2; ++c) {
            const int ct = 2 * hh + c;
            f32x16 acc;
#pragma unroll
            for (int q = 0; q < 16; ++q) acc[q] = 0.f;
            uint4 wf[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) wf[g] = p.w3[(size_t)ct * p.w3_tile + g * 64 + lane];
#pragma unroll
            for (int g = 0; g < 4; ++g) mfma_kgroup_xs<float>(wf[g], xf[g], acc);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const float4 b = *(const float4*)(sb + 64 + ct * 32 + 8 * k + 4 * h);
                *(float4*)(stage + opx * SROW + c * 32 + 8 * k + 4 * h) =
                    make_float4(acc[4 * k] + b.x, acc[4 * k + 1] + b.y, acc[4 * k + 2] + b.z, acc[4 * k + 3] + b.w);
            }
        }
        __syncthreads();
#pragma unroll
        for (int it = 0; it < kTH * kTW * 16 / kThreads; ++it) {          // 128 pixels x 16 float4 of this 64-channel half
            const int item = tid + it * kThreads;
            const int px = item >> 4, c4 = item & 15;
            const int oy = oy0 + (px >> 4), ox = ox0 + (px & 15);
            if (oy < p.H && ox < p.W) {
                const size_t o = (((size_t)img * p.H + oy) * p.W + ox) * 128 + hh * 64 + c4 * 4;
                const float4 r = *(const float4*)(p.in + o);
                const float4 v = *(const float4*)(stage + px * SROW + c4 * 4);
                *(float4*)(p.out + o) = make_float4(fmaxf(v.x + r.x, 0.f), fmaxf(v.y + r.y, 0.f), fmaxf(v.z + r.z, 0.f), fmaxf(v.w + r.w, 0.f));
            }
        }
        if (hh == 0) __syncthreads();                     // the staging tile is rewritten by the second half
    }
}

}  // namespace

}  // namespace cobevt

using namespace cobevt;

// C-ABI entry point, see include/cobevt_hip.h
extern "C" int cobevt_bottleneck_f32_nhwc(const void* in, const void* y1, const void* w1frag, const float* b1, const void* w2frag,
                                          const float* b2, const void* w3frag, const float* b3, void* out, const int* dims,
                                          hipStream_t stream) {
    // dims: [N, H, W, w3 tile stride in 16-byte units]
    if (!in || !w1frag || !w2frag || !w3frag || !b1 || !b2 || !b3 || !out || !dims) return COBEVT_ERR_ARG;
    BnF32Params p;
    p.in = (const float*)in; p.y1 = (const float*)y1; p.w1 = (const uint4*)w1frag; p.w2 = (const uint4*)w2frag; p.w3 = (const uint4*)w3frag;
    p.b1 = b1; p.b2 = b2; p.b3 = b3; p.out = (float*)out;
    p.N = dims[0]; p.H = dims[1]; p.W = dims[2]; p.w3_tile = dims[3];
    if (p.N < 1 || p.H < 1 || p.W < 1 || p.w3_tile < 4 * 64) return COBEVT_ERR_SHAPE;
    p.tiles_y = (p.H + kTH - 1) / kTH;
    p.tiles_x = (p.W + kTW - 1) / kTW;
    const long blocks = (long)p.N * p.tiles_y * p.tiles_x;
    if (blocks > 0x7fffffffL) return COBEVT_ERR_SHAPE;
    static cobevt::PerDeviceOnce attr_once;
    if (attr_once.first()) (void)hipFuncSetAttribute((const void*)bottleneck_f32_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, kLds);
    hipLaunchKernelGGL(bottleneck_f32_kernel, dim3((unsigned)blocks), dim3(kThreads), kLds, stream, p);
    return hipGetLastError() == hipSuccess ? COBEVT_OK : COBEVT_ERR_LAUNCH;
}
