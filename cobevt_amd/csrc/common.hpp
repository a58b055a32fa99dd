// Shared device helpers for the cobevt_amd HIP kernels (gfx950 / CDNA4 only).
//
// Two arithmetic modes share every kernel through one template parameter:
//   T = bf16_t : activations/weights stored bf16, v_mfma_f32_32x32x16_bf16, fp32 accumulate (perf mode)
//   T = float  : activations/weights stored fp32, v_mfma_f32_32x32x2_f32, exact fp32    (parity mode)
// Both modes address LDS tiles at byte level the same way: a "k-group" is 32 bytes per row
// (16 bf16 or 8 fp32); lane-half h = lane>>5 owns bytes [16h, 16h+16) of every k-group.  For fp32 the
// four floats of that 16-byte piece feed four 32x32x2 MFMAs, i.e. the contraction index is permuted
// identically on the A and B side, which leaves the sum unchanged.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace cobevt {

struct bf16_t { uint16_t bits; };

using f32x4 = __attribute__((ext_vector_type(4))) float;
using f32x16 = __attribute__((ext_vector_type(16))) float;
using bf16x8 = __attribute__((ext_vector_type(8))) __bf16;

__device__ __forceinline__ float bf2f(uint16_t b) { return __uint_as_float(((uint32_t)b) << 16); }

using bf16x2 = __attribute__((ext_vector_type(2))) __bf16;
using f32x2 = __attribute__((ext_vector_type(2))) float;

// fp32 pair -> packed bf16 pair with the hardware converter (v_cvt_pk_bf16_f32: round-to-nearest-even, the same
// rounding as torch.float32 -> torch.bfloat16); `lo` lands in bits 15:0.
__device__ __forceinline__ uint32_t pack_bf2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, bf16x2));
}

__device__ __forceinline__ uint16_t f2bf(float f) { return (uint16_t)(pack_bf2(f, 0.f) & 0xffffu); }

template <typename T> struct Elem;
template <> struct Elem<bf16_t> {
    static constexpr int kBytes = 2;
    static constexpr int kChunk = 8;   // elements per 16-byte chunk
    static constexpr bool kIsBf16 = true;
};
template <> struct Elem<float> {
    static constexpr int kBytes = 4;
    static constexpr int kChunk = 4;
    static constexpr bool kIsBf16 = false;
};

// A 16-byte chunk unpacked to fp32 lanes (8 values for bf16, 4 for fp32; unused tail left untouched).
template <typename T> __device__ __forceinline__ void chunk_to_f32(const uint4& c, float* v) {
    if constexpr (Elem<T>::kIsBf16) {
        v[0] = bf2f(c.x & 0xffff); v[1] = bf2f(c.x >> 16);
        v[2] = bf2f(c.y & 0xffff); v[3] = bf2f(c.y >> 16);
        v[4] = bf2f(c.z & 0xffff); v[5] = bf2f(c.z >> 16);
        v[6] = bf2f(c.w & 0xffff); v[7] = bf2f(c.w >> 16);
    } else {
        v[0] = __uint_as_float(c.x); v[1] = __uint_as_float(c.y);
        v[2] = __uint_as_float(c.z); v[3] = __uint_as_float(c.w);
    }
}
template <typename T> __device__ __forceinline__ uint4 f32_to_chunk(const float* v) {
    uint4 c;
    if constexpr (Elem<T>::kIsBf16) {
        c.x = pack_bf2(v[0], v[1]); c.y = pack_bf2(v[2], v[3]);
        c.z = pack_bf2(v[4], v[5]); c.w = pack_bf2(v[6], v[7]);
    } else {
        c.x = __float_as_uint(v[0]); c.y = __float_as_uint(v[1]);
        c.z = __float_as_uint(v[2]); c.w = __float_as_uint(v[3]);
    }
    return c;
}

template <typename T> __device__ __forceinline__ float load_elem(const T* p, size_t i);
template <> __device__ __forceinline__ float load_elem<bf16_t>(const bf16_t* p, size_t i) { return bf2f(p[i].bits); }
template <> __device__ __forceinline__ float load_elem<float>(const float* p, size_t i) { return p[i]; }
template <typename T> __device__ __forceinline__ void store_elem(T* p, size_t i, float v);
template <> __device__ __forceinline__ void store_elem<bf16_t>(bf16_t* p, size_t i, float v) { p[i].bits = f2bf(v); }
template <> __device__ __forceinline__ void store_elem<float>(float* p, size_t i, float v) { p[i] = v; }

// COBEVT_F32_SPLIT = 1 builds the SECOND library of the package (cobevt_amd/build.py: libcobevt_hip_f32s.so, same sources, same
// C ABI): every fp32-storage kernel of the inference path then takes its matrix products through the split-bf16 form below
// instead of v_mfma_f32_32x32x2_f32 - the strict-parity mode that is not 16x off the bf16 matrix rate.
#ifndef COBEVT_F32_SPLIT
#define COBEVT_F32_SPLIT 0
#endif

// (x, y) -> packed bf16 pairs (hi(x), hi(y)) and (lo(x), lo(y)): hi = round-to-nearest-even bf16, lo = bf16(x - hi) (x - hi is exact)
__device__ __forceinline__ void split_bf16_pair(float x, float y, uint32_t& hi, uint32_t& lo) {
    hi = pack_bf2(x, y);
    lo = pack_bf2(x - __uint_as_float(hi << 16), y - __uint_as_float(hi & 0xffff0000u));
}

// COBEVT_F32_SPLIT = 2 builds the THIRD library (libcobevt_hip_f32h.so; round 6): fp32 storage, ONE
// v_mfma_f32_32x32x16_f16 per 16-byte piece.  The activation-side piece x[0..3] goes in as an fp16 pair per value -
// hi = fp16(x), lo = fp16((x - hi) * 2^kF16LoShift): 22 significand bits - and the weight-side piece w[0..3] as ONE fp16 term, once
// at full scale against the hi slots and once scaled by 2^-kF16LoShift against the lo slots: sum_k (x_hi + x_lo) w_hi, i.e. the
// weights are rounded to 11 bits and nothing else is.  Half the matrix time of the split-bf16 form; only the ResNet encoder's
// convolutions are routed through this library (host/resnet_ms.py under set_compute_dtype("fp32_fast")): measured with the CPU
// oracle (tools/precision_emul.py, mode fp16_we) that costs 2.7-2.8e-4 max-rel / 1.0-1.5e-4 rms-rel on the 5-agent frame's logits,
// where fp16 weights in EVERY product cost 0.8-1.0e-3 and fp16 operands on both sides 1.6e-3 (the north-star's gate is 1e-3).
// The shift keeps the lo slots out of fp16's subnormal range for |x| >= 4e-3 (lo ~ 2^-12 |x|) whether or not the matrix pipe
// flushes subnormals; the scaled weight copy is normal for |w| >= 4e-3 and below that the lo term it would carry is < 2^-12 * 4e-3 |x|.
// Range: |x|, |w| <= 65504 (fp16) - the same precondition as the reference's own fp16 autocast (train_camera.py:157-160).
using f16x2 = __attribute__((ext_vector_type(2))) _Float16;
using f16x8 = __attribute__((ext_vector_type(8))) _Float16;
constexpr float kF16LoScale = 64.0f;             // 2^6
__device__ __forceinline__ uint32_t pack_h2(float lo, float hi) {
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, f16x2));
}
// activation-side piece -> {hi(x0,x1), hi(x2,x3), lo(x0,x1), lo(x2,x3)}
__device__ __forceinline__ uint4 split_f16_piece(const uint4& x) {
    const float x0 = __uint_as_float(x.x), x1 = __uint_as_float(x.y), x2 = __uint_as_float(x.z), x3 = __uint_as_float(x.w);
    const uint32_t h01 = pack_h2(x0, x1), h23 = pack_h2(x2, x3);
    const f16x2 a = __builtin_bit_cast(f16x2, h01), b = __builtin_bit_cast(f16x2, h23);
    const uint32_t l01 = pack_h2((x0 - (float)a[0]) * kF16LoScale, (x1 - (float)a[1]) * kF16LoScale);
    const uint32_t l23 = pack_h2((x2 - (float)b[0]) * kF16LoScale, (x3 - (float)b[1]) * kF16LoScale);
    return make_uint4(h01, h23, l01, l23);
}
// weight-side piece -> {w01, w23, w01 / 2^s, w23 / 2^s} as fp16
__device__ __forceinline__ uint4 dup_f16_piece(const uint4& w) {
    const uint32_t h01 = pack_h2(__uint_as_float(w.x), __uint_as_float(w.y)), h23 = pack_h2(__uint_as_float(w.z), __uint_as_float(w.w));
    const f16x2 inv = {(_Float16)(1.0f / kF16LoScale), (_Float16)(1.0f / kF16LoScale)};
    const f16x2 s01 = __builtin_bit_cast(f16x2, h01) * inv, s23 = __builtin_bit_cast(f16x2, h23) * inv;
    return make_uint4(h01, h23, __builtin_bit_cast(uint32_t, s01), __builtin_bit_cast(uint32_t, s23));
}

// One 32-byte k-group of a 32x32 MFMA tile.  `a` and `b` are the 16-byte pieces this lane read from
// row (lane&31) of the A tile and the B tile at byte offset 16*(lane>>5) of the k-group.
// C/D layout (both dtypes): col = lane&31 (B row), row = (r&3) + 8*(r>>2) + 4*(lane>>5) (A row).
// kWeightsFirst: which operand is the weight-like one (D = W . X^T kernels pass the weights as `a`) - only the fp16 form of the
// third library distinguishes the two sides.
template <typename T, bool kWeightsFirst = true>
__device__ __forceinline__ void mfma_kgroup(const uint4& a, const uint4& b, f32x16& acc) {
    if constexpr (Elem<T>::kIsBf16) {
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a),
                                                      __builtin_bit_cast(bf16x8, b), acc, 0, 0, 0);
    } else {
#if COBEVT_F32_SPLIT == 2
        const uint4 av = kWeightsFirst ? dup_f16_piece(a) : split_f16_piece(a);
        const uint4 bv = kWeightsFirst ? split_f16_piece(b) : dup_f16_piece(b);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, av), __builtin_bit_cast(f16x8, bv), acc, 0, 0, 0);
#elif COBEVT_F32_SPLIT
        // fp32 storage, split-bf16 matrix path (libcobevt_hip_f32s.so): x = hi + lo with hi = bf16(x), lo = bf16(x - hi),
        // |x - hi - lo| <= 2^-17 |x|.  The bf16 instruction contracts 8 element pairs per lane half where the piece holds 4
        // values, so A carries {hi[0..3], lo[0..3]} and B {hi[0..3], hi[0..3]} / {lo[0..3], lo[0..3]}: two
        // v_mfma_f32_32x32x16_bf16 (64 matrix-pipe cycles) add all FOUR cross terms (hi+lo)(hi+lo) of the 8 products that the
        // four v_mfma_f32_32x32x2_f32 (256 cycles) form exactly; bf16 x bf16 products are exact in the fp32 accumulator.
        uint32_t ah[2], al[2], bh[2], bl[2];
        split_bf16_pair(__uint_as_float(a.x), __uint_as_float(a.y), ah[0], al[0]);
        split_bf16_pair(__uint_as_float(a.z), __uint_as_float(a.w), ah[1], al[1]);
        split_bf16_pair(__uint_as_float(b.x), __uint_as_float(b.y), bh[0], bl[0]);
        split_bf16_pair(__uint_as_float(b.z), __uint_as_float(b.w), bh[1], bl[1]);
        const uint4 av = make_uint4(ah[0], ah[1], al[0], al[1]);
        const uint4 bhv = make_uint4(bh[0], bh[1], bh[0], bh[1]);
        const uint4 blv = make_uint4(bl[0], bl[1], bl[0], bl[1]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, bhv), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, av), __builtin_bit_cast(bf16x8, blv), acc, 0, 0, 0);
#else
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.x), __uint_as_float(b.x), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.y), __uint_as_float(b.y), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.z), __uint_as_float(b.z), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(__uint_as_float(a.w), __uint_as_float(b.w), acc, 0, 0, 0);
#endif
    }
}

// Activation pieces split ONCE, when a kernel stages them into an LDS patch that only feeds MFMA operand reads (fp32 storage in the
// second and third library): the in-loop forms above spend 12-14 VALU instructions per MFMA on the split, in a dependent chain between
// the LDS read and the MFMA (third library, 256 -> 256 on 20 x 32 x 32: 77 us in-loop, 54.7 us staged), while a patch piece is read by
// 9 taps x every cout tile of the workgroup.  Staged form of a piece {x0..x3}: {hi(x0,x1), hi(x2,x3), lo(x0,x1), lo(x2,x3)} - bf16 halves
// in the second library (x = hi + lo to 2^-17), fp16 halves with the scaled lo in the third.  stage_x_piece is the identity in the native
// library and for bf16 storage, and mfma_kgroup_xs is mfma_kgroup<T> (weights first) there.
template <typename T> constexpr bool kXSplit = (COBEVT_F32_SPLIT != 0) && !Elem<T>::kIsBf16;
__device__ __forceinline__ void split_pair_staged(float x, float y, uint32_t& hi, uint32_t& lo) {
#if COBEVT_F32_SPLIT == 2
    hi = pack_h2(x, y);
    const f16x2 hv = __builtin_bit_cast(f16x2, hi);
    lo = pack_h2((x - (float)hv[0]) * kF16LoScale, (y - (float)hv[1]) * kF16LoScale);
#else
    split_bf16_pair(x, y, hi, lo);
#endif
}
template <typename T> __device__ __forceinline__ uint4 stage_x_piece(const uint4& x) {
    if constexpr (kXSplit<T>) {
        uint32_t h01, h23, l01, l23;
        split_pair_staged(__uint_as_float(x.x), __uint_as_float(x.y), h01, l01);
        split_pair_staged(__uint_as_float(x.z), __uint_as_float(x.w), h23, l23);
        return make_uint4(h01, h23, l01, l23);
    } else {
        return x;
    }
}
// w = the raw fp32 weight piece (a register fragment shared by the MT strips / pixel tiles of the wave: its split is amortised), xs = staged
template <typename T> __device__ __forceinline__ void mfma_kgroup_xs(const uint4& w, const uint4& xs, f32x16& acc) {
    if constexpr (kXSplit<T>) {
#if COBEVT_F32_SPLIT == 2
        const uint4 wv = dup_f16_piece(w);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, wv), __builtin_bit_cast(f16x8, xs), acc, 0, 0, 0);
#else
        // (w_hi, w_hi) . (x_hi, x_lo) + (w_lo, w_lo) . (x_hi, x_lo): the same four cross terms as mfma_kgroup's split form
        uint32_t h01, h23, l01, l23;
        split_bf16_pair(__uint_as_float(w.x), __uint_as_float(w.y), h01, l01);
        split_bf16_pair(__uint_as_float(w.z), __uint_as_float(w.w), h23, l23);
        const uint4 wh = make_uint4(h01, h23, h01, h23), wl = make_uint4(l01, l23, l01, l23);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xs), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, xs), acc, 0, 0, 0);
#endif
    } else {
        mfma_kgroup<T>(w, xs, acc);
    }
}

// Both operands staged, weights first: a kernel that also keeps its WEIGHTS in LDS (the stem) converts them once with stage_w_piece - the
// duplicated form in the third library, the (hi, lo) form in the second, whose second product takes the weight piece with its halves swapped.
template <typename T> __device__ __forceinline__ uint4 stage_w_piece(const uint4& w) {
    if constexpr (kXSplit<T>) {
#if COBEVT_F32_SPLIT == 2
        return dup_f16_piece(w);
#else
        return stage_x_piece<T>(w);          // {hi01, hi23, lo01, lo23}: the second product of mfma_kgroup_staged takes it with the halves swapped
#endif
    } else {
        return w;
    }
}
template <typename T> __device__ __forceinline__ void mfma_kgroup_staged(const uint4& ws, const uint4& xs, f32x16& acc) {
    if constexpr (kXSplit<T>) {
#if COBEVT_F32_SPLIT == 2
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, ws), __builtin_bit_cast(f16x8, xs), acc, 0, 0, 0);
#else
        const uint4 wsw = make_uint4(ws.z, ws.w, ws.x, ws.y);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, ws), __builtin_bit_cast(bf16x8, xs), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wsw), __builtin_bit_cast(bf16x8, xs), acc, 0, 0, 0);
#endif
    } else {
        mfma_kgroup<T>(ws, xs, acc);
    }
}

// BOTH operands staged through LDS (the LDS-staged 3x3 kernel, the 128 x 128 dense-row kernel, the implicit GEMM): activations first.
// Second library: both tiles hold the {hi01, hi23, lo01, lo23} form; x . w = (x_hi, x_lo) . (w_hi, w_lo) + (x_hi, x_lo) . (w_lo, w_hi) -
// the second product takes the weight piece with its halves swapped (four register moves), so the loop has no conversion at all where the
// in-loop split spent 36 VALU instructions per two MFMA pairs.  Third library: x staged, w in the duplicated form, one fp16 MFMA.
template <typename T> __device__ __forceinline__ uint4 stage_ws_piece(const uint4& w) {
    if constexpr (kXSplit<T>) {
#if COBEVT_F32_SPLIT == 2
        return dup_f16_piece(w);
#else
        return stage_x_piece<T>(w);
#endif
    } else {
        return w;
    }
}
template <typename T> __device__ __forceinline__ void mfma_kgroup_ss(const uint4& xs, const uint4& ws, f32x16& acc) {
    if constexpr (kXSplit<T>) {
#if COBEVT_F32_SPLIT == 2
        acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, xs), __builtin_bit_cast(f16x8, ws), acc, 0, 0, 0);
#else
        const uint4 wsw = make_uint4(ws.z, ws.w, ws.x, ws.y);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xs), __builtin_bit_cast(bf16x8, ws), acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, xs), __builtin_bit_cast(bf16x8, wsw), acc, 0, 0, 0);
#endif
    } else {
        mfma_kgroup<T, false>(xs, ws, acc);
    }
}

// PACKED fp16 form (third library, round 6 second step): tests/precision_emul.py mode fp16_e2 - BOTH operands of the ResNet encoder's
// convolutions rounded to fp16 - measures the same 2.7-2.8e-4 on the 5-agent frame as fp16 weights alone (the weights' rounding is what is
// seen; post-ReLU activations add nothing measurable), so the (hi, lo) pair of the activations buys nothing there and its four MFMA slots can
// carry a second k-group instead: a staged patch keeps ONE fp16 per activation, the 16-byte operand of lane half h is
// {k-group 2q: channels 4h .. 4h + 3, k-group 2q + 1: channels 4h .. 4h + 3}, and a pair of weight fragments is packed the same way -
// one v_mfma_f32_32x32x16_f16 per TWO k-groups, the bf16 kernels' matrix time on fp32 storage.  Kernels whose wave owns an even number of
// k-groups per tap use it (kXPack); the others keep the (hi, lo) form above.
template <typename T, int KGW> constexpr bool kXPack = (COBEVT_F32_SPLIT == 2) && !Elem<T>::kIsBf16 && (KGW % 2 == 0);
__device__ __forceinline__ uint2 pack_f16_hi(const uint4& x) {
    return make_uint2(pack_h2(__uint_as_float(x.x), __uint_as_float(x.y)), pack_h2(__uint_as_float(x.z), __uint_as_float(x.w)));
}
__device__ __forceinline__ uint4 pack_f16_pair(const uint4& w0, const uint4& w1) {
    const uint2 a = pack_f16_hi(w0), b = pack_f16_hi(w1);
    return make_uint4(a.x, a.y, b.x, b.y);
}
// byte offset of 16-byte piece j = 2 * kgroup + half of a 128-byte channel chunk inside the packed image of that chunk (its 8 bytes)
__device__ __forceinline__ constexpr int packed_piece_offset(int j) { return (j >> 2) * 32 + (j & 1) * 16 + ((j >> 1) & 1) * 8; }
__device__ __forceinline__ void mfma_f16_packed(const uint4& w, const uint4& x, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, w), __builtin_bit_cast(f16x8, x), acc, 0, 0, 0);
}

// THREE-TERM form on k-group pairs (second library): where a wave owns an even number of k-groups per tap, the staged
// patch keeps the hi halves and the lo halves of a PAIR of k-groups as two separate 16-byte operands per lane half - {hi(2q): 4 values,
// hi(2q + 1): 4 values} and the same for lo - and a pair of weight fragments is split the same way: w_hi . x_hi + w_hi . x_lo + w_lo . x_hi,
// THREE v_mfma_f32_32x32x16_bf16 per two k-groups instead of four.  The dropped w_lo . x_lo term is 2^-16 of a product, the same order as
// the operands' own 2^-17 residuals: ~2e-5 end to end instead of 1e-5.
template <typename T, int KGW> constexpr bool kXPack3 = (COBEVT_F32_SPLIT == 1) && !Elem<T>::kIsBf16 && (KGW % 2 == 0);
// byte offset of the hi half (8 bytes; the lo half sits 16 bytes further) of piece j = 2 * kgroup + half inside the image of a 128-byte chunk
__device__ __forceinline__ constexpr int packed3_piece_offset(int j) { return (j >> 2) * 64 + (j & 1) * 32 + ((j >> 1) & 1) * 8; }
__device__ __forceinline__ void store_piece_packed3(unsigned char* dst, const uint4& x, bool zero) {
    uint32_t h01, h23, l01, l23;
    split_bf16_pair(__uint_as_float(x.x), __uint_as_float(x.y), h01, l01);
    split_bf16_pair(__uint_as_float(x.z), __uint_as_float(x.w), h23, l23);
    *(uint2*)dst = zero ? make_uint2(0, 0) : make_uint2(h01, h23);
    *(uint2*)(dst + 16) = zero ? make_uint2(0, 0) : make_uint2(l01, l23);
}
__device__ __forceinline__ void split_w_pair(const uint4& w0, const uint4& w1, uint4& wh, uint4& wl) {
    uint32_t h[4], l[4];
    split_bf16_pair(__uint_as_float(w0.x), __uint_as_float(w0.y), h[0], l[0]);
    split_bf16_pair(__uint_as_float(w0.z), __uint_as_float(w0.w), h[1], l[1]);
    split_bf16_pair(__uint_as_float(w1.x), __uint_as_float(w1.y), h[2], l[2]);
    split_bf16_pair(__uint_as_float(w1.z), __uint_as_float(w1.w), h[3], l[3]);
    wh = make_uint4(h[0], h[1], h[2], h[3]);
    wl = make_uint4(l[0], l[1], l[2], l[3]);
}
__device__ __forceinline__ void mfma_3term(const uint4& wh, const uint4& wl, const uint4& xh, const uint4& xl, f32x16& acc) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wh), __builtin_bit_cast(bf16x8, xl), acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, wl), __builtin_bit_cast(bf16x8, xh), acc, 0, 0, 0);
}

// accumulator register r of the 32x32 C/D fragment -> row within the tile
__device__ __forceinline__ int acc_row(int r, int lane) { return (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }

// erf by Abramowitz & Stegun 7.1.26 (|error| <= 1.5e-7, below fp32 epilogue noise): one v_exp + one v_rcp + a degree-5
// polynomial instead of libm's multi-range erff (~40 instructions), which dominated the GELU epilogues.
__device__ __forceinline__ float erf_as(float x) {
    const float ax = fabsf(x);
    const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, ax, 1.0f));
    float poly = fmaf(1.061405429f, t, -1.453152027f);
    poly = fmaf(poly, t, 1.421413741f);
    poly = fmaf(poly, t, -0.284496736f);
    poly = fmaf(poly, t, 0.254829592f);
    const float e = __builtin_amdgcn_exp2f(-ax * ax * 1.4426950408889634f);
    const float y = fmaf(-poly * t, e, 1.0f);
    return copysignf(y, x);
}

// exact (erf) GELU of the reference: nn.GELU() default
__device__ __forceinline__ float gelu_erf(float x) { return 0.5f * x * (1.0f + erf_as(x * 0.70710678118654752440f)); }

// GELU of the bf16 kernels (fax_modules.py:309-313, base_transformer.py:108): x Phi(x) as x * sigmoid(x (p0 + p1 x^2 + p2 x^4))
// with (p0, p1, p2) = (1.59433946, 0.0745210325, -7.69554552e-4) fitted to the exact x Phi(x) (the tanh form's 0.044715 cubic
// plus a quintic term): |error| <= 6.2e-5 for x >= -8 (below that the result is x * 7e-12 instead of a value tending to -0: < 1e-9 in magnitude down to x = -100) and <= 0.38 of the bf16 rounding error 2^-9 max(|gelu(x)|, 0.01) the result
// meets next - it is stored as bf16 in every kernel that calls this.  Seven VALU instructions (one v_exp_f32, one v_rcp_f32)
// against ~25 for the A&S erf form, which was 3.2k of the 5.5k VALU instructions of a 32-row block of the level-0 row chain.
// The polynomial turns over beyond |x| ~ 8.2, so its argument is clamped to [-8, 8] (sigmoid there: 1 - 7e-12 / 7e-12);
// the coefficients below carry the factor -log2(e) of exp(-z) = exp2(-z log2 e).  fp32 mode keeps gelu_erf.
__device__ __forceinline__ float gelu_bf16(float x) {
    const float xc = __builtin_amdgcn_fmed3f(x, -8.0f, 8.0f);
    const float u = xc * xc;
    float w = fmaf(u, 0.0011102325515821576f, -0.10751112550497055f);
    w = fmaf(w, u, -2.3001456260681152f);
    const float e = __builtin_amdgcn_exp2f(xc * w);
    return x * __builtin_amdgcn_rcpf(1.0f + e);
}
#ifdef COBEVT_GELU_EXACT          // A/B builds only (tools/build_variant.py): the erf form in the bf16 kernels too.  In front of gelu_t,
#define gelu_bf16 gelu_erf        // so that apply_act<T> (conv3x3, gemm_rows, gemm_rows3, igemm epilogues) switches with the direct callers
#endif
template <typename T> __device__ __forceinline__ float gelu_t(float x) {
    if constexpr (sizeof(T) == 2) return gelu_bf16(x);
    else return gelu_erf(x);
}

// epilogue activations by code: 0 none, 1 ReLU, 2 exact GELU, 3 swish x * sigmoid(x) (EfficientNet MBConv), 4 sigmoid
__device__ __forceinline__ float sigmoid_f(float x) { return __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.4426950408889634f * x)); }
template <typename T = float> __device__ __forceinline__ float apply_act(float x, int act) {
    if (act == 1) return fmaxf(x, 0.f);
    if (act == 2) return gelu_t<T>(x);
    if (act == 3) return x * sigmoid_f(x);
    if (act == 4) return sigmoid_f(x);
    return x;
}

__device__ __forceinline__ float wave_sum_xor(float v, int width) {
    for (int o = width >> 1; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) is a PER-DEVICE attribute: a launch site keeps one of these (static) and
// raises the limit once per device the process launches on, not once per process.
struct PerDeviceOnce {
    unsigned long long done = 0;
    bool first() {
        int d = 0;
        if (hipGetDevice(&d) != hipSuccess || d < 0 || d > 63) return true;
        const unsigned long long bit = 1ull << d;
        if (done & bit) return false;
        done |= bit;
        return true;
    }
};

}  // namespace cobevt

// error codes shared with include/cobevt_hip.h
#define COBEVT_OK 0
#define COBEVT_ERR_ARG 1
#define COBEVT_ERR_SHAPE 2
#define COBEVT_ERR_LAUNCH 3
#define COBEVT_ERR_UNSUPPORTED 4
