/*
 * cobevt_hip.h — C ABI of libcobevt_hip.so: the MI355X (gfx950) kernels behind the CoBEVT FAX hot path.
 *
 * The reference (DerrickXuNu/CoBEVT) has no FFI: its hot path is Python nn.Modules calling ATen.  The
 * drop-in boundary is therefore the nn.Module API (cobevt_amd/host/*, same class names / constructor
 * arguments / state_dict keys / forward contracts); those modules bind the entry points below through
 * ctypes (cobevt_amd/lib.py).  Every entry point cites the reference code whose arithmetic it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - plain C types only: device pointers (hipMalloc'ed / torch data_ptr()), ints, floats, a hipStream_t.
 *   - all work is enqueued asynchronously on `stream`; nothing is allocated, nothing synchronises.
 *   - return value: 0 = COBEVT_OK, otherwise an error code (see cobevt_strerror); on error nothing was launched.
 *   - dtype code: 0 = bf16 storage + bf16 MFMA + fp32 accumulate ; 1 = fp32 storage (parity modes); the matrix path of
 *     dtype 1 is a property of the LIBRARY - the same sources and the same ABI are built three times (cobevt_amd/build.py):
 *       libcobevt_hip.so       exact v_mfma_f32_32x32x2_f32
 *       libcobevt_hip_f32s.so  every product as two v_mfma_f32_32x32x16_bf16 over (hi, lo) bf16 halves (-DCOBEVT_F32_SPLIT=1; ~1e-5 end to end)
 *       libcobevt_hip_f32h.so  v_mfma_f32_32x32x16_f16 with fp16 operands, the weight-side operand as ONE fp16 term (-DCOBEVT_F32_SPLIT=2):
 *                              the ResNet encoder's library under host.set_compute_dtype("fp32_fast"), not meant for any other launch
 *     (csrc/common.hpp; dtype 0 is identical in all three).
 *   - activations are channels-last ("NHWC", token-major); weights are [Cout][Kpad] with
 *     k = (kh*Kw + kw)*Cin + c, zero padded to a multiple of 32 (bf16) / 16 (fp32) elements.
 */
#ifndef COBEVT_HIP_H
#define COBEVT_HIP_H

#include <hip/hip_runtime_api.h>

#ifdef __cplusplus
extern "C" {
#endif

#define COBEVT_OK 0
#define COBEVT_ERR_ARG 1
#define COBEVT_ERR_SHAPE 2
#define COBEVT_ERR_LAUNCH 3
#define COBEVT_ERR_UNSUPPORTED 4

int cobevt_abi_version(void);
const char* cobevt_strerror(int code);

/*
 * Implicit-GEMM convolution / linear layer with fused prologue + epilogue.
 * Replaces: torchvision ResNet convs + folded BatchNorm + ReLU + residual add called from
 *   opv2v/opencood/models/backbones/resnet_ms.py:67-74; nn.Linear / 1x1 convs of
 *   opv2v/opencood/models/sub_modules/fax_modules.py:107,114-117,189-193,281-292,311-312,472-489;
 *   opv2v/opencood/models/fusion_modules/swap_fusion_modules.py:45-53; base_transformer.py:112-124;
 *   naive_decoder.py:78-87 (nearest x2 up-sampling folded into the gather); bev_seg_head.py:35-61;
 *   nuscenes/cross_view_transformer/model/{encoder_pyramid_axial.py:515-526, decoder.py:12-34, cvt.py:29-33}.
 * dims (int32[21]): dtype, N, H, W, Cin, Ho, Wo, Cout, Kh, Kw, stride, pad, K, Kpad, upsample(0/1),
 *   pre_relu(0/1), act(0 none,1 ReLU,2 exact GELU), store_mode(0 NHWC dtype,1 PixelUnshuffle(2) NHWC dtype,
 *   2 NCHW fp32, 3 NHWC fp32), out_H, out_W (dims of a possibly zero-padded output map, modes 0/3),
 *   smallc (1: `in` is fp32 NHWC with tiny Cin, taps decoded through klut[Kpad] = kh<<20|kw<<10|c, -1 = pad).
 * pre_scale/pre_shift (fp32[Cin], nullable): a <- a*scale+shift (+ReLU if pre_relu) applied to the gathered
 *   input (pre-activation BatchNorm->ReLU->1x1 conv).  bias fp32[Cout] nullable.  residual: same dtype,
 *   (N,Ho,Wo,Cout), nullable, added before the activation.
 */
int cobevt_conv2d_nhwc(const void* in, const void* wgt, const float* bias, const void* residual,
                       const float* pre_scale, const float* pre_shift, const int* klut, void* out,
                       const int* dims, hipStream_t stream);

/*
 * 3x3 / stride 1 / pad 1 convolution with an LDS-resident input patch (the fast path of cobevt_conv2d_nhwc for the
 * ResNet BasicBlocks resnet_ms.py:67-74, FAX Bottleneck / downsample convs fax_modules.py:472-489 and NaiveDecoder
 * naive_decoder.py:78-87).  Weights [Cout][Cin/cc][9][cc].  dims (int32[10]): dtype, N, H, W, Cin, Cout,
 * upsample(0/1: input is read through a nearest x2 up-sampling), act, store_mode (0 NHWC, 1 PixelUnshuffle(2)),
 * cc = channels per chunk (bf16: 64|32, fp32: 32|16; Cin % cc == 0).
 */
int cobevt_conv3x3_nhwc(const void* in, const void* wgt, const float* bias, const void* residual, void* out,
                        const int* dims, hipStream_t stream);

/*
 * The same 3x3 / stride 1 / pad 1 convolution (same reference call sites as cobevt_conv3x3_nhwc) with the weights
 * pre-ordered as MFMA B fragments, [Cout_p/32][Cin/cc][9 taps][4 k-groups][64 lanes][16 bytes] where lane = 32*half +
 * (cout % 32) holds bytes [32*kgroup + 16*half, +16) of that cout's 128-byte channel chunk and Cout_p >= Cout is the
 * zero-padded row count (multiple of 128): every weight operand is one coalesced 1-KB wave load from L2, LDS only holds
 * the input patch and there is one barrier per channel chunk instead of one per tap.  A workgroup owns MT strips of
 * 2 x 16 output pixels (numbered across image, row pair, column block) x 128 or 64 couts.  dims (int32[13]): dtype, N,
 * H, W, Cin, Cout, upsample, act, store_mode (0 NHWC, 1 PixelUnshuffle(2)), cc (bf16: 64, fp32: 32), Cout_p, variant =
 * 100 + 10*MT + (1 for 64-cout tiles; 3 = 32-cout tiles in four-wave workgroups, bf16 / stride 1, MT in 1..5), MT in 3..6
 * (0 = MT 5); the host picks MT so that the grid is a whole number of
 * workgroups per CU (cobevt_amd/ops.py conv3_tiling); stride (1, or 2 = the first conv of a down-sampling BasicBlock:
 * out (Ho, Wo) = ((H-1)/2+1, (W-1)/2+1), plain NHWC store, no up-sampling).  Needs N*H*W*Cin < 2^31.
 */
int cobevt_conv3x3_wfrag_nhwc(const void* in, const void* wfrag, const float* bias, const void* residual, void* out,
                              const int* dims, hipStream_t stream);

/*
 * The SECOND 3x3 convolution of a down-sampling BasicBlock together with the block's projection shortcut (torchvision
 * resnet.BasicBlock.forward with `downsample`, layer3.0 / layer4.0 as reached from resnet_ms.py:67-74), bf16:
 *     out = act(conv3x3(in; W2) + bias + bf16(conv1x1/stride2(in2; Wd) + bias2))      eval BatchNorms folded
 * in (N, H, W, Cin) = the block's first convolution's output, in2 (N, H2, W2, Cin2) = the block's input with
 * ((H2-1)/2+1, (W2-1)/2+1) = (H, W).  One launch instead of two (the shortcut's dense-row launch and its 2 x N*H*W*Cout
 * bytes of round trip go): behind the 3x3's channel chunks every wave computes the complete shortcut sum of some of its
 * output tiles from in2(2 oy, 2 ox), rounds it to bf16 as the separate launch stores it and adds it to its partial sum of
 * the 3x3 - the result differs from the two-launch path by fp32 summation order only.  wfrag: per
 * 32-cout tile the Cin/64 * 9 steps of cobevt_conv3x3_wfrag_nhwc's table followed by Cin2/64 steps of the same
 * [4 k-groups][64 lanes][16 bytes] shape built from Wd.  dims (int32[12]): dtype (0), N, H, W, Cin, Cout, act, Cout_p,
 * variant (100 + 10*MT + (1 for 64-cout tiles), MT in 3..5), H2, W2, Cin2; Cin % 64 == Cin2 % 64 == 0, Cin2 <= 256.
 */
int cobevt_conv3x3_ds_wfrag_nhwc(const void* in, const void* in2, const void* wfrag, const float* bias, const float* bias2, void* out,
                                 const int* dims, hipStream_t stream);

/*
 * Fused ResNet BasicBlock (stride 1, no downsample, C in {64, 128}): out = ReLU(conv2(ReLU(conv1(x) + b1)) + b2 + x) with
 * the eval BatchNorms folded into the weights / biases - torchvision resnet.BasicBlock.forward as reached from
 * resnet_ms.py:67-74 (layer1 / layer2 of the camera encoder).  One launch instead of two 3x3 launches: the intermediate
 * map stays in LDS (conv1 is recomputed on the 1-pixel halo conv2 needs) and is rounded to the storage type exactly as
 * the two-launch path stores it.  wfrag1 / wfrag2: the fragment-ordered tables of cobevt_conv3x3_wfrag_nhwc.
 * dims (int32[6]): dtype, N, H, W, C, tile_rows (0 = default; 8 | 16 pins the output tile height for C = 64 bf16).
 * Needs N*H*W*C < 2^31.
 */
int cobevt_basicblock_nhwc(const void* in, const void* wfrag1, const float* bias1, const void* wfrag2, const float* bias2,
                           void* out, const int* dims, hipStream_t stream);

/*
 * Fused down-sampling BasicBlock (ResNet layer2's first block: 64 -> 128 channels, stride 2, projection shortcut), bf16:
 * out = ReLU(conv2(ReLU(conv1/s2(x) + b1)) + b2 + conv1x1/s2(x) + b_ds) with the eval BatchNorms folded - torchvision
 * resnet.BasicBlock.forward with its `downsample` branch, as reached from resnet_ms.py:67-74.  One launch instead of three;
 * the intermediate map stays in LDS and is rounded to bf16 as the unfused path stores it, the shortcut is accumulated in
 * fp32 with conv2 (the unfused path rounds it to bf16 first).  wfrag1 / wfrag2: the fragment-ordered tables of
 * cobevt_conv3x3_wfrag_nhwc; wfrag_ds: the dense-row fragment table of cobevt_linear_rows_wfrag (K padded to 128).
 * dims (int32[6]): dtype (0 = bf16), N, H, W (even), Cin (64), Cout (128).  out: (N, H/2, W/2, 128).
 */
int cobevt_dsblock_nhwc(const void* in, const void* wfrag1, const float* bias1, const void* wfrag2, const float* bias2,
                        const void* wfrag_ds, const float* bias_ds, void* out, const int* dims, hipStream_t stream);

/*
 * ResNet stem: 7x7 / stride 2 / pad 3 conv on the fp32 3-channel channels-last image (+ folded BN, ReLU), computed as a
 * 4x4 stride-1 conv over the 2x2 space-to-depth image; torchvision resnet conv1/bn1/relu via resnet_ms.py:67-69.
 * wgt [Cout][16 taps][16] (12 real (dy,dx,c) channels + 4 zeros).  dims (int32[6]): dtype, N, H, W (even), Cout, act.
 */
int cobevt_stem_conv7x7s2(const float* in, const void* wgt, const float* bias, void* out, const int* dims,
                          hipStream_t stream);

/*
 * The stem followed by MaxPool2d(3, 2, 1) in one launch (torchvision resnet conv1/bn1/relu/maxpool, resnet_ms.py:67-71):
 * a workgroup computes the conv on the (2 R + 1) x 17 region an R x 8 tile of pooled pixels covers (R = 5 bf16, 3 fp32) and writes
 * the maxima, so the
 * (N, H/2, W/2, 64) stem map never reaches HBM.  Cout = 64, ReLU.  out (N, H/4, W/4, 64).
 * dims (int32[4]): dtype, N, H, W (multiples of 4).
 */
int cobevt_stem_conv7x7s2_pool(const float* in, const void* wgt, const float* bias, void* out, const int* dims,
                               hipStream_t stream);

/*
 * cobevt_stem_conv7x7s2_pool on the uint8 camera frames themselves: the ingest step of the reference's loop - RgbPreProcessor's
 * /255, (x - mean) / std on the host (opv2v/opencood/data_utils/pre_processor/rgb_preprocessor.py:14-31), the collate cast to
 * fp32 and `.to(device)` of the fp32 image (opv2v/opencood/tools/inference_camera.py:56-61) - folded into the stem's patch
 * gather.  in (N, H, W, 3) uint8; lut float[3][256]: lut[c][u] = the fp32 value the reference's pipeline gives byte u of channel c
 * (host/rgb_preprocessor.py: normalisation_table) - the kernel output is bit-identical to the fp32-image entry point fed lut[c][u].
 * A quarter of the bytes over PCIe and out of HBM.  dims as above.
 */
int cobevt_stem_conv7x7s2_pool_u8(const unsigned char* in, const float* lut, const void* wgt, const float* bias, void* out,
                                  const int* dims, hipStream_t stream);

/*
 * Dense-row GEMM with fused LayerNorm / pre-activation on the A operand and fused bias / residual / activation:
 * the fast path of every nn.Linear and 1x1 stride-1 convolution (fax_modules.py:189-193,281-292,309-313,411,435,472;
 * swap_fusion_modules.py:45-53; base_transformer.py:102-124).  wgt [N][Kp] (Kp = K rounded up to 128 bf16 / 64 fp32
 * elements).  dims (int64[16]): dtype, M, N, K, Kp, lda, pre_relu, act, src_H, src_W, out_H, out_W (these four remap
 * output rows into a zero-padded map; equal values = plain rows), ln (1 = normalise each A row over K first; the
 * LayerNorm affine is folded into wgt / bias by the host, or passed as ln_gamma/ln_beta fp32[K]; needs K <= one K-tile),
 * in_stride, in_H, in_W (in_stride s > 1: a 1x1 / stride-s convolution - the BasicBlock downsample path
 * resnet_ms.py:67-74 via torchvision resnet.py `downsample` - output row (n, oy, ox) of the (src_H, src_W) map reads
 * input pixel (n, s*oy, s*ox) of the (in_H, in_W) map; 1 = rows as stored).
 * pre_scale/pre_shift fp32[K] (nullable).
 */
int cobevt_linear_rows(const void* in, const void* wgt, const float* bias, const void* residual, const float* ln_gamma,
                       const float* ln_beta, const float* pre_scale, const float* pre_shift, void* out, const long* dims,
                       float ln_eps, hipStream_t stream);

/*
 * The same dense-row GEMM (same reference call sites and dims as cobevt_linear_rows) with the weights in MFMA fragment
 * order - rows zero-padded to a multiple of 128, [rows/32][Kp*esz/32 k-groups][64 lanes][16 bytes], lane = 32*half +
 * row%32 holding bytes [32*kgroup + 16*half, +16) of its row - and persistent workgroups: a workgroup walks the row tiles
 * of one 128-column tile with its weight fragments resident in registers (K <= one K-tile) and the next tile's rows in
 * flight under the current tile's MFMAs, epilogue and stores.  N % (8 bf16 | 4 fp32) == 0.  No ln_gamma / ln_beta
 * (the LayerNorm affine is folded into the weights by the host).
 */
int cobevt_linear_rows_wfrag(const void* in, const void* wfrag, const float* bias, const void* residual,
                             const float* pre_scale, const float* pre_shift, void* out, const long* dims, float ln_eps,
                             hipStream_t stream);

/*
 * Fused row-local chain after an attention (bf16 storage, or fp32 storage for C = 128 / hidden 256):  y = a.Wp^T (+bp) + skip ;  z = y + fc2(GELU(fc1'(norm(y)))) ;
 * out = post-LayerNorm(z) (optional) ;  out_next = act(norm?(out).Wn'^T + bn') (optional).  Replaces
 * fax_modules.py:240,246-247 + :411 / :435-437 and swap_fusion_modules.py:126,177 + base_transformer.py:102-124 in one
 * launch (hidden activations stay in LDS); the optional next projection is the row-local GEMM that reads `out` next in
 * the reference graph (to_qkv behind PreNormResidual.norm swap_fusion_modules.py:93, to_q of the second cross attention
 * fax_modules.py:201, the first 1x1 conv + BN + ReLU of the following ResNetBottleNeck fax_modules.py:472).
 * wp (C x 128), w1 (Hd x 128, LayerNorm affine folded in), w2 (C x Hdp), wnext (Nn x 128, LayerNorm affine / BN folded
 * in; nullable together with out_next [M][Nn]) are given in MFMA fragment order: rows zero-padded to a multiple of 128,
 * [rows/32][Kp/16][64 lanes][16 bytes] with lane = 32*half + row%32 holding bytes [32*kgroup + 16*half, +16) of its row,
 * so a wave's weight operand is one coalesced 1-KB load and no weight panel passes through LDS.  dims (int32[10]): dtype (0 bf16; 1 fp32
 * storage - round 6, csrc/row_chain_f32.hip: C == 128, Hd == Hdp == 256, fragments [rows/32][Kp/8][64 lanes][16 bytes] of fp32, the matrix
 * path of the library it is called from; other fp32 widths return COBEVT_ERR_UNSUPPORTED and the caller runs the GEMMs separately), M, C(<=128), Hd(<=256), Hdp, Nn (<= 768),
 * next_ln (1 = normalise the stored `out` rows first), next_act (0 none, 1 ReLU, 2 GELU), rows per workgroup (0 = 32; 64),
 * skip_rows (0 = M; a divisor of M: `skip` has that many rows and row m adds skip[m % skip_rows] - the learned prior
 * broadcast over the batch, fax_modules.py:509-510).
 */
int cobevt_attn_mlp_chain(const void* a, const void* skip, void* out, const void* wp, const float* bp, const void* w1,
                          const float* b1, const void* w2, const float* b2, const float* post_gamma,
                          const float* post_beta, const void* wnext, const float* bnext, void* out_next, const int* dims,
                          float eps1, float eps_post, float eps_next, hipStream_t stream);

/*
 * Fused gathered attention: window / dilated-grid partition -> QK^T -> (+relative position bias, key mask)
 * -> softmax -> PV -> (mean over query cameras) -> partition reverse, for projected token matrices.
 * Replaces: CrossWinAttention core, fax_modules.py:211-237,243 with the partitions of :399-404,:417-424 and
 *   reverses of :409,:433; swap Attention core, swap_fusion_modules.py:93-123 with :172-190;
 *   FAX global Attention core, fax_modules.py:137-171.
 * dims (int32[40]): dtype, B, L(windows), heads, ldq, ldk, ldv, ldo, qoff, koff, voff, ooff,
 *   bias_mode(0/1), bias_rows, bias_L, mean_q (0 none | 1 camera mean of per-camera query copies, fax_modules.py:243 | 2 per-camera
 *   query copies scored against their own camera's keys under ONE softmax over all cameras: CVT CrossAttention,
 *   cvt_modules.py:142-153), then three token maps {mode(0 window,1 grid,2 stored
 *   partitioned), ncam, HH, WW, w1, w2, X, Y} for q, k/v and out.  Head dim is 32.
 * bias_table fp32[bias_rows][heads] indexed ((dl+bias_L-1)(2w1-1)+(di+w1-1))(2w2-1)+(dj+w2-1);
 * mask fp32 (B,HH,WW,ncam) over the key map, 0 = masked (nullable).  `scale` multiplies QK^T.
 */
int cobevt_window_attention(const void* q, const void* k, const void* v, void* out, const float* bias_table,
                            const float* mask, const int* dims, float scale, hipStream_t stream);

/* Training slice (fp32 storage, dims[0] dtype = 1).  cobevt_window_attention_lse: the forward above that also stores the base-2
 * log-sum-exp of every query's logits, lse[B][L][heads][Nq] (mean_q must be 0).  cobevt_window_attention_bwd: given the forward
 * tensors, `out`, lse and dout (layout of out), writes dq, dk, dv (layouts of q, k, v; rows no window covers are left
 * untouched) and ADDS the bias-table gradient into dbias[bias_rows][heads] (zero-initialised; nullable without bias).  dlse (nullable,
 * layout of lse, drop_p must be 0): gradient w.r.t. the NATURAL log-sum-exp ln(2) * lse when the caller combines the lse further - CVT's
 * CrossAttention (cvt_modules.py:142-153: one softmax over all cameras' keys) trains as per-camera attentions merged by softmax(lse).
 * Replaces torch autograd through the einsum / softmax / einsum of fax_modules.py:219-237, swap_fusion_modules.py:100-121
 * (train_camera.py:143-179 loss.backward()). */
int cobevt_window_attention_lse(const void* q, const void* k, const void* v, void* out, float* lse, const float* bias_table,
                                const float* mask, const int* dims, float scale, float drop_p, unsigned drop_seed,
                                const unsigned* drop_seed_dev, hipStream_t stream);
int cobevt_window_attention_bwd(const void* q, const void* k, const void* v, const void* out, const float* lse,
                                const void* dout, const float* dlse, void* dq, void* dk, void* dv, float* dbias, const float* bias_table,
                                const float* mask, const int* dims, float scale, float drop_p, unsigned drop_seed,
                                const unsigned* drop_seed_dev, hipStream_t stream);
/* drop_p > 0: nn.Dropout on the attention probabilities (FAX global attention in train mode, fax_modules.py:114,161): element
 * (query, key) is kept with probability 1 - drop_p and scaled by 1 / (1 - drop_p); the decision is a counter-based hash of
 * (seed, batch, window, head, query, key), regenerated by the backward kernels, seed = drop_seed + *drop_seed_dev (drop_seed_dev: a
 * nullable DEVICE word - a training step replayed from a captured HIP graph bumps it inside the graph, the scalar arguments being
 * frozen in the graph's nodes; forward and backward of one step must see the same value).  cobevt_attention_dropout_mask dumps
 * the mask of an effective seed (test hook): keep uint8 [B][L][heads][Nq][Nk]. */
int cobevt_attention_dropout_mask(int B, int L, int heads, int Nq, int Nk, float drop_p, unsigned drop_seed, unsigned char* keep,
                                  hipStream_t stream);

/* Row-local backward kernels of the training slice (fp32).  cobevt_layernorm_bwd: x, dy (rows, C) -> dx; ADDS dy * xhat / dy
 * column sums into zero-initialised dgamma[C] / dbeta[C] (both nullable together); gamma nullable (= 1).  C % 4 == 0, C <= 1024.
 * cobevt_gelu: out = GELU(x) (dy null) or dy * GELU'(x) (exact erf form, nn.GELU()); n % 4 == 0.
 * (nn.LayerNorm / nn.GELU under autograd: base_transformer.py:102-124, swap_fusion_modules.py:275-279, fax_modules.py:189-191.) */
int cobevt_layernorm_bwd(const float* x, const float* dy, const float* gamma, float* dx, float* dgamma, float* dbeta, int rows,
                         int C, float eps, hipStream_t stream);
int cobevt_gelu(const float* x, const float* dy, float* out, long n, hipStream_t stream);
/*
 * The same inside a bf16 autocast region (train_camera.py:157-160: torch's autocast runs layer_norm in fp32 and hands its result to the
 * bf16 projection that follows; nn.GELU runs in its input's bf16): every tensor fp32 or bf16 (dtype codes 0 = bf16, 1 = fp32), fp32
 * arithmetic.  cobevt_layernorm_fwd_t dtypes = [x, y]; cobevt_layernorm_bwd_t dtypes = [x, dy, dx]; cobevt_gelu_bf16: all bf16, n % 8 == 0.
 */
int cobevt_layernorm_fwd_t(const void* x, const float* gamma, const float* beta, void* y, int rows, int C, float eps, const int* dtypes,
                           hipStream_t stream);
int cobevt_layernorm_bwd_t(const void* x, const void* dy, const float* gamma, void* dx, float* dgamma, float* dbeta, int rows, int C,
                           float eps, const int* dtypes, hipStream_t stream);
int cobevt_gelu_bf16(const void* x, const void* dy, void* out, long n, hipStream_t stream);
/* Weight gradient of a k x k convolution, channels-last: dw fp32 (Cout, Cin, k, k) += sum over output pixels of dy (N, Ho, Wo, Cout)
 * x the tap-shifted x (N, H, W, Cin); dw must be zero-initialised (fp32 atomics).  dims (int32[11]): N, H, W, Cin, Ho, Wo, Cout, k,
 * stride, pad, storage type of x and dy (0 bf16 - the autocast path -, 1 fp32).  (cuDNN under autograd in the reference: every
 * nn.Conv2d on the path, train_camera.py:166-173.) */
int cobevt_conv_wgrad(const void* x, const void* dy, float* dw, const int* dims, hipStream_t stream);
/* The same weight gradient on the bf16 matrix path from BLOCKED bf16 copies of the operands (8 pixels of one channel = one 16-byte
 * piece, so a lane's matrix operand is one coalesced load and zero padding replaces every bounds test; cobevt_wgrad_block_operand
 * makes them): xb [N][Hp][XB][P][Cin][8] = x zero-padded by `pad` rows / columns, db [N][Ho][DB][Cout][8] = dy with rows zero-padded
 * to an even number DB of blocks; dw fp32 (Cout, Cin, k, k), zero-initialised.  Output row oy, tap row a reads padded row oy * sy + a
 * (Hp >= (Ho - 1) sy + k).  mode 0: stride 1, k = 1 / 3, P = 1, the taps of a row are pixel shifts (XB >= DB + (k > 1));
 * mode 1: stride 2 (sy = 2), k = 3 / pad 1 with P = 2 planes = even / odd padded columns (XB >= DB + 1), or k = 1 / pad 0 with the even
 * columns only (P = 1); mode 2: the k tap columns as P = k planes, k * Cin <= 32, k odd <= 7 (the 7x7 / stride 2 stem on 3 channels:
 * a tap row is one matrix instruction).  dims (int32[10]): N, Hp, XB, Cin, Ho, DB, Cout, k, sy, mode. */
int cobevt_conv_wgrad_blocked(const void* xb, const void* db, float* dw, const int* dims, hipStream_t stream);

/* out[b][i] = max over l of in[b][l][i] (F-Cooper max-out fusion over the max_cav agent slots, SpatialFusionMask,
 * opv2v/opencood/models/fusion_modules/f_cooper_fuse.py:30-36).  in (B, L, per) contiguous, dtype 0 bf16 / 1 fp32, per % 8 == 0. */
int cobevt_agent_max(const void* in, void* out, int dtype, int B, int L, long per, hipStream_t stream);

/* Fused torchvision Bottleneck(128, 32) (1x1 128->32, 3x3 32->32, 1x1 32->128, eval BatchNorms folded, ReLUs, identity skip) on
 * a channels-last bf16 map: ResNetBottleNeck(dim) of the FAX pyramid, fax_modules.py:10,472,512.  w1 / w2 / w3: MFMA fragment
 * tables [8][64][8] / [9][2][64][8] / [4][2][64][8] bf16 (layout: cobevt_amd/ops.py BottleneckPlan; w3's contraction index in
 * accumulator-register order), b1[32] / b2[32] / b3[128] fp32.  dims (int32[7]): dtype (0), N, H, W, C (128), mid (32),
 * tile rows (0 = automatic | 8 | 16). */
int cobevt_bottleneck_nhwc(const void* in, const void* w1, const void* w2, const void* w3, const float* b1, const float* b2,
                           const float* b3, void* out, const int* dims, hipStream_t stream);

/*
 * The same block for fp32 STORAGE (round 6, csrc/bottleneck_f32.hip; the matrix path is the library's): in (N, H, W, 128) fp32, out the same;
 * y1 (N, H, W, 32) = ReLU(conv1(x) + b1) when the producer of x already computed it (the row chain's `next` projection), else null and the
 * kernel computes conv1 on the tile's halo region itself.  w1frag / w3frag: the fp32 fragment tables of cobevt_linear_rows_small_k for the
 * folded 32 x 128 / 128 x 32 matrices ([rows/32][Kp/8][64 lanes][16 bytes], Kp = K rounded up to 64), w2frag: the table of
 * cobevt_conv3x3_wfrag_nhwc for the folded 3x3 ([tile][1 chunk][9 taps][4 k-groups][64 lanes][16 bytes]; tile 0 is used).
 * dims (int32[4]): N, H, W, w3 tile stride in 16-byte units (= Kp / 8 * 64).
 */
int cobevt_bottleneck_f32_nhwc(const void* in, const void* y1, const void* w1frag, const float* b1, const void* w2frag, const float* b2,
                               const void* w3frag, const float* b3, void* out, const int* dims, hipStream_t stream);

/* Test hooks for the integer part of the attention kernels (north-star: window index arithmetic bit-exact), computed by the
 * same device functions the attention kernels use.  cobevt_attention_index_map: rows[b][l][t] (int32, B * X*Y * ncam*w1*w2) =
 * row of token t of window l in the (B*ncam, HH, WW) token matrix, for one token map {mode, ncam, HH, WW, w1, w2, X, Y}:
 * the einops partitions of fax_modules.py:399-404 (window), :417-424 (grid), swap_fusion_modules.py:172-190.
 * cobevt_attention_bias_index: idx[tq][tk] (int32, Nq * Nk) = relative-position table row for query token tq / key token tk
 * (swap_fusion_modules.py:63-85 with bias_L = agent_size; fax_modules.py:123-130 with bias_L = 1). */
int cobevt_attention_index_map(const int* map8, int B, int* rows, hipStream_t stream);
int cobevt_attention_bias_index(const int* qmap8, const int* kmap8, int bias_L, int* idx, hipStream_t stream);

/* LayerNorm over channels (gamma/beta both null = normalisation only), optionally after a mean over `navg` slices (mlp_head).
 * Replaces: nn.LayerNorm of fax_modules.py:189-191,309-313,435-437; swap_fusion_modules.py:275-279;
 *   base_transformer.py:102-109. */
int cobevt_layernorm(const void* in, const float* gamma, const float* beta, void* out, int dtype, int rows, int C,
                     float eps, int navg, long avg_stride, long in_batch_stride, int rows_per_batch,
                     hipStream_t stream);

/* Camera-ray positional embedding, fax_modules.py:346-358.  out (BN, hw, D) channels-last. */
int cobevt_fax_ray_embed(const float* I_inv, const float* E_inv, const float* image_plane, const float* w_img,
                         const float* w_cam, void* out, int dtype, int BN, int hw, int D, hipStream_t stream);

/* BEV query embedding + prior, fax_modules.py:370-375,387-388.  out (B, n, hw, D).  x (B, hw, D), or with x_bcast = 1
 * one (hw, D) prior shared by every b (the first pyramid level, fax_modules.py:509-510 repeat of the learned prior). */
int cobevt_fax_bev_embed(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                         const float* w_cam, const void* x, void* out, int dtype, int B, int n, int hw, int D,
                         int x_bcast, hipStream_t stream);

/*
 * BEV query embedding + prior fused with the to_q LayerNorm + Linear that consumes it (fax_modules.py:370-375,387-388
 * followed by CrossWinAttention.to_q :193-195,201): the (B, n, hw, D) query never reaches HBM - the dense-row GEMM
 * computes its A rows  x[b][pix] + L2norm_c(w_bev.world[pix] + b_bev - w_cam.E_inv[b,cam][:,3])  on the fly, rounds them
 * to the storage type exactly as cobevt_fax_bev_embed would have stored them, normalises (LayerNorm affine folded into
 * wgt / bias) and multiplies.  out (B*n*hw, N).  wgt [N][Kp] as for cobevt_linear_rows.  dims (int64[8]): dtype, B, n,
 * hw (multiple of 128), D (= K <= one K-tile), N, Kp, ln.
 */
int cobevt_bev_embed_linear_rows(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                                 const float* w_cam, const void* x, const void* wgt, const float* bias, void* out,
                                 const long* dims, float ln_eps, hipStream_t stream);

/* MaxPool2d(3, 2, 1) channels-last; torchvision ResNet stem reached from resnet_ms.py:71. */
int cobevt_maxpool3x3s2(const void* in, void* out, int dtype, int N, int H, int W, int C, hipStream_t stream);

/* Strided (n,c,h,w) view <-> contiguous channels-last with dtype cast (module boundary; resnet_ms.py:62-65). */
int cobevt_to_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W,
                   const long* strides, hipStream_t stream);
int cobevt_from_nhwc(const void* in, int in_dtype, void* out, int out_dtype, int N, int C, int H, int W,
                     const long* strides, hipStream_t stream);

/* Pairwise-warp fusion baselines (V2VNet fusion_modules/v2v_fuse.py:72-144, DiscoNet fusion_modules/disconet_fuse.py:106-168) on the
 * un-grouped channels-last agent batch x (sum(record_len), H, W, C), H == W; record_len device int32[B]; pairwise (B, L, L, 4, 4)
 * fp32 = batch['pairwise_t_matrix'] (intermediate_fusion_dataset.py:110-150).  All maps stay in their original orientation (the
 * reference's transpose + flip is folded into the kernels' indexing).
 * cobevt_pairwise_warp: nb[b][i][j] (B, L, L, H, W, C) = agent j's map warped into agent i's frame (zeros unless i, j <
 *   record_len[b]); roi (B, L, L, H, W) fp32 = get_rotated_roi's mask (torch_transformation_utils.py:77-105) at the position the
 *   reference multiplies that output pixel with.
 * cobevt_agent_message_reduce: out[off_b + i] = mean (mode 0) | max (mode 1) over j < record_len[b] of
 *   (msg[b][i][j] + ego[off_b + i]) * roi[b][i][j]   (v2v_fuse.py:108-119; ego = the ego half of msg_cnn + its bias).
 * cobevt_gru_zero_state: out[row] = sigmoid(in[row][0:C]) * tanh(in[row][C:2C]): the ConvGRU cell with the zero hidden state
 *   the fusions always pass (convgru.py:57-78, v2v_fuse.py:125-130).
 * cobevt_agent_softmax_sum: out[off_b + i] = sum_j softmax_j(score or -inf where roi == 0) * nb[b][i][j] * roi[b][i][j]
 *   (disconet_fuse.py:35-42,141-150); score = column 0 of a (B*L*L*H*W, lds) matrix. */
int cobevt_pairwise_warp(const void* x, const float* pairwise, const int* record_len, void* nb, float* roi, int dtype, int B, int L,
                         int H, int W, int C, float discrete_ratio, float downsample_rate, hipStream_t stream);
/* Adjoint of cobevt_pairwise_warp w.r.t. x (training; torch autograd through warp_affine in the reference): dnb (B, L, L, H, W, C) ->
 * dx fp32 (N, H, W, C), zero-initialised; the same bilinear sample positions, scattered with fp32 atomics. */
int cobevt_pairwise_warp_bwd(const void* dnb, const float* pairwise, const int* record_len, float* dx, int dtype, int B, int L, int H,
                             int W, int C, float discrete_ratio, float downsample_rate, hipStream_t stream);
int cobevt_agent_message_reduce(const void* msg, const void* ego, const float* roi, const int* record_len, void* out, int dtype,
                                int B, int L, int HW, int C, int mode, hipStream_t stream);
int cobevt_gru_zero_state(const void* in, void* out, int dtype, long rows, int C, hipStream_t stream);
int cobevt_agent_softmax_sum(const void* score, int lds, const void* nb, const float* roi, const int* record_len, void* out, int dtype,
                             int B, int L, int HW, int C, int use_mask, hipStream_t stream);

/* regroup: split by record_len (device int32[B]), zero pad to max_cav, agent mask (B,max_cav) fp32.
 * Replaces fuse_utils.py:8-61 without its host synchronisation (:26). */
int cobevt_regroup(const void* in, const int* record_len, void* out, float* mask, int dtype, int B, int max_cav,
                   long elems_per_agent, hipStream_t stream);

/* STTF warp into the ego frame + ROI/agent mask.  x (B*L,H,W,C) -> out (B,L,H,W,C), com_mask (B,H,W,1,L).
 * Replaces corpbevt.py:28-64 and torch_transformation_utils.py:11-134,160-355. tmat: (B,L,4,4) fp32.
 * With record_len (device int32[B]) x is the un-grouped agent batch (sum(record_len),H,W,C) and regroup
 * (fuse_utils.py:8-61) happens inside: agent l of sample b is row sum(record_len[:b]) + l, absent agents warp to zeros and
 * get mask 0; cav_out (B,L) fp32 (nullable) receives the agent mask; cav_mask is then ignored. */
int cobevt_sttf_warp(const void* x, const float* tmat, const float* cav_mask, void* out, float* com_mask,
                     const int* record_len, float* cav_out, int dtype, int B, int L, int H, int W, int C,
                     float discrete_ratio, float downsample_rate, hipStream_t stream);

/* Batched inverse of n (dim x dim) fp32 matrices, dim in {3, 4}; fax_modules.py:500-501 (intrinsic.inverse()),
 * nuscenes encoder_pyramid_axial.py:538-539. */
int cobevt_invert_small(const float* in, float* out, int n, int dim, hipStream_t stream);

/* Channels-last resize: mode 0 nearest (F.interpolate default), mode 1 bilinear align_corners=True;
 * nuscenes/cross_view_transformer/model/decoder.py:12,31. */
int cobevt_resize_nhwc(const void* in, void* out, int dtype, int N, int H, int W, int C, int Ho, int Wo, int mode,
                       hipStream_t stream);

/* y = x*scale[c] + shift[c] on contiguous (N,C,HW) fp32; Normalize, nuscenes encoder_pyramid_axial.py:41-49. */
int cobevt_channel_affine(const float* in, const float* scale, const float* shift, float* out, long N, int C, long HW,
                          hipStream_t stream);

/*
 * Pull `bytes` (a multiple of 16) from pinned, device-visible HOST memory (hipHostMalloc / torch .pin_memory(): the host pointer is
 * the device pointer) into device memory with `blocks` (0 = 128; the captured step uses 16) workgroups of 256 lanes, each lane with eight
 * 8-byte SYSTEM-SCOPE relaxed atomic loads in flight (global_load_dwordx2 sc0 sc1 behind an acquire fence: a ring slot the host has
 * rewritten is never served out of the GPU's caches) and plain 8-byte stores: the ingest step of the reference's loop
 * (opv2v/opencood/tools/inference_camera.py:56-61 `.to(device)`) as a kernel that can be CAPTURED in the step's HIP graph and runs
 * beside the compute kernels (no LDS, 52 VGPRs) - host.pipeline.PipelinedCorpBEVT(host_ingest=True).
 */
int cobevt_host_fetch(const void* host_src, void* dst, long bytes, int blocks, hipStream_t stream);

/* ---- downstream of the hot path: the logits -> scored maps -------------------------------------- */

/* CameraBevPostprocessor.softmax_argmax, opv2v/opencood/data_utils/post_processor/camera_bev_postprocessor.py:55-59:
 * prob = Softmax(dim=1)(logits) (fp32) and map = argmax(prob, dim=1) (int64; the first maximum wins, and it is the
 * argmax of the rounded probabilities as in the reference, not of the logits).  logits (N, C, hw) planar, dtype 0 bf16 /
 * 1 fp32, C <= 8; prob (N, C, hw) fp32; map (N, hw) int64. */
int cobevt_softmax_argmax(const void* logits, float* prob, long long* map, int dtype, int N, int C, int hw,
                          hipStream_t stream);

/* The pixel counts mean_IU / mean_precision reduce their class masks to, opv2v/opencood/utils/seg_utils.py:6-50:
 * counts[n][c] = (n_ii = |pred==c & gt==c|, t_i = |gt==c|, n_ij = |pred==c|) for c < K, and counts[n][K] =
 * (|pred outside [0,K)|, |gt outside [0,K)|, 0).  pred, gt (N, hw) int64; counts (N, K + 1, 3) uint64, zeroed here. */
int cobevt_seg_class_counts(const long long* pred, const long long* gt, unsigned long long* counts, int N, int hw, int K,
                            hipStream_t stream);

/* Class-weighted cross entropy nn.CrossEntropyLoss(weight=w) computes in VanillaSegLoss (validation loss of
 * train_camera.py:182-196; opv2v/opencood/loss/vanilla_seg_loss.py:18-23,58-70): out[0] = sum_i w[y_i] (logsumexp(x_i) - x_i[y_i])
 * / sum_i w[y_i], out[1] / out[2] = numerator / denominator, out[3] = number of labels outside [0, C) other than -100 (the
 * ignore_index of nn.CrossEntropyLoss, which contributes to neither sum); the reference raises for such labels, so the caller
 * must refuse a result with out[3] != 0.  logits (N, C, hw) planar (dtype 0 bf16 / 1 fp32, 2 <= C <= 8), target (N, hw) int64,
 * weight [C] fp32, out >= 4 floats, scratch >= 3 * N * ceil(hw / 4096) floats, fixed summation order. */
int cobevt_weighted_cross_entropy(const void* logits, const long long* target, const float* weight, float* scratch, float* out,
                                  int dtype, int N, int C, int hw, hipStream_t stream);
/* Backward of the loss above for fp32 logits: dlogits (N, C, hw) = upstream[0] * w[y] (softmax_c - [c == y]) / sum w[y]; stats = the
 * forward's out[4]; upstream = the incoming gradient of the scalar loss (device fp32[1]).  train_camera.py:166-173 loss.backward(). */
int cobevt_weighted_cross_entropy_bwd(const float* logits, const long long* target, const float* weight, const float* stats,
                                      const float* upstream, float* dlogits, int N, int C, int hw, hipStream_t stream);

/* 3x3 / stride 1 / pad 1 convolution with 1..4 output channels on a channels-last map -> fp32 NCHW logits: the BevSegHead heads
 * (bev_seg_head.py:20-33,44-58).  wgt fp32 [Cout][9 taps][Cin] (tap = kh * 3 + kw), bias fp32[Cout] nullable, Cin a multiple of
 * the 16-byte chunk with at most 256 bytes per pixel. */
int cobevt_conv3x3_head_nchw(const void* in, const float* wgt, const float* bias, float* out, int dtype, int N, int H, int W, int Cin,
                             int Cout, hipStream_t stream);

/* The same GEMM as cobevt_linear_rows for K <= 512 in bf16 (the to_q / to_k / to_v / to_qkv projections behind a LayerNorm,
 * feature_proj / feature_linear behind BN + ReLU, the Bottleneck 1x1 convs), on the row chain's structure: 32-row workgroups,
 * weights as MFMA fragments straight from L2.  wfrag: [N_p/32][8][64 lanes][16 bytes] as for cobevt_attn_mlp_chain (rows
 * zero-padded to a multiple of 128 columns K_p, K_p / 16 k-groups per tile, N_p = N rounded up to 128).  dims (int64[13]):
 * dtype (0), M, N (multiple of 8, <= 4096), K (multiple of 8, <= 512; ln needs K <= 128), lda, pre_relu, act (0..4), ln,
 * in_stride, src_H, src_W, in_H, in_W (strided row gather of a 1x1 / stride-2 conv as in cobevt_linear_rows; 1 = dense rows).
 * residual [M][N] (dense rows only), pre_scale / pre_shift [K], bias [N] nullable. */
int cobevt_linear_rows_small_k(const void* in, const void* wfrag, const float* bias, const void* residual, const float* pre_scale,
                               const float* pre_shift, void* out, const long* dims, float ln_eps, hipStream_t stream);

/* cobevt_bev_embed_linear_rows on the 32-row kernel: the BEV query  x[b][pix] + L2norm_c(w_bev . world[pix] + b_bev -
 * w_cam . E_inv[b, cam][:, 3])  (fax_modules.py:370-375,387-388) is produced while the A rows are staged, rounded as
 * cobevt_fax_bev_embed would have stored it, normalised and multiplied by to_q (fax_modules.py:193-195,201) - the
 * (B, n, hw, D) query never reaches HBM.  wfrag as cobevt_linear_rows_small_k.  dims (int64[8]): dtype (0), B, n, hw (multiple
 * of 32), D (= K <= 128), N, ln, x_bcast (1: x is one (hw, D) prior shared by every b).  out (B*n*hw, N). */
int cobevt_bev_embed_linear_rows_small_k(const float* E_inv, const float* world, const float* w_bev, const float* b_bev,
                                         const float* w_cam, const void* x, const void* wfrag, const float* bias, void* out,
                                         const long* dims, float ln_eps, hipStream_t stream);

/* Forward of BinarySegmentationLoss / CenterLoss, nuscenes/cross_view_transformer/losses.py:27-84: sigmoid focal loss (fvcore's
 * published definition) of pred (N, C, hw) fp32 logits against label_c = max over the label channels in label_mask[c]
 * (soft_label: label channel c itself, NL == C), over the pixels with visibility >= min_visibility (< 0: all), mean.
 * out[0] = mean, out[1] = sum, out[2] = count; scratch >= 2 * N * ceil(hw / 2048) floats; fixed summation order. */
int cobevt_sigmoid_focal_loss(const float* pred, const float* label, const unsigned char* visibility, const unsigned int* label_mask,
                              float* scratch, float* out, int N, int C, int NL, int hw, int min_visibility, float alpha,
                              float gamma, int soft_label, hipStream_t stream);

/* nuScenes IoU metric, nuscenes/cross_view_transformer/metrics.py:22-31,56-72: counts[t] += (tp, fp, fn) of
 * sigmoid(pred) >= thresholds[t] against label_c = any(label[l] != 0 for l in the bit mask label_mask[c]) over the pixels with
 * visibility >= min_visibility (min_visibility < 0: all pixels, visibility may be null).  pred (N, C, hw) fp32 logits, label
 * (N, NL <= 32, hw) fp32, visibility (N, hw) uint8, thresholds [T <= 8] fp32, counts (T, 3) uint64 - accumulated, not zeroed. */
int cobevt_iou_counts(const float* pred, const float* label, const unsigned char* visibility, const unsigned int* label_mask,
                      const float* thresholds, unsigned long long* counts, int N, int C, int NL, int hw, int T,
                      int min_visibility, hipStream_t stream);

/* ---- upstream of the nuScenes path: MBConv pieces of the EfficientNet image backbone wrapped by
 * nuscenes/cross_view_transformer/model/backbones/efficientnet.py:24-96 (efficientnet-pytorch 0.7.1 MBConvBlock.forward;
 * the 1x1 convolutions of a block go through cobevt_linear_rows / cobevt_conv2d_nhwc with act 3 = swish) -------------- */

/* k x k depthwise convolution (groups == channels) + folded BatchNorm + activation on a channels-last map.  wgt [k*k][C]
 * fp32 (BN scale folded), bias [C] fp32.  dims (int32[12]): dtype, N, H, W, C (multiple of 8), k (<= 7), stride, pad_top,
 * pad_left (the TensorFlow-"same" extra row / column at the bottom / right is implicit: taps outside the map read zero),
 * Ho, Wo, act (0 none, 1 ReLU, 2 GELU, 3 swish, 4 sigmoid). */
int cobevt_depthwise_conv_nhwc(const void* in, const float* wgt, const float* bias, void* out, const int* dims,
                               hipStream_t stream);

/* Squeeze: out[n][c] = mean over the hw pixels of in (N, hw, C), fp32, fixed summation order. */
int cobevt_spatial_mean_nhwc(const void* in, float* out, int dtype, int N, int hw, int C, hipStream_t stream);

/* Excitation: gate (N, C) = sigmoid(w_expand (C x Cs) . swish(w_reduce (Cs x C) . mean (N, C) + b_reduce) + b_expand), fp32. */
int cobevt_se_gate(const float* mean, const float* w_reduce, const float* b_reduce, const float* w_expand,
                   const float* b_expand, float* gate, int N, int C, int Cs, hipStream_t stream);

/* out (N, hw, C) = in * gate (N, C). */
int cobevt_channel_gate_nhwc(const void* in, const float* gate, void* out, int dtype, int N, int hw, int C,
                             hipStream_t stream);

/* SwapFusionEncoder.mlp_head in one launch (swap_fusion_modules.py:275-281: mean over the agents, LayerNorm, Linear): out (B * R, N) =
 * act(LayerNorm?(mean over j < L of in[b][j][r][:]) . W^T + bias), in (B, L, R, K) bf16, K <= 128; the dense-row kernel of
 * cobevt_linear_rows_small_k with the agent mean taken while the A rows are staged.  Weights in MFMA fragment order, the LayerNorm
 * affine folded in.  dims (int64[8]): dtype (0), B, L, R, K, N, ln, act. */
int cobevt_mean_linear_rows_small_k(const void* in, const void* wfrag, const float* bias, void* out, const long* dims,
                                    float ln_eps, hipStream_t stream);

/* cobevt_window_attention with the KEYS of every window shared out over `ksplit` (2..8) workgroups per query tile and a merge
 * pass (same reference code: fax_modules.py:211-237, 137-171): for launches whose grid leaves the chip idle while every query
 * walks a long key list (FAX level 2 / global attention: 1024 keys, 160 workgroups).  Plain inference attention only (no camera
 * mean / pairing, no lse, no dropout); bias table and mask as there.  part_out: [ksplit][out_rows][heads * 32] in the storage
 * dtype, part_lse: fp32 [ksplit][out_rows][heads] - scratch; out_rows = rows of `out` (ld = dims' ldo). */
int cobevt_window_attention_ksplit(const void* q, const void* k, const void* v, void* out, const float* bias_table,
                                   const float* mask, void* part_out, float* part_lse, const int* dims, float scale,
                                   int ksplit, long out_rows, hipStream_t stream);

/*
 * Projection chain (bf16, 128 channels): the key / value side of a FAX cross-view level in one launch per operand -
 *   y = ReLU?(a * pre_scale[c] + pre_shift[c]) . Wp^T + bp + skip     pre-activation BatchNorm -> ReLU -> 1x1 conv (feature_proj /
 *                                                                     feature_linear, fax_modules.py:281-292,377-396) + the ray embedding
 *   next = act(LayerNorm?(y) . Wn'^T + bn')                           to_k / to_v of BOTH cross attentions stacked (fax_modules.py:201-205)
 * with y (the "key" / "val" map of the reference) kept in LDS: it is written to `out` only when out != null.  The row_chain
 * kernel without its MLP phases (csrc/row_chain.hip).  Weights in MFMA fragment order as for cobevt_attn_mlp_chain.
 * dims (int32[9]): dtype (0), M, C (128), Nn (<= 768), next_ln, next_act, skip_rows (0 = M), pre_relu, variant (0 = automatic:
 * maps of >= 32768 rows run as independent waves with the rows in registers and the weights in LDS, csrc/proj_chain128.hip;
 * 1 = always the barrier-phased kernel).
 */
int cobevt_proj_chain(const void* a, const float* pre_scale, const float* pre_shift, const void* skip, const void* wp,
                      const float* bp, void* out, const void* wnext, const float* bnext, void* out_next, const int* dims,
                      float eps_next, hipStream_t stream);

/*
 * Key AND value side of a FAX pyramid level whose image features are wider than the level, one launch (csrc/proj_chain_k.hip):
 *   key = ReLU(BN(feature)) . Wk^T + ray embedding, val = ReLU(BN'(feature)) . Wv^T   (feature_proj / feature_linear, fax_modules.py:281-292,392-396)
 *   kk = LN(key) . [to_k of attention 1 | attention 2]^T, vv = LN(val) . [to_v | to_v]^T            (fax_modules.py:201-205)
 * for K = 256 / 384 / 512 feature channels and 128 level channels; the key / value maps stay in LDS.  a (M, K) bf16.  ptrs: host
 * array of 18 device pointers, side s (0 key, 1 value) at [9 s ..]: pre_scale, pre_shift (float[K] or null), wp (fragment-ordered
 * [4][K/16]), bp (float[128] or null), skip ((skip_rows, 128) bf16 or null), wn (fragment-ordered stacked next projection), bn
 * (float[Nn]), out ((M, 128) or null), out_next (M, Nn).  dims (int32[10]): dtype (0), M, K, Nn, next_ln, nsides (1 | 2),
 * pre_relu_0, skip_rows_0 (0 = M), pre_relu_1, skip_rows_1.
 */
int cobevt_proj_chain_kv(const void* a, const void* const* ptrs, const int* dims, float eps_next, hipStream_t stream);

/*
 * One half of a SwapFusionBlock in ONE launch (bf16 mode, 128 channels = 4 heads of 32): PreNormResidual(Attention) +
 * PreNormResidual(FeedForward) over the window (map mode 0) or dilated-grid (mode 1) partition of (B, L, H, W, 128) agent maps
 * = opv2v/opencood/models/fusion_modules/swap_fusion_modules.py:87-128 (attention with the 3-D relative position bias and the
 * key mask), :126,172-190 (to_out, residuals, rearranges) and base_transformer.py:102-124 (PreNormResidual / FeedForward), plus
 * the LayerNorm + to_qkv (:93) of the NEXT half while the rows are in LDS.  A workgroup carries 32 query tokens of one
 * window / grid group through attention (one wave per head, K rows straight from L2, V^T in LDS) and the row chain; the
 * attention output never reaches memory (csrc/swap_stage.hip).
 * qkv [rows][384] = to_qkv(LayerNorm(x)) (q | k | v), x / out [rows][128], qkv_next [rows][Nn] (nullable with wn / bn);
 * rows = (b, l, h, w).  map (int32[8]): mode, L, H, W, w1, w2, X, Y as for cobevt_window_attention.  bias_table: the
 * relative-position table one column per head, [4][bias_rows rounded up to a multiple of 4] fp32, multiplied by log2(e) (the
 * kernel's softmax runs in the base-2 domain) and zero-padded to 10240 floats (copied whole), bias_rows = (2L-1)(2w1-1)(2w2-1); mask (B, H, W, L) fp32 (0 = key masked out) or null.  Weights in MFMA fragment order as for
 * cobevt_attn_mlp_chain (wp: to_out, w1 / b1: fc1 with the LayerNorm affine folded in, w2 / b2: fc2, wn / bn: next to_qkv
 * with its LayerNorm folded in).  dims (int32[9]): dtype (0), B, C (128), heads (4), Hd, Hdp, Nn, bias_rows, bias_L.
 */
int cobevt_swap_fusion_stage(const void* qkv, const void* x, void* out, void* qkv_next, const int* map,
                             const float* bias_table, const float* mask, const void* wp, const float* bp, const void* w1,
                             const float* b1, const void* w2, const float* b2, const void* wn, const float* bn,
                             const int* dims, float scale, float eps1, float eps_next, hipStream_t stream);

/* ---- training glue, both directions (csrc/train_glue.hip): what torch autograd + cuDNN run between the convolutions under
 * opv2v/opencood/tools/train_camera.py:143-179.  Channels-last maps flattened to (rows, C) / (N, H, W, C); dtype 0 bf16, 1 fp32;
 * C a multiple of 8 (<= 2048) except where noted. --------------------------------------------------------------------------- */

/* sum[c] = sum over rows of (x - shift[c]) and (sumsq nullable; must be sum + C) sumsq[c] = sum of its squares, fp64 (shift nullable
 * = 0; BatchNorm passes its running mean so that the variance is not a cancellation); out_f (nullable): the same values as fp32.
 * scratch: fp64 [scratch_blocks][2][C] for per-workgroup partial sums (no atomics: they serialise on the C result words).
 * BatchNorm batch statistics (torchvision BasicBlock / Bottleneck bn1-3 reached from resnet_ms.py:67-74, fax_modules.py:10,472-489,
 * naive_decoder.py:78-87) and bias gradients (any C: channel counts off the 8-channel piece take a scalar kernel, without shift /
 * out_f). */
int cobevt_channel_sums(const void* x, const float* shift, double* sum, double* sumsq, float* out_f, double* scratch,
                        int scratch_blocks, int dtype, long rows, int C, hipStream_t stream);
int cobevt_f64_to_f32(const double* in, float* out, int n, hipStream_t stream);
/* nn.BatchNorm2d statistics -> per-channel scale / shift (scale = gamma rstd, shift = beta - mean scale), mean / rstd for backward.
 * training != 0: batch statistics from the sums (shifted != 0: they were taken of x - running_mean), running_mean / running_var
 * (nullable) updated in place with `momentum` and the unbiased variance; training == 0: the frozen running statistics.
 * batches_tracked (nullable): the module's int64 num_batches_tracked word, incremented by one. */
int cobevt_bn_finalize(const double* sum, const double* sumsq, const float* gamma, const float* beta, float* running_mean,
                       float* running_var, float* scale, float* shift, float* mean, float* rstd, int C, long rows, float eps,
                       float momentum, int training, int shifted, long long* batches_tracked, hipStream_t stream);
/* cobevt_channel_sums (x and x^2, shifted by running_mean when given) + cobevt_bn_finalize(training = 1) as two launches instead of three:
 * the batch statistics of a training BatchNorm2d incl. the running-stat update; running_mean / running_var nullable together; 8 | C. */
int cobevt_bn_batch_stats(const void* x, const float* gamma, const float* beta, float* running_mean, float* running_var, float* scale,
                          float* shift, float* mean, float* rstd, double* scratch, int scratch_blocks, int dtype, long rows, int C,
                          float eps, float momentum, long long* batches_tracked, hipStream_t stream);
/* y = act(x * scale[c] + shift[c] (+ residual)), act 0 none / 1 ReLU. */
int cobevt_bn_apply(const void* x, const void* residual, const float* scale, const float* shift, void* y, int dtype, long rows,
                    int C, int act, hipStream_t stream);
/* Backward of cobevt_bn_apply behind the statistics: g = dy [y > 0 when act == 1]; dgamma_dbeta (fp64 [2][C], written) = sum g xhat,
 * sum g; grads_f (nullable) the same as fp32; dx = gamma rstd (g - (dbeta + xhat dgamma) / rows) with batch statistics, gamma rstd g
 * with frozen ones; dres (nullable) = g.  scratch: fp64 [scratch_blocks][2][C]. */
int cobevt_bn_backward(const void* x, const void* y, const void* dy, const float* mean, const float* rstd, const float* gamma,
                       double* dgamma_dbeta, float* grads_f, double* scratch, int scratch_blocks, void* dx, void* dres, int dtype,
                       long rows, int C, int act, int training, hipStream_t stream);
/* nn.MaxPool2d(3, 2, 1) backward (resnet_ms.py:70): dx fp32 (N, H, W, C), every element written; first maximum of a window wins. */
int cobevt_maxpool3x3s2_bwd(const void* x, const void* dy, float* dx, int dtype, int N, int H, int W, int C, hipStream_t stream);
/* the same with dx in the maps' own type (dtype 0 bf16 | 1 fp32): one thread per 2 x 2 input block and 8 channels */
/*
 * Training forms of the FAX BEV query (CrossViewSwapAttention, fax_modules.py:344-372), channels-last, d = 128:
 *   v = W grid + bias - c[b, cam];  query[b, cam, pixel] = v / (||v|| + 1e-7) + x[b, pixel]
 * grid (K, H, W) - or (B, K, H, W): the per-camera homogeneous ray directions of the key-side image embedding (fax_modules.py:330-343,
 * K = 4, n = 1, x null) - w (d, K), bias (d) | null, c (B * n, d), x (B, H, W, d) | null, out / dq (B, n, H, W, d), all fp32; n <= 8.
 * dims (int32[8]): B, n, H, W, d, round_bf16 (1 inside a bf16 autocast region: W, grid, W grid + bias and v rounded to bf16 like torch's
 * autocast conv / subtraction), K (2 | 4), per_batch_grid (0 | 1).  The backward WRITES dx (B, H, W, d) (nullable) and ADDS into
 * dw (d, K), dbias (d) | null, dc (B * n, d).
 * What torch autograd does with a K = 2 convolution, sub, norm, div, add and a permute over the (B, n, d, H, W) tensor
 * (train_camera.py:143-179).
 */
int cobevt_fax_bev_query_train(const float* grid, const float* w, const float* bias, const float* c, const float* x, float* out,
                               const int* dims, hipStream_t stream);
int cobevt_fax_bev_query_train_bwd(const float* grid, const float* w, const float* bias, const float* c, const float* dq, float* dx,
                                   float* dw, float* dbias, float* dc, const int* dims, hipStream_t stream);
/* mean over the n slabs of a contiguous (B, n, inner) tensor -> (B, inner) (backward = 0), or its backward: (B, inner) -> (B, n, inner),
 * every slab = in / n (backward = 1); dtype 0 bf16 | 1 fp32, 8 | inner.  The camera mean of CrossWinAttention (fax_modules.py:243) under
 * train_camera.py:143-179. */
int cobevt_group_mean(const void* in, void* out, int dtype, long B, int n, long inner, int backward, hipStream_t stream);
int cobevt_maxpool3x3s2_bwd_t(const void* x, const void* dy, void* dx, int dtype, int N, int H, int W, int C, hipStream_t stream);
/* nn.PixelUnshuffle(2) (fax_modules.py:479) on channels-last maps: inverse 0: (N, 2Ho, 2Wo, C) -> (N, Ho, Wo, 4C); 1: back. */
int cobevt_pixel_unshuffle2_nhwc(const void* in, void* out, int dtype, int N, int Ho, int Wo, int C, int inverse,
                                 hipStream_t stream);
/* nearest x2 up-sampling (naive_decoder.py:84): backward 0: (N, H, W, C) -> (N, 2H, 2W, C); 1: dy (N, 2H, 2W, C) -> dx (N, H, W, C). */
int cobevt_upsample_nearest2_nhwc(const void* in, void* out, int dtype, int N, int H, int W, int C, int backward,
                                  hipStream_t stream);
/* Input gradient of cobevt_sttf_warp (corpbevt.py:28-64, torch_transformation_utils.py:317-355): dout (B, L, H, W, C) fp32 ->
 * dx (agents, H, W, C) fp32 (zeroed by the caller), scattered through the same sample positions; record_len as there. */
int cobevt_sttf_warp_bwd(const float* dout, const float* tmat, const int* record_len, float* dx, int B, int L, int H, int W, int C,
                         float discrete_ratio, float downsample_rate, hipStream_t stream);

/* Operand preparation of the training convolutions (what cuDNN does internally for the reference's nn.Conv2d / nn.Linear under
 * train_camera.py:143-179; csrc/train_prep.hip).  cobevt_conv_weight_rows: fp32 master weight (Cout, Cin, kh, kw) -> the weight rows
 * cobevt_conv2d_nhwc reads, in the compute type dims[0] (0 bf16, 1 fp32): rows_fwd [Cout][Kpad_fwd] with k = (r * kw + s) * Cin + c,
 * rows_dgrad [Cin][Kpad_dgrad] with k = (r * kw + s) * Cout + o of w[o][c][kh-1-r][kw-1-s] (the input gradient = the same
 * convolution, taps flipped, channel roles swapped); columns past K are zero; either output may be null.
 * dims: [dtype, Cout, Cin, kh, kw, Kpad_fwd, Kpad_dgrad]. */
int cobevt_conv_weight_rows(const float* w, void* rows_fwd, void* rows_dgrad, const int* dims, hipStream_t stream);
/*
 * The same for a 3x3 / pad-1 convolution with 64 | channel counts that runs on the inference kernels in training (bf16): the fp32
 * master weight (Cout, Cin, 3, 3) -> `frag` (cobevt_conv3x3_wfrag_nhwc's fragment table, [Op/32][I/64][9][4][64 lanes][8], Op = O
 * rounded up to 128) and / or `rows3` (cobevt_conv3x3_nhwc's [O][I/64][9][64]); either nullable.  dims (int32[3]): Cout, Cin, dgrad.
 * dgrad = 0: the forward convolution (O = Cout, I = Cin); dgrad = 1: its input gradient, i.e. the convolution with the channel
 * roles swapped and the taps flipped (O = Cin, I = Cout) - what cuDNN's backward-data does under train_camera.py:143-179.
 */
int cobevt_conv3_weight_operands(const float* w, void* frag, void* rows3, const int* dims, hipStream_t stream);
/* ... both directions from one launch: outs[4] = {forward frag, forward rows3, input-gradient frag, input-gradient rows3}, each nullable;
 * dims (int32[2]): Cout, Cin. */
int cobevt_conv3_weight_operands2(const float* w, void* const* outs, const int* dims, hipStream_t stream);
/* The same for a dense projection / 1x1 stride-1 convolution on the inference row-GEMM kernel (cobevt_linear_rows_small_k) in training:
 * fp32 master weight (N, K) -> the bf16 fragment tables of cobevt_linear_rows_small_k ([Rp/32][Cp/16][64 lanes][8], both padded to 128):
 * `frag` for the projection (rows N, contraction K), `frag_t` for its input gradient dx = dy W (rows K, contraction N); either nullable,
 * one launch.  dims (int32[2]): N, K. */
int cobevt_linear_weight_frags(const float* w, void* frag, void* frag_t, const int* dims, hipStream_t stream);
/*
 * Weight gradient of a 3x3 / stride-1 / pad-1 convolution from the channels-last bf16 maps themselves (csrc/wgrad3.hip: the
 * pixel-major operands of the matrix instruction come out of LDS through ds_read_b64_tr_b16, no blocked copies, no atomics):
 * dw (Cout, Cin, 3, 3) fp32 is WRITTEN (not accumulated).  x (N, H, W, Cin), dy (N, H, W, Cout) bf16, 32 | Cin, 32 | Cout, 16 | W.
 * cobevt_conv_wgrad3_chunks(dims[5]: N, H, W, Cin, Cout) returns the number of partial copies of dw the launch writes (scratch must hold
 * that many x Cout * Cin * 9 floats), or minus an error code when the shape is not served (the caller then takes
 * cobevt_conv_wgrad_blocked); cobevt_conv_wgrad3 takes dims (int32[6]): N, H, W, Cin, Cout, that chunk count.
 * cuDNN's backward-filter under train_camera.py:143-179.
 */
int cobevt_conv_wgrad3_chunks(const int* dims);
int cobevt_conv_wgrad3(const void* x, const void* dy, float* dw, float* scratch, const int* dims, hipStream_t stream);
/*
 * The same for a dense projection / 1x1 stride-1 convolution: dw (Cout, Cin) fp32 = dy^T x over R bf16 rows (x (R, Cin), dy (R, Cout),
 * 128 | Cin, 128 | Cout), written, not accumulated - nn.Linear's backward-filter under train_camera.py:143-179.
 * cobevt_linear_wgrad_chunks(dims[3]: R, Cin, Cout) -> partial copies of dw in `scratch` (or minus an error code);
 * cobevt_linear_wgrad dims (int64[4]): R, Cin, Cout, that chunk count.
 */
int cobevt_linear_wgrad_chunks(const long* dims);
int cobevt_linear_wgrad(const void* x, const void* dy, float* dw, float* scratch, const long* dims, hipStream_t stream);
/* bf16 channels-last map (N, H, W, C) -> the blocked operand of cobevt_conv_wgrad_blocked, dst [N][Hp][NB][C][8]: pixel x of input
 * row y sits in block (x + pad_left) / 8 of row y + pad_top; everything else is zero.  In general dst is [N][Hp][NB][P][C][8] and slot j
 * of plane q of block b holds input pixel sx (8 b + j) + q - pad_left (P = 1, sx = 1 above; P = 2, sx = 2: even / odd columns of a
 * stride-2 convolution; P = k, sx = stride: the k tap columns of a tap row as planes).
 * dims: [N, H, W, C, Hp, NB, pad_top, pad_left, P, sx]. */
int cobevt_wgrad_block_operand(const void* src, void* dst, const int* dims, hipStream_t stream);

/* Backward pieces of the nuScenes SinBEVT training path (csrc/train_nusc.hip; torch autograd + cuDNN in the reference:
 * nuscenes/cross_view_transformer/model/model_module.py:35-60 over backbones/efficientnet.py:85-96, decoder.py:27-36, losses.py:27-84).
 * cobevt_swish: dy == NULL: out = x sigmoid(x) (efficientnet-pytorch's MemoryEfficientSwish); else out = dy d/dx; n elements (multiple of 8),
 * dtype 0 bf16 / 1 fp32. */
int cobevt_swish(const void* x, const void* dy, void* out, int dtype, long n, hipStream_t stream);
/* Weight gradient of the depthwise k x k convolution (k = 3 / 5): dw fp32 [k * k][C] += sum over output pixels of dy (N, Ho, Wo, C) x the
 * tap-shifted x (N, H, W, C); dw zero-initialised.  dims (int32[11]): dtype, N, H, W, C, k, stride, pad_top, pad_left, Ho, Wo.  (The input
 * gradient is cobevt_depthwise_conv_nhwc on the flipped taps, on the zero-stuffed gradient when strided.) */
int cobevt_depthwise_wgrad(const void* x, const void* dy, float* dw, const int* dims, hipStream_t stream);
/* Adjoint of cobevt_resize_nhwc mode 1 (align_corners bilinear; nn.Upsample of decoder.py:12): dy (N, Ho, Wo, C) -> dx fp32 (N, H, W, C),
 * zero-initialised. */
int cobevt_resize_bilinear_bwd(const void* dy, float* dx, int dtype, int N, int H, int W, int C, int Ho, int Wo, hipStream_t stream);
/* Gradient of cobevt_sigmoid_focal_loss's mean w.r.t. the logits: dpred (N, C, hw) fp32 = gscale[0] / count x d loss / d logit on the kept
 * elements, 0 elsewhere.  stats = the forward's out[3] (mean, sum, count) and gscale (the upstream gradient) are DEVICE words: no host sync. */
int cobevt_sigmoid_focal_loss_bwd(const float* pred, const float* label, const unsigned char* visibility, const unsigned int* label_mask,
                                  const float* stats, const float* gscale, float* dpred, int N, int C, int NL, int hw, int min_visibility,
                                  float alpha, float gamma, int soft_label, hipStream_t stream);

/* ---- multi-GPU: the V2V feature-sharing step in front of FuseBEVT.  The reference keeps all agents in one
 * process (opv2v/opencood/models/corpbevt.py:112-124, sub_modules/fuse_utils.py:8-61: agents are a batch dimension up to
 * `regroup`); its only collective call sites are the DDP set-up in opv2v/opencood/tools/multi_gpu_utils.py:32-37 and
 * train_camera.py:105-110.  With one agent per GPU the (32, 32, 128) per-agent blocks (256 KiB bf16) are exchanged once per
 * frame: either by RCCL (torch.distributed.all_gather_into_tensor, cobevt_amd/dist.py) or by the one-shot direct exchange
 * below - every rank owns a WINDOW of uncached device memory mapped by all peers through hipIpc, and one launch pair per
 * frame stores the local blocks straight into the destination windows over xGMI (csrc/peer_gather.hip). ------------------ */

/* Allocate this rank's window: `bytes` of block storage (multiple of 16) followed by the flag words, zero-filled; writes the
 * device pointer and the 64-byte hipIpc handle peers open with cobevt_peer_window_open.  Synchronous (set-up time only). */
int cobevt_peer_window_alloc(long bytes, void** dptr, void* handle64);
/* Map a peer's window into this process (hipIpcOpenMemHandle with lazy peer access). */
int cobevt_peer_window_open(const void* handle64, void** dptr);
int cobevt_peer_window_close(void* dptr);
int cobevt_peer_window_free(void* dptr);
/* Synchronise `stream`, then read the local window's status (0 = every bounded wait so far completed, 1 = a peer's
 * "window free" acknowledgement timed out, 2 = a peer's data-ready flag timed out) and the count of completed exchanges. */
int cobevt_peer_window_status(const void* window, long bytes, int* status, int* epoch, hipStream_t stream);
/* The same read in stream order, no synchronisation: the window's 32 flag words (status at [18], completed exchanges at [16]) are copied
 * to `host_words` (pinned host memory); look at them after an event recorded behind this call has completed. */
int cobevt_peer_window_status_async(const void* window, long bytes, unsigned int* host_words, hipStream_t stream);
/* One exchange, two launches on `stream` (capturable in a HIP graph: the epoch lives in the window).  windows: HOST array
 * of `world` (<= 8) device pointers = this process's mappings of every rank's window, the own window at [rank].  local:
 * n_local (<= 16) contiguous blocks of block_bytes; block j is stored at block slot dest_block[j] of rank dest_rank[j]'s
 * window, or of every rank's window when dest_rank[j] < 0 (all-gather).  When the second launch retires, every block
 * addressed to this rank has landed in its window.  spin_limit: bound on the flag polls (<= 0: default, ~30 s). */
int cobevt_peer_exchange(const void* local, void* const* windows, int world, int rank, int n_local, long block_bytes,
                         const int* dest_rank, const int* dest_block, long window_bytes, long spin_limit,
                         hipStream_t stream);

/*
 * Box calibration (bench.py `box_calibration`; no reference counterpart - the reference's benchmark protocol,
 * nuscenes/scripts/benchmark.py:42-55, reports wall time only).  cobevt_calibrate_mfma: `blocks` workgroups of 4 waves issue
 * 4 * iters independent v_mfma_f32_32x32x16_bf16 each (no operand traffic): 2 * 32*32*16 * 16 * iters flops per workgroup;
 * clk[0] / clk[1] (device int64[2]) = shader-clock / 100-MHz wall-clock ticks workgroup 0 spent in the loop.  out: fp32
 * [blocks * 256].  cobevt_calibrate_copy: streaming copy of `bytes` (multiple of 16) src -> dst.
 */
int cobevt_calibrate_mfma(float* out, long long* clk, int blocks, int iters, hipStream_t stream);
int cobevt_calibrate_copy(const void* src, void* dst, long bytes, hipStream_t stream);
/* rate of the wall clock behind clk[1] and the device's maximum shader clock (kHz), of the current device */
int cobevt_calibrate_clock_khz(int* wall_khz, int* sclk_max_khz);

#ifdef __cplusplus
}
#endif
#endif /* COBEVT_HIP_H */
