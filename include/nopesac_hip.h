/*
 * nopesac_hip.h — C ABI of libnopesac_hip.so: the MI355X (gfx950) kernels behind NopeSAC's inference
 * hot path.
 *
 * The reference (IceTTTb/NopeSAC) is pure Python on torch/cuDNN and has NO native interface
 * (SURVEY.md fact 1); this ABI is therefore build-defined (SURVEY.md §8b last row).  Each entry point
 * names the reference computation it replaces (path:line under NopeSAC_Net/modeling/).  The host side
 * that binds it is nopesac_amd/_lib.py (ctypes); INTEGRATION.md shows the binding a maintainer of the
 * reference would add.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer on the current device (tensor.data_ptr()); the caller owns all
 *    buffers and allocates outputs; nothing is retained after return;
 *  - `stream` is a hipStream_t passed as void* (torch.cuda.current_stream().cuda_stream); all work is
 *    enqueued on it, no call synchronises, so every call is capturable into a hipGraph;
 *  - return value 0 = enqueued, <0 = argument error (NPS_E_*), >0 = hipError_t from the launch;
 *  - activations are NHWC ("pixels x channels", channels contiguous); conv weights are
 *    [Cout][KH][KW][Cin] (K-contiguous rows); token matrices are [rows][features] row-major;
 *  - dtype codes: 0 = f32, 1 = bf16.  Accumulation is always f32.
 *  - ragged per-pair sizes (n1, n2, m) live in int32 device arrays; padded rows are ignored/zeroed.
 */
#ifndef NOPESAC_HIP_H
#define NOPESAC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define NPS_DT_F32 0
#define NPS_DT_BF16 1
#define NPS_DT_F32_BF16W 2 /* conv in_dt only: x is f32 in memory, w is bf16; x is rounded to bf16 while staged */
#define NPS_DT_FP8 3       /* OCP e4m3fn bytes (gfx950's fp8).  As out_dt: the epilogue result is saturated to +-448 and rounded to
                            * nearest even; only with a channel count / stride that is a multiple of 8 and no residual */

#define NPS_ACT_NONE 0
#define NPS_ACT_RELU 1
#define NPS_ACT_LEAKY 2   /* LeakyReLU(0.01), camera_modules.py:47 */
#define NPS_ACT_SIGMOID 3
#define NPS_ACT_RES_AFTER 0x100 /* OR-able flag: add `residual` AFTER the activation (planeTR_head.py:244-250) */
#define NPS_ACT_BIAS_BATCHED 0x200 /* OR-able flag, batched weights only: `bias` is [B][Cout] (one row per batch entry) */

#define NPS_E_ARG (-1)
#define NPS_E_UNSUPPORTED (-2)

/* library / ABI version (major*10000 + minor*100 + patch) */
int nopesac_version(void);
/* last error string of this thread's most recent failing call ("" if none) */
const char* nopesac_last_error(void);

/* ---------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / linear layer with fused epilogue (MFMA).
 *   y[b,oh,ow,n] = act( (sum_{kh,kw,c} x[b, oh*s-p+kh, ow*s-p+kw, c] * w[n,kh,kw,c]) * scale[n] + bias[n]
 *                       + residual[b,oh,ow,n] )        (or act(...) + residual with NPS_ACT_RES_AFTER)
 * Replaces every torch conv2d/linear(+BN/bias/residual/activation) on the path: d2 ResNet-50
 * (meta_arch/siamese_planeTR.py:456), planeTR_net/planeTR_head.py:126,148-162,209-215,
 * camera_net/camera_modules.py:36-48,271-321, camera_net/camera_head.py:957-962,983-990,
 * transformer linears, matching_net/matching_head.py:101-113.
 *   in_dt/out_dt : NPS_DT_*; weights have dtype in_dt (bf16 for NPS_DT_F32_BF16W).
 *   x_cstride / y_cstride / r_cstride : elements between consecutive pixels (>= channels) so that
 *       inputs/outputs can live inside wider (concatenated) buffers.
 *   w_bstride : elements between per-image weight sets (0 = shared weights).  With w_bstride != 0
 *       the call is a batched GEMM  y[b] = x[b] * w[b]^T  (mask einsum planeTR_head.py:150, attention-
 *       free correlations camera_head.py:1128, descriptor scores matching_head.py:113).
 *   scale/bias : f32[Cout] or NULL; residual: out_dt NHWC or NULL.
 */
int nopesac_conv2d_nhwc(const void* x, const void* w, const float* scale, const float* bias,
                        const void* residual, void* y,
                        int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                        int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int64_t w_bstride,
                        int act, int in_dt, int out_dt, void* stream);

/* Same, with an explicit kernel configuration (chosen per layer shape by the load-time autotuner of
 * nopesac_amd/ops.py; every configuration computes the same result): NPS_CONV_AUTO = built-in heuristic,
 * T128 / T64 = register-staged 128x128 / 64x64 tiles, DMA64 / DMA32 = LDS-DMA kernel with BK = 64 / 32
 * (bf16 with Cin % 64 == 0 only; silently falls back to the heuristic when not applicable). */
#define NPS_CONV_AUTO 0
#define NPS_CONV_T128 1
#define NPS_CONV_T64 2
#define NPS_CONV_DMA64 3
#define NPS_CONV_DMA32 4
int nopesac_conv2d_nhwc_ex(const void* x, const void* w, const float* scale, const float* bias,
                           const void* residual, void* y,
                           int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int64_t w_bstride,
                           int act, int in_dt, int out_dt, int kernel_cfg, void* stream);

/* bf16 conv with the weights in MFMA FRAGMENT-MAJOR order ([Cout][KH*KW*Cin] re-ordered as in nopesac_bottleneck_tail_bf16):
 * activations go through a 3-deep LDS-DMA ring (variant 3: K-tile 64, 3 workgroups/CU; variant 32: K-tile 32, 4-deep ring, 4 workgroups/CU),
 * every wave streams the weight fragments of its own 32 output channels from L2 (csrc/conv_igemm.hip, conv_igemm_bfrag_kernel).  Same semantics as nopesac_conv2d_nhwc for x / w bf16;
 * needs Cin % 64 == 0, Cout % 128 == 0; act may carry NPS_ACT_RES_AFTER; out_dt may also be NPS_DT_FP8 (no residual). */
int nopesac_conv2d_nhwc_bfrag(const void* x, const void* w_frag, const float* scale, const float* bias, const void* residual,
                              void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                              int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant,
                              void* stream);

/* bf16 conv on 256 x 256 x 64 workgroup tiles (csrc/conv_p8.hip, conv_igemm_p8_kernel): 8 waves, both operands through LDS-DMA,
 * the two waves of a SIMD alternate LDS-read / MFMA phases, counted vmcnt across barriers.  For the MFMA-bound layers: twice the
 * FLOPs per byte pulled from L2 of the 128 x 128 kernels.  x / w as for nopesac_conv2d_nhwc with in_dt BF16 (w = plain
 * [Cout][KH][KW][Cin] rows, NOT fragment-major); needs Cin % 64 == 0, Cout % 256 == 0, x_cstride % 8 == 0; act may carry
 * NPS_ACT_RES_AFTER; out_dt F32 / BF16 / FP8 (FP8: no residual).  variant: 0 = static wave priority for the lagging half (default),
 * 1 = priority raised around every MFMA cluster, 2 = no priority hints (tuning aid; results are identical).
 * Replaces: the same reference convolutions as nopesac_conv2d_nhwc (detectron2 Conv2d + FrozenBN + ReLU; camera_modules.py conv stacks). */
int nopesac_conv2d_nhwc_p8(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                           int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                           int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant, void* stream);

/* The same kernel with STREAM-K work distribution (conv_igemm_p8_kernel<.., SK>; round 5): the K-tiles of each XCD's run of output tiles
 * are cut evenly over that XCD's persistent workgroups, so a layer of 300 (150) tiles costs 1.17 (0.59) tile times on 256 CUs instead of
 * 2 (1) rounds.  A tile whose K range is shared is completed by whichever contributor arrives LAST on the tile's counter (nobody waits for
 * another workgroup); partial accumulators travel through `workspace` as f32 slabs and are summed in K order - results do not depend on
 * the arrival order, and equal the plain kernel's up to the K-split's summation order.
 * workspace: nopesac_conv2d_p8_sk_workspace_bytes() bytes, 16-byte aligned; its first 16 KB (arrival counters) must be ZERO on entry and
 * are zero again on completion - one workspace serves all launches of ONE stream.  variant: as nopesac_conv2d_nhwc_p8 (24 not allowed).
 * Replaces: the same reference convolutions as nopesac_conv2d_nhwc_p8. */
int64_t nopesac_conv2d_p8_sk_workspace_bytes(void);
int nopesac_conv2d_nhwc_p8_sk(const void* x, const void* w, const float* scale, const float* bias, const void* residual, void* y,
                              int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                              int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant,
                              void* workspace, int64_t workspace_bytes, void* stream);

/* The 256 x 128-tile sibling for layers with Cout % 128 == 0 that the 256-wide kernel cannot take (csrc/conv_p8n.hip,
 * conv_igemm_p8n_kernel; round 5): the 3x3 128 -> 128 convs of res3 and of the top-down / pose-net stacks.  Same structure (persistent
 * 8-wave workgroups, LDS-DMA operands, counted vmcnt, staggered wave groups), two 8-MFMA phases per K-tile, a three-deep K-tile ring.
 * bf16 in / bf16 out, y = act(conv * scale + bias) with act in {NPS_ACT_NONE, NPS_ACT_RELU, NPS_ACT_LEAKY}, no residual; scale / bias may
 * be NULL; Cin % 64 == 0, x_cstride % 8 == 0, y_cstride % 8 == 0; variant: 0, + 32 = channel-major K order.
 * Replaces: the same reference convolutions as nopesac_conv2d_nhwc (detectron2 BottleneckBlock conv2 + FrozenBN + ReLU, Base.yaml:2-12;
 * camera_modules.py:271-321 pixel-decoder convs). */
int nopesac_conv2d_nhwc_p8n(const void* x, const void* w, const float* scale, const float* bias, void* y, int B, int H, int W,
                            int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride, int act,
                            int variant, void* stream);
/* The same kernel with SPLIT-K work units (round 6) for layers with few tiles and a long K loop - the pose net's first conv (3x3, 2048 -> 128
 * at 15 x 20: 75 tiles x 288 K-tiles; reference camera_head.py:97-112 convs_trans / convs_rots are fed by it through the correlation, the
 * layer itself is the pixel decoder's `layer_3` output conv, camera_modules.py:271-321).  A unit = (tile, slice of the 64-channel groups);
 * partial tiles leave as f32 into workspace[splits][M][Cout] (M = B * OH * OW; >= splits * M * Cout * 4 bytes, 16-byte aligned, < 2 GB) and a
 * second launch on the same stream sums the slices in fixed order (deterministic) and applies the epilogue.  variant must carry + 32
 * (channel-major K order); 2 <= splits <= min(Cin / 64, 16).  Everything else as nopesac_conv2d_nhwc_p8n. */
int nopesac_conv2d_nhwc_p8n_splitk(const void* x, const void* w, const float* scale, const float* bias, void* y, int B, int H, int W,
                                   int Cin, int Cout, int KH, int KW, int stride, int pad, int64_t x_cstride, int64_t y_cstride, int act,
                                   int variant, int splits, void* workspace, int64_t workspace_bytes, void* stream);

/* ---- a stack of Linear(+bias)(+activation) layers in ONE launch (csrc/mlp_chain.hip) --------------------------------------------
 * Replaces one nopesac_conv2d_nhwc launch per layer for the row-wise MLP stacks of the heads in bf16 mode (reference:
 * camera_net/camera_head.py:957-990 geo_encoder / geo_proj_s1 / decoder_rot / geo_proj_s2 / decoder_tran / decoder_rot2 /
 * decoder_tran2 / rots / trans; :685-735 rot_emb_proj / trans_emb_proj; :1090-1100 normal_score_proj / param_score_proj).
 * Row r of the chain input = [ xb[r / xb_rows_per][0 .. xb_width) | x[r][0 .. x_width) ] (the broadcast prefix is optional:
 * xb_width = 0).  Layer l: y = act(y_prev . W_l^T + bias_l); f32 accumulation, operands rounded to bf16 (RNE) exactly where the
 * per-layer launches round them.  A layer with `out` != NULL also writes its f32 output rows to out[r * out_ld + n] (n < N); the
 * last layer must.  Weights: bf16, packed for K padded to nopesac_mlp_padded_k(N, K) and N padded to a multiple of 32 (zeros) in
 * MFMA fragment-major order [Np/32][Kp/16][64 lanes][8] (nopesac_amd.ops.mlp_pack); bias: f32[Np] (zero padded) or NULL.
 * Limits: chain input <= NOPESAC_MLP_MAX_IN columns, every N <= NOPESAC_MLP_MAX_WIDTH, <= NOPESAC_MLP_MAX_LAYERS layers. */
#define NOPESAC_MLP_MAX_IN 1280
#define NOPESAC_MLP_MAX_WIDTH 1024
#define NOPESAC_MLP_MAX_LAYERS 12
#define NOPESAC_MLP_RESTART 1   /* nopesac_mlp_layer.reserved: this layer reads the chain INPUT again instead of the previous layer's
                                 * output - several stacks over the same rows (e.g. the plane head's embedding / prob / param / center
                                 * MLPs, planeTR_head.py:170-188) run in one launch */
typedef struct nopesac_mlp_layer {
    const void* w;
    const float* bias;
    float* out;
    int64_t out_ld;
    int K, N, act, reserved;
} nopesac_mlp_layer;
typedef struct nopesac_mlp_chain {
    const float* x;
    int64_t x_ld;
    const float* xb;
    int64_t xb_ld;
    int x_width, xb_width, xb_rows_per, rows, n_layers, reserved;
    nopesac_mlp_layer layers[NOPESAC_MLP_MAX_LAYERS];
} nopesac_mlp_chain;
int nopesac_mlp_padded_k(int N, int K);
int64_t nopesac_mlp_packed_elems(int N, int K);
int nopesac_mlp_chain_bf16(const nopesac_mlp_chain* chain, void* stream);

/* fp8 (OCP e4m3fn) conv on the gfx950 K=64 fp8 MFMA (v_mfma_f32_32x32x64_f8f6f4, unit block scales; 2x the bf16 MFMA rate): x is
 * fp8 NHWC, w_frag8 the fp8 [Cout][KH*KW*Cin] matrix in the fp8 fragment-major order
 *   [Cout/32][K/64][2][64][16]:  byte j of piece h of lane l = w[nt*32 + (l & 31)][kf*64 + 32*(l >> 5) + 16*h + j]
 * (ops.mfma_fragment_major_fp8).  Accumulation is f32; `scale` must already contain the de-quantisation factors
 * (bn_scale[n] * w_scale[n] * x_scale) - the kernel is otherwise nopesac_conv2d_nhwc_bfrag (same structure: activations through an
 * LDS-DMA ring - half the bytes per K step -, weights streamed from L2).  out_dt: F32, BF16 or FP8.  residual (if any) has out_dt's
 * type (not allowed with FP8).  Needs Cin % 64 == 0, Cout % 128 == 0, x_cstride % 16 == 0.  variant 3: K-tile 128 (Cin % 128 == 0),
 * variant 32: K-tile 64.  Replaces: the reference has no reduced-precision path; BASELINE config 5 ("fp8 MFMA backbone weights"). */
int nopesac_conv2d_nhwc_fp8(const void* x, const void* w_frag8, const float* scale, const float* bias, const void* residual,
                            void* y, int B, int H, int W, int Cin, int Cout, int KH, int KW, int stride, int pad,
                            int64_t x_cstride, int64_t y_cstride, int64_t r_cstride, int act, int out_dt, int variant,
                            void* stream);

/* bf16 3x3 / stride 1 / pad 1 conv with 64 -> 64 channels + folded BN + activation (conv2 of the res2 bottlenecks): 16x16 pixel
 * tiles computed out of an 18x18 halo kept in LDS (csrc/conv3x3_c64.hip).  x, y bf16 NHWC [B,H,W,64]; w_frag = the
 * [64][3*3*64] weight matrix in MFMA fragment-major order. */
int nopesac_conv3x3_c64_bf16(const void* x, const void* w_frag, const float* scale, const float* bias, void* y, int B, int H, int W,
                             int act, void* stream);

/* bf16 3x3 / stride 1 / pad 1 conv, Cin % 64 == 0, Cout % 128 == 0, + folded BN + activation, computed from LDS halo tiles
 * (csrc/conv3x3_halo.hip): tile 0 = 16x16 pixels, 1 = 16 rows x 8 columns per workgroup.  x [B,H,W,Cin], y [B,H,W,Cout] bf16
 * (pixel-dense), w_frag = the [Cout][3*3*Cin] weights in MFMA fragment-major order. */
int nopesac_conv3x3_halo_bf16(const void* x, const void* w_frag, const float* scale, const float* bias, void* y, int B, int H, int W,
                              int Cin, int Cout, int act, int tile, void* stream);

/* Fused bf16 ResNet stem: y = maxpool3x3/s2/p1( relu( bn( conv7x7/s2/p3(x) ) ) ) in one kernel (d2 BasicStem).
 *   x bf16 NHWC [B,H,W,4] (RGB + zero pad channel); w bf16 [64][7][8][4] (kw padded 7 -> 8 with zeros, i.e. 224 per
 *   output channel); scale/bias f32[64] (folded FrozenBN); y bf16 NHWC [B,PH,PW,64]. */
int nopesac_stem_fused_bf16(const void* x, const void* w, const float* scale, const float* bias, void* y,
                            int B, int H, int W, void* stream);
/* Same, fed directly with the f32 NCHW images [B,3,H,W] of preprocess_image (siamese_planeTR.py:534-542): the normalisation
 * (v - mean[c]) / std[c] and the rounding to bf16 happen while the input patch is staged, bit-identical to
 * nopesac_preprocess_nchw_to_nhwc followed by nopesac_stem_fused_bf16. */
int nopesac_stem_fused_raw_bf16(const float* x_nchw, const float* mean, const float* std, const void* w, const float* scale,
                                const float* bias, void* y, int B, int H, int W, void* stream);
/* Raw-image stem with the normalisation folded into its operands (round 4): the staged patch holds v - 128, exact in bf16 for the
 * 8-bit pixel values preprocess_image receives, so the input operand carries no rounding error (the normalised image of the entry
 * above is rounded to 8 significant bits).  The caller folds: w_folded[o][kh][kw][c] = bf16(w[o][kh][kw][c] / std[c]),
 * bias_folded[o] = bias[o] + scale[o] * sum_{kh,kw,c} (w / std[c]) * (128 - mean[c]), pad3[c] = mean[c] - 128 (the raw value whose
 * folded contribution is the reference's zero padding of the NORMALISED image; siamese_planeTR.py:534-542 + d2 BasicStem). */
int nopesac_stem_fused_raw_shifted_bf16(const float* x_nchw, const float* pad3, const void* w_folded, const float* scale,
                                        const float* bias_folded, void* y, int B, int H, int W, void* stream);

/* Fused tail of a bf16 ResNet bottleneck (d2 BottleneckBlock.forward: conv3 + shortcut + ReLU) plus, optionally, the NEXT
 * block's 1x1 reduce conv, in one launch (all tensors bf16 NHWC, pixel-dense; FrozenBN as f32 scale/bias):
 *   y = relu(scale3 * (w3 . b) + bias3 + shortcut),  shortcut = residual [B,OH,OW,C4]                      (identity block)
 *                                                    or scale_sc * (wsc . x2[:, ::s, ::s]) + bias_sc        (projection block)
 *   o = relu(scale1 * (w1 . y) + bias1)  if CN > 0   [B,OH,OW,CN]
 * b [B,OH,OW,C], x2 [B,x2_H,x2_W,C2].  Exactly one of residual / x2 is non-NULL.
 * The three weight matrices w3 [C4][C], wsc [C4][C2], w1 [CN][C4] are passed FRAGMENT-MAJOR: a [N][K] matrix is stored as
 * [N/32][K/16][2][32][8], i.e. element (n, k) at ((((n/32)*(K/16) + k/16)*2 + (k%16)/8)*32 + n%32)*8 + k%8 - the order in which
 * the 64 lanes of a wave consume it as MFMA operands, so every weight load is 1 KB contiguous (nopesac_amd.ops.mfma_fragment_major).
 * Supported (C, C4, CN, C2): (64,256,{0,64,128},0), (64,256,{0,64},64), (128,512,{0,128,256},0), (128,512,{0,128},256),
 * (256,1024,{0,256,512},0), (256,1024,{0,256},512);
 * anything else returns an error (the caller then uses
 * nopesac_conv2d_nhwc per layer). */
int nopesac_bottleneck_tail_bf16(const void* b, const void* w3, const float* scale3, const float* bias3, const void* residual,
                                 const void* x2, const void* wsc, const float* scale_sc, const float* bias_sc, int B, int OH, int OW,
                                 int x2_H, int x2_W, int x2_stride, int C, int C4, int C2, void* y, const void* w1,
                                 const float* scale1, const float* bias1, int CN, void* o, void* stream);
/* Same; o_dt = NPS_DT_BF16, or NPS_DT_FP8: o is written as OCP e4m3fn bytes (the next block's 3x3 conv then runs on
 * nopesac_conv2d_nhwc_fp8; scale1 / bias1 must already contain the 1 / x_scale of that conv's input quantisation). */
int nopesac_bottleneck_tail_bf16_ex(const void* b, const void* w3, const float* scale3, const float* bias3, const void* residual,
                                    const void* x2, const void* wsc, const float* scale_sc, const float* bias_sc, int B, int OH, int OW,
                                    int x2_H, int x2_W, int x2_stride, int C, int C4, int C2, void* y, const void* w1,
                                    const float* scale1, const float* bias1, int CN, void* o, int o_dt, void* stream);

/* (x - mean[c]) / std[c], NCHW f32 -> NHWC (C padded with zeros to Cpad), out_dt f32/bf16.
 * siamese_planeTR.py:85-89,534-542. */
int nopesac_preprocess_nchw_to_nhwc(const float* x, void* y, const float* mean, const float* std,
                                    int B, int C, int H, int W, int Cpad, int out_dt, void* stream);

/* max-pool K x K / stride / pad on NHWC (d2 BasicStem 3/2/1; camera_head.py:81-87 2/2/0). */
int nopesac_maxpool_nhwc(const void* x, void* y, int B, int H, int W, int C, int K, int stride, int pad,
                         int dt, void* stream);

/* y = act(bilinear_x2(x) [align_corners=False]) (+ addend) ; planeTR_head.py:225,246-250. */
int nopesac_upsample2x_bilinear_nhwc(const void* x, const void* addend, void* y, int B, int H, int W, int C,
                                     int act, int dt, void* stream);
/* y = lateral + nearest_x2(x) ; camera_modules.py:346. */
int nopesac_upsample2x_nearest_add_nhwc(const void* x, const void* lateral, void* y, int B, int H, int W,
                                        int C, int dt, void* stream);

/* GroupNorm(G groups, eps) [+ReLU] on NHWC, per image; camera_modules.py:271-303 (d2 get_norm "GN").
 * workspace: f32[B*16*G*2] scratch for the split statistics (NULL selects the slower single-kernel path). */
int nopesac_groupnorm_nhwc(const void* x, const float* gamma, const float* beta, void* y,
                           int B, int HW, int C, int G, float eps, int act, int dt, float* workspace, void* stream);

/* y = LayerNorm(x [+ res]) * gamma + beta over the last dim D (<= 1024, multiple of 64);
 * optional second output y2 = y + addend (addend row index = row % addend_rows), used for the
 * "with_pos_embed" sums of transformer/transformer.py:180-199,304-321.  f32 only. */
int nopesac_layernorm(const float* x, const float* res, const float* gamma, const float* beta,
                      float* y, const float* addend, int addend_rows, float* y2,
                      int rows, int D, float eps, void* stream);

/* Same, with optional bf16 copies of both results (y_bf16, y2_bf16) for the GEMMs that consume them; any of the four
 * outputs may be NULL (at least one must not be). */
int nopesac_layernorm_ex(const float* x, const float* res, const float* gamma, const float* beta,
                         float* y, const float* addend, int addend_rows, float* y2, void* y_bf16, void* y2_bf16,
                         int rows, int D, float eps, void* stream);

/* out = a + b (b row index = row % b_rows); f32. */
int nopesac_add_rows(const float* a, const float* b, float* out, int rows, int D, int b_rows, void* stream);

/* rows[b] = [t(3) | q(4) | n1 | n2 | m | t_err | r_err | pair_idx0 + b | nonfinite[0] | 0 | 0] as f32 [B][16]: the per-pair result rows
 * the runner all-gathers (the reference gathers the same numbers as python objects over Gloo, mp3d_evaluation.py:316-319).  t_err, r_err
 * and nonfinite (device int32[1]) may be NULL.  One launch, no host synchronisation. */
int nopesac_metric_rows(const float* trans, const float* rot, const int32_t* n1, const int32_t* n2, const int32_t* m,
                        const float* t_err, const float* r_err, const int32_t* nonfinite, int pair_idx0, float* rows, int B, void* stream);

/* out[r] = [a[r] | b[r]] (f32 rows of Da and Db elements): the 7-vector (t, q) the matcher takes (camera_head.py:455). */
int nopesac_concat_cols(const float* a, int Da, const float* b, int Db, float* out, int rows, void* stream);

/* a and a + b as bf16 (a_bf16, ab_bf16: [rows, D], D % 4 == 0): the operands of the first encoder layer's v and q|k projections
 * (src and src + pos, transformer.py: TransformerEncoderLayer.forward_post) in one pass over the f32 rows. */
int nopesac_add_rows_bf16(const float* a, const float* b, void* a_bf16, void* ab_bf16, int rows, int D, int b_rows, void* stream);

/* row softmax over the last dim (<= 1024); camera_head.py:1131 (after the NHWC re-layout the 300
 * view-2 positions are the channel dim). f32. */
int nopesac_softmax_rows(const float* x, float* y, int rows, int D, void* stream);

/* The same softmax written into rows of out_ld >= D elements (columns D .. out_ld-1 = 0), f32 or bf16 (out_dt = NPS_DT_F32 /
 * NPS_DT_BF16): the affinity volume in the channel-padded layout and the type the branch convs read (camera_head.py:1131-1133). */
int nopesac_softmax_rows_pad(const float* x, void* y, int rows, int D, int out_ld, int out_dt, void* stream);

/* Multi-head softmax attention for short sequences (<= 512 keys), head dim 32.
 *   o[b,i,h,:] = softmax_j( scale * q[b,i,h,:].k[b,j,h,:] ) v[b,j,h,:]
 * q/k/v/o are row-major token matrices with arbitrary row strides (elements); batch b's rows start at
 * b*Lq (resp. b*Lk).  qlen/klen: optional int32[B] of valid rows per batch (NULL = all);
 * rows >= qlen[b] are written as zeros.  nn.MultiheadAttention (transformer/transformer.py:166,242-243)
 * and FullAttention (transformer/gnn.py:19-44). f32. */
int nopesac_attention_small(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                            const float* v, int64_t v_stride, float* o, int64_t o_stride,
                            int B, int Lq, int Lk, int heads, float scale,
                            const int32_t* qlen, const int32_t* klen, void* stream);

/* Same contract, mixed-precision MFMA kernel: q/k/v are rounded to bf16 while staged, scores / softmax
 * statistics / output accumulate in f32 (used when MODEL.AMD.COMPUTE_DTYPE = bfloat16). Rows 16-byte aligned. */
int nopesac_attention_small_bf16(const float* q, int64_t q_stride, const float* k, int64_t k_stride,
                                 const float* v, int64_t v_stride, float* o, int64_t o_stride,
                                 int B, int Lq, int Lk, int heads, float scale,
                                 const int32_t* qlen, const int32_t* klen, void* stream);

/* Same kernel with q/k/v/o stored as bf16 (strides in elements, rows 16-byte aligned): the producing / consuming GEMMs round
 * to bf16 anyway, so this only halves the traffic. */
int nopesac_attention_small_bf16io(const void* q, int64_t q_stride, const void* k, int64_t k_stride,
                                   const void* v, int64_t v_stride, void* o, int64_t o_stride,
                                   int B, int Lq, int Lk, int heads, float scale,
                                   const int32_t* qlen, const int32_t* klen, void* stream);

/* gather rows of a [B, H*W, C] map into (w,h) order: y[b, w*H + h, :] = x[b, h*W + w, :]
 * (camera_head.py:1120-1124). f32. */
int nopesac_transpose_hw_rows(const float* x, float* y, int B, int H, int W, int C, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Plane post-selection, fused (meta_arch/siamese_planeTR.py:625-803), per image.
 *   cls_logits f32[B,nq,2]; mask_prob f32[B,h,w,nq] = sigmoid(mask logits) at 1/4 resolution;
 *   params f32[B,nq,3]; query_feat f32[B,nq,D].
 * Outputs (caller-allocated, per image b):
 *   n_kept int32[B]; kept_idx int32[B,nq] (ascending query index, -1 padded);
 *   planes f32[B,nq,3]; feats f32[B,nq,D]; scores f32[B,nq]; areas int32[B,nq]; centers f32[B,nq,2];
 *   winner uint8[B,H,W]: low 7 bits = arg-max query id, bit 7 = weighted prob > mask_thr
 *       (plane masks are decoded from it: mask_q = (id==q) & bit7, or id==q in the fallback case);
 *   flags int32[B]: bit0 = zero_flag (no query passed the score test), bit1 = overlap fallback used.
 *   work int32[B, 9*nq+8]: scratch, zeroed by the call.
 */
int nopesac_postselect_planes(const float* cls_logits, const float* mask_prob, const float* params,
                              const float* query_feat,
                              int B, int nq, int D, int h, int w, int H, int W,
                              float score_thr, float mask_thr, float overlap_thr,
                              int32_t* n_kept, int32_t* kept_idx, float* planes, float* feats, float* scores,
                              int32_t* areas, float* centers, uint8_t* winner, int32_t* flags, int32_t* work,
                              void* stream);
/* Same; prob_planar = 1: mask_prob is laid out [B,nq,h,w] (what nopesac_mask_head_bf16 writes with flag bit 1): only the planes
 * of the queries that pass the score test are read. */
int nopesac_postselect_planes_ex(const float* cls_logits, const float* mask_prob, const float* params,
                                 const float* query_feat,
                                 int B, int nq, int D, int h, int w, int H, int W,
                                 float score_thr, float mask_thr, float overlap_thr,
                                 int32_t* n_kept, int32_t* kept_idx, float* planes, float* feats, float* scores,
                                 int32_t* areas, float* centers, uint8_t* winner, int32_t* flags, int32_t* work,
                                 int prob_planar, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Matching head tail: geometric priors + log-space Sinkhorn with dustbin + mutual-NN assignment,
 * one persistent workgroup per pair (matching_net/matching_head.py:75-96,113-128,228-306;
 * camera_net/camera_modules.py:15-34).
 *   desc_dot f32[B,nq,nq] = D1.D2^T / sqrt(256) (from nopesac_conv2d_nhwc batched);
 *   planes1/2 f32[B,nq,3]; cam7 f32[B,7] = (t, q); n1,n2 int32[B].
 *   log_scores f32[B,nq+1,nq+1]: rows/cols [0,n) are planes, index nq is the dustbin, the rest -inf-like
 *       padding (-1e30); assignment f32[B,nq,nq] in {0,1}.
 */
int nopesac_matcher_sinkhorn(const float* desc_dot, const float* planes1, const float* planes2,
                             const float* cam7, const int32_t* n1, const int32_t* n2,
                             const float* bin_score, float offset_mult, float normal_mult,
                             int iters, float match_thr, int B, int nq,
                             float* log_scores, float* assignment, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Neural one-plane RANSAC, wavefront-level parts (camera_net/camera_head.py:512-569,925-1115).
 *
 * geo_sequence: matched pairs in row-major order of `assignment` -> geo_local f32[B,nq,6],
 *   geo_global f32[B,nq,6] (under init pose), sig f32[B,nq], geo_enc f32[B,nq,8] (the MLP input,
 *   :937-957), m int32[B].  (:1352-1425, :568-569)
 */
int nopesac_geo_sequence(const float* assignment, const float* planes1, const float* planes2,
                         const int32_t* n1, const int32_t* n2, const float* init_trans, const float* init_rot,
                         int B, int nq, int warp_in_ref,
                         float* geo_local, float* geo_global, float* sig, float* geo_enc, int32_t* m,
                         void* stream);

/* hypothesis scoring maps (:990-1006,1018-1035): for pair b, hypothesis h in [0,nq] (0 = initial pose,
 * h>=1 from rot_raw/trans_raw row h-1) and matched pair j in [0,nq):
 *   normal_score[b,h,j] = exp(-|n(R_h p0_j) - n(flip p1_j)|) * mask, param_score[b,h,j] =
 *   exp(-|warp(p0_j; R_h,t_h) - flip p1_j|) * mask, mask = (h<=m)&(j<m).
 * Also writes rots_all f32[B,nq+1,4] (normalised) and trans_all f32[B,nq+1,3], and the diagnostic maps
 * l2_dist / normal_angle / offset_dist f32[B,nq+1,nq] if non-NULL, and the row sums used by
 * 'min-cost' (dn_sum, dl2_sum f32[B,nq+1]). */
int nopesac_ransac_score_maps(const float* geo_local, const float* rot_raw, const float* trans_raw,
                              const float* init_rot, const float* init_trans, const int32_t* m,
                              int B, int nq,
                              float* rots_all, float* trans_all, float* normal_score, float* param_score,
                              float* l2_dist, float* normal_angle, float* offset_dist,
                              float* dn_sum, float* dl2_sum, void* stream);

/* masked softmax over the m+1 hypotheses + soft / average aggregation of the 256-d pose features +
 * final pose regression (:1009-1014,1038-1087) and the m==0 / m<=1 special cases (:964-969,1068-1075).
 *   score_feat_{rot,trans} f32[B,nq+1,64] (outputs of the score MLPs), reg_{rot,trans}_{w f32[64], b f32[1]};
 *   init_{rot,trans}_feat f32[B,256]; fused_{rot,trans}_feat f32[B,nq,256] (already ReLU'd);
 *   rots_{w f32[4,256], b f32[4]}, trans_{w f32[3,256], b f32[3]}.
 *   mode: 0 soft, 1 avg-all, 2 min-cost, 3 max-score; + 16 = the TRAINING-side twin (__forward_PlaneCamRefHead,
 *   camera_head.py:737-923): no m == 0 / m <= 1 shortcuts, scores clamped to [0.01, 0.9], masked and renormalised
 *   (:814-818, :852-854), avg_* from the per-plane features only (:858-867), pred_* = the soft pose (:869-875).
 * Outputs: pred_rot f32[B,4], pred_trans f32[B,3], avg_rot f32[B,4], avg_trans f32[B,3],
 *   score_rot/score_trans f32[B,nq+1]. */
int nopesac_ransac_soft_vote(const float* score_feat_rot, const float* score_feat_trans,
                             const float* reg_rot_w, const float* reg_rot_b,
                             const float* reg_trans_w, const float* reg_trans_b,
                             const float* init_rot_feat, const float* init_trans_feat,
                             const float* fused_rot_feat, const float* fused_trans_feat,
                             const float* rots_w, const float* rots_b, const float* trans_w, const float* trans_b,
                             const float* rots_all, const float* trans_all, const float* dn_sum, const float* dl2_sum,
                             const float* init_rot, const float* init_trans, const int32_t* m,
                             int B, int nq, int mode,
                             float* pred_rot, float* pred_trans, float* avg_rot, float* avg_trans,
                             float* score_rot, float* score_trans, void* stream);

/* The seven refinement losses of the training-side twin (camera_head.py:883-921, CameraPoseLoss camera_modules.py:355-365)
 * from the outputs of nopesac_ransac_score_maps (rots_all, trans_all, l2_dist - the diagnostic map is REQUIRED here) and
 * nopesac_ransac_soft_vote in mode 16 (pred_* = soft pose, avg_*, score_*); gt_pose f32[B,7] = trans | quaternion (normalised
 * inside), m int32[B] (>= 1 each, as in the reference: the parameter loss divides by it).
 * losses f32[7] = { tran_planeAvgReg, rot_planeAvgReg, tran_planeSoftReg, rot_planeSoftReg, rotIdx * 0.01, transIdx * 0.02,
 * paramL2_dist * 0.1 } * weight.  Forward only: no gradient kernels exist for this path. */
int nopesac_plane_cam_ref_losses(const float* pred_rot, const float* pred_trans, const float* avg_rot,
                                 const float* avg_trans, const float* rots_all, const float* trans_all,
                                 const float* score_rot, const float* score_trans, const float* l2_dist,
                                 const int32_t* m, const float* gt_pose, int B, int nq, float weight,
                                 float* losses, void* stream);

/* CameraPoseLoss, reduce=True without mask (camera_modules.py:355-365) - also the form of the AIM's reconstruction losses
 * (camera_head.py:700-705, :725-731, with trans_eps = 1e-10 as the reference adds it to the re-embedded translation):
 *   out[0] = mean_b |gt_trans[b] + trans_eps - est_trans[b]|_2 * weight, out[1] = mean_b |n(gt_rot[b]) - n(est_rot[b])|_2 * weight.
 * est_trans f32[B,3], est_rot f32[B,4]; gt_trans / gt_rot are rows of `stride` floats (7 and 7 for a [B,7] pose tensor with
 * gt_rot = gt_pose + 3).  Forward only. */
int nopesac_camera_pose_loss(const float* est_trans, const float* est_rot, const float* gt_trans, int gt_trans_stride,
                             const float* gt_rot, int gt_rot_stride, int B, float trans_eps, float weight, float* out,
                             void* stream);

/* assignment re-filter under the refined pose (:605-629): keep matches with normal angle < 45 deg and
 * offset distance < 1; `rot` is sign-canonicalised inside (w >= 0). In/out f32[B,nq,nq]. */
int nopesac_refilter_assignment(const float* assignment_in, const float* planes1, const float* planes2,
                                const int32_t* n1, const int32_t* n2, const float* rot, const float* trans,
                                int B, int nq, float* assignment_out, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Launch tape (csrc/tape.hip): the host-side executor of a captured forward.  The reference submits its ~1700 kernels per pair from
 * the Python interpreter (inference_on_dataset, test_NopeSAC.py:157-179: one model(inputs) call per pair); here a static-shape
 * forward is captured once into a hipGraph, `nopesac_tape_create` reads the graph's nodes back (kernel function / grid / block /
 * dynamic LDS / argument block, memset and memcpy parameters) into a flat list in a dependency-respecting order, and
 * `nopesac_tape_replay` issues them as plain launches on `stream` - what the eager path enqueues, without Python (4-5 ms -> ~1 ms
 * per 32-pair forward) and without the whole-graph launch that serialises batches in flight.  `hip_graph` is a hipGraph_t that must
 * outlive the tape (its nodes own the argument blocks).  counts4 (optional) receives {kernels, memsets, memcpys, streams used}.  Unsupported node kinds (host callbacks, child graphs, memory nodes, module launches with packed arguments) -> NPS_E_ARG. */
int nopesac_tape_create(void* hip_graph, void** tape_out, int32_t* counts4);
/* the same with a cap on the number of streams the tape may use (1 = everything on the caller's stream; default 4): parallel branches
 * of the captured graph (the capture's side streams) are issued on the tape's own side streams, joined by events. */
int nopesac_tape_create_ex(void* hip_graph, int max_streams, void** tape_out, int32_t* counts4);
int nopesac_tape_replay(void* tape, void* stream);
/* nopesac_tape_replay with the side chains on the caller's streams: side_streams[k] (hipStream_t) runs chain k + 1, a null / missing
 * entry falls back to the tape's own stream.  Two streams that share a hardware queue execute in submission order, so the placement
 * of the side chains decides how well replays of several tapes overlap (nopesac_amd/streams.py picks streams by queue). */
int nopesac_tape_replay_on(void* tape, void* stream, void* const* side_streams, int n_side);
int nopesac_tape_destroy(void* tape);

/* Diagnostic: `workgroups` x 256 threads fill `lds_bytes` of their LDS with a pattern, spin `spin_cycles`, verify, `rounds` times.
 * *count (uint32, zeroed by the caller) += mismatching dwords; log4 (optional, log_cap x 4 uint32) = {workgroup, dword, expected, found}
 * of the first mismatches.  Run on one stream while other kernels run on others (scripts/lds_victim.py). */
int nopesac_lds_canary(int workgroups, int lds_bytes, int64_t spin_cycles, int rounds, uint32_t* count, uint32_t* log4, int log_cap,
                       void* stream);

/* BENCHMARK-ONLY K control (not part of the reference: it has no such knob - the reference benchmark would need
 * trained weights to keep K planes per view).  Per pair b: the K highest-scoring queries of view 1 (score = logits[b,q,0] -
 * logits[b,q,1], logits f32 [>=B, nq, n_cls]) in ascending query order become rows 0..K-1 of feats[b] (query_feat f32 [>=B,nq,D]);
 * feats[B+b, k] = feats[b, perm[b,k]] + noise[b,k] (perm int64 [B,K], noise f32 [B,K,D]); rows >= K are zeroed; n_kept[0..2B) = K.
 * One launch (the first version was 18 torch launches inside bench.py's timed region). */
int nopesac_force_k_select(const float* logits, int n_cls, const float* query_feat, const int64_t* perm, const float* noise,
                           int B, int nq, int K, int D, float* feats, int32_t* n_kept, void* stream);

/* small pose utilities on [B,*]: L2-normalise rows of a [rows,D] matrix (F.normalize, eps 1e-12),
 * optionally flip the sign so that component 0 >= 0 (camera_head.py:436-437,695-696). */
int nopesac_normalize_rows(const float* x, float* y, int rows, int D, int canonical_sign, void* stream);

/* Adds the number of non-finite (Inf / NaN) values of x[0..n) to *count (int32 on the device, zeroed by the caller).  The reference
 * traps NaN poses with pdb.set_trace() (camera_net/camera_head.py:185-187, 681-682, 1072-1074); the drop-in counts them on the
 * device and raises FloatingPointError when the results are fetched (MODEL.AMD.CHECK_FINITE). */
int nopesac_count_nonfinite(const float* x, int64_t n, int32_t* count, void* stream);
/* uint8 -> float32, exact (8-bit image planes handed over by the data mapper; the reference mapper converts on the host,
 * data/planercnn_transforms.py:225-227, and sends 4 bytes per sample over PCIe).  x, y 16-byte aligned device pointers. */
int nopesac_u8_to_f32(const uint8_t* x, float* y, int64_t n, void* stream);

/* Engine-clock probe: one wave spins for spin_cycles shader cycles; out2[0] = shader cycles, out2[1] = 100 MHz reference ticks of
 * the same interval (device memory, 2 x uint64).  Shader clock [MHz] = 100 * out2[0] / out2[1].  Meant to be launched on a side
 * stream while a workload runs: it reads the clock the firmware grants under that load. */
int nopesac_clock_probe(uint64_t* out2, int64_t spin_cycles, void* stream);

/* the same for up to NOPESAC_NONFINITE_MAX_TENSORS tensors in ONE launch: x / n are HOST arrays of device pointers / element counts */
#define NOPESAC_NONFINITE_MAX_TENSORS 16
int nopesac_count_nonfinite_batch(const float* const* x, const int64_t* n, int n_tensors, int32_t* count, void* stream);

/* ---- Baseline-JPEG decode (the reference's image reader: detectron2 utils.read_image = PIL / libjpeg-turbo,
 * NopeSAC_Net/data/planercnn_transforms.py:210-227 and :306-314 - the ScanNet colour frames).  Bit-exact with libjpeg-turbo's default
 * decompression (JDCT_ISLOW, fancy upsampling, RGB).  The host (nopesac_amd/jpeg.py) walks the markers, splits the entropy-coded data
 * at the restart markers, removes the byte stuffing and lays out the tables; all arrays below are DEVICE memory:
 *   img32 [n_images][NOPESAC_JPEG_IMG_I32] int32: 0 width, 1 height, 2 components (1 | 3), 3 / 4 luma sampling h / v (1 | 2; chroma is
 *     1x1), 5 / 6 MCUs per row / column, 7 MCUs per restart interval (all of them when the file has none), 8..10 blocks per row of
 *     component c, 11..13 block rows, 14..16 downsampled_width, 17..19 downsampled_height, 20..22 index of the component's DC table
 *     (0 | 1), 23..25 index of its AC table (2 | 3), 26 first block of the image in the batch-wide block list, 27 its block count
 *   img64 [n_images][NOPESAC_JPEG_IMG_I64] int64: 0..2 element offset of component c's coefficients in `coef` ([block rows][blocks per
 *     row][64] int16, ZIGZAG order, DC prediction resolved), 3..5 byte offset of its sample plane in `planes` ([block rows * 8][blocks
 *     per row * 8] uint8), 6 byte offset (a multiple of 16) of the image in `out` ([height][width][3] uint8), 7 offset of the image's first interval in
 *     `words`; img32 28 / 29: first lane / subsequence count for nopesac_jpeg_huffman_parallel (29 = 0: not a parallel image)
 *   tables [n_images][NOPESAC_JPEG_TABLES_BYTES]: Huffman tables DC0, DC1, AC0, AC1 (NOPESAC_JPEG_HUFF_BYTES each: 9-bit look-ahead
 *     uint16[512] = (code length << 8) | symbol, 0 = longer code; maxcode int32[18]; valoffset int32[18]; huffval uint8[256] - jdhuff.c's
 *     derived table), then the components' quantisation tables uint16[64] in NATURAL order
 *   seg32 [n_segments][NOPESAC_JPEG_SEG_I32] int32: image, first MCU, MCU count, 0; seg64 [n_segments][NOPESAC_JPEG_SEG_I64] int64:
 *     offset into `words`, word count.  words: the intervals' bytes without the stuffing as 32-bit words, first bit = most significant,
 *     each interval followed by four zero words.
 * nopesac_jpeg_huffman: one wave per segment, n_words = length of `words`, `coef` zero-filled by the caller.  nopesac_jpeg_idct: n_blocks = sum of img32[.][27].
 * nopesac_jpeg_color: max_pixels = the largest width * height of the batch; bgr != 0 writes B, G, R. */
#define NOPESAC_JPEG_HUFF_BYTES 1536
#define NOPESAC_JPEG_TABLES_BYTES (4 * NOPESAC_JPEG_HUFF_BYTES + 3 * 128)
#define NOPESAC_JPEG_IMG_I32 32
#define NOPESAC_JPEG_IMG_I64 8
#define NOPESAC_JPEG_SEG_I32 4
#define NOPESAC_JPEG_SEG_I64 2
int nopesac_jpeg_huffman(const int32_t* img32, const int64_t* img64, const uint8_t* tables, const int32_t* seg32, const int64_t* seg64,
                         int n_segments, const uint32_t* words, int64_t n_words, int16_t* coef, const int32_t* par_done, void* stream);
/* Files WITHOUT restart markers (one serial chain of codes per image): self-synchronising parallel decode.  The stream of image i is
 * cut into img32[i][29] subsequences of NOPESAC_JPEG_SUB_WORDS words, one lane each (lanes img32[i][28] .. of the batch-wide lane list,
 * every image's share padded to a multiple of 64 lanes; lane_img[lane] = image; img32[i][29] = 0: the image is left to
 * nopesac_jpeg_huffman; img64[i][7] = offset of the image's words).  Lanes decode from guessed states, then re-decode from their
 * predecessor's exit state for NOPESAC_JPEG_SYNC_PASSES passes (csrc/jpeg.hip); an image whose lanes no longer change gets
 * par_done[i] = 1 and its coefficients written (DC prediction resolved by a scan); par_done[i] = 0: call nopesac_jpeg_huffman with
 * the same par_done afterwards - it decodes exactly the images still open.  Work arrays (device, n_lanes entries unless noted):
 * exit_state, entry_used (int64, entry_used initialised to -1), n_blk (int32), first_block (int64), changed (int32
 * [NOPESAC_JPEG_SYNC_PASSES][n_images], zeroed), par_done (int32 [n_images], zeroed). */
/* HOST function (no device work, thread-safe): one pass over the entropy-coded bytes that follow a scan header - stuffed zero bytes
 * removed, the stream cut at RSTn markers when has_restart != 0, every interval written to `words` as 32-bit words whose most
 * significant bit is the first bit of the stream, padded to whole words and followed by four zero words (the layout
 * nopesac_jpeg_huffman / _parallel read).  seg_off / seg_cnt / seg_bytes [max_segs]: word offset, word count and byte length of every
 * interval; *consumed = offset of the marker that ended the scan (or n).  Returns the number of intervals, -1 when `words` (capacity
 * words_cap; n / 4 + 5 * max_segs + 8 always suffices) or max_segs is too small.  (RSTn inside a scan without a restart interval
 * ends the scan like any other marker: has_restart = 0.) */
int64_t nopesac_jpeg_prepare_scan(const uint8_t* data, int64_t n, int has_restart, uint32_t* words, int64_t words_cap, int64_t* seg_off,
                                  int64_t* seg_cnt, int64_t* seg_bytes, int64_t max_segs, int64_t* consumed);
/* HOST functions: the whole host side of a BATCH of JPEG files on the library's own threads (csrc/jpeg_host.hip) - file read, marker walk,
 * supported-subset checks, nopesac_jpeg_prepare_scan, derived Huffman tables and the launch arrays above; replaces the per-file Python of
 * nopesac_amd/jpeg.py (parse_markers / prepare_batch), i.e. the reference's per-image utils.read_image call (planercnn_transforms.py:306-314),
 * which held the interpreter lock next to the thread that launches the model.
 *   scan:  status[n] = 0 or why file i keeps the batch off this path (-1 not a JPEG, -2 unsupported, -3 malformed, -6 unreadable);
 *          totals[8] = words, restart intervals, lanes, 8x8 blocks, largest image (pixels), coef / plane / output elements of the batch;
 *          returns an opaque batch (NULL on bad arguments);
 *   fill:  only if every status is 0: img32 [n][32], img64 [n][8], tables [n][TABLES_BYTES], seg32 [segs][4], seg64 [segs][2], words,
 *          lane_img [lanes] (NULL if none), geometry [n][2] = (height, width); 0, -1 bad arguments, -2 bad Huffman table;
 *   free:  always. */
void* nopesac_jpeg_batch_scan_host(const char* const* paths, int n, int threads, int parallel, int* status, int64_t* totals);
int nopesac_jpeg_batch_fill_host(void* batch, int32_t* img32, int64_t* img64, uint8_t* tables, int32_t* seg32, int64_t* seg64, uint32_t* words,
                                 int32_t* lane_img, int32_t* geometry);
void nopesac_jpeg_batch_free_host(void* batch);
#define NOPESAC_JPEG_SUB_WORDS 64
#define NOPESAC_JPEG_SYNC_PASSES 12
int nopesac_jpeg_huffman_parallel(const int32_t* img32, const int64_t* img64, const uint8_t* tables, int n_images, const int32_t* lane_img,
                                  int64_t n_lanes, const uint32_t* words, int64_t n_words, int64_t* exit_state, int64_t* entry_used,
                                  int32_t* n_blk, int64_t* first_block, int32_t* changed, int32_t* par_done, int16_t* coef, void* stream);
int nopesac_jpeg_idct(const int32_t* img32, const int64_t* img64, const uint8_t* tables, int n_images, int n_blocks, const int16_t* coef,
                      uint8_t* planes, void* stream);
int nopesac_jpeg_color(const int32_t* img32, const int64_t* img64, int n_images, int max_pixels, const uint8_t* planes, uint8_t* out,
                       int bgr, void* stream);

/* Result fetch (replaces the per-tensor `.cpu()` copies of siamese_planeTR.py:384-450 at the drop-in boundary): n_segments byte
 * ranges (HOST arrays of device pointers / sizes / destination offsets, n_segments <= NOPESAC_GATHER_MAX_SEGMENTS) are copied into
 * `dst` by ONE kernel.  dst is device memory or device-mapped pinned host memory (hipHostMalloc / torch pin_memory): the latter is the
 * intended use - the batch's small result tensors reach the host in one launch that a captured graph can hold.  Ranges may be
 * unaligned (16-byte vectors are used when a segment's source and destination addresses allow it).  size_dev: NULL, or a HOST array
 * of device pointers (entries may be NULL) to the number of VALID bytes of a segment when only the device knows it (the run-length
 * strings of a batch: a buffer of fixed capacity, filled to a data-dependent length); min(size[i], *size_dev[i]) bytes are copied. */
#define NOPESAC_GATHER_MAX_SEGMENTS 32
int nopesac_gather_bytes(const void* const* src, const int64_t* size, const int64_t* const* size_dev, const int64_t* dst_off,
                         int n_segments, void* dst, void* stream);

/* One GNN layer of the plane matcher (transformer/gnn.py:73-96) for n_sets plane sets, one workgroup per set, bf16 MFMA
 * operands / f32 residual stream (csrc/gnn_layer.hip).  Feature buffers are f32 [sets][nq][256] (nq <= 64); workgroup b updates
 * set x_off + b attending to set src_off + b (same buffer and offset = 'self' layer) and writes set out_off + b of `out`
 * (out may share a buffer with x / src only for disjoint set ranges).  qlen / klen: int32 valid rows per set, indexed like
 * x / src (NULL = nq).  All six weight matrices are bf16 in MFMA fragment-major order (see nopesac_bottleneck_tail_bf16):
 * wq (pre-multiplied by 1/sqrt(32)), wk, wv, wmerge [256][256]; w0 = mlp.0 [512][512] (input = [x | msg]); w2 = mlp.2 [256][512]. */
int nopesac_gnn_layer_bf16(const float* x, int x_off, const float* src, int src_off, float* out, int out_off, int n_sets, int nq,
                           const int32_t* qlen, const int32_t* klen, const void* wq, const void* wk, const void* wv,
                           const void* wmerge, const void* w0, const void* w2, const float* ln1_g, const float* ln1_b,
                           const float* ln2_g, const float* ln2_b, void* stream);
/* The same launch + a weight prefetch for the NEXT one (one pair per call: one or two workgroups per launch, each layer's 1.28 MB of
 * weights cold in L2): next_weights6 = HOST array of the next launch's six fragment-major weight pointers (wq, wk, wv, wmerge, w0,
 * w2), next_sets = its n_sets; 128 extra workgroups of this launch, dealt over the XCDs like the next launch's workgroups will be,
 * read those weights into the L2s that will need them and exit.  No prefetch workgroups are added when this launch has more than 16
 * layer workgroups.  next_weights6 = NULL: exactly nopesac_gnn_layer_bf16. */
int nopesac_gnn_layer_bf16_pf(const float* x, int x_off, const float* src, int src_off, float* out, int out_off, int n_sets,
                              int nq, const int32_t* qlen, const int32_t* klen, const void* wq, const void* wk, const void* wv,
                              const void* wmerge, const void* w0, const void* w2, const float* ln1_g, const float* ln1_b,
                              const float* ln2_g, const float* ln2_b, const void* const* next_weights6, int next_sets, void* stream);

/* Finest top-down level + mask head of the PlaneTR head in one launch (planeTR_head.py:148-162, 241-252), bf16:
 *   p1 = relu(scale * (w_lateral . c1) + bias) + relu(bilinear_2x(t1));   prob = [sigmoid](mask_w[b] . p1 + mask_b[b])
 * c1 [B,H,W,256], t1 [B,H/2,W/2,256] bf16; w_lateral [256][256] and mask_w [B][NQP][256] (rows >= nq zero) bf16 in MFMA
 * fragment-major order; mask_b f32 [B][NQP]; NQP = 64 for nq <= 64, 128 for nq <= 128; prob f32 [B,H,W,nq] (nq even, <= 128);
 * p1_out optional bf16 [B,H,W,256].
 * H*W must be a multiple of 128.  apply_sigmoid: bit 0 = apply the sigmoid, bit 1 = write prob planar, [B,nq,H,W], bit 2 = tuning
 * aid / test: one pixel per item in the bilinear phase (default: four consecutive pixels share their eight taps; identical results). */
int nopesac_mask_head_bf16(const void* c1, const void* t1, const void* w_lateral, const float* scale, const float* bias,
                           const void* mask_w, const float* mask_b, float* prob, void* p1_out, int B, int H, int W, int nq,
                           int apply_sigmoid, void* stream);

/* The two per-image operands of nopesac_mask_head_bf16 from the folded plane embeddings (fold f32 [B * nq][ld]: columns 0..255 mask
 * weights, column 256 mask bias - the pixel-embedding conv folded into the plane-embedding MLP, planeTR_head.py:170-188): mask_w bf16
 * in the kernel's per-image MFMA fragment order for nqp = 64 / 128 planes (planes >= nq zero), mask_b f32 [B][nqp].  One launch. */
int nopesac_mask_operands(const float* fold, int ld, void* mask_w, float* mask_b, int B, int nq, int nqp, void* stream);

/* Tail of one post-norm transformer encoder layer (transformer/transformer.py:183-199) for M tokens of width 256, FFN 1024:
 *   y1 = LN1(src + attn . wo^T + bo);  y2 = LN2(y1 + relu(y1 . w1^T + b1) . w2^T + b2)
 * attn bf16 [M][256] (attention output), src f32 [M][256]; wo [256][256], w1 [1024][256], w2 [256][1024] bf16 in MFMA
 * fragment-major order; outputs (each nullable): y f32, y_bf16, ypos_bf16 = bf16(y2 + pos[token % pos_rows]). */
int nopesac_encoder_tail_bf16(const void* attn, const float* src, const void* wo, const float* bo, const float* ln1_g,
                              const float* ln1_b, const void* w1, const float* b1, const void* w2, const float* b2,
                              const float* ln2_g, const float* ln2_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                              void* ypos_bf16, int M, void* stream);

/* cv2.resize(img, (OW, OH)) with the default INTER_LINEAR on uint8 HWC images (the ScanNet input path,
 * data/planercnn_transforms.py:314): OpenCV's 11-bit fixed-point algorithm (csrc/resize.hip). src [H][W][C], dst [OH][OW][C]. */
int nopesac_resize_bilinear_u8(const uint8_t* src, int H, int W, int C, uint8_t* dst, int OH, int OW, void* stream);
/* The same for n images of one size that lie src_stride bytes apart (the GPU JPEG decoder's output buffer), one launch; chw = 1: dst is
 * [n][C][OH][OW] (the mapper's CHW tensors: the reference's transpose after cv2.resize, planercnn_transforms.py:314-320), else [n][OH][OW][C]. */
int nopesac_resize_bilinear_u8_batch(const uint8_t* src, int n, int64_t src_stride, int H, int W, int C, uint8_t* dst, int OH, int OW, int chw,
                                     void* stream);

/* Pre-norm counterpart for the decoder layers (transformer/transformer.py:293-322, after the cross-attention), same kernel:
 *   s = tgt + attn . wo^T + bo;  u = s + relu(LN3(s) . w1^T + b1) . w2^T + b2;  n = LN_next(u)
 * y = u (f32 residual stream), y_bf16 = bf16(n), ypos_bf16 = bf16(n + pos[row % pos_rows]), yn = n in f32 (each nullable);
 * LN_next = the next layer's norm1, or the decoder's final norm after the last layer. */
int nopesac_decoder_tail_bf16(const void* attn, const float* tgt, const void* wo, const float* bo, const float* ln3_g,
                              const float* ln3_b, const void* w1, const float* b1, const void* w2, const float* b2,
                              const float* lnn_g, const float* lnn_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                              void* ypos_bf16, float* yn, int M, void* stream);

/* General form of the two tails above, followed by the input projections of the NEXT attention - one launch:
 *   pre_norm = 0 (encoder layer, transformer.py:183-199):  n = LN_b(y1 + FFN(y1)), y1 = LN_a(src + out_proj(attn));  y = n
 *   pre_norm = 1 (decoder layer after the cross-attention, :308-322):  u = s + FFN(LN_a(s)), s = src + out_proj(attn); n = LN_b(u); y = u
 *   skip_ffn = 1 (decoder layer after the SELF-attention, :300-306):  s = src + out_proj(attn), n = LN_a(s), y = s (w1 / w2 / ln_b unused)
 *   proj_pos [M][n_pos] = bf16((n + pos) Wpos^T + bpos),  proj [M][n_proj] = bf16(n Wp^T + bp): the next self-attention's q|k and v, or
 *   the cross-attention's q (nn.MultiheadAttention in_proj slices); weights in mfma_fragment_major order (K = 256), widths multiples of 32.
 * Outputs y / y_bf16 / ypos_bf16 / yn as in the two entries above; every output is optional (at least one). */
int nopesac_transformer_tail_bf16(const void* attn, const float* src, const void* wo, const float* bo, const float* lna_g,
                                  const float* lna_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                  const float* lnb_g, const float* lnb_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                  void* ypos_bf16, float* yn, int pre_norm, int skip_ffn, const void* w_pos, const float* b_pos,
                                  void* proj_pos, int n_pos, const void* w_proj, const float* b_proj, void* proj, int n_proj, int M,
                                  void* stream);
/* The same launch + a weight prefetch for the NEXT one (few workgroups - one pair per call): next_ptrs / next_bytes = HOST arrays of up to
 * 8 device byte ranges (the next tail's wo / w1 / w2 and projection matrices), next_workgroups = the next launch's workgroup count; with
 * at most 64 layer workgroups in THIS launch, 128 extra workgroups read those ranges into the L2s of the XCDs the next launch will
 * run on and exit (csrc/enc_tail.hip, like nopesac_gnn_layer_bf16_pf).  n_next = 0: exactly nopesac_transformer_tail_bf16. */
int nopesac_transformer_tail_bf16_pf(const void* attn, const float* src, const void* wo, const float* bo, const float* lna_g,
                                     const float* lna_b, const void* w1, const float* b1, const void* w2, const float* b2,
                                     const float* lnb_g, const float* lnb_b, const float* pos, int pos_rows, float* y, void* y_bf16,
                                     void* ypos_bf16, float* yn, int pre_norm, int skip_ffn, const void* w_pos, const float* b_pos,
                                     void* proj_pos, int n_pos, const void* w_proj, const float* b_proj, void* proj, int n_proj, int M,
                                     const void* const* next_ptrs, const int64_t* next_bytes, int n_next, int next_workgroups, void* stream);

/* ---- COCO RLE of the kept plane masks (replaces pycocotools.mask.encode / toBbox at
 *      meta_arch/siamese_planeTR.py:703-704, 747-748; consumed by evaluation/mp3d_evaluation.py:203-205) ----
 * labels: winner uint8[V,H,W] (+ kept_idx int32[V,nq], n_kept int32[V], flags int32[V] from nopesac_postselect_planes)
 *   -> uint8[V,W,H] COLUMN-major map of kept-plane ordinals (0xFF = none; bit-7 test skipped when flags bit1 = fallback). */
int nopesac_rle_labels(const uint8_t* winner, const int32_t* kept_idx, const int32_t* n_kept, const int32_t* flags,
                       uint8_t* labels, int V, int H, int W, int nq, void* stream);
/* transitions: counts int32[V,nq] = number of 0<->1 flips of plane p's mask along the column-major scan (N = H*W);
 *   if positions != NULL the ascending flip positions of (v,p) are written at positions[offsets[v*nq+p] ...]
 *   (call once with NULLs to size, exclusive-scan the counts, call again). */
int nopesac_rle_transitions(const uint8_t* labels, const int32_t* n_kept, const int64_t* offsets, int32_t* counts,
                            uint32_t* positions, int V, int N, int nq, void* stream);
/* Dense boolean masks of the kept planes of V views in one launch (`pred_plane_masks`, siamese_planeTR.py:685 / :743): for view v and
 * its p-th kept plane, masks[(offsets[v] + p) * H * W + pixel] = 1 iff the pixel's arg-max query is kept_idx[v][p] and the pixel
 * passed the mask threshold (bit 7 of the winner byte) or the view is a fallback view (flags bit 1).  offsets: device int64 [V],
 * exclusive prefix sums of n_kept.  masks: uint8 [sum(n_kept), H, W]. */
int nopesac_decode_masks(const uint8_t* winner, const int32_t* kept_idx, const int32_t* n_kept, const int32_t* flags,
                         const int64_t* offsets, uint8_t* masks, int V, int H, int W, int nq, void* stream);

/* HOST function (no device work): one mask's flip positions -> COCO compressed "counts" string (not NUL terminated,
 *   returns its length or < 0) and bbox4 = [x, y, w, h] (cocoapi rleToString / rleToBbox). */
int nopesac_rle_compress_host(const uint32_t* positions, int n_pos, int H, int W, char* out, int cap, double* bbox4);
/* The same on the device for n_masks masks at once (mask i owns positions[offsets[i] .. offsets[i] + counts[i]); all pointers device
 * memory).  Pass 1, out == NULL: lens[i] = length of mask i's string, bbox4[4 i ..] = its [x, y, w, h] box.  Pass 2, out != NULL: the
 * strings are written at out + out_off[i] (the caller's exclusive prefix sums of lens). */
int nopesac_rle_compress_device(const uint32_t* positions, const int64_t* offsets, const int32_t* counts, int n_masks, int H, int W,
                                int32_t* lens, double* bbox4, char* out, const int64_t* out_off, void* stream);
/* Pass 2 of nopesac_rle_compress_device into a buffer of fixed capacity `cap` bytes, so that the caller needs no host round trip for
 * the total string length: string i is written at out + out_off[i] only when out_off[i] + lens[i] <= cap (lens: pass 1's output).
 * The strings' consumer (siamese_planeTR.py:703-720, instances[k]["segmentation"]) gets the same bytes as from the unbounded form. */
int nopesac_rle_compress_device_capped(const uint32_t* positions, const int64_t* offsets, const int32_t* counts, int n_masks, int H, int W,
                                       const int32_t* lens, char* out, const int64_t* out_off, int64_t cap, void* stream);
/* Batch form of nopesac_rle_compress_host: mask i owns positions[offsets[i] .. +counts[i]); strings are packed back to back into
 * `out` (mask i at out + out_off[i], out_off[n_masks] = total), boxes at bbox4 + 4 i.  Returns the total length or < 0. */
long long nopesac_rle_compress_batch_host(const uint32_t* positions, const long long* offsets, const int* counts, int n_masks,
                                          int H, int W, char* out, long long cap, long long* out_off, double* bbox4);


/* Layers 1..5 of both branches of the pixel pose net (camera_net/camera_modules.py `convs_trans` / `convs_rots`: Conv3x3 + BatchNorm +
 * LeakyReLU(0.01), strides 2,1,2,1,2 behind the stride-1 first layer; call site camera_head.py:642-735) in one launch, one workgroup per
 * (image pair, branch), activations resident in LDS.  x_trans / x_rots: the branches' layer-0 outputs [B][15][20][128] bf16;
 * w10 / scale10 / bias10: 10 pointers each, index = branch * 5 + (layer - 1), branch 0 = trans: bf16 weights [128][3*3*128] in
 * nopesac MFMA fragment-major order (K index = (kh*3 + kw)*128 + c), folded-BatchNorm f32 scale / shift [128];
 * y_trans / y_rots: [B][2][3][128] f32 (NHWC: the FC stack's input).  H, W, C must be 15, 20, 128 (the 480 x 640 geometry). */
int nopesac_posenet_branch_tail_bf16(const void* x_trans, const void* x_rots, const void* const* w10, const float* const* scale10,
                                     const float* const* bias10, float* y_trans, float* y_rots, int B, int H, int W, int C, void* stream);

/* ---- backward kernels of the camera head's training-side twin (csrc/refine_bwd.hip) ----------------------------
 * Vector-Jacobian products of nopesac_plane_cam_ref_losses, nopesac_ransac_soft_vote (mode | 16) and nopesac_ransac_score_maps - the
 * forward of the reference's __forward_PlaneCamRefHead (camera_net/camera_head.py:737-923; CameraPoseLoss camera_modules.py:355-365).
 * All f32, [B, ...] layouts exactly as the forward kernels'; no atomics: parameter gradients of the vote kernel are written PER PAIR
 * (pb_* = [B, size of the parameter]) and summed over the pairs by nopesac_col_sum_f32.  The Linear / MLP stacks in between are
 * differentiated with nopesac_conv2d_nhwc (f32): dX = dY W, dW = dY^T X, plus the helpers below (nopesac_amd/training.py).
 *  losses_backward:     g_loss [7] (d total / d each loss of nopesac_plane_cam_ref_losses) -> gradients of the four poses ([B,4] / [B,3],
 *                       with respect to the NORMALISED quaternions the vote kernel returns), of the scores [B,nq+1] (the hypothesis the
 *                       index losses pick is a constant, as in autograd) and of l2_dist [B,nq+1,nq] (its diagonal entries).
 *  vote_backward:       -> gradients of the score features [B,nq+1,64] x 2, the initial pose features [B,256] x 2, the per-plane features
 *                       [B,nq,256] x 2 and, per pair, of rots / trans weights + biases and of the two score regressors.
 *  score_maps_backward: gradients of normal_score / param_score / l2_dist [B,nq+1,nq] -> rot_raw [B,nq,4] (through its normalisation),
 *                       trans_raw [B,nq,3], init_rot [B,4], init_trans [B,3]. */
int nopesac_refine_losses_backward(const float* pred_rot, const float* pred_trans, const float* avg_rot, const float* avg_trans,
                                   const float* rots_all, const float* trans_all, const float* score_rot, const float* score_trans,
                                   const int32_t* m, const float* gt_pose, const float* g_loss, int B, int nq, float weight,
                                   float* g_pred_rot, float* g_pred_trans, float* g_avg_rot, float* g_avg_trans, float* g_score_rot,
                                   float* g_score_trans, float* g_l2_dist, void* stream);
int nopesac_refine_vote_backward(const float* sf_rot, const float* sf_trans, const float* reg_rot_w, const float* reg_rot_b,
                                 const float* reg_trans_w, const float* reg_trans_b, const float* init_rot_feat,
                                 const float* init_trans_feat, const float* fused_rot, const float* fused_trans, const float* rots_w,
                                 const float* rots_b, const float* trans_w, const float* trans_b, const int32_t* m, int B, int nq,
                                 const float* g_pred_rot, const float* g_pred_trans, const float* g_avg_rot, const float* g_avg_trans,
                                 const float* g_score_rot, const float* g_score_trans, float* g_sf_rot, float* g_sf_trans,
                                 float* g_init_rot_feat, float* g_init_trans_feat, float* g_fused_rot, float* g_fused_trans,
                                 float* pb_rots_w, float* pb_rots_b, float* pb_trans_w, float* pb_trans_b, float* pb_reg_rot_w,
                                 float* pb_reg_rot_b, float* pb_reg_trans_w, float* pb_reg_trans_b, void* stream);
int nopesac_refine_score_maps_backward(const float* geo_local, const float* rot_raw, const float* trans_raw, const float* init_rot,
                                       const float* init_trans, const int32_t* m, int B, int nq, const float* g_normal_score,
                                       const float* g_param_score, const float* g_l2_dist, float* g_rot_raw, float* g_trans_raw,
                                       float* g_init_rot, float* g_init_trans, void* stream);
/* backward of nopesac_camera_pose_loss (CameraPoseLoss camera_modules.py:355-365 and the AIM's reconstruction losses camera_head.py:700-705,
 * :725-731): g_out [2] -> gradients of BOTH pose arguments ([B,3] / [B,4] dense; the "ground truth" of a reconstruction loss is the pixel
 * pose, an output of trainable layers). */
int nopesac_camera_pose_loss_backward(const float* est_trans, const float* est_rot, const float* gt_trans, int gt_trans_stride,
                                      const float* gt_rot, int gt_rot_stride, int B, float trans_eps, float weight, const float* g_out,
                                      float* g_est_trans, float* g_est_rot, float* g_gt_trans, float* g_gt_rot, void* stream);
/* helpers of the Linear backward and the optimiser (f32): y [cols,rows] = x^T (x rows strided by x_ld); out [cols] = column sums (fixed
 * summation order); out = y > 0 ? g : 0; J^T g of row-wise x / max(|x|, 1e-12) (D <= 4; canonical_sign: the forward also flipped rows with x[0] < 0); torch.optim.AdamW / SGD(momentum) updates of one
 * tensor (train_NopeSAC.py:150-157). */
int nopesac_transpose_f32(const float* x, int rows, int cols, int64_t x_ld, float* y, void* stream);
int nopesac_col_sum_f32(const float* x, int rows, int cols, int64_t x_ld, float* out, void* stream);
int nopesac_relu_backward_f32(const float* g, const float* y, int64_t n, float* out, void* stream);
int nopesac_normalize_rows_backward(const float* x, const float* g, int rows, int D, int canonical_sign, float* out, void* stream);
/* full-model gradient clipping (train_NopeSAC.py:139-148 -> torch.nn.utils.clip_grad_norm_): out[0] += sum x^2 (one launch per tensor on the
 * same accumulator, fixed order); coef[0] = min(1, max_norm / (sqrt(sumsq[0]) + 1e-6)); x *= coef[0]. */
int nopesac_sumsq_accumulate_f32(const float* x, int64_t n, float* out, void* stream);
int nopesac_clip_coefficient(const float* sumsq, float max_norm, float* coef, void* stream);
int nopesac_scale_by_f32(float* x, int64_t n, const float* coef, void* stream);
int nopesac_adamw_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1, float beta2,
                       float eps, float weight_decay, int step, void* stream);
int nopesac_sgd_step(float* param, const float* grad, float* momentum_buf, int64_t n, float lr, float momentum, float weight_decay,
                     int first_step, void* stream);

/* ---- host-side PNG decode for the data mapper (csrc/png_host.hip; no kernel) ---------------------------------------------------------
 * The mp3d split stores 480 x 640 PNG frames (reference: data/planercnn_transforms.py:210-227 -> detectron2 utils.read_image -> PIL).
 * PIL decodes PNGs with the interpreter lock held; these entry points are called through ctypes with the lock released, so the reader
 * threads scale with the host's cores.  nopesac_png_info_host: 0 + geometry (*supported = 1 if the decoder below takes the file), negative
 * if the bytes are not a PNG.  nopesac_png_decode_host: the file's samples as interleaved RGB (bgr = 0) / BGR (bgr = 1) in out
 * [H * W * 3], converted like PIL's convert("RGB") (grey replicated, palette looked up, alpha dropped); 0, or -1 not a PNG, -2
 * unsupported (16-bit / sub-byte / interlaced: the caller falls back to PIL), -3 truncated / corrupt (CRC, inflate, filter type), -4 out
 * too small.  (zlib is optional: only the NOPESAC_PNG_ZLIB_INFLATE=1 A/B path uses it; the default inflate is csrc/inflate_host.h.) */
int nopesac_png_info_host(const unsigned char* data, int64_t n, int* height, int* width, int* channels, int* supported);
int nopesac_png_decode_host(const unsigned char* data, int64_t n, unsigned char* out, int64_t out_bytes, int bgr);
/* A batch of PNG FILES on `threads` threads of the call itself (replaces the per-image Python of the reference mapper,
 * planercnn_transforms.py:210-227, whose open / read / array / transpose steps hold the interpreter lock): file paths[i] -> out + i *
 * image_stride, H x W x 3 interleaved or (flags bit 1) 3 x H x W channel-major - the mapper's CHW layout - in RGB or (flags bit 0) BGR
 * order.  status[i] = 0, a code of nopesac_png_decode_host, -5 geometry is not H x W, -6 unreadable file.  Returns the number of files
 * with a non-zero status (the caller hands those to PIL), negative on bad arguments. */
int nopesac_png_decode_files_host(const char* const* paths, int n, unsigned char* out, int64_t image_stride, int H, int W, int flags,
                                  int threads, int* status);
/* The decoder's own inflate (csrc/inflate_host.h: a whole RFC 1950 / 1951 stream in memory -> a buffer of known size; replaces zlib's
 * streaming inflate, the floor of the PNG path) on its own, for tests: bytes written, or -1 on any malformed stream / overflow. */
int64_t nopesac_inflate_zlib_host(const unsigned char* in, int64_t n, unsigned char* out, int64_t out_cap);

#ifdef __cplusplus
}
#endif
#endif /* NOPESAC_HIP_H */
