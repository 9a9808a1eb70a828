/*
 * xview_hip.h -- C ABI of libxview_hip.so: the MI355X (gfx950 / CDNA4) kernels behind the
 * two-stream FCN + probabilistic-fusion hot path of ethz-asl/modular_semantic_segmentation.
 *
 * The reference has NO native boundary for this path: every op below is a TensorFlow-1 graph
 * op reached through `sess.run` (xview/models/base_model.py:257-258,288-289,310-311).  Each
 * entry point therefore cites the reference graph call site it replaces; INTEGRATION.md shows
 * the ctypes binding a maintainer of the reference would add.
 *
 * Conventions
 *   - every function returns 0 on success, a negative XV_E* code for a bad argument, or a
 *     positive hipError_t from the launch; nothing throws, allocates or synchronises
 *   - all pointers are DEVICE pointers owned by the caller; `stream` is a hipStream_t
 *   - activations are bf16 "padded NHWC": a dense [N][H+2][W+2][C] buffer whose 1-pixel border is
 *     zero and is never written by any kernel (3x3 'same' convolutions read it as their zero
 *     padding; custom_layers.py:124-139 `padding='same'`).  `xv_act.data` points at the first
 *     element of the padded buffer.  Network inputs/outputs (images, probabilities, label maps)
 *     are dense unpadded NHWC / NHW tensors in the reference's dtypes (float32 / int64 / int32).
 */
#ifndef XVIEW_HIP_H
#define XVIEW_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define XV_OK 0
#define XV_EINVAL (-1)      /* bad argument (null pointer, non-positive size, ...)          */
#define XV_ESHAPE (-2)      /* shape not supported by the kernel (see each function)        */
#define XV_EWORKSPACE (-3)  /* workspace too small                                          */

/* padded-NHWC activation: dense [n][h+2][w+2][c], zero border, data -> padded origin.
 * dtype XV_BF16 (every entry point) or XV_FP8 (OCP e4m3fn, one byte per element; xv_conv2d_fwd* only, BASELINE
 * config "fp8 MFMA conv path"): a stored value q stands for q * 2^scale_exp (per-tensor power-of-two scale, 0 for
 * bf16).  Entry points other than the forward convolutions take bf16 maps only and return XV_EINVAL for an XV_FP8
 * descriptor (they would otherwise reinterpret one-byte elements as bf16).                                          */
#define XV_BF16 0
#define XV_FP8 1
typedef struct xv_act {
  void* data;
  int32_t n, h, w, c;
  int32_t dtype;
  int32_t scale_exp;
} xv_act;

/* Library version (major*10000 + minor*100 + patch). */
int xv_version(void);
/* Name of the gfx target the device code was compiled for ("gfx950"). */
const char* xv_arch(void);
/* First 16 hex digits of the sha256 over the sources this binary was compiled from (csrc/Makefile SRC_HASH);
 * the Python loader compares it with the sources next to it and refuses a stale library. */
const char* xv_source_hash(void);

/* ---- weight packing ------------------------------------------------------------------------
 * Conv kernels arrive in the reference npz schema: float32 HWIO [k][k][cin][cout]
 * (base_model.py:361-393 export_weights; tf.layers.conv2d kernel layout).  The MFMA kernels read
 * them as bf16 [tap][cin/64][cout][64 swizzled], 128 B per (tap, cin-chunk, cout) row with
 * 16-byte slot s of row `co` stored at slot s ^ (co & 6) (so a weight tile is a linear copy
 * into bank-conflict-free LDS).  For k = 3 a second image follows in the same buffer for the
 * 32-channel-chunk kernel: bf16 [tap][cin/32][cout][32], 64-byte rows, slot s of row `co` at
 * s ^ ((co >> 1) & 2); and a third for generations 4 / 5 and the fused first pair (16x16x32 MFMA blocks,
 * configurations 26-28): the same [tap][cin/32][row][32] with slot s of row r at s ^ ((r >> 1) & 2) and the rows of
 * every 64-row block permuted (row 16j + 4g + q = channel 16g + 4j + q, so that a lane's 16 accumulator registers of a
 * pixel are 16 consecutive output channels).  xv_packed_weight_bytes is the size of all three.
 * k = 1 or 3, cin % 64 == 0, cout % 64 == 0.                                                       */
size_t xv_packed_weight_bytes(int k, int cin, int cout);
int xv_pack_conv_weights(const float* w_hwio, void* packed, int k, int cin, int cout, void* stream);
/* Forward image and data-gradient image (xv_pack_conv_weights_dgrad) of one layer in a single launch: what a training
 * step does for every conv after the optimizer update.                                                               */
int xv_pack_conv_weights_pair(const float* w_hwio, void* packed, void* packed_dgrad, int k, int cin, int cout,
                              void* stream);

/* fp8 weights (config "fp8 MFMA conv path", vgg16.py:7-51 / custom_layers.py:124-139 call sites): OCP e4m3fn,
 * [tap][cin/128][cout][128] (128-byte rows, slots swizzled like the bf16 image: a weight tile has the same bytes
 * geometry), preceded by a 256-byte header whose first int32 is the per-tensor scale exponent e_w (stored value q
 * stands for q * 2^e_w; the kernel feeds it to the MFMA as a uniform E8M0 block scale).  Values are rounded to
 * nearest-even after the scaling and saturate at +-448.  For k = 3 a second image follows for the generation-4 kernel:
 * [tap][cin/64][row][64], 64-byte rows, slot s of row r at s ^ ((r >> 2) & 3), rows permuted as in the bf16 image;
 * with it cin % 64 == 0 suffices for k = 3 (the first image is left empty unless cin % 128 == 0).  k = 1: cin % 128 == 0.  cout % 64 == 0.                    */
size_t xv_packed_weight_bytes_f8(int k, int cin, int cout);
int xv_pack_conv_weights_f8(const float* w_hwio, void* packed, int k, int cin, int cout, int scale_exp, void* stream);

/* ---- conv2d forward ------------------------------------------------------------------------
 * y = act(conv(x, W) + b): tf.layers.conv2d(padding='same', strides 1) via
 * custom_layers.py:124-139 (call sites simple_fcn.py:39-79).  Implicit GEMM on bf16 MFMA
 * (v_mfma_f32_16x16x32_bf16, fp32 accumulate), input halo tile + weight tile staged in LDS.
 * k = 3 (conv1_2 .. conv5_3) or 1 (score_conv4 / score_conv5, AdapNet's block stages: a flat GEMM over the padded rows
 * from 256 input channels, conv1x1_gemm.hip); cin % 64 == 0, cout % 64 == 0.
 * Inference batch-norm is folded into (W, b) by the host before packing.
 * If pooled != NULL (k = 3 only, h and w even) the 2x2/2 max-pool of y
 * (max_pooling2d, simple_fcn.py:41,44,48,58) is written to `pooled` from the same accumulators;
 * y itself may then be NULL (data == NULL) to skip the full-resolution store.
 * fp8 (the dtype fields select it, no second entry point): x->dtype == XV_FP8 runs the block-scaled
 * v_mfma_scale_f32_16x16x128_f8f6f4 kernel (e4m3 operands, 128-channel chunks, fp32 accumulate; x->scale_exp and the
 * weight header's exponent go in as uniform E8M0 scales, so the accumulators are in real units) and needs weights
 * packed by xv_pack_conv_weights_f8 and cin % 128 == 0; y->dtype / pooled->dtype == XV_FP8 make the epilogue write
 * e4m3 (value * 2^-scale_exp, round-to-nearest-even, saturating at 448) from either kernel.  3x3 convs run on the
 * generation-4 kernel where it is ahead (configurations 24 / 25, conv_f8_dma.hip: v_mfma_scale_f32_32x32x64_f8f6f4 on
 * 64-channel e4m3 chunks, so cin % 64 == 0 suffices for k = 3; v_mfma_f32_32x32x16_bf16 for bf16 maps -- bf16 in / e4m3
 * out included; any map size, partial tiles by clamped DMA offsets and predicated stores): conv2_1 (64 input channels)
 * is an e4m3 convolution there.                                                                                  */
int xv_conv2d_fwd(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                  const xv_act* pooled, int k, int relu, void* stream);

/* The 3x3 conv of a conv -> batch norm block with the batch statistics taken in the conv's own epilogue (generation 4: bf16
 * maps that tile exactly in 16x32 pixels, cout <= 512): y = conv(x, W) + b as xv_conv2d_fwd(relu = 0), and row w of
 * stats_rows ([xv_conv2d_stats_rows()][2 cout] floats) = workgroup w's per-channel sum | sum of squares of the STORED
 * (bf16) outputs; xv_bn_sums_from_rows(stats_rows, xv_conv2d_stats_rows(), 2 cout, sums) adds the rows in a fixed tree and
 * leaves what xv_bn_stats would have.  Returns XV_ESHAPE where that kernel does not apply (use xv_conv2d_fwd + xv_bn_stats). */
int xv_conv2d_stats_rows(void);
int xv_conv2d_fwd_stats(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, float* stats_rows,
                        size_t stats_bytes, void* stream);
int xv_bn_sums_from_rows(const float* rows, int nrows, int len, double* sums, void* stream);

/* The routed pool of training (generation 4: bf16 maps that tile exactly in 16x32 pixels; XV_ESHAPE elsewhere -- keep the full
 * map and xv_maxpool2x2_bwd there).  Replaces, bit for bit, tf.layers.conv2d(relu) + max_pooling2d forward and
 * Conv2DBackpropInput + MaxPoolGrad + ReluGrad backward (simple_fcn.py:41,44,48 under base_model.py:153-162) without the
 * full-resolution map and without the pooled gradient in memory:
 *   xv_conv2d_fwd_route:      pooled = maxpool2x2(relu(conv3x3(x) + b)); route[n][h/2][w/2][cout] = one byte per pooled value:
 *                             0 = the window's maximum is not positive, else 0x80 >> the first position of the maximum
 *                             (row 0: columns 0, 1; row 1: columns 0, 1 -- MaxPoolGrad's order).
 *   xv_conv2d_bwd_data_route: dx (twice dy's size) = conv3x3(dy, Wd) stored through `route`: every value at the position its
 *                             byte names, zeros at the other three (every interior value of dx is written).
 * xv_conv2d_route_bytes(n, h, w, cout): size of the route buffer of a conv over [n][h][w] maps (0: bad dimensions).      */
size_t xv_conv2d_route_bytes(int n, int h, int w, int cout);
int xv_conv2d_fwd_route(const xv_act* x, const void* w_packed, const float* bias, const xv_act* pooled, void* route,
                        size_t route_bytes, void* stream);
int xv_conv2d_bwd_data_route(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias, const void* route,
                             size_t route_bytes, const xv_act* dx, void* stream);

/* Residual form for the 1x1 convs that close a ResNet block: y = act(conv1x1(x, W) + b) + residual
 * (block_a / block_b of adapnet.py:38-51,80-100: stage_3 carries its own relu, the block's outer relu is the identity
 * on the sum of two non-negative maps).                                                                               */
int xv_conv2d_fwd_residual(const xv_act* x, const void* w_packed, const float* bias, const xv_act* residual,
                           const xv_act* y, int relu, void* stream);

/* Dense transposed convolution: tf.layers.conv2d_transpose(filters, k, strides=s, padding='same', use_bias=False) with
 * an ARBITRARY kernel (custom_layers.py:71-121 `deconv2d`; the FCN's upscore layers are constants and take the
 * depthwise bilinear kernels below, a trained or imported non-bilinear kernel -- AdapNet trains its deconvs,
 * adapnet.py:156-163 -- takes this path), followed by the optional inference batch norm (scale / shift per channel),
 * the activation and a residual add:  y = act(deconv(x, W) * scale + shift) + residual.
 * k = 2s only ([4,4]/2 and [16,16]/8, every call site of the reference).  w_phases_packed: the kernel re-arranged by
 * the host into a 3x3 conv cin -> s*s*cout (one 2x2-tap sub-conv per output phase; custom_layers.
 * dense_deconv_as_conv3x3) and packed by xv_pack_conv_weights; zero_bias: s*s*cout zeros; workspace: the phase map.  */
size_t xv_deconv_dense_workspace_bytes(int n, int h, int w, int cout, int stride);
int xv_deconv_dense_fwd(const xv_act* x, const void* w_phases_packed, const float* zero_bias, const float* scale,
                        const float* shift, const xv_act* residual, const xv_act* y, int stride, int relu,
                        void* workspace, size_t workspace_bytes, void* stream);

/* Tuning / test entry: the same op with an explicit tile configuration 0 <= cfg < xv_conv2d_num_cfgs()
 * (cfg < 0 = the library's own choice, i.e. xv_conv2d_fwd).  Results are bit-identical across
 * configurations; XV_ESHAPE if the configuration cannot tile this shape.                            */
int xv_conv2d_fwd_cfg(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y,
                      const xv_act* pooled, int k, int relu, int cfg, void* stream);
/* The SAME 3x3 layer of TWO models in one launch: the two experts of a fusion model run identical layer shapes from conv1_2
 * on (basic_fusion_model.py:25-60 builds one fcn() per modality), each with its own maps, weights and bias.  The persistent
 * generation-4 / 5 kernel takes the concatenated tile lists, so the launch makes whole rounds of workgroups where each
 * expert alone leaves its last round half empty (conv4_x at 16 images of 768x384: 2 x 1152 tiles = 9 rounds of 256 CUs
 * against 2 x 5; conv5_x: 3 against 2 x 2).  Arguments per model as xv_conv2d_fwd (k = 3; y descriptors are required, their
 * data may be null with a pooled output); both models give the same set of outputs.  Results are bit-identical to two
 * xv_conv2d_fwd calls.  XV_ESHAPE -- nothing launched -- where the shape does not run on generation 4 / 5 (bf16 maps that
 * tile exactly in 16x32, or in 24x16 / 32x16 without a pooled output): launch the two convs separately then.          */
int xv_conv2d_fwd_pair(const xv_act* xa, const void* wa_packed, const float* bias_a, const xv_act* ya, const xv_act* pooled_a,
                       const xv_act* xb, const void* wb_packed, const float* bias_b, const xv_act* yb, const xv_act* pooled_b,
                       int relu, void* stream);
/* Stream-K tail of the generation-2 3x3 kernel (conv_dma_kernel).  A persistent grid walks (pixel patch, cout tile)
 * tiles round by round; when the last round is incomplete (conv4_x / conv5_x at 16 images: 4.5 / 1.5 rounds; one image:
 * fewer tiles than CUs from conv2 on) its (tile, 32-channel chunk) items are dealt out evenly over ALL workgroups, the
 * partial sums of a split tile meet in fp32 slabs of this workspace and the workgroup that arrives last adds them in
 * a fixed order (bitwise reproducible) and runs the epilogue.  `workspace`: xv_conv2d_streamk_workspace_bytes() bytes,
 * 16-byte aligned, ZERO when first used (the arrival counters; every launch leaves them zero again), owned by ONE
 * stream at a time (two experts on two streams need one each).  NULL = every tile whole (the plain entry points).
 * Same arguments and results otherwise as xv_conv2d_fwd_cfg / xv_conv2d_bwd_data; kernels other than generation 2
 * ignore the workspace.                                                                                               */
size_t xv_conv2d_streamk_workspace_bytes(void);
int xv_conv2d_fwd_ws(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, const xv_act* pooled,
                     int k, int relu, int cfg, void* workspace, size_t workspace_bytes, void* stream);
/* The forward conv with a workspace for the SPLIT form of the 24x16-tile kernel (round 5, batch-1 latency): where a layer's
 * whole tiles fill less than half the CUs AND its shape sums its input channels in chunk groups (maps that tile in 24x16 and
 * not in 16x32 with >= 256 input channels: conv5_x of a 768x384 input), every (tile, group) is a work item of its own that
 * leaves fp32 accumulators in the workspace, and a second launch adds the groups in order.  The unsplit launch of such a
 * layer adds the same groups in the same order in registers: identical bits at every batch size.
 * xv_conv2d_split_workspace_bytes: bytes this shape needs (0: it is never split; the call below then equals xv_conv2d_fwd_cfg).
 * The workspace belongs to the calling stream until the launch has completed.                                          */
size_t xv_conv2d_split_workspace_bytes(int n, int h, int w, int cin, int cout);
int xv_conv2d_fwd_split(const xv_act* x, const void* w_packed, const float* bias, const xv_act* y, const xv_act* pooled, int k,
                        int relu, int cfg, void* split_workspace, size_t split_workspace_bytes, void* stream);
int xv_conv2d_bwd_data_ws(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias, const xv_act* relu_ref,
                          const xv_act* addend, const xv_act* dx, int k, void* workspace, size_t workspace_bytes,
                          void* stream);
int xv_conv2d_num_cfgs(void);
/* The tile configuration the library would choose for this shape (what cfg < 0 resolves to), without launching anything:
 * host arithmetic only.  in_dtype / out_dtype: XV_BF16 | XV_FP8; flags: bit 0 = with a pooled output, bit 1 = the
 * data-gradient epilogue (addend / relu mask), bit 2 = with a stream-K workspace.  Returns the configuration index or
 * XV_ESHAPE / XV_EINVAL.                                                                                              */
int xv_conv2d_choose_cfg(int n, int h, int w, int cin, int cout, int k, int in_dtype, int out_dtype, int flags);

/* First layer: conv1_1 = relu(conv3x3(x) + b) on the RAW float32 network input (dense unpadded
 * NHWC, cin = 1..4; simple_fcn.py:39), fp32 weights HWIO [3][3][cin][64], fp32 math, bf16 out.
 * cin = 1 / 3 with w % 16 == 0 runs on the matrix cores with every fp32 operand split exactly into
 * three bf16 terms (fp32 accumulation; rounds like the fp32 FMA chain the other shapes use).        */
int xv_conv2d_first_fwd(const float* x, int n, int h, int w, int cin, const float* w_hwio,
                        const float* bias, const xv_act* y, int relu, void* stream);
/* AdapNet's block_0_1 written straight into the operand of block_0_2 (adapnet.py:126-127): the same values as
 * xv_conv2d_first_fwd + xv_gather_conv7s2, without the 64-channel full-resolution map in between.  cin = 1 or 3, w % 16 == 0,
 * h even; z [N,h/2,w/2,576] bf16, which must hold zeros where the gather has no source pixel (variant 1 of the last operand
 * row / column: neither this kernel nor xv_gather_conv7s2 writes anything else there).                                   */
int xv_conv2d_first_gather7s2_fwd(const float* x, int n, int h, int w, int cin, const float* w_hwio,
                                  const float* bias, const xv_act* z, int relu, void* stream);
/* conv1_1 AND conv1_2 (+ max_pooling2d) of the FCN trunk in one launch (simple_fcn.py:39-41), inference only: conv1_1 is
 * evaluated tile by tile straight into conv1_2's LDS patch buffers, so its 64-channel map never reaches memory.  x / w1_hwio
 * / b1 as xv_conv2d_first_fwd (cin = 1 or 3), w2_packed / b2 as xv_conv2d_fwd (64 -> 64 channels, 3x3); y and / or pooled
 * as xv_conv2d_fwd (either may be NULL or have NULL data).  Bit-identical to the two separate calls.  Maps that tile exactly
 * in 16x32 only: XV_ESHAPE otherwise (run the two kernels).
 * y / pooled may be e4m3 maps (XV_FP8, one scale for both: value * 2^-scale_exp, saturating at +-448): the first e4m3 map of
 * the fp8 graph, the same bytes as xv_conv2d_first_fwd + xv_conv2d_fwd onto e4m3 maps.                                  */
int xv_conv_first_pair_fwd(const float* x, int n, int h, int w, int cin, const float* w1_hwio, const float* b1, int relu1,
                           const void* w2_packed, const float* b2, int relu2, const xv_act* y, const xv_act* pooled,
                           void* stream);


/* 2x2 stride-2 'valid' max pooling (max_pooling2d, simple_fcn.py:41,44,48,58).                  */
int xv_maxpool2x2_fwd(const xv_act* x, const xv_act* y, void* stream);

/* y = relu(bilinear_x2(x)) [+ residual]: deconv2d(k=4, strides 2, 'same', constant bilinear
 * diagonal kernel) + relu, then tf.add_n with score_conv4 (custom_layers.py:8-25,71-121;
 * simple_fcn.py:82-85), computed as the depthwise 2x2-tap interpolation the diagonal kernel
 * collapses to.  residual may be NULL.                                                           */
int xv_upsample2x_relu_add(const xv_act* x, const xv_act* residual, const xv_act* y, void* stream);

/* The same with an inference batch norm between the deconv and its relu (deconv2d(..., batch_normalization=True),
 * custom_layers.py:112-119): y = relu(bilinear_x2(x) * scale[c] + shift[c]) [+ residual]; scale / shift float32 [C]
 * (gamma / sqrt(moving_variance + 1e-3), beta - moving_mean * scale), both NULL = no batch norm.                   */
int xv_upsample2x_affine_relu_add(const xv_act* x, const float* scale, const float* shift, const xv_act* residual,
                                  const xv_act* y, void* stream);

/* The general form: y = act(bilinear_x2(x) * scale[c] + shift[c]) [+ residual], relu = 0 for AdapNet's
 * first_deconvolution_upconv (deconv2d(activation=None, batch_normalization=True) then tf.add with the block-7
 * shortcut, adapnet.py:157-163).                                                                                      */
int xv_upsample2x_affine_act_add(const xv_act* x, const float* scale, const float* shift, const xv_act* residual,
                                 const xv_act* y, int relu, void* stream);

/* tf.layers.dropout(x, rate, training=True) at the MC-dropout sites of encoder / decoder (simple_fcn.py:50-62,71-78,
 * 124-126; enabled only by the uncertainty models, bayesian_fcn.py:74-89): y = x / (1 - rate) where a counter-based
 * random draw per element (a function of `seed` and the element index only) keeps it, else 0.  Not TensorFlow's random
 * stream: masks are reproducible per seed, not comparable with the reference's.                                    */
int xv_dropout(const xv_act* x, const xv_act* y, float rate, uint64_t seed, void* stream);

/* y = concat(a, b) along channels (tf.concat(axis=3) of the two trunks' conv4_3 / conv5_3, fusion_fcn.py:27-28). */
int xv_concat_channels(const xv_act* a, const xv_act* b, const xv_act* y, void* stream);

/* ---- gathers that map AdapNet's strided / dilated convs onto the stride-1 kernels above (adapnet.py:12-173) ----
 * xv_subsample2: y[i][j] = x[2i][2j], the input of a 1x1 stride-2 conv (block_a with strides 2, adapnet.py:39,45).
 * xv_gather_conv7s2: z [N,H/2,W/2,9C] such that the 7x7 stride-2 'same' conv of block_0_2 (adapnet.py:127) equals a 3x3
 *   stride-1 conv of z: group g = 3*rv + cv, variants (parity, shift) = (0,0), (0,1), (1,0) per axis,
 *   z[j][i][g] = x[2(j+sr)+pr][2(i+sc)+pc] (zero outside); the 7x7 tap k = 2(t+s)+p lands on 3x3 tap t.
 * xv_im2col_dilated_pair: z [N,H,W,18C], z[y][x][t] = x[y+ty*d][x+tx*d] with t = 0..8 at rate d1 and 9..17 at rate d2:
 *   the two atrous convs of block_b (adapnet.py:84-88) become one 1x1 conv whose output is their concatenation.     */
int xv_subsample2(const xv_act* x, const xv_act* y, void* stream);
int xv_gather_conv7s2(const xv_act* x, const xv_act* z, void* stream);
int xv_im2col_dilated_pair(const xv_act* x, int dilation1, int dilation2, const xv_act* z, void* stream);
/* The same two atrous convs + concat WITHOUT the 18C operand: an implicit GEMM over nine taps per output half, the taps
 * gathered by the kernel's own loads (adapnet.py:84-88: stage_2_1, stage_2_2, tf.concat).  wpk: the packed [1,1,18C,F] image
 * the materialised form's 1x1 conv takes (rows [0,9C) x columns [0,F/2) = conv 1, rows [9C,18C) x [F/2,F) = conv 2; the two
 * off-diagonal blocks are never read -- except F = 64, where both halves share a tile and all 18 taps are multiplied).  C a
 * multiple of 64, F of 256 or F = 64; y [N,H,W,F]; the same bits as xv_im2col_dilated_pair +
 * xv_conv2d_fwd(ksize 1).  XV_ESHAPE otherwise (callers then take the materialised form).                              */
int xv_conv_dilated_pair_fwd(const xv_act* x, const void* wpk, const float* bias, int dilation1, int dilation2, int relu,
                             const xv_act* y, void* stream);
/* The pair's gradients, also without the operand (training; adapnet.py:84-88 behind base_model.py:153-162).
 * xv_conv_dilated_pair_bwd_data: dx [N,H,W,C] = gradient of the pair's input from dy [N,H,W,F] (both halves): an implicit GEMM
 *   over 18 taps x F/2 channels; wpk_dgrad = the packed forward-format image (xv_pack_conv_weights) of the [1,1,18 F/2,C]
 *   kernel whose row (h * 9 + t) * F/2 + co holds k_h[t][:, co] of conv h; F and C multiples of 128.
 * xv_conv_dilated_pair_bwd_filter_ws: dw1 / dw2 (+=) = HWIO [3][3][C][F/2] gradients of the two kernels; the taps of x gathered
 *   by the kernel's loads, split-K slabs in the workspace (>= ..._workspace_bytes; 0 = shape not taken: C and F multiples of
 *   256) added in a fixed order -- bitwise reproducible.  XV_ESHAPE: callers take the im2col operand + the 1x1 entries.   */
int xv_conv_dilated_pair_bwd_data(const xv_act* dy, const void* wpk_dgrad, const float* zero_bias, int dilation1, int dilation2,
                                  const xv_act* dx, void* stream);
size_t xv_conv_dilated_pair_bwd_filter_workspace_bytes(int n, int h, int w, int c, int f);
int xv_conv_dilated_pair_bwd_filter_ws(const xv_act* x, const xv_act* dy, int dilation1, int dilation2, float* dw1, float* dw2,
                                       void* workspace, size_t workspace_bytes, void* stream);
/* Their transposes for the training graph (each destination element summed by one thread: deterministic), and the
 * residual add y = a + b of a block whose closing conv is followed by a batch norm (adapnet.py:38-51).             */
int xv_subsample2_bwd(const xv_act* dy, const xv_act* dx, void* stream);
int xv_gather_conv7s2_bwd(const xv_act* dz, const xv_act* dx, void* stream);
int xv_im2col_dilated_pair_bwd(const xv_act* dz, int dilation1, int dilation2, const xv_act* dx, void* stream);
int xv_add(const xv_act* a, const xv_act* b, const xv_act* y, void* stream);
/* Phase shuffles of AdapNet's two TRAINABLE transposed convs (adapnet.py:155-163 calls custom_layers.deconv2d:71-121
 * without trainable=False): a k = 2*stride conv2d_transpose is one 3x3 conv onto stride*stride*C phase channels at the
 * input resolution (xv_conv2d_fwd on the kernel custom_layers.dense_deconv_as_conv3x3 arranges) + a depth-to-space
 * shuffle; its filter / data gradients are xv_conv2d_bwd_filter / xv_conv2d_bwd_data of the space-to-depth shuffle of
 * the upstream gradient.  Phase channel (py*stride + px)*C + c <-> output pixel (stride*qy + py, stride*qx + px).
 *   xv_space_to_depth        g [N,s*H,s*W,C] -> out [N,H,W,s*s*C]                         (C % 8 == 0)
 *   xv_space_to_depth_dense  the same from a dense float32 [N,s*H,s*W,num_classes] gradient; out->c = s*s*Cp with
 *                            Cp % 8 == 0, Cp >= num_classes, the padding channels written as zeros
 *   xv_depth_to_space_dense  phase map z [N,H,W,s*s*Cp] -> dense float32 [N,s*H,s*W,num_classes], optional per-class
 *                            scale / shift (the inference batch norm of second_deconvolution_upconv)               */
int xv_space_to_depth(const xv_act* g, int stride, const xv_act* out, void* stream);
int xv_space_to_depth_dense(const float* g, int num_classes, int stride, const xv_act* out, void* stream);
int xv_depth_to_space_dense(const xv_act* z, int stride, int num_classes, const float* scale, const float* shift,
                            float* out, void* stream);
/* The float32 route of AdapNet's trained x8 score deconv (adapnet.py:155-163 computes the class scores in float32): the
 * padded bf16 map as a dense float32 one (exact), the 3x3 conv onto the 64 phases x Cp classes through xv_conv2d_f32 (fp32
 * matrix instruction), and the phases unshuffled from the dense float32 phase map [n][hq][wq][stride*stride*cp]
 * (cp a multiple of 4) [* scale + shift] -- no bf16 rounding between the deconv and the batch norm / softmax.          */
int xv_act_to_dense_f32(const xv_act* x, float* out, void* stream);
int xv_depth_to_space_dense_f32(const float* z, int n, int hq, int wq, int stride, int cp, int num_classes, const float* scale,
                                const float* shift, float* out, void* stream);

/* Decoder head: upscore = relu(bilinear_x8(fused)) (deconv2d k=16 s=8, simple_fcn.py:129-130),
 * score = conv1x1(upscore, Ws) + bs (no activation, simple_fcn.py:131-133), prob = softmax(score),
 * label = argmax(prob, 3) (basic_fusion_model.py:21-22 / simple_fcn.py:223-224), fp32 from the bf16 `fused`
 * features.  `fused` is non-negative by construction (a sum of two relu outputs) and the bilinear
 * weights are positive, so the relu is the identity and the linear x8 deconv and 1x1 conv commute: the
 * 1x1 conv runs at 1/8 resolution into `workspace` and C class scores are interpolated instead of U
 * features; the full-resolution U-channel tensor is never formed.
 * w_score: float32 [U][C] (HWIO of the 1x1 kernel), b_score: float32 [C], C <= 32, U % 8 == 0.
 * Outputs (each may be NULL): score / prob float32 [N][8h][8w][C], label int64 [N][8h][8w].
 * workspace: xv_decoder_head_workspace_bytes(n, h, w, C) bytes, 16-byte aligned (h, w of `fused`).      */
size_t xv_decoder_head_workspace_bytes(int n, int h, int w, int num_classes);
int xv_decoder_head_fwd(const xv_act* fused, const float* w_score, const float* b_score, int num_classes,
                        float* score, float* prob, int64_t* label, void* workspace, size_t workspace_bytes,
                        void* stream);

/* General decoder head for a batch norm with a non-zero shift between the x8 deconv and its relu (the default of
 * decoder() as called by fusion_fcn.py:38; custom_layers.py:112-119): upscore = relu(bilinear_x8(fused) * scale[u] +
 * shift[u]), then score / softmax / argmax as above.  The 1x1 conv no longer commutes with the interpolation, so
 * all U features are interpolated per pixel (16x the FMAs of xv_decoder_head_fwd; no workspace).                 */
int xv_decoder_head_affine_fwd(const xv_act* fused, const float* scale, const float* shift, const float* w_score,
                               const float* b_score, int num_classes, float* score, float* prob, int64_t* label,
                               void* stream);

/* prob = softmax(score), label = argmax(prob) on a dense float32 [npix][C] score tensor
 * (tf.nn.softmax + tf.argmax, basic_fusion_model.py:21-22); lowest index wins ties.  prob / label
 * may be NULL.                                                                                    */
int xv_softmax_argmax(const float* score, int64_t npix, int num_classes, float* prob, int64_t* label,
                      void* stream);

/* ---- fusion --------------------------------------------------------------------------------
 * Bayes fusion of E experts' label maps (bayes_mix.py:12-58 + tf.argmax at :161):
 *   score[c] = sum_e loglik[e][label_e][c] + logprior[c],  fused = argmax_c score[c]
 * labels: E device pointers (host array of device pointers) to int64 [npix]; loglik: float32
 * [E][C][C] = log(1e-20 + cond_e) computed by the host in fp32; logprior float32 [C].
 * score_out (float32 [npix][C]) may be NULL.  E <= 4, C <= 32.                                     */
int xv_bayes_fuse(const int64_t* const* labels, int num_experts, const float* loglik, const float* logprior,
                  int num_classes, int64_t npix, int64_t* fused, float* score_out, void* stream);

/* Same decision through the C^2 lookup table of bayes_decision_matrix (bayes_mix.py:61-112;
 * experiments/timing.py:87-115): fused = lut[label_a][label_b], lut int64 [C][C].                  */
int xv_bayes_fuse_lut(const int64_t* label_a, const int64_t* label_b, const int64_t* lut, int num_classes,
                      int64_t npix, int64_t* fused, void* stream);

/* Dirichlet fusion of E experts' softmax vectors (dirichlet_mix.py:14-36,96-136):
 *   p_e <- p_e / sum(p_e);  L_e[c] = sum_k am1[e][c][k] * log(1e-20 + p_e[k]) - lognorm[e][c]
 *   fused_score[c] = sum_e L_e[c] + logprior[c];  label = argmax_c
 * am1: float32 [E][C][C] with am1[e][c][k] = sigma*A_e[k][c] - 1; lognorm: float32 [E][C] =
 * sum_k lgamma(sigma*A_e[k][c]) - lgamma(sum_k ...); logprior = log(1e-20 + prior) float32 [C].
 * probs: E device pointers to float32 [npix][C].  score_out may be NULL.  E <= 4, C <= 32.          */
int xv_dirichlet_fuse(const float* const* probs, int num_experts, const float* am1, const float* lognorm,
                      const float* logprior, int num_classes, int64_t npix, int64_t* fused, float* score_out,
                      void* stream);

/* Mean of the experts' probabilities then argmax (average_mix.py:18-21).                         */
int xv_average_fuse(const float* const* probs, int num_experts, int num_classes, int64_t npix,
                    int64_t* fused, void* stream);

/* ---- training: backward kernels and optimizers ---------------------------------------------------
 * These replace the gradient graph tf.train.{Adam,RMSProp,Adagrad}Optimizer.minimize(self.loss)
 * builds (base_model.py:153-162) over SimpleFCN's training graph (simple_fcn.py:200-214).
 * Activations and activation gradients are bf16 padded NHWC; weight gradients and optimizer
 * state are float32 in the reference's HWIO layout.  Gradient buffers are ACCUMULATED into
 * (callers zero them once per step).                                                               */

/* Packed weights of the data-gradient convolution: Wd[k*k-1-tap][co][ci] = W[tap][ci][co]
 * (xv_packed_weight_bytes(k, cout, cin) bytes).                                                     */
/* All kernels of a model in one launch (the re-pack after every optimizer step): `table_device` = n descriptors in DEVICE
 * memory; packed_dgrad may be NULL per entry.  Shapes as xv_pack_conv_weights / _dgrad (k in {1, 3}, cin, cout % 64 == 0). */
typedef struct xv_pack_desc {
  const float* w_hwio;
  void* packed;
  void* packed_dgrad;
  int32_t k, cin, cout, reserved;
} xv_pack_desc;
int xv_pack_conv_weights_multi(const xv_pack_desc* table_device, int n, void* stream);
/* hipMemsetAsync(p, 0, bytes) on `stream`: the step's own clear of its gradient / loss accumulators (no framework kernel). */
int xv_memset_zero(void* p, size_t bytes, void* stream);
int xv_pack_conv_weights_dgrad(const float* w_hwio, void* packed, int k, int cin, int cout, void* stream);

/* dx = (conv(dy, Wd) + addend) * (relu_ref > 0): Conv2DBackpropInput of tf.layers.conv2d, the AddN
 * where two gradient paths meet (addend, may be NULL) and the ReluGrad of the layer below
 * (relu_ref = that layer's forward output, may be NULL), fused in the forward MFMA kernel.
 * zero_bias: float32 [dx->c] zeros.                                                                */
int xv_conv2d_bwd_data(const xv_act* dy, const void* w_packed_dgrad, const float* zero_bias,
                       const xv_act* relu_ref, const xv_act* addend, const xv_act* dx, int k, void* stream);

/* dW[tap][cin][cout] += sum_pixels x[pix+tap][cin] * dy[pix][cout] (Conv2DBackpropFilter), on MFMA with
 * transposing LDS reads; dbias[cout] += sum_pixels dy (BiasAddGrad; may be NULL).  k = 1 or 3,
 * cin % 64 == 0, cout % 64 == 0.  fp32 atomics: summation order is not fixed run to run.            */
int xv_conv2d_bwd_filter(const xv_act* x, const xv_act* dy, float* dw_hwio, float* dbias, int k, void* stream);
/* Same with a workspace of xv_conv2d_bwd_filter_workspace_bytes(n, h, w, cin, cout, k) bytes: the split-K
 * partial sums go to per-split slabs with plain stores and are added into dw by a second kernel in a fixed
 * order -- bitwise reproducible and faster than the fp32 atomics (~1.3 TB/s) for 3x3 layers with at least four
 * 64x64 channel-block pairs; smaller layers keep the atomics (the slabs would be all reduce and no compute). */
size_t xv_conv2d_bwd_filter_workspace_bytes(int n, int h, int w, int cin, int cout, int k);
int xv_conv2d_bwd_filter_ws(const xv_act* x, const xv_act* dy, float* dw_hwio, float* dbias, int k,
                            void* workspace, size_t workspace_bytes, void* stream);

/* Test / tuning switch for the 3x3 filter-gradient kernel: 1 = register-staged tiles, two 4-wave workgroups
 * per CU; 2 = LDS-DMA double-buffered tiles, one 8-wave workgroup per CU (round 5's default); 3 (default) = the
 * same tiles with four loader waves beside four compute waves that walk the halo rows (three X fragments per
 * row serve all nine taps); 0 = back to the default (XV_WGRAD_VARIANT in the environment, else 3).  Same sums,
 * equal on integers; the fp32 summation order differs between the variants.                           */
int xv_set_wgrad_variant(int variant);

/* dbias[c] += sum over all pixels of dy[.., c] (BiasAddGrad).                                        */
int xv_bias_grad(const xv_act* dy, float* dbias, void* stream);

/* conv1_1: dW[3][3][cin][64] += sum_pixels x[pix+tap][cin] * dy[pix][:] on the raw float32 input, and
 * dbias[64] += sum_pixels dy (may be NULL).                                                         */
int xv_conv2d_first_bwd_filter(const float* x, int n, int h, int w, int cin, const xv_act* dy, float* dw_hwio,
                               float* dbias, void* stream);
/* The same with per-workgroup partial sums in `workspace` (xv_conv2d_first_bwd_filter_workspace_bytes) and a fixed-order
 * reduce instead of fp32 atomics: bitwise reproducible.  dbias must not be NULL.  NULL workspace = the plain form.       */
size_t xv_conv2d_first_bwd_filter_workspace_bytes(int n, int h, int w, int cin);
int xv_conv2d_first_bwd_filter_ws(const float* x, int n, int h, int w, int cin, const xv_act* dy, float* dw_hwio,
                                  float* dbias, void* workspace, size_t workspace_bytes, void* stream);

/* MaxPoolGrad + ReluGrad: dy = dpooled routed to the first maximum of each 2x2 window, zero where y <= 0. */
int xv_maxpool2x2_bwd(const xv_act* y, const xv_act* dpooled, const xv_act* dy, void* stream);

/* ReluGrad: out = ref > 0 ? g : 0.                                                                 */
int xv_relu_bwd(const xv_act* g, const xv_act* ref, const xv_act* out, void* stream);

/* Gradient of fused = s4 + relu(bilinear_x2(s5)) w.r.t. s5 (through score_conv5's own relu).       */
int xv_upsample2x_bwd(const xv_act* dfused, const xv_act* s5, const xv_act* ds5, void* stream);

/* count += #pixels with 0 <= label < C  (the denominator of utils.py:52).                          */
int xv_count_valid_labels(const int32_t* labels, int num_classes, int64_t npix, int64_t* count, void* stream);

/* S = fused . Ws at 1/8 resolution into a zero-bordered float32 [N][h+2][w+2][CM] buffer, CM = C rounded up
 * to 4: the first half of the decoder head in its commuted form (see xv_decoder_head_fwd).          */
int xv_score_lowres(const xv_act* fused, const float* w_score, int num_classes, float* S, void* stream);

/* Fused head of a two-expert fusion model: both experts' low-resolution class scores (xv_score_lowres of each `fused`
 * map) -> per-pixel x8 bilinear logits + bias -> each expert's softmax / argmax (basic_fusion_model.py:21-22) -> Bayes
 * fusion of the two labels (mode 0; bayes_mix.py:33-58,161: tab = loglik [2][C][C]) or Dirichlet fusion of the two
 * probability vectors (mode 1; dirichlet_mix.py:14-36,96-136: tab = alpha-1 [2][C][C], lognorm [2][C]) -> the fused
 * label map int64 [n][8hi][8wi], and nothing else: no per-pixel intermediate reaches HBM.  Same arithmetic, term for
 * term, as xv_decoder_head_fwd + xv_bayes_fuse / xv_dirichlet_fuse (bit-identical labels).                        */
int xv_fused_head_fwd(const float* Sa, const float* Sb, const float* bias_a, const float* bias_b, int n, int hi, int wi,
                      int num_classes, int mode, const float* tab, const float* lognorm, const float* logprior,
                      int64_t* fused_label, void* stream);

/* Loss and head backward (simple_fcn.py:212-214, utils.py:43-53) in the same commuted form: recomputes
 * score = bilinear_x8(fused . Ws) + bs, adds -sum(onehot*log_softmax)/(1e-20+count) to *loss, accumulates
 * d(score kernel) [U][C] and d(score bias) [C], and writes dfused = d(loss)/d(fused) (bf16 padded NHWC; the
 * relu masks below `fused` are applied by the callers' next kernels).  valid_count must already hold the
 * batch's count.  workspace: xv_decoder_head_bwd_workspace_bytes(n, h, w, C) bytes, 16-byte aligned (the padded
 * 1/8-resolution scores, the row-weighted column sums of the score gradient -- the dense gradient is never stored --
 * and one slab of score-weight partial sums per 256 low-resolution pixels, reduced in a fixed order).   */
size_t xv_decoder_head_bwd_workspace_bytes(int n, int h, int w, int num_classes);
int xv_decoder_head_bwd(const xv_act* fused, const float* w_score, const float* b_score, const int32_t* labels,
                        const int64_t* valid_count, int num_classes, double* loss, float* dw_score,
                        float* db_score, const xv_act* dfused, void* workspace, size_t workspace_bytes,
                        void* stream);

/* [TF1] optimizers on flat float32 buffers; grad_scale multiplies the gradient first (1/world_size).
 * Adam: m,v updated, p -= lr_t*m/(sqrt(v)+eps) with lr_t = lr*sqrt(1-b2^t)/(1-b1^t) from the host.
 * RMSProp: ms = decay*ms+(1-decay)g^2 (ms initialised to 1), p -= lr*g/sqrt(ms+eps).
 * Adagrad: a += g^2 (a initialised to 0.1), p -= lr*g/sqrt(a).                                      */
int xv_adam_step(float* param, const float* grad, float* m, float* v, int64_t n, float lr_t, float beta1,
                 float beta2, float eps, float grad_scale, void* stream);
int xv_rmsprop_step(float* param, const float* grad, float* ms, int64_t n, float lr, float decay, float eps,
                    float grad_scale, void* stream);
int xv_adagrad_step(float* param, const float* grad, float* accum, int64_t n, float lr, float grad_scale,
                    void* stream);

/* ---- statistics ----------------------------------------------------------------------------
 * Dirichlet sufficient statistics (dirichlet_mix.py:142-163): for every pixel with 0 <= label < C
 *   S[label][k] += log(1e-10 + prob[k]),  counts[label] += 1
 * S: float64 [C][C] (accumulated, not zeroed), counts: int64 [C] (accumulated).  labels int32.     */
int xv_dirichlet_suffstats(const float* prob, const int32_t* labels, int num_classes, int64_t npix,
                           double* S, int64_t* counts, void* stream);

/* Confusion matrix (base_model.py:136-151): cm[label][pred] += 1 for 0 <= label < C (negative
 * labels are the dropped (C+1)-th row); cm int64 [C][C], accumulated, rows = ground truth.         */
int xv_confusion_matrix(const int32_t* labels, const int64_t* pred, int num_classes, int64_t npix,
                        int64_t* cm, void* stream);
/* Label maps on their way to the host (predict(), base_model.py:279-288): int64 labels in [0, 256) -> one byte per pixel,
 * out[i] = (uint8) labels[i].  The pipelined host boundary sends this image over PCIe (an eighth of the bytes) and widens it
 * to the reference's np.int64 while it fills the result array.  labels 16-byte, out 8-byte aligned.                       */
int xv_narrow_labels(const int64_t* labels, int64_t n, uint8_t* out, void* stream);

/* ---- training-mode batch normalisation -----------------------------------------------------------
 * tf.layers.batch_normalization(training=True) between a conv / deconv and its activation
 * (custom_layers.py:112-119,124-139): per-channel statistics over (N, H, W), normalisation with the biased batch
 * variance, [TF1] epsilon 1e-3 and momentum 0.99 (moving variance fed the unbiased estimate), and its gradient.
 * z: pre-normalisation conv output, y: post-relu output (relu mask of the backward; NULL = no activation).
 * sums: double [2*C] scratch (zeroed inside).  C in {64, 128, 256, 512, ...}: C/8 must divide 256.              */
int xv_bn_stats(const xv_act* z, double* sums, void* stream);
int xv_bn_finalize(const double* sums, int channels, int64_t count, const float* gamma, const float* beta, float eps,
                   float momentum, float* moving_mean, float* moving_var, float* mean, float* invstd, float* scale,
                   float* shift, void* stream);
/* xv_bn_sums_from_rows + xv_bn_finalize, and xv_bn_stats_ws + xv_bn_finalize, with the row sums and the per-channel results
 * in ONE launch (one workgroup per channel; `sums` is left as the two-call form leaves it, bit for bit).  Single-process
 * statistics only: a data-parallel run all-reduces `sums` between the two calls and keeps them.                        */
int xv_bn_finalize_from_rows(const float* rows, int nrows, int channels, int64_t count, const float* gamma, const float* beta,
                             float eps, float momentum, float* moving_mean, float* moving_var, float* mean, float* invstd,
                             float* scale, float* shift, double* sums, void* stream);
int xv_bn_stats_finalize_ws(const xv_act* z, double* sums, void* workspace, size_t workspace_bytes, const float* gamma,
                            const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* mean,
                            float* invstd, float* scale, float* shift, void* stream);
int xv_bn_apply(const xv_act* z, const float* scale, const float* shift, int relu, const xv_act* y, void* stream);
/* dz = gamma*invstd*(g - mean(g) - zhat*mean(g*zhat)), g = dy*(y>0); dgamma += sum g*zhat, dbeta += sum g.      */
int xv_bn_bwd(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean, const float* invstd,
              const float* gamma, double* sums, float* dgamma, float* dbeta, const xv_act* dz, void* stream);
/* The gradient in two steps for data-parallel callers (Sync-BN): xv_bn_bwd_reduce fills `sums` with the LOCAL
 * sum g / sum g*zhat and adds them into dgamma / dbeta; the caller may all-reduce `sums`; xv_bn_bwd_apply then forms
 * dz from `sums` and `count` (the global N*H*W).                                                                    */
int xv_bn_bwd_reduce(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean, const float* invstd,
                     double* sums, float* dgamma, float* dbeta, void* stream);
int xv_bn_bwd_apply(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean, const float* invstd,
                    const float* gamma, const double* sums, int64_t count, const xv_act* dz, void* stream);
/* The same two steps for a batch norm FOLLOWED BY A RELU, the mask recomputed from z (z * scale + shift > 0, the forward
 * pass's own expression: scale / shift as xv_bn_finalize left them) instead of read from the activation map: a third / a
 * quarter less HBM traffic.  64 <= C, C divides 2048.                                                                    */
int xv_bn_bwd_reduce_zmask(const xv_act* dy, const xv_act* z, const float* mean, const float* invstd, const float* scale,
                           const float* shift, double* sums, float* dgamma, float* dbeta, void* workspace,
                           size_t workspace_bytes, void* stream);
/* y = relu(z * scale + shift) and pooled = maxpool2x2(y) in one pass; y may be a NULL-data descriptor (only the pooled map is
 * written: the training step recomputes y from z in xv_bn_pool_bwd_*).  64 <= C, C divides 2048; h, w even.               */
int xv_bn_apply_pool(const xv_act* z, const float* scale, const float* shift, const xv_act* y, const xv_act* pooled,
                     void* stream);
/* The batch-norm gradient of a conv -> batch norm -> relu -> 2x2 max-pool block straight from the gradient of the POOLED map:
 * each pass recomputes the window's activations from z, routes the pooled gradient to the first positive maximum
 * (xv_maxpool2x2_bwd's rule on the same values) and reduces / applies -- no routed-gradient map in HBM.  count = n h w of z
 * (x ranks under Sync-BN); workspace as in the _ws forms below (may be NULL).                                            */
int xv_bn_pool_bwd_reduce(const xv_act* dpooled, const xv_act* z, const float* mean, const float* invstd, const float* scale,
                          const float* shift, double* sums, float* dgamma, float* dbeta, void* workspace,
                          size_t workspace_bytes, void* stream);
int xv_bn_pool_bwd_apply(const xv_act* dpooled, const xv_act* z, const float* mean, const float* invstd, const float* scale,
                         const float* shift, const float* gamma, const double* sums, int64_t count, const xv_act* dz,
                         void* stream);
/* Workspace forms of the reductions: with a workspace of xv_bn_workspace_bytes(C) bytes (16-byte aligned, owned by one
 * stream) every workgroup writes its partial sums to its own row and a second kernel adds the rows in a fixed tree --
 * bitwise reproducible statistics and gamma / beta gradients; workspace == NULL: f64 atomics in arrival order (the
 * forms without _ws).                                                                                                   */
size_t xv_bn_workspace_bytes(int channels);
int xv_bn_stats_ws(const xv_act* z, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int xv_bn_bwd_reduce_ws(const xv_act* dy, const xv_act* y, const xv_act* z, const float* mean, const float* invstd,
                        double* sums, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);
int xv_bn_dense_stats_ws(const float* z, int64_t rows, int channels, double* sums, void* workspace, size_t workspace_bytes,
                         void* stream);
int xv_bn_dense_bwd_reduce_ws(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                              const float* invstd, double* sums, float* dgamma, float* dbeta, void* workspace,
                              size_t workspace_bytes, void* stream);
int xv_bn_bwd_apply_zmask(const xv_act* dy, const xv_act* z, const float* mean, const float* invstd, const float* scale,
                          const float* shift, const float* gamma, const double* sums, int64_t count, const xv_act* dz,
                          void* stream);

/* The batch norm BEHIND the x8 deconv without its input in memory: `low` is the deconv's input (padded bf16, 64 .. 2048
 * channels), the normalised map is z = bilinear_x8(low) rounded to bf16 -- exactly what xv_upsample_raw_fwd(low, 8) would
 * store -- recomputed per element in the four passes that would read it back (custom_layers.py:112-119 behind
 * simple_fcn.py:117-119 in training).  Same results, bit for bit, as the entry points without `_ups8` on the stored map.      */
int xv_bn_stats_finalize_ups8_ws(const xv_act* low, double* sums, void* workspace, size_t workspace_bytes, const float* gamma,
                                 const float* beta, float eps, float momentum, float* moving_mean, float* moving_var, float* mean,
                                 float* invstd, float* scale, float* shift, void* stream);
/* (round 6) the statistics pass alone, for data-parallel runs: sums[c] = sum z, sums[C + c] = sum z^2 over the recomputed map;
 * the caller all-reduces `sums` and finalises with xv_bn_finalize on the global pixel count.                                   */
int xv_bn_stats_ups8_ws(const xv_act* low, double* sums, void* workspace, size_t workspace_bytes, void* stream);
int xv_bn_apply_ups8(const xv_act* low, const float* scale, const float* shift, int relu, const xv_act* y, void* stream);
/* ... the forward apply pass FUSED with the dense score conv behind it: y (as xv_bn_apply_ups8 with relu) and score = y . W + b
 * (as xv_score_dense_fwd) in one launch -- 64 units, at most 16 classes; XV_ESHAPE elsewhere.                                  */
int xv_score_dense_fwd_ups8(const xv_act* low, const float* scale, const float* shift, const float* w_score,
                            const float* b_score, int num_classes, const xv_act* y, float* score, void* stream);
int xv_bn_bwd_reduce_zmask_ups8(const xv_act* dy, const xv_act* low, const float* mean, const float* invstd, const float* scale,
                                const float* shift, double* sums, float* dgamma, float* dbeta, void* workspace,
                                size_t workspace_bytes, void* stream);
int xv_bn_bwd_apply_zmask_ups8(const xv_act* dy, const xv_act* low, const float* mean, const float* invstd, const float* scale,
                               const float* shift, const float* gamma, const double* sums, int64_t count, const xv_act* dz,
                               void* stream);
/* The same on a dense float32 [rows][C] tensor, C <= 32 (the batch norm on `score`, simple_fcn.py:131-133).        */
int xv_bn_dense_stats(const float* z, int64_t rows, int channels, double* sums, void* stream);
int xv_bn_dense_apply(const float* z, int64_t rows, int channels, const float* scale, const float* shift, float* y,
                      void* stream);
int xv_bn_dense_bwd(const float* dy, const float* z, int64_t rows, int channels, const float* mean, const float* invstd,
                    const float* gamma, double* sums, float* dgamma, float* dbeta, float* dz, void* stream);
int xv_bn_dense_bwd_reduce(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                           const float* invstd, double* sums, float* dgamma, float* dbeta, void* stream);
int xv_bn_dense_bwd_apply(const float* dy, const float* z, int64_t rows, int channels, const float* mean,
                          const float* invstd, const float* gamma, const double* sums, int64_t count, float* dz,
                          void* stream);

/* ---- un-commuted training head (batch norm between the x8 deconv and its relu) ---------------------
 * y = bilinear_x{2,8}(x) with no activation (deconv2d with the constant kernel, custom_layers.py:8-25,71-110) and its
 * transpose; score = u . Ws + bs per full-resolution pixel (dense float32 [N*H*W][C]); softmax cross-entropy over the
 * labelled pixels with the loss normalised by *valid_count (models/utils.py:43-53); gradients of the score conv.   */
int xv_upsample_raw_fwd(const xv_act* x, int factor, const xv_act* y, void* stream);
int xv_upsample_raw_bwd(const xv_act* dy, int factor, const xv_act* dx, void* stream);
/* The x8 gradient through per-block sums in a workspace (xv_upsample_raw_bwd_workspace_bytes(n, h, w, c) for the LOW-resolution
 * map [n][h][w][c]): every element of dy is read once instead of from four source pixels' 16x16 footprints.  Same sums in
 * another (fixed) order.  factor 2 or workspace == NULL: xv_upsample_raw_bwd.                                              */
size_t xv_upsample_raw_bwd_workspace_bytes(int n, int h, int w, int c);
int xv_upsample_raw_bwd_ws(const xv_act* dy, int factor, const xv_act* dx, void* workspace, size_t workspace_bytes, void* stream);
int xv_score_dense_fwd(const xv_act* u, const float* w_score, const float* b_score, int num_classes, float* score,
                       void* stream);
int xv_softmax_ce_dense(const float* logits, const int32_t* labels, const int64_t* valid_count, int num_classes,
                        int64_t npix, double* loss, float* dlogits, void* stream);
/* The same with the batch norm's affine on the scores applied inside (logits = scores * scale + shift, the expression of
 * xv_bn_dense_apply; scale == shift == NULL: plain logits): the normalised scores never go to HBM.                      */
int xv_softmax_ce_dense_affine(const float* scores, const float* scale, const float* shift, const int32_t* labels,
                               const int64_t* valid_count, int num_classes, int64_t npix, double* loss, float* dlogits,
                               void* stream);
/* The same with the loss added up in a FIXED order: every workgroup leaves its partial sum in `ws` (at least
 * xv_softmax_ce_dense_workspace_bytes(npix) bytes, 8-byte aligned, owned by the calling stream) and a second one-workgroup
 * launch adds them to *loss -- bit for bit the same loss on every run (the two entries above add the partials with a double
 * atomic in arrival order: equal to ~1e-16 relative).  ws == NULL: the arrival-order form.  scale == shift == NULL: plain
 * logits.                                                                                                               */
size_t xv_softmax_ce_dense_workspace_bytes(int64_t npix);
int xv_softmax_ce_dense_ws(const float* scores, const float* scale, const float* shift, const int32_t* labels,
                           const int64_t* valid_count, int num_classes, int64_t npix, double* loss, float* dlogits, void* ws,
                           size_t ws_bytes, void* stream);
int xv_score_dense_bwd(const xv_act* u, const float* dscore, const float* w_score, int num_classes, float* dw_score,
                       float* db_score, const xv_act* du, void* stream);
/* The same with the filter and bias gradients added in a FIXED order (64 units, at most 16 classes -- the matrix-core form):
 * every workgroup leaves its partial sums in `workspace` (xv_score_dense_bwd_workspace_bytes(n, h, w) bytes, 16-byte aligned),
 * two small launches add them onto dw_score / db_score -- bitwise reproducible from run to run (the entry above adds them by
 * fp32 atomics in arrival order).  workspace == NULL: that form.                                                          */
size_t xv_score_dense_bwd_workspace_bytes(int n, int h, int w);
int xv_score_dense_bwd_ws(const xv_act* u, const float* dscore, const float* w_score, int num_classes, float* dw_score,
                          float* db_score, const xv_act* du, void* workspace, size_t workspace_bytes, void* stream);

/* ---- "exact" mode (conv_dtype='fp32'): the FCN trunk in plain float32 on dense UNPADDED NHWC maps -- the reference
 * graph's own arithmetic type (tf.layers.conv2d / max_pooling2d / conv2d_transpose on float32: simple_fcn.py:39-87,
 * custom_layers.py:71-139), no bf16 storage.  The convs run on the fp32 matrix instruction (v_mfma_f32_32x32x2_f32: bit for
 * bit a k-ordered fmaf chain, at the fp32 vector rate = 1/16 of the bf16 MFMA rate); the mode exists so that label maps
 * EQUAL to the fp32 oracle's on trained weights are a product mode and not only a test (basic_fusion_model.py:21-22).
 *   xv_conv2d_f32             y = [relu](conv_kxk_same(x, w_hwio) + bias), k in {1, 3}, any channel counts and map sizes
 *   xv_conv2d_f32_pool        the same; y and / or the 2x2 max-pooled map (h, w even) written from the accumulators
 *                             (max_pooling2d fused, simple_fcn.py:41-63); y == NULL or pooled == NULL skips that output
 *   xv_conv2d_f32_scalar      round 4's kernel (fp32 FMAs on the vector ALU): the A/B baseline of the bench record
 *   xv_maxpool2x2_f32         2x2 / stride 2
 *   xv_upsample2x_f32         y = relu(bilinear_x2(x)) [+ residual]            (upscore_conv5 + add_score)
 *   xv_score_lowres_f32       S[n][i+1][j+1][k] = fused[n][i][j][:] . Ws[:][k] into the zero-bordered [N][h+2][w+2][CP]
 *                             buffer (CP = num_classes rounded up to 4) that xv_decoder_head_from_scores reads
 *   xv_decoder_head_from_scores  x8 bilinear interpolation of S + bias -> score / prob / label (the second half of
 *                             xv_decoder_head_fwd; also serves the bf16 path's S)                                       */
int xv_conv2d_f32(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k, int cout,
                  int relu, float* y, void* stream);
int xv_conv2d_f32_pool(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k, int cout,
                       int relu, float* y, float* pooled, void* stream);
int xv_conv2d_f32_scalar(const float* x, int n, int h, int w, int cin, const float* w_hwio, const float* bias, int k, int cout,
                         int relu, float* y, void* stream);
int xv_maxpool2x2_f32(const float* x, int n, int h, int w, int c, float* y, void* stream);
int xv_upsample2x_f32(const float* x, int n, int h, int w, int c, const float* residual, float* y, void* stream);
/* ... with the inference batch norm of the deconv between it and its relu: y = relu(bilinear_x2(x) * scale + shift) + residual */
int xv_upsample2x_affine_f32(const float* x, int n, int h, int w, int c, const float* scale, const float* shift,
                             const float* residual, float* y, void* stream);
/* The decoder head without the commutation, in float32 (a batch norm with a shift between the x8 deconv and its relu:
 * custom_layers.py:112-119 under simple_fcn.py:89-135): fused [n][h][w][u] -> score / prob [n][8h][8w][C] float32, label
 * [n][8h][8w] int64 (any of the three may be NULL).  scale / shift: the folded batch norm of `upscore` ([u]).            */
int xv_decoder_head_affine_f32(const float* fused, int n, int h, int w, int u, const float* scale, const float* shift,
                               const float* w_score, const float* b_score, int num_classes, float* score, float* prob,
                               int64_t* label, void* stream);
int xv_score_lowres_f32(const float* fused, int n, int h, int w, int u, const float* w_score, int num_classes, float* S,
                        void* stream);
int xv_decoder_head_from_scores(const float* S, const float* b_score, int n, int hi, int wi, int num_classes, float* score,
                                float* prob, int64_t* label, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* XVIEW_HIP_H */
