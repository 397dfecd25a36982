/* zs3hip.h -- C ABI of libzs3hip.so, the MI355X (gfx950) kernel library under the zs3_amd Python host.
 *
 * Conventions: plain pointers are DEVICE pointers unless said otherwise; tensors
 * are fp32, activations NHWC ("channels_last") with an explicit pixel stride in ELEMENTS; every call
 * is asynchronous on the HIP stream passed last (hipStream_t as void*); return value 0 = launched,
 * >0 = hipError_t, <0 = argument error.  No call allocates, frees or synchronises.
 *
 * The reference (valeoai/ZS3) has no FFI layer: its hot path is torch.nn modules.  Each entry point
 * names the reference call sites it replaces; the Python binding is zs3_amd/_lib.py (ctypes).
 */
#ifndef ZS3HIP_H
#define ZS3HIP_H
/* `io` (the argument before the stream of every call that reads or writes ACTIVATION tensors): their element type in HBM.
 * Bit 0 (ZS3_IO_IN16) = the call's activation inputs are bf16, bit 1 (ZS3_IO_OUT16) = its activation outputs are; 0 = all
 * fp32 (BASELINE configs[1-3]), 3 = all bf16 (the 2-byte mode of configs[4]).  The pointers stay declared `float*`; strides
 * count elements; all arithmetic is fp32 in registers either way, and statistics / parameters / weight gradients are always
 * fp32.  Convolutions: bit 0 = x, bit 1 = y together with res, the accumulate target and bn_y; bf16 x needs prec = 1.  Weight
 * gradients: bit 0 = dy, bit 1 = x.  Element-wise calls accept 0 or 3; zs3_affine_act accepts all four (1 and 2 are the casts). */
#define ZS3_IO_IN16 1
#define ZS3_IO_OUT16 2
#ifdef __cplusplus
extern "C" {
#endif

/* ---- operand preparation ------------------------------------------------------------------ */
/* Split a conv / linear weight [cout][taps][cin] fp32 into bf16 hi/lo halves, stored
 * [row][K/32][{hi,lo}][32] (one 128-byte line per row and 32-wide K chunk): f_pk rows = cout with
 * K = taps*cin_pad (forward operand), and when t_pk != NULL rows = cin with K = taps*cout_pad (dgrad
 * operand).  cin_pad, cout_pad: multiples of 32.  Byte size of each: rows*K*4.  Replaces nothing in the
 * reference (cuDNN consumed fp32 weights directly). */
int zs3_prep_weight(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad, int cout_pad,
                    void* stream);
/* Exact-fp32 operands for prec = 0 of zs3_conv_igemm (test mode: every product on v_mfma_f32_32x32x2_f32, the arithmetic of the
 * reference's own fp32 convolutions; register-staged kernels only, ~1/16 of the bf16 rate): the same two planes as plain fp32
 * [row][K] rows with the same zero padding and the same byte sizes (a {hi,lo} bf16 pair and a float are both 4 bytes).
 * Exists so that a parity gap can be attributed to bf16x3 arithmetic rather than structure: tests/test_gpu_model.py runs the
 * reference's default-init train-mode goldens through it. */
int zs3_prep_weight_f32(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad, int cout_pad,
                        void* stream);
/* Operands of a layer whose FORWARD launches run prec = 4 ("f16x3": fp16 hi/lo halves, three v_mfma_f32_32x32x16_f16 per
 * operand pair, 2^-22-class products -- the default arithmetic of the fp32-storage forward pass since round 4, because it puts
 * the default-initialised train step at 0.8x / 1.3x / 0.9x the reference's own fp32 error where bf16x3 sat at 22x / 31x / 5.6x):
 * f_pk holds fp16 hi/lo of 2^6 * w (the prec = 4 kernels scale their accumulators back: a typical weight's lo half would be
 * an fp16 subnormal otherwise; |w| < 1023) in the same [row][K/32][{hi,lo}][32] layout, t_pk (data-gradient operand, prec = 3)
 * bf16 hi/lo.
 * fp16 has the precision but not the range for back-propagated gradients, so data- and weight-gradient launches stay bf16x3. */
int zs3_prep_weight_f16fwd(const float* w, void* f_pk, void* t_pk, int cout, int taps, int cin, int cin_pad, int cout_pad,
                           void* stream);
/* the same for many weights in ONE launch (after an optimizer step): table[e] = {w, f_pk, t_pk, cout, taps, cin, cin_pad,
 * cout_pad} as 8 int64 (device memory; bit 32 of the `taps` word set = the entry's forward plane is fp16 hi/lo as written by
 * zs3_prep_weight_f16fwd), blockmap[b] = {entry, chunk} as 2 int32 with chunk in
 * [0, zs3_prep_chunks(cout_pad, taps, cin_pad)) (one tap x 32 output channels x <= 256 input channels each). */
int zs3_prep_chunks(int cout_pad, int taps, int cin_pad);
int zs3_prep_weight_multi(const long* table, const int* blockmap, int nblocks, void* stream);
/* NCHW 3-channel image -> [N][H][Wp][4] zero-padded NHWC4 (image at columns [left,left+W)).  Feeds
 * the 7x7/s2 stem (resnet.py:79) as a 7x1 conv over 32-float (8 pixel x 4 ch) windows. */
int zs3_nchw3_to_nhwc4(const float* img, float* out, int N, int H, int W, int Wp, int left, void* stream);

/* ---- implicit-GEMM convolution (forward and data-gradient) --------------------------------- */
/* y[m,co] (+)= act(scale[co]*conv(x,w)[m,co] + shift[co] + res[m,co]),  m = (n,ho,wo).
 * x: NHWC fp32, spatial N x H x W, pixel stride ldx; K axis per tap = cin_pad channels of which
 * cin_valid (multiple of 4) are read, the rest are zeros.  w_pk: operand from zs3_prep_weight (K = KH*KW*cin_pad).  stat_partial (optional): [mtiles][2][ncols] per-row-tile sums and
 * sums of squares of the raw conv output (BatchNorm batch statistics), mtiles = zs3_conv_igemm_mtiles.
 * dgrad=1: rows are input-gradient pixels (N x Ho x Wo = the conv's input extent), x is dy (N x H x W
 * = the conv's output extent), w_pk is the t_pk operand.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(leak).  prec: 3 = bf16x3 split (2^-16-class products, fp32 exponent range), 4 = f16x3 split
 * (2^-22-class products, fp16 exponent range: |x| <= 65504, full precision for |x| >= 1e-4 or so; w_pk from
 * zs3_prep_weight_f16fwd; store-only epilogues, i.e. forward launches), 1 = plain bf16, 0 = exact fp32 (test mode:
 * w_pk from zs3_prep_weight_f32, tile_cfg 1-4 / 11-14 only, -7 otherwise).
 * tile_cfg: 0 auto, 1 128x128, 2 128x64, 3 64x128, 4 64x64 (+10: two-deep register prefetch).  zero_page: >= 256 bytes of
 * device zeros (16-byte aligned) that masked loads are redirected to; stride must be a power of two.
 * Replaces nn.Conv2d fwd / convolution_backward(input) at resnet.py:16-28,79,125-131; aspp.py:11-19,
 * 86,97; decoder.py:12,16,20,26; and nn.Linear of gmmn.py:18,33. */
int zs3_conv_igemm(const float* x, const void* w_pk, float* y, const float* scale,
                   const float* shift, const float* res, float* stat_partial, int N, int H, int W, int Ho, int Wo,
                   int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride, int pad_h, int pad_w, int dil,
                   int ncols, int ldy, int ldr, int act, float leak, int accumulate, int dgrad, int prec,
                   int tile_cfg, const void* zero_page, int io, void* stream);
int zs3_conv_igemm_mtiles(int M, int ncols, int tile_cfg);
/* zs3_conv_igemm whose input is read through x' = max(x * in_scale[c] + in_shift[c], 0) by the producer waves of the strip-resident and
 * persistent pointwise kernels (tile_cfg 41 / 42 / 51 / 52; -7 for any other kernel): BatchNorm-apply + ReLU of the layer that produced
 * x in the consumer's operand path -- that layer's activation tensor is never stored (resnet.py:33-53: bn1 -> relu -> conv2,
 * bn2 -> relu -> conv3).  in_scale / in_shift: >= cin_valid floats, 16-byte aligned. */
int zs3_conv_igemm_in(const float* x, const void* w_pk, float* y, const float* scale, const float* shift, const float* res,
                      float* stat_partial, int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW,
                      int stride, int pad_h, int pad_w, int dil, int ncols, int ldy, int ldr, int act, float leak, int accumulate,
                      int dgrad, int prec, int tile_cfg, const void* zero_page, const float* in_scale, const float* in_shift,
                      int io, void* stream);
/* tile_cfg 31: wave-specialised 256x128 LDS-DMA kernel, one tile per workgroup. */
/* tile_cfg 41 / 42 (csrc/conv_halo.hip): strip-resident kernel for stride-1, same-size multi-tap (3x3, dilated 3x3)
 * convolutions and their data gradients, 256- / 192-row tiles.  The input strip of a tile (tile rows + the halo the taps
 * reach, one channel chunk) is loaded ONCE, split to bf16 hi/lo by two producer waves and kept in LDS while all taps read
 * shifted windows of it; weight tiles are register-staged by the same producer waves.  zs3_conv_halo_ok() > 0 when a launch with
 * these arguments can run on tile_cfg 41 / 42 (the value is the kernel instantiation's strip-passes-per-step template argument;
 * 0: zs3_conv_igemm returns -7 for them, use tile_cfg 31).
 * Replaces the 3x3 nn.Conv2d of resnet.py:18-26, aspp.py:11-19 (atrous branches), decoder.py:15-24. */
/* tile_cfg 141 / 142: the same kernel reading x STORED AS BF16 ([pixel][channel], ldx in bf16 elements, 8-channel granularity;
 * prec = 1 only): the producers copy instead of converting -- half the input bytes, no conversion VALU (kernel-level result of
 * round 3; the network's tensors are fp32). */
int zs3_conv_halo_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                     int pad_h, int pad_w, int dil, int dgrad, int prec, int tile_cfg);
/* Persistent pointwise convolution (csrc/conv_pw.hip, tile_cfg 51 = 256-row tiles, 52 = 128-row tiles) of the 1x1 stride-1
 * layers and their input gradients: min(tiles, CUs) resident workgroups walk the output tiles; producer waves split the
 * activations to bf16 hi/lo once and prefetch across tile boundaries, MFMA waves run the shared fused epilogue.
 * zs3_conv_pw_ok: 1 when zs3_conv_igemm(..., tile_cfg 51 / 52) can run the layer (callers fall back to tile_cfg 31).
 * zs3_conv_pw_set_wgs: resident workgroups per launch (default 256), returns the previous value.
 * Replaces F.conv2d of the 1x1 nn.Conv2d at resnet.py:33-53, aspp.py:86-88. */
int zs3_conv_pw_ok(int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride,
                   int pad_h, int pad_w, int tile_cfg);
int zs3_conv_pw_set_wgs(int wgs);
/* zs3_conv_halo_set_wgs: workgroups per launch of the strip-resident kernel (tile_cfg 41 / 42): 0 (default) = one per tile, n > 0 =
 * launches with more tiles run on n workgroups walking the tiles, all handed out at once -- a kernel of another stream then starts beside
 * the launch instead of waiting for its rounds (the GMMN step's frozen feature pass beside the generator's update chain).  Returns the
 * previous setting. */
int zs3_conv_halo_set_wgs(int wgs);
/* zs3_conv_igemm (no affine / activation) with the two backward-pass epilogue fusions of the residual network:
 * (1) bn_partial != NULL: the epilogue also produces the BatchNorm-backward sums of the layer the output gradient belongs
 *     to: bn_partial[mtiles][2][ncols] = (sum dz, sum dz*xhat) per row tile, with dz = stored value * ReLU mask and
 *     xhat = (bn_y - bn_mean) * bn_invstd.  Mask: mask_bits ([M][ncols/4] sign bytes of zs3_affine_act), else
 *     bn_y*mask_scale + mask_shift > 0, else none.  Saves the separate zs3_bn_bwd_stats pass (a full read of the gradient).
 * (2) res_mask_bits != NULL: `res` is added through a ReLU mask (sign bytes [M][ncols/4]): the skip gradient of a residual
 *     block, (block-output gradient) * mask, is taken from the block-output gradient itself (res may alias y) instead of
 *     a copy written by zs3_bn_act_bwd.
 * resnet.py:33-53 / aspp.py / decoder.py backward. */
int zs3_conv_igemm_bnstats(const float* x, const void* w_pk, float* y, const float* res, const unsigned char* res_mask_bits,
                           int N, int H, int W, int Ho, int Wo, int cin_pad, int cin_valid, int ldx, int KH, int KW,
                           int stride, int pad_h, int pad_w, int dil, int ncols, int ldy, int ldr, int accumulate, int dgrad,
                           int prec, int tile_cfg, const void* zero_page, const float* bn_y, int bn_ldy,
                           const float* bn_mean, const float* bn_invstd, const float* mask_scale, const float* mask_shift,
                           const unsigned char* mask_bits, float* bn_partial, int io, void* stream);

/* ---- weight gradient ------------------------------------------------------------------------- */
/* Strip-resident weight gradient (csrc/conv_wgrad_strip.hip) of the stride-1, same-size 3x3 convolutions (pad = dilation):
 * the reduction runs over zero-padded pixel positions, so every filter tap is a constant shift of one LDS-resident strip of x
 * (read from L2 once instead of nine times); producer waves split both operands to bf16 hi/lo once, the MFMA waves fetch
 * their position-major fragments with ds_read_b64_tr_b16.  zs3_conv_wgrad_strip_plan returns 1 when the layer is eligible
 * (and the split-K factor / workspace floats the launch needs), 0 otherwise (use zs3_conv_wgrad).  Arguments as zs3_conv_wgrad;
 * dw: [co_write][3][3][ci_write].  Replaces convolution_backward(weight) at resnet.py:18-26, aspp.py:11-19, decoder.py:15-24. */
/* x_scale / x_shift (both kernels below; nullable, 16-byte aligned, >= ci_read floats): the producer waves read x through
 * x' = max(x * x_scale[c] + x_shift[c], 0) -- the BatchNorm-apply + ReLU of the layer that produced x, whose activation tensor is then
 * never stored (resnet.py:33-53: bn1 -> relu -> conv2, bn2 -> relu -> conv3); pad positions and rows past M stay zero. */
int zs3_conv_wgrad_strip_plan(int N, int H, int W, int Ho, int Wo, int KH, int KW, int stride, int pad_h, int pad_w, int dil,
                              int co, int ci, int* splitk_out, long* workspace_floats);
int zs3_conv_wgrad_strip(const float* dy, const float* x, float* dw, float* workspace, int N, int H, int W, int dil,
                         int co_read, int co_write, int ci_read, int ci_write, int lddy, int ldx, int prec,
                         const void* zero_page, const float* x_scale, const float* x_shift, int io, void* stream);
/* Pointwise weight gradient (csrc/conv_wgrad_strip.hip) of the stride-1 1x1 convolutions: dw[co][ci] = sum over the M = N*H*W
 * positions of dy[p][co] * x[p][ci], with the strip kernel's division of labour (producer waves split both operands to bf16
 * hi/lo once per (64 or 128)^2 tile and K step, MFMA waves read position-major fragments with ds_read_b64_tr_b16).
 * zs3_conv_wgrad_pw_plan returns 1 when the layer is eligible (>= 64 channels on both sides) and the split-K factor /
 * workspace floats of the launch, 0 otherwise (use zs3_conv_wgrad).  Channel arguments as zs3_conv_wgrad.
 * Replaces convolution_backward(weight) of the 1x1 nn.Conv2d at resnet.py:33-53, aspp.py:11-19,86-88. */
int zs3_conv_wgrad_pw_plan(long M, int co, int ci, int* splitk_out, long* workspace_floats);
int zs3_conv_wgrad_pw(const float* dy, const float* x, float* dw, float* workspace, long M, int co_read, int co_write,
                      int ci_read, int ci_write, int lddy, int ldx, int prec, const void* zero_page, const float* x_scale,
                      const float* x_shift, int io, void* stream);
/* dw[co][kh][kw][ci] = sum_m dy[m][co] * x[gather(m,kh,kw)][ci]  (channels_last weight layout).
 * dy: [M][lddy] with co_read (multiple of 4) readable channels of which co_write rows are produced;
 * x likewise (ci_read / ci_write).  Split-K over pixels: call zs3_conv_wgrad_plan (same M = N*Ho*Wo, Wo, channel
 * counts and tap count as the launch; it also fixes the kernel choice) for the workspace size (floats), pass NULL
 * when it returns 0.  Replaces convolution_backward(weight) of the same
 * call sites as zs3_conv_igemm. */
int zs3_conv_wgrad_plan(int M, int Wo, int co, int ci, int taps, int* splitk_out, long* workspace_floats);
int zs3_conv_wgrad(const float* dy, const float* x, float* dw, float* workspace, int N, int H, int W, int Ho, int Wo,
                   int KH, int KW, int stride, int pad_h, int pad_w, int dil, int co_read, int co_write, int ci_read,
                   int ci_write, int lddy, int ldx, int prec, const void* zero_page, int io, void* stream);
/* Kernel choice of zs3_conv_wgrad / zs3_conv_wgrad_plan: 0 = the library's rules, 1 = the register-staged kernel for every layer
 * (required before prec = 0, which exists on that kernel only: -7 otherwise), 2 = the LDS-DMA kernel wherever the channel counts
 * allow.  Returns the previous setting; plans made under another setting are stale. */
int zs3_conv_wgrad_set_kernel(int kernel);

/* ---- BatchNorm / ReLU / residual (bn.hip) ---------------------------------------------------- */
/* Replaces native_batch_norm fwd/bwd, relu_, threshold_backward, residual add_ at resnet.py:33-53,
 * aspp.py:25-29,111-116, decoder.py:30-32,15-24.  Partial-sum buffers are [chunks][2][C] floats. */
int zs3_colstats_plan(int M, int C, int* chunks, int* rows_per_block);
int zs3_colstats(const float* x, int ldx, int M, int C, float* partial, int io, void* stream);
/* ReLU mask, first that is given: mask_bits (one byte per 4 channels, dense [M][C/4], written by zs3_affine_act),
   a_out (> 0), or recomputed as y*mask_scale + mask_shift > 0 (layers without residual).
   drop_p > 0: dA is the gradient of a dropout fused behind the activation (zs3_affine_act): its mask is applied first. */
int zs3_bn_bwd_stats(const float* dA, int ldd, const float* a_out, int lda, const float* y, int ldy, const float* mean,
                     const float* invstd, const float* mask_scale, const float* mask_shift,
                     const unsigned char* mask_bits, int M, int C, float* partial, float drop_p,
                     unsigned long long drop_seed, int io, void* stream);
/* num_batches_tracked (nullable): BatchNorm's int64 step counter, incremented by the kernel.
   count_dev (nullable): device-resident sample count that overrides `count` (cross-rank SyncBN: the count is
   all-reduced together with the sums and never visits the host) */
/* chunks == -1 (both finalize calls): `partial` is a SyncBN exchange buffer of zs3_bn_sync_pack -- fp64 [sum x C | sum of
   squares (or second backward sum) x C | count] after its all-reduce; pass count_dev = the buffer's element 2C. */
/* SyncBN (sync_batchnorm/batchnorm.py:60-67,101-122): this rank's per-channel fp64 totals of the [chunks][2][C] partial sums and
   its sample count, written as the 2C+1 doubles ONE all-reduce carries across ranks. */
int zs3_bn_sync_pack(const float* partial, int chunks, int C, double count, double* totals, void* stream);
/* the same for the BACKWARD sums (sum dz, sum dz * xhat), which are this rank's dbeta / dgamma as they stand: also written as fp32
   vectors (dgamma, dbeta: C floats each, either may be NULL) -- the per-rank finalize launch of the SyncBN backward folded into the
   pack (113 launches per step) */
int zs3_bn_sync_pack_bwd(const float* partial, int chunks, int C, double count, double* totals, float* dgamma, float* dbeta,
                         void* stream);
int zs3_bn_fwd_finalize(const float* partial, int chunks, int C, double count, const double* count_dev,
                        const float* gamma, const float* beta, float eps, float momentum, float* running_mean,
                        float* running_var, float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                        long* num_batches_tracked, int* range_flag, void* stream);
/* range_flag (optional, device int32, sticky): set to 1 when a channel's batch sums are not finite -- what an operand beyond fp16's
   range does to an f16x3 (prec 4) forward convolution; the running statistics are then left untouched.  zs3_sgd_multi skips its
   update while the flag it is given is up (zs3_amd.functional.check_forward_range lowers it and switches the forward to bf16x3). */
int zs3_bn_eval_affine(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                       float eps, int C, float* mean_out, float* invstd_out, float* scale_out, float* shift_out,
                       void* stream);
int zs3_bn_bwd_finalize(const float* partial, int chunks, int C, double count, const double* count_dev, float* dgamma,
                        float* dbeta, float* c1, float* c2, int use_batch_stats, void* stream);
/* out[m][c] = act(alpha*(x[m/div][c]*scale[c] + shift[c]) + res[m][c]) (+= if accumulate); act 0/1/2.
   mask_out (nullable): [M][C/4] bytes, bit k of byte q = "pre-activation value of channel 4q+k was > 0" -- the
   residual blocks' ReLU mask for the backward pass, 1/16 of the bytes of re-reading the output twice.
   drop_p > 0: nn.Dropout fused behind the activation (aspp.py:100, decoder.py:19,23): out = keep ? act(...)/(1-p) : 0 with
   the mask zs3_dropout draws for (drop_seed, element m*C + c) -- saves a read + write of the activation per dropout. */
int zs3_affine_act(const float* x, int ldx, const float* scale, const float* shift, float alpha, const float* res,
                   int ldr, float* out, int ldo, long M, int C, int div, int act, float leak, int accumulate,
                   unsigned char* mask_out, float drop_p, unsigned long long drop_seed, int io, void* stream);
/* dz = act'(a_out)*dA; dres (=|+=) dz; dy = gamma*invstd*(dz - c1 - xhat*c2)  (c1==NULL: dy = gamma*invstd*dz);
   drop_p > 0: dA is first passed through the backward of the fused dropout (same mask, recomputed) */
int zs3_bn_act_bwd(const float* dA, int ldd, const float* a_out, int lda, const float* y, int ldy, const float* mean,
                   const float* invstd, const float* gamma, const float* c1, const float* c2, const float* mask_scale,
                   const float* mask_shift, const unsigned char* mask_bits, float* dy, int ldo, float* dres, int ldr,
                   int dres_accumulate, long M, int C, int act, float leak, float drop_p, unsigned long long drop_seed,
                   int io, void* stream);
/* out[g][c] = scale * sum_{r<R} x[g*R + r][c]: AdaptiveAvgPool2d((1,1)) of aspp.py:85 and its broadcast backward */
int zs3_group_colsum(const float* x, int ldx, int G, int R, int C, float scale, float* out, int ldo, int io, void* stream);

/* ---- pooling / resize (pool_resize.hip) ------------------------------------------------------- */
/* nn.MaxPool2d(3,2,1) of resnet.py:82 (idx: one byte per output element, the winning tap) */
int zs3_maxpool_fwd(const float* x, int ldx, float* out, int ldo, void* idx, int N, int H, int W, int Ho, int Wo, int C,
                    int K, int stride, int pad, int io, void* stream);
int zs3_maxpool_bwd(const float* dy, int ldd, const void* idx, float* dx, int ldo, int N, int H, int W, int Ho, int Wo,
                    int C, int K, int stride, int pad, int io, void* stream);
/* F.interpolate(mode="bilinear", align_corners=True) of aspp.py:109, decoder.py:34-36, deeplab.py:44,55 */
int zs3_bilinear_fwd(const float* x, int ldx, float* out, int ldo, int N, int H, int W, int Ho, int Wo, int C,
                     int io, void* stream);
int zs3_bilinear_bwd(const float* dout, int ldd, float* dx, int ldo, int N, int H, int W, int Ho, int Wo, int C,
                     int accumulate, int io, void* stream);

/* Validation without shipping logits to the host (train_pascal.py:130-134, Evaluator._generate_matrix metrics.py:73-79):
 * conf[gt*C + pred] += 1 for every target pixel with 0 <= gt < C, pred = first argmax over the C channels of x
 * [N,H,W,C] bilinearly resized (align_corners=True, same arithmetic as zs3_bilinear_fwd) to Ho x Wo -- pass the
 * low-resolution logits and the full-resolution label map and the upsampled logits never exist.  target: float32 or
 * int64 [N,Ho,Wo]; conf: C*C int64 counters, accumulated (zero them first). */
int zs3_argmax_confusion(const float* x, int ldx, int N, int H, int W, int C, const void* target, int target_is_i64,
                         int Ho, int Wo, void* conf, void* stream);

/* ---- losses (loss.hip) ------------------------------------------------------------------------ */
/* SegmentationLosses.CrossEntropyLoss (zs3/utils/loss.py:31-46): logits [P][ld] (P = B*H*W pixels, C classes),
 * target float32 or int64 [P]; loss_ws (3 floats) = {loss, sum of weights, sum of w*nll}; partial_ws: zs3_ce_ws_doubles()
 * doubles.  batch = B for batch_average, 0 for none.  gout: device scalar (upstream gradient). */
int zs3_ce_ws_doubles(void);
int zs3_ce_fwd(const float* logits, int ld, const void* target, int target_is_i64, const float* weight, long P, int C,
               int ignore_index, int batch, double* partial_ws, float* loss_ws, void* stream);
int zs3_ce_bwd(const float* logits, int ld, const void* target, int target_is_i64, const float* weight, long P, int C,
               int ignore_index, int batch, const float* loss_ws, const float* gout, float* dlogits, int ldo,
               void* stream);
/* GMMNLoss.moment_loss (zs3/utils/loss.py:92-115) for M == N samples of dimension D (even).  sigma: HOST array.
 * G: [2N][2N] floats kept for backward; tile_ws: 2*ceil(2N/32)^2 doubles; loss: device scalar. */
int zs3_mmd_fwd(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* sigma, int nsig,
                float* G, double* tile_ws, float* loss, void* stream);
int zs3_mmd_bwd(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* G, const float* loss,
                const float* gout, float* dgen, int ldo, void* stream);
/* zs3_mmd_bwd that takes the loss from zs3_mmd_fwd's tile_ws (zs3_mmd_fwd may then be called with loss = NULL); with
   loss_ring != NULL it also stores the loss value to loss_ring[slot_dev[0]] (slot_dev: device int64, < ring_len) */
int zs3_mmd_bwd_ws(const float* gen, int ldg, const float* real, int ldr, int N, int D, const float* G,
                   const double* tile_ws, const float* gout, float* dgen, int ldo, float* loss_ring, const void* slot_dev,
                   int ring_len, void* stream);
/* end of one generator update: loss_ring[slot_dev[0]++] = MMD loss from tile_ws; step_dev[0] += 1; seed_dev[0] += seed_inc */
int zs3_gmmn_update_epilogue(const double* tile_ws, int N, float* loss_ring, void* slot_dev, int ring_len, void* step_dev,
                             void* seed_dev, long seed_inc, void* stream);

/* ---- the GMMN generator update as latency-shaped kernels (gmmn.hip; train_pascal_GMMN.py:209-242, gmmn.py:17-22) ---------
 * Row-GEMMs on 32x16 output tiles whose workgroups load their whole reduction extent in one burst (<= 20 chunks of 32), bf16x3
 * on v_mfma_f32_16x16x32_bf16; w_pk / wt_pk are the forward / transposed operands of zs3_prep_weight (kchunks = K_pad / 32).
 * fwd1: x[r] = [emb[pix[r]][0:Ca] | U[0,1)^Cb keyed on key[r] | 0] (stored to x_out, ld ldx; pix == NULL: row r itself);
 *       h = LeakyReLU(x W1^T + b1);
 *       hd = Dropout(h; p_drop, mask keyed on key[r]) -- replaces zs3_gather_cat_noise + nn.Linear + LeakyReLU + nn.Dropout.
 * fwd2: gen = hd W2^T + b2; also real_out[r] = real[gidx[r]] (the MMD's real samples; pass NULL to skip).
 * dgrad: dpre = LeakyReLU'(h) * Dropout'(dgen W2) -- backward of fwd2, the dropout and the activation in one launch.
 * wgrad: dw2 = dy2^T x2 ([co2][ci2]), db2 = column sums of dy2, and the same for layer 1, reduction over R <= 128 rows;
 *        one launch for both layers.  All leading dimensions and channel counts: multiples of 4. */
/* table-driven update (no host argument per replay): row u = upd_dev[0] of `table` ([S sample indices | order offset | pixel
 * base], int64, row stride ld_table) selects the (image, class); writes x[j] = [emb[base + order[offset + ridx[j]]] | noise
 * keyed on ridx[j]], pix_global[j] (row of the real features) and key[j] = ridx[j].  Replaces zs3_sample_rows + the
 * per-update host-to-device copy of the sample indices + zs3_gather_cat_noise.  adam_bc (optional): receives Adam's bias
 * corrections {1 - b1^t, sqrt(1 - b2^t)}, t = adam_step_dev[0] + 1, of this update for zs3_gmmn_mlp_wgrad_adam. */
int zs3_gmmn_prep(const long* table, int ld_table, const void* upd_dev, const long* order, const float* emb, int ld_emb,
                  int Ca, int Cb, float* x, int ldx, long* pix_global, long* key, int S, unsigned long long seed,
                  const void* seed_dev, const void* adam_step_dev, float b1, float b2, float* adam_bc, void* stream);
int zs3_gmmn_mlp_fwd1(const float* emb, int ld_emb, const long* pix, const long* key, int Ca, int Cb, const void* w_pk,
                      int kchunks, const float* bias, float* x_out, int ldx, float* h, float* hd, int ldo, int M, int N,
                      float leak, float p_drop, unsigned long long seed_noise, unsigned long long seed_drop,
                      const void* seed_dev, void* stream);
/* zs3_gmmn_prep + zs3_gmmn_mlp_fwd1 in ONE launch (round 4: six launches per generator update instead of seven): every workgroup of
 * the first GEMM reads its sample indices from the update's table row itself (two dependent index loads ahead of its operand
 * burst instead of a kernel boundary), the workgroups of the first column tile publish pix_global / key, workgroup 0 the Adam
 * bias corrections.  Same draws, same results as the two-launch form (tests/test_gpu_gmmn_kernels.py). */
int zs3_gmmn_mlp_fwd1_table(const long* table, int ld_table, const void* upd_dev, const long* order, const float* emb, int ld_emb,
                            int Ca, int Cb, const void* w_pk, int kchunks, const float* bias, float* x_out, int ldx, float* h,
                            float* hd, int ldo, int S, int N, float leak, float p_drop, unsigned long long seed_noise,
                            unsigned long long seed_drop, const void* seed_dev, long* pix_global, long* key,
                            const void* adam_step_dev, float b1, float b2, float* adam_bc, void* stream);
int zs3_gmmn_mlp_fwd2(const float* hd, int lda, const void* w_pk, int kchunks, const float* bias, float* gen, int ldo, int M,
                      int N, int K, const float* real, int ld_real, const long* gidx, float* real_out, void* stream);
int zs3_gmmn_mlp_dgrad(const float* dgen, int lda, const void* wt_pk, int kchunks, const float* h, int ldh, const long* key,
                       float* dpre, int ldo, int M, int N, int K, float leak, float p_drop, unsigned long long seed_drop,
                       const void* seed_dev, void* stream);
int zs3_gmmn_mlp_wgrad(const float* dy2, int ldy2, const float* x2, int ldx2, int co2, int ci2, float* dw2, float* db2,
                       const float* dy1, int ldy1, const float* x1, int ldx1, int co1, int ci1, float* dw1, float* db1, int R,
                       void* stream);
/* wgrad + torch.optim.Adam (train_pascal_GMMN.py:65-67,240) + end-of-update bookkeeping in one launch: every 64x64 tile of a
 * weight gradient is consumed where it is produced -- Adam moments and weight updated, the weight's bf16 hi/lo operand planes
 * rewritten -- and the last workgroup to finish advances the device-resident counters of the captured update: slot_dev += 1,
 * step_dev += 1, seed_dev += seed_inc.  state2 / state1: 8 device pointers per layer (HOST arrays): {weight, exp_avg,
 * exp_avg_sq, bias, bias exp_avg, bias exp_avg_sq, f_pk, t_pk}; done_dev: a zeroed uint32 arrival counter (left zero);
 * adam_bc: the two bias corrections written by zs3_gmmn_prep for this update, or NULL (then every thread computes them). */
int zs3_gmmn_mlp_wgrad_adam(const float* dy2, int ldy2, const float* x2, int ldx2, int co2, int ci2, const float* dy1,
                            int ldy1, const float* x1, int ldx1, int co1, int ci1, int R, const void* const* state2,
                            const void* const* state1, int cin_pad2, int cout_pad2, int cin_pad1, int cout_pad1, float lr,
                            float b1, float b2, float eps, float wd, void* slot_dev, void* step_dev, void* seed_dev,
                            long seed_inc, void* done_dev, const float* adam_bc, void* stream);

/* ---- GMMN step helpers and optimisers (misc.hip) ---------------------------------------------- */
/* nn.Dropout (aspp.py:100, decoder.py:19,23, gmmn.py:20): y = keep ? x/(1-p) : 0 with a counter-based mask
 * that is a pure function of (seed, element index); the backward is the same call on dy.  row_idx (optional,
 * int64[M]): row m of x is row row_idx[m] of the tensor the mask was drawn for (sampled-row backward). */
/* seed_dev (optional): device uint64 added to `seed`, so a launch captured in a hipGraph draws a fresh mask per replay */
int zs3_dropout(const float* x, int ldx, float* y, int ldy, long M, int C, float p, unsigned long long seed,
                const long* row_idx, const void* seed_dev, int io, void* stream);
int zs3_uniform(float* out, long n, unsigned long long seed, const void* seed_dev, void* stream);
/* counter[0] += v (device int64 / uint64): advances the stream position / step count between graph replays */
int zs3_counter_add(void* counter, long v, void* stream);
/* ---- fused pieces of the GMMN generator update (train_pascal_GMMN.py:205-236, replayed per image and class) ---- */
/* zs3_uniform + zs3_gather_cat in one launch: out[r] = [a[idx[r]][0:Ca] | U[0,1)^Cb | 0...].  The noise of row r is keyed
   on noise_key[r] (r itself when NULL): with noise_key = the sampled within-class pixel index, duplicate samples share
   their noise row exactly like z[random_idx] of train_pascal_GMMN.py:216,229-236 */
int zs3_gather_cat_noise(const float* a, int lda, const long* idx, int Ca, int Cb, float* out, int ldo, long n,
                         unsigned long long seed, const void* seed_dev, const long* noise_key, void* stream);
/* out = dropout_backward(dy; p, seed, row_idx) * leaky_relu'(h; leak): Dropout + LeakyReLU backward of gmmn.py:19-22 */
int zs3_dropout_act_bwd(const float* dy, int ldd, const float* h, int ldh, float* out, int ldo, long M, int C, float p,
                        unsigned long long seed, const long* row_idx, const void* seed_dev, float leak, void* stream);
/* out = srcs[0] + ... + srcs[n-1] in that order, 2 <= n <= 8 dense fp32 arrays of `count` elements: the gradient of a tensor
 * with several consumers (aspp.py:104-108: x feeds five branches; deeplab.py:41-42: layer1's output feeds layer2 and the
 * decoder) in one pass -- autograd would add pairwise.  srcs: HOST array of device pointers; out may alias srcs[0]. */
int zs3_sum_n(const void* const* srcs, int n, float* out, long count, int io, void* stream);
/* The two fills / copies the step used to leave to the tensor library, as entry points of this one so that a recorded plan
 * (below) carries them: zs3_fill_zero = `bytes` zero bytes at dst (the zero-initialised channel pad of a gradient buffer);
 * zs3_pad_rows = dst[m][0:C] = src[m][0:C], dst[m][C:ldd] = 0 for M rows (io 0: 4-byte elements, 3: 2-byte) -- the 21 -> 24
 * channel pad of the class-score gradient in front of pred_conv's data / weight gradient (decoder.py:26). */
int zs3_fill_zero(void* dst, long bytes, void* stream);
/* rows x [g][c] floats <-> rows x [G][C] floats (g <= G, c <= C), zero padded: unpack = 0 pads (dst = the [G][C] rows), unpack = 1
 * selects the [g][c] corner back (src = the [G][C] rows).  The 7x7x3 stem weight as the 7 x (8 pixels x 4 channels) operand of
 * the stem's window convolution (resnet.py:79 -> zs3_nchw3_to_nhwc4) and its gradient on the way back -- F.pad and its backward
 * until round 5, the last tensor-library kernels inside the supervised step. */
int zs3_repack_pad(const float* src, long rows, int g, int c, float* dst, int G, int C, int unpack, void* stream);
int zs3_pad_rows(const void* src, int lds, int C, void* dst, int ldd, long M, int io, void* stream);
/* out[c] = sum_m x[m][c] (rows in order): bias gradients of the generator's Linear layers */
int zs3_colsum(const float* x, int ldx, int M, int C, float* out, void* stream);
/* torch.optim.Adam for several tensors in one launch, device-resident step count; table[e] = {p, g, exp_avg, exp_avg_sq,
 * n, f_pk, t_pk, cout, cin, cin_pad, cout_pad} (11 int64): entries with f_pk != 0 are Linear weights whose bf16 hi/lo
 * operands (zs3_prep_weight layout, taps = 1) are rewritten in the same pass; blockmap[b] = {entry, chunk of
 * zs3_adam_chunk() elements}. */
int zs3_adam_chunk(void);
int zs3_adam_multi(const long* table, const int* blockmap, int nblocks, float lr, float b1, float b2, float eps, float wd,
                   const void* step_dev, void* stream);
int zs3_counter_add2(void* c0, long v0, void* c1, long v1, void* stream);
/* pix_local[j] = order[ridx[j]], pix_global[j] = pix_local[j] + base (the sampled rows of train_pascal_GMMN.py:229-231) */
int zs3_sample_rows(const long* order, const long* ridx, long base, long* pix_local, long* pix_global, int s,
                    void* stream);
/* The label bookkeeping at the head of a GMMN step in one launch (train_pascal_GMMN.py:175-180 nearest resize of the label
 * maps, :183 unique classes, :226 per-class pixel lists): target [B][H][W] float32 or int64 -> tgt_l [B][ho*wo] int64 (values
 * outside 0..255 become 255), tgt_cls (255 -> 0, the dataloader's embedding class, datasets/base.py:47-48), hist [B][256]
 * int64 class histogram, order [B][ho*wo] int64 = the pixels of every image grouped by class in ascending pixel order (a stable
 * argsort of tgt_l). */
int zs3_label_order(const void* target, int target_is_i64, int B, int H, int W, int ho, int wo, long* tgt_l, long* tgt_cls,
                    long* hist, long* order, void* stream);
/* F.interpolate(mode="nearest") of one [C][H][W] image into pixel rows [ho*wo][ldo] (train_pascal_GMMN.py:175-195) */
int zs3_nearest_rows(const float* src, int C, int H, int W, int ho, int wo, float* rows, int ldo, void* stream);
/* out[r] = [a[idx[r]][0:Ca] | b[r][0:Cb] | 0...]: torch.cat((embd, noise), 1) of gmmn.py:44 fused with the class mask */
int zs3_gather_cat(const float* a, int lda, const long* idx, int Ca, const float* b, int ldb, int Cb, float* out,
                   int ldo, long n, void* stream);
int zs3_gather_rows(const float* src, int lds, const long* idx, float* out, int ldo, long n, int C, void* stream);
int zs3_scatter_rows(const float* src, int lds, const long* idx, float* out, int ldo, long n, int C, void* stream);
int zs3_index_add_rows(const float* src, int lds, const long* idx, float* out, int ldo, int n, int C, void* stream);
/* torch.optim.SGD (train_pascal.py:55-60) and torch.optim.Adam (train_pascal_GMMN.py:65-67) update rules */
int zs3_sgd_step(float* p, const float* g, float* buf, long n, float lr, float momentum, float wd, int nesterov,
                 int first, void* stream);
/* One launch for a whole parameter set: table = int64[E][6] {p, g, momentum_buf, n, lr | wd<<32 (float bits), first},
 * blockmap = int32[nblocks][2] {entry, chunk}; every block updates zs3_sgd_chunk() consecutive elements.
 * skip_flag (optional, device int32): when it reads non-zero the launch leaves parameters and existing momentum buffers alone
 * (first-step buffers are zeroed) -- the forward range guard's flag (zs3_bn_fwd_finalize), so that a step whose forward left
 * fp16's range never reaches the weights. */
int zs3_sgd_chunk(void);
int zs3_sgd_multi(const void* table, const void* blockmap, int nblocks, float momentum, int nesterov, const int* skip_flag,
                  void* stream);
/* zs3_sgd_multi with the groups' hyper-parameters as LAUNCH ARGUMENTS: table[e][4] = the entry's parameter-group index,
 * group_lr_wd = HOST array [ngroups][2] {lr, wd}, 1 <= ngroups <= zs3_sgd_max_groups() (-3 otherwise: use zs3_sgd_multi).  The
 * table then holds nothing that a learning-rate schedule changes (lr_scheduler.py:46-76 sets a new lr before every step,
 * base_trainer.py:15): no per-step table upload for the schedule, and a recorded plan (below) follows the schedule by patching
 * this one argument (zs3_plan_patch). */
int zs3_sgd_max_groups(void);
int zs3_sgd_multi_g(const void* table, const void* blockmap, int nblocks, float momentum, int nesterov, const int* skip_flag,
                    const float* group_lr_wd, int ngroups, void* stream);
/* step_dev (optional): device int64 holding the number of steps taken so far; overrides `step` (= step_dev[0] + 1) */
int zs3_adam_step(float* p, const float* g, float* m, float* v, long n, float lr, float b1, float b2, float eps,
                  float wd, int step, const void* step_dev, void* stream);


/* ---- GCN-context cluster graph (train_context_GMMN_GCNcontext.py:33-102, construct_adj_mat) ------------------- */
/* seg: [H][W] int32 class map (H*W <= zs3_cluster_graph_max_pixels()).  cmap[p] = cluster id (8-connected components
 * of equal class, numbered in raster order of their first pixel); seed[c] = that first pixel's flat index, labels[c] its
 * class; ncluster[0] = number of clusters; adj: [cap][cap] floats, zeroed by the caller, adj[c1][c2] = 1 when clusters of
 * different class touch in the 8-neighbourhood (clusters >= cap are counted but not recorded).  One launch, no host sync. */
int zs3_cluster_graph_max_pixels(void);
int zs3_cluster_graph(const int* seg, int H, int W, int* cmap, int* seed, int* labels, int* ncluster, float* adj, int cap,
                      void* stream);
/* B label maps in one launch (one workgroup each): seg/cmap [B][H][W], seed/labels [B][cap], ncluster [B], adj [B][cap][cap].
 * The GCN-context step (train_context_GMMN_GCNcontext.py:307-322 inside the per-image loop) builds the graphs of the whole
 * batch up front and reads the B cluster counts back once. */
int zs3_cluster_graph_batch(const int* seg, int B, int H, int W, int* cmap, int* seed, int* labels, int* ncluster, float* adj,
                            int cap, void* stream);

/* ---- collectives: RCCL under the C ABI (round 6) -------------------------------------------------------------------------- */
/* The reference's multi-GPU exchanges (nn.DataParallel's replicate / gather and the vendored SyncBN's
 * master-slave reduction, sync_batchnorm/batchnorm.py:101-122, comm.py:98-126) as one-process-per-GPU collectives issued BY THIS
 * LIBRARY on the stream the caller names: the SyncBN sums on the compute stream itself (pack -> all-reduce -> finalize in stream
 * order: no hand-over to a framework's collective stream, 208 times per step), the gradient buckets on the weight-gradient side
 * stream.  RCCL is bound at run time: zs3_comm_load(path of the librccl the process uses; NULL / "" = "librccl.so.1").
 * Communicator set-up: rank 0 calls zs3_comm_unique_id (zs3_comm_unique_id_bytes() bytes, HOST memory), the bytes reach the other
 * ranks by any means (zs3_amd/parallel.py: torch.distributed's store), every rank calls zs3_comm_create(id, nranks, rank) -> handle
 * (0 = failure).  zs3_allreduce / zs3_broadcast work IN PLACE on `count` elements of dtype 0 fp32 / 1 fp64 / 2 int32 / 3 int64; op 0 =
 * SUM, 1 = MAX.  Return codes: 0, < 0 argument / state error, 1000 + ncclResult_t for RCCL's own errors (message on stderr).
 * All ranks must issue the collectives of one communicator in the same order; use one communicator per stream that issues them.
 * Being entry points with a trailing stream, collectives are recorded and replayed by launch plans like kernel launches. */
int zs3_comm_load(const char* librccl_path);
int zs3_comm_unique_id_bytes(void);
int zs3_comm_unique_id(void* id_out);
long zs3_comm_create(const void* unique_id, int nranks, int rank);
int zs3_comm_destroy(long comm);
int zs3_comm_abort(long comm);     /* ncclCommAbort: ends the communicator's collectives in flight (peers that never arrive), frees it */
int zs3_comm_ranks(long comm);
/* (a communicator of ONE rank: both collectives return without enqueuing anything -- in place they are the identity) */
int zs3_allreduce(long comm, void* buf, long count, int dtype, int op, void* stream);
int zs3_broadcast(long comm, void* buf, long count, int dtype, int root, void* stream);
/* zs3_bn_sync_pack + the SUM all-reduce of its 2C + 1 doubles in one call (batchnorm.py:101-122: the replicas' sum / sum of
 * squares / element count reduced on the master and broadcast back); zs3_bn_*_finalize(chunks = -1) read `totals` afterwards. */
int zs3_bn_sync_exchange(long comm, const float* partial, int chunks, int C, double count, double* totals, void* stream);
int zs3_bn_sync_exchange_bwd(long comm, const float* partial, int chunks, int C, double count, double* totals, float* dgamma,
                             float* dbeta, void* stream);      /* zs3_bn_sync_pack_bwd + the all-reduce of the totals */
/* global normalisation of the CE behind the all-reduce of loss_ws[1..2] (utils/loss.py: the loss of the gathered batch that
 * nn.DataParallel hands the reference's criterion, loss.py:33-46): loss_ws[0] = loss_ws[2] / loss_ws[1] / global_batch
 * (global_batch <= 0: no batch averaging). */
int zs3_ce_global_finish(float* loss_ws, int global_batch, void* stream);

/* ---- recorded launch plans: the step loop under the C ABI (round 6) ------------------------------------------------------- */
/* The loop body of base_trainer.py:16-20 (zero_grad / forward / loss / backward / step) is ~1000 calls of the entry points above
 * per iteration, identical from iteration to iteration up to a few scalars.  A PLAN records them once and replays them from C:
 *
 *   plan = zs3_plan_create();  zs3_plan_record_begin(plan);  <one eager step>;  nops = zs3_plan_record_end(plan);
 *   per iteration:  zs3_plan_patch / zs3_plan_replace_u64 (learning rate, dropout seeds);  zs3_plan_replay(plan, 0, -1);
 *
 * While a plan records, every entry point whose last parameter is `void* stream` appends {its arguments} to the plan and then
 * executes as usual (the recording wrappers are generated from this header: zs3_amd/build.py, csrc/gen/plan_wrappers.hip); HOST
 * arrays among the arguments (zs3_sum_n's srcs, zs3_mmd_fwd's sigma, zs3_sgd_multi_g's group_lr_wd) are copied into the plan.
 * zs3_plan_replay(plan, first, count) issues ops [first, first + count) (count < 0: to the end) again: same entry points, same
 * arguments, same streams -- plain launches on the real streams, not a hipGraph, so side-stream launches overlap exactly as
 * in the eager step.  Returns 0, or the first failing op's return code (its index: zs3_plan_failed_op).  The CALLER keeps every
 * device buffer the recorded step touched alive and at its address (zs3_amd/plan.py records under a private allocator pool).
 * One plan records at a time, process-wide (launches come from the caller's thread and from autograd's device thread).
 * Calls return 0 / a count on success, < 0 on misuse (-1 bad handle / index, -2 recording state, -3 size, -4 kind). */
long zs3_plan_create(void);
int zs3_plan_destroy(long plan);
int zs3_plan_record_begin(long plan);
int zs3_plan_record_end(long plan);                 /* -> number of recorded ops */
int zs3_plan_size(long plan);
int zs3_plan_truncate(long plan, int nops);         /* forget the ops behind the first nops */
int zs3_plan_replay(long plan, int first, int count);
int zs3_plan_failed_op(long plan);
int zs3_plan_op_name(long plan, int op, char* buf, int cap);          /* entry-point name of op (debugging, tests) */
int zs3_plan_find_op(long plan, const char* name, int nth);           /* index of the nth op of that entry point, -1 if none */
/* every recorded `unsigned long long` argument equal to old_value (a dropout seed: the forward launch and its backward launches
 * carry the same one) becomes new_value; -> number of arguments rewritten */
int zs3_plan_replace_u64(long plan, unsigned long long old_value, unsigned long long new_value);
/* Rebinding a buffer of the recorded step (the input batch; the optimizer's table): zs3_plan_find_ptr lists where a device pointer
 * occurs -- up to cap (op, arg) pairs into where[2k], where[2k + 1], -> the total -- and zs3_plan_set_ptr rewrites ONE argument.
 * Ask right after the recording, while the buffer is alive and its address means nothing else.  zs3_plan_replace_ptr rewrites every
 * pointer argument equal to old_ptr: only safe for addresses outside the recording's allocator pool -- inside it one address serves
 * several tensors in the course of a step. */
int zs3_plan_find_ptr(long plan, const void* ptr, int* where, int cap);
int zs3_plan_set_ptr(long plan, int op, int arg, const void* ptr);
int zs3_plan_replace_ptr(long plan, const void* old_ptr, const void* new_ptr);
/* overwrite scalar / host-array argument `arg` (0-based position in the entry point's parameter list) of op with `bytes` bytes */
int zs3_plan_patch(long plan, int op, int arg, const void* data, int bytes);
/* HIP-event timing of chosen ops inside ONE replay: zs3_plan_time_ops arms a set of n ascending op indices (-> the set's id, or
 * < 0; n = 0 disarms); the NEXT zs3_plan_replay records an event pair around each of them on the op's own stream and disarms; after
 * the caller has synchronised, zs3_plan_timed_ms(plan, set, out_ms, cap) -> the n durations in milliseconds.  Sets stay readable until
 * the plan is destroyed.  (bench.py: the dominant kernel's launches timed inside replayed steps.) */
int zs3_plan_time_ops(long plan, const int* ops, int n);
int zs3_plan_timed_ms(long plan, int set, float* out_ms, int cap);
/* read argument `arg` of op back (-> its size in bytes; cap = room at out): tests and debugging */
int zs3_plan_get_arg(long plan, int op, int arg, void* out, int cap);
/* kind of argument `arg` of op as a character code: 'p' pointer, 's' stream, 'i' int, 'l' long, 'f' float, 'd' double, 'u' unsigned long
 * long, 'h' host array copied into the plan; 0 behind the last argument */
int zs3_plan_arg_kind(long plan, int op, int arg);
/* waiter waits for everything queued on producer so far (event record + stream wait, one reusable event per waiting stream):
 * the cross-stream dependencies of the step (weight-gradient side streams, functional.py) as a recordable call. */
int zs3_stream_wait(void* waiter, void* producer);

#ifdef __cplusplus
}
#endif
#endif
