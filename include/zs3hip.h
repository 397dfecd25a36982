/* zs3hip.h -- C ABI of libzs3hip.so, the MI355X (gfx950) kernel library under the zs3_amd Python host.
 *
 * Conventions: plain pointers are DEVICE pointers unless said otherwise; tensors
 * are fp32, activations NHWC ("channels_last") with an explicit pixel stride in floats; every call
 * is asynchronous on the HIP stream passed last (hipStream_t as void*); return value 0 = launched,
 * >0 = hipError_t, <0 = argument error.  No call allocates, frees or synchronises.
 *
 * The reference (valeoai/ZS3) has no FFI layer: its hot path is torch.nn modules.  Each entry point
 * names the reference call sites it replaces; the Python binding is zs3_amd/_lib.py (ctypes).
 */
#ifndef ZS3HIP_H
#define ZS3HIP_H
#ifdef __cplusplus
extern "C" {
#endif

/* ---- operand preparation ------------------------------------------------------------------ */
/* Split a conv / linear weight [cout][taps][cin] fp32 into bf16 hi/lo planes [cout][taps][cin_pad]
 * (f_*) and, when t_hi != NULL, the dgrad planes [cin][taps][cout_pad] (t_*).  cin_pad, cout_pad:
 * multiples of 32.  Replaces nothing in the reference (cuDNN consumed fp32 weights directly). */
int zs3_prep_weight(const float* w, void* f_hi, void* f_lo, void* t_hi, void* t_lo, int cout, int taps, int cin,
                    int cin_pad, int cout_pad, void* stream);
/* NCHW 3-channel image -> [N][H][Wp][4] zero-padded NHWC4 (image at columns [left,left+W)).  Feeds
 * the 7x7/s2 stem (resnet.py:79) as a 7x1 conv over 32-float (8 pixel x 4 ch) windows. */
int zs3_nchw3_to_nhwc4(const float* img, float* out, int N, int H, int W, int Wp, int left, void* stream);

/* ---- implicit-GEMM convolution (forward and data-gradient) --------------------------------- */
/* y[m,co] (+)= act(scale[co]*conv(x,w)[m,co] + shift[co] + res[m,co]),  m = (n,ho,wo).
 * x: NHWC fp32, spatial N x H x W, pixel stride ldx; K axis per tap = cin_pad channels of which
 * cin_valid (multiple of 8) are read, the rest are zeros.  w_hi/w_lo: planes from zs3_prep_weight
 * with row stride KH*KW*cin_pad.  stat_partial (optional): [mtiles][2][ncols] per-row-tile sums and
 * sums of squares of the raw conv output (BatchNorm batch statistics), mtiles = zs3_conv_igemm_mtiles.
 * dgrad=1: rows are input-gradient pixels (N x Ho x Wo = the conv's input extent), x is dy (N x H x W
 * = the conv's output extent), w planes are the transposed t_* planes.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(leak).  prec: 3 = bf16x3 split (fp32-class), 1 = plain bf16.
 * tile_cfg: 0 auto, 1 128x128, 2 128x64, 3 64x128, 4 64x64.
 * Replaces nn.Conv2d fwd / convolution_backward(input) at resnet.py:16-28,79,125-131; aspp.py:11-19,
 * 86,97; decoder.py:12,16,20,26; and nn.Linear of gmmn.py:18,33. */
int zs3_conv_igemm(const float* x, const void* w_hi, const void* w_lo, float* y, const float* scale,
                   const float* shift, const float* res, float* stat_partial, int N, int H, int W, int Ho, int Wo,
                   int cin_pad, int cin_valid, int ldx, int KH, int KW, int stride, int pad_h, int pad_w, int dil,
                   int ncols, int ldy, int ldr, int act, float leak, int accumulate, int dgrad, int prec,
                   int tile_cfg, void* stream);
int zs3_conv_igemm_mtiles(int M, int ncols, int tile_cfg);

#ifdef __cplusplus
}
#endif
#endif
