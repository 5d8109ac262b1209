/* libtcct_hip.so — C-ABI of the MI355X (gfx950) kernels for the TCCT `stc_tt` training hot path.
 *
 * The reference (tyb311/TCCT, task1) has NO native/FFI layer: every op of the path is a stock ATen call made
 * from Python (nets/tcct.py, nets/reg.py, nets/fcs.py, nets/fcp.py, kite/losses/loss.py, kite/loopback.py).
 * Each entry below therefore cites the reference *call site(s)* whose ATen op family it replaces
 * (paths relative to /root/reference/task1).  INTEGRATION.md shows the ctypes binding a maintainer adds.
 *
 * Conventions
 *   - activations are NHWC ("channels-last") device buffers, dtype TCCT_F32 or TCCT_BF16; a token tensor
 *     [B,N,C] of the ViT branch is the same memory as the NHWC image [B,H,W,C] (N = H*W).
 *   - parameters, statistics, gradients of parameters and loss scalars are fp32 (BN sums fp64).
 *   - the caller owns every buffer (library never allocates or frees); all calls are asynchronous on
 *     `stream` (a hipStream_t passed as void*), never synchronise, and are hipGraph-capturable.
 *   - return 0 on success, <0 on error; tcct_last_error() gives a thread-local message.
 *   - no RNG inside: DropPath masks / Gumbel noise / jitter are inputs.
 */
#ifndef TCCT_HIP_H
#define TCCT_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

typedef void* tcct_stream_t;

enum { TCCT_F32 = 0, TCCT_BF16 = 1 };
enum { TCCT_ACT_NONE = 0, TCCT_ACT_LRELU = 1, TCCT_ACT_HSWISH = 2, TCCT_ACT_GELU = 3, TCCT_ACT_SIGMOID = 4,
       TCCT_ACT_ABS = 5 };

int tcct_version(void);
const char* tcct_last_error(void);
/* process-wide switch: 1 = accumulation outputs (dw, dbias, BN/LN sums) handed to the calls are already zero, so the
 * entry points skip their own hipMemsetAsync (the caller clears one pooled buffer per step instead); returns 0 */
int tcct_set_outputs_prezeroed(int on);

/* ---- layout: loader-side `img.to(device)` (kite/loop_seg.py:116) + 1->3 channel replicate / W pad ---------
 * img [N,Csrc,H,Wsrc] fp32 NCHW (Csrc 1 or 3) -> out [N,H,Wdst,4] NHWC (channel 3 and columns >= Wsrc zero). */
int tcct_image_to_nhwc4(const float* img, void* out, int N, int Csrc, int H, int Wsrc, int Wdst, int dtype,
                        tcct_stream_t stream);
/* GOALS B-scan / label-map preprocessing on uint8 HWC images (reference data/octnpy.py:82-85,91-112,119-130 and
 * data/octgen.py:9-24 do this with OpenCV + albumentations on the CPU): dst[n, dy0+y, dx0+x, c] = src[n, sy0 + ny(y), sx0 + nx(x), c] * mul / div
 * for (y, x) in the dh x dw destination window, `fill` elsewhere; ny/nx = cv2.INTER_NEAREST source index of a sh x sw -> dh x dw resize
 * (min(floor(d * (1.0 / (dn / sn))), sn - 1) in double precision), applied after the optional vertical / horizontal flip of the window.
 * Covers row crop, nearest resize, flips, label code changes (//30, *30) and pasting into the 800 x 1100 canvas in one gather. */
int tcct_u8_gather2d(const uint8_t* src, uint8_t* dst, int N, int SH, int SW, int C, int sy0, int sx0, int sh, int sw, int DH, int DW,
                     int dy0, int dx0, int dh, int dw, int flipy, int flipx, int mul, int div, int fill, tcct_stream_t stream);
/* labels: one-hot int64 [N,C,H,W] (kite/loop_seg.py:119) -> class index uint8 [N,H,W] */
int tcct_onehot_to_index(const int64_t* onehot, uint8_t* lab, int N, int C, int64_t HW, tcct_stream_t stream);
/* int64 [N,H,Wsrc] class labels -> uint8 [N,H,Wdst] (columns >= Wsrc get class 0) */
int tcct_labels_to_u8(const int64_t* lab, uint8_t* out, int N, int H, int Wsrc, int Wdst, tcct_stream_t stream);
/* NHWC [M,C] -> NCHW-contiguous fp32 copy and back (API-boundary materialisation only) */
int tcct_nhwc_to_nchw_f32(const void* x, float* y, int N, int64_t HW, int C, int dtype, tcct_stream_t stream);

/* ---- elementwise: nn.LeakyReLU / nn.Hardswish / nn.GELU / F.gelu / abs / sigmoid and residual adds
 * (nets/tcct.py:35,89,126,811,817,822,826,894,980; nets/reg.py:75,115) ------------------------------------ */
int tcct_act_fwd(const void* x, void* y, int64_t n, int kind, int dtype, tcct_stream_t stream);
int tcct_act_bwd(const void* x, const void* dy, void* dx, int64_t n, int kind, int dtype, tcct_stream_t stream);
int tcct_add(const void* a, const void* b, void* y, int64_t n, int dtype, tcct_stream_t stream);
int tcct_add_act_fwd(const void* a, const void* b, void* y, int64_t n, int kind, int dtype, tcct_stream_t stream);
int tcct_add_act_bwd(const void* a, const void* b, const void* dy, void* dx, int64_t n, int kind, int dtype,
                     tcct_stream_t stream);
/* y = x + scale[b] * z   (DropPath residual, nets/tcct.py:465,468; scale NULL -> 1) */
int tcct_residual_fwd(const void* x, const void* z, const float* scale, void* y, int B, int64_t per_sample,
                      int dtype, tcct_stream_t stream);
/* y[b,:] = scale[b] * x[b,:] */
int tcct_scale_rows(const void* x, const float* scale, void* y, int B, int64_t per_sample, int dtype,
                    tcct_stream_t stream);
/* y = alpha * (a + b + c)  (norm_add mean, nets/tcct.py:942);  y = alpha * x */
/* z = (a1[c] y1 + b1[c]) + (a2[c] y2 + b2[c]), ab = {a[C], b[C]}: the encoder fusion BN(tran_vit(v)) + BN(tran_cnn(c)) (nets/tcct.py:1016-1024) with both
 * train-mode BatchNorms applied in ONE pass; C % 8 == 0 */
int tcct_affine2_add(const void* y1, const float* ab1, const void* y2, const float* ab2, void* z, int64_t M, int C, int dtype, tcct_stream_t stream);
int tcct_add3_scale(const void* a, const void* b, const void* c, void* y, int64_t n, float alpha, int dtype,
                    tcct_stream_t stream);
int tcct_scale(const void* x, void* y, int64_t n, float alpha, int dtype, tcct_stream_t stream);
/* torch.cat(dim=1) on NHWC rows (nets/tcct.py:614) and its backward */
int tcct_concat2(const void* a, const void* b, void* y, int64_t M, int Ca, int Cb, int dtype, tcct_stream_t stream);
int tcct_split2(const void* dy, void* da, void* db, int64_t M, int Ca, int Cb, int dtype, tcct_stream_t stream);
/* fp32 accumulate: y += x  (gradient accumulation of parameter grads) */
int tcct_axpy_f32(const float* x, float* y, int64_t n, float alpha, tcct_stream_t stream);

/* ---- BatchNorm2d, train mode (nets/tcct.py:80,125,544,811,817,822,874,893,966-974,979; reg.py:73) with the
 * neighbouring activation fused: y = post( a[c]*pre(x) + b[c] ).  x viewed as [M,C], M = N*H*W. ----------- */
int tcct_bn_stats(const void* x, int64_t M, int C, int pre_act, double* sums /*[2C]*/, int dtype, tcct_stream_t stream);
int tcct_bn_finalize(const double* sums, int64_t M, int C, const float* gamma, const float* beta, float eps,
                     float momentum, float* running_mean, float* running_var, int64_t* num_batches_tracked,
                     float* mean_rstd /*[2C]*/, float* ab /*[2C]*/, tcct_stream_t stream);
/* eval mode: ab from running stats (KiteSeg.val, kite/loop_seg.py:68) */
int tcct_bn_eval_ab(int C, const float* gamma, const float* beta, float eps, const float* running_mean,
                    const float* running_var, float* mean_rstd, float* ab, tcct_stream_t stream);
int tcct_bn_apply(const void* x, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act, int dtype,
                  tcct_stream_t stream);
/* train-mode apply straight from the batch sums: tcct_bn_finalize + tcct_bn_apply[_add] in one launch (res nullable) */
int tcct_bn_apply_train(const void* x, const void* res, void* y, int64_t M, int C, const double* sums, const float* gamma,
                        const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                        int64_t* num_batches_tracked, float* mean_rstd, float* ab, int pre_act, int post_act, int dtype,
                        tcct_stream_t stream);
/* The last BatchNorm of an encoder level and the `self.pool` behind it (reference nets/tcct.py:820-823 block5, :876-884 CrossResNet.forward:
 * the level's output is kept AND pooled) in one pass: z [N,H,W,C] = post(BN_train(pre(x))), pooled [N,H/2,W/2,C] = MaxPool2d(2)(z), amax
 * [N,H/2,W/2,C/4] bytes = positions of the window maxima (2 bits per channel) for the backward call; sums, running statistics, mean_rstd,
 * ab as in tcct_bn_apply_train.  Even H, W; C/4 must divide 256. */
int tcct_bn_pool_fwd_train(const void* x, void* z, void* pooled, void* amax, int N, int H, int W, int C, const double* sums, const float* gamma,
                           const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                           int64_t* num_batches_tracked, float* mean_rstd, float* ab, int pre_act, int post_act, int dtype,
                           tcct_stream_t stream);
/* its backward: dx = BatchNorm_backward(dskip + MaxPool_backward(dpool)) -- what autograd computes for the two consumers of z -- without
 * writing that sum (dskip nullable); sums fp64 [2C] scratch, dgamma / dbeta fp32 [C] overwritten */
int tcct_bn_pool_bwd(const void* x, const void* dpool, const void* dskip, const void* amax, void* dx, int N, int H, int W, int C, const float* mean_rstd,
                     const float* ab, int pre_act, int post_act, double* sums, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream);
/* y = post(a*pre(x)+b) + res (res, y like x): the normalisation pass with the residual / branch sum that follows it folded in
 * (InvRes `x + conv2(f)`, nets/tcct.py:563-572; `tran_vit(x) + tran_cnn(c)`, :1012-1015) */
int tcct_bn_apply_add(const void* x, const void* res, void* y, int64_t M, int C, const float* ab, int pre_act, int post_act, int dtype,
                      tcct_stream_t stream);
int tcct_bn_bwd_reduce(const void* x, const void* dy, int64_t M, int C, const float* mean_rstd, const float* ab,
                       int pre_act, int post_act, double* sums /*[2C]*/, int dtype, tcct_stream_t stream);
int tcct_bn_bwd_apply(const void* x, const void* dy, void* dx, int64_t M, int C, const float* mean_rstd,
                      const float* ab, const float* gamma, const double* sums, int pre_act, int post_act,
                      float* dgamma, float* dbeta, int dtype, tcct_stream_t stream);
/* per-channel constants of the train-mode BatchNorm backward for kernels that rebuild the input gradient on load instead of reading it from
 * a tcct_bn_bwd_apply pass: du = a (dz' - S1/M - xhat S2/M) = c1 dz' + c2 y + c3 with y the BatchNorm input (pre = none).  sums [2][C] fp64
 * = {S1 = sum dz', S2 = sum dz' xhat} (tcct_bn_bwd_reduce) or, raw = 1, {sum dz', sum dz' y} (reduction epilogues).  Writes coef [5][C] =
 * {c1, c2, c3, a, b}, dgamma [C] = S2, dbeta [C] = S1. */
int tcct_bn_bwd_coef(const double* sums, int raw, int64_t M, int C, const float* mean_rstd, const float* ab, float* coef, float* dgamma,
                     float* dbeta, tcct_stream_t stream);
/* {sum dz', sum dz' y} (raw, from a reduction epilogue) -> {S1, S2 = sum dz' xhat}: the form tcct_bn_bwd_apply reads */
/* two BatchNorms (no activation) whose outputs were added and so share dz: raw_i fp64 [2C] (zero on entry) += {sum dz, sum dz y_i} from one pass (the raw form
 * tcct_bn_sums_from_raw / tcct_pw_bwd_bn_sums(raw = 1) take); C % 4 == 0 */
int tcct_bn_bwd_reduce2_raw(const void* y1, const void* y2, const void* dz, int64_t M, int C, double* raw1, double* raw2, int dtype, tcct_stream_t stream);
int tcct_bn_sums_from_raw(const double* raw, const float* mean_rstd, int C, double* sums, tcct_stream_t stream);

/* fused CrossCNNBlock junction y = act(BN_A(pre(xa)) + BN_B(pre(xb))) (nets/tcct.py:811,817,825-826: LeakyReLU -> BN on both
 * branches, then F.gelu of the sum), train mode; abA, abB and mean_rstdA, mean_rstdB come from tcct_bn_finalize; sums fp64 [4C] */
int tcct_bn2_add_act_fwd(const void* xa, const void* xb, void* y, int64_t M, int C, const float* abA, const float* abB, int pre_act,
                         int act, int dtype, tcct_stream_t stream);
/* train-mode junction straight from the two sets of batch sums: 2 x tcct_bn_finalize + tcct_bn2_add_act_fwd in one launch */
int tcct_bn2_add_act_train(const void* xa, const void* xb, void* y, int64_t M, int C, const double* sumsA, const float* gammaA,
                           const float* betaA, float* running_meanA, float* running_varA, int64_t* nbtA, float* mean_rstdA,
                           float* abA, const double* sumsB, const float* gammaB, const float* betaB, float* running_meanB,
                           float* running_varB, int64_t* nbtB, float* mean_rstdB, float* abB, float eps, float momentum,
                           int pre_act, int act, int dtype, tcct_stream_t stream);
int tcct_bn2_add_act_bwd_reduce(const void* xa, const void* xb, const void* dy, int64_t M, int C, const float* mean_rstdA,
                                const float* abA, const float* mean_rstdB, const float* abB, int pre_act, int act, double* sums,
                                int dtype, tcct_stream_t stream);
int tcct_bn2_add_act_bwd_apply(const void* xa, const void* xb, const void* dy, void* dxa, void* dxb, int64_t M, int C,
                               const float* mean_rstdA, const float* abA, const float* mean_rstdB, const float* abB,
                               const double* sums, int pre_act, int act, float* dgammaA, float* dbetaA, float* dgammaB,
                               float* dbetaB, int dtype, tcct_stream_t stream);

/* ---- nn.LayerNorm(C, eps=1e-6) over the channel dim of tokens [M,C] (nets/tcct.py:427,454-455,461,467) ---- */
int tcct_layernorm_fwd(const void* x, void* y, int64_t M, int C, const float* gamma, const float* beta, float eps,
                       float* mean_rstd /*[M,2]*/, int dtype, tcct_stream_t stream);
int tcct_layernorm_bwd(const void* x, const void* dy, void* dx, int64_t M, int C, const float* gamma,
                       const float* mean_rstd, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream);
/* dx = LN_bwd(dy) + res, res [M,C]: gradient of the residual path around the normalisation (x + f(LN(x)), tcct.py:461-468) */
int tcct_layernorm_bwd_add(const void* x, const void* dy, const void* res, void* dx, int64_t M, int C, const float* gamma,
                           const float* mean_rstd, float* dgamma, float* dbeta, int dtype, tcct_stream_t stream);

/* ---- dense conv2d / nn.Linear (nets/tcct.py:41-43,72,124,809-821,873,892,897,966-997) ------------------- */
/* x [N,H,W,Cin] (Cin % 4 == 0; weight input channels Cin_w <= Cin, extra channels ignored),
 * w OIHW fp32 [Cout,Cin_w,KH,KW], y [N,Ho,Wo,Cout]. */
int tcct_conv2d_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int Cin,
                    int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int in_dtype, int out_dtype,
                    tcct_stream_t stream);
/* stride-1 input gradient: dy [N,H,W,Cout] -> dx [N,H,W,Cin] (Cin == Cin_w here) */
int tcct_conv2d_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int Cin, int Cout, int KH,
                      int KW, int padh, int padw, int dy_dtype, int dx_dtype, tcct_stream_t stream);
/* dw OIHW fp32 (overwritten), dbias fp32 [Cout] (nullable, overwritten) */
int tcct_conv2d_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int Cin,
                      int Cin_w, int Cout, int KH, int KW, int stride, int padh, int padw, int x_dtype, int dy_dtype,
                      tcct_stream_t stream);

/* fp32 MFMA convolution of the parity mode (compute dtype float32): dense 32 -> 32 channels, stride 1, 'same' padding, 3x3 / 1xk / kx1 with
 * k <= 13 (reference nets/tcct.py:808-822,892,978), fp32 NHWC tensors, v_mfma_f32_32x32x2_f32 (fp32 products, fp32 accumulation).
 * pack: OIHW fp32 -> [tap][co][ci] fp32 (transposed = 1: flipped taps, co/ci swapped = the weights of the input-gradient convolution);
 * fwd: y = conv(x) + bias [+ yadd] (yadd nullable: the gradient reaching x through its other consumers); wgrad: dw OIHW, dbias nullable. */
int tcct_conv32f_pack_weights(const float* w, float* wp, int KH, int KW, int transposed, tcct_stream_t stream);
int tcct_conv32f_fwd(const float* x, const float* wp, const float* bias, const float* yadd, float* y, int N, int H, int W, int KH, int KW, int PH,
                     int PW, tcct_stream_t stream);
int tcct_conv32f_wgrad(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW,
                       tcct_stream_t stream);
/* the same kernels on 32-channel slabs of wider fp32 tensors (MPViT stem[1] 32 -> 64, the wide CNN encoder of stc_tb / gtc_tb): *_sub packs one 32 x 32
 * block of the weight, fwd_strided reads slab xo of xs channels and writes (accumulate = 1: adds to) slab yo of ys, wgrad_strided accumulates one block */
int tcct_conv32f_pack_weights_sub(const float* w, float* wp, int KH, int KW, int transposed, int cin_total, int o_off, int i_off, tcct_stream_t stream);
int tcct_conv32f_fwd_strided(const float* x, const float* wp, const float* bias, float* y, int N, int H, int W, int KH, int KW, int PH, int PW, int xs,
                             int xo, int ys, int yo, int accumulate, tcct_stream_t stream);
int tcct_conv32f_wgrad_strided(const float* x, const float* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH, int PW, int xs,
                               int xo, int ds, int dof, int ldi, int o_off, int i_off, tcct_stream_t stream);
/* ... and the 1x1 convolutions / nn.Linear of the same mode (nets/tcct.py:41-43,124,532-546,600,966-997): y [M,N] = x [M,K] W^T + bias with
 * fp32 rows, K and N multiples of 32; transposed = 1 reads w as [K,N] (the input gradient dx = dy W); wgrad: dw [N,K], dbias [N] nullable, N <= 160 */
int tcct_pwf_fwd(const float* x, const float* w, const float* bias, float* y, int64_t M, int K, int N, int transposed, tcct_stream_t stream);
int tcct_pwf_wgrad(const float* x, const float* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream);
/* MFMA implicit-GEMM path for the hot family: 32 -> 32 channels, stride 1, 'same' padding, bf16 NHWC, any KHxKW
 * (the 3x3 and 1xk / kx1 cross-convolutions of CrossCNNBlock, reference nets/tcct.py:808-822, and the decoder 3x3s).
 * wp = weights packed by tcct_conv32_pack_weights to bf16 [KH*KW][32 co][32 ci]; transposed=1 packs the flipped/transposed
 * weights so that the same kernel computes the input gradient (dx = conv32_fwd(dy, wp_T, NULL)). */
int tcct_conv32_pack_weights(const float* w, void* wp, int KH, int KW, int transposed, tcct_stream_t stream);
/* all 32 -> 32 convolution weights of a step in one launch: desc = device array of n records {const float* w (OIHW fp32); void* wp2 (2 KH KW 1024
 * bf16: forward pack, then the flipped / transposed input-gradient pack); int64 KH; int64 KW} */
int tcct_conv32_pack_weights_multi(const void* desc, int n, tcct_stream_t stream);
/* both packs in one launch: wp2 [2][KH*KW*1024] = {forward pack, input-gradient pack}; the backward pass reuses the second half */
int tcct_conv32_pack_weights_both(const float* w, void* wp2, int KH, int KW, tcct_stream_t stream);
int tcct_conv32_fwd(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH,
                    int PW, tcct_stream_t stream);

/* forward + fused train-mode BatchNorm statistics of the consumer: stats[64] fp64 (zero on entry) += per-channel {sum, sum of
 * squares} of pre_act(y), y as stored -- replaces the tcct_bn_stats pass (conv -> [LeakyReLU ->] BN, nets/tcct.py:808-822,892) */
int tcct_conv32_fwd_bnstats(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH,
                            int PW, double* stats, int pre_act, tcct_stream_t stream);
/* Inference epilogues (eval-mode nn.BatchNorm2d folded into the producing convolution; reference kite/loop_seg.py:21-33
 * runs the same modules under model.eval()):  y = post_act(a[c] * pre_act(conv(x) + bias[c]) + b[c]),  ab = {a[C], b[C]} from
 * tcct_bn_eval_ab (NULL: a = 1, b = 0, activations only).  Same operands as tcct_conv32_fwd / tcct_pw_fwd. */
int tcct_conv32_fwd_affine(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW,
                           int PH, int PW, const float* ab, int pre_act, int post_act, tcct_stream_t stream);
int tcct_pw_fwd_affine(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, const float* ab,
                       int pre_act, int post_act, int out_dtype, tcct_stream_t stream);
/* inference: y = res + (a[c] * (x W^T + bias[c]) + b[c]): `x + BN_eval(conv2(f))` of the InvRes block (nets/tcct.py:563-572) as one GEMM; K = N in {64, 96, 128}; ab as
 * tcct_pw_fwd_affine (NULL: a = 1, b = 0) */
int tcct_pw_fwd_affine_residual(const void* x, const float* w, const float* bias, const float* ab, const void* res, void* y, int64_t M, int K, int N,
                                tcct_stream_t stream);
/* inference: y = post_act(a[c] * ([x1 | x2] W^T + bias[c]) + b[c]) over the never-materialised concatenation of two 64-channel tensors, N = 96 (`aggregate` of MHCA stage 0,
 * nets/tcct.py:600-616) */
int tcct_pw_fwd_cat2_affine(const void* x1, const void* x2, const float* w, const float* bias, const float* ab, int post_act, void* y, int64_t M, int K, int N,
                            tcct_stream_t stream);
/* inference: y = res + (a[c] * (hswish(a_prev[k] * y_prev + b_prev[k]) W^T + bias[c]) + b[c]): the InvRes tail (nets/tcct.py:563-572) as one GEMM over the depthwise
 * convolution's raw output; ab_prev = {a[K], b[K]} of `norm`, ab = {a[N], b[N]} of conv2.bn (both eval mode); K = N in {64, 96, 128} */
int tcct_pw_fwd_xaff_affine_residual(const void* y_prev, const float* ab_prev, const float* w, const float* bias, const float* ab, const void* res, void* y,
                                     int64_t M, int K, int N, tcct_stream_t stream);
/* y = conv(x) + bias + res, res bf16 NHWC [N,H,W,32] (not overlapping y): used as the input gradient of a convolution whose input has
 * a second consumer (CrossCNNBlock: x feeds block12 and block34, nets/tcct.py:826) -- no separate gradient accumulation pass */
int tcct_conv32_fwd_add(const void* x, const void* wp, const float* bias, const void* res, void* y, int N, int H, int W, int KH, int KW,
                        int PH, int PW, tcct_stream_t stream);
/* Two dense 3x3 32 -> 32 convolutions with nothing between them as ONE launch (csrc/conv_chain.hip, round 5): y = conv(conv(x; wp1, bias1); wp2, bias2),
 * mid = the first convolution's output -- written (the backward pass needs it) but not read back, the second convolution takes it from LDS.  Replaces
 * nn.Conv2d -> nn.Conv2d of CrossCNNBlock.block12 (reference nets/tcct.py:808-810: no nonlinearity between the two) and, with the flipped / transposed
 * packs, the input-gradient chain of the same pair.  stats (fp64 [64] zero on entry, or NULL): += {sum, sum of squares} per channel of LeakyReLU(y) as
 * stored (the BatchNorm behind block12, :811); res (or NULL): y += res before the store (as tcct_conv32_fwd_add); mid == NULL (inference): the intermediate is
 * not written at all.  Bit-identical to two tcct_conv32_fwd calls. */
int tcct_conv32_chain33(const void* x, const void* wp1, const float* bias1, void* mid, const void* wp2, const float* bias2, void* y,
                        const void* res, int N, int H, int W, double* stats, tcct_stream_t stream);
/* the same kernels on 32-channel slabs of wider NHWC tensors (x: xs channels/pixel, slab at xo; y: ys, yo; accumulate adds
 * into y) and on 32x32 sub-blocks (o_off, i_off) of an OIHW weight with cin_total input channels: 32->64 / 64->32 convolutions
 * (MPViT stem[1], nets/tcct.py:682-689) run as 32x32 sub-GEMMs.  wgrad_strided ACCUMULATES: zero dw/dbias first. */
int tcct_conv32_pack_weights_sub(const float* w, void* wp, int KH, int KW, int transposed, int cin_total, int o_off, int i_off,
                                 tcct_stream_t stream);
int tcct_conv32_fwd_strided(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH,
                            int PW, int xs, int xo, int ys, int yo, int accumulate, tcct_stream_t stream);
/* the same slab call + the statistics of the train-mode BatchNorm behind the wide convolution (MPViT stem[1], nets/tcct.py:682-689 with the
 * BatchNorm of Conv2d_BN :80): use it for the LAST input slab of an output slab; stats fp64 {sum[ys], sum of squares[ys]} of the whole tensor,
 * zero before the first slab */
int tcct_conv32_fwd_strided_bnstats(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH,
                                    int PW, int xs, int xo, int ys, int yo, int accumulate, double* stats, int pre_act, tcct_stream_t stream);
/* inference: one 32-channel output slab of a wider convolution with ONE 32-channel input slab (MPViT stem[1], 32 -> 64 3x3, nets/tcct.py:682-689) with the eval-mode
 * BatchNorm + activation of those channels in the epilogue; ab = {a[32], b[32]} of the slab (tcct_bn_eval_ab on the 32-channel slices).  Replaces
 * tcct_conv32_fwd_strided + tcct_bn_apply in `KiteSeg.predict`. */
int tcct_conv32_fwd_strided_affine(const void* x, const void* wp, const float* bias, void* y, int N, int H, int W, int KH, int KW, int PH, int PW, int xs,
                                   int xo, int ys, int yo, const float* ab, int pre_act, int post_act, tcct_stream_t stream);
/* selects the kernel behind tcct_conv32_wgrad for plain 3x3 convolutions: 0 (default) = wave-private row streams where a wave gets >= 32 rows (levels 0-1), else the rolling-row form (one x fragment per halo row meets a register window of three dy fragments, 6 waves x 2 blocks per
 * CU); 1 = the generic register-staged kernel every other shape takes (comparison arm of the bit-compatibility test; bench.py --wgrad-mode 1 for a whole run);
 * 2 = row streams for every plain 3x3; any other value only queries.  Returns the previous mode.  (No reference counterpart: the reference calls ATen's
 * convolution backward, nets/tcct.py:808-822 through autograd.) */
int64_t tcct_conv32_wgrad_mode(int mode);
/* the same for the plain 32-channel 3x3 forward / input gradient (tcct_conv32_fwd, tcct_conv32_fwd_bnstats with no / LeakyReLU pre-activation): 0 (default) =
 * the row-stream kernel where a wave gets >= 48 rows, 1 = the tiled kernel everywhere, 2 = the row-stream kernel for every
 * plain 3x3.  The two kernels give bit-identical outputs.  Returns the previous mode; any other value only queries. */
int64_t tcct_conv32_fwd_mode(int mode);
int tcct_conv32_wgrad_strided(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH,
                              int PW, int xs, int xo, int ds, int dof, int cin_total, int o_off, int i_off, tcct_stream_t stream);
/* weight/bias gradient of the same family (ds_read_b64_tr_b16 transposing LDS reads feed the pixel-contraction MFMA);
 * dw OIHW fp32 [32,32,KH,KW] and dbias fp32 [32] (nullable) are overwritten */
int tcct_conv32_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int KH, int KW, int PH,
                      int PW, tcct_stream_t stream);
/* Fused backward of a dense 3x3 32->32 'same' convolution (autograd of nets/tcct.py:809-810,816,821,892,978): dx = conv(dy, flipped
 * weights) (+ dskip, nullable), dw [32,32,3,3] += dy (x) x, dbias [32] += sum dy (nullable) with dy read from HBM once.  wp_t = the
 * input-gradient weight pack (tcct_conv32_pack_weights_both's second half).  dw / dbias are cleared first unless prezeroed. */
int tcct_conv32_bwd3x3(const void* x, const void* dy, const void* wp_t, const void* dskip, void* dx, float* dw, float* dbias, int N, int H,
                       int W, tcct_stream_t stream);

/* MFMA pointwise (1x1 conv / nn.Linear) path, bf16 rows [M,K] with K % 32 == 0 (ViT 1x1s and MLPs, FTC tran_x, t32x and
 * aux heads; reference nets/tcct.py:41-43,124,532-546,600,966-997).  w fp32 [N,K] (transposed=0) or [K,N] (transposed=1,
 * i.e. dx = dy * W for the input gradient); y [M,N] bf16 or fp32.  wgrad: dw fp32 [N,K], dbias [N] nullable, N <= 160. */
int tcct_pw_fwd(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, int transposed,
                int out_dtype, tcct_stream_t stream);
int tcct_pw_wgrad(const void* x, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream);
/* Fused backward of a 1x1 convolution / nn.Linear (autograd of nets/tcct.py:41-43,124,532-546,966-997): dx = dy W (+ res, nullable: the
 * gradient reaching x through its other consumers), dw += dy^T x, dbias += sum dy in ONE pass over dy.  bf16 rows, w fp32 [N,K],
 * K and N in {32,64,96,128}; dw / dbias are cleared first unless tcct_set_outputs_prezeroed(1). */
int tcct_pw_bwd(const void* x, const void* dy, const float* w, const void* res, void* dx, float* dw, float* dbias, int64_t M, int K, int N,
                tcct_stream_t stream);
/* ... writing both dx_sum = dy W + res and dx_plain = dy W (MPUpBlock tail + the `x_i + y_i` of FTC.forward, nets/tcct.py:908-914,1028-1031) */
/* ... over a concatenation x = [x1 | x2] (two tensors of 64 channels), dx = [dx1 | dx2]: backward of tcct_pw_fwd_cat2 (MHCA_stage.aggregate,
 * nets/tcct.py:600-616), no bias */
int tcct_pw_bwd_cat2(const void* x1, const void* x2, const void* dy, const float* w, void* dx1, void* dx2, float* dw, int64_t M, int K, int N,
                     tcct_stream_t stream);
int tcct_pw_bwd_residual2(const void* x, const void* dy, const float* w, const void* res, void* dx_sum, void* dx_plain, float* dw, float* dbias,
                          int64_t M, int K, int N, tcct_stream_t stream);
/* Backward of z = post(BN_train(x W^T + bias)) [+ residual] (Conv2d_BN / DWConv2d_BN.pwconv / FTC.tran_*, reference nets/tcct.py:55-97,
 * 124-126,966-974) given dz, the gradient of the BatchNorm OUTPUT: the BatchNorm backward apply pass is folded into the staging of this
 * kernel (dy_conv = c1 dz post'(a y + b) + c2 y + c3 per element, coef [5][N] = {c1, c2, c3, a, b} from tcct_bn_bwd_coef), and, with
 * red_post >= 0, the reduction pass of the BatchNorm IN FRONT of the convolution (x = post_prev(a_prev y_prev + b_prev), ab_prev [2][K]) is
 * folded into its dx epilogue: sums_prev [2][K] fp64 += {sum dz', sum dz' y_prev} (raw form).  x2 / dx2 non-NULL: the concatenated operands
 * of tcct_pw_bwd_cat2 (the reduction then covers the first half).  tcct_pw_bwd_bn_supported says which (K, N, kinds) have a kernel. */
int64_t tcct_pw_bwd_bn_supported(int K, int N, int post, int red_post, int split);
int tcct_pw_bwd_bn(const void* x, const void* x2, const void* dz, const void* y, const float* coef, int post, const float* w, const void* res,
                   void* dx, void* dx2, float* dw, float* dbias, int64_t M, int K, int N, const void* y_prev, const float* ab_prev,
                   int red_post, double* sums_prev, tcct_stream_t stream);
/* ... with the constants derived inside the kernel from the BatchNorm's two batch sums (sums [2][N] fp64: {sum dz', sum dz' xhat} from
 * tcct_bn_bwd_reduce, or raw = 1 {sum dz', sum dz' y} from a reduction epilogue); writes dgamma / dbeta [N]: one launch per BatchNorm fewer */
int tcct_pw_bwd_bn_sums(const void* x, const void* x2, const void* dz, const void* y, const double* sums, int raw, const float* mean_rstd,
                        const float* ab, float* dgamma, float* dbeta, int post, const float* w, const void* res, void* dx, void* dx2, float* dw,
                        float* dbias, int64_t M, int K, int N, const void* y_prev, const float* ab_prev, int red_post, double* sums_prev,
                        tcct_stream_t stream);
/* the same for an N-column slab of a wider output: dy rows have stride ldy elements (multiple of 8) and dy / dw / dbias point at the slab
 * (nn.Linear(dim, 3 dim) of FactorAtt_ConvRelPosEnc, nets/tcct.py:307, runs as slabs of <= 160 columns) */
int tcct_pw_wgrad_strided(const void* x, const void* dy, int64_t ldy, float* dw, float* dbias, int64_t M, int K, int N,
                          tcct_stream_t stream);
/* Linear + residual: y = res + scale[m / per_sample] * (x W^T + bias), bf16, N % 32 == 0, scale fp32 [M / per_sample] nullable
 * (Mlp.fc2 followed by `x + drop_path(...)`, nets/tcct.py:468); y_plain (nullable) additionally receives x W^T + bias itself: the decoder's
 * `post` convolution whose output feeds both the next stage and `x_i + y_i` (nets/tcct.py:1028-1031) */
int tcct_pw_fwd_residual(const void* x, const float* w, const float* bias, const void* res, const float* scale, int64_t per_sample,
                         void* y, void* y_plain, int64_t M, int K, int N, tcct_stream_t stream);
/* InvRes.norm -> conv2 (nets/tcct.py:563-572) without the normalisation pass (round 4): the convolution reads the BatchNorm's INPUT y_prev and applies
 * z = hswish(a_prev y_prev + b_prev) while it stages its tiles, forward and backward; ab_prev = {a[K], b[K]} from tcct_bn_finalize.  K = N in {64, 96}.
 *   fwd : y = z W^T + bias + batch statistics of y (stats fp64 [2N], zero on entry) for the BatchNorm behind
 *   bwd : tcct_pw_bwd_bn_sums with x rebuilt from y_prev; red_post = TCCT_ACT_HSWISH (K = 64: + the reduction sums of the BatchNorm in front) or -1 (96) */
int tcct_pw_fwd_bnstats_xaff(const void* y_prev, const float* ab_prev, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats,
                             tcct_stream_t stream);
int tcct_pw_bwd_bn_sums_xaff(const void* y_prev, const float* ab_prev, const void* dz, const void* y, const double* sums, int raw, const float* mean_rstd,
                             const float* ab, float* dgamma, float* dbeta, const float* w, const void* res, void* dx, float* dw, float* dbias, int64_t M,
                             int K, int N, int red_post, double* sums_prev, tcct_stream_t stream);
/* Mlp (nets/tcct.py:29-53: fc1 -> GELU -> fc2) without the activation passes (round 4): x1 is the PRE-activation fc1 wrote; GELU is applied while the
 * tile is staged, so h = gelu(x1) and, backwards, dh never exist in HBM.  K = N in {64, 96} (mpvit_tiny: hidden = dim; stages 0 and 1 = 97 % of the bytes).
 *   fwd: y = res + scale[m / per_sample] * (gelu(x1) W^T + bias)   (fc2 + DropPath scale + residual, nets/tcct.py:468; scale nullable)
 *   bwd: dx1 = (dy W) gelu'(x1), dw += dy^T gelu(x1), dbias += sum dy   (dw / dbias cleared here unless tcct_set_outputs_prezeroed) */
int tcct_pw_fwd_gelu_residual(const void* x1, const float* w, const float* bias, const void* res, const float* scale, int64_t per_sample, void* y,
                              int64_t M, int K, int N, tcct_stream_t stream);
/* tcct_pw_bwd_residual2 for the decoder blocks whose output has TWO gradients (its own consumer and `x_i + y_i`): dy = dy_a + dy_b is summed while the tile is
 * staged, no separate add pass; K = N = 32 */
int tcct_pw_bwd_residual2_sum(const void* x, const void* dy_a, const void* dy_b, const float* w, const void* res, void* dx_sum, void* dx_plain, float* dw,
                              float* dbias, int64_t M, int K, int N, tcct_stream_t stream);
/* fused backward of Mlp.fc1 BEHIND MHCABlock.norm2 (nets/tcct.py:466-468): x = LN(t) as stored, t and mean_rstd [M][2] the LayerNorm's input and saved statistics,
 * res = the gradient that reaches t through the residual path; dt = LN^T(dy W) + res (the gradient of x is never written), dw / dbias of the Linear, dgamma / dbeta
 * [K] of the LayerNorm.  K = N = 64. */
int tcct_pw_bwd_lnb(const void* x, const void* dy, const float* w, const void* t, const float* mean_rstd, const float* gamma, const void* res, void* dt,
                    float* dw, float* dbias, float* dgamma, float* dbeta, int64_t M, int K, int N, tcct_stream_t stream);
int tcct_pw_bwd_gelu(const void* x1, const void* dy, const float* w, void* dx1, float* dw, float* dbias, int64_t M, int K, int N, tcct_stream_t stream);
/* input gradient with a second gradient folded in: dx_plain = dy W, dx_sum = dy W + res (w [Nout,K] as stored; res, dx_* [M,K] bf16; dx_plain nullable):
 * backward of the decoder block tail (MPUpBlock, tcct.py:908-914): dx_plain continues into the resize, dx_sum is the skip's gradient */
int tcct_pw_dgrad_residual(const void* dy, const float* w, const void* res, void* dx_sum, void* dx_plain, int64_t M, int Nout, int K,
                           tcct_stream_t stream);
/* Concatenation-free pointwise convolution over [x1 | x2] (MHCA_stage.aggregate, nets/tcct.py:600-616): forward (+ optional fused
 * BN statistics), input gradient written to two tensors, weight gradient -- the channel concatenation is never materialised */
int tcct_pw_fwd_cat2(const void* x1, const void* x2, int K1, const float* w, const float* bias, void* y, int64_t M, int K, int N,
                     double* stats, int pre_act, tcct_stream_t stream);
/* y fp32 [B*2uH*2uW, N] = x W^T + bias + resize_x2(up)[..., :N] (bilinear, align_corners=True), N <= 8; up fp32 [B,uH,uW,8] (channels >= N zero): the level-0 aux
 * head with the resize commuted behind the convolution adds the low-resolution product in the epilogue of the full-resolution GEMM */
int tcct_pw_fwd_f32_upadd(const void* x, const float* w, const float* bias, float* y, int B, int uH, int uW, int K, int N, const float* up, tcct_stream_t stream);
int tcct_pw_dgrad_split2(const void* dy, const float* w, void* dx1, void* dx2, int K1, int64_t M, int Nout, int K, tcct_stream_t stream);
/* fused backward over the concatenation incl. the bias gradient: dx1 | dx2 = dy W, dw += dy^T [x1 | x2], dbias += sum dy; two halves of 64 channels
 * (K = 128) or, for N = 32, of 32 channels (K = 64: the decoder's composed tail below) */
int tcct_pw_bwd_cat2_bias(const void* x1, const void* x2, const void* dy, const float* w, void* dx1, void* dx2, float* dw, float* dbias, int64_t M,
                          int K, int N, tcct_stream_t stream);
/* The last decoder block's tail as one GEMM (csrc/decoder_tail.hip): MPUpBlock.post (nets/tcct.py:913), `x_0 + y_0` (:1031) and t324 (:1035-1040) are
 * 1x1 convolutions / adds with nothing nonlinear between them, so g0 = Wc [up(y) | skip] + c with Wc = [W2 W1 | W2 W1 + W2], c = W2 b1 + b2
 * (w1, b1 = post; w2, b2 = t324; all fp32, 32 channels).  compose: Wc [32][64], c [32]; compose_bwd: the four gradients from dWc, dc. */
int tcct_tail_compose(const float* w1, const float* b1, const float* w2, const float* b2, float* wc, float* c, tcct_stream_t stream);
int tcct_tail_compose_bwd(const float* w1, const float* b1, const float* w2, const float* dwc, const float* dc, float* dw1, float* db1, float* dw2,
                          float* db2, tcct_stream_t stream);
/* ... and through the level-0 aux head (FTC.aux0, nets/tcct.py:994,1041) when nothing else reads g0: logits0 = (W3 Wc) [up(y) | skip] + (W3 c + b3);
 * w3 [C][32], b3 [C], C <= 8.  compose3: wcc [C][64] + its halves wa, wb [C][32] (what the small-N input-gradient kernel takes), ccc [C];
 * compose3_bwd: the six gradients from d wcc (as halves dwa, dwb) and d ccc.  tcct_pw_fwd_cat2_f32: the forward GEMM with fp32 logits. */
int tcct_tail_compose3(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, const float* b3, int C, float* wcc, float* wa,
                       float* wb, float* ccc, tcct_stream_t stream);
int tcct_tail_compose3_bwd(const float* w1, const float* b1, const float* w2, const float* b2, const float* w3, int C, const float* dwa, const float* dwb,
                           const float* dccc, float* dw1, float* db1, float* dw2, float* db2, float* dw3, float* db3, tcct_stream_t stream);
int tcct_pw_fwd_cat2_f32(const void* x1, const void* x2, int K1, const float* w, const float* bias, float* y, int64_t M, int K, int N, tcct_stream_t stream);
/* Levels 1-3, same idea (nets/tcct.py:1036-1044): when nothing but aux_i reads g_i = t32x(x_i + y_i), logits_i = (Wa Wt) s_i + (Wa bt + ba) is ONE
 * 32 -> C pointwise convolution on s_i (tcct_pw_fwd with the composed weight); head_compose: wh [C][32], ch [C] from wt [32][32], bt [32] (t32x) and
 * wa [C][32], ba [C] (aux_i), C <= 16; head_compose_bwd: the four gradients from dwh, dch (outputs overwritten). */
int tcct_head_compose(const float* wt, const float* bt, const float* wa, const float* ba, int C, float* wh, float* ch, tcct_stream_t stream);
int tcct_head_compose_bwd(const float* wt, const float* bt, const float* wa, int C, const float* dwh, const float* dch, float* dwt, float* dbt,
                          float* dwa, float* dba, tcct_stream_t stream);
int tcct_pw_wgrad_cat2(const void* x1, const void* x2, int K1, const void* dy, float* dw, float* dbias, int64_t M, int K, int N,
                       tcct_stream_t stream);
/* pw_fwd + fused train-mode BatchNorm statistics of the consumer (bf16 output, N in {32,64,96,128}); stats fp64 [2N], zero on entry */
int tcct_pw_fwd_bnstats(const void* x, const float* w, const float* bias, void* y, int64_t M, int K, int N, double* stats, int pre_act,
                        tcct_stream_t stream);

/* first-layer helper: 4-channel NHWC image -> 32-channel 3x3 patch pixels (k = (ky*3+kx)*3+ch, zero for k >= 27) so that
 * CrossResNet.cnn[0] (nets/tcct.py:873) and MPViT stem[0] (stride 2, :674-681) run as 32->32 pointwise MFMA GEMMs */
int tcct_im2col3x3_c3(const void* x4, void* out, int N, int H, int W, int stride, int dtype, tcct_stream_t stream);
/* The same first layers WITHOUT the patch tensor (bf16): direct 3 -> 32 channel 3x3 convolution (pad 1, stride 1 or 2) of the 4-channel
 * image x4 [B,H,W,4]; w fp32 [32,3,3,3] as stored by nn.Conv2d (reference nets/tcct.py:873 `self.cnn[0]`, :674-681 `stem[0]`), y bf16
 * [B,Ho,Wo,32].  stats (nullable, fp64 [64], zero on entry): fused statistics of pre_act(y) for the train-mode BatchNorm behind it
 * (nets/tcct.py:873, :80); ab / pre_act / post_act (inference, exclusive with stats): y = post(a[c]*pre(conv+bias)+b[c]), ab from tcct_bn_eval_ab */
int tcct_c3_fwd(const void* x4, const float* w, const float* bias, void* y, int B, int H, int W, int stride, double* stats, int stat_pre,
                const float* ab, int pre_act, int post_act, tcct_stream_t stream);
/* its weight / bias gradient: dw fp32 [32,3,3,3], dbias fp32 [32] (nullable), overwritten; dy bf16 [B,Ho,Wo,32] (what autograd computes for
 * `F.conv2d(x, w, b, stride, 1)` at nets/tcct.py:873 / :80; the image needs no gradient) */
int tcct_c3_wgrad(const void* x4, const void* dy, float* dw, float* dbias, int B, int H, int W, int stride, tcct_stream_t stream);
/* The same first layers TOGETHER with the train-mode BatchNorm behind them as one store (round 4; csrc/c3_bn.hip): `self.cnn = Sequential(Conv2d(3, 32,
 * 3, 1, 1), BatchNorm2d(32))` (nets/tcct.py:873, post_act none) and `stem[0] = Conv2d_BN(3, 32, 3, 2, 1, act=Hardswish)` (nets/tcct.py:55-97,674-681,
 * post_act hswish).  The convolution output y is recomputed from the 4-channel image instead of being stored and re-read:
 *   fwd_train : statistics pass (no store) + normalising pass, z = post_act(a y + b) bf16 [B,Ho,Wo,32]; sums fp64 [64] (zero on entry) receives the
 *               batch sums of y; running_mean / running_var / num_batches_tracked are updated (nullable); mean_rstd [64], ab [64] for the backward
 *   bwd_reduce: raw fp64 [64] (zero on entry) = {sum dz', sum dz' y}, dz' = dz post_act'(a y + b); feed it to tcct_bn_bwd_coef(raw = 1), which
 *               also writes dgamma / dbeta
 *   bwd_wgrad : dw fp32 [32,3,3,3], dbias fp32 [32] (nullable) of the convolution from dy = c1 dz' + c2 y + c3 rebuilt in registers
 *               (coef fp32 [160] = {c1, c2, c3, a, b} of tcct_bn_bwd_coef); both overwritten */
int tcct_c3_bn_fwd_train(const void* x4, const float* w, const float* bias, void* z, int B, int H, int W, int stride, double* sums,
                         const float* gamma, const float* beta, float eps, float momentum, float* running_mean, float* running_var,
                         int64_t* num_batches_tracked, float* mean_rstd, float* ab, int post_act, tcct_stream_t stream);
/* inference: z = post_act(a y + b) with the eval-mode coefficients ab fp32 [64] = {a[32], b[32]} (tcct_bn_eval_ab) -- `KiteSeg.predict` / `val`
 * (kite/loop_seg.py:21-33,66-106) through the same first layers, one launch, y never stored */
int tcct_c3_bn_fwd_eval(const void* x4, const float* w, const float* bias, void* z, int B, int H, int W, int stride, const float* ab, int post_act,
                        tcct_stream_t stream);
int tcct_c3_bn_bwd_reduce(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride, const float* ab,
                          double* raw, int post_act, tcct_stream_t stream);
int tcct_c3_bn_bwd_wgrad(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride, const float* coef,
                         float* dw, float* dbias, int post_act, tcct_stream_t stream);
/* kernel A/B of the one-pass backward below (tools/c3bwd_bench.py, tests): 4 (default) = wave-private 32-pixel tiles (k_c3_bn_bwd_wave, round 6); 1..3 = the block-tile kernel
 * (128 pixels, two barriers per tile) with that many tiles requested ahead; 0 = the block-tile kernel on up to 1024 blocks (the round-5 launch); returns the previous value.
 * Test / measurement only. */
int64_t tcct_c3_bn_bwd_prefetch(int tiles_ahead);
/* ... or both in ONE pass over dz (round 4): dy = a (dz' - s1 - yh s2) is linear in per-pixel quantities, so dW = a (A1 - s2 A2 - s1 A3) with A1 = sum dz' patch,
 * A2 = sum yh patch, A3 = sum patch accumulated together with the batch sums; work fp32 [4160] and sums fp64 [96] are scratch (zero on entry when the outputs are
 * pre-zeroed, cleared here otherwise); dw [32,3,3,3], dbias [32] (nullable), dgamma, dbeta [32] are overwritten. */
int tcct_c3_bn_bwd_onepass(const void* x4, const float* w, const float* bias, const void* dz, int B, int H, int W, int stride, const float* mean_rstd,
                           const float* ab, float* work, double* sums, float* dw, float* dbias, float* dgamma, float* dbeta, int post_act, tcct_stream_t stream);
/* weight gradient of 1x1 convs with <= 8 outputs (5-class aux heads, nets/tcct.py:994-997); dy fp32 or bf16 */
int tcct_pw_wgrad_smalln(const void* x, const void* dy, float* dw, float* dbias, int64_t M, int K, int N, int x_dtype,
                         int dy_dtype, tcct_stream_t stream);

/* achievable-bandwidth yardstick of bench.py (`roofline.copy_ceiling`): a streaming copy with 16-byte accesses, one 8 KB chunk per block */
int tcct_stream_copy(const void* src, void* dst, int64_t nbytes, tcct_stream_t stream);

/* test tooling, no reference counterpart: number of launches of kernel family `which` since the last reset (0 k_conv32_chain33, 1 k_conv32_wgradk_stream,
 * 2 k_conv32_wgrad33_stream, 3 k_conv32_fwd33_stream); reset != 0 clears it after reading; -1 for an unknown family.  The entry points choose a kernel per shape:
 * tests/test_fullsize_gpu.py asserts with this that the bench-shape step was served by the row-stream / chain kernels it means to check. */
int64_t tcct_kernel_census(int which, int reset);
/* measurement tooling (tools/attrib_trace.py), no reference counterpart: an empty launch of `id` blocks x 64 threads -- the grid size is the only thing a
 * kernel trace / PMC table records about a dispatch, so a marker in front of every C-ABI call lets the tables be cut into calls; 1 <= id < 2^20 */
int tcct_marker(int id, tcct_stream_t stream);

/* ---- depthwise 3x3 (nets/tcct.py:114-122,206,535-543; nets/reg.py:66-67,72,74 as C=1 / groups=C) -------- */
int tcct_dwconv3x3_fwd(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C,
                       int stride, int add_input, int dtype, tcct_stream_t stream);
/* forward + fused statistics of the train-mode BatchNorm behind it (ResBlock: `self.dwconv` -> `self.norm`, reference nets/tcct.py:548-551,565):
 * stats fp64 [2C] (zero on entry) += {sum, sum of squares} per channel of y as stored -- replaces a tcct_bn_stats pass; C % 4 == 0 */
int tcct_dwconv3x3_fwd_bnstats(const void* x, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                               int add_input, double* stats, int dtype, tcct_stream_t stream);
int tcct_dwconv3x3_dgrad(const void* dy, const float* w, void* dx, int N, int H, int W, int C, int stride,
                         int add_input, int dtype, tcct_stream_t stream);
/* input gradient + res [N,H,W,C]: the gradient reaching the convolution's input through its other consumers (the stage input
 * feeds ConvPosEnc, InvRes.conv1 and the InvRes residual, tcct.py:563-572,604-616) is added in the same pass */
int tcct_dwconv3x3_dgrad_add(const void* dy, const float* w, const void* res, void* dx, int N, int H, int W, int C, int stride,
                             int add_input, int dtype, tcct_stream_t stream);
int tcct_dwconv3x3_wgrad(const void* x, const void* dy, float* dw, float* dbias, int N, int H, int W, int C,
                         int stride, int dtype, tcct_stream_t stream);
/* The depthwise convolution behind a train-mode BatchNorm + Hardswish whose normalisation pass is not run (round 4; Conv2d_BN -> DWConv2d_BN.dwconv and
 * InvRes.conv1 -> InvRes.dwconv, nets/tcct.py:55-97,114-122,535-543): x is that BatchNorm's INPUT y_prev, xab = {a[C], b[C]}; the kernels apply
 * z = hswish(a y_prev + b) (rounded to the activation type, zero outside the image) when a row enters the register window.  C % 4 == 0, C <= 256. */
int tcct_dwconv3x3_fwd_xaff(const void* x, const float* xab, const float* w, const float* bias, void* y, int N, int H, int W, int C, int stride,
                            double* stats, int dtype, tcct_stream_t stream);
int tcct_dwconv3x3_wgrad_xaff(const void* x, const float* xab, const void* dy, float* dw, float* dbias, int N, int H, int W, int C, int stride, int dtype,
                              tcct_stream_t stream);
/* ... and the stride-1 input gradient that also accumulates THAT BatchNorm's two backward sums (raw fp64 [2C], zero on entry: {sum dz', sum dz' y_prev} with
 * dz' = dz hswish'(a y_prev + b), dz = dx as stored -- the form tcct_bn_sums_from_raw and tcct_pw_bwd_bn_sums(raw = 1) take): no separate reduction pass */
int tcct_dwconv3x3_dgrad_bnred(const void* dy, const float* w, const void* y_prev, const float* ab_prev, void* dx, double* raw, int N, int H, int W, int C,
                               int dtype, tcct_stream_t stream);

/* ---- MetaPool on tokens [B,N,C] (nets/tcct.py:405-415,463): AvgPool2d(3,1,1,count_include_pad=False)(x) - x
 * taken over the (token, channel) plane, exactly as torch treats the 3-D tensor --------------------------- */
int tcct_metapool_fwd(const void* x, void* y, int B, int64_t N, int C, int dtype, tcct_stream_t stream);
int tcct_metapool_bwd(const void* dy, void* dx, int B, int64_t N, int C, int dtype, tcct_stream_t stream);
/* MHCABlock mixer branch with its residual: y = res + scale[b] * metapool(x) (nets/tcct.py:464-465; scale = DropPath mask / keep, fp32 [B],
 * NULL = 1) and its input gradient dx = scale[b] * metapool^T(dy) -- no separate residual / scale passes */
int tcct_metapool_residual_fwd(const void* x, const void* res, const float* scale, void* y, int B, int64_t N, int C, int dtype,
                               tcct_stream_t stream);
int tcct_metapool_scaled_bwd(const void* dy, const float* scale, void* dx, int B, int64_t N, int C, int dtype, tcct_stream_t stream);
/* ... and with the LayerNorm in front of the mixer folded in (csrc/ln_pool.hip; nets/tcct.py:457-465): y = t + scale[b] * (pool(a) - a), a = LN(t; gamma, beta,
 * eps) rounded to the activation type, in ONE pass (read t, write y); backward dt = dy + LN^T(scale[b] * (pool^T(dy) - dy)) in one pass (read dy, read t,
 * write dt; the LayerNorm statistics are recomputed from t).  C a multiple of 8 in 16..256; dgamma / dbeta fp32 [C] overwritten. */
int tcct_ln_metapool_residual_fwd(const void* t, void* y, int B, int64_t N, int C, const float* gamma, const float* beta, float eps, const float* scale,
                                  int dtype, tcct_stream_t stream);
/* ... and MHCABlock.norm2 of the row just produced (nets/tcct.py:466): y2 = LN(y; gamma2, beta2, eps2), mean_rstd2 fp32 [B*N*2] as tcct_layernorm_fwd writes it */
int tcct_ln_metapool_residual_ln_fwd(const void* t, void* y, void* y2, int B, int64_t N, int C, const float* gamma, const float* beta, float eps,
                                     const float* scale, const float* gamma2, const float* beta2, float eps2, float* mean_rstd2, int dtype, tcct_stream_t stream);
int tcct_ln_metapool_residual_bwd(const void* t, const void* dy, void* dt, int B, int64_t N, int C, const float* gamma, float eps, const float* scale,
                                  float* dgamma, float* dbeta, int dtype, tcct_stream_t stream);
/* ---- nn.MaxPool2d(2) (nets/tcct.py:867,883); even H, W ---------------------------------------------------- */
int tcct_maxpool2_fwd(const void* x, void* y, int N, int H, int W, int C, int dtype, tcct_stream_t stream);
int tcct_maxpool2_bwd(const void* x, const void* dy, void* dx, int N, int H, int W, int C, int dtype, tcct_stream_t stream);
/* dx = maxpool_bwd(dy) + res: res [N,H,W,C] is the gradient reaching x through its other consumers (c_i also feeds tran_cnn / the
 * decoder skip, tcct.py:1012-1031), folded into the scatter pass instead of a separate accumulation pass */
int tcct_maxpool2_bwd_add(const void* x, const void* dy, const void* res, void* dx, int N, int H, int W, int C, int dtype,
                          tcct_stream_t stream);
/* ---- bilinear resize: nn.Upsample(x2, align_corners=True) (tcct.py:890) and F.interpolate(size, align_corners=False)
 * (tcct.py:941,1042-1044).  bwd: dy [N,Ho,Wo,C] -> dx [N,H,W,C], gather form (no atomics) ------------------ */
int tcct_bilinear_fwd(const void* x, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners, int dtype,
                      tcct_stream_t stream);
/* separable form of tcct_bilinear_bwd for narrow fp32 tensors (the 5-class aux logits): a pass along W into `workspace`
 * (N*Ho*W*C floats, caller-allocated), then a pass along H; same result up to fp32 summation order */
int tcct_bilinear_bwd_separable(const float* dy, float* dx, float* workspace, int N, int H, int W, int C, int Ho, int Wo,
                                int align_corners, tcct_stream_t stream);
/* y = resize(x) + res (res, y [N,Ho,Wo,C]): upsampling with the decoder's skip-connection add folded in (nets/tcct.py:908-912) */
int tcct_bilinear_add_fwd(const void* x, const void* res, void* y, int N, int H, int W, int C, int Ho, int Wo, int align_corners,
                          int dtype, tcct_stream_t stream);
/* kernel A/B of tcct_bilinear_bwd (tools/bilinear_bwd_bench.py, tests): 0 = exact x2 resizes (align_corners = 0) take the tiled gather kernel like every other scale instead of the
 * separable lane-exchange kernel; returns the previous value.  Test / measurement only. */
int64_t tcct_bilinear_bwd_x2(int on);
int tcct_bilinear_bwd(const void* dy, void* dx, int N, int H, int W, int C, int Ho, int Wo, int align_corners, int dtype,
                      tcct_stream_t stream);
/* ---- F.normalize(dim=channel, p=2, eps) (nets/tcct.py:940) ------------------------------------------------ */
int tcct_l2norm_fwd(const void* x, void* y, int64_t M, int C, float eps, int dtype, tcct_stream_t stream);
int tcct_l2norm_bwd(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, int dtype, tcct_stream_t stream);
int tcct_l2norm_bwd_scaled(const void* x, const void* dy, void* dx, int64_t M, int C, float eps, float scale, int dtype,
                           tcct_stream_t stream);
/* the same + res: the gradient that reaches x through its OTHER consumer (the inputs of norm_add, nets/tcct.py:937-942, also feed the aux heads
 * :1035-1040) is added in this pass instead of by autograd's accumulation add */
int tcct_l2norm_bwd_scaled_add(const void* x, const void* dy, const void* res, void* dx, int64_t M, int C, float eps, float scale, int dtype,
                               tcct_stream_t stream);
/* norm_add (nets/tcct.py:937-942), the `feats` side output of FTC.forward: out = (l2n(g0) + resize(l2n(g1)) + resize(l2n(g2))) / 3 with
 * F.interpolate(bilinear, align_corners=False) to g0's size, in one pass over g0 / out.  g0 [N,H,W,C], g1 [N,h1,w1,C], g2 [N,h2,w2,C];
 * inv1 / inv2: fp32 workspaces [N*h1*w1] / [N*h2*w2] (the coarse maps' inverse norms, written by a small pre-pass) */
int tcct_normadd_fwd(const void* g0, const void* g1, const void* g2, float* inv1, float* inv2, void* out, int N, int H, int W, int C,
                     int h1, int w1, int h2, int w2, float eps, int dtype, tcct_stream_t stream);

/* ---- MultiLoss(DiceLoss) (kite/losses/loss.py:83-99,15-32): softmax over C fused with the batch-global sums
 * sums[3][C] = {sum p*g, sum p, sum g}; loss = sum_c 1 - (1+2I)/(1+P+G).  labels: class index uint8 [M] ----- */
int tcct_softmax_dice_fwd(const void* logits, const uint8_t* labels, int64_t M, int C, double* sums, float* loss,
                          int dtype, tcct_stream_t stream);
/* dlogits = grad_scale * (*grad_out) * dLoss/dlogits  (grad_out: device scalar, NULL -> 1) */
int tcct_softmax_dice_bwd(const void* logits, const uint8_t* labels, int64_t M, int C, const double* sums,
                          const float* grad_out, float grad_scale, void* dlogits, int dtype, tcct_stream_t stream);
/* Deep-supervision heads (FTC.forward, nets/tcct.py:1042-1044 + MultiLoss, kite/losses/loss.py:83-99) without the full-size logits:
 * low fp32 NHWC [B,h,w,C] is resized to H x W (F.interpolate bilinear, align_corners=False, integer scale 2, 4, 8 or 16) on the fly inside the
 * softmax-Dice kernels.  fwd: sums[3][C] + loss as tcct_softmax_dice_fwd.  bwd: dlow [B,h,w,C] = grad_scale * (*grad_out) * dLoss/dlow,
 * ws = fp32 workspace [B,H,w,C] */
int tcct_updice_fwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, double* sums, float* loss,
                    tcct_stream_t stream);
int tcct_updice_bwd(const float* low, const uint8_t* labels, int B, int h, int w, int H, int W, int C, const double* sums,
                    const float* grad_out, float grad_scale, float* ws, float* dlow, tcct_stream_t stream);
/* the whole deep-supervision criterion of KiteSeg.grad_calc (kite/loopback.py:62-73) in one launch sequence: loss = sum_{i=3,2,1} coff * Dice(resize(low_i)) +
 * Dice(logits), fp32 scalar arithmetic in the reference's order; one memset, one finalisation.  low_i fp32 [B,h_i,w_i,C], nullable from the back; sums fp64 [4*3C]
 * (head i at i*3C: pass the slices to tcct_softmax_dice_bwd / tcct_updice_bwd with grad_scale = coff). */
int tcct_dice_ds_fwd(const void* logits, int dtype, const uint8_t* labels, int B, int H, int W, int C, const float* low1, int h1, int w1, const float* low2,
                     int h2, int w2, const float* low3, int h3, int w3, float coff, double* sums, float* loss, tcct_stream_t stream);
/* softmax prob of the labelled class (regular_udh sort key, nets/reg.py:89) and/or argmax class (KiteSeg.predict,
 * kite/loop_seg.py:32); either output may be NULL */
int tcct_softmax_pick(const void* logits, const uint8_t* labels, int64_t M, int C, float* prob_lab, uint8_t* argmax,
                      int dtype, tcct_stream_t stream);
/* out[N][C][3] = per-sample {|pred&lab|, |pred|, |lab|} per class (MDiceLoss/MIouLoss.score, kite/losses/miou.py:28-91) */
int tcct_confusion_counts(const uint8_t* pred, const uint8_t* labels, int N, int64_t HW, int C, float* out,
                          tcct_stream_t stream);

/* GateFusion in training mode (nets/tcct.py:916-932, gtc_* variants): y = x1*alpha + x2*(1-alpha) with
 * alpha = clamp(bicubic_upsample(field -> H x W), 0, 1), field fp32 NHWC [N,hs,ws,C] = the caller's torch.rand draw; torch's bicubic
 * (align_corners=False, A = -0.75, clamped border) is evaluated on the fly.  bwd: dx1 = dy*alpha, dx2 = dy*(1-alpha). */
int tcct_gate_fusion_fwd(const void* x1, const void* x2, const float* field, void* y, int N, int H, int W, int C, int hs, int ws,
                         int dtype, tcct_stream_t stream);
int tcct_gate_fusion_bwd(const void* dy, const float* field, void* dx1, void* dx2, int N, int H, int W, int C, int hs, int ws,
                         int dtype, tcct_stream_t stream);

/* layer-boundary coordinates of a class-index mask [N,H,W] (KiteSeg.predict output): out int32 [N][C-1][W],
 * out[n][k-1][w] = number of rows h with mask[n,h,w] < k = the row where layer k starts in column w of a layered segmentation
 * (the boundary-coordinate extractor the reference lacks -- it only carries the unused soft_argmax, nets/reg.py:27-35) */
int tcct_mask_boundaries(const uint8_t* mask, int32_t* out, int N, int H, int W, int C, tcct_stream_t stream);

/* ---- boundary-regression loss pieces (RegNet.regular_reg, nets/reg.py:109-156); this pipeline is fp32 -------- */
int tcct_slice_channels_fwd(const void* x, float* y, int64_t M, int C, int start, int n, int dtype, tcct_stream_t stream);
int tcct_slice_channels_bwd(const float* dy, void* dx, int64_t M, int C, int start, int n, int dtype, tcct_stream_t stream);
/* onehot [M,n] fp32 of classes start..start+n-1 (NULL to skip) and edge[N,H,W] = clamp1(sum_c |onehot[h]-onehot[h-1]|) */
int tcct_label_planes(const uint8_t* labels, float* onehot, float* edge, int N, int H, int W, int start, int n,
                      tcct_stream_t stream);
/* sampling_softmax summed over channels (reg.py:118-128): x, eps fp32 [N,H,W,CH]; out [N,H,W]; stats [N,W,CH,3] */
int tcct_gumbel_colsoftmax_fwd(const float* x, const float* eps, float* out, float* stats, int N, int H, int W, int CH,
                               tcct_stream_t stream);
int tcct_gumbel_colsoftmax_bwd(const float* x, const float* eps, const float* stats, const float* dout, float* dx, int N,
                               int H, int W, int CH, tcct_stream_t stream);
/* softmax over H of fp32 [N,H,W] (reg.py:155) */
int tcct_colsoftmax_fwd(const float* x, float* y, int N, int H, int W, tcct_stream_t stream);
int tcct_colsoftmax_bwd(const float* y, const float* dy, float* dx, int N, int H, int W, tcct_stream_t stream);
/* column soft-argmax edge[n,w] = sum_h x[n,h,w]*wts[h] (reg.py:146-150) */
int tcct_colwsum_fwd(const float* x, const float* wts, float* out, int N, int H, int W, tcct_stream_t stream);
int tcct_colwsum_bwd(const float* dout, const float* wts, float* dx, int N, int H, int W, tcct_stream_t stream);
/* nn.MSELoss (reg.py:108,154-155): out = mean((a-b)^2); bwd da = 2/n*(a-b)*g, db = -da (either may be NULL) */
int tcct_mse_fwd(const float* a, const float* b, int64_t n, double* acc, float* out, tcct_stream_t stream);
int tcct_mse_bwd(const float* a, const float* b, int64_t n, const float* grad_out, float grad_scale, float* da, float* db,
                 tcct_stream_t stream);

/* ---- feature-polarization loss (RegNet.regular_udh nets/reg.py:86-105; points_selection_bins nets/fcs.py:25-50;
 * cosinesim fcs.py:63-80; FeatConPolar.choice nets/fcp.py:72-75) ------------------------------------------- */
/* Bin assignment without a sort (round 3): radix multi-select of the 32 bin boundaries per class on the unique key (~prob, pixel index) --
 * the order of a stable descending sort by probability -- then bin map + bin sums in ONE coalesced pass over the feature rows (reference nets/fcs.py:25-50).
 * counts [16] uint32, binmap [M] (255 = dropped), pro_sum [C][32][32] fp32 (cleared here); workspace: tcct_fpl_select_workspace_bytes(). */
int64_t tcct_fpl_select_workspace_bytes();
int tcct_fpl_select(const void* feat, const uint8_t* labels, const float* prob, int64_t M, int C, void* workspace, uint32_t* counts,
                    uint8_t* binmap, float* pro_sum, int dtype, tcct_stream_t stream);
/* prototypes, loss and d loss / d prototype (already divided by the bin size) from the bin sums; pro_sum/pro/dpro_over_n fp32 [C,32,32] */
int tcct_fpl_loss(const float* pro_sum, const uint32_t* counts, const float* buf_grad, int C, float* pro, float* loss, float* dpro_over_n,
                  tcct_stream_t stream);
/* binmap uint8 [M] (bin 0..31, 255 = not selected); dfeat [M,32] */
int tcct_fpl_backward(const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out,
                      float grad_scale, int64_t M, void* dfeat, int dtype, tcct_stream_t stream);
/* norm_add's backward when its gradient is the feature-polarization loss's (round 4, csrc/pool_resize.hip): d loss / d feats is a function of two
 * bytes per pixel (label, bin) and the [classes][32][32] table dpro_over_n, so it is looked up instead of being written by tcct_fpl_backward and read
 * three times.  l2norm: dx = scale * l2norm_bwd(x, dfeat) (+ res); bilinear: dx [N,H,W,32] = resize_bwd(dfeat [N,Ho,Wo,32]); 32 feature channels. */
int tcct_l2norm_bwd_fplgrad(const void* x, const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out,
                            float grad_scale, int ncls, const void* res, void* dx, int64_t M, float eps, float scale, int dtype, tcct_stream_t stream);
int tcct_bilinear_bwd_fplgrad(const uint8_t* labels, const uint8_t* binmap, const float* dpro_over_n, const float* grad_out, float grad_scale, int ncls,
                              void* dx, int N, int H, int W, int Ho, int Wo, int align_corners, int dtype, tcct_stream_t stream);

/* ---- factorised attention with convolutional relative position encoding: FactorAtt_ConvRelPosEnc.forward
 * nets/tcct.py:311-341 and ConvRelPosEnc.forward nets/tcct.py:265-287, the token mixer the reference keeps commented out in MHCABlock
 * (tcct.py:436-449).  qkv [B,N,3C] is the output of the qkv Linear (tcct.py:316-318: channel = which*C + head*Ch + ch, Ch = C/heads a
 * multiple of 4, C <= 256); M, dM fp32 [B,heads,Ch,Ch]; stats fp32 [B,C,2] = (max_n k, 1 / sum_n exp(k - max)); cv [B,N,C] = the crpe
 * depthwise convolutions of v.  The per-head contractions are Ch x Ch (8..20): VALU with M in LDS; qkv / proj are tcct_pw_* GEMMs. */
int64_t tcct_fatt_kstats_workspace_bytes(int B, int64_t N, int C);
/* k.softmax(dim=2) statistics (tcct.py:321) */
int tcct_fatt_kstats(const void* qkv, void* workspace, float* stats, int B, int64_t N, int C, int heads, int dtype, tcct_stream_t stream);
/* M = softmax_N(k)^T v (tcct.py:322); accumulation output: cleared here unless tcct_set_outputs_prezeroed(1) */
int tcct_fatt_ktv(const void* qkv, const float* stats, float* M, int B, int64_t N, int C, int heads, int dtype, tcct_stream_t stream);
/* dM = scale * q^T dout (gradient of M; same clearing rule) */
int tcct_fatt_dktv(const void* qkv, const void* dout, float* dM, float scale, int B, int64_t N, int C, int heads, int dtype,
                   tcct_stream_t stream);
/* out [B,N,C] = scale * q M + q * cv (tcct.py:323-331, 284-285; already in the transpose(1,2).reshape(B,N,C) layout) */
int tcct_fatt_apply_fwd(const void* qkv, const float* M, const void* cv, void* out, float scale, int B, int64_t N, int C, int heads,
                        int dtype, tcct_stream_t stream);
/* dqkv [B,N,3C] = (scale dout M^T + dout cv | P (v dM^T - D) | P dM), dcv [B,N,C] = dout * q; the crpe convolution's share of dv is added by
 * tcct_dwk_strided_fwd(flip=1, accumulate=1) on dcv afterwards */
int tcct_fatt_apply_bwd(const void* qkv, const float* stats, const float* M, const float* dM, const void* cv, const void* dout,
                        void* dqkv, void* dcv, float scale, int B, int64_t N, int C, int heads, int dtype, tcct_stream_t stream);
/* depthwise K x K (K in 3,5,7), stride 1, 'same' zero padding on one channel group of NHWC rows (ConvRelPosEnc.conv_list, tcct.py:247-262):
 * x / y point at the group's first channel, pixel strides ldx / ldy in elements (multiples of 4, like Cg); w fp32 OIHW [Cg,1,K,K]; bias fp32
 * [Cg] or NULL; flip = taps rotated by 180 degrees (the input gradient); accumulate = add to y */
int tcct_dwk_strided_fwd(const void* x, int64_t ldx, const float* w, const float* bias, void* y, int64_t ldy, int B, int H, int W, int Cg,
                         int K, int flip, int accumulate, int dtype, tcct_stream_t stream);
/* dw [Cg,1,K,K] = sum dy * shifted x, dbias [Cg] = sum dy (nullable); accumulation outputs */
int tcct_dwk_strided_wgrad(const void* x, int64_t ldx, const void* dy, int64_t ldy, float* dw, float* dbias, int B, int H, int W, int Cg,
                           int K, int dtype, tcct_stream_t stream);

/* ---- clip_grad_norm_(12) + AdamW on flat fp32 buffers (kite/loop_seg.py:128-130, kite/loopback.py:127) ------ */
int tcct_grad_sumsq(const float* g, int64_t n, double* acc, tcct_stream_t stream);
/* grad_mul pre-scales the raw gradient (1/world_size after a sum all-reduce); total_norm_out nullable */
int tcct_clip_adamw(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm,
                    float grad_mul, float lr, double beta1, double beta2, float eps, float weight_decay, int step,
                    float* total_norm_out, tcct_stream_t stream);

/* the same step with its two per-step scalars in device memory -- state[0] = learning rate (host-written), state[1] = steps taken so far
 * (float, incremented here) -- so that the launch arguments are constant and the whole training step can be replayed from a hipGraph */
int tcct_clip_adamw_dev(float* p, const float* g, float* m, float* v, int64_t n, const double* sumsq, float max_norm,
                        float grad_mul, float* state, double beta1, double beta2, float eps, float weight_decay,
                        float* total_norm_out, tcct_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif
