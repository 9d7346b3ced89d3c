/*
 * scan_hip.h -- C ABI of libscan_hip.so, the MI355X (gfx950) hot path of the
 * SCAN FCOS-based domain-adaptive detector.
 *
 * Every entry point takes plain device pointers + sizes and a HIP stream
 * (void* == hipStream_t; NULL = the null stream).  Nothing here allocates,
 * frees or synchronises unless stated, so calls are graph-capturable.
 * Return value: 0 on success, <0 on error (scan_last_error() gives the text;
 * the Python host raises RuntimeError, mirroring the reference's AT_ASSERTM /
 * AT_ERROR -> RuntimeError behaviour, csrc/nms.h:10-28, csrc/SigmoidFocalLoss.h:10-41).
 *
 * Layout convention ("pyramid"): an activation is a row-major fp32 matrix
 * [M, C] whose rows are pixels in level-major, then image, then y, then x
 * order -- exactly the order the reference flattens to before its losses
 * (rpn/fcos/loss.py:191-202, condgraph.py:348-351).  A scan_pyramid_t says how
 * the rows split into levels.  A plain NHWC tensor is a pyramid with 1 level.
 *
 * Each function cites the reference interface it replaces
 * (paths relative to the reference checkout, fcos_core/...).
 */
#ifndef SCAN_HIP_H
#define SCAN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SCAN_MAX_LEVELS 5

typedef struct {
  int32_t n_levels;              /* 1..5 */
  int32_t n_images;              /* N */
  int32_t h[SCAN_MAX_LEVELS];    /* per-level height */
  int32_t w[SCAN_MAX_LEVELS];    /* per-level width  */
  int64_t row_off[SCAN_MAX_LEVELS + 1]; /* first row of each level; row_off[n_levels] = M */
} scan_pyramid_t;

const char* scan_last_error(void);
int scan_abi_version(void);

/* Tuning knob for A/B measurements and tests (no reference counterpart): scan_tune(key, value) sets an integer
 * launch-selection parameter and returns its previous value, SCAN_TUNE_UNKNOWN for an unknown key.  Every setting of every
 * key gives correct results (timing-ablation instances are not part of this library).
 *   "conv_bn256"  3x3 convs whose output channels are a multiple of 256 -- 2 (default): 256-channel tiles whenever they do
 *                 not cost a round of 256 CUs against 128-channel tiles (a 256-channel workgroup runs twice as long); 1: when
 *                 the launch keeps >= 2 workgroups per CU (rounds 2-4); 0: always 128-channel tiles.  Same results bit for bit.
 *   "conv_v2"     1 (default): bf16x3 forward / data-gradient convs run on the v_mfma_f32_16x16x32_bf16 kernel
 *                 (csrc/conv_fwd.hip); 0: on the independent v_mfma_f32_32x32x16_bf16 kernel kept for cross-checks
 *                 (csrc/conv_bf16x3.hip; same arithmetic, different summation order inside a 32-channel chunk).
 *   "conv_wg1024" (bf16x3) 1 (default): the 128- / 256-channel forward / dgrad instances run 16 waves per workgroup; 0: 8
 *                 waves; 2: 16 waves only for the 256-channel tile on multi-level pyramids.  Same results bit for bit.
 *   "conv_w8"     (bf16x3) 1 (default): the 256-channel LDS-DMA forward / dgrad tile runs on 8 waves (64 px x 128 ch per
 *                 wave); 0: 16 waves.  Same results bit for bit.
 *   "conv_tpb3"   (bf16x3) bit 0 (default on) / bit 1: the 128- / 64-channel instance stages three taps per barrier (same
 *                 results bit for bit).
 *   "conv_bn64_th16" pixel tile of convs with <= 64 output channels on single-level pyramids: 0: 8x16; 1: 16x16 when the
 *                 sizes are multiples of 16; 2 (default): bf16x6 additionally takes 32x16 when H is a multiple of 32.  Same
 *                 results bit for bit.
 *   "conv_glds"   1 (default): the 128- / 256-channel 3x3 instances stage weight tiles by LDS-DMA (buffer_load ... lds) on
 *                 whole tiles; 0: through registers.  Same results bit for bit.
 *   "conv1x1"     bf16x6 1x1 convs (FPN laterals, ResNet bottlenecks): bit 0 = weight tiles by LDS-DMA, bit 1 = the 256-channel tile
 *                 when the output channels fill it without costing a round; 3 (default), 0 = rounds 4-5's 128-channel register-staged
 *                 instance.  Same results bit for bit.
 *   "wgrad_v6"    1 (default): the 3x3 weight-gradient launches take the producer / consumer kernel (12 waves: 8 issue
 *                 MFMAs, 4 stage); 0: the kernel in which all 8 waves stage and multiply in turn (always used by the 1x1
 *                 convs).  bf16x3: bit-identical; bf16x6: other K-chunk length, i.e. other split-K boundaries.
 *   "wgrad_prio"  1: the producer waves of that kernel run at s_setprio 3; 0 (default): at the consumers' priority.  Same
 *                 results bit for bit.
 *   "wgrad_tile"  consumer wave tile of that kernel: 0 = 64 (o) x 32 (c), 1 = 32 (o) x 64 (c), 2 (default, and any other value) = 1
 *                 for bf16x6, 0 for bf16x3.  bf16x3: same results bit for bit.  bf16x6: 1 sums the six piece products of a
 *                 32-pixel step in a temporary first (one rounding per step at the accumulator's magnitude, DESIGN.md 3.0:
 *                 0.23-0.5x the fp32-MFMA kernel's distance from fp64); 0 adds them straight into the running accumulator (an
 *                 fp32 sum in another order, 1.0-2.1x that distance; the golden suite is green on it, 4-5 % faster on random
 *                 operands, no faster in the training step: profiles/r06_wgrad_tile_ab.txt).
 *   "wgrad_wgs"   768 (default): workgroups a weight-gradient launch aims at (tiles x split-K slabs); 512 / 640 / 896 / 1024
 *                 / 1280 / 1536 are 2...25 % slower on the 256- and 512-channel layers (bf16x3).
 *   "gconv_mfma"  0 (default): the grouped class-branch conv runs on fp32 FMAs; 1: tap products and data gradient on the
 *                 fp32 matrix cores (same products, different summation order; measured no faster in the step).
 *   "reduce_blocks" 2048 (default): most workgroups a loss-reduction kernel is launched with (each ends in one or two float
 *                 atomics on one cache line); the IoU and CKA forward kernels take half of it.  Same sums up to the order of
 *                 the atomics.
 *   "dbscan_bf16x3" 1 (default): scan_dbscan_prepare's pairwise-distance GEMM runs as bf16x3 with a wider exact re-check
 *                 band; 0: exact fp32 matrix cores.  Same neighbour bits (pairs inside the band are decided in fp64). */
#define SCAN_TUNE_UNKNOWN (-2147483647 - 1)
int scan_tune(const char* key, int value);
/* read-only: the current value of a knob (nothing is written), SCAN_TUNE_UNKNOWN for an unknown key */
int scan_tune_get(const char* key);
/* the value the library was built with; scan_tune_key(i): name of the i-th knob, NULL past the end (bench.py stamps every
 * knob that differs from its default into the line it prints) */
int scan_tune_default(const char* key);
const char* scan_tune_key(int index);

/* Measurement (no reference counterpart): the bf16 matrix-pipe rate this board sustains under its package power cap -- a
 * register-only v_mfma_f32_16x16x32_bf16 loop on every CU for about `seconds` (<= 30), two waves per SIMD; random != 0:
 * random-sign / random-mantissa operands, 0: zeros.  Writes TFLOP/s; blocking (synchronises `stream`).  bench.py reports
 * it as roofline.board_sustained beside the fraction of the nominal 2.5 PFLOP/s. */
int scan_mfma_sustained_bf16(double seconds, int32_t random, double* tflops, void* stream);

/* Measurement (no reference counterpart; what it stands in for: the RCCL ring all-reduce the data-parallel step issues where the
 * reference's DistributedDataParallel does, tools/train_net_da.py:421-515): one rank's footprint of an all-reduce of
 * buf[0, n_floats) -- `wgs` workgroups of 256 threads (RCCL: one per channel) that read-modify-write (x * 1.0f: values
 * unchanged) their slice `traffic` times (ring: 2 (N - 1) / N), paced to `gbps` aggregate read rate by sleeping on the wall
 * clock (0 = unpaced).  Asynchronous on `stream`.  tools/dp_emulate.py runs it where the gradient buckets fire. */
int scan_comm_standin(float* buf, int64_t n_floats, double traffic, int32_t wgs, double gbps, void* stream);

/* Output-channel tile (64, 128 or 256) the bf16x3 3x3 kernel uses for a launch on pyramid d with Nout channels;
 * 1128 / 1256 = the 128- / 256-channel tile on 16-wave workgroups, 2256 = the 256-channel tile on the 8-wave LDS-DMA instance. */
int scan_conv3x3_bf16x3_instance(const scan_pyramid_t* d, int32_t Nout);
/* the same for a bf16x6 1x1 launch on output pyramid yd with Nout channels and weight-plane row length Csw: 64 / 128 = register-staged
 * tiles, 1128 / 1256 = the 128- / 256-channel tile with LDS-DMA weight tiles (scan_tune "conv1x1") */
int scan_conv1x1_bf16x6_instance(const scan_pyramid_t* yd, int32_t Nout, int32_t Csw);

/* ---- SigmoidFocalLoss  (replaces _C.sigmoid_focalloss_forward / _backward,
 *      csrc/SigmoidFocalLoss.h:10-41, csrc/cuda/SigmoidFocalLoss_cuda.cu:20-187) ----
 * logits [M,C] fp32, targets [M] int32 (0 = bg, c = class c, <0 = ignore).
 * losses may be NULL; loss_sum (1 float, pre-zeroed by caller) may be NULL:
 * when given, the wavefront-reduced total is atomically added to it, which is
 * what layers/sigmoid_focal_loss.py:56-70 (`loss.sum()`) consumes. */
int scan_sigmoid_focal_loss_forward(const float* logits, const int32_t* targets, int64_t M, int32_t C,
                                    float gamma, float alpha, float* losses, float* loss_sum, void* stream);
/* d_losses [M,C] or NULL; when NULL every element uses d_scale (the fused
 * `sum()/(num_pos+N)` backward, rpn/fcos/loss.py:204-208). */
int scan_sigmoid_focal_loss_backward(const float* logits, const int32_t* targets, const float* d_losses,
                                     float d_scale, int64_t M, int32_t C, float gamma, float alpha,
                                     float* d_logits, void* stream);

/* ---- IOULoss (replaces layers/iou_loss.py:5-36) ----
 * pred/target [P,4] (l,t,r,b), weight [P] or NULL.  out[0] += sum(loss*w),
 * out[1] += sum(w) (w = 1 when NULL); caller pre-zeroes out[2] and divides. */
int scan_iou_loss_forward(const float* pred, const float* target, const float* weight, int64_t P,
                          float* out2, void* stream);
/* d_pred[i,:] = g_num * w_i * dloss_i/dpred  with g_num = upstream / sum(w) read from g_num_dev[0]. */
int scan_iou_loss_backward(const float* pred, const float* target, const float* weight, int64_t P,
                           const float* g_num_dev, float* d_pred, void* stream);

/* ---- BCE-with-logits, optionally weighted (replaces F.binary_cross_entropy_with_logits as used by
 *      discriminator/fcos_head_discriminator_con.py:117-123 and rpn/fcos/loss.py:221-224) ----
 * logits [M] (stride 1), weight element i at weight[i*w_stride] or NULL, constant target.
 * out[0] += sum(w * bce), out[1] += sum(w). */
int scan_bce_logits_forward(const float* logits, const float* targets, float const_target,
                            const float* weight, int64_t w_stride, int64_t M, float* out2, void* stream);
/* d_logits[i] = g_dev[0] * w_i * (sigmoid(x_i) - t_i) */
int scan_bce_logits_backward(const float* logits, const float* targets, float const_target,
                             const float* weight, int64_t w_stride, int64_t M, const float* g_dev,
                             float* d_logits, void* stream);

/* ---- CKA class-conditional adversarial loss (replaces the per-class loop of
 *      discriminator/fcos_head_discriminator_con.py:105-124, num_classes > 1 branch) ----
 * logits [M,Cf] (one column per foreground class), act [M,Cf+1] softmax maps (column 0 = background),
 * constant domain target t.  out[2*c] += sum_m act[m][c+1]*bce(logits[m][c], t), out[2*c+1] += sum_m act[m][c+1].
 * The host forms  sum_c (out[2c]/out[2c+1]) / Cf.   Cf <= 16. */
int scan_cka_bce_forward(const float* logits, const float* act, int64_t M, int32_t Cf, float target, float* out,
                         void* stream);
/* d_logits[m][c] = g_dev[c] * act[m][c+1] * (sigmoid(x) - t),  g_dev[c] = upstream / (Cf * sum_w[c]) */
int scan_cka_bce_backward(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                          const float* g_dev, float* d_logits, void* stream);
/* The same pair with the layer's scalar arithmetic folded in: out = 2 * Cf + 2 floats zeroed by the caller (sums, a ticket
 * word, the loss  sum_c (num_c / den_c) / Cf  written by the block that finishes last); the backward takes the gradient
 * of that scalar (g_loss [1], device) and the forward's out and forms the per-class coefficients itself. */
int scan_cka_bce_forward_loss(const float* logits, const float* act, int64_t M, int32_t Cf, float target, float* out,
                              void* stream);
int scan_cka_bce_backward_loss(const float* logits, const float* act, int64_t M, int32_t Cf, float target,
                               const float* g_loss, const float* sums, float* d_logits, void* stream);

/* ---- y = alpha * x  (GradientReversalFunction: forward alpha = 1 copy, backward alpha = -lambda;
 *      discriminator/layer.py:6-24) ---- */
int scan_scale(const float* x, float alpha, float* y, int64_t n, void* stream);
/* dst[r][c] = src[r][c] for c < ncols, zeros for ncols <= c < ncols + ztail; row pitches ld_src / ld_dst in floats (callers
 * offset the pointers to the first column).  Replaces the torch-tier spellings of F.pad(act_maps, (0, 3)) in front of head_out's
 * act-map share (rpn/fcos/condgraph.py:344-362: cat(features, act_maps)) and of cat(x, act_maps[:, 1:]) in front of the class
 * branches (discriminator/fcos_head_discriminator_con.py:104-118). */
int scan_copy_cols(const float* src, int32_t ld_src, float* dst, int32_t ld_dst, int64_t M, int32_t ncols, int32_t ztail,
                   void* stream);
/* GRAPHModule.update_prototype_nx1_rnn with COSINE_UPDATE_ON (rpn/fcos/condgraph.py:586-606) on the paradigm buffer P [K][C][T]
 * in place: slot = it - 1 if it == T else it; m = cosine_similarity(P[:, :, slot], pb); P[:, :, slot] <- cur * m + pb * (1 - m) for the
 * classes whose batch mean pb [K][C] is not all zero; it == T shifts the slots down by one first.  C <= 1024. */
int scan_paradigm_update(float* P, const float* pb, int32_t K, int32_t C, int32_t T, int32_t it, void* stream);

/* ---- semantic-conditioned dynamic 1x1 conv + channel softmax
 *      (replaces GRAPHModule.dynamic_conv + softmax, rpn/fcos/condgraph.py:619-629, 344-346) ----
 * feat [M,C] (C % 4 == 0), kernels [K,C] (K <= 16) -> logits [M,K], probs [M,K]. */
int scan_dynconv_softmax_forward(const float* feat, const float* kernels, int64_t M, int32_t C, int32_t K,
                                 float* logits, float* probs, void* stream);
/* d_logits_in [M,K] or NULL (grad arriving at the logits, e.g. from the act loss), d_probs [M,K] or NULL.
 * Writes d_feat [M,C] (overwrites) and d_kernels [K,C] (overwrites; uses ws of size >= grid*K*C floats,
 * see scan_dynconv_ws_floats). */
int64_t scan_dynconv_ws_floats(int64_t M, int32_t C, int32_t K);
int scan_dynconv_softmax_backward(const float* feat, const float* kernels, const float* probs,
                                  const float* d_logits_in, const float* d_probs, int64_t M, int32_t C,
                                  int32_t K, float* d_feat, float* d_kernels, float* ws, void* stream);

/* ---- softmax focal loss of the activation maps (replaces layers/sigmoid_focal_loss_wbg.py:38-64,
 *      alpha = 1, gamma = 2, mean over rows) ----
 * logits [M,K], labels int64 [M]; loss_sum[0] += sum_i -(1-p_i)^g log p_i  (caller divides by M). */
int scan_softmax_focal_forward(const float* logits, const int64_t* labels, int64_t M, int32_t K, float gamma,
                               float* loss_sum, void* stream);
int scan_softmax_focal_backward(const float* logits, const int64_t* labels, int64_t M, int32_t K, float gamma,
                                float d_scale, float* d_logits, void* stream);

/* ---- NMS (replaces _C.nms, csrc/nms.h:10-28, csrc/cpu/nms_cpu.cpp:5-65, csrc/cuda/nms.cu:23-131;
 *      and _C.ml_nms, csrc/ml_nms.h:10-27, csrc/cuda/ml_nms.cu:13-136) ----
 * dets [n,4] xyxy, scores [n], labels [n] float or NULL (NULL = plain nms).
 * rule_ge != 0: suppress when IoU >= thr (the CPU rule, nms_cpu.cpp:60 -- the oracle);
 * rule_ge == 0: suppress when IoU >  thr (the CUDA rule, nms.cu:60).
 * keep_out [n] int64 receives the kept ORIGINAL indices ascending, num_keep_out[0] their count
 * (both device memory).  workspace: scan_nms_ws_bytes(n) bytes (n <= SCAN_NMS_PANEL: ~n*n/8 + 64 KB; the mask of a
 * larger n is n * ceil(n/64) * 8 bytes, the size of the reference's own, cuda/nms.cu:95-100).
 * n <= SCAN_NMS_PANEL runs as three single-workgroup-chain launches (the post-processor's case: <= 5,000 candidates per
 * image); a larger n (nms_cuda has no limit, cuda/nms.cu:70-131) takes the panel path, up to SCAN_NMS_MAX -- a bound on
 * the chain state one workgroup keeps in LDS (two bit words per 64 candidates), not on the algorithm. */
#define SCAN_NMS_PANEL 8192
#define SCAN_NMS_MAX 262144
int64_t scan_nms_ws_bytes(int64_t n);
int scan_nms(const float* dets, const float* scores, const float* labels, int64_t n, float thr, int32_t rule_ge,
             int64_t* keep_out, int32_t* num_keep_out, void* workspace, void* stream);

/* ---- convolution as fp32-MFMA implicit GEMM (replaces nn.Conv2d / F.conv2d at the call sites of
 *      backbone/mmdetection/vgg.py:8-33, backbone/fpn.py:52-66,118-130,
 *      rpn/fcos/condgraph.py:86-106, rpn/fcos/fcos.py:25-64, discriminator/fcos_head_discriminator_con.py:20-62) ----
 * x: pyramid [Mi, Cin_s] (row stride Cin_s floats, Cin_s % 4 == 0, channels >= Cin are zero padding)
 * w: [Cout][k*k][Cin_s] fp32 ("OHWI", what a channels_last torch weight is physically)
 * y: pyramid [Mo, Cout_s]; bias [Cout] or NULL; relu != 0 fuses max(0,.).
 * ksize in {1,3,5,7}; stride in {1,2}; pad = ksize/2. */
int scan_conv2d_forward(const float* x, const scan_pyramid_t* xd, int32_t Cin_s, const float* w, const float* bias,
                        float* y, const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t ksize,
                        int32_t stride, int32_t relu, void* stream);
/* dX = conv_transpose(dY, W).  wt: [Cin][k*k][Cout_s] (the transposed copy made by scan_weight_transpose).
 * mask (optional, same shape as dx): dx is zeroed where mask <= 0 (ReLU backward of the producer). */
int scan_conv2d_dgrad(const float* dy, const scan_pyramid_t* yd, int32_t Cout_s, const float* wt, float* dx,
                      const scan_pyramid_t* xd, int32_t Cin, int32_t Cin_s, int32_t ksize, int32_t stride,
                      const float* mask, void* stream);
/* dW[Cout][k*k][Cin_s] = sum_m dY[m][o] * X[tap(m)][c]; deterministic split-K through ws
 * (scan_conv2d_wgrad_ws_floats) followed by an in-order reduction.  accumulate != 0: dW += result. */
int64_t scan_conv2d_wgrad_ws_floats(const scan_pyramid_t* yd, int32_t Cin_s, int32_t Cout, int32_t ksize);
int scan_conv2d_wgrad(const float* x, const scan_pyramid_t* xd, int32_t Cin_s, const float* dy,
                      const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t ksize, int32_t stride,
                      float* dw, int32_t accumulate, float* ws, void* stream);
/* ---- 3x3 / stride-1 convolution with fp32-grade accuracy on the bf16 matrix cores ("bf16x3": every fp32
 *      operand is split hi + lo into two bf16 values; hi*hi + hi*lo + lo*hi accumulated in fp32).  Same call
 *      sites as scan_conv2d_forward; ~5x the fp32-MFMA rate.
 * scan_weight_split: w [O][T][Cs] fp32 -> bf16 planes wh, wl.
 *    mode 0: [O][T][Csw]   (forward);   mode 1: [Cs][T][Csw] holding w[o][T-1-t][c] at [c][t][o] (dgrad).
 *    Csw % 8 == 0, zero padded.
 * scan_conv3x3_bf16x3: y[M][Ns] = conv3x3_s1(x[M][Cs], w) + bias, optional ReLU, optional mask (y = 0 where
 *    mask <= 0).  The data gradient is the same call on dY with the mode-1 weights. */
int scan_weight_split(const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* wh, void* wl,
                      int32_t Csw, void* stream);
int scan_conv3x3_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                        int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout, int32_t Ns,
                        int32_t relu, void* stream);
/* 1x1 convolutions (FPN laterals backbone/fpn.py:52-66; ResNet bottleneck conv1 / conv3 / downsample,
 * backbone/resnet.py:228-314, incl. the stride-2 ones: STRIDE_IN_1X1) on the same bf16x3 kernel (one tap, no halo).
 * wh/wl: scan_weight_split planes with T = 1 (mode 0 forward, mode 1 data gradient).  map 0: stride 1 (xd == yd);
 * map 1: stride-2 forward (yd = ceil(xd / 2)); map 2: data gradient of a stride-2 1x1 conv (x = dY on the coarse
 * pyramid xd, y = dX on the fine pyramid yd, zero where a coordinate is odd).  mask as in scan_conv3x3_bf16x3. */
int scan_conv1x1_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wl,
                        int32_t Csw, const float* bias, const float* mask, float* y, const scan_pyramid_t* yd,
                        int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream);
/* weight gradient of a 1x1 conv (stride 1 or 2): dw [Cout][1][Cs] (+)= dY^T X over all pixels of the dY pyramid yd; x on
 * xd (yd = ceil(xd / stride)); db / accumulate / ws as for scan_conv3x3_wgrad_bf16x3. */
int64_t scan_conv1x1_wgrad_bf16x3_ws_floats(const scan_pyramid_t* yd, int32_t Cs, int32_t Cout);
int scan_conv1x1_wgrad_bf16x3(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy,
                              const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t stride, float* dw,
                              float* db, int32_t accumulate, float* ws, void* stream);
/* first-layer convolutions (3 input channels stored as 4; forward only, both live in frozen stages): the ResNet stem
 * 7x7 / stride 2 / pad 3 (backbone/resnet.py:316-336) and VGG conv1_1 3x3 / stride 1 / pad 1
 * (backbone/mmdetection/vgg.py:8-33).  x [N,H,W,4] NHWC rows, w [Cout][k*k][4] fp32, Cout <= 64,
 * y [N,Ho,Wo,Cout_s]; bf16x3 product, optional bias and ReLU. */
int scan_conv_smallcin_bf16x3(const float* x, int32_t N, int32_t H, int32_t W, const float* w, const float* bias,
                              float* y, int32_t Cout, int32_t Cout_s, int32_t ksize, int32_t stride, int32_t relu,
                              void* stream);
/* conv3x3 + bias whose 256-channel output feeds GroupNorm(32, 256) (the [conv, GN, ReLU] towers of condgraph.py:86-106,
 * fcos.py:25-49, fcos_head_discriminator_con.py:20-34): the epilogue also accumulates the GroupNorm sums into gn_ws
 * (8-byte aligned, n_levels * N * 32 * 2 fp64 values, zeroed by the call); scan_groupnorm_stats_from_sums turns them into
 * the (mean, rstd) table scan_groupnorm_stats would have produced. */
int scan_conv3x3_gn_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                           int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                           void* stream);
/* The same with the sums ADDED to gn_ws as it stands: the caller has cleared it (one memset per training iteration over a
 * buffer all such workspaces are slices of, instead of one memset launch per call). */
int scan_conv3x3_gn_acc_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                               int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, float* gn_ws,
                               void* stream);
int scan_groupnorm_stats_from_sums(const float* ws, const scan_pyramid_t* d, int32_t C, int32_t G, float eps,
                                   float* stats, void* stream);
/* conv3x3 + bias (+ ReLU) + nn.MaxPool2d(2, 2) in one launch (last conv of a FROZEN VGG stage, vgg.py:8-33: forward only):
 * single-level pyramid d with even H, W; y [N, H/2, W/2, Ns]. */
int scan_conv3x3_pool2_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wl,
                              int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns, int32_t relu,
                              void* stream);
/* weight gradient of the same conv, bf16x3 on the matrix cores, deterministic split-K through ws
 * (scan_conv3x3_wgrad_bf16x3_ws_floats floats).  dw [Cout][9][Cs]; db [Cout] or NULL = bias gradient (column
 * sums of dy, fused); accumulate != 0: dw += result, db += result. */
int64_t scan_conv3x3_wgrad_bf16x3_ws_floats(const scan_pyramid_t* d, int32_t Cs, int32_t Cout);
int scan_conv3x3_wgrad_bf16x3(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy, int32_t Cout,
                              int32_t Cout_s, float* dw, float* db, int32_t accumulate, float* ws, void* stream);
/* ---- the same convolutions at the reference's arithmetic on the bf16 matrix cores ("bf16x6") ----
 * The reference multiplies fp32 by fp32 and accumulates in fp32 (torch.nn.Conv2d: backbone/mmdetection/vgg.py:8-33,
 * backbone/fpn.py:52-66, rpn/fcos/condgraph.py:86-106, rpn/fcos/fcos.py:25-64,
 * discriminator/fcos_head_discriminator_con.py:20-62).  Here every fp32 operand is cut into THREE bf16 pieces
 * hi + mid + lo -- 8 + 8 + 8 = all 24 significand bits, the split is exact -- and a product is accumulated in fp32 from the
 * six piece products hi*lo, mid*mid, lo*hi, hi*mid, mid*hi, hi*hi (smallest first); the three dropped products are
 * <= 2^-23 of the product, below the rounding of the fp32 accumulation itself.  bf16 x bf16 is exact in fp32, so the result
 * is an fp32 convolution up to summation order: measured against an fp64 convolution it is as close as the exact
 * v_mfma_f32_32x32x2_f32 kernels of scan_conv2d_* (tests/test_gpu_kernels.py::test_conv_error_vs_fp64), on a pipe whose
 * ceiling is 2.5 PFLOP/s / 6 = 417 TFLOP/s fp32-equivalent instead of 157.
 * Arguments as for the _bf16x3 functions with a third plane: wh / wm / wl = scan_weight_split3 planes (same layout and
 * modes as scan_weight_split).  y and mask must be 16-byte aligned, Ns % 4 == 0.
 * scan_conv3x3_gn_bf16x6: clear != 0 zeroes gn_ws first (scan_conv3x3_gn_bf16x3), 0 adds to it (.._gn_acc_bf16x3). */
int scan_weight_split3(const float* w, int32_t O, int32_t T, int32_t Cs, int32_t mode, void* wh, void* wm, void* wl,
                       int32_t Csw, void* stream);
int scan_conv3x3_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wm,
                        const void* wl, int32_t Csw, const float* bias, const float* mask, float* y, int32_t Nout,
                        int32_t Ns, int32_t relu, void* stream);
int scan_conv1x1_bf16x6(const float* x, const scan_pyramid_t* xd, int32_t Cs, const void* wh, const void* wm,
                        const void* wl, int32_t Csw, const float* bias, const float* mask, float* y,
                        const scan_pyramid_t* yd, int32_t Nout, int32_t Ns, int32_t relu, int32_t map, void* stream);
int scan_conv3x3_gn_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wm,
                           const void* wl, int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns,
                           float* gn_ws, int32_t clear, void* stream);
int scan_conv3x3_pool2_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const void* wh, const void* wm,
                              const void* wl, int32_t Csw, const float* bias, float* y, int32_t Nout, int32_t Ns,
                              int32_t relu, void* stream);
int scan_conv_smallcin_bf16x6(const float* x, int32_t N, int32_t H, int32_t W, const float* w, const float* bias,
                              float* y, int32_t Cout, int32_t Cout_s, int32_t ksize, int32_t stride, int32_t relu,
                              void* stream);
/* weight gradients: x and dy are split inside the kernel, no planes; Cout_s % 4 == 0 */
int64_t scan_conv3x3_wgrad_bf16x6_ws_floats(const scan_pyramid_t* d, int32_t Cs, int32_t Cout);
int scan_conv3x3_wgrad_bf16x6(const float* x, const scan_pyramid_t* d, int32_t Cs, const float* dy, int32_t Cout,
                              int32_t Cout_s, float* dw, float* db, int32_t accumulate, float* ws, void* stream);
int64_t scan_conv1x1_wgrad_bf16x6_ws_floats(const scan_pyramid_t* yd, int32_t Cs, int32_t Cout);
int scan_conv1x1_wgrad_bf16x6(const float* x, const scan_pyramid_t* xd, int32_t Cs, const float* dy,
                              const scan_pyramid_t* yd, int32_t Cout, int32_t Cout_s, int32_t stride, float* dw,
                              float* db, int32_t accumulate, float* ws, void* stream);
/* output-channel tile (64 / 128 / 256) a scan_conv3x3_bf16x6 launch on pyramid d with Nout channels takes; 1064 = the
 * 64-channel tile on 16x16-pixel tiles (single-level pyramids with sizes that are multiples of 16), 2064 = on 32x16-pixel
 * tiles (H a multiple of 32 too) */
int scan_conv3x3_bf16x6_instance(const scan_pyramid_t* d, int32_t Nout);
/* w [Cout][T][Cin_s] -> wt [Cin_s][T][Cout_s] (zero padded) */
int scan_weight_transpose(const float* w, int32_t Cout, int32_t T, int32_t Cin_s, float* wt, int32_t Cout_s,
                          void* stream);
/* column sums: db[c] (+)= sum_m dy[m][c], c < C; ws >= scan_colsum_ws_floats */
int64_t scan_colsum_ws_floats(int64_t M, int32_t C);
int scan_colsum(const float* dy, int64_t M, int32_t C, int32_t ld, float* db, int32_t accumulate, float* ws,
                void* stream);
/* out = dy * (y > 0)   (ReLU backward; y is the ReLU output; out may alias dy) */
int scan_relu_backward(const float* dy, const float* y, float* out, int64_t n, void* stream);

/* ---- GroupNorm(32 groups) + ReLU on a pyramid (replaces nn.GroupNorm(32,C)+nn.ReLU in the towers,
 *      condgraph.py:99-105, fcos.py:36-49, fcos_head_discriminator_con.py:31-32) ----
 * stats [n_levels*N*G*2] = (mean, rstd) per (level, image, group), eps = 1e-5.
 * ws (8-byte aligned): scan_groupnorm_ws_floats floats, used as fp64 accumulators. */
int scan_groupnorm_stats(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G, float eps, float* stats,
                         float* ws, void* stream);
int scan_groupnorm_relu_forward(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G, const float* stats,
                                const float* gamma, const float* beta, int32_t relu, float* y, void* stream);
/* scan_groupnorm_stats_from_sums + scan_groupnorm_relu_forward as ONE launch (sums: fp64 [n_levels*N*G][2] accumulated
 * by scan_conv3x3_gn_bf16x3; stats [n_levels*N*G][2] is written for scan_groupnorm_relu_backward). */
int scan_groupnorm_relu_forward_from_sums(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                          const float* sums, float eps, const float* gamma, const float* beta,
                                          int32_t relu, float* y, float* stats, void* stream);
int64_t scan_groupnorm_ws_floats(const scan_pyramid_t* d, int32_t C, int32_t G);
/* dx, dgamma[C] (+)=, dbeta[C] (+)=; beta [C] is the forward's shift: the ReLU mask is recomputed from x with the
 * forward's exact operation order instead of reading y back (may be NULL when relu == 0);
 * accumulate: bit 0 = add to dgamma / dbeta instead of overwriting, bit 1 = ws arrives cleared (else the call clears it);
 * ws: scan_groupnorm_ws_floats */
int scan_groupnorm_relu_backward(const float* x, const float* beta, const float* dy, const scan_pyramid_t* d, int32_t C,
                                 int32_t G, const float* stats, const float* gamma, int32_t relu, float* dx,
                                 float* dgamma, float* dbeta, int32_t accumulate, float* ws, void* stream);
/* The same three with y (forward) / dy (backward) as a column slice of a wider row-major matrix: ldy / lddy = its row
 * stride in floats (a multiple of 4, >= C), the slice starts at the pointer.  The reference concatenates the tower output
 * with the act maps (torch.cat, fcos_head_discriminator_con.py:104-118); here the tower's last GroupNorm writes straight
 * into the first C columns of that matrix and reads its gradient from the same columns of the conv's data gradient. */
int scan_groupnorm_relu_forward_ld(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G, const float* stats,
                                   const float* gamma, const float* beta, int32_t relu, float* y, int32_t ldy,
                                   void* stream);
int scan_groupnorm_relu_forward_from_sums_ld(const float* x, const scan_pyramid_t* d, int32_t C, int32_t G,
                                             const float* sums, float eps, const float* gamma, const float* beta,
                                             int32_t relu, float* y, int32_t ldy, float* stats, void* stream);
int scan_groupnorm_relu_backward_ld(const float* x, const float* beta, const float* dy, int32_t lddy,
                                    const scan_pyramid_t* d, int32_t C, int32_t G, const float* stats, const float* gamma,
                                    int32_t relu, float* dx, float* dgamma, float* dbeta, int32_t accumulate, float* ws,
                                    void* stream);

/* ---- the source pass's ground-truth plan on the device (reference rpn/fcos/loss.py:40-133: FCOS location -> GT
 *      assignment and centerness targets; :428-463: graph-node sampling of the source branch).  Rows = pyramid rows
 *      (level-major, image, y, x); strides [n_levels], soi [n_levels][2] = sizes of interest; boxes [N][G][4] xyxy padded
 *      to G per image, glabels [N][G], ng [N] = boxes of each image.
 *      scan_fcos_assign: labels [M] (+ an int32 copy), reg [M][4] = (l, t, r, b) of the assigned box, level_pos
 *      [SCAN_MAX_LEVELS] = positives per level (device counters; the host reads them once to size the rest).
 *      scan_fcos_compact: pos_list / neg_list [M]: per level, at its row offset, the rows with label > 0 / == 0 in row
 *      order.  scan_fcos_nodes (level_pos: HOST copy of the counters): node_index / node_labels [scan_fcos_nodes_count]
 *      in the reference's order [all negatives picked, all positives], pos_inds [n_pos], reg_pos [n_pos][4], ctr_pos
 *      [n_pos]. ---- */
int scan_fcos_assign(const scan_pyramid_t* d, const int32_t* strides, const float* soi, const float* boxes,
                     const int64_t* glabels, const int32_t* ng, int32_t G, int64_t* labels, int32_t* labels_i32,
                     float* reg, int32_t* level_pos, void* stream);
int scan_fcos_compact(const scan_pyramid_t* d, const int64_t* labels, int32_t* pos_list, int32_t* neg_list, void* stream);
int64_t scan_fcos_nodes_count(const scan_pyramid_t* d, const int32_t* level_pos);
int scan_fcos_nodes(const scan_pyramid_t* d, const int32_t* level_pos, const int64_t* labels, const float* reg,
                    const int32_t* pos_list, const int32_t* neg_list, int64_t* node_index, int64_t* node_labels,
                    int64_t* pos_inds, float* reg_pos, float* ctr_pos, void* stream);

/* ---- FPN top-down join on NHWC rows (reference backbone/fpn.py:62-75: inner = lateral + F.interpolate(top,
 *      scale_factor=2, mode="nearest")).  lat / y [N, 2h, 2w, C], coarse [N, h, w, C], C % 4 == 0.  Backward:
 *      d_lateral is the incoming gradient itself, d_coarse its 2x2 window sums (scan_downsample2x_sum: g [N, 2h, 2w, C]
 *      -> d [N, h, w, C]). ---- */
int scan_upsample2x_add(const float* lat, const float* coarse, int32_t N, int32_t h, int32_t w, int32_t C, float* y,
                        void* stream);
int scan_downsample2x_sum(const float* g, int32_t N, int32_t h, int32_t w, int32_t C, float* d, void* stream);

/* ---- 2x2 / stride-2 max pooling on NHWC rows (replaces nn.MaxPool2d(2, 2) of the VGG body,
 *      backbone/mmdetection/vgg.py:33).  x [N,H,W,C], y [N,H/2,W/2,C]; H, W even, C % 4 == 0.
 *      backward routes the gradient to the first maximum of each window (F.max_pool2d's rule);
 *      relu_mask != 0 additionally zeroes it where that maximum is <= 0 (x is a ReLU output whose own
 *      backward was left to its consumers, see scan_conv3x3_bf16x3's mask). ---- */
int scan_maxpool2x2_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream);
int scan_maxpool2x2_backward(const float* x, const float* y, const float* dy, int32_t N, int32_t H, int32_t W,
                             int32_t C, float* dx, int32_t relu_mask, void* stream);

/* ---- ResNet body pieces (reference backbone/resnet.py): stem pooling F.max_pool2d(x, 3, 2, 1) on NHWC rows
 *      (:335; forward only -- the stem is frozen for FREEZE_CONV_BODY_AT >= 1), y [N,(H-1)/2+1,(W-1)/2+1,C], C % 4 == 0;
 *      and the residual join y = max(a + b, 0) (:312-313; backward = scan_relu_backward(dy, y) to both branches).
 *      FrozenBatchNorm2d (layers/batch_norm.py:5-24) is an affine per channel and is folded into the conv weights /
 *      bias by the host side. ---- */
int scan_maxpool3x3s2_forward(const float* x, int32_t N, int32_t H, int32_t W, int32_t C, float* y, void* stream);
int scan_add_relu(const float* a, const float* b, float* y, int64_t n, void* stream);

/* ---- target-node density clustering on the device (replaces sklearn.cluster.DBSCAN(eps, min_samples = 5) inside
 *      PrototypeComputation.DBSCAN_batch_cpu, rpn/fcos/loss.py:397-423, which only asks "is the point in cluster 0?"
 *      -- noise -> 1, cluster 0 -> 0, selected = non-zero).  pts [n][D] fp32 rows, D % 4 == 0, n <= SCAN_DBSCAN_MAX.
 *      ws: scan_dbscan_ws_bytes(n) bytes (adjacency bit matrix n^2/8 + masks), 8-byte aligned.
 *      prepare: neighbour bit matrix (fp32 matrix-core P P^T, fp64 re-check at the threshold), core mask, seed;
 *               info[0] <- lowest core index (n when there is no core point).
 *      bfs_step: one breadth-first level over the core graph; call with parity 0, 1, 0, ... while *changed reads 1.
 *      finish:  in_cluster0[i] = 1 iff sklearn would label point i with 0. ---- */
#define SCAN_DBSCAN_MAX 1200000
int64_t scan_dbscan_ws_bytes(int64_t n);
int scan_dbscan_prepare(const float* pts, int64_t n, int32_t D, float eps, int32_t min_samples, void* ws,
                        int32_t* info, void* stream);
int scan_dbscan_bfs_step(int64_t n, void* ws, int32_t parity, int32_t* changed, void* stream);
int scan_dbscan_finish(int64_t n, void* ws, uint8_t* in_cluster0, void* stream);

/* ---- fused SGD with momentum (replaces torch.optim.SGD as configured by solver/build.py:7-43) ----
 * g' = g + wd*p ; buf = momentum*buf + g' ; p -= lr*buf   (first_step != 0: buf = g') */
int scan_sgd_momentum(float* p, const float* g, float* buf, int64_t n, float lr, float wd, float momentum,
                      int32_t first_step, void* stream);

/* The same update over up to SCAN_SGD_MAX_SEGMENTS (p, g, buf) ranges in ONE launch: the weights and the biases of the
 * eight sub-models each have their own lr / weight decay (solver/build.py:20-28: one param group per parameter, biases
 * with BIAS_LR_FACTOR and WEIGHT_DECAY_BIAS), which torch.optim.SGD walks one parameter at a time.  segs is HOST memory
 * (copied into the kernel arguments); element arithmetic identical to scan_sgd_momentum. */
#define SCAN_SGD_MAX_SEGMENTS 32
typedef struct {
  float* p;
  const float* g;
  float* buf;
  int64_t n;
  float lr;
  float wd;
  int32_t first_step;
  int32_t reserved;
} scan_sgd_segment_t;
int scan_sgd_momentum_multi(const scan_sgd_segment_t* segs, int32_t n_segs, float momentum, void* stream);

/* ---- bf16 hi / lo planes of many conv weights in one launch (same element mapping as scan_weight_split) ----
 * jobs: DEVICE array of n_jobs records of SCAN_SPLIT_JOB_WORDS int64: {w, wh, wl (device addresses), O, T, Cs, mode,
 * rows (= mode ? Cs : O), Csw, first_block, third plane or 0}; with a third plane the job is the three-piece split of
 * scan_weight_split3 and {wh, wl, third} = its {wh, wm, wl}; job_words = SCAN_SPLIT_JOB_WORDS as the caller compiled it (a table of another layout is refused); first_block = running sum of scan_weight_split_job_blocks() over the jobs
 * before it, total_blocks the sum over all.  No reference counterpart (the reference convolves in fp32 through
 * cuDNN); it exists because a DA iteration re-splits ~120 weights after every optimizer step. */
#define SCAN_SPLIT_JOB_WORDS 11
int64_t scan_weight_split_job_blocks(int32_t O, int32_t T, int32_t Cs, int32_t mode, int32_t Csw);
int scan_weight_split_batched(const int64_t* jobs, int32_t n_jobs, int32_t job_words, int64_t total_blocks, void* stream);

/* ---- CKA discriminator class branches as two stacked convolutions ----
 * replaces the per-class loop of FCOSDiscriminator_con.forward (modeling/discriminator/
 * fcos_head_discriminator_con.py:104-121: for each foreground class c, classifier_cls_c = conv3x3(C+1 -> H) -> ReLU ->
 * conv3x3(H -> 1) on cat(x, act[:, c+1])).  branches[c] holds the four parameter tensors of class c (HOST array of
 * device pointers): w0 [H][C+1][3][3], b0 [H], w2 [1][H][3][3], b2 [1]; element (o, ci, k = 3*ky + kx) of w0 lives at
 * o*s1o + ci*s1c + k*s1k, element (h, k) of w2 at h*s2c + k*s2k (so NCHW- and channels-last-stored parameters both
 * work).  stack writes
 *   w1 [Cf*H][9][Cs1]: row c*H+o = w0_c[o] on columns 0..C-1, w0_c[o][C] on column C+c, zero elsewhere
 *   b1 [Cf*H], w2 [Cf][9][Cs2]: row c = w2_c on columns c*H..c*H+H-1, zero elsewhere; b2 [Cf]
 * in the [O][T][Cs] layout the convolution entry points take.  unstack is its adjoint: the gradient of w1 / b1 / w2 /
 * b2 (any of them NULL = absent) scattered into grads[c] (same strides), added to what is there when accumulate != 0. */
#define SCAN_CKA_MAX_CLASSES 16
typedef struct {
  const float* w0;
  const float* b0;
  const float* w2;
  const float* b2;
} scan_cka_branch_t;
int scan_cka_stack_weights(const scan_cka_branch_t* branches, int32_t Cf, int32_t C, int32_t H, int64_t s1o, int64_t s1c,
                           int64_t s1k, int64_t s2c, int64_t s2k, int32_t Cs1, int32_t Cs2, float* w1, float* b1,
                           float* w2, float* b2, void* stream);
int scan_cka_unstack_grads(const scan_cka_branch_t* grads, int32_t Cf, int32_t C, int32_t H, int64_t s1o, int64_t s1c,
                           int64_t s1k, int64_t s2c, int64_t s2k, int32_t Cs1, int32_t Cs2, const float* dw1,
                           const float* db1, const float* dw2, const float* db2, int32_t accumulate, void* stream);

/* ---- the class branches' second convolution: G groups of 128 channels -> one output channel per group ----
 * replaces `classifier_cls_c[2]` = nn.Conv2d(128, 1, 3, padding=1) applied per foreground class
 * (fcos_head_discriminator_con.py:44-62,118-119), for all classes at once: x [M][G*128] rows of a pyramid (the ReLU-ed
 * hidden maps of the G classes side by side), w = the stacked weight [G][9][G*128] of scan_cka_stack_weights (only the
 * diagonal blocks w[g][t][g*128 + i] are read), y [M][Ns >= G] (columns >= G are written as zero).  Plain fp32 FMA
 * arithmetic, HBM-bound (one read of x).  ws: scan_gconv3x3_to1_ws_floats floats.
 *   dgrad: dx[q][c] = sum_t dy[q - off(t)][g(c)] * w[g][t][c], multiplied by (mask[q][c] > 0) when mask != NULL (the
 *          deferred ReLU of the producer); wgrad: dw[g][t][g*128 + i] (diagonal blocks only; added to dw when
 *          accumulate != 0), deterministic. */
int64_t scan_gconv3x3_to1_ws_floats(const scan_pyramid_t* d, int32_t G, int32_t Cg);
int scan_gconv3x3_to1_forward(const float* x, const scan_pyramid_t* d, int32_t G, int32_t Cg, const float* w,
                              const float* bias, float* y, int32_t Ns, float* ws, void* stream);
int scan_gconv3x3_to1_dgrad(const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G, int32_t Cg, const float* w,
                            const float* mask, float* dx, void* stream);
int scan_gconv3x3_to1_wgrad(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G, int32_t Cg,
                            float* dw, int32_t accumulate, float* ws, void* stream);
/* both gradients (relu_mask != 0: dx *= (x > 0)) */
int scan_gconv3x3_to1_backward(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G, int32_t Cg,
                               const float* w, int32_t relu_mask, float* dx, float* dw, int32_t accumulate, float* ws,
                               void* stream);
/* forward that also leaves the ReLU mask of x as bits (relu_bits: M * G * 4 uint32; bit 4 j + e of word
 * [(row * G + g) * 4 + q] = x[row][g * 128 + 16 j + 4 q + e] > 0), and the backward that masks dx with them instead of
 * re-reading x for the mask. */
int scan_gconv3x3_to1_forward_bits(const float* x, const scan_pyramid_t* d, int32_t G, int32_t Cg, const float* w,
                                   const float* bias, float* y, int32_t Ns, float* ws, uint32_t* relu_bits, void* stream);
int scan_gconv3x3_to1_backward_bits(const float* x, const float* dy, int32_t Ns, const scan_pyramid_t* d, int32_t G,
                                    int32_t Cg, const float* w, const uint32_t* relu_bits, float* dx, float* dw,
                                    int32_t accumulate, float* ws, void* stream);

/* ---- input pipeline: the step in front of the path ----
 * scan_resize_bilinear_u8 replaces torchvision F.resize on a PIL image = PIL Image.resize(size, BILINEAR), as
 * called by Resize.__call__ (reference data/transforms/transforms.py:57-61): Pillow's two-pass fixed-point resampler.
 * src uint8 [H, W, 3] -> dst uint8 [OH, OW, 3] (device pointers); tmp = [H, OW, 3] intermediate (only when both
 * sizes change).  xbounds/ybounds int32 [O][2] = (first input index, tap count), xcoef/ycoef int32 [O][k] fixed-point
 * (22 fractional bits) triangle-filter weights -- the host computes them in double precision as Pillow's
 * precompute_coeffs does (scan_amd/data.py: bilinear_tables); bit-exact against PIL. */
int scan_resize_bilinear_u8(const uint8_t* src, int32_t H, int32_t W, uint8_t* tmp, uint8_t* dst, int32_t OH, int32_t OW,
                            const int32_t* xbounds, const int32_t* xcoef, int32_t kx, const int32_t* ybounds,
                            const int32_t* ycoef, int32_t ky, void* stream);

/* ToTensor + Normalize(to_bgr255) (+ RandomHorizontalFlip's F.hflip) (transforms.py:64-90) + the collator's zero
 * padding (data/collate_batch.py:5-20, structures/image_list.py:54-66) for ONE image: src uint8 [H, W, 3] RGB ->
 * fp32 ((x / 255)[2,1,0] * 255 - mean) / std written into a zero-padded Hp x Wp slot.  layout 0: CHW planes
 * [3, Hp, Wp] (the reference tensor); layout 1: NHWC rows [Hp * Wp, 4] (what the first convolution reads; channel 3
 * is zero).  mean3 / std3 are HOST arrays of 3 floats (BGR order when to_bgr255).  Bit-exact vs torch-CPU. */
int scan_normalize_image_u8(const uint8_t* src, int32_t H, int32_t W, int32_t flip, int32_t to_bgr255,
                            const float* mean3, const float* std3, float* dst, int32_t Hp, int32_t Wp, int32_t layout,
                            void* stream);

/* Gradient of taking the rows of images [i0, i1) out of a pyramid matrix [M, C] (the paired step splits source and target
 * frames that way, reference trainer.py:284-352 runs them as separate batches): out [M, C] = g's rows at the taken images'
 * places, zero elsewhere, in one pass.  g: [(i1 - i0) * sum_l h_l w_l, C], level-major like the pyramid. */
int scan_take_images_backward(const float* g, const scan_pyramid_t* d, int32_t i0, int32_t i1, int32_t C, float* out,
                              void* stream);

/* Conditioned-kernel generator of the graph middle head (rpn/fcos/condgraph.py:313-319 get_conded_weight: paradigm
 * [K, 256, T] -> nn.RNN(256, 512, num_layers=2, nonlinearity tanh, h0 = 0) over the T slots, batch = the K classes ->
 * Conv2d(512, 256, (T, 1)) -> kernels [K, 256]) as one launch per link of the dependent chain (2 T + 1 forward, 2 T + 2
 * backward) instead of ~120 torch launches.  K <= 9, T <= 3.  x: the paradigm slot-major, [T, K, 256].
 * weights / grads: ten device pointers in the order weight_ih_l0 [512,256], weight_hh_l0 [512,512], bias_ih_l0, bias_hh_l0,
 * weight_ih_l1 [512,512], weight_hh_l1, bias_ih_l1, bias_hh_l1, cond_nx1.weight [256,512,T,1] (contiguous), cond_nx1.bias.
 * forward: h0 / h1 [T, K, 512] = the two layers' states (kept for the backward), kernels [K, 256].
 * backward: every gradient is overwritten (the paradigm is a buffer: no input gradient); ws: scan_cond_rnn_ws_floats(). */
int scan_cond_rnn_forward(const float* x, int32_t K, int32_t T, const float* const* weights, float* h0, float* h1,
                          float* kernels, void* stream);
int64_t scan_cond_rnn_ws_floats(void);
int scan_cond_rnn_backward(const float* x, int32_t K, int32_t T, const float* const* weights, const float* h0,
                           const float* h1, const float* dkernels, float* const* grads, float* ws, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SCAN_HIP_H */
