/* libdealyolo_hip.so -- C ABI of the MI355X (gfx950) DEAL-YOLO hot path.
 *
 * The reference (adityaX1412/Experiment-YOLO, an Ultralytics-YOLOv8 fork) has NO native code and no FFI on this
 * path: every operator below replaces a *sequence of ATen calls* issued by the reference's Python modules.  Each
 * entry therefore cites the reference Python interface (file:line under /root/reference/ultralytics) whose
 * arithmetic it takes over; INTEGRATION.md shows the ctypes binding a maintainer adds on the reference side.
 *
 * Conventions
 *   - plain pointers + sizes only; all pointers are DEVICE pointers unless named host_*;
 *   - activations are NHWC fp16 ("f16"), addressed as (pointer, ld) where ld is the pixel stride in ELEMENTS, so a
 *     channel slice of a wider buffer (Concat / chunk) is just an offset pointer with the parent's ld;
 *     channel counts and ld are multiples of 8, pointers 16-byte aligned;
 *   - every function enqueues work on `stream` and returns immediately: no allocation, no synchronisation, no
 *     global state; workspaces are caller-owned;  return value: DY_OK (0) or a negative DY_ERR_* code.
 */
#ifndef DEALYOLO_HIP_H
#define DEALYOLO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#ifndef __HIP__
typedef struct ihipStream_t* hipStream_t;
#endif

#define DY_OK 0
#define DY_ERR_ARG (-1)
#define DY_ERR_LAUNCH (-2)
#define DY_ERR_ALIGN (-3)

/* A channel CONCATENATION that is never materialised: the tensor's channels [c_end[s-1], c_end[s]) live in segment s, an fp16 NHWC
 * tensor (or channel slice) of its own pixel stride.  C2f's ``torch.cat(y, 1)`` (nn/modules/block.py:222-226), SPPF's and the model's
 * Concat layers (nn/modules/conv.py:338-348) feed 1x1 convolutions only; those read their input, write their input gradient and read
 * their weight-gradient operand through this table, so every concat member stays a contiguous tensor for the kernels that touch it
 * alone (BatchNorm apply / backward reduce, the 3x3 convs of a Bottleneck) instead of a strided slice of a wide buffer. */
#define DY_MAX_SEGS 8
typedef struct DySegs {
  int nseg;                      /* 1..DY_MAX_SEGS */
  int c_end[DY_MAX_SEGS];        /* exclusive end channel of segment s in the concatenated tensor (multiples of 8, increasing) */
  int ld[DY_MAX_SEGS];           /* pixel stride of segment s in elements */
  int acc[DY_MAX_SEGS];          /* as an OUTPUT: 1 = add to what the segment holds (gradient fan-in), 0 = store.  As the INPUT of
                                    dy_conv1x1_forward_segs: 2 = the segment is nn.Upsample(None, 2, 'nearest') of a (n, h/2, w/2) tensor
                                    (ptr / ld are that tensor's): pixel (y, x) reads (y >> 1, x >> 1), the up-sampled copy in front
                                    of the reference's Concat (the head of every cfg/models YAML) is never written */
  const void* ptr[DY_MAX_SEGS];  /* first channel of segment s */
} DySegs;
int dy_segs_bytes(void); /* sizeof(DySegs) in the library (bindings check their layout) */

/* conv epilogue flags */
#define DY_EPI_STATS 1   /* write per-workgroup sum / sum-of-squares partials of the (fp16-rounded) output */
#define DY_EPI_BIAS 2    /* add bias[cout] */
#define DY_EPI_SILU 4    /* apply SiLU */
#define DY_EPI_F32OUT 8  /* y is fp32 (Detect's final 1x1 convs feed the loss in fp32) */
#define DY_EPI_ACCUM 16  /* y += result (gradient fan-in) */
#define DY_EPI_STATS_ACC 32 /* with DY_EPI_STATS: `partials` is a double accumulator [DY_BN_COPIES][2][round16(cout)], zeroed by the
                             * caller, that every workgroup ADDS its sums into (copy = workgroup index % DY_BN_COPIES); consumed by
                             * dy_bn_act_apply_acc -- no dy_bn_finalize launch */
#define DY_BN_COPIES 16

#define DY_ACT_NONE 0
#define DY_ACT_SILU 1
#define DY_ACT_LEAKY 2 /* LeakyReLU(0.1), ScalSeq */

int dy_abi_version(void);

/* ---- Conv.forward / forward_fuse, nn/modules/conv.py:41-59; nn.Conv2d heads nn/modules/head.py:38-42;
 *      LDConv.p_conv nn/modules/conv.py:356; input-gradient half of aten::convolution_backward. ------------------ */
int dy_conv_geometry(int cin, int cout, int ks, int stride, int* cin_p, int* cout_p, int* cc, int* nch, int* mt,
                     int* ngroups, int* ksteps, int* packed_elems);
/* fp32 OIHW master weights -> fp16 MFMA-packed; scale: optional per-Cout factor (BN folding, utils/torch_utils.py:171-198);
 * transposed=1 builds the (Cout->Cin, flipped taps) pack consumed by the stride-1 input-gradient pass. */
int dy_pack_weights(const float* w, const float* scale, void* out, int cout, int cin, int ks, int stride,
                    int transposed, hipStream_t stream);
/* every pack of a model in ONE launch: fill host descriptors (dy_pack_desc_bytes() each, first_block = running sum of the
 * returned block counts), copy them to the device once, then call dy_pack_weights_batched every step. */
int dy_pack_desc_bytes(void);
int dy_pack_desc_fill(void* desc, const float* w, const float* scale, void* out, int cout, int cin, int ks, int stride,
                      int transposed, int ld_taps, int ld_cphys, int first_block);
int dy_pack_weights_batched(const void* descs_device, int n, int total_blocks, hipStream_t stream);
/* y[n,ho,wo,:cout] = conv(x)[...]; x: (n,h,w,cin) with cin a multiple of 8 (zero-padded channels).
 * dil=2: x is read as a zero-dilated map of size (2h,2w) -- the input gradient of a stride-2 conv; out_h/out_w (>0)
 * then give the extent of the forward input (2h or 2h-1), otherwise pass 0 to derive the output extent.
 * partials: [num_partials][2][cout_p] floats, required with DY_EPI_STATS. */
/* size limit: the input is addressed with 32-bit byte offsets -- n*h*w*ldx*2 (1x1) or h*w*ldx*2 (3x3, per image) < 2 GiB, else DY_ERR_ARG */
int dy_conv_forward(const void* x, int ldx, const void* w_packed, const float* bias, void* y, int ldy, float* partials,
                    int n, int h, int w, int cin, int cout, int ks, int stride, int dil, int out_h, int out_w, int epi,
                    int* num_partials, hipStream_t stream);
int dy_conv_num_partials(int n, int h, int w, int cin, int cout, int ks, int stride, int dil);
/* Input gradient of a stride-1 conv (the dgrad half of aten::convolution_backward: dy (n,h,w,cin) x transposed pack -> dx (n,h,w,cout))
 * that is the ONLY writer of dx = the gradient w.r.t. the activated output of a Conv (nn/modules/conv.py:49-55): the first pass of that
 * Conv's BatchNorm backward (sums of g = dx * silu'(raw*scale+shift) and g*xhat over the pixels, dy_bn_act_bwd_reduce_acc) runs in
 * the epilogue on the values being stored and is added into acc [DY_BN_COPIES][2][C].  C == cout; dy_conv_red_supported(cin, cout, ks)
 * tells whether the geometry has this form (else DY_ERR_ARG). */
/* Conv.forward_fuse followed by Bottleneck's shortcut add (nn/modules/conv.py:57-59, nn/modules/block.py:333-335) in one launch:
 * y = fp16(fp16(SiLU(conv(x) + bias)) + res), the bits of dy_conv_forward(DY_EPI_BIAS | DY_EPI_SILU) followed by dy_add.  3x3 stride-1
 * convolutions the ping-pong kernel takes (dy_conv_res_supported); res: fp16 (N, Ho, Wo, ldres). */
int dy_conv_res_supported(int cin, int cout, int ks, int stride);
int dy_conv_forward_res(const void* x, int ldx, const void* w_packed, const float* bias, const void* res, int ldres, void* y, int ldy,
                        int n, int h, int w, int cin, int cout, int ks, int stride, hipStream_t stream);
int dy_conv_red_supported(int cin, int cout, int ks);
int dy_conv_input_grad_red(const void* dy, int lddy, const void* w_packed_t, void* dx, int lddx, int n, int h, int w, int cin, int cout,
                           int ks, const void* raw, int ldraw, const float* coef, double* acc, int C, hipStream_t stream);
/* host-side: the kernel instantiation dy_conv_forward launches for this geometry, spelled as rocprofv3 prints it */
int dy_conv_kernel_name(int cin, int cout, int ks, int stride, char* out, int cap);
/* ... for a launch with this output width, dilation and epilogue: maps exactly 40 pixels wide run 3x3 stride-1 convs of 32-channel
 * chunks on full-width tiles (conv_mfma_pp_kernel's last template argument), which rocprofv3 lists as a kernel of their own */
int dy_conv_kernel_name_at(int cin, int cout, int ks, int stride, int out_w, int dil, int epi, char* out, int cap);

/* ---- weight-gradient half of aten::convolution_backward for the same modules. dw: fp32 OIHW (cout,cin,ks,ks). -- */
int dy_wgrad_workspace(int n, int h, int w, int cin, int cout, int ks, int stride, int* nslabs, long* slab_elems);
/* host-side: the kernel instantiation dy_conv_wgrad launches for this geometry, spelled as rocprofv3 prints it */
int dy_wgrad_kernel_name(int cin, int cout, int ks, int stride, char* out, int cap);
/* the same for a given map: on small maps a workgroup owns a smaller (ci, co) channel block, so that fewer fp32 weight slabs exist */
int dy_wgrad_kernel_name_at(int n, int h, int w, int cin, int cout, int ks, int stride, char* out, int cap);
int dy_conv_wgrad(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h, int w,
                  int cin, int cout, int ks, int stride, int accumulate, hipStream_t stream);
/* Weight gradient of a Conv (conv + BatchNorm + SiLU, nn/modules/conv.py:49-55) from the gradient w.r.t. its ACTIVATED output
 * (autograd: silu_backward + native_batch_norm_backward + the weight half of convolution_backward in one kernel): the
 * BatchNorm / SiLU backward apply pass runs on the dY operand while it is staged -- coef [4][cout] from the forward, acc
 * [DY_BN_COPIES][2][cout] from dy_bn_act_bwd_reduce_acc -- and d(raw conv output) is written to draw (N,Ho,Wo,ldraw; the geometry
 * of raw) for the input-gradient pass; dgamma / dbeta (may be NULL) receive the BatchNorm parameter gradients.  cout % 16 == 0. */
int dy_conv_wgrad_bn(const void* x, int ldx, const void* dy, int lddy, const void* raw, int ldraw, void* draw, const float* coef,
                     const double* acc, float* dgamma, float* dbeta, float count, float* slabs, float* dw, int n, int h, int w,
                     int cin, int cout, int ks, int stride, int accumulate, hipStream_t stream);
int dy_conv_wgrad_ld_bn(const void* x, int ldx, const void* dy, int lddy, const void* raw, int ldraw, void* draw,
                        const float* coef, const double* acc, float* dgamma, float* dbeta, float count, float* slabs, float* dw,
                        int n, int h, int w, int cout, int ld_cin, int ld_taps, int ld_cphys, int accumulate, hipStream_t stream);
/* 1x1 convolutions over a never-materialised concatenation (DySegs above): forward (dy_conv_forward with xs in place of (x, ldx); epi as
 * there), input gradient (every 8-channel piece of W^T dy stored in / added to its segment) and weight gradient with the BatchNorm backward
 * apply inside (dy_conv_wgrad_bn with xs in place of (x, ldx)).  dy_conv1x1_segs_supported: a Cin chunk exists that no segment boundary
 * cuts (the geometry's own, or 32 where that is 64: same packed weights) and the ping-pong kernel takes the shape. */
int dy_conv1x1_segs_supported(int cin, int cout, const DySegs* xs);
int dy_conv1x1_segs_kernel_name(int cin, int cout, const DySegs* xs, char* out, int cap); /* as dy_conv_kernel_name, for dy_conv1x1_forward_segs */
int dy_conv1x1_forward_segs(const DySegs* xs, const void* w_packed, const float* bias, void* y, int ldy, float* partials, int n, int h,
                            int w, int cin, int cout, int epi, hipStream_t stream);
int dy_conv1x1_input_grad_segs(const void* dy, int lddy, const void* w_packed_t, const DySegs* dxs, int n, int h, int w, int cin, int cout,
                               hipStream_t stream);
int dy_conv1x1_wgrad_bn_segs(const DySegs* xs, const void* dy, int lddy, const void* raw, int ldraw, void* draw, const float* coef,
                             const double* acc, float* dgamma, float* dbeta, float count, float* slabs, float* dw, int n, int h, int w,
                             int cin, int cout, int accumulate, hipStream_t stream);
/* C2f's ``cv1(x).chunk(2, 1)`` (nn/modules/block.py:223) with each half a tensor of its own ("two planes": channels [0, csplit) at one
 * base, [csplit, C) at another): the BatchNorm apply that writes them (dy_bn_act_apply_acc, no residual), the backward reduce that
 * reads their gradients (dy_bn_act_bwd_reduce_acc, no shortcut gradient) and the 1x1 weight gradient with the BatchNorm backward
 * inside whose dY operand they are (dy_conv_wgrad_bn; xs != NULL: dy_conv1x1_wgrad_bn_segs).  The weight gradient wants both planes
 * in one allocation, dy2 >= dy + n*h*w*lddy, same pixel stride. */
int dy_bn_act_apply_acc_split(const void* x, int ldx, void* y, int ldy, void* y2, int ldy2, int csplit, const double* acc,
                              const float* gamma, const float* beta, float* running_mean, float* running_var, float* coef, long npix,
                              int C, int act, float count, float eps, float momentum, hipStream_t stream);
int dy_bn_act_bwd_reduce_acc_split(const void* dy, int lddy, const void* dy2, int lddy2, int csplit, const void* x, int ldx,
                                   const float* coef, double* acc, long npix, int C, int act, hipStream_t stream);
int dy_conv1x1_wgrad_bn_planes(const DySegs* xs, const void* x, int ldx, const void* dy, const void* dy2, int lddy, int csplit,
                               const void* raw, int ldraw, void* draw, const float* coef, const double* acc, float* dgamma,
                               float* dbeta, float count, float* slabs, float* dw, int n, int h, int w, int cin, int cout,
                               int accumulate, hipStream_t stream);
/* The stem Conv(3 -> 16, k 3, s 2, p 1) of the model YAMLs (nn/modules/conv.py:41-55 as model.0) read straight from the image
 * batch the trainer hands the model (models/yolo/detect/train.py:57-59: fp32 NCHW, img * mul): no import pass, no padded copy.
 * dy_stem_forward writes the raw conv output (N,Ho,Wo,ldraw) fp16 and ADDS the BatchNorm sums into acc [DY_BN_COPIES][2][16]
 * (then dy_bn_act_apply_acc as for any Conv).  dy_stem_wgrad_bn: weight gradient with the BatchNorm / SiLU backward apply inside
 * (as dy_conv_wgrad_bn; no d(raw) output -- the image needs no gradient): slabs [dy_stem_grid()][9][16][16] fp32 for
 * dy_wgrad_reduce_batched (descriptor: cin 3, cout 16, ks 3, stride 2). */
int dy_stem_grid(int n, int h, int w);
int dy_stem_forward(const float* img_nchw, const float* weight, void* raw, int ldraw, double* acc, int n, int h, int w, float mul,
                    hipStream_t stream);
/* The same stem in eval mode (Conv.forward_fuse, nn/modules/conv.py:57-59; get_FPS.py:49 fuses the model before timing it):
 * y = act(conv(img * mul) + bias) as fp16 NHWC, no statistics.  bias NULL: none (un-fused eval: BatchNorm with running statistics follows
 * as its own pass); silu 0: no activation. */
int dy_stem_forward_eval(const float* img_nchw, const float* weight, const float* bias, void* y, int ldy, int n, int h, int w, float mul,
                         int silu, hipStream_t stream);
int dy_stem_wgrad_bn(const float* img_nchw, const void* dy, int lddy, const void* raw, int ldraw, const float* coef,
                     const double* acc, float* dgamma, float* dbeta, float count, float* slabs, int n, int h, int w, float mul,
                     hipStream_t stream);
/* Weight gradient of a conv with bias (Detect's final nn.Conv2d nn/modules/head.py:38-42, LDConv.p_conv nn/modules/conv.py:356) that also
 * takes the bias gradient -- the sum of dY over the pixels -- from the dY granules it stages: added into bias_acc
 * [DY_BN_COPIES][round8(cout)] (fp64, zeroed by the caller), finished by dy_wgrad_reduce_batched for descriptors marked with
 * dy_wgrad_reduce_desc_bias(desc, bias_acc, dbias, round8(cout)).  Not for 48- / 96-channel outputs (DY_ERR_ARG). */
int dy_conv_wgrad_bias(const void* x, int ldx, const void* dy, int lddy, double* bias_acc, float* slabs, float* dw, int n, int h, int w,
                       int cin, int cout, int ks, int stride, int accumulate, hipStream_t stream);
int dy_wgrad_reduce_desc_bias(void* desc, const double* bias_acc, float* dbias, int c);
/* dw == NULL in dy_conv_wgrad / dy_conv_wgrad_ld defers the slab reduction: the caller keeps that layer's slabs alive, fills one
 * descriptor per layer (host side, sizeof = dy_wgrad_reduce_desc_bytes(); returns the layer's block count, first_block = the
 * exclusive prefix sum of those counts) and reduces every layer of the backward pass in ONE launch. */
int dy_wgrad_reduce_desc_bytes(void);
int dy_wgrad_reduce_desc_fill(void* desc, const float* slabs, int nslabs, float* dw, int cin, int cout, int ks, int stride,
                              int accumulate, int ld_taps, int ld_cphys, int ld_cin, int first_block);
int dy_wgrad_reduce_batched(const void* descs_device, int n, int total_blocks, hipStream_t stream);

/* ---- LDConv, nn/modules/conv.py:350-503: sampling (offset -> 4-corner bilinear gather, :366-410, :456-489) writes
 *      x_off[pix][n*C + c]; the (N,1) column conv (:354) then is a 1x1 conv with K = N*C whose weights are packed /
 *      whose weight gradient is unpacked by the *_ld variants (master layout (cout, cin, N, 1)). ------------------- */
int dy_ldconv_sample(const void* x, int ldx, const float* off, int ldoff, const int* pn, void* xo, int ldxo, int n, int H,
                     int W, int h, int w, int C, int Np, int stride, hipStream_t stream);
/* dx32: (n,H,W,C) fp32 scatter accumulator, zeroed by the caller (NULL: skip); doff: fp16 (n,h,w,lddoff), ch [0,2Np) */
int dy_ldconv_sample_backward(const void* x, int ldx, const float* off, int ldoff, const int* pn, const void* dxo,
                              int lddxo, float* dx32, void* doff, int lddoff, int n, int H, int W, int h, int w, int C,
                              int Np, int stride, hipStream_t stream);
/* The same backward with the input gradient computed by a deterministic gather and written (accumulate=0) or added
 * (accumulate=1) straight into the fp16 gradient map dx (n,H,W,lddx).  Samples whose offsets stay within rmax pixels
 * (1..16) are gathered, with the candidate radius shrunk to the largest |offset| of the layer, measured on the device into
 * scratch (>= 4 bytes); the far ones go through fp32 atomics into dx32 (n*H*W*C floats, caller-owned, need not be
 * zeroed, untouched when the layer has no far sample) and are folded in by the gather -- no host decision anywhere. */
int dy_ldconv_sample_backward_gather(const void* x, int ldx, const float* off, int ldoff, const int* pn, const void* dxo,
                                     int lddxo, void* dx, int lddx, int accumulate, float* dx32, void* doff, int lddoff,
                                     void* scratch, int rmax, int n, int H, int W, int h, int w, int C, int Np, int stride,
                                     hipStream_t stream);
int dy_f32_to_f16_add(const float* src, void* dst, int ld, long npix, int C, int accumulate, hipStream_t stream);
int dy_pack_weights_ld(const float* w, void* out, int cout, int cin, int ld_taps, int ld_cphys, int transposed,
                       hipStream_t stream);
int dy_conv_wgrad_ld(const void* x, int ldx, const void* dy, int lddy, float* slabs, float* dw, int n, int h, int w,
                     int cout, int ld_cin, int ld_taps, int ld_cphys, int accumulate, hipStream_t stream);

/* ---- nn.BatchNorm2d / BatchNorm3d (training statistics) + SiLU / LeakyReLU, nn/modules/conv.py:49-55,
 *      nn/extra_modules/block.py:3440-3441, utils/torch_utils.py:347-349, and their autograd backward. ------------- */
/* coef: [4][C] = scale, shift, mean, invstd.  Up to three weighted partial sets (ScalSeq: three resolutions). */
int dy_bn_finalize(const float* p0, int n0, float w0, const float* p1, int n1, float w1, const float* p2, int n2,
                   float w2, const float* gamma, const float* beta, float* running_mean, float* running_var,
                   float* coef, int C, float count, float eps, float momentum, int update_running, hipStream_t stream);
int dy_bn_eval_coef(const float* gamma, const float* beta, const float* running_mean, const float* running_var,
                    float* coef, int C, float eps, hipStream_t stream);
/* y = act(x*scale+shift) (+ res) */
int dy_bn_act_apply(const void* x, int ldx, const void* res, int ldr, void* y, int ldy, const float* coef, long npix,
                    int C, int act, hipStream_t stream);
/* The same three passes WITHOUT the finalize launches (nn/modules/conv.py:49-55 and its autograd backward, as above): the
 * statistics travel in fp64 accumulators acc[DY_BN_COPIES][2][C] that the producer adds into (dy_conv_forward with
 * DY_EPI_STATS | DY_EPI_STATS_ACC; dy_bn_act_bwd_reduce_acc) and the consumer sums in its prologue; block 0 of the consumer leaves
 * coef [4][C] + the running statistics (forward) / dgamma, dbeta (backward) behind.  The caller zeroes acc before the producer. */
int dy_bn_act_apply_acc(const void* x, int ldx, const void* res, int ldr, void* y, int ldy, const double* acc,
                        const float* gamma, const float* beta, float* running_mean, float* running_var, float* coef,
                        long npix, int C, int act, float count, float eps, float momentum, hipStream_t stream);
/* res_grad (may be NULL): gradient of a residual operand added after the activation (Bottleneck shortcut, nn/modules/block.py:333-335)
 * -- it equals dy, and is stored (res_accumulate 0) or added (1) while dy streams through, instead of by a pass of its own */
int dy_bn_act_bwd_reduce_acc(const void* dy, int lddy, const void* x, int ldx, const float* coef, double* acc, long npix,
                             int C, int act, void* res_grad, int ldrg, int res_accumulate, hipStream_t stream);
/* The accumulator forms of the forward apply / backward reduce (SiLU, no residual / shortcut operand) for SEVERAL tensors in one launch
 * (n <= dy_bn_group_max(); array arguments have n entries): independent Convs of one stage, whose passes would otherwise queue behind
 * each other although the small ones cannot fill the chip. */
int dy_bn_group_max(void);
int dy_bn_act_apply_acc_group(int n, const void* const* x, const int* ldx, void* const* y, const int* ldy, const double* const* acc,
                              const float* const* gamma, const float* const* beta, float* const* running_mean, float* const* running_var,
                              float* const* coef, const long* npix, const int* C, const float* count, const float* eps,
                              const float* momentum, hipStream_t stream);
int dy_bn_act_bwd_reduce_acc_group(int n, const void* const* dy, const int* lddy, const void* const* x, const int* ldx,
                                   const float* const* coef, double* const* acc, const long* npix, const int* C, hipStream_t stream);
/* dy_bn_act_bwd_reduce_acc for a gradient that has ROWS (what dy_conv1x1_rows_backward passes down: zero at every background anchor):
 * the sums visit the foreground pixels of `assigned` (B = n images, A anchors each, this tensor's pixel (b, r) is anchor a0 + r) only */
int dy_bn_act_bwd_reduce_rows(const void* dy, int lddy, const void* x, int ldx, const float* coef, double* acc, int n, int hw, int C,
                              int act, const int* assigned, int A, int a0, hipStream_t stream);
int dy_bn_act_bwd_apply_acc(const void* dy, int lddy, const void* x, int ldx, void* dx, int lddx, const float* coef,
                            const double* acc, float* dgamma, float* dbeta, long npix, int C, int act, float count,
                            hipStream_t stream);
int dy_bn_act_bwd_reduce(const void* dy, int lddy, const void* x, int ldx, const float* coef, float* partials,
                         int max_partials, long npix, int C, int act, int* nparts, hipStream_t stream);
/* bwdcoef: [2][C] = mean(g), mean(g*xhat); dgamma/dbeta fp32 (may be NULL) */
int dy_bn_bwd_finalize(const float* partials, int nparts, float* dgamma, float* dbeta, float* bwdcoef, int C,
                       float count, int accumulate, hipStream_t stream);
int dy_bn_act_bwd_apply(const void* dy, int lddy, const void* x, int ldx, void* dx, int lddx, const float* coef,
                        const float* bwdcoef, long npix, int C, int act, int frozen_stats, hipStream_t stream);

/* ---- data movement: image import (models/yolo/detect/train.py:57-59), nn.Upsample(2,'nearest') (model YAMLs),
 *      SPPF's MaxPool2d(5,1,2) chain nn/modules/block.py:166-171, Add nn/extra_modules/block.py:3483-3484,
 *      Concat nn/modules/conv.py:338-348, ScalSeq tail nn/extra_modules/block.py:3437-3443. -------------------------- */
int dy_import_image(const float* x_nchw, void* y, int n, int c, int h, int w, int cp, float mul, hipStream_t stream);
/* The loader's batch: uint8 NHWC RGB (n,h,w,3), 4-byte aligned -> fp16 NHWC with channels zero-padded to cp, value u8/255
 * (models/yolo/detect/train.py:59 `batch["img"].float() / 255`).  flip: NULL, or n bytes -- bit 0 mirrors image i left-right,
 * bit 1 up-down while converting (RandomFlip, data/augment.py:651-683; the loader then ships unflipped pixels).  index: NULL,
 * or n ints -- batch slot i reads image index[i] of x, a pool of decoded images resident in HBM (no per-step host copy).
 * hsv: NULL, or n x 3 floats -- RandomHSV's hue / saturation / value gains of image i (data/augment.py:605-624). */
int dy_import_image_u8(const void* x, void* y, int n, int h, int w, int cp, const void* flip, const int* index,
                       const float* hsv, hipStream_t stream);
/* Mosaic (data/augment.py:208-241) + random affine / perspective (cv2.warpAffine | warpPerspective of RandomPerspective :384-435,
 * border 114) + MixUp (:326-345) + flips, composed from an HBM-resident pool of letterboxed s x s uint8 images straight into the
 * fp16 NHWC stem input.  slots: n x 2 records (dy_warp_slot_bytes() bytes per sample: the sample, then its MixUp partner) of
 * 48 words = { float minv[6] (output pixel -> canvas, rows 0-1 of the inverse map); int canvas_w, canvas_h, xc, yc, flip, npatch;
 * int patch[4][7] = pool index, destination x1,y1,x2,y2 on the canvas, source x,y; float hsv[3] (RandomHSV gains, 0 = off);
 * float pinv[3] (row 2 of the inverse map; 0,0,1 when affine); double mix_r (MixUp ratio of this image; < 0: no partner) },
 * built on the host from the random draws. */
int dy_warp_import_u8(const void* pool, const void* slots, void* y, int n, int s, int cp, hipStream_t stream);
int dy_warp_slot_bytes(void);
int dy_add(const void* a, int lda, const void* b, int ldb, const void* c, int ldc, void* y, int ldy, long npix, int C,
           hipStream_t stream);
int dy_upsample2x(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int C, int backward, int accumulate,
                  hipStream_t stream);
int dy_maxpool5(const void* x, int ldx, void* y, int ldy, void* argmax, int n, int h, int w, int C, hipStream_t stream);
int dy_maxpool5_backward(const void* dy, int lddy, const void* argmax, void* dx, int lddx, int n, int h, int w, int C,
                         int accumulate, hipStream_t stream);
/* SPPF's three chained pools (nn/modules/block.py:166-171) in one launch when a whole map fits in LDS.  cat: [n*h*w][ld] fp16, slice 0
 * (C channels) = cv1's output, slices 1..3 receive y1..y3; argK: [n*h*w][C] uint8 arg-max of pool K (as dy_maxpool5 writes it; may be
 * NULL in forward when no backward follows).  dy_sppf_pool3_supported: channels per workgroup (16 / 8) or 0 = use dy_maxpool5.
 * Backward: gcat holds the gradients of the four slices (from cv2's input gradient); on return slice 0 holds (accK: is added to)
 * the gradient of cv1's output; the gradients of y1 / y2 are consumed inside the kernel and not written back. */
int dy_sppf_pool3_supported(int h, int w, int C);
int dy_sppf_pool3(void* cat, int ld, int C, void* arg0, void* arg1, void* arg2, int n, int h, int w, hipStream_t stream);
int dy_sppf_pool3_backward(void* gcat, int ld, int C, const void* arg0, const void* arg1, const void* arg2, int n, int h, int w,
                           int acc0, int acc1, int acc2, hipStream_t stream);
int dy_scalseq_tail(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2, const void* res,
                    int ldres, void* y, int ldy, const float* coef, int n, int h, int w, int C, hipStream_t stream);
/* mode 0: BN3d backward partial sums for `level`; mode 1: dr_level */
int dy_scalseq_tail_backward(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2, const void* dy,
                             int lddy, void* dr, int lddr, const float* coef, const float* bwdcoef, float* partials,
                             int max_partials, int n, int h, int w, int C, int level, int mode, int* nparts,
                             hipStream_t stream);
/* all three levels in one pass (mode 0: BN partial sums [nparts][2][C]; mode 1: dr0/dr1/dr2 at full, 1/2, 1/4 resolution) */
int dy_scalseq_tail_backward_all(const void* r0, int ld0, const void* r1, int ld1, const void* r2, int ld2,
                                 const void* dy, int lddy, void* dr0, int lddr0, void* dr1, int lddr1, void* dr2,
                                 int lddr2, const float* coef, const float* bwdcoef, float* partials, int max_partials,
                                 int n, int h, int w, int C, int mode, int* nparts, hipStream_t stream);
/* Zoom_cat fine branch nn/extra_modules/block.py:3406-3412: 2x2 max + 2x2 mean to half resolution (h, w = output extent) */
int dy_zoom_pool(const void* x, int ldx, void* y, int ldy, int n, int h, int w, int C, hipStream_t stream);
int dy_zoom_pool_backward(const void* x, int ldx, const void* dy, int lddy, void* dx, int lddx, int n, int h, int w, int C,
                          int accumulate, hipStream_t stream);
int dy_copy_slice(const void* x, int ldx, void* y, int ldy, long npix, int C, hipStream_t stream);
int dy_fill_zero(void* p, size_t bytes, hipStream_t stream);

/* ---- v8DetectionLoss.__call__ utils/loss.py:356-457 + TaskAlignedAssigner utils/tal.py:39-258 + BboxLoss
 *      utils/loss.py:202-250 + bbox_iou / wasserstein_loss / WiseIouLoss utils/metrics.py:75-126,540-565,591-645,
 *      forward AND backward (gradients w.r.t. the head logits are written as fp16 * gscale). ------------------------ */
typedef struct DyLossArgs {
  int nl, B, nc, ncp, nmax;      /* levels (<=4), batch, classes, classes padded to 8, gt capacity per image */
  const float* box[4];           /* (B,H,W,64) fp32 DFL logits per level */
  const float* cls[4];           /* (B,H,W,ncp) fp32 class logits per level */
  void* dbox[4];                 /* (B,H,W,64) fp16 out */
  void* dcls[4];                 /* (B,H,W,ncp) fp16 out */
  int H[4], W[4];
  float stride[4];
  const float* t_batch_idx;      /* (n,) targets as the dataloader emits them, data/dataset.py:207-224 */
  const float* t_cls;            /* (n,) */
  const float* t_boxes;          /* (n,4) normalised xywh */
  int n_targets;
  const int* n_targets_dev;      /* optional device copy of n_targets (read instead of n_targets when non-NULL, so a
                                    captured hipGraph sees per-step counts) */
  float img_w, img_h;
  float hyp_box, hyp_cls, hyp_dfl; /* cfg/default.yaml box/cls/dfl gains */
  int use_wiou, use_nwd;         /* BboxLoss.use_wiseiou / nwd_loss toggles, utils/loss.py:194,197 */
  float iou_ratio;
  const float* gscale;           /* device scalar: loss scale folded into every gradient */
  float* scalars;                /* 16 persistent floats: [1] target_scores_sum [3] #fg [4] WIoU iou_mean (init 1)
                                    [5..7] loss_items box,cls,dfl [8] loss.sum()*B [9] error flag */
  void* workspace;               /* dy_loss_workspace_bytes(B, A, nmax) bytes */
  int dbox_rows_only;            /* non-zero: dbox is written at foreground anchors only and NOT zeroed elsewhere -- for a consumer that
                                    reads it through the assignment (dy_conv1x1_rows_backward); 0: every row is defined */
  int box_from_input;            /* non-zero: box[] is not read.  pred_box in the workspace was written by dy_head_box_decode, and the DFL
                                    logits of foreground anchors are recomputed from the final box convolution's operands: */
  const void* box_in[4];         /* its input (B,H,W,box_in_ld) fp16, 64 channels */
  int box_in_ld[4];
  const float* box_w[4];         /* its fp32 weight (64,64) and bias (64) */
  const float* box_b[4];
  const float* box_in_coef[4];   /* NULL, or the BatchNorm coefficient table [4][64] of the Conv that produced box_in when box_in is that
                                    Conv's RAW output (no apply launch ran): BatchNorm + SiLU are applied where the rows are read */
} DyLossArgs;
int dy_loss_args_bytes(void); /* sizeof(DyLossArgs) in the library: bindings compare it with their own layout */
size_t dy_loss_workspace_bytes(int B, int A, int nmax);
/* byte offsets of pred_box (B,A,4 f32, grid units), assigned gt index (B,A i32, -1 = background) and target score
 * (B,A f32) inside the workspace after a call -- used by the parity tests */
int dy_loss_workspace_layout(int B, int A, int nmax, size_t* off_pred_box, size_t* off_asg_gt, size_t* off_tscore);
int dy_detection_loss(const DyLossArgs* args, hipStream_t stream);
/* Backward of Detect's final box convolution (nn/modules/head.py:38-40, Conv2d(c2, 4*reg_max, 1) with bias) from a gradient that has
 * ROWS: utils/loss.py:436-445 gives box / DFL terms to foreground anchors only, every other row of d(box logits) is zero.
 * assigned: the loss's per-anchor gt index (B, A) int32, -1 = background (workspace + off_asg_gt of dy_loss_workspace_layout);
 * this level's pixel (b, r) is anchor a0 + r.  Weight gradient: dy_conv1x1_rows_slabs(n, h, w) fp32 slabs [64][64] for
 * dy_wgrad_reduce_batched (descriptor: cin 64, cout 64, ks 1); bias gradient: fp64 sums into bias_acc [DY_BN_COPIES][64] (finished by
 * dy_wgrad_reduce_desc_bias); input gradient dx (may be NULL): W^T dy on foreground pixels, zeros (or untouched when dx_accumulate)
 * elsewhere.  Rows of dy whose anchor is background are never read.  Supported: cin == cout == 64. */
/* Forward of the same convolution fused with the loss's bbox_decode (utils/loss.py:347-354): pred_box (B, A, 4) in grid units, the
 * buffer at off_pred_box of the loss workspace; no logits are written (DyLossArgs.box_from_input). */
/* x_coef (both entries; may be NULL): x is the RAW output of the Conv below and x_coef its coefficient table [4][64] (scale, shift,
 * mean, invstd as dy_bn_act_apply_acc leaves them): BatchNorm + SiLU are applied on load -- what nn/modules/conv.py:49-55 computes,
 * without the launch and the tensor in between. */
int dy_head_box_decode(const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* pred_box, int A, int a0,
                       int n, int h, int w, int cin, int cout, hipStream_t stream);
/* Detect's final CLASS convolution (nn/modules/head.py:41-42, Conv2d(c3, nc, 1) with bias) for nc <= 8, cin 32 or 64, as stand-alone
 * kernels.  Forward: logits (npix, 8) fp32, channels >= nc written as 0.  x_coef as above (RAW input + the coefficient table of the
 * Conv in front, or NULL for an activated input).  Backward, one walk: weight gradient as dy_cls_head_slabs() fp32 slabs [16][cin] for
 * dy_wgrad_reduce_batched, bias gradient as fp64 sums in bias_acc [DY_BN_COPIES][8], input gradient dx = W^T dy (stored, or added
 * when dx_accumulate; may be NULL).  dy: (npix, 8) fp16. */
int dy_cls_head_supported(int cin, int nc);
int dy_cls_head_slabs(void);
int dy_cls_head_forward(const void* x, int ldx, const float* x_coef, const float* weight, const float* bias, float* logits, long npix,
                        int cin, int nc, hipStream_t stream);
int dy_cls_head_backward(const void* x, int ldx, const float* x_coef, const void* dy, const float* weight, void* dx, int lddx,
                         int dx_accumulate, float* slabs, double* bias_acc, long npix, int cin, int nc, hipStream_t stream);
/* The four head entries for SEVERAL detection levels in one launch each (nl <= 4; array arguments have nl entries; the same kernels,
 * blockIdx picks the level): the small levels' latency-bound workgroups run beside the large level's instead of after it. */
int dy_head_box_decode_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const float* const* weight,
                              const float* const* bias, float* pred_box, int A, const int* a0, int n, const int* h, const int* w, int cin,
                              int cout, hipStream_t stream);
int dy_conv1x1_rows_backward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const void* const* dy,
                                    const int* lddy, const int* assigned, int A, const int* a0, const float* const* weight, void* const* dx,
                                    const int* lddx, const int* dx_accumulate, float* const* slabs, double* const* bias_acc, int n,
                                    const int* h, const int* w, int cin, int cout, hipStream_t stream);
int dy_cls_head_forward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const float* const* weight,
                               const float* const* bias, float* const* logits, const long* npix, int cin, int nc, hipStream_t stream);
int dy_cls_head_backward_levels(int nl, const void* const* x, const int* ldx, const float* const* x_coef, const void* const* dy,
                                const float* const* weight, void* const* dx, const int* lddx, const int* dx_accumulate, float* const* slabs,
                                double* const* bias_acc, const long* npix, int cin, int nc, hipStream_t stream);
int dy_conv1x1_rows_supported(int cin, int cout);
int dy_conv1x1_rows_slabs(int n, int h, int w);
int dy_conv1x1_rows_backward(const void* x, int ldx, const float* x_coef, const void* dy, int lddy, const int* assigned, int A, int a0,
                             const float* weight, void* dx, int lddx, int dx_accumulate, float* slabs, double* bias_acc, int n,
                             int h, int w, int cin, int cout, hipStream_t stream);
/* TaskAlignedAssigner.forward utils/tal.py:39-88 as a call of its own (topk 10, alpha 0.5, beta 6.0: what v8DetectionLoss builds,
 * utils/loss.py:311): scores[l] (B,H,W,ncp) class PROBABILITIES (pd_scores re-laid per level), pd_boxes_grid (B,A,4) xyxy in grid
 * units (pd_bboxes / stride), gt_labels (B,n) int32, gt_bboxes (B,n,4) xyxy pixels, mask_gt (B,n) int32.  Out: asg_gt (B,A) int32,
 * the assigned gt slot or -1 (fg_mask / target_gt_idx), tscore (B,A) = the normalised alignment metric at the assigned class
 * (target_scores).  workspace: dy_loss_workspace_bytes(B, A, n) bytes. */
int dy_tal_assign(const float* const* scores, const int* H, const int* W, const float* stride, int nl, int B, int nc, int ncp,
                  int n, const float* pd_boxes_grid, const int* gt_labels, const float* gt_bboxes, const int* mask_gt, int* asg_gt,
                  float* tscore, void* workspace, hipStream_t stream);

/* ---- Detect inference decode nn/modules/head.py:50-74 (+ DFL nn/modules/block.py:52-55, dist2bbox utils/tal.py:310-318)
 *      -> y (B, 4+nc, A) fp32 [xywh pixels, sigmoid class scores] ------------------------------------------------- */
/* The whole inference tail of Detect in ONE launch (nn/modules/head.py:50-74 `_inference` with its two final convs head.py:38-42):
 * per level l the box conv Conv2d(64, 64, 1) over x_box[l] and the class conv Conv2d(cin_cls, nc, 1) over x_cls[l] (fp16 NHWC, the
 * activated outputs of cv2[l][1] / cv3[l][1]; fp32 master weights (cout, cin) + bias), DFL expectation, dist2bbox(xywh) * stride and
 * sigmoid, written straight as y (B, 4+nc, A): no fp32 logits are written or re-read.  Same bits as dy_conv_forward (fp32 out + bias)
 * followed by dy_decode_predictions.  dy_head_infer_supported: 64 -> 64 box conv, nc <= 80, cin_cls a multiple of 16 up to 128. */
int dy_head_infer_supported(int cin_box, int cout_box, int cin_cls, int nc);
int dy_head_infer_levels(int nl, const void* const* x_box, const int* ld_box, const float* const* w_box, const float* const* b_box,
                         const void* const* x_cls, const int* ld_cls, const float* const* w_cls, const float* const* b_cls, const int* h,
                         const int* w, const float* stride, int n, int cin_cls, int nc, float* y, hipStream_t stream);
int dy_decode_predictions(const float* const* box, const float* const* cls, const int* H, const int* W,
                          const float* stride, int nl, int B, int nc, int ncp, float* y, hipStream_t stream);
/* ---- ops.non_max_suppression utils/ops.py:292-427: candidate extraction (:344-392, order-preserving) ... */
int dy_nms_candidates(const float* pred, int B, int nc, int A, float conf, int multi_label, const int* classes,
                      int n_classes, float* cbox, float* cscore, float* ccls, int* ccount, int cap, hipStream_t stream);
/* ---- ... the candidate cap utils/ops.py:395-396 ``x = x[x[:, 4].argsort(descending=True)[:max_nms]]``: per image with more than
 *      max_nms candidates the max_nms most confident ones in descending confidence (ties in candidate order: the reference's
 *      unstable argsort leaves them unspecified); other images are copied through.  Outputs (B,max_nms,4) / (B,max_nms); count is
 *      clamped in place.  workspace: dy_nms_presort_workspace(B, max_nms) bytes. */
size_t dy_nms_presort_workspace(int B, int max_nms);
int dy_nms_presort(const float* cbox, const float* cscore, const float* ccls, int* count, int B, int cap, int max_nms, float* obox,
                   float* oscore, float* ocls, void* workspace, hipStream_t stream);
/* ---- ... and ops.soft_nms utils/ops.py:260-290 with bbox_iou_for_nms :162-199 (sequential semantics kept; scores are
 *      decayed in place; keep/nkeep receive the kept candidate indices per image) ---------------------------------- */
int dy_soft_nms(const float* boxes, float* scores, const float* cls, const int* count, int* order_a, int* order_b,
                int* keep, int* nkeep, int B, int cap, float iou_thr, float sigma, float score_thr, float class_offset,
                hipStream_t stream);

/* ---- BaseTrainer.optimizer_step engine/trainer.py:949-957 + build_optimizer groups :1146-1174 + ModelEMA.update
 *      utils/torch_utils.py:447-458 + GradScaler policy, over flat fp32 buffers.  mode: 0 SGD-nesterov, 1 Adam, 2 AdamW,
 *      3 RMSprop, 4 RAdam, 5 Adamax, 6 NAdam (betas = (momentum, 0.999) as build_optimizer :1159-1164 sets them). -------------- */
int dy_optimizer_step(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n, long g0_end,
                      long g1_end, const unsigned char* frozen, const float* buffers, float* ema_buffers,
                      long n_buffers, const float* hyper, float* state, float* partials, int mode, hipStream_t stream);
/* The same step over a flat buffer of SIX segments [bucket A: bias | decayed | norm][bucket B: bias | decayed | norm] (seg_ends: the ends
 * of the first five, host array): each data-parallel gradient bucket is then ONE contiguous slice that RCCL reduces in place -- DDP's
 * reducer works on contiguous buckets too (engine/trainer.py:694-695).  Segment k takes the hyper-parameters of group k % 3. */
int dy_optimizer_step_seg(float* params, const float* grads, float* mom, float* adam_v, float* ema, long n, const long* seg_ends,
                          const unsigned char* frozen, const float* buffers, float* ema_buffers, long n_buffers, const float* hyper,
                          float* state, float* partials, int mode, hipStream_t stream);
/* hyper (16 device floats read by dy_optimizer_step: lr per group, momentum, weight decay per group, EMA decay, max grad norm,
 * beta2, eps) set from 16 HOST floats that are copied at enqueue time (kernel arguments): safe however far the host runs ahead. */
int dy_set_hyper(float* hyper_dev, const float* host_values16, hipStream_t stream);
int dy_axpy_f32(float* y, const float* x, float a, long n, hipStream_t stream);

/* ---- validation: the per-batch half of DetectionValidator.update_metrics in ONE launch --
 *      _prepare_batch/_prepare_pred (models/yolo/detect/val.py:93-115: xywh2xyxy * imgsz, scale_boxes + clip_boxes),
 *      box_iou (utils/metrics.py:53-73) and BaseValidator.match_predictions (engine/validator.py:217-257, numpy path).
 *      preds (Ntot,6) x1 y1 x2 y2 conf cls packed over the batch, pred_off (B+1); targets as the collate function gives
 *      them (batch_idx, cls, xywh normalised); geom (B,5) = gain, padw, padh, ori_h, ori_w per image; iouv (niou <= 16);
 *      tp (Ntot,niou) uint8; predn (Ntot,6) native-space predictions or NULL; *status |= 1 if an image has > 1024 labels. */
int dy_match_predictions(const float* preds, const int* pred_off, const float* t_batch_idx, const float* t_cls,
                         const float* t_boxes, int n_targets, const float* geom, const float* iouv, int niou, int B,
                         int img_h, int img_w, unsigned char* tp, float* predn, int* status, hipStream_t stream);
/* utils/metrics.py:53-73 box_iou: (n,4) x (m,4) xyxy fp32 -> (n,m) */
int dy_box_iou(const float* box1, int n, const float* box2, int m, float* out, hipStream_t stream);

/* ---- two-stage ("double") inference, double_inference.py:98-305 ----------------------------------------------------
 * crop + resize + letterbox of K rectangles of one HBM-resident uint8 HWC image into a (K,S,S,3) batch
 * (prepare_cropped_image_cv2 :129-149; rects x1,y1,x2,y2 with exclusive x2/y2; geom new_w,new_h,pad_x,pad_y per crop). */
int dy_crop_letterbox_u8(const void* img, int H, int W, const int* rects, const int* geom, int K, int S, void* out,
                         hipStream_t stream);
/* per first-stage detection: map its crop's second-stage rows back (scale_boxes_vectorized :152-161) and pick the
 * replacement of process_refined_boxes_optimized :263-303.  dets: packed (x1,y1,x2,y2,conf,cls) rows, offsets (K+1),
 * orig (K,6), scale (K,3) ratio,pad_x,pad_y; out (K,6), found (K). */
int dy_refine_select(const float* dets, const int* offsets, const float* orig, const int* rects, const float* scale, int K,
                     float img_w, float img_h, float* out, int* found, hipStream_t stream);
/* greedy per-class NMS in descending score order, suppress IoU > iou_thr (torchvision_nms :164-203); n <= 2048; keep: n bytes */
int dy_nms_hard(const float* boxes, const float* scores, const float* labels, int n, float iou_thr, void* keep,
                hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif
