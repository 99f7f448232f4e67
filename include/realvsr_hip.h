/*
 * realvsr_hip.h -- C ABI of librealvsr_hip.so: the MI355X (gfx950) implementation of RealVSR's
 * EDVR alignment / fusion / reconstruction / pyramid-loss hot path.
 *
 * Conventions (all entry points):
 *   - plain device pointers (float32, NCHW, contiguous unless a batch stride is stated) + sizes;
 *     no torch / ATen types.  The caller owns every buffer; the library allocates nothing
 *     (temporaries come in through `workspace`) and keeps no state between calls EXCEPT one
 *     setting, the GEMM arithmetic: a process-wide default (rvsr_set_gemm_mode below; set it once
 *     before the first call) and a per-host-thread override (rvsr_set_gemm_mode_thread), both read
 *     by every conv / DCN entry point at call time on the calling thread -- so the reference's
 *     nn.DataParallel threads (VideoSR_AllPair_model_YCbCr_Split.py:35-36), which all CALL
 *     concurrently, each on its own device / stream, can each choose without racing
 *     (tests/test_gpu_misc.py::test_two_host_threads_two_streams, ::test_gemm_mode_per_thread).  The host
 *     mirror realvsr_amd.functional keeps two per-process caches on top (bf16 weight images keyed
 *     on the parameter object, per-layer DCN offset statistics); both are keyed per parameter
 *     object + device, so DataParallel replicas (distinct parameter objects) get their own entries.
 *   - `stream` is a hipStream_t (NULL = default stream).  Launches are asynchronous; the call is
 *     re-entrant across devices/streams (the caller sets the current device, as
 *     at::DeviceGuard does in the reference: deform_conv_cuda.cpp:499,581).
 *   - return value: 0 = ok, RVSR_ERR_* otherwise; rvsr_last_error() gives the message for the
 *     calling thread.  (The reference only printf()s launch errors, kernel.cu:794-798; shape
 *     errors there are c10::Error -> RuntimeError, deform_conv_cuda.cpp:497-516.)
 *
 * Section 1 is the drop-in for the reference's pybind module `deform_conv_cuda`
 * (codes/models/archs/dcn/src/deform_conv_cuda.cpp:687-701).  Sections 2-4 replace the ATen /
 * cuDNN calls the reference's Python makes on the same path (nn.Conv2d, F.interpolate, pooling,
 * pixel_shuffle, the TSA element-wise chain, utils/util.py pyramids, Charbonnier loss).
 */
#ifndef REALVSR_HIP_H
#define REALVSR_HIP_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

#define RVSR_OK 0
#define RVSR_ERR_UNSUPPORTED 1 /* valid request, not implemented on the HIP path (e.g. 5x5 DCN) */
#define RVSR_ERR_BAD_ARG 2     /* null pointer / inconsistent shapes                              */
#define RVSR_ERR_LAUNCH 3      /* HIP launch error                                                */
#define RVSR_ERR_WORKSPACE 4   /* workspace missing or too small                                  */

const char* rvsr_last_error(void);

/* GEMM arithmetic of the conv blocks: 0 (default) = 3-term bf16 split on the bf16 matrix cores
 * (a = a_hi + a_lo; a_hi*b_hi + a_hi*b_lo + a_lo*b_hi accumulated in f32; ~2^-17 relative error
 * per product), 1 = exact f32 MFMA (an fmaf chain; ~5x slower GEMMs).  Opt-in speed modes:
 * 2 = two terms (the weights -- in a weight gradient: the output gradient -- rounded to bf16, the
 * other operand hi + lo: exactly the result of mode 0 on bf16-rounded weights), 3 = one term (both
 * operands rounded to bf16); accumulation stays f32, ~2^-9 per product.  Modes 2 / 3 act in the
 * kernels that dominate a training step and ONLY there:
 *   conv_fwd5   3x3 stride-1 forward and data gradient, when C_out > 32 (64-row m-blocks) and
 *               the input view is one of the vector-staged ones (16-byte aligned, W % 4 == 0);
 *   conv_wgrad2 3x3 stride-1 weight gradient (W_out % 4 == 0, 16-byte aligned);
 *   dcn_fwd3    fused DCN forward with C_out in {64, 128} (3x3, C % 8 == 0);
 *   dcn_bwdin5 / dcn_bwdin6  fused DCN input / offset / mask gradient (3x3, stride 1, C % 8 == 0, C_out <= 128);
 *   dcn_bwdw4 / dcn_bwdw6    fused DCN weight gradient (same gate; bwdw4 also W_out % 4 == 0).
 * Every other kernel (1x1 and strided convs, narrow m-blocks such as the 3-channel output conv, scalar-staged views,
 * the generic DCN path of section 1c) computes three terms in modes 2 / 3, so a network off those shapes gets
 * mode 0's result and speed.  Other values select 0.  Process-wide switch. */
void rvsr_set_gemm_mode(int mode);
int rvsr_get_gemm_mode(void);   /* the mode the CALLING thread's next call computes in */
/* The same choice for the calling host thread only (every entry point reads the mode at call time on the calling thread): mode 0-3
 * overrides the process-wide setting for this thread's calls, any other value (-1) returns the thread to the process-wide setting.
 * This is the race-free way to select the arithmetic per call -- set it, call, set it back -- when several host threads drive the
 * library at once (the reference's nn.DataParallel replicas); rvsr_set_gemm_mode stays the default for threads that never ask. */
void rvsr_set_gemm_mode_thread(int mode);

/* ---------------------------------------------------------------------------------------------
 * 1. Modulated deformable convolution (DCNv2)
 * --------------------------------------------------------------------------------------------- */

/* Replaces modulated_deform_conv_cuda_forward (deform_conv_cuda.cpp:490-569) +
 * modulated_deformable_im2col_cuda (deform_conv_cuda_kernel.cu:571-633, 769-799).
 *   input (B,C,H,W)  weight (Co,C,kh,kw)  bias (Co) or NULL  offset (B,2*dg*kh*kw,Ho,Wo)
 *   mask (B,dg*kh*kw,Ho,Wo)  output (B,Co,Ho,Wo) -- fully overwritten.
 * The reference's `ones` / `columns` temporaries do not exist here (the column tile lives in LDS).
 * HIP path: kh = kw = 3, group = 1, isotropic stride/pad/dilation, C/dg dividing or a multiple of 8.
 * workspace: rvsr_modulated_deform_conv_forward_workspace_bytes(channels, channels_out) bytes for
 * the bf16 hi/lo re-packed weights of the bf16x3 kernel; NULL selects the exact-f32 kernel. */
size_t rvsr_modulated_deform_conv_forward_workspace_bytes(int channels, int channels_out);
int rvsr_modulated_deform_conv_forward(const float* input, const float* weight, const float* bias,
                                       const float* offset, const float* mask, float* output,
                                       int batch, int channels, int height, int width, int channels_out,
                                       int kernel_h, int kernel_w, int stride_h, int stride_w,
                                       int pad_h, int pad_w, int dilation_h, int dilation_w,
                                       int group, int deformable_group, int with_bias,
                                       void* workspace, size_t workspace_bytes, void* stream);

/* Replaces modulated_deform_conv_cuda_backward (deform_conv_cuda.cpp:571-685) + the col2im /
 * col2im_coord / im2col kernels (deform_conv_cuda_kernel.cu:571-767).
 *   grad_input (B,C,H,W): must be zero on entry (scatter-add, as the reference: deform_conv.py:127)
 *   grad_offset, grad_mask: overwritten.   grad_weight, grad_bias: accumulated into (+=), the
 *   caller zeroes them (deform_conv.py:130-131, cpp:659-671).
 *   grad_input/grad_offset/grad_mask may be NULL together (skip), grad_weight may be NULL (skip).
 *   workspace: rvsr_modulated_deform_conv_backward_workspace_bytes() bytes, needed iff grad_weight.
 *   SIZE: for stride 1 / dilation 1 / channels % 8 == 0 / channels_out <= 128 (what EDVR and TDAN instantiate) the workspace also holds the
 *   hand-off between the input-gradient and the weight-gradient kernel -- a bf16 hi + lo copy of grad_output in matrix-core operand order,
 *   batch x ceil8(Ho) x ceil32(Wo) x ceil64(channels_out) x 4 bytes (~575 MiB at 40 x 64 x 180 x 320) -- on top of the weight image and the
 *   weight-gradient partials (tens of MB).  Other geometries: partials only.
 *   The kernel for grad_input/grad_offset/grad_mask is chosen ON THE DEVICE from a sampled statistic of `offset` (no host
 *   synchronisation): one shared fixed-point LDS window per 8-channel chunk with a halo of 2 / 4 / 5 / 8 / 12 px; samples beyond the halo
 *   go through global gathers / atomics with the reference's rule set; all candidates give the same result up to the summation order of
 *   the scatter.  grad_weight / grad_bias / grad_offset / grad_mask are bit-reproducible run to run, grad_input is not (atomics, as the
 *   reference's col2im, kernel.cu:688). */
size_t rvsr_modulated_deform_conv_backward_workspace_bytes(int batch, int channels, int height, int width,
                                                           int channels_out, int stride, int pad, int dil);
int rvsr_modulated_deform_conv_backward(const float* input, const float* weight, const float* bias,
                                        const float* offset, const float* mask, float* grad_input,
                                        float* grad_weight, float* grad_bias, float* grad_offset,
                                        float* grad_mask, const float* grad_output,
                                        int batch, int channels, int height, int width, int channels_out,
                                        int kernel_h, int kernel_w, int stride_h, int stride_w,
                                        int pad_h, int pad_w, int dilation_h, int dilation_w,
                                        int group, int deformable_group, int with_bias,
                                        void* workspace, size_t workspace_bytes, void* stream);

/* 1b. DCNv1 -- the other three functions of the reference's pybind module (deform_conv_cuda.cpp:152-260 deform_conv_forward_cuda,
 * :262-374 deform_conv_backward_input_cuda, :376-488 deform_conv_backward_parameters_cuda; registration :687-701), same argument order
 * (kW, kH, dW, dH, padW, padH, dilationW, dilationH, group, deformable_group[, scale], im2col_step) with the tensors as plain pointers:
 *   input (B,C,H,W), offset (B, 2*dg*kH*kW, Ho, Wo), weight (Co, C, kH, kW), output / grad_output (B,Co,Ho,Wo); no bias, no mask.
 * The reference's `columns` / `ones` temporaries do not exist here (no column matrix); `im2col_step` has to divide the batch
 * (deform_conv.py:41) and is otherwise unused.  Same geometry limits as section 1 (3x3, group 1, isotropic stride / pad / dilation).
 * backward_input accumulates into the caller-zeroed grad_input and writes grad_offset; backward_parameters accumulates scale * gradient
 * into the caller-zeroed grad_weight (scale must be 1, what deform_conv.py:76 passes).
 * workspace: rvsr_deform_conv_workspace_bytes(...) bytes for all three. */
size_t rvsr_deform_conv_workspace_bytes(int batch, int channels, int height, int width, int channels_out, int kW, int kH, int dW,
                                        int padW, int dilationW, int deformable_group);
int rvsr_deform_conv_forward(const float* input, const float* weight, const float* offset, float* output, int batch, int channels,
                             int height, int width, int channels_out, int kW, int kH, int dW, int dH, int padW, int padH,
                             int dilationW, int dilationH, int group, int deformable_group, int im2col_step, void* workspace,
                             size_t workspace_bytes, void* stream);
int rvsr_deform_conv_backward_input(const float* input, const float* offset, const float* grad_output, float* grad_input,
                                    float* grad_offset, const float* weight, int batch, int channels, int height, int width,
                                    int channels_out, int kW, int kH, int dW, int dH, int padW, int padH, int dilationW,
                                    int dilationH, int group, int deformable_group, int im2col_step, void* workspace,
                                    size_t workspace_bytes, void* stream);
int rvsr_deform_conv_backward_parameters(const float* input, const float* offset, const float* grad_output, float* grad_weight,
                                         int batch, int channels, int height, int width, int channels_out, int kW, int kH, int dW,
                                         int dH, int padW, int padH, int dilationW, int dilationH, int group, int deformable_group,
                                         float scale, int im2col_step, void* workspace, size_t workspace_bytes, void* stream);

/* Fused core of ModulatedDeformConvPack.forward (deform_conv.py:274-292): `om` is the raw
 * (B,3*dg*9,Ho,Wo) output of conv_offset_mask; torch.chunk/torch.cat become addressing (channels
 * [0,2*dg*9) are the offsets, the rest mask logits) and torch.sigmoid runs in-kernel.
 * act: 0 none, 1 ReLU, 2 LeakyReLU(slope) applied to the output (EDVR_arch.py:107,130); act | 0x100: `workspace`
 * already holds this layer's packed weight image (rvsr_dcn_pack_weights / rvsr_pack_weights_batched), skip the per-call pack.
 * probe (NULL or 8 zeroed uint32 on the device): receives the sampled offset statistic (components beyond 2.5 .. 11.5 px) from
 * which the forward picks the halo of its LDS tile on the device (3 / 7 / 11 px, no host round trip); hand the same buffer to
 * rvsr_dcn_pack_backward, which then skips its own probe pass.  NULL: the halo in act bits 10..13 (3 / 7 / 11, chosen by the caller
 * from an earlier statistic, rvsr_dcn_offset_probe), else 3 px; out-of-tile samples gather from global memory in every case. */
int rvsr_dcn_offset_probe(const float* om, int batch, int height_out, int width_out, int deformable_group, void* probe, void* stream);
size_t rvsr_dcn_pack_weights(const float* weight, int channels, int channels_out, void* out, size_t out_bytes,
                             long long* desc, void* stream);
int rvsr_dcn_pack_forward(const float* input, const float* weight, const float* bias, const float* om,
                          float* output, int batch, int channels, int height, int width, int channels_out,
                          int stride, int pad, int dilation, int deformable_group, int act, float slope,
                          void* probe, void* workspace, size_t workspace_bytes, void* stream);
/* act_out (NULL or the saved activation output) fuses the activation derivative into the
 * grad_output load.  grad_om (B,3*dg*9,Ho,Wo) is overwritten (d/d logit for the mask part);
 * grad_input zero on entry; grad_weight/grad_bias accumulated.
 * probe: NULL, or the counters rvsr_dcn_pack_forward filled for the same `om`. */
int rvsr_dcn_pack_backward(const float* input, const float* weight, const float* om, const float* grad_output,
                           const float* act_out, float act_slope, float* grad_input, float* grad_weight,
                           float* grad_bias, float* grad_om, int batch, int channels, int height, int width,
                           int channels_out, int stride, int pad, int dilation, int deformable_group,
                           const void* probe, void* workspace, size_t workspace_bytes, void* stream);

/* 1c. The operator over its whole argument space: any kernel_h x kernel_w, anisotropic stride / padding / dilation, group >= 1, any
 * number of channels per deformable group, DCNv1 (mask == NULL) and DCNv2, element types f32 / f64 / f16 (dtype 0 / 1 / 2) -- what
 * modulated_deform_conv_cuda_forward / _backward and the three deform_conv_*_cuda functions accept (deform_conv_cuda.cpp:490-685,
 * 152-488; AT_DISPATCH_FLOATING_TYPES_AND_HALF, deform_conv_cuda_kernel.cu:781).  Sections 1 / 1b above are the fused f32 kernels for
 * the geometries the reference's architectures instantiate (3 x 3, isotropic); this is the general path behind the same Python
 * operator, organised as the reference organises it (per batch element: columns in the workspace + one GEMM per group).  All tensors
 * of a call have the element type `dtype`; arithmetic is f32 for f16 / f32 tensors and f64 for f64.
 *   backward: grad_input (zero on entry: scatter-add) and grad_offset are given together or both NULL; grad_mask NULL for DCNv1;
 *   grad_weight / grad_bias are accumulated into, NULL = skip. */
size_t rvsr_deform_conv_generic_workspace_bytes(int dtype, int channels, int height, int width, int kernel_h, int kernel_w,
                                                int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h, int dilation_w);
/* what the forward alone needs (one column buffer instead of two + the f16 scatter planes) */
size_t rvsr_deform_conv_generic_forward_workspace_bytes(int dtype, int channels, int height, int width, int kernel_h, int kernel_w,
                                                        int stride_h, int stride_w, int pad_h, int pad_w, int dilation_h,
                                                        int dilation_w);
int rvsr_deform_conv_generic_forward(int dtype, const void* input, const void* weight, const void* bias, const void* offset,
                                     const void* mask, void* output, int batch, int channels, int height, int width,
                                     int channels_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                     int dilation_h, int dilation_w, int group, int deformable_group, void* workspace,
                                     size_t workspace_bytes, void* stream);
int rvsr_deform_conv_generic_backward(int dtype, const void* input, const void* weight, const void* offset, const void* mask,
                                      const void* grad_output, void* grad_input, void* grad_offset, void* grad_mask,
                                      void* grad_weight, void* grad_bias, int batch, int channels, int height, int width,
                                      int channels_out, int kernel_h, int kernel_w, int stride_h, int stride_w, int pad_h, int pad_w,
                                      int dilation_h, int dilation_w, int group, int deformable_group, void* workspace,
                                      size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 2. Convolution blocks (nn.Conv2d 3x3 / 1x1, padding = ksize/2) with fused neighbours
 *    (EDVR_arch.py:96-132, 166-208, 256-319; arch_util.py:121-139)
 * --------------------------------------------------------------------------------------------- */

/* out = act(conv(cat(x1, x2)) + bias) [+ residual]
 *   x1 (B,C1,..), x2 (B,C2,..) or NULL/0: torch.cat([x1,x2],1) without materialising it.
 *   in_mode 0: x1 is (B,C1,Hs,Ws).
 *   in_mode 1: x1 is read through a zero-inserted x2 view of virtual size (Hout,Wout)
 *              (data gradient of a stride-2 conv).
 *   in_mode 2: x1 is stored (B,C1/4,Hs,Ws) and read pixel-unshuffled as (B,C1,Hs/2,Ws/2)
 *              (gradient arriving through nn.PixelShuffle(2), EDVR_arch.py:311-312).
 *   xact (NULL or a tensor stored like x1): value *= (xact > 0 ? 1 : xact_slope), i.e. the
 *              ReLU / LeakyReLU derivative taken from the saved activation output.
 *   weight: w_mode 0 -> (Co, C1+C2, k, k) used as is;
 *           w_mode 1 -> (C1+C2, Co, k, k) used transposed + spatially flipped (data gradient);
 *           w_mode | 2: `workspace` already holds the packed image of these weights (section 2b), the per-call pack is skipped.
 *           w_mode | 4 (opt-in, with w_mode & 1 == 0, ksize 3, stride 1, Co1+Co2 > 32, in_mode 0, xact NULL, Ws % 4 == 0, 16-byte aligned
 *               inputs, a second input only behind a multiple of 16 channels; RVSR_ERR_UNSUPPORTED otherwise): this forward conv forms its
 *               products in the f16 + fp8 format -- a1*b1 on v_mfma_f32_32x32x16_f16 plus the cross terms a1*b2 + a2*b1 on
 *               v_mfma_scale_f32_32x32x64_f8f6f4 (a1 = f16(a), a2 = a - a1, fp8 e4m3 with the 2^12 in the block scales): 56 instead of
 *               108 matrix instructions per 16-channel stage, ~1.2e-5 instead of ~4.6e-6 relative l2 error per convolution (f32: 3e-7).
 *               Activations must lie inside f16's range (|x| < 65504; below 6e-8 they round to zero).  A packed image (w_mode | 2) must
 *               have been written with the same flag.  realvsr_amd.set_gemm_mode('f16fp8') sets it on every eligible forward conv.
 *   out1 (B,Co1,Hout,Wout) [+ out2 (B,Co2,Hout,Wout): rows split, for the gradient of a cat].
 *   residual (NULL or shaped like out1, out2 must be NULL): added after the activation.
 *   act: 0 none, 1 ReLU, 2 LeakyReLU(slope); 3 (with `residual`, ksize 3, stride 1, one output): out = (conv + bias) * (residual > 0 ? 1 : slope),
 *        i.e. a data gradient multiplied by the derivative of the activation whose saved OUTPUT `residual` is -- the consumers of that
 *        gradient then need no mask.  Built for the 8 x 64-tile kernel only: returns RVSR_ERR_UNSUPPORTED (without an error message) for
 *        frames that kernel does not take, and the caller applies the mask on the consumer side (xact / gout_act) instead.
 *   pixel_shuffle 1: out1 is (B,Co/4,2*Hout,2*Wout), written through PixelShuffle(2).
 *   stride 2 only with ksize 3 and in_mode 0.
 *   workspace: rvsr_conv2d_forward_workspace_bytes(C1, C2, Co1+Co2, ksize) bytes (holds the weights
 *   re-packed as bf16 hi/lo for the matrix cores; unused in exact-f32 mode).
 *   Sizes: the fast kernels address one batch element of a tensor with 32-bit byte offsets (raw
 *   buffers); an input whose C*Hs*Ws*4 bytes reach 2 GB, or a concat whose first input is not a
 *   multiple of 16 channels, takes the scalar-staging / earlier-generation kernels (same results).
 *   3x3 stride-1 layers with Co <= 4 (conv_last) run on the vector ALU in exact f32 in both modes. */
size_t rvsr_conv2d_forward_workspace_bytes(int C1, int C2, int Co, int ksize);
int rvsr_conv2d_forward(const float* x1, int C1, const float* x2, int C2, const float* xact, float xact_slope,
                        int in_mode, int Hs, int Ws, const float* weight, const float* bias,
                        const float* residual, float* out1, int Co1, float* out2, int Co2, int B, int ksize,
                        int stride, int w_mode, int act, float slope, int pixel_shuffle, int Hout, int Wout,
                        void* workspace, size_t workspace_bytes, void* stream);

/* 2b. Packed weight images, once per optimizer step.  The matrix-core kernels stage weights as bf16 hi/lo images
 *   ([m-block][chunk][hi|lo][tap][octet][row][8]); rvsr_conv2d_forward builds that image in its workspace on every call.
 *   rvsr_conv2d_pack_weights / rvsr_dcn_pack_weights write the image of one layer (forward: w_mode 0, data gradient: w_mode 1; w_mode 4:
 *   the forward image of a 3x3 layer with Co > 32 in the f16 + fp8 format of rvsr_conv2d_forward's w_mode | 4, for stride-1 use only) to
 *   caller-owned memory of rvsr_conv2d_forward_workspace_bytes(C_in, 0, Co, ksize) /
 *   rvsr_modulated_deform_conv_forward_workspace_bytes(channels, channels_out) bytes and return that size (0 = bad argument);
 *   `desc` (NULL or 10 x long long, host; 20 x long long for rvsr_dcn_pack_weights) receives {weight, out, Co, C_in, taps, MP, CCG,
 *   nchunks, nmb, mode}.  The DCN forward keeps TWO images in that buffer -- the tap-major one of dcn_fwd2 / dcn_fwd3 (mode 0)
 *   and, behind it, the k-step-major one of dcn_fwd4 (mode 2; desc[10..19], all zero where that kernel does not apply: C % 16 != 0).
 *   rvsr_pack_weights_batched re-packs n images in ONE launch from a DEVICE table of 48-byte records
 *   {const float* w; void* out; int Co, C_in, taps, MP, CCG, nchunks, nmb, mode;} built from those descriptors: the host
 *   (realvsr_amd.functional.PackedWeights) calls it once after the optimizer has updated the parameters in place. */
size_t rvsr_conv2d_pack_weights(const float* weight, int C_in, int Co, int ksize, int w_mode, void* out, size_t out_bytes,
                                long long* desc, void* stream);
int rvsr_pack_weights_batched(const void* descs, int n, void* stream);

/* grad_weight (Co,C1+C2,k,k) and grad_bias (Co) (NULL = skip) of the conv above.
 *   gout: gradient w.r.t. the conv output.  g_mode 0: stored (B,Co,Gs_h,Gs_w) = (.., Hout, Wout);
 *         g_mode 2: stored (B,Co/4,Gs_h,Gs_w) pixel-shuffled (2*Hout, 2*Wout).
 *   gact/gact_slope: fused activation derivative (stored like gout), or NULL.
 *   accumulate 0: overwrite, 1: += .  Deterministic (fixed-order reduction of partials). */
size_t rvsr_conv2d_wgrad_workspace_bytes(int C1, int C2, int Co, int B, int ksize, int stride, int Hout, int Wout);
int rvsr_conv2d_backward_weight(const float* x1, int C1, const float* x2, int C2, int Hin, int Win,
                                const float* gout, const float* gact, float gact_slope, int g_mode,
                                int Gs_h, int Gs_w, float* grad_weight, float* grad_bias, int Co, int B,
                                int ksize, int stride, int Hout, int Wout, int accumulate,
                                void* workspace, size_t workspace_bytes, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 3. Fusion / resampling element-wise chain
 * --------------------------------------------------------------------------------------------- */

/* scale * F.interpolate(x, scale_factor=factor, mode='bilinear', align_corners=False), factor 2|4
 * (EDVR_arch.py:111-112,115,120-121,124,194,200,316).  in: `planes` images of HxW. */
int rvsr_upsample_bilinear_forward(const float* in, float* out, size_t planes, int H, int W, int factor,
                                   float scale, void* stream);
int rvsr_upsample_bilinear_backward(const float* gout, float* gin, size_t planes, int H, int W, int factor,
                                    float scale, void* stream);

/* cat([MaxPool2d(3,2,1)(x), AvgPool2d(3,2,1)(x)], 1) (EDVR_arch.py:154-155,188-194):
 * in (B,C,H,W) -> out (B,2C,Ho,Wo), argmax (B,C,Ho,Wo) uint8 saved for the backward. */
int rvsr_maxavgpool_forward(const float* in, float* out, unsigned char* argmax, int B, int C, int H, int W,
                            void* stream);
int rvsr_maxavgpool_backward(const float* gout, const unsigned char* argmax, float* gin, int B, int C, int H,
                             int W, void* stream);

/* TSA temporal attention (EDVR_arch.py:171-181): prob[b,n] = sigmoid(sum_c emb[b,n,c]*emb_ref[b,c]);
 * mod[b,n,c] = aligned[b,n,c] * prob[b,n].  mod (B,N,C,H,W), emb_ref (B,C,H,W), prob (B,N,H,W);
 * emb / aligned (and galigned / gemb): (B,N,C,H,W) as the reference stacks them (frame_major 0), or (N,B,C,H,W) =
 * the frame-major batch the alignment stage works on (frame_major 1: no transposing copy between the two stages). */
int rvsr_tsa_temporal_forward(const float* emb, const float* emb_ref, const float* aligned, float* mod,
                              float* prob, int B, int N, int C, int H, int W, int frame_major, void* stream);
int rvsr_tsa_temporal_backward(const float* gmod, const float* emb, const float* emb_ref, const float* aligned,
                               const float* prob, float* galigned, float* gemb, float* gemb_ref, int B, int N,
                               int C, int H, int W, int frame_major, void* stream);

/* out = fea * sigmoid(att) * 2 + att_add (EDVR_arch.py:204-207); grad w.r.t. att_add is g itself. */
int rvsr_tsa_output_forward(const float* fea, const float* att, const float* att_add, float* out, size_t n,
                            void* stream);
int rvsr_tsa_output_backward(const float* g, const float* fea, const float* att, float* gfea, float* gatt,
                             size_t n, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 4. Laplacian pyramid decomposition + Charbonnier loss (utils/util.py:491-554, loss.py:10-23)
 * --------------------------------------------------------------------------------------------- */

/* downsample(conv_gauss(x)): reflect-pad 2, 5x5 binomial /256, keep even rows/cols.
 * in: `planes` images HxW -> out ceil(H/2) x ceil(W/2).  (utils/util.py:503-510) */
int rvsr_pyr_down_forward(const float* in, float* out, size_t planes, int H, int W, void* stream);
int rvsr_pyr_down_backward(const float* gout, float* gin, size_t planes, int H, int W, void* stream);
/* out = cur - upsample(down): zero-insert at even positions, reflect-pad 2, 5x5 binomial * 4/256
 * (utils/util.py:513-516, 548-551).  H, W even; down is (H/2, W/2).
 * backward: grad_cur = gout (identity), grad_down computed here. */
int rvsr_pyr_updiff_forward(const float* cur, const float* down, float* out, size_t planes, int H, int W,
                            void* stream);
int rvsr_pyr_updiff_backward(const float* gout, float* gdown, size_t planes, int H, int W, void* stream);

/* out[0] = scale * sum(sqrt((x-y)^2 + eps))  (scale = 1/n for reduction='mean'). */
size_t rvsr_charbonnier_workspace_bytes(void);
int rvsr_charbonnier_forward(const float* x, const float* y, size_t n, float eps, double scale, float* out,
                             void* workspace, void* stream);
/* gx = gscalar[0] * scale * (x-y)/sqrt((x-y)^2+eps); gscalar is a device pointer (no host sync). */
int rvsr_charbonnier_backward(const float* x, const float* y, const float* gscalar, float scale, float eps,
                              float* gx, size_t n, void* stream);

/* GWLoss (codes/models/loss.py:54-80): out[0] = scale * sum (1 + w|Sx(d)|)(1 + w|Sy(d)|)|d|, d = x1 - x2, depthwise 3x3 Sobel
 * with zero padding.  fa/fbx/fby (NULL together, or three buffers shaped like x1) receive the per-pixel factors the
 * backward needs.  workspace: rvsr_charbonnier_workspace_bytes().  backward: gx1 = gscalar[0]*scale * dL/dx1 (dL/dx2 = -gx1). */
int rvsr_gwloss_forward(const float* x1, const float* x2, size_t planes, int H, int W, float w, double scale, float* out,
                        float* fa, float* fbx, float* fby, void* workspace, void* stream);
int rvsr_gwloss_backward(const float* fa, const float* fbx, const float* fby, const float* gscalar, float scale, float* gx,
                         size_t planes, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 5. Test-time output conversion
 * --------------------------------------------------------------------------------------------- */

/* ycc: planar f32 [3][H][W] network output (Y, Cb, Cr in [0, 1] nominal) -> bgr: uint8 [H][W][3].
 * Replaces the host-side numpy chain of codes/test_RealVSR_wi_GT.py:122-123
 * (utils/util.py:151-181 tensor2img(float32, reverse_channel=False) -> data/util.py:397-416 ycbcr2bgr ->
 * clip, *255, round, uint8), same arithmetic order and precisions: bit-exact. */
int rvsr_ycbcr_to_bgr_u8(const float* ycc, unsigned char* bgr, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 6. The rest of the reference's pixel criteria (codes/models/loss.py) and pyramid helpers
 * --------------------------------------------------------------------------------------------- */

/* Bytes of the partial-sum workspace the reductions below need. */
size_t rvsr_reduce_workspace_bytes(void);

/* out[0] = scale * sum f(x - y) over n elements; mode 0: |d| (nn.L1Loss, loss.py:167-168), 1: d^2 (nn.MSELoss, :169-170),
 * 2: Huber with delta = param (HuberLoss, loss.py:26-40), 3: sqrt(d^2 + param) (CharbonnierLoss, loss.py:10-23).
 * backward: gx = gscalar[0] * scale * f'(x - y)  (grad w.r.t. y is -gx); gscalar is a device pointer. */
int rvsr_pixel_loss_forward(const float* x, const float* y, size_t n, int mode, float param, double scale, float* out,
                            void* workspace, void* stream);
int rvsr_pixel_loss_backward(const float* x, const float* y, const float* gscalar, int mode, float param, float scale,
                             float* gx, size_t n, void* stream);

/* SSIM loss as LapPyrLoss(lf_mode='ssim') uses it (loss.py:203,209,222 -> IQA_pytorch.SSIM(channels=1)(x, y, as_loss=True);
 * third-party, un-vendored: algorithm restated in oracle/ssim_oracle.py, parity unpinned): 11x11 Gaussian window sigma 1.5,
 * 'valid' correlation, C1 = 0.01^2, C2 = 0.03^2, contrast-structure map clamped at 0, out[0] = 1 - scale * sum(ssim_map)
 * with scale = 1 / (planes (H-10) (W-10)).  ga/gb/gc (NULL together, or planes x (H-10) x (W-10) each) receive
 * dS/dmu_x, dS/dE[xx], dS/dE[xy] for the backward: gx = -gscalar[0] * scale * window-adjoint(ga + 2 x gb + y gc). */
int rvsr_ssim_forward(const float* x, const float* y, size_t planes, int H, int W, double scale, float* out, float* ga,
                      float* gb, float* gc, void* workspace, void* stream);
int rvsr_ssim_backward(const float* x, const float* y, const float* ga, const float* gb, const float* gc,
                       const float* gscalar, float scale, float* gx, size_t planes, int H, int W, void* stream);

/* conv_gauss(img, gain * gauss_kernel) (utils/util.py:503-506): reflect-pad 2 + depthwise 5x5 binomial /256, same size. */
int rvsr_conv_gauss_forward(const float* in, float* out, size_t planes, int H, int W, float gain, void* stream);
int rvsr_conv_gauss_backward(const float* gout, float* gin, size_t planes, int H, int W, float gain, void* stream);
/* upsample(x) (utils/util.py:513-516): zero-insert to 2H x 2W, conv_gauss with 4 * kernel.  in planes x H x W. */
int rvsr_pyr_upsample_forward(const float* in, float* out, size_t planes, int H, int W, void* stream);
int rvsr_pyr_upsample_backward(const float* gout, float* gin, size_t planes, int H, int W, void* stream);

/* ---------------------------------------------------------------------------------------------
 * 7. The two steps either side of netG in optimize_parameters
 * --------------------------------------------------------------------------------------------- */

/* One torch.optim.Adam update (VideoSR_AllPair_model_YCbCr_Split.py:122-124,187) of n contiguous f32 parameters, in place:
 *   g' = grad + weight_decay * param; exp_avg += (g' - exp_avg)(1 - beta1); exp_avg_sq = beta2 exp_avg_sq + (1 - beta2) g'^2;
 *   param -= step_size * exp_avg / (sqrt(exp_avg_sq) / bias_correction2_sqrt + eps)
 * with step_size = lr / (1 - beta1^t), bias_correction2_sqrt = sqrt(1 - beta2^t) computed by the caller. 16-byte aligned buffers. */
int rvsr_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, size_t n, float step_size, float beta1,
                   float beta2, float eps, float weight_decay, float bias_correction2_sqrt, void* stream);

/* CutBlur / rgb-permute / blend of a clip pair (data/augments_video_allpair.py:38-88) in one pass over `frames` x 3 x H x W:
 *   a = im1[f, perm[c]], b = im2[f, perm[c]];  out1 = a;  out2 = b            (box_mode 0)
 *                                              out2 = inside box ? a : b      (box_mode 1: im1's box pasted into im2)
 *                                              out2 = inside box ? b : a      (box_mode 2: im2's box pasted into a copy of im1)
 *   colour != NULL (frames x 3): out = v * out + (1 - v) * colour[f, c]        (blend)
 * The random decisions (which augmentation, box, permutation, v) are drawn by the caller in the reference's host-RNG order. */
int rvsr_augment_clips(const float* im1, const float* im2, float* out1, float* out2, const float* colour, size_t frames, int H,
                       int W, int perm0, int perm1, int perm2, int box_mode, int y0, int y1, int x0, int x1, float v,
                       void* stream);

/* The algebraic split of the PCD "concat with the repeated reference" convs (EDVR_arch.py:100-101,109,118,127: conv(cat([nbr, ref]))
 * with ref identical for the N frames of a window): conv(cat(x, repeat(ref))) = conv_a(x) + conv_b(ref).  These two kernels are the
 * elementwise glue: a[n][j] = act(a[n][j] + b[j]) in place (n < N repeats of `per` floats; act 0 none, 1 ReLU, 2 LeakyReLU(slope)),
 * and its adjoint gb[j] = sum_n gout[n][j] * act'(out[n][j]) (out NULL: no activation). */
int rvsr_bcast_add_act(float* a, const float* b, size_t per, int N, int act, float slope, void* stream);
int rvsr_bcast_reduce_act(const float* gout, const float* out, float* gb, size_t per, int N, float gslope, void* stream);

/* Measurement aid, not part of the reference's interface (bench.py: roofline_conv.sustained_peak): `workgroups` x 8 waves loop `iters`
 * times over 8 register-resident v_mfma_f32_32x32x16_bf16 whose operands come from `ops` (8 x 512 x 16 B of bf16: [operand][thread][8]);
 * out: workgroups x 512 floats (checksums).  MFMA work issued = workgroups * 8 waves * iters * 8 * 32768 FLOP. */
int rvsr_debug_mfma_rate(const void* ops, float* out, int workgroups, int iters, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* REALVSR_HIP_H */
