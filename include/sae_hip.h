/*
 * sae_hip.h — C-ABI of the MI355X (gfx950) hot-path library for the Swapping-Autoencoder
 * GAN training step.
 *
 * This is the drop-in boundary.  Every entry point replaces one native
 * interface of the reference (paths relative to the reference checkout):
 *
 *   sae_upfirdn2d_f32        <- upfirdn2d_op.upfirdn2d(...)   models/networks/stylegan2_op/upfirdn2d.cpp:12-19
 *                               (kernel: upfirdn2d_kernel.cu:52-137, host: :140-272)
 *   sae_bias_act_f32         <- fused.fused_bias_act(...)      models/networks/stylegan2_op/fused_bias_act.cpp:11-17
 *                               (kernel: fused_bias_act_kernel.cu:18-49, host: :52-99)
 *   sae_noise_bias_act_f32   <- NoiseInjection.forward + FusedLeakyReLU.forward   models/networks/stylegan2_layers.py:340-351, :54-65
 *   sae_noise_bias_act_bwd_f32, sae_plane_scale_dot_f32 <- the autograd graph of StyledConv's elementwise ops
 *                               (stylegan2_layers.py:280-286 style modulation, :340-351 noise, :54-65 bias + leaky-ReLU)
 *   sae_reflect_pad_f32, sae_reflect_pad_adj_f32 <- nn.ReflectionPad2d / F.pad(mode="reflect")   stylegan2_layers.py:57-63,100-105,643
 *   sae_random_crop_f32, sae_random_crop_bwd_f32 <- util.apply_random_crop's F.grid_sample   util/util.py:323-343
 *   sae_bias_act_bwd_f32     <- fused_bias_act(grad=1) followed by grad_input.sum(dim)
 *                               models/networks/stylegan2_op/fused_act.py:32-41 (one fused pass here)
 *   sae_conv2d_{fwd,dgrad,wgrad}_f32
 *                            <- F.conv2d / F.conv_transpose2d and their ATen backward
 *                               models/networks/stylegan2_layers.py:136,175,182,306,315,321
 *   sae_modconv2d_{fwd,dgrad,wgrad}_f32
 *                            <- ModulatedConv2d.forward (style scale, demodulation, conv / conv_transpose) and its
 *                               ATen backward   models/networks/stylegan2_layers.py:266-325
 *   sae_gemm_f32, sae_gemm_ws_f32 <- F.linear and its backward  models/networks/stylegan2_layers.py:177,186
 *   sae_upsample2x_bilinear_{add,bwd}_f32
 *                            <- F.interpolate(bilinear x2) + residual   models/networks/generator.py:51-53
 *   sae_l2_normalize_{,bwd_}f32 <- util.normalize                       util/util.py:18-22
 *   sae_plane_affine_{,bwd_}f32 <- GeneratorModulation.forward           models/networks/generator.py:62-67
 *   sae_softplus_mean_{,bwd_}f32 <- gan_loss                             models/networks/loss.py:10-16
 *   sae_adam_multi_f32, sae_adam_multi_dev_f32
 *                            <- torch.optim.Adam(...).step()            optimizers/swapping_autoencoder_optimizer.py:34-42,77,95,107
 *
 * Conventions (what the reference's pybind layer did implicitly is explicit here):
 *   - plain pointers and sizes only, no torch types; all tensors are dense fp32 in device memory
 *     of ONE device, the one `stream` belongs to;
 *   - the caller owns every buffer, including outputs and workspaces (no hidden allocation, so a
 *     caching allocator and stream ordering stay intact); workspace sizes come from the
 *     *_workspace() queries, which are pure host functions;
 *   - every call is asynchronous on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *     nothing here synchronises the device;
 *   - return value: 0 on success, a negative SAE_E* code otherwise; sae_last_error() returns a
 *     thread-local human-readable message for the last failing call on this thread.  Nothing
 *     throws, nothing exits;
 *   - calls are re-entrant; the library keeps no mutable state besides that thread-local message and ONE
 *     process-wide switch, the conv arithmetic of sae_set_conv_math() below (an atomic, meant to be chosen once
 *     at start-up: PyTorch runs backward on its own engine threads, so a per-thread setting would not reach the
 *     dgrad / wgrad calls; a conv call that races with a switch may see either value, and fails with
 *     SAE_EWORKSPACE rather than misbehaving if its workspace was sized for the other one).  The product build
 *     reads ONE environment variable, SAE_CONV_MATH (the initial value of that switch, once).  Kernel-selection
 *     knobs for A/B measurements and the recorded-experiment kernels exist only in builds with -DSAE_TUNING
 *     (tools/build_variant*.sh, tests/tuning, tests/emu); `strings libsae_hip.so | grep ^SAE_` shows the product has none.
 */
#ifndef SAE_HIP_H
#define SAE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SAE_ABI_VERSION 12  /* 12: sae_s2wino_* (stride-2 3x3 family on the polyphase minimal-filtering form);  11: sae_gemm_ws_f32 (K split across workgroups);  10: sae_adam_multi_dev_f32 (step counts in device memory: hipGraph replays);  9: sae_wino_fused_wgrad_*;  8: sae_wino_fused_* (one-kernel Winograd convolution);  7: sae_wino_* (Winograd F(2x2,3x3) transforms);  6: prepared weights (sae_conv2d_desc::prepped*, sae_conv2d_wprep_*);  2: modconv / adam / glue entry points (round 2), 3: sae_upfirdn2d_epilogue_f32, 4: sae_conv2d_fwd_residual_f32, 5: sae_weight_demod_*, sae_modconv2d_fwd_noise_bias_act_f32, sae_upfirdn2d_noise_bias_act_f32, sae_plane_scale_dot_act_f32 */

#define SAE_OK 0
#define SAE_EINVAL (-1)    /* bad argument (null pointer, non-positive size, unsupported mode) */
#define SAE_ELAUNCH (-2)   /* the HIP runtime reported a launch error */
#define SAE_EWORKSPACE (-3)/* workspace missing or too small */

typedef void* sae_stream_t; /* hipStream_t */

int sae_abi_version(void);
const char* sae_last_error(void);

/* Arithmetic of the 3x3 stride-1 convolution kernels (process-wide; initial value from the
 * environment variable SAE_CONV_MATH = "f32" | "bf16x6").  Inputs, outputs and accumulation are
 * fp32 in both modes.
 *   SAE_CONV_MATH_F32     v_mfma_f32_32x32x2_f32, bitwise an fp32 fma chain (default; what
 *                         F.conv2d computes in the reference, stylegan2_layers.py:136).
 *   SAE_CONV_MATH_BF16X6  every operand split exactly into three bf16 pieces, the six leading
 *                         cross products on v_mfma_f32_32x32x16_bf16 (dropped terms <= 2^-26
 *                         relative); same error class against fp64 as the fp32 chain.
 * sae_conv2d_workspace() depends on the mode: query it after switching. */
#define SAE_CONV_MATH_F32 0
#define SAE_CONV_MATH_BF16X6 1
int sae_set_conv_math(int32_t mode);
int sae_get_conv_math(void);

/* ------------------------------------------------------------------------------------------
 * upfirdn2d: zero-insertion upsample -> pad/crop -> 2-D FIR (true convolution) -> decimate.
 * x: [major, in_h, in_w, minor]   k: [kh, kw] (row-major, as passed by the caller, NOT flipped)
 * y: [major, out_h, out_w, minor] with out = (in*up + pad0 + pad1 - k + down) / down  (C division,
 * upfirdn2d_kernel.cu:167-168).  Negative pads crop.  Unlike the reference (which launches nothing
 * and returns uninitialised memory when no template matches, upfirdn2d_kernel.cu:172-268) every
 * (up, down, kh, kw) >= 1 is supported; kh*kw <= 1024.
 * ------------------------------------------------------------------------------------------ */
int sae_upfirdn2d_f32(const float* x, const float* k, float* y,
                      int64_t major, int64_t in_h, int64_t in_w, int64_t minor,
                      int32_t kh, int32_t kw,
                      int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                      int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1,
                      sae_stream_t stream);

/* upfirdn2d on planes (minor = 1, down = 1, up = 1 or 2 on both axes, at most 4 x 4 taps) with the elementwise work that
 * FOLLOWS it in the backward pass of a ResBlock (stylegan2_layers.py:672-693) done on its way out, so that the tensor
 * between the two never exists in HBM:
 *     v = upfirdn2d(x, k, up, pad)                                     exactly sae_upfirdn2d_f32's value
 *     if (accumulate) v += y                                           the sum of a forked gradient (autograd's add;
 *                                                                      the block input feeds conv1 AND the skip path)
 *     if (act_ref)  { v = (act_ref > 0 ? v : slope * v) * scale;       FusedLeakyReLUFunctionBackward, fused_act.py:32-41
 *                     gb[c] = sum over planes p with p % channels == c and all pixels of v }   (grad_bias, :36-41)
 *     y = v
 * act_ref: output-shaped (the saved output of the activation whose backward this is) or NULL; gb [channels] and the
 * workspace (sae_upfirdn2d_epilogue_workspace floats) are used with act_ref only; the bias-gradient reduction is
 * two-stage and fixed-order (deterministic, no atomics). */
int64_t sae_upfirdn2d_epilogue_workspace(int64_t major, int64_t out_h, int64_t out_w, int64_t channels, int32_t up);
int sae_upfirdn2d_epilogue_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                               int32_t kh, int32_t kw, int32_t up, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0,
                               int32_t pad_y1, const float* act_ref, float slope, float scale, float* gb,
                               int64_t channels, int32_t accumulate, float* workspace, int64_t workspace_floats,
                               sae_stream_t stream);

/* The blur that ends StyledConv's upsampling conv, followed by NoiseInjection and FusedLeakyReLU (stylegan2_layers.py:313-321,
 * :398-405, :340-351), in one kernel: up = down = 1, taps <= 4 x 4,
 *   y = lrelu((upfirdn2d(x, k, pad) + noise_weight[0] * noise[n][oy][ox]) + bias[c], slope) * scale,  plane p = n * channels + c
 * noise: [major / channels][out_h][out_w] or NULL; noise_weight: one float ON THE DEVICE; bias: [channels] or NULL. */
int sae_upfirdn2d_noise_bias_act_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                                     int32_t kh, int32_t kw, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1,
                                     const float* noise, const float* noise_weight, const float* bias, int64_t channels,
                                     float slope, float scale, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * bias_act: y[i] = act'(x[i] + b[(i / step_b) % size_b]) * scale       fused_bias_act_kernel.cu:18-49
 *   act: 1 = linear, 3 = leaky-ReLU(alpha);  grad: 0 = value, 1 = first derivative applied to x
 *   using sign(ref), 2 = second derivative (identically 0 for both activations).
 *   b may be NULL (no bias), ref may be NULL only when grad == 0.
 * ------------------------------------------------------------------------------------------ */
int sae_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                     int64_t numel, int64_t step_b, int64_t size_b,
                     int32_t act, int32_t grad, float alpha, float scale,
                     sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * StyledConv glue (generator), x / y / gy / gx: [outer][channels][hw] contiguous, hw % 4 == 0, 16-byte
 * aligned; noise: [outer][hw] (one map per sample, broadcast over channels) or NULL; noise_weight: one float
 * ON THE DEVICE (the NoiseInjection parameter, stylegan2_layers.py:334); bias: [channels] or NULL.
 *   sae_noise_bias_act_f32      y = lrelu((x + noise_weight * noise) + bias, alpha) * scale
 *   sae_noise_bias_act_bwd_f32  gx = (y_ref > 0 ? gy : alpha gy) * scale;  gbias[c] = sum_{n,hw} gx;
 *                               gnoise_weight[0] = sum gx * noise   (two-stage, fixed order; gbias /
 *                               gnoise_weight may be NULL; workspace: sae_noise_bias_act_bwd_workspace floats)
 *   sae_plane_scale_dot_f32     backward of y = x * s[plane] (style modulation of the conv input):
 *                               gx = g * s[plane], gs[plane] = sum_hw g * x;  g, x, gx: [planes][hw]
 * ------------------------------------------------------------------------------------------ */
int sae_noise_bias_act_f32(const float* x, const float* noise, const float* noise_weight, const float* bias,
                           float* y, int64_t outer, int64_t channels, int64_t hw, float alpha, float scale,
                           sae_stream_t stream);
int64_t sae_noise_bias_act_bwd_workspace(int64_t outer, int64_t channels, int64_t hw);
int sae_noise_bias_act_bwd_f32(const float* gy, const float* y_ref, const float* noise, float* gx, float* gbias,
                               float* gnoise_weight, float* workspace, int64_t workspace_floats, int64_t outer,
                               int64_t channels, int64_t hw, float alpha, float scale, sae_stream_t stream);
int sae_plane_scale_dot_f32(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t planes,
                            int64_t hw, sae_stream_t stream);
/* sae_plane_scale_dot_f32 followed by sae_noise_bias_act_bwd_f32 of the StyledConv that PRODUCED x (inside a generator block conv2's
 * input is conv1's activated output, generator.py:30-53): gs[n][c] = sum_hw g x;  gx = (x > 0 ? g s : alpha g s) * scale;
 * gbias[c] = sum_{n,hw} gx;  gnoise_weight[0] = sum gx * noise.  g, x, gx: [outer][channels][hw], s, gs: [outer][channels],
 * noise: [outer][hw] or NULL; gbias / gnoise_weight may be NULL.  Workspace: sae_plane_scale_dot_act_workspace floats. */
int64_t sae_plane_scale_dot_act_workspace(int64_t outer, int64_t channels);
int sae_plane_scale_dot_act_f32(const float* g, const float* x, const float* s, const float* noise, float* gx, float* gs,
                                float* gbias, float* gnoise_weight, float* workspace, int64_t workspace_floats, int64_t outer,
                                int64_t channels, int64_t hw, float alpha, float scale, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Demodulation factor of ModulatedConv2d and its backward (reference: models/networks/stylegan2_layers.py:290-292,
 *   weight = scale * weight;  demod = rsqrt(weight.pow(2).sum([2, 3, 4]) + 1e-8);  weight = weight * demod
 * computed from the un-modulated weight, `new_demodulation` :258).  w, geff, gw: [rows][cols] = [out channels][in * k * k].
 *   sae_weight_demod_f32      d[o] = rsqrt(sum_j (alpha w[o][j])^2 + eps)
 *   sae_weight_demod_bwd_f32  the weight gradient of a conv that ran on W_eff = alpha d[o] w, from geff = alpha dL/dW_eff
 *                             (what sae_modconv2d_wgrad_f32 returns):
 *                             gw[o][j] = d[o] geff[o][j] - (sum_j geff[o][j] w[o][j]) d[o]^3 alpha^2 w[o][j]
 * ------------------------------------------------------------------------------------------ */
int sae_weight_demod_f32(const float* w, float* d, int64_t rows, int64_t cols, float alpha, float eps, sae_stream_t stream);
int sae_weight_demod_bwd_f32(const float* geff, const float* w, const float* d, float* gw, int64_t rows, int64_t cols,
                             float alpha, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Random-crop sampler (patch discriminator input): crop k of image b = k / crops_per_image is
 *   y[k, ch, r, c] = bilinear(x[b, ch], gx, gy)   gx = (lin[c] * flip) * sx + ox,  gy = lin[r] * sy + oy
 * with F.grid_sample's conventions (align_corners = False: pixel = ((g + 1) * extent - 1) / 2, zero padding),
 * exactly what util.apply_random_crop builds (util/util.py:323-343).  params: [images * crops][5] =
 * {flip (+1 / -1), sx, sy, ox, oy}; lin: [size] = linspace(-1, 1, size); x: [images, channels, h, w];
 * y / gy: [images * crops, channels, size, size].  The backward is a deterministic gather (no atomics).
 * ------------------------------------------------------------------------------------------ */
int sae_random_crop_f32(const float* x, const float* params, const float* lin, float* y, int64_t images,
                        int64_t channels, int64_t h, int64_t w, int64_t crops_per_image, int64_t size,
                        sae_stream_t stream);
int sae_random_crop_bwd_f32(const float* gy, const float* params, const float* lin, float* gx, int64_t images,
                            int64_t channels, int64_t h, int64_t w, int64_t crops_per_image, int64_t size,
                            sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Reflection padding (no edge repeat, as nn.ReflectionPad2d) of [planes][h][w] and its adjoint (the sum of gy over
 * the padded positions that mirror onto each input pixel; a deterministic gather).  Pads must be < the image size.
 * ------------------------------------------------------------------------------------------ */
int sae_reflect_pad_f32(const float* x, float* y, int64_t planes, int64_t h, int64_t w, int32_t left, int32_t right,
                        int32_t top, int32_t bottom, sae_stream_t stream);
int sae_reflect_pad_adj_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w, int32_t left,
                            int32_t right, int32_t top, int32_t bottom, sae_stream_t stream);

/* Fused backward of the leaky-ReLU form: gx = (y_ref > 0 ? gy : alpha*gy) * scale and
 * gb[c] = sum over everything but the channel axis of gx (deterministic two-stage reduction,
 * no atomics).  workspace: sae_bias_act_bwd_workspace() floats. */
int64_t sae_bias_act_bwd_workspace(int64_t numel, int64_t step_b, int64_t size_b);
int sae_bias_act_bwd_f32(const float* gy, const float* y_ref, float* gx, float* gb,
                         float* workspace, int64_t workspace_floats,
                         int64_t numel, int64_t step_b, int64_t size_b,
                         float alpha, float scale, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Dense 2-D convolution family on the fp32 matrix cores (v_mfma_f32_32x32x2_f32), NCHW.
 * One descriptor describes the forward problem
 *     y[n,m,oy,ox] = alpha * sum_{c,ky,kx} w[m,c,ky,kx] * x[n,c,oy*stride+ky-pad, ox*stride+kx-pad]
 * (cross-correlation, zero padding), and the same descriptor is passed to dgrad / wgrad:
 *     dgrad: gx[n,c,iy,ix]   = alpha * sum_{m,ky,kx} w[m,c,ky,kx] * gy[n,m,oy,ox],  iy = oy*stride+ky-pad
 *     wgrad: gw[m,c,ky,kx]   = alpha * sum_{n,oy,ox} gy[n,m,oy,ox] * x[n,c,oy*stride+ky-pad, ...]
 * A stride-2 transposed convolution (F.conv_transpose2d, stylegan2_layers.py:306) is dgrad with
 * the roles of x and y exchanged.  Weight element (m,c,ky,kx) lives at
 *     w[m*w_stride_m + c*w_stride_c + ky*kw + kx]
 * so both [M,C,kh,kw] and [C,M,kh,kw] parameter layouts are addressed without a copy.
 * Supported: kh == kw in {1,3}; stride in {1,2}; 0 <= pad < kh.  oh/ow must equal
 * (h + 2*pad - kh)/stride + 1 (floor).
 * ------------------------------------------------------------------------------------------ */
typedef struct sae_conv2d_desc {
    int64_t n;              /* batch */
    int64_t c, h, w;        /* x side: channels, height, width */
    int64_t m, oh, ow;      /* y side: channels, height, width */
    int32_t kh, kw, stride, pad;
    int64_t w_stride_m, w_stride_c;
    /* optional (ABI 6): weights already re-laid for this launch by sae_conv2d_wprep_f32 (NULL / 0 / 0 = none; the launch then
     * re-lays them into the head of its workspace as before).  Used only when `prepped_layout` is the identity
     * sae_conv2d_wprep_query reports for THIS call -- prepared weights of another layout are ignored, never misread. */
    const float* prepped;
    int64_t prepped_floats;
    int64_t prepped_layout;
} sae_conv2d_desc;

#define SAE_CONV_FWD 0
#define SAE_CONV_DGRAD 1
#define SAE_CONV_WGRAD 2
/* floats of workspace needed by the given operation (op = SAE_CONV_*) */
int64_t sae_conv2d_workspace(const sae_conv2d_desc* d, int32_t op);

int sae_conv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d,
                       float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);
/* forward fused with the bias + leaky-ReLU that follows it in ConvLayer (stylegan2_layers.py:642-659):
 *   y = lrelu_{act_slope}(alpha * conv(x, w) + bias[m]) * act_scale          (bias may be NULL)   */
int sae_conv2d_fwd_bias_act_f32(const float* x, const float* w, const float* bias, float* y,
                                const sae_conv2d_desc* d, float alpha, float act_slope, float act_scale,
                                float* workspace, int64_t workspace_floats, sae_stream_t stream);

/* Forward conv with the residual merge of a ResBlock on its way out (stylegan2_layers.py:689, `(out + skip) / sqrt(2)`; the
 * skip path's 1x1 conv adds the main branch's output and scales):
 *     y = (alpha * conv(x, w) + residual) * res_scale          residual: y-shaped, 16-byte aligned, never y itself
 * Bit-identical to sae_conv2d_fwd_f32 followed by sae_add_scale_f32 (the fp32 accumulator is the value the separate call
 * would have stored).  1x1 convolutions only (kh = kw = 1, either stride).  Workspace: sae_conv2d_workspace(d, SAE_CONV_FWD). */
int sae_conv2d_fwd_residual_f32(const float* x, const float* w, const float* residual, float* y,
                                const sae_conv2d_desc* d, float alpha, float res_scale,
                                float* workspace, int64_t workspace_floats, sae_stream_t stream);
int sae_conv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d,
                         float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);
int sae_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                         float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Style-modulated convolution (ModulatedConv2d, models/networks/stylegan2_layers.py:266-325) as ONE call per
 * operation: the three conv operations above applied to
 *     x * x_scale[n][c],   gy * y_scale[n][m],   w * alpha * wm_scale[m] * wc_scale[c]
 * where every factor array may be NULL (= 1).  The activation factors (the per-sample style `s`, :280-286) are
 * applied while the operand is staged into LDS and the weight factors (the demodulation, :290-292) while the weight
 * is re-laid for the launch, so neither the modulated activation nor the modulated weight exists in HBM.  Indices
 * follow the DESCRIPTOR's axes (c = x side, m = y side), whatever the operation:
 *     plain modulated conv  O <- I    (d.m = O, d.c = I):  forward = fwd(x_scale = s, wm_scale = demod),
 *         input gradient before the style factor = dgrad(wm_scale = demod), weight gradient = wgrad(x_scale = s);
 *     transposed modulated conv (:302-309; the weight is used in its own [O][I] orientation, d.c = O, d.m = I):
 *         forward = dgrad(gy := input, y_scale = s, wc_scale = demod), input gradient = fwd(wc_scale = demod),
 *         weight gradient = wgrad(x := output gradient, gy := input, y_scale = s).
 * Activation factors are staged by the exact-fp32 kernels; under SAE_CONV_MATH_BF16X6 a call with x_scale / y_scale
 * returns SAE_EINVAL (modulate the activation first; weight factors work in both arithmetics).
 * Workspaces: sae_conv2d_workspace() of the same descriptor and operation.
 * ------------------------------------------------------------------------------------------ */
typedef struct sae_conv2d_mod {
    const float* x_scale;   /* [n][c] factor of the x-side activation (forward, wgrad) */
    const float* y_scale;   /* [n][m] factor of the y-side activation (dgrad, wgrad) */
    const float* wm_scale;  /* [m]    factor of the weight along the m axis (forward, dgrad) */
    const float* wc_scale;  /* [c]    factor of the weight along the c axis (forward, dgrad) */
} sae_conv2d_mod;
int sae_modconv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                          float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);
/* StyledConv's plain (stride-1) form in ONE kernel: ModulatedConv2d -> NoiseInjection -> FusedLeakyReLU
 * (stylegan2_layers.py:398-405, :340-351, fused_act.py:75-86):
 *   y = lrelu((alpha * conv(x * x_scale, w * wm_scale * wc_scale) + noise_weight[0] * noise[n][oy][ox]) + bias[m], act_slope) * act_scale
 * noise: [n][oh][ow] (one map per sample) or NULL; noise_weight: one float ON THE DEVICE; bias: [m] or NULL.  mod->x_scale is
 * required (this entry exists for the modulated kernels only); exact-fp32 arithmetic only (SAE_EINVAL under bf16x6). */
int sae_modconv2d_fwd_noise_bias_act_f32(const float* x, const float* w, const float* noise, const float* noise_weight,
                                         const float* bias, float* y, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                                         float alpha, float act_slope, float act_scale, float* workspace,
                                         int64_t workspace_floats, sae_stream_t stream);
int sae_modconv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                            float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);
int sae_modconv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                            float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Prepared weights.  Every forward / data-gradient launch first re-lays its weights for its tile shape ([tap][C_pad][M_pad],
 * `alpha`, the flip of the data gradient and the weight factors of `mod` folded in).  The layout depends on the descriptor's
 * geometry, not on the batch, and a parameter changes once per optimiser step (optimizers/swapping_autoencoder_optimizer.py:
 * 77,95,107), while it is used by 2 - 6 launches in between: a caller may keep the re-laid copy and hand it to the calls
 * through sae_conv2d_desc::prepped / prepped_floats / prepped_layout.
 *   sae_conv2d_wprep_query   floats (0: the launch has no weight layout, e.g. the streaming 1x1 kernels) and identity of the
 *                            layout the call (d, mod, op) would build; pure host function.  Two calls with equal identity,
 *                            equal `alpha`, equal weight and weight-factor VALUES can share one prepared buffer.
 *   sae_conv2d_wprep_f32     builds it into `out` (out_floats = the query's floats), asynchronously on `stream`.
 * op = SAE_CONV_FWD (also for the fused forward entry points) or SAE_CONV_DGRAD; `mod` as for the call itself (NULL = none).
 * The CALLER answers for freshness: a buffer prepared before the weights (or alpha, or the weight factors) changed holds the
 * old values -- the library cannot see an in-place update through a raw pointer.
 * ------------------------------------------------------------------------------------------ */
int sae_conv2d_wprep_query(const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, int64_t* floats, int64_t* layout);
int sae_conv2d_wprep_f32(const float* w, const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, float alpha,
                         float* out, int64_t out_floats, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Strided GEMM on the fp32 matrix cores:  C[i*ldc + j] = alpha * sum_k A[i*a_si + k*a_sk] * B[k*b_sk + j*b_sj]
 * (+ bias[j] when bias != NULL).  Serves F.linear forward (A = x, B = W^T), its dgrad and wgrad.
 * ------------------------------------------------------------------------------------------ */
int sae_gemm_f32(const float* a, const float* b, const float* bias, float* c,
                 int64_t m, int64_t n, int64_t k,
                 int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc,
                 float alpha, sae_stream_t stream);

/* The same product with the contraction split ACROSS workgroups for the skinny shapes of the linear layers (16 - 128 rows
 * against 2048 x 2048 ... 8192 x 512 weights: F.linear of the Dpatch pair MLP, the D head and the style projections,
 * models/networks/stylegan2_layers.py:177,186): K slices in a caller-owned workspace of sae_gemm_workspace(m, n, k) floats
 * (0: the shape takes sae_gemm_f32's one-launch kernels, workspace may be NULL), then one reduction launch that adds the slices
 * in order, scales and adds the bias.  Same result contract as sae_gemm_f32; the summation order differs (per slice). */
int64_t sae_gemm_workspace(int64_t m, int64_t n, int64_t k);
int sae_gemm_ws_f32(const float* a, const float* b, const float* bias, float* c,
                    int64_t m, int64_t n, int64_t k,
                    int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc,
                    float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Generator skip path: bilinear x2 upsampling (align_corners = false) fused with the residual add
 * and 1/sqrt(2) scale,  y[2h,2w] = alpha * (up2x(x[h,w]) + res[2h,2w])  (res may be NULL), and its
 * adjoint  gx[h,w] = alpha * up2x^T(gy[2h,2w]).  Replaces F.interpolate(scale_factor=2,
 * mode='bilinear', align_corners=False) + (skip + res) / sqrt(2), models/networks/generator.py:51-53.
 * Tensors are [planes, h, w] / [planes, 2h, 2w].
 * ------------------------------------------------------------------------------------------ */
int sae_upsample2x_bilinear_add_f32(const float* x, const float* res, float* y, int64_t planes,
                                    int64_t h, int64_t w, float alpha, sae_stream_t stream);
int sae_upsample2x_bilinear_bwd_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w,
                                    float alpha, sae_stream_t stream);

/* Residual merge y = alpha * (a + b): the (out + skip) / sqrt(2) of ResBlock (stylegan2_layers.py:689)
 * and of the generator's resolution-preserving block (generator.py:36) as one elementwise pass. */
int sae_add_scale_f32(const float* a, const float* b, float* y, int64_t numel, float alpha, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Small glue of the train step, one launch each way instead of a chain of ATen launches (csrc/glue.hip):
 *   sae_l2_normalize_f32      y = x * rsqrt(sum_c x^2 + eps) over the channel axis of [outer][channels][inner]
 *                             (util.normalize, util/util.py:18-22; inner = 1 for the [N, C] global code)
 *   sae_l2_normalize_bwd_f32  gx = r * gy - r^3 * <gy, x>_c * x,  r = rsqrt(sum_c x^2 + eps)
 *   sae_plane_affine_f32      y[p][i] = x[p][i] * a[p] + b[p] for p = (n, c) planes of hw elements
 *                             (GeneratorModulation, models/networks/generator.py:62-67)
 *   sae_plane_affine_bwd_f32  gx = g * a[p], ga[p] = sum_i g * x, gb[p] = sum_i g
 *   sae_softplus_mean_f32     y[b] = mean_i softplus(sign * x[b][i])   (gan_loss, models/networks/loss.py:10-16;
 *                             softplus with beta 1 and threshold 20 as F.softplus)
 *   sae_softplus_mean_bwd_f32 gx[b][i] = gy[b] * sign * sigmoid(sign * x[b][i]) / inner
 * ------------------------------------------------------------------------------------------ */
int sae_l2_normalize_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner, float eps,
                         sae_stream_t stream);
int sae_l2_normalize_bwd_f32(const float* gy, const float* x, float* gx, int64_t outer, int64_t channels, int64_t inner,
                             float eps, sae_stream_t stream);
int sae_plane_affine_f32(const float* x, const float* a, const float* b, float* y, int64_t planes, int64_t hw,
                         sae_stream_t stream);
int sae_plane_affine_bwd_f32(const float* g, const float* x, const float* a, float* gx, float* ga, float* gb,
                             int64_t planes, int64_t hw, sae_stream_t stream);
int sae_softplus_mean_f32(const float* x, float* y, int64_t batch, int64_t inner, float sign, sae_stream_t stream);
int sae_softplus_mean_bwd_f32(const float* gy, const float* x, float* gx, int64_t batch, int64_t inner, float sign,
                              sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Multi-tensor Adam step: the update torch.optim.Adam applies to every parameter of a group
 * (optimizers/swapping_autoencoder_optimizer.py:34-42 construct the two instances; :77,:95,:107 step them),
 * for `count` tensors in a handful of launches instead of four elementwise launches per tensor:
 *     g = grads[i] * grad_scale
 *     m = m + (g - m) * (1 - beta1)            v = v * beta2 + ((1 - beta2) * g) * g
 *     p = p - lr / (1 - beta1^step[i]) * m / (sqrt(v) / sqrt(1 - beta2^step[i]) + eps)
 * (amsgrad = False, weight_decay = 0, maximize = False: the configuration the reference uses).
 * params / grads / exp_avg / exp_avg_sq / numel / step are HOST arrays of `count` entries; the pointers they hold
 * are device pointers.  step[i] is the 1-based update count of tensor i AFTER this update (torch keeps it per
 * parameter: a parameter that received no gradient is skipped and its count does not advance).  grad_scale folds
 * the 1 / world_size of the gradient all-reduce into the update, so grads may point straight into the all-reduced
 * flat buckets.
 * ------------------------------------------------------------------------------------------ */
int sae_adam_multi_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                       const int64_t* numel, const int64_t* step, int64_t count, double lr, double beta1, double beta2,
                       double eps, double grad_scale, sae_stream_t stream);

/* The same update with the update counts kept in DEVICE memory: step_dev is a HOST array of `count` device pointers, each to
 * the int64 count of its tensor BEFORE this update (0 for a fresh parameter).  The kernels read it, form the two bias
 * corrections in double as the host form does, and a trailing launch adds 1 to every count.  Nothing about the call depends
 * on a host-side value that changes from step to step, so a hipGraph captured around a whole train step (the eager loop of
 * train.py:22-28 replayed as one graph launch) applies the right bias correction on every replay. */
int sae_adam_multi_dev_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                           const int64_t* numel, int64_t* const* step_dev, int64_t count, double lr, double beta1, double beta2,
                           double eps, double grad_scale, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) transforms for the 3x3 stride-1 convolutions (pad 1 or 0) (F.conv2d at models/networks/stylegan2_layers.py:136,
 * 315 -- the algorithm class the reference's cuDNN / MIOpen back end picks for these shapes).  csrc/winograd.hip has the
 * matrices.  The 16 products of the transform domain are 1x1 convolutions (c channels of tiles_h x tiles_w "pixels" each) on the
 * MFMA gather: sae_wino_gemm_f32.
 *   sae_wino_weights_f32   u[16][m][c] = (G g G^T) of g = alpha * w[m * w_stride_m + c * w_stride_c + tap] * row_scale[m] *
 *                          col_scale[c] (either factor may be NULL: the wm_scale / wc_scale of sae_conv2d_mod);  flip != 0: taps
 *                          reversed -- with the two strides (and factors) swapped by the caller that is the data gradient's filter
 *   sae_wino_input_f32     x [planes][h][w] (h, w even) with zero padding `pad` (1: the layer's "same" form; 0: valid; 2: the
 *                          data gradient of a valid layer) -> v [16][planes][(h + 2 pad - 2) / 2][(w + 2 pad - 2) / 2];
 *                          plane_scale: NULL or one factor per plane (x_scale / y_scale of sae_conv2d_mod)
 *   sae_wino_output_f32    md [16][planes][h/2][w/2] -> y [planes][h][w], times plane_scale[plane] if given; act != 0:
 *                          y = lrelu((y + noise_weight[0] * noise[plane / channels][pixel]) + bias[plane % channels], slope) *
 *                          act_scale (noise: [planes / channels][h][w] or NULL, noise_weight: one float on the device; bias may
 *                          be NULL) -- the epilogues of sae_conv2d_fwd_bias_act_f32 / sae_modconv2d_fwd_noise_bias_act_f32
 * Weight gradient on the same sixteen points (the transposition of the forward algorithm):
 *     gw = G^T [ sum over images and tiles of (A e A^T) o (B^T d B) ] G,    e: the 2x2 tile of the output gradient
 *   sae_wino_gy_f32            gy [planes][h][w] -> e [16][planes][h/2][w/2]  (plane_scale as in sae_wino_input_f32)
 *   sae_wino_wgrad_gemm_f32    gu[xi][m][c] = sum_{n,t} e[xi][n][m][t] v[xi][n][c][t]  (sixteen 1x1 weight gradients; v from
 *                              sae_wino_input_f32 on the layer's input; workspace: sae_wino_wgrad_gemm_workspace floats)
 *   sae_wino_wgrad_output_f32  gw[m * w_stride_m + c * w_stride_c + tap] = alpha * (G^T gu G)[tap]
 * Exact fp32; results differ from the direct kernels' by rounding only (another association of the same sums).
 *   sae_wino_gemm_f32      md[xi] = u[xi] v[xi] for the 16 xi: v [16][n][c][tiles_h][tiles_w], u [16][m][c],
 *                          md [16][n][m][tiles_h][tiles_w]; workspace: sae_wino_gemm_workspace floats
 */
int64_t sae_wino_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w);
int sae_wino_gemm_f32(const float* v, const float* u, float* md, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                      int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream);
int sae_wino_gy_f32(const float* gy, const float* plane_scale, float* e, int64_t planes, int64_t h, int64_t w, sae_stream_t stream);
int64_t sae_wino_wgrad_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w);
int sae_wino_wgrad_gemm_f32(const float* v, const float* e, float* gu, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                            int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream);
int sae_wino_wgrad_output_f32(const float* gu, float* gw, int64_t m, int64_t c, int64_t w_stride_m, int64_t w_stride_c, float alpha,
                              sae_stream_t stream);
int sae_wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* u, int64_t m, int64_t c,
                         int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha, sae_stream_t stream);
int sae_wino_input_f32(const float* x, const float* plane_scale, float* v, int64_t planes, int64_t h, int64_t w, int32_t pad,
                       sae_stream_t stream);
int sae_wino_output_f32(const float* md, const float* plane_scale, const float* noise, const float* noise_weight,
                        const float* bias, float* y, int64_t planes, int64_t channels, int64_t h, int64_t w, int32_t act,
                        float slope, float act_scale, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * Fused Winograd F(2x2, 3x3) convolution (csrc/winograd_fused.hip): the same layers (F.conv2d at models/networks/
 * stylegan2_layers.py:136,315; their data gradient with flip = 1, the two weight strides swapped and pad -> 2 - pad) with the
 * input transform, the sixteen products and the output transform + epilogue in ONE kernel: only x, the prepared weights and y
 * touch HBM.
 *   sae_wino_fused_weights_floats(m, c)   floats of the prepared weights: ceil(m / 64) * ceil(c / 8) * 8192
 *   sae_wino_fused_weights_f32   uf[((mb * chunks + chunk) * 16 + xi) * 512 + (half * 64 + ml) * 4 + s] = U[xi][64 mb + ml][8 chunk +
 *                                4 half + s], U as sae_wino_weights_f32's u (same factors, same flip), zero beyond m / c;
 *                                chunks = ceil(c / 8).  uf 16-byte aligned.  Prepared once per weight update.
 *   sae_wino_fused_conv_f32      x [n][c][h][w] (h, w even; zero padding `pad` 0, 1 or 2) times x_scale[n * c + ci] if given ->
 *                                y [n][m][h + 2 pad - 2][w + 2 pad - 2], times out_scale[n * m + mi] if given; act != 0:
 *                                y = lrelu((y + noise_weight[0] * noise[n][pixel]) + bias[mi], slope) * act_scale (noise
 *                                [n][oh][ow] or NULL, bias may be NULL).  No workspace.  y 8-byte aligned.
 */
int64_t sae_wino_fused_weights_floats(int64_t m, int64_t c);
int sae_wino_fused_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t m, int64_t c,
                               int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha, sae_stream_t stream);
int sae_wino_fused_conv_f32(const float* x, const float* x_scale, const float* uf, const float* out_scale, const float* noise,
                            const float* noise_weight, const float* bias, float* y, int64_t n, int64_t c, int64_t m, int64_t h,
                            int64_t w, int32_t pad, int32_t act, float slope, float act_scale, sae_stream_t stream);
/*   sae_wino_fused_wgrad_f32     the weight gradient of the same layers (the reference reaches it through autograd's
 *                                conv2d backward on stylegan2_layers.py:136,315), both transforms in registers:
 *                                gw[mi * w_stride_m + ci * w_stride_c + tap] = alpha * sum over images and pixels of
 *                                (gy * y_scale[n * m + mi]) (x) (x * x_scale[n * c + ci]) -- x [n][c][h][w], gy [n][m][h + 2 pad - 2]
 *                                [w + 2 pad - 2], pad 0 or 1, output rows a multiple of 16 pixels, gy 16-byte aligned; either
 *                                factor may be NULL.  Pixel slices are summed in slice order (deterministic).
 *   sae_wino_fused_wgrad_workspace   floats of workspace it needs (0: shape not supported)
 */
int64_t sae_wino_fused_wgrad_workspace(int64_t n, int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad);
int sae_wino_fused_wgrad_f32(const float* x, const float* x_scale, const float* gy, const float* y_scale, float* gw, int64_t n,
                             int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad, int64_t w_stride_m, int64_t w_stride_c,
                             float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream);

/* ------------------------------------------------------------------------------------------
 * The 3x3 STRIDE-2 family on its polyphase minimal-filtering form (csrc/s2wino.hip): the reference's
 * F.conv_transpose2d(stride 2) of ModulatedConv2d(upsample=True) (models/networks/stylegan2_layers.py:296-309) and the data
 * gradient autograd forms for F.conv2d(stride 2) of ConvLayer(downsample=True) (:136,627-648).  Along an axis the even positions
 * of the (2h+1)-long side meet the 2-tap filter (w0, w2) -- F(2,2): 3 products per 2 outputs -- and the odd positions the 1-tap
 * filter w1: 25 instead of 36 multiplications per 2x2 of small-side positions, transforms in registers, one kernel.
 *   sae_s2wino_weights_floats(cout, cin)  floats of the prepared weights (opaque layout; both point sets of the kernel)
 *   sae_s2wino_weights_f32    w[co * w_stride_out + ci * w_stride_in + tap] (co: the channel the PRODUCT writes, ci: the one it
 *                             contracts), times alpha, row_scale[co], col_scale[ci] if given; flip = 1: the taps reversed (the
 *                             data gradient / transposed convolution meets the reversed filter).  uf 16-byte aligned.
 *   sae_s2wino_dgrad_f32      g [n][cin][h][w] (h, w even, w >= 4) times g_scale[n * cin + ci] if given ->
 *                             dx [n][cout][2h+1][2w+1], dx[2 oy + ky][2 ox + kx] += w[ky][kx] g[oy][ox] (uf prepared with
 *                             flip = 1), times out_scale[n * cout + co] if given.  No workspace.
 */
int64_t sae_s2wino_weights_floats(int64_t cout, int64_t cin);
int sae_s2wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t cout, int64_t cin,
                           int64_t w_stride_out, int64_t w_stride_in, int32_t flip, float alpha, sae_stream_t stream);
int sae_s2wino_dgrad_f32(const float* g, const float* g_scale, const float* uf, const float* out_scale, float* dx, int64_t n,
                         int64_t cin, int64_t cout, int64_t h, int64_t w, sae_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* SAE_HIP_H */
