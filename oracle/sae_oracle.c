/*
 * sae_oracle.c — CPU restatement of the reference's algorithms for the hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
 * leg may load this library, and only as the checker / reported baseline — never as the thing
 * shipped or measured as the product.  The product path (swapping_autoencoder_pytorch_amd)
 * must fail loudly when the HIP library is missing; it never falls back to this file.
 *
 * Parity pin: the reference ships no tests or golden vectors.  This
 * restatement is pinned against outputs of the reference's own Python modules run in the build
 * container (native-fallback path, upfirdn2d.py:162-222, fused_act.py:93-96,
 * stylegan2_layers.py) — fixtures under tests/golden/, generator tests/golden/make_golden.py —
 * and checked in tests/test_reference_parity.py.
 *
 * Each function carries the same signature as its sae_* counterpart in include/sae_hip.h
 * (host pointers; the stream argument is ignored) and cites the reference lines it follows.
 * Accumulation is in double so the oracle is a strictly better approximation of the real-number
 * result than either fp32 implementation (the reference's or ours).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <stdio.h>
#include <math.h>

#define SAE_OK 0
#define SAE_EINVAL (-1)
#define SAE_EWORKSPACE (-3)

typedef void* sae_stream_t;

typedef struct sae_conv2d_desc {
    int64_t n;
    int64_t c, h, w;
    int64_t m, oh, ow;
    int32_t kh, kw, stride, pad;
    int64_t w_stride_m, w_stride_c;
    const float* prepped;          /* include/sae_hip.h (ABI 6): prepared weights -- a GPU launch detail; the oracle convolves */
    int64_t prepped_floats;        /* with the weights themselves and ignores them                                            */
    int64_t prepped_layout;
} sae_conv2d_desc;

static __thread char g_err[256];

int oracle_abi_version(void) { return 12; }
const char* oracle_last_error(void) { return g_err; }

/* include/sae_hip.h: the mode only selects GPU arithmetic; the oracle always accumulates in double */
static int g_conv_math = 0;
int oracle_set_conv_math(int32_t mode) {
    if (mode != 0 && mode != 1) return -1;
    g_conv_math = mode;
    return 0;
}
int oracle_get_conv_math(void) { return g_conv_math; }

static int set_err_early(const char* msg) {
    snprintf(g_err, sizeof g_err, "%s", msg);
    return SAE_EINVAL;
}

/* upfirdn2d_kernel.cu:18-26 */
static inline int64_t floor_div(int64_t a, int64_t b) {
    int64_t c = a / b;
    if (c * b > a) c--;
    return c;
}

/*
 * upfirdn2d_kernel.cu:52-137 (per-output arithmetic :114-129, flipped taps :71-81, zero reads
 * outside the input :98-104) with the output size of :167-168.  The CUDA kernel bounds its tap
 * loops by the template size kernel_h/up and relies on zero-padded taps; here the loop simply
 * stops when the tap index leaves [0,kh) — the same sum.
 */
int oracle_upfirdn2d_f32(const float* x, const float* k, float* y,
                         int64_t major, int64_t in_h, int64_t in_w, int64_t minor,
                         int32_t kh, int32_t kw,
                         int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y,
                         int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1,
                         sae_stream_t stream) {
    (void)stream;
    if (!x || !k || !y || up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1) {
        snprintf(g_err, sizeof g_err, "oracle_upfirdn2d_f32: bad argument");
        return SAE_EINVAL;
    }
    int64_t out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    int64_t out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    if (out_h <= 0 || out_w <= 0) return SAE_OK;
#pragma omp parallel for schedule(static)
    for (int64_t mj = 0; mj < major; ++mj)
        for (int64_t oy = 0; oy < out_h; ++oy)
            for (int64_t ox = 0; ox < out_w; ++ox) {
                int64_t mid_x = ox * down_x + up_x - 1 - pad_x0;
                int64_t mid_y = oy * down_y + up_y - 1 - pad_y0;
                int64_t in_x = floor_div(mid_x, up_x);
                int64_t in_y = floor_div(mid_y, up_y);
                int64_t kx0 = (in_x + 1) * up_x - mid_x - 1;
                int64_t ky0 = (in_y + 1) * up_y - mid_y - 1;
                for (int64_t mn = 0; mn < minor; ++mn) {
                    double v = 0.0;
                    for (int64_t yy = 0; ky0 + yy * up_y < kh; ++yy) {
                        int64_t iy = in_y + yy;
                        if (iy < 0 || iy >= in_h) continue;
                        int64_t ky = ky0 + yy * up_y;
                        for (int64_t xx = 0; kx0 + xx * up_x < kw; ++xx) {
                            int64_t ix = in_x + xx;
                            if (ix < 0 || ix >= in_w) continue;
                            int64_t kx = kx0 + xx * up_x;
                            /* sk[ky][kx] = kernel[kh-1-ky][kw-1-kx]   (:71-81) */
                            double tap = k[(kh - 1 - ky) * kw + (kw - 1 - kx)];
                            v += (double)x[((mj * in_h + iy) * in_w + ix) * minor + mn] * tap;
                        }
                    }
                    y[((mj * out_h + oy) * out_w + ox) * minor + mn] = (float)v;
                }
            }
    return SAE_OK;
}

/* include/sae_hip.h sae_upfirdn2d_epilogue_f32: the K1 call above followed by what the backward pass of a ResBlock
 * (stylegan2_layers.py:672-693) does next with its result -- autograd's sum of the two gradients of the block input
 * (accumulate) and / or FusedLeakyReLUFunctionBackward (fused_act.py:32-41: grad_input = grad_output * (out > 0 ? 1 :
 * slope) * scale in float, as fused_bias_act_kernel.cu:30,47 does; grad_bias = grad_input.sum over all but the channel
 * axis, here in double).  The FIR value is rounded to float first, as the separate kernels would have stored it. */
int64_t oracle_upfirdn2d_epilogue_workspace(int64_t major, int64_t out_h, int64_t out_w, int64_t channels, int32_t up) {
    (void)major; (void)out_h; (void)out_w; (void)channels; (void)up;
    return 0;
}
int oracle_upfirdn2d_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w, int64_t minor,
                         int32_t kh, int32_t kw, int32_t up_x, int32_t up_y, int32_t down_x, int32_t down_y, int32_t pad_x0,
                         int32_t pad_x1, int32_t pad_y0, int32_t pad_y1, sae_stream_t stream);
int oracle_upfirdn2d_epilogue_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                                  int32_t kh, int32_t kw, int32_t up, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0,
                                  int32_t pad_y1, const float* act_ref, float slope, float scale, float* gb,
                                  int64_t channels, int32_t accumulate, float* workspace, int64_t workspace_floats,
                                  sae_stream_t stream) {
    (void)workspace; (void)workspace_floats;
    if (!x || !k || !y || (up != 1 && up != 2) || kh < 1 || kw < 1 || kh > 4 || kw > 4 ||
        (act_ref && (!gb || channels < 1 || major % channels != 0))) {
        snprintf(g_err, sizeof g_err, "oracle_upfirdn2d_epilogue_f32: bad argument");
        return SAE_EINVAL;
    }
    const int64_t out_h = in_h * up + pad_y0 + pad_y1 - kh + 1, out_w = in_w * up + pad_x0 + pad_x1 - kw + 1;
    if (out_h < 1 || out_w < 1) return SAE_EINVAL;
    const int64_t hw = out_h * out_w;
    float* t = (float*)malloc(sizeof(float) * (size_t)(major * hw > 0 ? major * hw : 1));
    if (!t) return SAE_EWORKSPACE;
    int rc = oracle_upfirdn2d_f32(x, k, t, major, in_h, in_w, 1, kh, kw, up, up, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, stream);
    if (rc != SAE_OK) { free(t); return rc; }
    double* acc = act_ref ? (double*)calloc((size_t)channels, sizeof(double)) : NULL;
    for (int64_t p = 0; p < major; ++p)
        for (int64_t i = 0; i < hw; ++i) {
            float v = t[p * hw + i];
            if (accumulate) v += y[p * hw + i];
            if (act_ref) {
                v = (act_ref[p * hw + i] > 0.0f) ? v : v * slope;
                v *= scale;
                acc[p % channels] += (double)v;
            }
            y[p * hw + i] = v;
        }
    if (act_ref) {
        for (int64_t c = 0; c < channels; ++c) gb[c] = (float)acc[c];
        free(acc);
    }
    free(t);
    return SAE_OK;
}

/* the blur, then NoiseInjection + FusedLeakyReLU on its result (include/sae_hip.h) */
int oracle_upfirdn2d_noise_bias_act_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                                        int32_t kh, int32_t kw, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0, int32_t pad_y1,
                                        const float* noise, const float* noise_weight, const float* bias, int64_t channels,
                                        float slope, float scale, sae_stream_t stream) {
    if (!x || !k || !y || kh < 1 || kw < 1 || kh > 4 || kw > 4 || channels < 1 || major % channels != 0 || (noise && !noise_weight)) {
        snprintf(g_err, sizeof g_err, "oracle_upfirdn2d_noise_bias_act_f32: bad argument");
        return SAE_EINVAL;
    }
    const int64_t out_h = in_h + pad_y0 + pad_y1 - kh + 1, out_w = in_w + pad_x0 + pad_x1 - kw + 1;
    if (out_h < 1 || out_w < 1) return SAE_EINVAL;
    int rc = oracle_upfirdn2d_f32(x, k, y, major, in_h, in_w, 1, kh, kw, 1, 1, 1, 1, pad_x0, pad_x1, pad_y0, pad_y1, stream);
    if (rc != SAE_OK) return rc;
    const int64_t hw = out_h * out_w;
    const float wn = noise ? noise_weight[0] : 0.0f;
    for (int64_t p = 0; p < major; ++p)
        for (int64_t i = 0; i < hw; ++i) {
            float t = y[p * hw + i] + (noise ? wn * noise[(p / channels) * hw + i] : 0.0f);
            t += bias ? bias[p % channels] : 0.0f;
            y[p * hw + i] = (t > 0.0f ? t : t * slope) * scale;
        }
    return SAE_OK;
}

/* fused_bias_act_kernel.cu:18-49; the arithmetic is done in float exactly as the CUDA kernel
 * does for scalar_t = float (one add, one select/multiply, one multiply). */
int oracle_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                        int64_t numel, int64_t step_b, int64_t size_b,
                        int32_t act, int32_t grad, float alpha, float scale,
                        sae_stream_t stream) {
    (void)stream;
    if (!x || !y || numel < 0 || (b && (step_b < 1 || size_b < 1)) || (grad == 1 && !ref) ||
        (act != 1 && act != 3) || grad < 0 || grad > 2) {
        snprintf(g_err, sizeof g_err, "oracle_bias_act_f32: bad argument");
        return SAE_EINVAL;
    }
    for (int64_t i = 0; i < numel; ++i) {
        float v = x[i];
        if (b) v += b[(i / step_b) % size_b];
        float r = ref ? ref[i] : 0.0f;
        float o;
        switch (act * 10 + grad) {
            default:
            case 10: o = v; break;
            case 11: o = v; break;
            case 12: o = 0.0f; break;
            case 30: o = (v > 0.0f) ? v : v * alpha; break;
            case 31: o = (r > 0.0f) ? v : v * alpha; break;
            case 32: o = 0.0f; break;
        }
        y[i] = o * scale;
    }
    return SAE_OK;
}

int64_t oracle_bias_act_bwd_workspace(int64_t numel, int64_t step_b, int64_t size_b) {
    (void)numel; (void)step_b; (void)size_b;
    return 0;
}

/* fused_act.py:32-41: grad_input = fused_bias_act(grad_output, empty, out, 3, 1, a, s);
 * grad_bias = grad_input.sum(all dims but 1).  The sum is accumulated in double. */
int oracle_bias_act_bwd_f32(const float* gy, const float* y_ref, float* gx, float* gb,
                            float* workspace, int64_t workspace_floats,
                            int64_t numel, int64_t step_b, int64_t size_b,
                            float alpha, float scale, sae_stream_t stream) {
    (void)stream; (void)workspace; (void)workspace_floats;
    if (!gy || !y_ref || !gx || !gb || step_b < 1 || size_b < 1) {
        snprintf(g_err, sizeof g_err, "oracle_bias_act_bwd_f32: bad argument");
        return SAE_EINVAL;
    }
    double* acc = (double*)calloc((size_t)size_b, sizeof(double));
    if (!acc) return SAE_EWORKSPACE;
    for (int64_t i = 0; i < numel; ++i) {
        float g = gy[i];
        float o = (y_ref[i] > 0.0f) ? g : g * alpha;
        o *= scale;
        gx[i] = o;
        acc[(i / step_b) % size_b] += (double)o;
    }
    for (int64_t c = 0; c < size_b; ++c) gb[c] = (float)acc[c];
    free(acc);
    return SAE_OK;
}

/* ---------------------------------------------------------------------------------------------
 * Dense conv family.  The reference delegates these to PyTorch ATen (un-vendored, unpinned
 * dependency: README.md:29 "PyTorch 1.7.1"); call sites stylegan2_layers.py:136 (EqualConv2d),
 * :175,:182 (EqualLinear on 4-D input), :306 (conv_transpose2d), :315,:321 (ModulatedConv2d).
 * What is restated here is the published definition of torch.nn.functional.conv2d
 * (cross-correlation, zero padding, floor output size) and its two adjoints.
 * ------------------------------------------------------------------------------------------- */
static int conv_desc_ok(const sae_conv2d_desc* d) {
    if (!d || d->n < 0 || d->c < 1 || d->m < 1 || d->h < 1 || d->w < 1 || d->kh < 1 || d->kw < 1 ||
        d->stride < 1 || d->pad < 0)
        return 0;
    if (d->oh != (d->h + 2 * d->pad - d->kh) / d->stride + 1) return 0;
    if (d->ow != (d->w + 2 * d->pad - d->kw) / d->stride + 1) return 0;
    return d->oh >= 1 && d->ow >= 1;
}

int64_t oracle_conv2d_workspace(const sae_conv2d_desc* d, int32_t op) {
    (void)d; (void)op;
    return 0;
}

/* Loop structure of the three conv functions: the parallel axis is (image, channel, block of
 * ORB output rows) so that a one-image / few-channel slice of a BASELINE-size layer (what the
 * full-size GPU parity tests ask for) still spreads over a few hundred host cores; every output
 * element is the double sum over (c, ky, kx) [forward], (m, ky, kx) [dgrad] or (n, oy, ox)
 * [wgrad] of single-rounded-to-double products, accumulated in row buffers so the innermost loop
 * runs along W. */
#define ORB 16

int oracle_conv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d,
                          float alpha, float* workspace, int64_t workspace_floats,
                          sae_stream_t stream) {
    (void)workspace; (void)workspace_floats; (void)stream;
    if (!x || !w || !y || !conv_desc_ok(d)) {
        snprintf(g_err, sizeof g_err, "oracle_conv2d_fwd_f32: bad argument");
        return SAE_EINVAL;
    }
    const int64_t nblk = (d->oh + ORB - 1) / ORB;
    const int64_t S = d->stride, P = d->pad;
#pragma omp parallel for collapse(3) schedule(dynamic, 1)
    for (int64_t n = 0; n < d->n; ++n)
        for (int64_t m = 0; m < d->m; ++m)
            for (int64_t blk = 0; blk < nblk; ++blk) {
                const int64_t oy0 = blk * ORB, oy1 = oy0 + ORB < d->oh ? oy0 + ORB : d->oh;
                double* acc = (double*)calloc((size_t)(ORB * d->ow), sizeof(double));
                for (int64_t c = 0; c < d->c; ++c)
                    for (int ky = 0; ky < d->kh; ++ky)
                        for (int kx = 0; kx < d->kw; ++kx) {
                            const double wv = w[m * d->w_stride_m + c * d->w_stride_c + ky * d->kw + kx];
                            /* ox range with 0 <= ox*S + kx - P < W */
                            int64_t lo = P - kx > 0 ? (P - kx + S - 1) / S : 0;
                            int64_t hi = (d->w - 1 + P - kx) / S + 1;
                            if (d->w - 1 + P - kx < 0) hi = 0;
                            if (hi > d->ow) hi = d->ow;
                            for (int64_t oy = oy0; oy < oy1; ++oy) {
                                const int64_t iy = oy * S + ky - P;
                                if (iy < 0 || iy >= d->h) continue;
                                const float* xr = x + ((n * d->c + c) * d->h + iy) * d->w + (kx - P);
                                double* ar = acc + (oy - oy0) * d->ow;
                                for (int64_t ox = lo; ox < hi; ++ox) ar[ox] += wv * (double)xr[ox * S];
                            }
                        }
                for (int64_t oy = oy0; oy < oy1; ++oy)
                    for (int64_t ox = 0; ox < d->ow; ++ox)
                        y[((n * d->m + m) * d->oh + oy) * d->ow + ox] =
                            (float)(acc[(oy - oy0) * d->ow + ox] * (double)alpha);
                free(acc);
            }
    return SAE_OK;
}

int oracle_conv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d,
                            float alpha, float* workspace, int64_t workspace_floats,
                            sae_stream_t stream) {
    (void)workspace; (void)workspace_floats; (void)stream;
    if (!gy || !w || !gx || !conv_desc_ok(d)) {
        snprintf(g_err, sizeof g_err, "oracle_conv2d_dgrad_f32: bad argument");
        return SAE_EINVAL;
    }
    const int64_t nblk = (d->h + ORB - 1) / ORB;
    const int64_t S = d->stride, P = d->pad;
#pragma omp parallel for collapse(3) schedule(dynamic, 1)
    for (int64_t n = 0; n < d->n; ++n)
        for (int64_t c = 0; c < d->c; ++c)
            for (int64_t blk = 0; blk < nblk; ++blk) {
                const int64_t iy0 = blk * ORB, iy1 = iy0 + ORB < d->h ? iy0 + ORB : d->h;
                double* acc = (double*)calloc((size_t)(ORB * d->w), sizeof(double));
                for (int64_t m = 0; m < d->m; ++m)
                    for (int ky = 0; ky < d->kh; ++ky)
                        for (int kx = 0; kx < d->kw; ++kx) {
                            const double wv = w[m * d->w_stride_m + c * d->w_stride_c + ky * d->kw + kx];
                            int64_t lo = P - kx > 0 ? (P - kx + S - 1) / S : 0;
                            int64_t hi = (d->w - 1 + P - kx) / S + 1;
                            if (d->w - 1 + P - kx < 0) hi = 0;
                            if (hi > d->ow) hi = d->ow;
                            for (int64_t oy = 0; oy < d->oh; ++oy) {
                                const int64_t iy = oy * S + ky - P;
                                if (iy < iy0 || iy >= iy1) continue;
                                const float* gr = gy + ((n * d->m + m) * d->oh + oy) * d->ow;
                                double* ar = acc + (iy - iy0) * d->w + (kx - P);
                                for (int64_t ox = lo; ox < hi; ++ox) ar[ox * S] += wv * (double)gr[ox];
                            }
                        }
                for (int64_t iy = iy0; iy < iy1; ++iy)
                    for (int64_t ix = 0; ix < d->w; ++ix)
                        gx[((n * d->c + c) * d->h + iy) * d->w + ix] =
                            (float)(acc[(iy - iy0) * d->w + ix] * (double)alpha);
                free(acc);
            }
    return SAE_OK;
}

int oracle_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                            float alpha, float* workspace, int64_t workspace_floats,
                            sae_stream_t stream) {
    (void)workspace; (void)workspace_floats; (void)stream;
    if (!x || !gy || !gw || !conv_desc_ok(d)) {
        snprintf(g_err, sizeof g_err, "oracle_conv2d_wgrad_f32: bad argument");
        return SAE_EINVAL;
    }
    const int64_t S = d->stride, P = d->pad;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
    for (int64_t m = 0; m < d->m; ++m)
        for (int64_t c = 0; c < d->c; ++c)
            for (int ky = 0; ky < d->kh; ++ky)
                for (int kx = 0; kx < d->kw; ++kx) {
                    int64_t lo = P - kx > 0 ? (P - kx + S - 1) / S : 0;
                    int64_t hi = (d->w - 1 + P - kx) / S + 1;
                    if (d->w - 1 + P - kx < 0) hi = 0;
                    if (hi > d->ow) hi = d->ow;
                    double acc = 0.0;
                    for (int64_t n = 0; n < d->n; ++n)
                        for (int64_t oy = 0; oy < d->oh; ++oy) {
                            const int64_t iy = oy * S + ky - P;
                            if (iy < 0 || iy >= d->h) continue;
                            const float* gr = gy + ((n * d->m + m) * d->oh + oy) * d->ow;
                            const float* xr = x + ((n * d->c + c) * d->h + iy) * d->w + (kx - P);
                            double row = 0.0;
#pragma omp simd reduction(+ : row)
                            for (int64_t ox = lo; ox < hi; ++ox) row += (double)gr[ox] * (double)xr[ox * S];
                            acc += row;
                        }
                    gw[m * d->w_stride_m + c * d->w_stride_c + ky * d->kw + kx] =
                        (float)(acc * (double)alpha);
                }
    return SAE_OK;
}

/* F.linear (stylegan2_layers.py:177,186) and its adjoints as one strided product. */
int oracle_gemm_f32(const float* a, const float* b, const float* bias, float* c,
                    int64_t m, int64_t n, int64_t k,
                    int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc,
                    float alpha, sae_stream_t stream) {
    (void)stream;
    if (!a || !b || !c || m < 0 || n < 0 || k < 0) {
        snprintf(g_err, sizeof g_err, "oracle_gemm_f32: bad argument");
        return SAE_EINVAL;
    }
    for (int64_t i = 0; i < m; ++i)
        for (int64_t j = 0; j < n; ++j) {
            double acc = 0.0;
            for (int64_t kk = 0; kk < k; ++kk)
                acc += (double)a[i * a_si + kk * a_sk] * (double)b[kk * b_sk + j * b_sj];
            acc *= (double)alpha;
            if (bias) acc += (double)bias[j];
            c[i * ldc + j] = (float)acc;
        }
    return SAE_OK;
}

/* sae_gemm_ws_f32 / sae_gemm_workspace: the K split is a launch detail of the GPU; the oracle needs no workspace. */
int64_t oracle_gemm_workspace(int64_t m, int64_t n, int64_t k) { (void)m; (void)n; (void)k; return 0; }
int oracle_gemm_ws_f32(const float* a, const float* b, const float* bias, float* c, int64_t m, int64_t n, int64_t k,
                       int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc, float alpha, float* workspace,
                       int64_t workspace_floats, sae_stream_t stream) {
    (void)workspace; (void)workspace_floats;
    return oracle_gemm_f32(a, b, bias, c, m, n, k, a_si, a_sk, b_sk, b_sj, ldc, alpha, stream);
}

/* ---------------------------------------------------------------------------------------------
 * Bilinear x2 upsampling, align_corners = false (ATen UpSampleBilinear2d, un-vendored dependency;
 * call site generator.py:51) + the residual add / scale of generator.py:53, and the adjoint.
 * Published formula: src = max((d + 0.5)/scale - 0.5, 0), i0 = floor(src), l1 = src - i0,
 * i1 = i0 + (i0 < n-1).
 * ------------------------------------------------------------------------------------------- */
static void up2_src(int64_t d, int64_t n, int64_t* i0, int64_t* i1, double* l0, double* l1) {
    double src = ((double)d + 0.5) * 0.5 - 0.5;
    if (src < 0.0) src = 0.0;
    *i0 = (int64_t)src;
    *l1 = src - (double)*i0;
    *l0 = 1.0 - *l1;
    *i1 = *i0 + ((*i0 < n - 1) ? 1 : 0);
}

int oracle_upsample2x_bilinear_add_f32(const float* x, const float* res, float* y, int64_t planes,
                                       int64_t h, int64_t w, float alpha, sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || h < 1 || w < 1 || (planes > 0 && (!x || !y))) {
        snprintf(g_err, sizeof g_err, "oracle_upsample2x_bilinear_add_f32: bad argument");
        return SAE_EINVAL;
    }
#pragma omp parallel for schedule(static)
    for (int64_t pl = 0; pl < planes; ++pl)
        for (int64_t oy = 0; oy < 2 * h; ++oy) {
            int64_t y0, y1; double hl0, hl1;
            up2_src(oy, h, &y0, &y1, &hl0, &hl1);
            for (int64_t ox = 0; ox < 2 * w; ++ox) {
                int64_t x0, x1; double wl0, wl1;
                up2_src(ox, w, &x0, &x1, &wl0, &wl1);
                const float* xp = x + pl * h * w;
                double v = hl0 * (wl0 * xp[y0 * w + x0] + wl1 * xp[y0 * w + x1]) +
                           hl1 * (wl0 * xp[y1 * w + x0] + wl1 * xp[y1 * w + x1]);
                int64_t o = (pl * 2 * h + oy) * 2 * w + ox;
                if (res) v += (double)res[o];
                y[o] = (float)(v * (double)alpha);
            }
        }
    return SAE_OK;
}

int oracle_upsample2x_bilinear_bwd_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w,
                                       float alpha, sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || h < 1 || w < 1 || (planes > 0 && (!gy || !gx))) {
        snprintf(g_err, sizeof g_err, "oracle_upsample2x_bilinear_bwd_f32: bad argument");
        return SAE_EINVAL;
    }
    /* scatter form of the adjoint, straight from the forward definition */
#pragma omp parallel for schedule(static)
    for (int64_t pl = 0; pl < planes; ++pl) {
        double* acc = (double*)calloc((size_t)(h * w), sizeof(double));
        for (int64_t oy = 0; oy < 2 * h; ++oy) {
            int64_t y0, y1; double hl0, hl1;
            up2_src(oy, h, &y0, &y1, &hl0, &hl1);
            for (int64_t ox = 0; ox < 2 * w; ++ox) {
                int64_t x0, x1; double wl0, wl1;
                up2_src(ox, w, &x0, &x1, &wl0, &wl1);
                double g = gy[(pl * 2 * h + oy) * 2 * w + ox];
                acc[y0 * w + x0] += hl0 * wl0 * g;
                acc[y0 * w + x1] += hl0 * wl1 * g;
                acc[y1 * w + x0] += hl1 * wl0 * g;
                acc[y1 * w + x1] += hl1 * wl1 * g;
            }
        }
        for (int64_t i = 0; i < h * w; ++i) gx[pl * h * w + i] = (float)(acc[i] * (double)alpha);
        free(acc);
    }
    return SAE_OK;
}

/* ConvLayer's Conv -> Act pair (stylegan2_layers.py:642-659) in one call: the conv of
 * oracle_conv2d_fwd_f32 followed by fused_bias_act_kernel.cu:30,36-47 (act = 3, grad = 0). */
int oracle_conv2d_fwd_bias_act_f32(const float* x, const float* w, const float* bias, float* y,
                                   const sae_conv2d_desc* d, float alpha, float act_slope, float act_scale,
                                   float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    int rc = oracle_conv2d_fwd_f32(x, w, y, d, alpha, workspace, workspace_floats, stream);
    if (rc != SAE_OK) return rc;
    int64_t hw = d->oh * d->ow;
    return oracle_bias_act_f32(y, bias, NULL, y, d->n * d->m * hw, hw, d->m, 3, 0, act_slope, act_scale, stream);
}

/* include/sae_hip.h sae_conv2d_fwd_residual_f32: ResBlock's skip conv followed by the merge `(out + skip) / math.sqrt(2)`
 * (stylegan2_layers.py:683-689): the conv as oracle_conv2d_fwd_f32 (double accumulation, rounded to float as the separate
 * call would store it), then (conv + residual) * res_scale in float. */
int oracle_conv2d_fwd_residual_f32(const float* x, const float* w, const float* residual, float* y, const sae_conv2d_desc* d,
                                   float alpha, float res_scale, float* workspace, int64_t workspace_floats,
                                   sae_stream_t stream) {
    if (!residual) return set_err_early("oracle_conv2d_fwd_residual_f32: null residual");
    int rc = oracle_conv2d_fwd_f32(x, w, y, d, alpha, workspace, workspace_floats, stream);
    if (rc != SAE_OK) return rc;
    const int64_t numel = d->n * d->m * d->oh * d->ow;
    for (int64_t i = 0; i < numel; ++i) y[i] = (y[i] + residual[i]) * res_scale;
    return SAE_OK;
}

/* (out + skip) / sqrt(2), stylegan2_layers.py:689 / generator.py:36, as alpha * (a + b) in float. */
int oracle_add_scale_f32(const float* a, const float* b, float* y, int64_t numel, float alpha, sae_stream_t stream) {
    (void)stream;
    if (numel < 0 || (numel > 0 && (!a || !b || !y))) return SAE_EINVAL;
    for (int64_t i = 0; i < numel; ++i) y[i] = (a[i] + b[i]) * alpha;
    return SAE_OK;
}


static int set_err(const char* msg) {
    snprintf(g_err, sizeof g_err, "oracle: %s", msg);
    return -1;
}

/* ---- StyledConv glue (include/sae_hip.h): stylegan2_layers.py:340-351 (noise), :54-65 (bias + leaky-ReLU),
 * :280-286 (style modulation) and their autograd ---------------------------------------------------------- */
int oracle_noise_bias_act_f32(const float* x, const float* noise, const float* noise_weight, const float* bias,
                              float* y, int64_t outer, int64_t channels, int64_t hw, float alpha, float scale,
                              void* stream) {
    (void)stream;
    if (outer < 0 || channels < 1 || hw < 4 || hw % 4) return set_err("noise_bias_act: bad shape");
    const float wn = noise ? noise_weight[0] : 0.0f;
#pragma omp parallel for collapse(2)
    for (int64_t n = 0; n < outer; ++n)
        for (int64_t c = 0; c < channels; ++c) {
            const float* xp = x + (n * channels + c) * hw;
            float* yp = y + (n * channels + c) * hw;
            for (int64_t p = 0; p < hw; ++p) {
                float t = xp[p] + (noise ? wn * noise[n * hw + p] : 0.0f);
                t += bias ? bias[c] : 0.0f;
                yp[p] = (t > 0.0f ? t : t * alpha) * scale;
            }
        }
    return 0;
}

int64_t oracle_noise_bias_act_bwd_workspace(int64_t outer, int64_t channels, int64_t hw) {
    (void)outer; (void)channels; (void)hw;
    return 1;
}

int oracle_noise_bias_act_bwd_f32(const float* gy, const float* y_ref, const float* noise, float* gx, float* gbias,
                                  float* gnoise_weight, float* workspace, int64_t workspace_floats, int64_t outer,
                                  int64_t channels, int64_t hw, float alpha, float scale, void* stream) {
    (void)stream; (void)workspace; (void)workspace_floats;
    if (outer < 0 || channels < 1 || hw < 4 || hw % 4) return set_err("noise_bias_act_bwd: bad shape");
    double gw = 0.0;
    for (int64_t c = 0; c < channels; ++c) {
        double gb = 0.0;
        for (int64_t n = 0; n < outer; ++n) {
            const int64_t base = (n * channels + c) * hw;
            for (int64_t p = 0; p < hw; ++p) {
                const float o = (y_ref[base + p] > 0.0f ? gy[base + p] : gy[base + p] * alpha) * scale;
                gx[base + p] = o;
                gb += o;
                if (noise) gw += (double)o * noise[n * hw + p];
            }
        }
        if (gbias) gbias[c] = (float)gb;
    }
    if (gnoise_weight && noise) gnoise_weight[0] = (float)gw;
    return 0;
}

int oracle_plane_scale_dot_f32(const float* g, const float* x, const float* s, float* gx, float* gs, int64_t planes,
                               int64_t hw, void* stream) {
    (void)stream;
    if (planes < 0 || hw < 4 || hw % 4) return set_err("plane_scale_dot: bad shape");
#pragma omp parallel for
    for (int64_t q = 0; q < planes; ++q) {
        double acc = 0.0;
        for (int64_t p = 0; p < hw; ++p) {
            gx[q * hw + p] = g[q * hw + p] * s[q];
            acc += (double)g[q * hw + p] * x[q * hw + p];
        }
        gs[q] = (float)acc;
    }
    return 0;
}

int64_t oracle_plane_scale_dot_act_workspace(int64_t outer, int64_t channels) { (void)outer; (void)channels; return 1; }

/* the two passes one after the other, element by element (include/sae_hip.h) */
int oracle_plane_scale_dot_act_f32(const float* g, const float* x, const float* s, const float* noise, float* gx, float* gs,
                                   float* gbias, float* gnoise_weight, float* workspace, int64_t workspace_floats, int64_t outer,
                                   int64_t channels, int64_t hw, float alpha, float scale, void* stream) {
    (void)stream; (void)workspace; (void)workspace_floats;
    if (outer < 0 || channels < 1 || hw < 4 || hw % 4) return set_err("plane_scale_dot_act: bad shape");
    double gw = 0.0;
    for (int64_t c = 0; c < channels; ++c) {
        double gb = 0.0;
        for (int64_t n = 0; n < outer; ++n) {
            const int64_t q = n * channels + c;
            double acc = 0.0;
            for (int64_t p = 0; p < hw; ++p) {
                const float t = g[q * hw + p] * s[q];
                const float o = (x[q * hw + p] > 0.0f ? t : t * alpha) * scale;
                gx[q * hw + p] = o;
                acc += (double)g[q * hw + p] * x[q * hw + p];
                gb += o;
                if (noise) gw += (double)o * noise[n * hw + p];
            }
            gs[q] = (float)acc;
        }
        if (gbias) gbias[c] = (float)gb;
    }
    if (gnoise_weight && noise) gnoise_weight[0] = (float)gw;
    return 0;
}

/* ---- demodulation factor and its backward (include/sae_hip.h; stylegan2_layers.py:290-292): the reference's own sequence --
 * scale, square, sum, + eps, rsqrt -- with the sum in double; the backward is the closed form of autograd's chain. ---- */
int oracle_weight_demod_f32(const float* w, float* d, int64_t rows, int64_t cols, float alpha, float eps, void* stream) {
    (void)stream;
    if (rows < 0 || cols < 1) return set_err("weight_demod: bad shape");
#pragma omp parallel for
    for (int64_t o = 0; o < rows; ++o) {
        double acc = 0.0;
        for (int64_t j = 0; j < cols; ++j) {
            const float u = w[o * cols + j] * alpha;
            acc += (double)(u * u);
        }
        d[o] = 1.0f / sqrtf((float)acc + eps);
    }
    return 0;
}

int oracle_weight_demod_bwd_f32(const float* geff, const float* w, const float* d, float* gw, int64_t rows, int64_t cols,
                                float alpha, void* stream) {
    (void)stream;
    if (rows < 0 || cols < 1) return set_err("weight_demod_bwd: bad shape");
#pragma omp parallel for
    for (int64_t o = 0; o < rows; ++o) {
        double gd = 0.0;
        for (int64_t j = 0; j < cols; ++j) gd += (double)geff[o * cols + j] * w[o * cols + j];
        const double dv = d[o], cf = gd * dv * dv * dv * (double)alpha * alpha;
        for (int64_t j = 0; j < cols; ++j) gw[o * cols + j] = (float)(dv * geff[o * cols + j] - cf * w[o * cols + j]);
    }
    return 0;
}


/* ---- random-crop sampler (include/sae_hip.h): util/util.py:323-343, i.e. F.grid_sample(bilinear, zeros,
 * align_corners=False) on the grid lin * flip * scale + offset.  Coordinates in float exactly as ATen's
 * grid_sampler computes them; interpolation accumulated in double; the backward is the plain scatter. ------ */
static float crop_coord_o(float lin, float mul, float off, int extent) {
    volatile float g = lin * mul;
    g = g + off;
    volatile float t = g + 1.0f;
    t = t * (float)extent;
    t = t - 1.0f;
    return t * 0.5f;
}

int oracle_random_crop_f32(const float* x, const float* params, const float* lin, float* y, int64_t images,
                           int64_t channels, int64_t h, int64_t w, int64_t crops, int64_t size, void* stream) {
    (void)stream;
    if (images < 0 || channels < 1 || h < 1 || w < 1 || crops < 1 || size < 2) return set_err("random_crop: bad geometry");
#pragma omp parallel for
    for (int64_t k = 0; k < images * crops; ++k) {
        const int64_t b = k / crops;
        const float* pr = params + 5 * k;
        for (int64_t r = 0; r < size; ++r)
            for (int64_t c = 0; c < size; ++c) {
                volatile float lf = lin[c] * pr[0];
                const float ix = crop_coord_o(lf, pr[1], pr[3], (int)w), iy = crop_coord_o(lin[r], pr[2], pr[4], (int)h);
                const float fx = floorf(ix), fy = floorf(iy);
                const int64_t x0 = (int64_t)fx, y0 = (int64_t)fy;
                const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
                for (int64_t ch = 0; ch < channels; ++ch) {
                    const float* xp = x + (b * channels + ch) * h * w;
                    double acc = 0.0;
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int64_t yy = y0 + dy, xx = x0 + dx;
                            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                            acc += (double)xp[yy * w + xx] * (double)((dx ? wx1 : wx0) * (dy ? wy1 : wy0));
                        }
                    y[((k * channels + ch) * size + r) * size + c] = (float)acc;
                }
            }
    }
    return 0;
}

int oracle_random_crop_bwd_f32(const float* gy, const float* params, const float* lin, float* gx, int64_t images,
                               int64_t channels, int64_t h, int64_t w, int64_t crops, int64_t size, void* stream) {
    (void)stream;
    if (images < 0 || channels < 1 || h < 1 || w < 1 || crops < 1 || size < 2) return set_err("random_crop_bwd: bad geometry");
    const int64_t plane = h * w;
    double* acc = (double*)calloc((size_t)(images * channels * plane), sizeof(double));
    if (!acc) return set_err("random_crop_bwd: out of memory");
    for (int64_t k = 0; k < images * crops; ++k) {
        const int64_t b = k / crops;
        const float* pr = params + 5 * k;
        for (int64_t r = 0; r < size; ++r)
            for (int64_t c = 0; c < size; ++c) {
                volatile float lf = lin[c] * pr[0];
                const float ix = crop_coord_o(lf, pr[1], pr[3], (int)w), iy = crop_coord_o(lin[r], pr[2], pr[4], (int)h);
                const float fx = floorf(ix), fy = floorf(iy);
                const int64_t x0 = (int64_t)fx, y0 = (int64_t)fy;
                const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix, wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
                for (int64_t ch = 0; ch < channels; ++ch) {
                    const double g = gy[((k * channels + ch) * size + r) * size + c];
                    for (int dy = 0; dy < 2; ++dy)
                        for (int dx = 0; dx < 2; ++dx) {
                            const int64_t yy = y0 + dy, xx = x0 + dx;
                            if (yy < 0 || yy >= h || xx < 0 || xx >= w) continue;
                            acc[(b * channels + ch) * plane + yy * w + xx] += g * (double)((dx ? wx1 : wx0) * (dy ? wy1 : wy0));
                        }
                }
            }
    }
    for (int64_t i = 0; i < images * channels * plane; ++i) gx[i] = (float)acc[i];
    free(acc);
    return 0;
}


/* ---- reflection padding and its adjoint (include/sae_hip.h): nn.ReflectionPad2d, stylegan2_layers.py:57-63,100-105,643 */
static int64_t refl_o(int64_t i, int64_t n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

int oracle_reflect_pad_f32(const float* x, float* y, int64_t planes, int64_t h, int64_t w, int32_t left, int32_t right,
                           int32_t top, int32_t bottom, void* stream) {
    (void)stream;
    if (planes < 0 || h < 1 || w < 1 || left < 0 || right < 0 || top < 0 || bottom < 0 || left >= w || right >= w ||
        top >= h || bottom >= h) return set_err("reflect_pad: bad geometry");
    const int64_t oh = h + top + bottom, ow = w + left + right;
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t oy = 0; oy < oh; ++oy)
            for (int64_t ox = 0; ox < ow; ++ox)
                y[(p * oh + oy) * ow + ox] = x[(p * h + refl_o(oy - top, h)) * w + refl_o(ox - left, w)];
    return 0;
}

int oracle_reflect_pad_adj_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w, int32_t left,
                               int32_t right, int32_t top, int32_t bottom, void* stream) {
    (void)stream;
    if (planes < 0 || h < 1 || w < 1 || left < 0 || right < 0 || top < 0 || bottom < 0 || left >= w || right >= w ||
        top >= h || bottom >= h) return set_err("reflect_pad_adj: bad geometry");
    const int64_t oh = h + top + bottom, ow = w + left + right;
    double* acc = (double*)calloc((size_t)(planes * h * w), sizeof(double));
    if (!acc) return set_err("reflect_pad_adj: out of memory");
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t oy = 0; oy < oh; ++oy)
            for (int64_t ox = 0; ox < ow; ++ox)
                acc[(p * h + refl_o(oy - top, h)) * w + refl_o(ox - left, w)] += gy[(p * oh + oy) * ow + ox];
    for (int64_t i = 0; i < planes * h * w; ++i) gx[i] = (float)acc[i];
    free(acc);
    return 0;
}

/* ---- Adam (include/sae_hip.h: sae_adam_multi_f32).  The reference steps two torch.optim.Adam instances
 * (optimizers/swapping_autoencoder_optimizer.py:34-42,77,95,107); PyTorch is an un-vendored dependency, so what is
 * restated is the published update of torch.optim.Adam (amsgrad = False, weight_decay = 0), evaluated in double;
 * tests/test_adam.py pins this function to torch.optim.Adam itself. */
int oracle_adam_multi_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                          const int64_t* numel, const int64_t* step, int64_t count, double lr, double beta1, double beta2,
                          double eps, double grad_scale, sae_stream_t stream) {
    (void)stream;
    if (count < 0 || (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !step)))
        return set_err("oracle_adam_multi_f32: null table");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
        return set_err("oracle_adam_multi_f32: betas must be in [0, 1), eps >= 0");
    for (int64_t t = 0; t < count; ++t) {
        if (numel[t] < 0 || step[t] < 1) return set_err("oracle_adam_multi_f32: numel < 0 or step < 1");
        if (numel[t] > 0 && (!params[t] || !grads[t] || !exp_avg[t] || !exp_avg_sq[t]))
            return set_err("oracle_adam_multi_f32: null pointer");
        const double bc1 = 1.0 - pow((double)beta1, (double)step[t]);
        const double bc2 = 1.0 - pow((double)beta2, (double)step[t]);
        const double step_size = (double)lr / bc1, bc2_sqrt = sqrt(bc2);
        for (int64_t i = 0; i < numel[t]; ++i) {
            const double g = (double)grads[t][i] * (double)grad_scale;
            const double m = (double)exp_avg[t][i] + (g - (double)exp_avg[t][i]) * (1.0 - (double)beta1);
            const double v = (double)exp_avg_sq[t][i] * (double)beta2 + (1.0 - (double)beta2) * g * g;
            const double denom = sqrt(v) / bc2_sqrt + (double)eps;
            params[t][i] = (float)((double)params[t][i] - step_size * (m / denom));
            exp_avg[t][i] = (float)m;
            exp_avg_sq[t][i] = (float)v;
        }
    }
    return SAE_OK;
}

/* sae_adam_multi_dev_f32: the same update with the counts read from (here: host) memory and advanced afterwards. */
int oracle_adam_multi_dev_f32(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
                              const int64_t* numel, int64_t* const* step_dev, int64_t count, double lr, double beta1,
                              double beta2, double eps, double grad_scale, sae_stream_t stream) {
    if (count < 0 || (count > 0 && !step_dev)) return set_err("oracle_adam_multi_dev_f32: null step table");
    int64_t* step = (int64_t*)malloc(sizeof(int64_t) * (size_t)(count > 0 ? count : 1));
    if (!step) return set_err("oracle_adam_multi_dev_f32: out of memory");
    for (int64_t t = 0; t < count; ++t) {
        if (!step_dev[t]) { free(step); return set_err("oracle_adam_multi_dev_f32: null step counter"); }
        step[t] = *step_dev[t] + 1;
    }
    const int rc = oracle_adam_multi_f32(params, grads, exp_avg, exp_avg_sq, numel, step, count, lr, beta1, beta2, eps, grad_scale,
                                         stream);
    free(step);
    if (rc == SAE_OK)
        for (int64_t t = 0; t < count; ++t) *step_dev[t] += 1;
    return rc;
}

/* ---- Style-modulated convolution (include/sae_hip.h: sae_modconv2d_*).  ModulatedConv2d.forward,
 * stylegan2_layers.py:266-325: the style scales the INPUT per sample (:280-286: input * style), the demodulation
 * factor scales the weight per output channel (:290-292), then the plain conv / conv_transpose runs (:306,:315,:321).
 * Restated literally: the factors are applied to copies of the operands, then the plain oracle operation runs. */
typedef struct sae_conv2d_mod {
    const float* x_scale;
    const float* y_scale;
    const float* wm_scale;
    const float* wc_scale;
} sae_conv2d_mod;

static float* scaled_activation(const float* t, const float* scale, int64_t n, int64_t c, int64_t hw) {
    float* out = (float*)malloc(sizeof(float) * (size_t)(n * c * hw > 0 ? n * c * hw : 1));
    if (!out) return NULL;
    for (int64_t i = 0; i < n * c; ++i)
        for (int64_t j = 0; j < hw; ++j) out[i * hw + j] = scale ? t[i * hw + j] * scale[i] : t[i * hw + j];
    return out;
}

/* dense [m][c][k][k] copy of the weight with the factors applied; *dd describes it */
static float* scaled_weight(const float* w, const sae_conv2d_desc* d, const sae_conv2d_mod* mod, sae_conv2d_desc* dd) {
    const int64_t kk = (int64_t)d->kh * d->kw;
    float* out = (float*)malloc(sizeof(float) * (size_t)(d->m * d->c * kk));
    if (!out) return NULL;
    for (int64_t m = 0; m < d->m; ++m)
        for (int64_t c = 0; c < d->c; ++c)
            for (int64_t t = 0; t < kk; ++t) {
                float v = w[m * d->w_stride_m + c * d->w_stride_c + t];
                if (mod && mod->wm_scale) v *= mod->wm_scale[m];
                if (mod && mod->wc_scale) v *= mod->wc_scale[c];
                out[(m * d->c + c) * kk + t] = v;
            }
    *dd = *d;
    dd->w_stride_m = d->c * kk;
    dd->w_stride_c = kk;
    return out;
}

int oracle_modconv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                             float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    if (!x || !w || !y || !conv_desc_ok(d)) return set_err("oracle_modconv2d_fwd_f32: bad argument");
    if (mod && mod->y_scale) return set_err("oracle_modconv2d_fwd_f32: y_scale has no meaning for the forward operation");
    sae_conv2d_desc dd;
    float* xs = scaled_activation(x, mod ? mod->x_scale : NULL, d->n, d->c, d->h * d->w);
    float* ws = scaled_weight(w, d, mod, &dd);
    int rc = (xs && ws) ? oracle_conv2d_fwd_f32(xs, ws, y, &dd, alpha, workspace, workspace_floats, stream) : SAE_EWORKSPACE;
    free(xs); free(ws);
    return rc;
}

int oracle_modconv2d_fwd_noise_bias_act_f32(const float* x, const float* w, const float* noise, const float* noise_weight,
                                            const float* bias, float* y, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                                            float alpha, float act_slope, float act_scale, float* workspace,
                                            int64_t workspace_floats, sae_stream_t stream) {
    /* the three modules one after the other: the modulated conv, then NoiseInjection + FusedLeakyReLU */
    if (!mod || !mod->x_scale) return set_err("oracle_modconv2d_fwd_noise_bias_act_f32: x_scale is required");
    if (!conv_desc_ok(d) || d->stride != 1) return set_err("oracle_modconv2d_fwd_noise_bias_act_f32: stride 1 only");
    int rc = oracle_modconv2d_fwd_f32(x, w, y, d, mod, alpha, workspace, workspace_floats, stream);
    if (rc != 0) return rc;
    const int64_t hw = d->oh * d->ow;
    const float wn = noise ? noise_weight[0] : 0.0f;
#pragma omp parallel for collapse(2)
    for (int64_t n = 0; n < d->n; ++n)
        for (int64_t c = 0; c < d->m; ++c) {
            float* yp = y + (n * d->m + c) * hw;
            for (int64_t p = 0; p < hw; ++p) {
                float t = yp[p] + (noise ? wn * noise[n * hw + p] : 0.0f);      /* (image + weight * noise) + bias */
                t += bias ? bias[c] : 0.0f;
                yp[p] = (t > 0.0f ? t : t * act_slope) * act_scale;
            }
        }
    return 0;
}

int oracle_modconv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                               float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    if (!gy || !w || !gx || !conv_desc_ok(d)) return set_err("oracle_modconv2d_dgrad_f32: bad argument");
    if (mod && mod->x_scale) return set_err("oracle_modconv2d_dgrad_f32: x_scale has no meaning for the data gradient");
    sae_conv2d_desc dd;
    float* gs = scaled_activation(gy, mod ? mod->y_scale : NULL, d->n, d->m, d->oh * d->ow);
    float* ws = scaled_weight(w, d, mod, &dd);
    int rc = (gs && ws) ? oracle_conv2d_dgrad_f32(gs, ws, gx, &dd, alpha, workspace, workspace_floats, stream) : SAE_EWORKSPACE;
    free(gs); free(ws);
    return rc;
}

int oracle_modconv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                               float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    if (!x || !gy || !gw || !conv_desc_ok(d)) return set_err("oracle_modconv2d_wgrad_f32: bad argument");
    if (mod && (mod->wm_scale || mod->wc_scale))
        return set_err("oracle_modconv2d_wgrad_f32: weight factors have no meaning for the weight gradient");
    float* xs = scaled_activation(x, mod ? mod->x_scale : NULL, d->n, d->c, d->h * d->w);
    float* gs = scaled_activation(gy, mod ? mod->y_scale : NULL, d->n, d->m, d->oh * d->ow);
    int rc = (xs && gs) ? oracle_conv2d_wgrad_f32(xs, gs, gw, d, alpha, workspace, workspace_floats, stream) : SAE_EWORKSPACE;
    free(xs); free(gs);
    return rc;
}

/* include/sae_hip.h "Prepared weights": the oracle has no weight layout (floats = 0 tells the caller not to prepare) */
int oracle_conv2d_wprep_query(const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, int64_t* floats, int64_t* layout) {
    (void)mod; (void)op;
    if (!floats || !layout || !conv_desc_ok(d)) return set_err("oracle_conv2d_wprep_query: bad argument");
    *floats = 0; *layout = 0;
    return SAE_OK;
}
int oracle_conv2d_wprep_f32(const float* w, const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, float alpha,
                            float* out, int64_t out_floats, sae_stream_t stream) {
    (void)w; (void)d; (void)mod; (void)op; (void)alpha; (void)out; (void)out_floats; (void)stream;
    return set_err("oracle_conv2d_wprep_f32: the oracle has no weight layout (oracle_conv2d_wprep_query reports 0 floats)");
}

/* ---- small glue (include/sae_hip.h, csrc/glue.hip).  util.normalize (util/util.py:18-22): v * rsqrt(sum(v^2, dim 1) + 1e-8);
 * GeneratorModulation (generator.py:62-67): x * (1 * scale) + bias; gan_loss (loss.py:10-16): softplus(+-x).view(B,-1).mean(1).
 * Evaluated in double; tests/test_glue.py pins each to the ATen expression the reference writes. */
int oracle_l2_normalize_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner, float eps,
                            sae_stream_t stream) {
    (void)stream;
    if (outer < 0 || channels < 1 || inner < 1 || (outer > 0 && (!x || !y))) return set_err("oracle_l2_normalize_f32: bad argument");
    for (int64_t n = 0; n < outer; ++n)
        for (int64_t j = 0; j < inner; ++j) {
            double s = 0.0;
            for (int64_t c = 0; c < channels; ++c) { double v = x[(n * channels + c) * inner + j]; s += v * v; }
            const double r = 1.0 / sqrt(s + (double)eps);
            for (int64_t c = 0; c < channels; ++c) y[(n * channels + c) * inner + j] = (float)(x[(n * channels + c) * inner + j] * r);
        }
    return SAE_OK;
}

int oracle_l2_normalize_bwd_f32(const float* gy, const float* x, float* gx, int64_t outer, int64_t channels, int64_t inner,
                                float eps, sae_stream_t stream) {
    (void)stream;
    if (outer < 0 || channels < 1 || inner < 1 || (outer > 0 && (!gy || !x || !gx)))
        return set_err("oracle_l2_normalize_bwd_f32: bad argument");
    for (int64_t n = 0; n < outer; ++n)
        for (int64_t j = 0; j < inner; ++j) {
            double s = 0.0, d = 0.0;
            for (int64_t c = 0; c < channels; ++c) {
                const double v = x[(n * channels + c) * inner + j];
                s += v * v;
                d += (double)gy[(n * channels + c) * inner + j] * v;
            }
            const double r = 1.0 / sqrt(s + (double)eps);
            for (int64_t c = 0; c < channels; ++c) {
                const int64_t i = (n * channels + c) * inner + j;
                gx[i] = (float)(r * gy[i] - r * r * r * d * x[i]);
            }
        }
    return SAE_OK;
}

int oracle_plane_affine_f32(const float* x, const float* a, const float* b, float* y, int64_t planes, int64_t hw,
                            sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || hw < 1 || (planes > 0 && (!x || !a || !b || !y))) return set_err("oracle_plane_affine_f32: bad argument");
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t i = 0; i < hw; ++i) y[p * hw + i] = (float)((double)x[p * hw + i] * (double)a[p] + (double)b[p]);
    return SAE_OK;
}

int oracle_plane_affine_bwd_f32(const float* g, const float* x, const float* a, float* gx, float* ga, float* gb,
                                int64_t planes, int64_t hw, sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || hw < 1 || (planes > 0 && (!g || !x || !a || !gx || !ga || !gb)))
        return set_err("oracle_plane_affine_bwd_f32: bad argument");
    for (int64_t p = 0; p < planes; ++p) {
        double sa = 0.0, sb = 0.0;
        for (int64_t i = 0; i < hw; ++i) {
            gx[p * hw + i] = (float)((double)g[p * hw + i] * (double)a[p]);
            sa += (double)g[p * hw + i] * (double)x[p * hw + i];
            sb += (double)g[p * hw + i];
        }
        ga[p] = (float)sa;
        gb[p] = (float)sb;
    }
    return SAE_OK;
}

static double softplus_d(double v) { return v > 20.0 ? v : log1p(exp(v)); }

int oracle_softplus_mean_f32(const float* x, float* y, int64_t batch, int64_t inner, float sign, sae_stream_t stream) {
    (void)stream;
    if (batch < 0 || inner < 1 || (batch > 0 && (!x || !y))) return set_err("oracle_softplus_mean_f32: bad argument");
    for (int64_t b = 0; b < batch; ++b) {
        double s = 0.0;
        for (int64_t i = 0; i < inner; ++i) s += softplus_d((double)sign * (double)x[b * inner + i]);
        y[b] = (float)(s / (double)inner);
    }
    return SAE_OK;
}

int oracle_softplus_mean_bwd_f32(const float* gy, const float* x, float* gx, int64_t batch, int64_t inner, float sign,
                                 sae_stream_t stream) {
    (void)stream;
    if (batch < 0 || inner < 1 || (batch > 0 && (!gy || !x || !gx))) return set_err("oracle_softplus_mean_bwd_f32: bad argument");
    for (int64_t b = 0; b < batch; ++b)
        for (int64_t i = 0; i < inner; ++i) {
            const double z = (double)sign * (double)x[b * inner + i];
            const double dz = z > 20.0 ? 1.0 : 1.0 / (1.0 + exp(-z));
            gx[b * inner + i] = (float)((double)gy[b] * (double)sign * dz / (double)inner);
        }
    return SAE_OK;
}

/* ------------------------------------------------------------------------------------------
 * Winograd F(2x2, 3x3) transforms (include/sae_hip.h: sae_wino_*; csrc/winograd.hip).  Restated from the definition -- the
 * three matrix products written out as loops over the published matrices (Lavin & Gray 2016, the ones cuDNN's and MIOpen's
 * F(2x2,3x3) kernels use) in double, rounded once -- not from the kernels' factored sums.  The reference itself reaches
 * these through F.conv2d (models/networks/stylegan2_layers.py:136,315).
 * ------------------------------------------------------------------------------------------ */
static const double WINO_BT[4][4] = {{1, 0, -1, 0}, {0, 1, 1, 0}, {0, -1, 1, 0}, {0, 1, 0, -1}};
static const double WINO_G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
static const double WINO_AT[2][4] = {{1, 1, 1, 0}, {0, 1, -1, -1}};

int oracle_wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* u, int64_t m, int64_t c,
                            int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha, sae_stream_t stream) {
    (void)stream;
    if (m < 1 || c < 1 || !w || !u) return set_err("oracle_wino_weights_f32: bad argument");
    for (int64_t mi = 0; mi < m; ++mi)
        for (int64_t ci = 0; ci < c; ++ci) {
            const float* wp = w + mi * w_stride_m + ci * w_stride_c;
            double g[3][3];
            for (int t = 0; t < 9; ++t)
                g[t / 3][t % 3] = (double)alpha * (double)wp[flip ? 8 - t : t] * (row_scale ? (double)row_scale[mi] : 1.0) *
                                  (col_scale ? (double)col_scale[ci] : 1.0);
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    double acc = 0.0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) acc += WINO_G[a][i] * g[i][j] * WINO_G[b][j];
                    u[(int64_t)(4 * a + b) * m * c + mi * c + ci] = (float)acc;
                }
        }
    return SAE_OK;
}

int oracle_wino_input_f32(const float* x, const float* plane_scale, float* v, int64_t planes, int64_t h, int64_t w, int32_t pad,
                          sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || pad < 0 || pad > 2 || h < 1 || w < 1 || (h & 1) || (w & 1) || h + 2 * pad < 4 || w + 2 * pad < 4)
        return set_err("oracle_wino_input_f32: the map must have even sides and pad 0, 1 or 2");
    if (planes == 0) return SAE_OK;
    if (!x || !v) return set_err("oracle_wino_input_f32: null tensor");
    const int64_t th = (h + 2 * pad - 2) / 2, tw = (w + 2 * pad - 2) / 2, tiles = th * tw;
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t ty = 0; ty < th; ++ty)
            for (int64_t tx = 0; tx < tw; ++tx) {
                double d[4][4];
                for (int r = 0; r < 4; ++r)
                    for (int q = 0; q < 4; ++q) {
                        const int64_t iy = 2 * ty - pad + r, ix = 2 * tx - pad + q;
                        const int in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                        d[r][q] = in ? (double)(plane_scale ? x[(p * h + iy) * w + ix] * plane_scale[p] : x[(p * h + iy) * w + ix]) : 0.0;
                    }
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        double acc = 0.0;
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j) acc += WINO_BT[a][i] * d[i][j] * WINO_BT[b][j];
                        v[((int64_t)(4 * a + b) * planes + p) * tiles + ty * tw + tx] = (float)acc;
                    }
            }
    return SAE_OK;
}

int oracle_wino_output_f32(const float* md, const float* plane_scale, const float* noise, const float* noise_weight,
                           const float* bias, float* y, int64_t planes, int64_t channels, int64_t h, int64_t w, int32_t act,
                           float slope, float act_scale, sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || channels < 1 || h < 2 || w < 2 || (h & 1) || (w & 1)) return set_err("oracle_wino_output_f32: bad shape");
    if (planes == 0) return SAE_OK;
    if (!md || !y) return set_err("oracle_wino_output_f32: null tensor");
    if (noise && (!act || !noise_weight || planes % channels != 0)) return set_err("oracle_wino_output_f32: noise without its epilogue");
    const int64_t th = h / 2, tw = w / 2, tiles = th * tw;
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t ty = 0; ty < th; ++ty)
            for (int64_t tx = 0; tx < tw; ++tx)
                for (int a = 0; a < 2; ++a)
                    for (int b = 0; b < 2; ++b) {
                        double acc = 0.0;
                        for (int i = 0; i < 4; ++i)
                            for (int j = 0; j < 4; ++j)
                                acc += WINO_AT[a][i] * (double)md[((int64_t)(4 * i + j) * planes + p) * tiles + ty * tw + tx] * WINO_AT[b][j];
                        float o = plane_scale ? (float)acc * plane_scale[p] : (float)acc;
                        if (act) {
                            if (noise) o = o + noise_weight[0] * noise[((p / channels) * h + 2 * ty + a) * w + 2 * tx + b];
                            o = o + (bias ? bias[p % channels] : 0.0f);
                            o = ((o > 0.0f) ? o : o * slope) * act_scale;
                        }
                        y[(p * h + 2 * ty + a) * w + 2 * tx + b] = o;
                    }
    return SAE_OK;
}

int64_t oracle_wino_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w) {
    (void)n; (void)c; (void)m; (void)tiles_h; (void)tiles_w;
    return 0;
}

/* md[xi][n][m][t] = sum_c u[xi][m][c] v[xi][n][c][t], double accumulation */
int oracle_wino_gemm_f32(const float* v, const float* u, float* md, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                         int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    (void)workspace; (void)workspace_floats; (void)stream;
    if (n < 0 || c < 1 || m < 1 || tiles_h < 1 || tiles_w < 1) return set_err("oracle_wino_gemm_f32: bad shape");
    if (n == 0) return SAE_OK;
    if (!v || !u || !md) return set_err("oracle_wino_gemm_f32: null tensor");
    const int64_t t = tiles_h * tiles_w;
#pragma omp parallel for collapse(2)
    for (int64_t xi = 0; xi < 16; ++xi)
        for (int64_t ni = 0; ni < n; ++ni)
            for (int64_t mi = 0; mi < m; ++mi)
                for (int64_t ti = 0; ti < t; ++ti) {
                    double acc = 0.0;
                    for (int64_t ci = 0; ci < c; ++ci)
                        acc += (double)u[(xi * m + mi) * c + ci] * (double)v[((xi * n + ni) * c + ci) * t + ti];
                    md[((xi * n + ni) * m + mi) * t + ti] = (float)acc;
                }
    return SAE_OK;
}

/* ---- weight gradient on the sixteen points: restated from the matrices (A = the transpose of WINO_AT, G = WINO_G) */
int oracle_wino_gy_f32(const float* gy, const float* plane_scale, float* e, int64_t planes, int64_t h, int64_t w, sae_stream_t stream) {
    (void)stream;
    if (planes < 0 || h < 2 || w < 2 || (h & 1) || (w & 1)) return set_err("oracle_wino_gy_f32: the map must have even sides");
    if (planes == 0) return SAE_OK;
    if (!gy || !e) return set_err("oracle_wino_gy_f32: null tensor");
    const int64_t th = h / 2, tw = w / 2, tiles = th * tw;
    for (int64_t p = 0; p < planes; ++p)
        for (int64_t ty = 0; ty < th; ++ty)
            for (int64_t tx = 0; tx < tw; ++tx)
                for (int a = 0; a < 4; ++a)
                    for (int b = 0; b < 4; ++b) {
                        double acc = 0.0;
                        for (int i = 0; i < 2; ++i)
                            for (int j = 0; j < 2; ++j) {
                                const float g = gy[(p * h + 2 * ty + i) * w + 2 * tx + j];
                                acc += WINO_AT[i][a] * (double)(plane_scale ? g * plane_scale[p] : g) * WINO_AT[j][b];
                            }
                        e[((int64_t)(4 * a + b) * planes + p) * tiles + ty * tw + tx] = (float)acc;
                    }
    return SAE_OK;
}

int64_t oracle_wino_wgrad_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w) {
    (void)n; (void)c; (void)m; (void)tiles_h; (void)tiles_w;
    return 0;
}

int oracle_wino_wgrad_gemm_f32(const float* v, const float* e, float* gu, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                               int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    (void)workspace; (void)workspace_floats; (void)stream;
    if (n < 0 || c < 1 || m < 1 || tiles_h < 1 || tiles_w < 1 || !gu || (n > 0 && (!v || !e)))
        return set_err("oracle_wino_wgrad_gemm_f32: bad argument");
    const int64_t t = tiles_h * tiles_w;
#pragma omp parallel for collapse(2)
    for (int64_t xi = 0; xi < 16; ++xi)
        for (int64_t mi = 0; mi < m; ++mi)
            for (int64_t ci = 0; ci < c; ++ci) {
                double acc = 0.0;
                for (int64_t ni = 0; ni < n; ++ni)
                    for (int64_t ti = 0; ti < t; ++ti)
                        acc += (double)e[((xi * n + ni) * m + mi) * t + ti] * (double)v[((xi * n + ni) * c + ci) * t + ti];
                gu[(xi * m + mi) * c + ci] = (float)acc;
            }
    return SAE_OK;
}

int oracle_wino_wgrad_output_f32(const float* gu, float* gw, int64_t m, int64_t c, int64_t w_stride_m, int64_t w_stride_c, float alpha,
                                 sae_stream_t stream) {
    (void)stream;
    if (m < 1 || c < 1 || !gu || !gw) return set_err("oracle_wino_wgrad_output_f32: bad argument");
    for (int64_t mi = 0; mi < m; ++mi)
        for (int64_t ci = 0; ci < c; ++ci)
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double acc = 0.0;
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b)
                            acc += WINO_G[a][i] * (double)gu[((int64_t)(4 * a + b) * m + mi) * c + ci] * WINO_G[b][j];
                    gw[mi * w_stride_m + ci * w_stride_c + 3 * i + j] = (float)((double)alpha * acc);
                }
    return SAE_OK;
}

/* ---- fused Winograd entry points (include/sae_hip.h: sae_wino_fused_*; csrc/winograd_fused.hip).  The prepared weights are the
 * transform-domain weights U[xi][m][c] = alpha (G g G^T)[xi] in the layout the header documents:
 *     uf[((mb * chunks + chunk) * 16 + xi) * 512 + (half * 64 + ml) * 4 + s],  m = 64 mb + ml,  c = 8 chunk + 4 half + s,
 * zero beyond m / c.  The convolution is restated from the published matrices in double: V = B^T d B of each 4x4 patch,
 * M[xi] = sum_c U[xi][m][c] V[xi][c], Y = A^T M A, then the epilogue of oracle_wino_output_f32. */
int64_t oracle_wino_fused_weights_floats(int64_t m, int64_t c) {
    if (m < 1 || c < 1) return 0;
    return ((m + 63) / 64) * ((c + 7) / 8) * 8192;
}

int oracle_wino_fused_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t m, int64_t c,
                                  int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha, sae_stream_t stream) {
    (void)stream;
    if (m < 1 || c < 1 || !w || !uf) return set_err("oracle_wino_fused_weights_f32: bad argument");
    const int64_t chunks = (c + 7) / 8, mbs = (m + 63) / 64;
    memset(uf, 0, sizeof(float) * (size_t)(mbs * chunks * 8192));
    for (int64_t mi = 0; mi < m; ++mi)
        for (int64_t ci = 0; ci < c; ++ci) {
            const float* wp = w + mi * w_stride_m + ci * w_stride_c;
            double g[3][3];
            for (int t = 0; t < 9; ++t)
                g[t / 3][t % 3] = (double)alpha * (double)wp[flip ? 8 - t : t] * (row_scale ? (double)row_scale[mi] : 1.0) *
                                  (col_scale ? (double)col_scale[ci] : 1.0);
            const int64_t mb = mi / 64, ml = mi % 64, chunk = ci / 8, half = (ci % 8) / 4, s = ci % 4;
            for (int a = 0; a < 4; ++a)
                for (int b = 0; b < 4; ++b) {
                    double acc = 0.0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) acc += WINO_G[a][i] * g[i][j] * WINO_G[b][j];
                    uf[((mb * chunks + chunk) * 16 + (4 * a + b)) * 512 + (half * 64 + ml) * 4 + s] = (float)acc;
                }
        }
    return SAE_OK;
}

int oracle_wino_fused_conv_f32(const float* x, const float* x_scale, const float* uf, const float* out_scale, const float* noise,
                               const float* noise_weight, const float* bias, float* y, int64_t n, int64_t c, int64_t m, int64_t h,
                               int64_t w, int32_t pad, int32_t act, float slope, float act_scale, sae_stream_t stream) {
    (void)stream;
    if (n < 0 || c < 1 || m < 1 || pad < 0 || pad > 2 || h < 1 || w < 1 || (h & 1) || (w & 1) || h + 2 * pad < 4 || w + 2 * pad < 4)
        return set_err("oracle_wino_fused_conv_f32: the map must have even sides and pad 0, 1 or 2");
    if (n == 0) return SAE_OK;
    if (!x || !uf || !y) return set_err("oracle_wino_fused_conv_f32: null tensor");
    if (noise && (!act || !noise_weight)) return set_err("oracle_wino_fused_conv_f32: noise without its epilogue");
    const int64_t oh = h + 2 * pad - 2, ow = w + 2 * pad - 2, th = oh / 2, tw = ow / 2, chunks = (c + 7) / 8;
    double* V = (double*)malloc(sizeof(double) * (size_t)(16 * c));
    if (!V) return set_err("oracle_wino_fused_conv_f32: out of memory");
    for (int64_t ni = 0; ni < n; ++ni)
        for (int64_t ty = 0; ty < th; ++ty)
            for (int64_t tx = 0; tx < tw; ++tx) {
                for (int64_t ci = 0; ci < c; ++ci) {
                    double d[4][4];
                    const float* xp = x + (ni * c + ci) * h * w;
                    for (int r = 0; r < 4; ++r)
                        for (int q = 0; q < 4; ++q) {
                            const int64_t iy = 2 * ty - pad + r, ix = 2 * tx - pad + q;
                            const int in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                            d[r][q] = in ? (double)(x_scale ? xp[iy * w + ix] * x_scale[ni * c + ci] : xp[iy * w + ix]) : 0.0;
                        }
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) {
                            double acc = 0.0;
                            for (int i = 0; i < 4; ++i)
                                for (int j = 0; j < 4; ++j) acc += WINO_BT[a][i] * d[i][j] * WINO_BT[b][j];
                            V[(4 * a + b) * c + ci] = acc;
                        }
                }
                for (int64_t mi = 0; mi < m; ++mi) {
                    double M[16];
                    const int64_t mb = mi / 64, ml = mi % 64;
                    for (int xi = 0; xi < 16; ++xi) {
                        double acc = 0.0;
                        for (int64_t ci = 0; ci < c; ++ci)
                            acc += (double)uf[((mb * chunks + ci / 8) * 16 + xi) * 512 + (((ci % 8) / 4) * 64 + ml) * 4 + ci % 4] *
                                   V[xi * c + ci];
                        M[xi] = acc;
                    }
                    for (int a = 0; a < 2; ++a)
                        for (int b = 0; b < 2; ++b) {
                            double acc = 0.0;
                            for (int i = 0; i < 4; ++i)
                                for (int j = 0; j < 4; ++j) acc += WINO_AT[a][i] * M[4 * i + j] * WINO_AT[b][j];
                            float o = out_scale ? (float)acc * out_scale[ni * m + mi] : (float)acc;
                            if (act) {
                                if (noise) o = o + noise_weight[0] * noise[(ni * oh + 2 * ty + a) * ow + 2 * tx + b];
                                o = o + (bias ? bias[mi] : 0.0f);
                                o = ((o > 0.0f) ? o : o * slope) * act_scale;
                            }
                            y[((ni * m + mi) * oh + 2 * ty + a) * ow + 2 * tx + b] = o;
                        }
                }
            }
    free(V);
    return SAE_OK;
}

/* sae_wino_fused_wgrad_f32, restated from the published matrices in double: gU[xi][m][c] = sum over images and tiles of
 * (A e A^T)[xi] (B^T d B)[xi], gw = alpha G^T gU G.  The workspace is not used (same size contract as the product's). */
int64_t oracle_wino_fused_wgrad_workspace(int64_t n, int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad) {
    if (n < 1 || c < 1 || m < 1 || h < 2 || w < 2 || pad < 0 || pad > 1) return 0;
    const int64_t tw = (w + 2 * pad - 2) / 2;
    if (tw % 8 != 0) return 0;
    return 1;
}

int oracle_wino_fused_wgrad_f32(const float* x, const float* x_scale, const float* gy, const float* y_scale, float* gw, int64_t n,
                                int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad, int64_t w_stride_m, int64_t w_stride_c,
                                float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    (void)stream; (void)workspace; (void)workspace_floats;
    if (n < 1 || c < 1 || m < 1 || pad < 0 || pad > 1 || (h & 1) || (w & 1) || h + 2 * pad < 4)
        return set_err("oracle_wino_fused_wgrad_f32: bad shape");
    const int64_t oh = h + 2 * pad - 2, ow = w + 2 * pad - 2, th = oh / 2, tw = ow / 2;
    if (tw < 8 || tw % 8 != 0) return set_err("oracle_wino_fused_wgrad_f32: output rows of a multiple of 16 pixels");
    if (!x || !gy || !gw) return set_err("oracle_wino_fused_wgrad_f32: null tensor");
    static const double A4[4][2] = {{1, 0}, {1, 1}, {1, -1}, {0, -1}};
    #pragma omp parallel for collapse(2)
    for (int64_t mi = 0; mi < m; ++mi)
        for (int64_t ci = 0; ci < c; ++ci) {
            double gU[16];
            for (int xi = 0; xi < 16; ++xi) gU[xi] = 0.0;
            for (int64_t ni = 0; ni < n; ++ni) {
                const float* xp = x + (ni * c + ci) * h * w;
                const float* gp = gy + (ni * m + mi) * oh * ow;
                const float sx = x_scale ? x_scale[ni * c + ci] : 1.0f, sy = y_scale ? y_scale[ni * m + mi] : 1.0f;
                for (int64_t ty = 0; ty < th; ++ty)
                    for (int64_t tx = 0; tx < tw; ++tx) {
                        double d[4][4], e[2][2];
                        for (int r = 0; r < 4; ++r)
                            for (int q = 0; q < 4; ++q) {
                                const int64_t iy = 2 * ty - pad + r, ix = 2 * tx - pad + q;
                                const int in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                                d[r][q] = in ? (double)(x_scale ? xp[iy * w + ix] * sx : xp[iy * w + ix]) : 0.0;
                            }
                        for (int a = 0; a < 2; ++a)
                            for (int b = 0; b < 2; ++b)
                                e[a][b] = (double)(y_scale ? gp[(2 * ty + a) * ow + 2 * tx + b] * sy : gp[(2 * ty + a) * ow + 2 * tx + b]);
                        for (int a = 0; a < 4; ++a)
                            for (int b = 0; b < 4; ++b) {
                                double E = 0.0, V = 0.0;
                                for (int i = 0; i < 2; ++i)
                                    for (int j = 0; j < 2; ++j) E += A4[a][i] * e[i][j] * A4[b][j];
                                for (int i = 0; i < 4; ++i)
                                    for (int j = 0; j < 4; ++j) V += WINO_BT[a][i] * d[i][j] * WINO_BT[b][j];
                                gU[4 * a + b] += E * V;
                            }
                    }
            }
            for (int i = 0; i < 3; ++i)
                for (int j = 0; j < 3; ++j) {
                    double acc = 0.0;
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) acc += WINO_G[a][i] * gU[4 * a + b] * WINO_G[b][j];
                    gw[mi * w_stride_m + ci * w_stride_c + 3 * i + j] = (float)((double)alpha * acc);
                }
        }
    return SAE_OK;
}


/* ------------------------------------------------------------------------------------------------------------------------------
 * sae_s2wino_* (csrc/s2wino.hip): the stride-2 3x3 family on its polyphase minimal-filtering form, restated in double from the
 * identities themselves.  Along an axis, with the taps t0 t1 t2 as the product sees them (flip = 1: reversed), a tile takes
 * the inputs (a, b, c) = g[2t-1], g[2t], g[2t+1] to the outputs
 *     dx[4t]   = (a - b) t0 + b (t0 + t2)          dx[4t+2] = b (t0 + t2) + (c - b) t2         (F(2,2) on the 2-tap even filter)
 *     dx[4t+1] = b t1                              dx[4t+3] = c t1                             (the 1-tap odd filter)
 * i.e. five products P = {(a-b) t0, b (t0+t2), (c-b) t2, b t1, c t1}; in the plane the 25 products of a tile are the tensor
 * product of the two axes.  The prepared weights here are the oracle's own layout: uf[(p * cout + co) * cin + ci], p = 5 py + px.
 * Reference: F.conv_transpose2d(stride 2) (models/networks/stylegan2_layers.py:306) / autograd's data gradient of
 * F.conv2d(stride 2) (:136). */
static const double S2W_IN[5][3] = {{1, -1, 0}, {0, 1, 0}, {0, -1, 1}, {0, 1, 0}, {0, 0, 1}};      /* point <- (a, b, c) */
static const double S2W_WT[5][3] = {{1, 0, 0}, {1, 0, 1}, {0, 0, 1}, {0, 1, 0}, {0, 1, 0}};       /* point <- (t0, t1, t2) */
static const double S2W_OUT[4][5] = {{1, 1, 0, 0, 0}, {0, 0, 0, 1, 0}, {0, 1, 1, 0, 0}, {0, 0, 0, 0, 1}};   /* dx[4t + o] <- points */

int64_t oracle_s2wino_weights_floats(int64_t cout, int64_t cin) {
    if (cout < 1 || cin < 1) return 0;
    return 25 * cout * cin;
}

int oracle_s2wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t cout, int64_t cin,
                              int64_t w_stride_out, int64_t w_stride_in, int32_t flip, float alpha, sae_stream_t stream) {
    (void)stream;
    if (cout < 1 || cin < 1 || !w || !uf) return set_err("oracle_s2wino_weights_f32: bad argument");
    for (int64_t co = 0; co < cout; ++co)
        for (int64_t ci = 0; ci < cin; ++ci) {
            const float* wp = w + co * w_stride_out + ci * w_stride_in;
            double t[3][3];
            for (int k = 0; k < 9; ++k)
                t[k / 3][k % 3] = (double)alpha * (double)wp[flip ? 8 - k : k] * (row_scale ? (double)row_scale[co] : 1.0) *
                                  (col_scale ? (double)col_scale[ci] : 1.0);
            for (int py = 0; py < 5; ++py)
                for (int px = 0; px < 5; ++px) {
                    double acc = 0.0;
                    for (int i = 0; i < 3; ++i)
                        for (int j = 0; j < 3; ++j) acc += S2W_WT[py][i] * t[i][j] * S2W_WT[px][j];
                    uf[((5 * py + px) * cout + co) * cin + ci] = (float)acc;
                }
        }
    return SAE_OK;
}

int oracle_s2wino_dgrad_f32(const float* g, const float* g_scale, const float* uf, const float* out_scale, float* dx, int64_t n,
                            int64_t cin, int64_t cout, int64_t h, int64_t w, sae_stream_t stream) {
    (void)stream;
    if (n < 0 || cin < 1 || cout < 1 || h < 2 || w < 4 || (h & 1) || (w & 1))
        return set_err("oracle_s2wino_dgrad_f32: the small side must have even sides and rows of at least 4 floats");
    if (n == 0) return SAE_OK;
    if (!g || !uf || !dx) return set_err("oracle_s2wino_dgrad_f32: null tensor");
    const int64_t oh = 2 * h + 1, ow = 2 * w + 1, th = h / 2 + 1, tw = w / 2 + 1;
    #pragma omp parallel for collapse(2)
    for (int64_t ni = 0; ni < n; ++ni)
        for (int64_t ty = 0; ty < th; ++ty) {
            double* V = (double*)malloc(sizeof(double) * (size_t)(25 * cin));
            for (int64_t tx = 0; tx < tw; ++tx) {
                for (int64_t ci = 0; ci < cin; ++ci) {
                    double d[3][3];
                    const float* gp = g + (ni * cin + ci) * h * w;
                    for (int r = 0; r < 3; ++r)
                        for (int q = 0; q < 3; ++q) {
                            const int64_t iy = 2 * ty - 1 + r, ix = 2 * tx - 1 + q;
                            const int in = iy >= 0 && iy < h && ix >= 0 && ix < w;
                            d[r][q] = in ? (double)(g_scale ? gp[iy * w + ix] * g_scale[ni * cin + ci] : gp[iy * w + ix]) : 0.0;
                        }
                    for (int py = 0; py < 5; ++py)
                        for (int px = 0; px < 5; ++px) {
                            double acc = 0.0;
                            for (int i = 0; i < 3; ++i)
                                for (int j = 0; j < 3; ++j) acc += S2W_IN[py][i] * d[i][j] * S2W_IN[px][j];
                            V[(5 * py + px) * cin + ci] = acc;
                        }
                }
                for (int64_t co = 0; co < cout; ++co) {
                    double M[25];
                    for (int pt = 0; pt < 25; ++pt) {
                        double acc = 0.0;
                        for (int64_t ci = 0; ci < cin; ++ci) acc += (double)uf[(pt * cout + co) * cin + ci] * V[pt * cin + ci];
                        M[pt] = acc;
                    }
                    for (int a = 0; a < 4; ++a)
                        for (int b = 0; b < 4; ++b) {
                            const int64_t oy = 4 * ty + a, ox = 4 * tx + b;
                            if (oy >= oh || ox >= ow) continue;
                            double acc = 0.0;
                            for (int i = 0; i < 5; ++i)
                                for (int j = 0; j < 5; ++j) acc += S2W_OUT[a][i] * M[5 * i + j] * S2W_OUT[b][j];
                            dx[((ni * cout + co) * oh + oy) * ow + ox] = out_scale ? (float)acc * out_scale[ni * cout + co] : (float)acc;
                        }
                }
            }
            free(V);
        }
    return SAE_OK;
}
