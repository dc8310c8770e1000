// Elementwise glue of the generator's StyledConv, fused:
//
//   noise_bias_act      y = lrelu(x + w_noise * noise[n, hw] + bias[c]) * scale
//                       = NoiseInjection.forward + FusedLeakyReLU.forward
//                         (models/networks/stylegan2_layers.py:340-351 and :54-65 / fused_act.py:75-86)
//                       in one pass (8 B per element instead of 16)
//   noise_bias_act_bwd  gx = (y > 0 ? gy : alpha gy) * scale,  gbias[c] = sum gx,  gw_noise = sum gx * noise
//                       (autograd of the two modules: a leaky-ReLU backward pass, a broadcast multiply,
//                       two reductions) in one pass (12 B per element instead of ~28)
//   plane_scale_dot     gx = g * s[plane],  gs[plane] = sum_hw g * x
//                       = backward of the style modulation  x * s[:, :, None, None]
//                         (ModulatedConv2d.forward, stylegan2_layers.py:280-286) in one pass
//                       (12 B per element instead of 24)
// All reductions are two-stage with a fixed order (deterministic, no atomics).
#include "sae_common.h"

namespace sae {
namespace {

// The reductions here are long, heavily cancelling sums (the gradient of ONE scalar over N*C*H*W terms):
// they accumulate in double, which is free on an HBM-bound pass.
__device__ __forceinline__ double wave_sum_m(double v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the workgroup, result valid in thread 0
__device__ __forceinline__ double block_sum_m(double v, double* red) {
    v = wave_sum_m(v);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) red[wid] = v;
    __syncthreads();
    double t = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
    }
    return t;
}

// x, y: [outer][channels][hw]; noise: [outer][hw]; hw % 4 == 0
__global__ __launch_bounds__(kBlock) void noise_bias_act_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ noise,
                                                                const float* __restrict__ noise_weight,
                                                                const float* __restrict__ bias,
                                                                float* __restrict__ y, int64_t nvec, int hw4,
                                                                int channels, float alpha, float scale) {
    const float wn = noise ? noise_weight[0] : 0.0f;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v0 = (int64_t)blockIdx.x * kBlock + threadIdx.x; v0 < nvec; v0 += stride * 4) {
        f32x4 xv[4], nv[4];
        float bv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + stride * u;
            if (v < nvec) {
                const int64_t plane = v / hw4;
                const int p4 = (int)(v - plane * hw4);
                const int64_t n = plane / channels;
                const int c = (int)(plane - n * channels);
                xv[u] = reinterpret_cast<const f32x4*>(x)[v];
                nv[u] = noise ? reinterpret_cast<const f32x4*>(noise)[n * hw4 + p4] : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                bv[u] = bias ? bias[c] : 0.0f;
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int64_t v = v0 + stride * u;
            if (v < nvec) {
                f32x4 o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // same association as the reference: (image + weight * noise) + bias
                    const float t = (xv[u][e] + wn * nv[u][e]) + bv[u];
                    o[e] = ((t > 0.0f) ? t : t * alpha) * scale;
                }
                reinterpret_cast<f32x4*>(y)[v] = o;
            }
        }
    }
}

// block (c, s) walks the chunks (n, chunk-of-plane) s, s + S, ... of channel c; partial_b[c*S + s],
// partial_n[c*S + s]
__global__ __launch_bounds__(kBlock) void noise_bias_act_bwd_kernel(
    const float* __restrict__ gy, const float* __restrict__ yref, const float* __restrict__ noise,
    float* __restrict__ gx, double* __restrict__ partial_b, double* __restrict__ partial_n, int64_t outer, int hw,
    int channels, int chunks_per_plane, int nsplit, float alpha, float scale) {
    __shared__ double red[kBlock / kWave];
    constexpr int kChunk = kBlock * 4 * 2;
    const int c = blockIdx.x;
    const int s = blockIdx.y;
    const int64_t items = outer * chunks_per_plane;
    double acc_b = 0.0, acc_n = 0.0;
    for (int64_t it = s; it < items; it += nsplit) {
        const int64_t n = it / chunks_per_plane;
        const int ch = (int)(it - n * chunks_per_plane);
        const int64_t plane = (n * channels + c) * hw;
        const int64_t nplane = n * hw;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = ch * kChunk + (u * kBlock + threadIdx.x) * 4;
            if (e < hw) {
                const f32x4 g = *reinterpret_cast<const f32x4*>(gy + plane + e);
                const f32x4 r = *reinterpret_cast<const f32x4*>(yref + plane + e);
                const f32x4 z = noise ? *reinterpret_cast<const f32x4*>(noise + nplane + e) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) o[q] = ((r[q] > 0.0f) ? g[q] : g[q] * alpha) * scale;
                *reinterpret_cast<f32x4*>(gx + plane + e) = o;
                acc_b += (o[0] + o[1]) + (o[2] + o[3]);
                acc_n += (o[0] * z[0] + o[1] * z[1]) + (o[2] * z[2] + o[3] * z[3]);
            }
        }
    }
    const double tb = block_sum_m(acc_b, red);
    const double tn = block_sum_m(acc_n, red);
    if (threadIdx.x == 0) {
        partial_b[c * nsplit + s] = tb;
        partial_n[c * nsplit + s] = tn;
    }
}

// gb[c] = sum_q partial_b[c*Q + q] (one wave per channel); the LAST workgroup additionally reduces all of partial_n
// (channels * Q values, fixed order, four independent chains per thread) into gw[0]
__global__ __launch_bounds__(kBlock) void noise_bias_finalize_kernel(const double* __restrict__ partial_b,
                                                                     const double* __restrict__ partial_n,
                                                                     float* __restrict__ gb, float* __restrict__ gw,
                                                                     int channels, int q_count) {
    __shared__ double red[kBlock / kWave];
    if (blockIdx.x == gridDim.x - 1) {       // its own workgroup: the channel sums do not wait behind it
        if (!gw) return;
        double t0 = 0.0, t1 = 0.0, t2 = 0.0, t3 = 0.0;
        const int total = channels * q_count;
        int i = threadIdx.x;
        for (; i + 3 * kBlock < total; i += 4 * kBlock) {
            t0 += partial_n[i];
            t1 += partial_n[i + kBlock];
            t2 += partial_n[i + 2 * kBlock];
            t3 += partial_n[i + 3 * kBlock];
        }
        for (; i < total; i += kBlock) t0 += partial_n[i];
        const double t = block_sum_m((t0 + t1) + (t2 + t3), red);
        if (threadIdx.x == 0) gw[0] = (float)t;
        return;
    }
    const int lane = threadIdx.x & (kWave - 1);
    const int c = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    double acc = 0.0;
    if (c < channels)
        for (int q = lane; q < q_count; q += kWave) acc += partial_b[c * q_count + q];
    acc = wave_sum_m(acc);
    if (gb && c < channels && lane == 0) gb[c] = (float)acc;
}

// one workgroup per (n, c) plane
__global__ __launch_bounds__(kBlock) void plane_scale_dot_kernel(const float* __restrict__ g,
                                                                 const float* __restrict__ x,
                                                                 const float* __restrict__ s,
                                                                 float* __restrict__ gx, float* __restrict__ gs,
                                                                 int hw) {
    __shared__ double red[kBlock / kWave];
    const int64_t plane = blockIdx.x;
    const float sc = s[plane];
    const float* gp = g + plane * hw;
    const float* xp = x + plane * hw;
    float* op = gx + plane * hw;
    double acc = 0.0;
    for (int e0 = 0; e0 < hw; e0 += kBlock * 4 * 4) {
        f32x4 gv[4], xv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + (u * kBlock + threadIdx.x) * 4;
            if (e < hw) {
                gv[u] = *reinterpret_cast<const f32x4*>(gp + e);
                xv[u] = *reinterpret_cast<const f32x4*>(xp + e);
            }
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int e = e0 + (u * kBlock + threadIdx.x) * 4;
            if (e < hw) {
                *reinterpret_cast<f32x4*>(op + e) = gv[u] * sc;
                acc += (gv[u][0] * xv[u][0] + gv[u][1] * xv[u][1]) + (gv[u][2] * xv[u][2] + gv[u][3] * xv[u][3]);
            }
        }
    }
    const double t = block_sum_m(acc, red);
    if (threadIdx.x == 0) gs[plane] = (float)t;
}

// Demodulation factor of ModulatedConv2d (stylegan2_layers.py:290-292, from the un-modulated weight):
//   d[o] = rsqrt(sum_j (alpha w[o][j])^2 + eps),   j over (in channel, ky, kx)
// one workgroup per output channel; the reference's five ATen launches (mul, pow, sum, add, rsqrt over the [O, I, k, k]
// weight, 9 MB at 512 x 512 x 3 x 3) in one pass over the weight.
__global__ __launch_bounds__(kBlock) void weight_demod_kernel(const float* __restrict__ w, float* __restrict__ d, int64_t cols,
                                                              float alpha, float eps) {
    __shared__ double red[kBlock / kWave];
    const float* wr = w + (int64_t)blockIdx.x * cols;
    double a0 = 0.0, a1 = 0.0;
    int64_t j = threadIdx.x;
    for (; j + kBlock < cols; j += 2 * kBlock) {
        const float u = wr[j] * alpha, v = wr[j + kBlock] * alpha;       // scale first, then square: the reference's order
        a0 += (double)(u * u);
        a1 += (double)(v * v);
    }
    if (j < cols) {
        const float u = wr[j] * alpha;
        a0 += (double)(u * u);
    }
    const double t = block_sum_m(a0 + a1, red);
    if (threadIdx.x == 0) d[blockIdx.x] = 1.0f / sqrtf((float)t + eps);
}

// Its backward together with the product rule of the effective weight W_eff = alpha d[o] w: with geff = alpha dL/dW_eff
// (what the weight-gradient kernels return),
//   dL/dd[o] = sum_j geff[o][j] w[o][j],     dd[o]/dw[o][j] = -d[o]^3 alpha^2 w[o][j]
//   gw[o][j] = d[o] geff[o][j] - (sum_j geff w) d[o]^3 alpha^2 w[o][j]
// (autograd spent ~15 launches on this per modulated conv: geff * d, (geff * w).sum, the chain through rsqrt / sum / pow /
// mul, two gradient accumulations and the select-backward of weight[0]).
__global__ __launch_bounds__(kBlock) void weight_demod_bwd_kernel(const float* __restrict__ geff, const float* __restrict__ w,
                                                                  const float* __restrict__ d, float* __restrict__ gw,
                                                                  int64_t cols, float alpha) {
    __shared__ double red[kBlock / kWave];
    __shared__ float coef;
    const int64_t base = (int64_t)blockIdx.x * cols;
    double a0 = 0.0, a1 = 0.0;
    int64_t j = threadIdx.x;
    for (; j + kBlock < cols; j += 2 * kBlock) {
        a0 += (double)geff[base + j] * w[base + j];
        a1 += (double)geff[base + j + kBlock] * w[base + j + kBlock];
    }
    if (j < cols) a0 += (double)geff[base + j] * w[base + j];
    const double t = block_sum_m(a0 + a1, red);
    const float dv = d[blockIdx.x];
    if (threadIdx.x == 0) coef = (float)t * (dv * dv * dv) * (alpha * alpha);
    __syncthreads();
    const float cf = coef;
    for (int64_t q = threadIdx.x; q < cols; q += kBlock) gw[base + q] = dv * geff[base + q] - cf * w[base + q];
}

// plane_scale_dot followed by the noise + bias + activation backward of the layer that PRODUCED x, in one pass: inside a
// generator block conv2's input x is conv1's activated output, so the gradient g * s that the style modulation's backward hands
// to conv1 meets x again as the activation reference --
//   gs[plane] = sum_hw g x;   gx = (x > 0 ? g s : alpha g s) * scale;   partial_b[c][n] = sum_hw gx;   partial_n[c][n] = sum_hw gx noise
// (12 B per element instead of 12 + 12: the intermediate g * s is never written).  Same arithmetic per element as the two
// kernels it replaces (g * s rounded to float first).  One workgroup per plane.
__global__ __launch_bounds__(kBlock) void plane_scale_dot_act_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                     const float* __restrict__ s, const float* __restrict__ noise,
                                                                     float* __restrict__ gx, float* __restrict__ gs,
                                                                     double* __restrict__ partial_b, double* __restrict__ partial_n,
                                                                     int hw, int channels, int outer, float alpha, float scale) {
    __shared__ double red[kBlock / kWave];
    const int64_t plane = blockIdx.x;
    const int64_t n = plane / channels;
    const int c = (int)(plane - n * channels);
    const float sc = s[plane];
    const float* gp = g + plane * hw;
    const float* xp = x + plane * hw;
    const float* zp = noise ? noise + n * hw : nullptr;
    float* op = gx + plane * hw;
    double acc_s = 0.0, acc_b = 0.0, acc_n = 0.0;
    for (int e0 = 0; e0 < hw; e0 += kBlock * 4 * 2) {
        f32x4 gv[2], xv[2], zv[2];
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = e0 + (u * kBlock + threadIdx.x) * 4;
            if (e < hw) {
                gv[u] = *reinterpret_cast<const f32x4*>(gp + e);
                xv[u] = *reinterpret_cast<const f32x4*>(xp + e);
                zv[u] = zp ? *reinterpret_cast<const f32x4*>(zp + e) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            }
        }
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int e = e0 + (u * kBlock + threadIdx.x) * 4;
            if (e < hw) {
                f32x4 o;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const float t = gv[u][q] * sc;
                    o[q] = ((xv[u][q] > 0.0f) ? t : t * alpha) * scale;
                }
                *reinterpret_cast<f32x4*>(op + e) = o;
                acc_s += (gv[u][0] * xv[u][0] + gv[u][1] * xv[u][1]) + (gv[u][2] * xv[u][2] + gv[u][3] * xv[u][3]);
                acc_b += (o[0] + o[1]) + (o[2] + o[3]);
                acc_n += (o[0] * zv[u][0] + o[1] * zv[u][1]) + (o[2] * zv[u][2] + o[3] * zv[u][3]);
            }
        }
    }
    const double ts = block_sum_m(acc_s, red);
    const double tb = block_sum_m(acc_b, red);
    const double tn = block_sum_m(acc_n, red);
    if (threadIdx.x == 0) {
        gs[plane] = (float)ts;
        partial_b[(int64_t)c * outer + n] = tb;
        partial_n[(int64_t)c * outer + n] = tn;
    }
}

int bwd_nsplit(int64_t outer, int hw, int channels, int* chunks_per_plane) {
    const int chunk = kBlock * 4 * 2;
    *chunks_per_plane = (int)ceil_div64(hw, chunk);
    const int64_t items = outer * *chunks_per_plane;
    int64_t want = ceil_div64(2048, channels);
    if (want < 1) want = 1;
    if (want > items) want = items;
    if (want > 1024) want = 1024;
    return (int)want;
}

bool shape_ok(int64_t outer, int64_t channels, int64_t hw) {
    return outer >= 0 && channels >= 1 && hw >= 4 && hw % 4 == 0 && hw < ((int64_t)1 << 30) && channels < (1 << 24);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_noise_bias_act_f32(const float* x, const float* noise, const float* noise_weight,
                                      const float* bias, float* y, int64_t outer, int64_t channels, int64_t hw,
                                      float alpha, float scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!shape_ok(outer, channels, hw))
        return fail(SAE_EINVAL, "sae_noise_bias_act_f32: need hw %% 4 == 0, got [%lld, %lld, %lld]", (long long)outer,
                    (long long)channels, (long long)hw);
    if (outer == 0) return SAE_OK;
    if (!x || !y || (noise && !noise_weight)) return fail(SAE_EINVAL, "sae_noise_bias_act_f32: null tensor");
    if (!aligned16(x) || !aligned16(y) || (noise && !aligned16(noise)))
        return fail(SAE_EINVAL, "sae_noise_bias_act_f32: tensors must be 16-byte aligned");
    const int64_t nvec = outer * channels * hw / 4;
    int64_t blocks = ceil_div64(nvec, (int64_t)kBlock * 4);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(noise_bias_act_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, noise,
                       noise_weight, bias, y, nvec, (int)(hw / 4), (int)channels, alpha, scale);
    return check_launch("sae_noise_bias_act_f32");
}

extern "C" int64_t sae_noise_bias_act_bwd_workspace(int64_t outer, int64_t channels, int64_t hw) {
    if (!shape_ok(outer, channels, hw) || outer == 0) return 0;
    int cpp;
    return 4 * channels * (int64_t)bwd_nsplit(outer, (int)hw, (int)channels, &cpp) + 2;   // two double arrays
}

extern "C" int sae_noise_bias_act_bwd_f32(const float* gy, const float* y_ref, const float* noise, float* gx,
                                          float* gbias, float* gnoise_weight, float* workspace,
                                          int64_t workspace_floats, int64_t outer, int64_t channels, int64_t hw,
                                          float alpha, float scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!shape_ok(outer, channels, hw))
        return fail(SAE_EINVAL, "sae_noise_bias_act_bwd_f32: need hw %% 4 == 0, got [%lld, %lld, %lld]",
                    (long long)outer, (long long)channels, (long long)hw);
    hipStream_t s = (hipStream_t)stream;
    if (outer == 0) {
        if (gbias) hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)channels, s);
        if (gnoise_weight) hipMemsetAsync(gnoise_weight, 0, sizeof(float), s);
        return check_launch("sae_noise_bias_act_bwd_f32(memset)");
    }
    if (!gy || !y_ref || !gx) return fail(SAE_EINVAL, "sae_noise_bias_act_bwd_f32: null tensor");
    if (!aligned16(gy) || !aligned16(y_ref) || !aligned16(gx) || (noise && !aligned16(noise)))
        return fail(SAE_EINVAL, "sae_noise_bias_act_bwd_f32: tensors must be 16-byte aligned");
    int cpp;
    const int nsplit = bwd_nsplit(outer, (int)hw, (int)channels, &cpp);
    const int64_t need = 4 * channels * (int64_t)nsplit + 2;
    if (!workspace || workspace_floats < need)
        return fail(SAE_EWORKSPACE, "sae_noise_bias_act_bwd_f32: workspace %lld < %lld floats",
                    (long long)workspace_floats, (long long)need);
    double* pb = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 7) & ~(uintptr_t)7);
    double* pn = pb + channels * nsplit;
    hipLaunchKernelGGL(noise_bias_act_bwd_kernel, dim3((unsigned)channels, (unsigned)nsplit), dim3(kBlock), 0, s, gy,
                       y_ref, noise, gx, pb, pn, outer, (int)hw, (int)channels, cpp, nsplit, alpha, scale);
    hipLaunchKernelGGL(noise_bias_finalize_kernel, dim3((unsigned)ceil_div64(channels, kBlock / kWave) + 1), dim3(kBlock),
                       0, s, (const double*)pb, (const double*)pn, gbias, noise ? gnoise_weight : nullptr, (int)channels,
                       nsplit);
    return check_launch("sae_noise_bias_act_bwd_f32");
}

extern "C" int sae_plane_scale_dot_f32(const float* g, const float* x, const float* s, float* gx, float* gs,
                                       int64_t planes, int64_t hw, sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || hw < 4 || hw % 4 != 0 || hw >= ((int64_t)1 << 30) || planes >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_plane_scale_dot_f32: need hw %% 4 == 0, got planes=%lld hw=%lld",
                    (long long)planes, (long long)hw);
    if (planes == 0) return SAE_OK;
    if (!g || !x || !s || !gx || !gs) return fail(SAE_EINVAL, "sae_plane_scale_dot_f32: null tensor");
    if (!aligned16(g) || !aligned16(x) || !aligned16(gx))
        return fail(SAE_EINVAL, "sae_plane_scale_dot_f32: tensors must be 16-byte aligned");
    hipLaunchKernelGGL(plane_scale_dot_kernel, dim3((unsigned)planes), dim3(kBlock), 0, (hipStream_t)stream, g, x, s,
                       gx, gs, (int)hw);
    return check_launch("sae_plane_scale_dot_f32");
}

extern "C" int sae_weight_demod_f32(const float* w, float* d, int64_t rows, int64_t cols, float alpha, float eps,
                                    sae_stream_t stream) {
    sae::clear_stale_error();
    if (rows < 0 || cols < 1 || rows >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_weight_demod_f32: bad shape");
    if (rows == 0) return SAE_OK;
    if (!w || !d) return fail(SAE_EINVAL, "sae_weight_demod_f32: null tensor");
    hipLaunchKernelGGL(weight_demod_kernel, dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, w, d, cols, alpha, eps);
    return check_launch("sae_weight_demod_f32");
}

extern "C" int sae_weight_demod_bwd_f32(const float* geff, const float* w, const float* d, float* gw, int64_t rows,
                                        int64_t cols, float alpha, sae_stream_t stream) {
    sae::clear_stale_error();
    if (rows < 0 || cols < 1 || rows >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_weight_demod_bwd_f32: bad shape");
    if (rows == 0) return SAE_OK;
    if (!geff || !w || !d || !gw) return fail(SAE_EINVAL, "sae_weight_demod_bwd_f32: null tensor");
    hipLaunchKernelGGL(weight_demod_bwd_kernel, dim3((unsigned)rows), dim3(kBlock), 0, (hipStream_t)stream, geff, w, d, gw, cols,
                       alpha);
    return check_launch("sae_weight_demod_bwd_f32");
}

extern "C" int64_t sae_plane_scale_dot_act_workspace(int64_t outer, int64_t channels) {
    if (outer < 1 || channels < 1) return 0;
    return 4 * outer * channels + 2;       // two double arrays
}

extern "C" int sae_plane_scale_dot_act_f32(const float* g, const float* x, const float* s, const float* noise, float* gx, float* gs,
                                           float* gbias, float* gnoise_weight, float* workspace, int64_t workspace_floats,
                                           int64_t outer, int64_t channels, int64_t hw, float alpha, float scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!shape_ok(outer, channels, hw) || outer * channels >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_plane_scale_dot_act_f32: need hw %% 4 == 0, got [%lld, %lld, %lld]", (long long)outer,
                    (long long)channels, (long long)hw);
    hipStream_t st = (hipStream_t)stream;
    if (outer == 0) {
        if (gbias) hipMemsetAsync(gbias, 0, sizeof(float) * (size_t)channels, st);
        if (gnoise_weight) hipMemsetAsync(gnoise_weight, 0, sizeof(float), st);
        return check_launch("sae_plane_scale_dot_act_f32(memset)");
    }
    if (!g || !x || !s || !gx || !gs) return fail(SAE_EINVAL, "sae_plane_scale_dot_act_f32: null tensor");
    if (!aligned16(g) || !aligned16(x) || !aligned16(gx) || (noise && !aligned16(noise)))
        return fail(SAE_EINVAL, "sae_plane_scale_dot_act_f32: tensors must be 16-byte aligned");
    const int64_t need = 4 * outer * channels + 2;
    if (!workspace || workspace_floats < need)
        return fail(SAE_EWORKSPACE, "sae_plane_scale_dot_act_f32: workspace %lld < %lld floats", (long long)workspace_floats,
                    (long long)need);
    double* pb = reinterpret_cast<double*>((reinterpret_cast<uintptr_t>(workspace) + 7) & ~(uintptr_t)7);
    double* pn = pb + outer * channels;
    hipLaunchKernelGGL(plane_scale_dot_act_kernel, dim3((unsigned)(outer * channels)), dim3(kBlock), 0, st, g, x, s, noise, gx, gs,
                       pb, pn, (int)hw, (int)channels, (int)outer, alpha, scale);
    hipLaunchKernelGGL(noise_bias_finalize_kernel, dim3((unsigned)ceil_div64(channels, kBlock / kWave) + 1), dim3(kBlock), 0, st,
                       (const double*)pb, (const double*)pn, gbias, noise ? gnoise_weight : nullptr, (int)channels, (int)outer);
    return check_launch("sae_plane_scale_dot_act_f32");
}
