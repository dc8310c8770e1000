// Small elementwise / reduction glue of the train step that the reference leaves to chains of ATen ops
// (SURVEY.md §8f row 1), each as one launch forward and one launch backward:
//
//   l2_normalize   util.normalize (util/util.py:18-22): v * rsqrt(sum_c v^2 + 1e-8) over dim 1 — the encoder's two
//                  codes (encoder.py:112-113) and the generator's inputs (generator.py:147-148); 5 ATen launches
//                  forward, ~9 backward, per call
//   plane_affine   GeneratorModulation (generator.py:62-67): x * (1 * scale[n,c]) + bias[n,c]
//   softplus_mean  gan_loss (models/networks/loss.py:10-16): F.softplus(+-pred).view(B, -1).mean(dim=1)
//
// The tensors are tiny ([16,8,16,16], [16,2048], [B,1]): these kernels remove launches, not bytes.  One thread walks
// the reduced axis of one output position (consecutive threads = consecutive inner positions: coalesced); rows with
// no inner extent ([N, C] codes) are reduced by one wave per row.  Accumulation in fp32 in a fixed order.
#include "sae_common.h"

namespace sae {
namespace {

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the workgroup, the same value in every thread (fixed order: wave shuffles, then the four wave sums)
__device__ __forceinline__ float block_sum_f(float v) {
    __shared__ float red[kBlock / kWave];
    v = wave_sum_f(v);
    __syncthreads();                // protects `red` against its previous use
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
    return t;
}

// x, y: [outer][C][inner]
__global__ __launch_bounds__(kBlock) void l2_normalize_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                              int64_t outer, int C, int64_t inner, float eps) {
    if (inner == 1) {               // one workgroup per row (one wave walked the 2048 channels of the global code in 19 us)
        const int64_t row = blockIdx.x;
        const float* xr = x + row * C;
        float s = 0.0f;
        for (int c = threadIdx.x; c < C; c += kBlock) s = fmaf(xr[c], xr[c], s);
        s = block_sum_f(s);
        const float r = (1.0f / sqrtf(s + eps));
        for (int c = threadIdx.x; c < C; c += kBlock) y[row * C + c] = xr[c] * r;
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= outer * inner) return;
    const int64_t n = i / inner, j = i - n * inner;
    const float* xp = x + n * C * inner + j;
    float s = 0.0f;
    for (int c = 0; c < C; ++c) s = fmaf(xp[c * inner], xp[c * inner], s);
    const float r = (1.0f / sqrtf(s + eps));
    float* yp = y + n * C * inner + j;
    for (int c = 0; c < C; ++c) yp[c * inner] = xp[c * inner] * r;
}

// gx = r * gy - r^3 * <gy, x> * x,  r = rsqrt(sum_c x^2 + eps)
__global__ __launch_bounds__(kBlock) void l2_normalize_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                  float* __restrict__ gx, int64_t outer, int C,
                                                                  int64_t inner, float eps) {
    if (inner == 1) {
        const int64_t row = blockIdx.x;
        const float* xr = x + row * C;
        const float* gr = gy + row * C;
        float s = 0.0f, d = 0.0f;
        for (int c = threadIdx.x; c < C; c += kBlock) { s = fmaf(xr[c], xr[c], s); d = fmaf(gr[c], xr[c], d); }
        s = block_sum_f(s);
        d = block_sum_f(d);
        const float r = (1.0f / sqrtf(s + eps));
        const float k = r * r * r * d;
        for (int c = threadIdx.x; c < C; c += kBlock) gx[row * C + c] = r * gr[c] - k * xr[c];
        return;
    }
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= outer * inner) return;
    const int64_t n = i / inner, j = i - n * inner;
    const int64_t base = n * C * inner + j;
    float s = 0.0f, d = 0.0f;
    for (int c = 0; c < C; ++c) {
        const float xv = x[base + c * inner];
        s = fmaf(xv, xv, s);
        d = fmaf(gy[base + c * inner], xv, d);
    }
    const float r = (1.0f / sqrtf(s + eps));
    const float k = r * r * r * d;
    for (int c = 0; c < C; ++c) gx[base + c * inner] = r * gy[base + c * inner] - k * x[base + c * inner];
}

// y[p][i] = x[p][i] * a[p] + b[p]      (p = plane (n, c), i < hw)
__global__ __launch_bounds__(kBlock) void plane_affine_kernel(const float* __restrict__ x, const float* __restrict__ a,
                                                              const float* __restrict__ b, float* __restrict__ y,
                                                              int64_t planes, int64_t hw) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * hw) return;
    const int64_t pl = i / hw;
    y[i] = fmaf(x[i], a[pl], b[pl]);
}

// one wave per plane: gx = g * a, ga = sum g * x, gb = sum g
__global__ __launch_bounds__(kBlock) void plane_affine_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                                  const float* __restrict__ a, float* __restrict__ gx,
                                                                  float* __restrict__ ga, float* __restrict__ gb,
                                                                  int64_t planes, int64_t hw) {
    const int64_t pl = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (pl >= planes) return;
    const float av = a[pl];
    float sa = 0.0f, sb = 0.0f;
    for (int64_t i = lane; i < hw; i += kWave) {
        const float gv = g[pl * hw + i];
        gx[pl * hw + i] = gv * av;
        sa = fmaf(gv, x[pl * hw + i], sa);
        sb += gv;
    }
    sa = wave_sum_f(sa);
    sb = wave_sum_f(sb);
    if (lane == 0) { ga[pl] = sa; gb[pl] = sb; }
}

// F.softplus (beta 1, threshold 20): x > 20 ? x : log1p(exp(x))
__device__ __forceinline__ float softplus_f(float v) { return v > 20.0f ? v : log1pf(expf(v)); }

// y[b] = mean_i softplus(sign * x[b][i]); one wave per sample
__global__ __launch_bounds__(kBlock) void softplus_mean_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                               int64_t batch, int64_t inner, float sign) {
    const int64_t b = (int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (b >= batch) return;
    float s = 0.0f;
    for (int64_t i = lane; i < inner; i += kWave) s += softplus_f(sign * x[b * inner + i]);
    s = wave_sum_f(s);
    if (lane == 0) y[b] = s / (float)inner;
}

// gx[b][i] = gy[b] * sign * sigmoid(sign * x[b][i]) / inner   (x > 20: derivative 1, as ATen's softplus_backward)
__global__ __launch_bounds__(kBlock) void softplus_mean_bwd_kernel(const float* __restrict__ gy, const float* __restrict__ x,
                                                                   float* __restrict__ gx, int64_t batch, int64_t inner,
                                                                   float sign) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= batch * inner) return;
    const float z = sign * x[i];
    const float e = expf(z);
    const float dz = z > 20.0f ? 1.0f : e / (e + 1.0f);
    gx[i] = gy[i / inner] * sign * dz / (float)inner;
}

inline unsigned blocks_for(int64_t work, int per_block) {
    int64_t b = ceil_div64(work > 0 ? work : 1, per_block);
    return (unsigned)(b > 2147483647 ? 2147483647 : b);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_l2_normalize_f32(const float* x, float* y, int64_t outer, int64_t channels, int64_t inner, float eps,
                                    sae_stream_t stream) {
    sae::clear_stale_error();
    if (outer < 0 || outer >= ((int64_t)1 << 31) || channels < 1 || inner < 1 || channels >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_l2_normalize_f32: bad shape");
    if (outer == 0) return SAE_OK;
    if (!x || !y) return fail(SAE_EINVAL, "sae_l2_normalize_f32: null tensor");
    const unsigned blocks = inner == 1 ? (unsigned)outer : blocks_for(outer * inner, kBlock);
    hipLaunchKernelGGL(l2_normalize_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, x, y, outer, (int)channels,
                       inner, eps);
    return check_launch("sae_l2_normalize_f32");
}

extern "C" int sae_l2_normalize_bwd_f32(const float* gy, const float* x, float* gx, int64_t outer, int64_t channels,
                                        int64_t inner, float eps, sae_stream_t stream) {
    sae::clear_stale_error();
    if (outer < 0 || outer >= ((int64_t)1 << 31) || channels < 1 || inner < 1 || channels >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_l2_normalize_bwd_f32: bad shape");
    if (outer == 0) return SAE_OK;
    if (!gy || !x || !gx) return fail(SAE_EINVAL, "sae_l2_normalize_bwd_f32: null tensor");
    const unsigned blocks = inner == 1 ? (unsigned)outer : blocks_for(outer * inner, kBlock);
    hipLaunchKernelGGL(l2_normalize_bwd_kernel, dim3(blocks), dim3(kBlock), 0, (hipStream_t)stream, gy, x, gx, outer,
                       (int)channels, inner, eps);
    return check_launch("sae_l2_normalize_bwd_f32");
}

extern "C" int sae_plane_affine_f32(const float* x, const float* a, const float* b, float* y, int64_t planes, int64_t hw,
                                    sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || hw < 1) return fail(SAE_EINVAL, "sae_plane_affine_f32: bad shape");
    if (planes == 0) return SAE_OK;
    if (!x || !a || !b || !y) return fail(SAE_EINVAL, "sae_plane_affine_f32: null tensor");
    hipLaunchKernelGGL(plane_affine_kernel, dim3(blocks_for(planes * hw, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, x, a,
                       b, y, planes, hw);
    return check_launch("sae_plane_affine_f32");
}

extern "C" int sae_plane_affine_bwd_f32(const float* g, const float* x, const float* a, float* gx, float* ga, float* gb,
                                        int64_t planes, int64_t hw, sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || hw < 1) return fail(SAE_EINVAL, "sae_plane_affine_bwd_f32: bad shape");
    if (planes == 0) return SAE_OK;
    if (!g || !x || !a || !gx || !ga || !gb) return fail(SAE_EINVAL, "sae_plane_affine_bwd_f32: null tensor");
    hipLaunchKernelGGL(plane_affine_bwd_kernel, dim3(blocks_for(planes, kBlock / kWave)), dim3(kBlock), 0,
                       (hipStream_t)stream, g, x, a, gx, ga, gb, planes, hw);
    return check_launch("sae_plane_affine_bwd_f32");
}

extern "C" int sae_softplus_mean_f32(const float* x, float* y, int64_t batch, int64_t inner, float sign,
                                     sae_stream_t stream) {
    sae::clear_stale_error();
    if (batch < 0 || inner < 1) return fail(SAE_EINVAL, "sae_softplus_mean_f32: bad shape");
    if (batch == 0) return SAE_OK;
    if (!x || !y) return fail(SAE_EINVAL, "sae_softplus_mean_f32: null tensor");
    hipLaunchKernelGGL(softplus_mean_kernel, dim3(blocks_for(batch, kBlock / kWave)), dim3(kBlock), 0, (hipStream_t)stream,
                       x, y, batch, inner, sign);
    return check_launch("sae_softplus_mean_f32");
}

extern "C" int sae_softplus_mean_bwd_f32(const float* gy, const float* x, float* gx, int64_t batch, int64_t inner,
                                         float sign, sae_stream_t stream) {
    sae::clear_stale_error();
    if (batch < 0 || inner < 1) return fail(SAE_EINVAL, "sae_softplus_mean_bwd_f32: bad shape");
    if (batch == 0) return SAE_OK;
    if (!gy || !x || !gx) return fail(SAE_EINVAL, "sae_softplus_mean_bwd_f32: null tensor");
    hipLaunchKernelGGL(softplus_mean_bwd_kernel, dim3(blocks_for(batch * inner, kBlock)), dim3(kBlock), 0,
                       (hipStream_t)stream, gy, x, gx, batch, inner, sign);
    return check_launch("sae_softplus_mean_bwd_f32");
}
