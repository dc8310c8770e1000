// K2 — fused bias + activation (+ its fused backward with the bias-gradient reduction).
//
// Replaces fused.fused_bias_act (reference: models/networks/stylegan2_op/fused_bias_act.cpp:11-17,
// fused_bias_act_kernel.cu:18-99) and the separate grad_input.sum(dim) pass of
// fused_act.py:32-41.  HBM-bound elementwise work: the forward moves 8 B per element
// (read x, write y), the fused backward 12 B (read gy, read y, write gx) with the per-channel
// bias gradient reduced in the same pass (wave shuffle -> LDS -> one partial per block, then a
// fixed-order second stage: deterministic, no atomics).
//
// gfx950 notes: 16 B per lane accesses wherever the channel plane is a multiple of 4 floats
// and the pointers are 16 B aligned (1 KiB per wave instruction); 4 independent vectors in
// flight per thread; 32-bit index arithmetic unless the tensor has >= 2^32 elements.
#include "sae_common.h"

namespace sae {
namespace {

constexpr int kUnroll = 4;

__device__ __forceinline__ float act_apply(float v, float r, int mode, float alpha) {
    // mode = act * 10 + grad, fused_bias_act_kernel.cu:36-45
    switch (mode) {
        case 30: return (v > 0.0f) ? v : v * alpha;
        case 31: return (r > 0.0f) ? v : v * alpha;
        case 12:
        case 32: return 0.0f;
        default: return v;  // 10, 11
    }
}

template <typename IdxT, int VEC>
__global__ __launch_bounds__(kBlock) void bias_act_kernel(const float* __restrict__ x,
                                                          const float* __restrict__ b,
                                                          const float* __restrict__ ref,
                                                          float* __restrict__ y, IdxT nvec, IdxT step_b,
                                                          IdxT size_b, int mode, float alpha,
                                                          float scale) {
    const IdxT stride = (IdxT)gridDim.x * kBlock;
    for (IdxT v0 = (IdxT)blockIdx.x * kBlock + threadIdx.x; v0 < nvec; v0 += stride * kUnroll) {
        if constexpr (VEC == 4) {
            float4 xv[kUnroll], rv[kUnroll];
            float bv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const IdxT v = v0 + stride * u;
                if (v < nvec) {
                    xv[u] = reinterpret_cast<const float4*>(x)[v];
                    rv[u] = ref ? reinterpret_cast<const float4*>(ref)[v] : make_float4(0.f, 0.f, 0.f, 0.f);
                    bv[u] = b ? b[((v * 4) / step_b) % size_b] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const IdxT v = v0 + stride * u;
                if (v < nvec) {
                    float4 o;
                    o.x = act_apply(xv[u].x + bv[u], rv[u].x, mode, alpha) * scale;
                    o.y = act_apply(xv[u].y + bv[u], rv[u].y, mode, alpha) * scale;
                    o.z = act_apply(xv[u].z + bv[u], rv[u].z, mode, alpha) * scale;
                    o.w = act_apply(xv[u].w + bv[u], rv[u].w, mode, alpha) * scale;
                    reinterpret_cast<float4*>(y)[v] = o;
                }
            }
        } else {
            float xv[kUnroll], rv[kUnroll], bv[kUnroll];
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const IdxT v = v0 + stride * u;
                if (v < nvec) {
                    xv[u] = x[v];
                    rv[u] = ref ? ref[v] : 0.0f;
                    bv[u] = b ? b[(v / step_b) % size_b] : 0.0f;
                }
            }
#pragma unroll
            for (int u = 0; u < kUnroll; ++u) {
                const IdxT v = v0 + stride * u;
                if (v < nvec) y[v] = act_apply(xv[u] + bv[u], rv[u], mode, alpha) * scale;
            }
        }
    }
}

// ---- fused backward ---------------------------------------------------------------------------

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the 256 threads of the block, result valid in thread 0
__device__ __forceinline__ float block_sum(float v) {
    __shared__ float red[kBlock / kWave];
    v = wave_sum(v);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
    __syncthreads();  // protect `red` against a previous use
    if (lane == 0) red[wid] = v;
    __syncthreads();
    float t = 0.0f;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
    }
    return t;
}

// Large channel planes: tensor viewed as [outer][C][step_b]; block (c, s) walks the chunks
// (n, chunk-of-plane) s, s+S, ... of channel c and leaves one partial in partial[c*S + s].
template <int VEC>
__global__ __launch_bounds__(kBlock) void bias_act_bwd_plane_kernel(
    const float* __restrict__ gy, const float* __restrict__ yref, float* __restrict__ gx,
    float* __restrict__ partial, int64_t outer, int64_t step_b, int64_t size_b, int chunks_per_plane,
    int nsplit, float alpha, float scale) {
    constexpr int kChunk = kBlock * VEC * 2;  // elements per block iteration
    const int64_t c = blockIdx.x;
    const int s = blockIdx.y;
    const int64_t items = outer * chunks_per_plane;
    float acc = 0.0f;
    for (int64_t it = s; it < items; it += nsplit) {
        const int64_t n = it / chunks_per_plane;
        const int ch = (int)(it - n * chunks_per_plane);
        const int64_t plane = (n * size_b + c) * step_b;
        const int64_t e0 = (int64_t)ch * kChunk;
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int64_t e = e0 + ((int64_t)u * kBlock + threadIdx.x) * VEC;
            if (e < step_b) {
                if constexpr (VEC == 4) {
                    const float4 g = *reinterpret_cast<const float4*>(gy + plane + e);
                    const float4 r = *reinterpret_cast<const float4*>(yref + plane + e);
                    float4 o;
                    o.x = ((r.x > 0.0f) ? g.x : g.x * alpha) * scale;
                    o.y = ((r.y > 0.0f) ? g.y : g.y * alpha) * scale;
                    o.z = ((r.z > 0.0f) ? g.z : g.z * alpha) * scale;
                    o.w = ((r.w > 0.0f) ? g.w : g.w * alpha) * scale;
                    *reinterpret_cast<float4*>(gx + plane + e) = o;
                    acc += (o.x + o.y) + (o.z + o.w);
                } else {
                    const float g = gy[plane + e];
                    const float r = yref[plane + e];
                    const float o = ((r > 0.0f) ? g : g * alpha) * scale;
                    gx[plane + e] = o;
                    acc += o;
                }
            }
        }
    }
    const float t = block_sum(acc);
    if (threadIdx.x == 0) partial[c * nsplit + s] = t;
}

// Small channel planes (step_b < 256, including the [N, C] case step_b == 1): one thread per (c, hw) column j walks a
// slice of the outer dimension; consecutive threads touch consecutive addresses.  The outer dimension is cut into
// gridDim.y slices so that a few thousand workgroups are in flight (one slice left 24 workgroups on 256 CUs for the
// 384 x [384, 4, 4] maps of the patch discriminator: 0.2 TB/s), four rows per step so that eight loads are in flight
// per thread.   partial[(c * S + s) * step_b + hw] = sum over the slice's rows of gx[n][c][hw]
__global__ __launch_bounds__(kBlock) void bias_act_bwd_column_kernel(
    const float* __restrict__ gy, const float* __restrict__ yref, float* __restrict__ gx,
    float* __restrict__ partial, int64_t outer, int64_t cols, int64_t step_b, int rows_per_slice, float alpha, float scale) {
    const int64_t j = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (j >= cols) return;
    const int64_t n0 = (int64_t)blockIdx.y * rows_per_slice;
    int64_t n1 = n0 + rows_per_slice;
    if (n1 > outer) n1 = outer;
    float acc = 0.0f;
    int64_t n = n0;
    for (; n + 4 <= n1; n += 4) {
        float g[4], r[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            g[u] = gy[(n + u) * cols + j];
            r[u] = yref[(n + u) * cols + j];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const float o = ((r[u] > 0.0f) ? g[u] : g[u] * alpha) * scale;
            gx[(n + u) * cols + j] = o;
            acc += o;
        }
    }
    for (; n < n1; ++n) {
        const int64_t i = n * cols + j;
        const float g = gy[i];
        const float r = yref[i];
        const float o = ((r > 0.0f) ? g : g * alpha) * scale;
        gx[i] = o;
        acc += o;
    }
    const int64_t c = j / step_b, hw = j - c * step_b;
    partial[(c * gridDim.y + blockIdx.y) * step_b + hw] = acc;
}

// Second stage: gb[c] = sum_{q < Q} partial[c*Q + q], one workgroup per channel, fixed order: thread t sums q = t, t + 256,
// ... in four independent chains (one chain of Q / 64 dependent loads per wave took 50 us at the Q of a few thousand that
// the fused K1 epilogue produces), then wave shuffles and one LDS round.
__global__ __launch_bounds__(kBlock) void bias_grad_finalize_kernel(const float* __restrict__ partial,
                                                                     float* __restrict__ gb,
                                                                     int64_t size_b, int q_count) {
    const int64_t c = blockIdx.x;
    const float* pc = partial + c * q_count;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int q = threadIdx.x;
    for (; q + 3 * kBlock < q_count; q += 4 * kBlock) {
        a0 += pc[q];
        a1 += pc[q + kBlock];
        a2 += pc[q + 2 * kBlock];
        a3 += pc[q + 3 * kBlock];
    }
    for (; q < q_count; q += kBlock) a0 += pc[q];
    const float t = block_sum((a0 + a1) + (a2 + a3));
    if (threadIdx.x == 0) gb[c] = t;
}

struct BwdPlan {
    bool column;        // small-plane kernel
    int vec;            // 4 or 1 (plane kernel)
    int chunks_per_plane;
    int nsplit;         // plane kernel: workgroups per channel; column kernel: slices of the outer dimension
    int rows_per_slice; // column kernel
    int q_count;        // partials per channel
    int64_t outer;
};

BwdPlan plan_bwd(int64_t numel, int64_t step_b, int64_t size_b, bool ptr_aligned) {
    BwdPlan p{};
    p.outer = numel / (step_b * size_b);
    if (step_b < 256) {
        p.column = true;
        // ~2048 workgroups, at least 8 rows per slice
        const int64_t col_blocks = ceil_div64(step_b * size_b, kBlock);
        int64_t slices = ceil_div64(2048, col_blocks);
        if (slices > p.outer / 8) slices = p.outer / 8;
        if (slices < 1) slices = 1;
        if (slices > 512) slices = 512;
        p.rows_per_slice = (int)ceil_div64(p.outer, slices);
        p.nsplit = (int)ceil_div64(p.outer, p.rows_per_slice);
        p.q_count = (int)step_b * p.nsplit;
        return p;
    }
    p.column = false;
    p.vec = (step_b % 4 == 0 && ptr_aligned) ? 4 : 1;
    const int chunk = kBlock * p.vec * 2;
    p.chunks_per_plane = (int)ceil_div64(step_b, chunk);
    const int64_t items = p.outer * p.chunks_per_plane;
    int64_t want = ceil_div64(2048, size_b);
    if (want < 1) want = 1;
    if (want > items) want = items;
    if (want > 1024) want = 1024;
    p.nsplit = (int)want;
    p.q_count = p.nsplit;
    return p;
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_bias_act_f32(const float* x, const float* b, const float* ref, float* y,
                                int64_t numel, int64_t step_b, int64_t size_b, int32_t act,
                                int32_t grad, float alpha, float scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (numel < 0 || (act != 1 && act != 3) || grad < 0 || grad > 2)
        return fail(SAE_EINVAL, "sae_bias_act_f32: unsupported act=%d grad=%d or numel=%lld", act, grad,
                    (long long)numel);
    if (numel == 0) return SAE_OK;
    if (!x || !y) return fail(SAE_EINVAL, "sae_bias_act_f32: null tensor");
    if (b && (step_b < 1 || size_b < 1)) return fail(SAE_EINVAL, "sae_bias_act_f32: bad step_b/size_b");
    if (grad == 1 && !ref) return fail(SAE_EINVAL, "sae_bias_act_f32: grad=1 needs ref");
    if (!b) { step_b = 1; size_b = 1; }
    const int mode = act * 10 + grad;
    const bool vec4 = (numel % 4 == 0) && (step_b % 4 == 0) && aligned16(x) && aligned16(y) &&
                      (!ref || aligned16(ref));
    const int64_t nvec = vec4 ? numel / 4 : numel;
    int64_t blocks = ceil_div64(nvec, (int64_t)kBlock * kUnroll);
    if (blocks > 16384) blocks = 16384;
    hipStream_t s = (hipStream_t)stream;
    const bool wide = numel >= ((int64_t)1 << 32);
    if (vec4) {
        if (wide)
            hipLaunchKernelGGL((bias_act_kernel<uint64_t, 4>), dim3((unsigned)blocks), dim3(kBlock), 0, s, x, b,
                               ref, y, (uint64_t)nvec, (uint64_t)step_b, (uint64_t)size_b, mode, alpha, scale);
        else
            hipLaunchKernelGGL((bias_act_kernel<uint32_t, 4>), dim3((unsigned)blocks), dim3(kBlock), 0, s, x, b,
                               ref, y, (uint32_t)nvec, (uint32_t)step_b, (uint32_t)size_b, mode, alpha, scale);
    } else {
        if (wide)
            hipLaunchKernelGGL((bias_act_kernel<uint64_t, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, x, b,
                               ref, y, (uint64_t)nvec, (uint64_t)step_b, (uint64_t)size_b, mode, alpha, scale);
        else
            hipLaunchKernelGGL((bias_act_kernel<uint32_t, 1>), dim3((unsigned)blocks), dim3(kBlock), 0, s, x, b,
                               ref, y, (uint32_t)nvec, (uint32_t)step_b, (uint32_t)size_b, mode, alpha, scale);
    }
    return check_launch("sae_bias_act_f32");
}

extern "C" int64_t sae_bias_act_bwd_workspace(int64_t numel, int64_t step_b, int64_t size_b) {
    if (numel <= 0 || step_b < 1 || size_b < 1 || numel % (step_b * size_b) != 0) return 0;
    // alignment only changes the vector width, not the partial count
    const BwdPlan p = plan_bwd(numel, step_b, size_b, true);
    return size_b * (int64_t)p.q_count;
}

extern "C" int sae_bias_act_bwd_f32(const float* gy, const float* y_ref, float* gx, float* gb,
                                    float* workspace, int64_t workspace_floats, int64_t numel,
                                    int64_t step_b, int64_t size_b, float alpha, float scale,
                                    sae_stream_t stream) {
    sae::clear_stale_error();
    if (step_b < 1 || size_b < 1 || numel < 0 || numel % (step_b * size_b) != 0)
        return fail(SAE_EINVAL, "sae_bias_act_bwd_f32: numel=%lld is not outer*size_b*step_b (%lld,%lld)",
                    (long long)numel, (long long)size_b, (long long)step_b);
    if (!gb) return fail(SAE_EINVAL, "sae_bias_act_bwd_f32: null gb");
    hipStream_t s = (hipStream_t)stream;
    if (numel == 0) {
        hipMemsetAsync(gb, 0, sizeof(float) * (size_t)size_b, s);
        return check_launch("sae_bias_act_bwd_f32(memset)");
    }
    if (!gy || !y_ref || !gx) return fail(SAE_EINVAL, "sae_bias_act_bwd_f32: null tensor");
    const BwdPlan p = plan_bwd(numel, step_b, size_b, aligned16(gy) && aligned16(y_ref) && aligned16(gx));
    const int64_t need = size_b * (int64_t)p.q_count;
    if (!workspace || workspace_floats < need)
        return fail(SAE_EWORKSPACE, "sae_bias_act_bwd_f32: workspace %lld < %lld floats",
                    (long long)workspace_floats, (long long)need);
    if (p.column) {
        const int64_t cols = size_b * step_b;
        hipLaunchKernelGGL(bias_act_bwd_column_kernel, dim3((unsigned)ceil_div64(cols, kBlock), (unsigned)p.nsplit),
                           dim3(kBlock), 0, s, gy, y_ref, gx, workspace, p.outer, cols, step_b, p.rows_per_slice, alpha, scale);
    } else if (p.vec == 4) {
        hipLaunchKernelGGL((bias_act_bwd_plane_kernel<4>), dim3((unsigned)size_b, (unsigned)p.nsplit),
                           dim3(kBlock), 0, s, gy, y_ref, gx, workspace, p.outer, step_b, size_b,
                           p.chunks_per_plane, p.nsplit, alpha, scale);
    } else {
        hipLaunchKernelGGL((bias_act_bwd_plane_kernel<1>), dim3((unsigned)size_b, (unsigned)p.nsplit),
                           dim3(kBlock), 0, s, gy, y_ref, gx, workspace, p.outer, step_b, size_b,
                           p.chunks_per_plane, p.nsplit, alpha, scale);
    }
    hipLaunchKernelGGL(bias_grad_finalize_kernel, dim3((unsigned)size_b), dim3(kBlock), 0, s, (const float*)workspace, gb,
                       size_b, p.q_count);
    return check_launch("sae_bias_act_bwd_f32");
}


// ---- residual merge: y = alpha * (a + b)  (ResBlock / generator blocks: (out + skip) / sqrt(2),
// stylegan2_layers.py:689, generator.py:36) in one pass instead of an add and a divide ----------
namespace sae {
namespace {
template <int VEC>
__global__ __launch_bounds__(kBlock) void add_scale_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                           float* __restrict__ y, int64_t nvec, float alpha) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t v = (int64_t)blockIdx.x * kBlock + threadIdx.x; v < nvec; v += stride) {
        if constexpr (VEC == 4) {
            const f32x4 x = reinterpret_cast<const f32x4*>(a)[v];
            const f32x4 z = reinterpret_cast<const f32x4*>(b)[v];
            reinterpret_cast<f32x4*>(y)[v] = (x + z) * alpha;
        } else {
            y[v] = (a[v] + b[v]) * alpha;
        }
    }
}
}  // namespace
}  // namespace sae

extern "C" int sae_add_scale_f32(const float* a, const float* b, float* y, int64_t numel, float alpha,
                                 sae_stream_t stream) {
    sae::clear_stale_error();
    if (numel < 0) return fail(SAE_EINVAL, "sae_add_scale_f32: negative size");
    if (numel == 0) return SAE_OK;
    if (!a || !b || !y) return fail(SAE_EINVAL, "sae_add_scale_f32: null tensor");
    const bool vec4 = numel % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(y);
    const int64_t nvec = vec4 ? numel / 4 : numel;
    int64_t blocks = ceil_div64(nvec, kBlock);
    if (blocks > 16384) blocks = 16384;
    if (vec4)
        hipLaunchKernelGGL((add_scale_kernel<4>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, b, y,
                           nvec, alpha);
    else
        hipLaunchKernelGGL((add_scale_kernel<1>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, a, b, y,
                           nvec, alpha);
    return check_launch("sae_add_scale_f32");
}
