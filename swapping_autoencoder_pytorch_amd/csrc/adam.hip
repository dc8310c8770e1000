// Multi-tensor Adam update.
//
// Replaces the two torch.optim.Adam instances of the reference's driver
// (optimizers/swapping_autoencoder_optimizer.py:34-42, stepped at :77,:95,:107): one launch updates a whole
// batch of parameter tensors instead of four elementwise ATen launches per tensor (226 parameters per group at
// the church preset).  HBM-bound: 16 B read (p, g, m, v) + 12 B written (p, m, v) per element, 16 B per lane
// accesses on the aligned body of every tensor.
//
// Tensor pointers travel in the kernel-argument block (no device-side table to upload or keep alive): up to
// kTensors tensors and kBlocks 64 Ki-element chunks per launch; the host loop packs as many launches as the list
// needs.  Per-tensor scalars (step size lr / (1 - beta1^t), 1 / sqrt(1 - beta2^t)) are computed on the host in
// double, as torch.optim.Adam does (torch/optim/adam.py, _single_tensor_adam), and the element update follows
// its operation order (hyper-parameters arrive as doubles so that 1 - beta is rounded to fp32 once, as in torch):
//     g  = grad * grad_scale                      (grad_scale: the 1 / world_size of the gradient all-reduce)
//     m  = lerp(m, g, 1 - beta1)                  (ATen's two-sided lerp formula)
//     v  = v * beta2 + ((1 - beta2) * g) * g      (mul, addcmul)
//     p -= step_size * m / (sqrt(v) / sqrt(bias_correction2) + eps)
#include "sae_common.h"

#include <cmath>

namespace sae {
namespace {

constexpr int kTensors = 24;
constexpr int kBlocks = 224;
constexpr int kChunk = 65536;      // elements per workgroup

struct AdamArgs {
    float* p[kTensors];
    const float* g[kTensors];
    float* m[kTensors];
    float* v[kTensors];
    long long n[kTensors];
    float step_size[kTensors];
    float inv_bc2_sqrt[kTensors];
    const long long* step_dev[kTensors];    // DEV: the tensor's update count BEFORE this update, in device memory
    int block_chunk[kBlocks];
    unsigned char block_tensor[kBlocks];
};

__device__ __forceinline__ void adam_element(float& p, float g, float& m, float& v, float gscale, float omb1, float beta2,
                                             float omb2, float eps, float step_size, float inv_bc2_sqrt) {
    g *= gscale;
    // ATen's lerp: the form that is exact at the near end (weight = 1 - beta1 = 1, the reference's beta1 = 0, gives m = g)
    m = (omb1 < 0.5f) ? m + omb1 * (g - m) : g - (g - m) * (1.0f - omb1);
    v = v * beta2 + (omb2 * g) * g;
    const float denom = sqrtf(v) * inv_bc2_sqrt + eps;
    p = p - step_size * (m / denom);
}

// DEV (sae_adam_multi_dev_f32): the step counts live in device memory, so that a captured hipGraph of the train step replays with
// the right bias corrections -- as kernel ARGUMENTS they would be frozen at their capture-time values.  Every thread forms the
// two scalars in double, as the host form does (two pow() per workgroup of 64 Ki elements: nothing next to the 28 bytes per
// element the update moves).
template <bool DEV>
__global__ __launch_bounds__(kBlock) void adam_multi_kernel(AdamArgs a, float gscale, float omb1, float beta2, float omb2,
                                                            float eps, double lr, double beta1d, double beta2d) {
    const int t = a.block_tensor[blockIdx.x];
    const long long base = (long long)a.block_chunk[blockIdx.x] * kChunk;
    const long long n = a.n[t];
    const long long count = (n - base < kChunk) ? (n - base) : kChunk;
    float* __restrict__ p = a.p[t] + base;
    const float* __restrict__ g = a.g[t] + base;
    float* __restrict__ m = a.m[t] + base;
    float* __restrict__ v = a.v[t] + base;
    float ss = a.step_size[t], ib = a.inv_bc2_sqrt[t];
    if constexpr (DEV) {
        const double st = (double)(*a.step_dev[t] + 1);
        ss = (float)(lr / (1.0 - pow(beta1d, st)));
        ib = (float)(1.0 / sqrt(1.0 - pow(beta2d, st)));
    }
    const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                       reinterpret_cast<uintptr_t>(v)) & 15u) == 0;
    long long done = 0;
    if (vec) {
        const int nvec = (int)(count >> 2);
        // two independent 16-byte vectors per thread per iteration: 8 loads in flight per lane
        for (int i = threadIdx.x; i < nvec; i += 2 * kBlock) {
            const int j = i + kBlock;
            const bool two = j < nvec;
            float4 p0 = reinterpret_cast<float4*>(p)[i], g0 = reinterpret_cast<const float4*>(g)[i];
            float4 m0 = reinterpret_cast<float4*>(m)[i], v0 = reinterpret_cast<float4*>(v)[i];
            float4 p1, g1, m1, v1;
            if (two) {
                p1 = reinterpret_cast<float4*>(p)[j]; g1 = reinterpret_cast<const float4*>(g)[j];
                m1 = reinterpret_cast<float4*>(m)[j]; v1 = reinterpret_cast<float4*>(v)[j];
            }
            adam_element(p0.x, g0.x, m0.x, v0.x, gscale, omb1, beta2, omb2, eps, ss, ib);
            adam_element(p0.y, g0.y, m0.y, v0.y, gscale, omb1, beta2, omb2, eps, ss, ib);
            adam_element(p0.z, g0.z, m0.z, v0.z, gscale, omb1, beta2, omb2, eps, ss, ib);
            adam_element(p0.w, g0.w, m0.w, v0.w, gscale, omb1, beta2, omb2, eps, ss, ib);
            reinterpret_cast<float4*>(p)[i] = p0; reinterpret_cast<float4*>(m)[i] = m0; reinterpret_cast<float4*>(v)[i] = v0;
            if (two) {
                adam_element(p1.x, g1.x, m1.x, v1.x, gscale, omb1, beta2, omb2, eps, ss, ib);
                adam_element(p1.y, g1.y, m1.y, v1.y, gscale, omb1, beta2, omb2, eps, ss, ib);
                adam_element(p1.z, g1.z, m1.z, v1.z, gscale, omb1, beta2, omb2, eps, ss, ib);
                adam_element(p1.w, g1.w, m1.w, v1.w, gscale, omb1, beta2, omb2, eps, ss, ib);
                reinterpret_cast<float4*>(p)[j] = p1; reinterpret_cast<float4*>(m)[j] = m1; reinterpret_cast<float4*>(v)[j] = v1;
            }
        }
        done = (long long)nvec << 2;
    }
    for (long long i = done + threadIdx.x; i < count; i += kBlock) {
        float pv = p[i], mv = m[i], vv = v[i];
        adam_element(pv, g[i], mv, vv, gscale, omb1, beta2, omb2, eps, ss, ib);
        p[i] = pv; m[i] = mv; v[i] = vv;
    }
}

constexpr int kAdvance = 384;       // step counters advanced per launch (pointers in the argument block)
struct AdvanceArgs { long long* slot[kAdvance]; };
__global__ void adam_advance_kernel(AdvanceArgs a, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) *a.slot[i] += 1;
}

}  // namespace
}  // namespace sae

namespace {
int adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
               const int64_t* numel, const int64_t* step, int64_t* const* step_dev, int64_t count, double lr, double beta1,
               double beta2, double eps, double grad_scale, sae_stream_t stream);
}

extern "C" int sae_adam_multi_dev_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                      float* const* exp_avg_sq, const int64_t* numel, int64_t* const* step_dev, int64_t count,
                                      double lr, double beta1, double beta2, double eps, double grad_scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (count > 0 && !step_dev) return sae::fail(SAE_EINVAL, "sae_adam_multi_dev_f32: null step table");
    for (int64_t i = 0; i < count; ++i)
        if (!step_dev[i]) return sae::fail(SAE_EINVAL, "sae_adam_multi_dev_f32: tensor %lld: null step counter", (long long)i);
    return adam_multi(params, grads, exp_avg, exp_avg_sq, numel, nullptr, step_dev, count, lr, beta1, beta2, eps, grad_scale, stream);
}

extern "C" int sae_adam_multi_f32(float* const* params, const float* const* grads, float* const* exp_avg,
                                  float* const* exp_avg_sq, const int64_t* numel, const int64_t* step, int64_t count,
                                  double lr, double beta1, double beta2, double eps, double grad_scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (count > 0 && !step) return sae::fail(SAE_EINVAL, "sae_adam_multi_f32: null table");
    return adam_multi(params, grads, exp_avg, exp_avg_sq, numel, step, nullptr, count, lr, beta1, beta2, eps, grad_scale, stream);
}

namespace {
int adam_multi(float* const* params, const float* const* grads, float* const* exp_avg, float* const* exp_avg_sq,
               const int64_t* numel, const int64_t* step, int64_t* const* step_dev, int64_t count, double lr, double beta1,
               double beta2, double eps, double grad_scale, sae_stream_t stream) {
    using namespace sae;
    const bool dev = step_dev != nullptr;
    if (count < 0 || (count > 0 && (!params || !grads || !exp_avg || !exp_avg_sq || !numel)))
        return fail(SAE_EINVAL, "sae_adam_multi_f32: null table");
    if (!(beta1 >= 0.0 && beta1 < 1.0 && beta2 >= 0.0 && beta2 < 1.0 && eps >= 0.0))
        return fail(SAE_EINVAL, "sae_adam_multi_f32: betas must be in [0, 1), eps >= 0");
    for (int64_t i = 0; i < count; ++i) {
        if (numel[i] < 0 || (!dev && step[i] < 1))
            return fail(SAE_EINVAL, "sae_adam_multi_f32: tensor %lld: numel < 0 or step < 1", (long long)i);
        if (numel[i] > 0 && (!params[i] || !grads[i] || !exp_avg[i] || !exp_avg_sq[i]))
            return fail(SAE_EINVAL, "sae_adam_multi_f32: tensor %lld: null pointer", (long long)i);
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    AdamArgs a;
    int nt = 0, nb = 0;
    auto flush = [&]() -> int {
        if (nb == 0) { nt = 0; return SAE_OK; }
        if (dev)
            hipLaunchKernelGGL(adam_multi_kernel<true>, dim3(nb), dim3(kBlock), 0, s, a, (float)grad_scale, (float)(1.0 - beta1),
                               (float)beta2, (float)(1.0 - beta2), (float)eps, lr, beta1, beta2);
        else
            hipLaunchKernelGGL(adam_multi_kernel<false>, dim3(nb), dim3(kBlock), 0, s, a, (float)grad_scale, (float)(1.0 - beta1),
                               (float)beta2, (float)(1.0 - beta2), (float)eps, lr, beta1, beta2);
        nt = nb = 0;
        return check_launch("adam_multi_kernel");
    };
    for (int64_t i = 0; i < count; ++i) {
        if (numel[i] == 0) continue;
        float step_size = 0.0f, inv_bc2_sqrt = 0.0f;
        if (!dev) {
            const double bc1 = 1.0 - std::pow(beta1, (double)step[i]);
            const double bc2 = 1.0 - std::pow(beta2, (double)step[i]);
            step_size = (float)(lr / bc1);
            inv_bc2_sqrt = (float)(1.0 / std::sqrt(bc2));
        }
        const int64_t chunks = ceil_div64(numel[i], kChunk);
        int64_t c = 0;
        while (c < chunks) {
            if (nt == kTensors || nb == kBlocks) {
                int rc = flush();
                if (rc != SAE_OK) return rc;
            }
            // (re)open tensor i in the current argument block, offset to chunk c
            const int t = nt++;
            a.p[t] = params[i] + c * kChunk;
            a.g[t] = grads[i] + c * kChunk;
            a.m[t] = exp_avg[i] + c * kChunk;
            a.v[t] = exp_avg_sq[i] + c * kChunk;
            a.n[t] = numel[i] - c * kChunk;
            a.step_size[t] = step_size;
            a.inv_bc2_sqrt[t] = inv_bc2_sqrt;
            a.step_dev[t] = dev ? reinterpret_cast<const long long*>(step_dev[i]) : nullptr;
            int local = 0;
            while (c < chunks && nb < kBlocks) {
                a.block_tensor[nb] = (unsigned char)t;
                a.block_chunk[nb] = local++;
                ++nb;
                ++c;
            }
        }
    }
    int rc = flush();
    if (rc != SAE_OK || !dev) return rc;
    // every update above read the counts as they were; now they advance (tensors of zero elements included: torch counts the
    // step of every parameter that had a gradient)
    for (int64_t i0 = 0; i0 < count; i0 += kAdvance) {
        AdvanceArgs adv;
        const int nadv = (int)((count - i0 < kAdvance) ? (count - i0) : kAdvance);
        for (int j = 0; j < nadv; ++j) adv.slot[j] = reinterpret_cast<long long*>(step_dev[i0 + j]);
        hipLaunchKernelGGL(adam_advance_kernel, dim3(ceil_div(nadv, 128)), dim3(128), 0, s, adv, nadv);
        rc = check_launch("adam_advance_kernel");
        if (rc != SAE_OK) return rc;
    }
    return SAE_OK;
}
}  // namespace
