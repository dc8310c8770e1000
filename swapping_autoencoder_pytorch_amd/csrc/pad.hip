// Reflection padding of the encoder's ConvLayers / Blurs (reference: nn.ReflectionPad2d and F.pad(mode="reflect"),
// models/networks/stylegan2_layers.py:57-63, :100-105, :643) and its adjoint.
//   forward   y[p][oy][ox] = x[p][refl(oy - top, H)][refl(ox - left, W)],  refl(i, n) = i < 0 ? -i : (i >= n ? 2(n-1) - i : i)
//   adjoint   gx[p][y][x]  = sum of gy over the (at most 2 x 2) padded positions that mirror onto (y, x)
// The adjoint is a GATHER (ATen's reflection_pad2d_backward scatters with atomicAdd, the one non-reproducible
// kernel that was left in the train step); the two kernels are each other's backward, so the op is differentiable
// to any order.
#include "sae_common.h"

namespace sae {
namespace {

struct PadGeom {
    int h, w, oh, ow, left, top;
};

__device__ __forceinline__ int refl(int i, int n) { return i < 0 ? -i : (i >= n ? 2 * (n - 1) - i : i); }

// Row-wise: a wave owns whole output rows (kPadRows at a time), its lanes the columns in runs of 64, four runs in flight.  The
// (plane, row) split is one division per row and wave; the element-wise form spent three 64-bit divisions per 8 bytes moved
// and sat at 2.3 TB/s (profiles/r3_roofline_by_shape_church256.txt) where the streaming kernels reach 5.
constexpr int kPadRows = 2;

__global__ __launch_bounds__(kBlock) void reflect_pad_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                             int64_t planes, const PadGeom q) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t rows = planes * q.oh;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / kWave);
    for (int64_t row0 = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kPadRows; row0 < rows;
         row0 += waves * kPadRows) {
#pragma unroll
        for (int r = 0; r < kPadRows; ++r) {
            const int64_t row = row0 + r;
            if (row >= rows) break;
            const int64_t p = row / q.oh;
            const int oy = (int)(row - p * q.oh);
            const float* src = x + (p * q.h + refl(oy - q.top, q.h)) * q.w;
            float* dst = y + row * q.ow;
            for (int ox0 = 0; ox0 < q.ow; ox0 += 4 * kWave) {
                float v[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ox = ox0 + u * kWave + lane;
                    v[u] = src[ox < q.ow ? refl(ox - q.left, q.w) : 0];      // branch-free load, see upfirdn2d.hip
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ox = ox0 + u * kWave + lane;
                    if (ox < q.ow) dst[ox] = v[u];
                }
            }
        }
    }
}

// padded coordinates o in [0, on) with refl(o - before, n) == i: the direct one and up to two mirrored ones
__device__ __forceinline__ int pre_images(int i, int n, int before, int on, int (&out)[3]) {
    int cnt = 0;
    out[cnt++] = i + before;                                   // always inside [before, before + n)
    const int lo = before - i;                                 // mirrors across index 0: o - before = -i
    if (i > 0 && lo >= 0) out[cnt++] = lo;
    const int hi = before + 2 * (n - 1) - i;                   // mirrors across index n - 1
    if (i < n - 1 && hi < on) out[cnt++] = hi;
    return cnt;
}

// row-wise like the forward; the row's pre-images are shared by the wave, the column's (one, except next to the border) per lane
__global__ __launch_bounds__(kBlock) void reflect_pad_adj_kernel(const float* __restrict__ gy, float* __restrict__ gx,
                                                                 int64_t planes, const PadGeom q) {
    const int lane = threadIdx.x & (kWave - 1);
    const int64_t rows = planes * q.h;
    const int64_t waves = (int64_t)gridDim.x * (kBlock / kWave);
    for (int64_t row0 = ((int64_t)blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6)) * kPadRows; row0 < rows;
         row0 += waves * kPadRows) {
#pragma unroll
        for (int r = 0; r < kPadRows; ++r) {
            const int64_t row = row0 + r;
            if (row >= rows) break;
            const int64_t p = row / q.h;
            const int py = (int)(row - p * q.h);
            int ys[3];
            const int ny = pre_images(py, q.h, q.top, q.oh, ys);
            const float* g = gy + p * q.oh * q.ow;
            float* dst = gx + row * q.w;
            for (int px0 = 0; px0 < q.w; px0 += 4 * kWave) {
                float acc[4];
                // the direct pre-image (every element has one) is fetched branch-free for the four runs, the mirrored ones --
                // border rows and the few border columns -- in a rarely taken branch; the terms are added in the order
                // (row pre-image, column pre-image) of the element-wise form
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = px0 + u * kWave + lane;
                    acc[u] = g[(int64_t)ys[0] * q.ow + (px < q.w ? px + q.left : 0)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = px0 + u * kWave + lane;
                    if (px < q.w) {
                        int xs[3];
                        const int nx = pre_images(px, q.w, q.left, q.ow, xs);
                        if (ny > 1 || nx > 1) {
                            for (int a = 0; a < ny; ++a)
                                for (int b = (a == 0 ? 1 : 0); b < nx; ++b) acc[u] += g[(int64_t)ys[a] * q.ow + xs[b]];
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int px = px0 + u * kWave + lane;
                    if (px < q.w) dst[px] = acc[u];
                }
            }
        }
    }
}

bool pad_ok(int64_t planes, int64_t h, int64_t w, int l, int r, int t, int b) {
    return planes >= 0 && h >= 1 && w >= 1 && l >= 0 && r >= 0 && t >= 0 && b >= 0 && l < w && r < w && t < h && b < h &&
           h < (1 << 15) && w < (1 << 15);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_reflect_pad_f32(const float* x, float* y, int64_t planes, int64_t h, int64_t w, int32_t left,
                                   int32_t right, int32_t top, int32_t bottom, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!pad_ok(planes, h, w, left, right, top, bottom))
        return fail(SAE_EINVAL, "sae_reflect_pad_f32: pads must be non-negative and smaller than the image");
    if (planes == 0) return SAE_OK;
    if (!x || !y) return fail(SAE_EINVAL, "sae_reflect_pad_f32: null tensor");
    const PadGeom q{(int)h, (int)w, (int)h + top + bottom, (int)w + left + right, left, top};
    int64_t blocks = ceil_div64(planes * q.oh, (kBlock / kWave) * kPadRows);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(reflect_pad_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, y, planes, q);
    return check_launch("sae_reflect_pad_f32");
}

extern "C" int sae_reflect_pad_adj_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w, int32_t left,
                                       int32_t right, int32_t top, int32_t bottom, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!pad_ok(planes, h, w, left, right, top, bottom))
        return fail(SAE_EINVAL, "sae_reflect_pad_adj_f32: pads must be non-negative and smaller than the image");
    if (planes == 0) return SAE_OK;
    if (!gy || !gx) return fail(SAE_EINVAL, "sae_reflect_pad_adj_f32: null tensor");
    const PadGeom q{(int)h, (int)w, (int)h + top + bottom, (int)w + left + right, left, top};
    int64_t blocks = ceil_div64(planes * h, (kBlock / kWave) * kPadRows);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(reflect_pad_adj_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, gy, gx, planes,
                       q);
    return check_launch("sae_reflect_pad_adj_f32");
}
