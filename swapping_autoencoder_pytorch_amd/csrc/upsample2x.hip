// Bilinear x2 upsampling (align_corners = false) fused with the residual add and scale of the
// generator's upsampling blocks:  y = alpha * (up2x(x) + res)   and its adjoint.
//
// Replaces F.interpolate(scale_factor=2, mode='bilinear', align_corners=False) followed by
// (skip + res) / sqrt(2) in the reference (models/networks/generator.py:51-53): three elementwise
// passes over the full-resolution tensor (ATen upsample kernel, add, div) become one.  The
// per-output arithmetic follows ATen's published bilinear formula: for output index d,
//   src = max((d + 0.5) / 2 - 0.5, 0),  i0 = floor(src),  l1 = src - i0,  l0 = 1 - l1,
//   i1 = i0 + (i0 < n - 1),             out = l0 * in[i0] + l1 * in[i1]      (rows, then columns)
// HBM-bound: reads N + 4N (res), writes 4N floats per plane.  One thread per INPUT pixel produces
// the 2x2 output quad (8-byte stores, a wave writes 512 contiguous bytes per output row).
#include "sae_common.h"

namespace sae {
namespace {

struct Up2Params {
    int64_t planes;
    int h, w;
    float alpha;
};

__device__ __forceinline__ void src_index(int d, int n, int& i0, int& i1, float& l0, float& l1) {
    float src = (d + 0.5f) * 0.5f - 0.5f;
    src = src < 0.0f ? 0.0f : src;
    i0 = (int)src;
    l1 = src - (float)i0;
    l0 = 1.0f - l1;
    i1 = i0 + ((i0 < n - 1) ? 1 : 0);
}

__global__ __launch_bounds__(kBlock) void upsample2x_add_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ res,
                                                                float* __restrict__ y, const Up2Params p) {
    const int64_t per_plane = (int64_t)p.h * p.w;
    const int64_t total = p.planes * per_plane;
    for (int64_t idx = (int64_t)blockIdx.x * kBlock + threadIdx.x; idx < total;
         idx += (int64_t)gridDim.x * kBlock) {
        const int64_t plane = idx / per_plane;
        const int rem = (int)(idx - plane * per_plane);
        const int i = rem / p.w, j = rem - i * p.w;
        const float* xp = x + plane * per_plane;
        const int64_t obase = plane * per_plane * 4;
        float out[2][2];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            int y0, y1; float hl0, hl1;
            src_index(2 * i + r, p.h, y0, y1, hl0, hl1);
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                int x0, x1; float wl0, wl1;
                src_index(2 * j + c, p.w, x0, x1, wl0, wl1);
                const float a = xp[y0 * p.w + x0], b = xp[y0 * p.w + x1];
                const float cc = xp[y1 * p.w + x0], d = xp[y1 * p.w + x1];
                out[r][c] = hl0 * (wl0 * a + wl1 * b) + hl1 * (wl0 * cc + wl1 * d);
            }
        }
        // both rows' residual pairs are fetched before the first row is stored: fetched row by row, the second row's load waited
        // for the first row's store to reach the L2 as well (vmcnt counts stores)
        float2 q[2] = {make_float2(0.0f, 0.0f), make_float2(0.0f, 0.0f)};
        if (res) {
#pragma unroll
            for (int r = 0; r < 2; ++r) q[r] = *reinterpret_cast<const float2*>(res + obase + (int64_t)(2 * i + r) * (2 * p.w) + 2 * j);
        }
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const int64_t o = obase + (int64_t)(2 * i + r) * (2 * p.w) + 2 * j;
            float2 v = make_float2(out[r][0], out[r][1]);
            if (res) { v.x += q[r].x; v.y += q[r].y; }
            v.x *= p.alpha; v.y *= p.alpha;
            *reinterpret_cast<float2*>(y + o) = v;
        }
    }
}

// adjoint: gx[i][j] = alpha * sum over the 4x4 output neighbourhood rows 2i-1..2i+2, cols
// 2j-1..2j+2 (clamped to the image) with separable weights (1/4, 3/4, 3/4, 1/4).
// A lane fetches the two middle columns (2j, 2j+1) of each of the four rows as one 8-byte load -- a wave reads 512
// contiguous bytes per instruction -- and takes the outer two (2j-1, 2j+2) from its neighbours' registers; only the first and
// last lane of a wave fetch theirs.  (Sixteen stride-2 dword loads per lane before: 4x the bytes through the L1 and 2.0 TB/s,
// profiles/r3_roofline_by_shape_church256.txt.)  Consecutive lanes own consecutive input pixels of the flattened tensor;
// where a row ends inside a wave the clamp, not the neighbour, supplies the outer column.
template <typename IdxT>
__global__ __launch_bounds__(kBlock) void upsample2x_bwd_kernel(const float* __restrict__ gy,
                                                                float* __restrict__ gx, const Up2Params p, IdxT total) {
    const IdxT per_plane = (IdxT)p.h * (IdxT)p.w;
    const int oh = 2 * p.h, ow = 2 * p.w;
    const int lane = threadIdx.x & (kWave - 1);
    // whole waves iterate together (the trip count is taken from the wave's first lane) so that the register exchange below
    // always finds its neighbour
    const IdxT stride = (IdxT)gridDim.x * kBlock;
    for (IdxT base = (IdxT)blockIdx.x * kBlock + (threadIdx.x & ~(kWave - 1)); base < total; base += stride) {
        const IdxT idx = base + lane;
        const bool live = idx < total;
        const IdxT ii = live ? idx : total - 1;
        const IdxT plane = ii / per_plane;
        const int rem = (int)(ii - plane * per_plane);
        const int i = rem / p.w, j = rem - i * p.w;
        const float* gp = gy + (int64_t)plane * per_plane * 4;
        float2 mid[4];
        float left[4], right[4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            int yy = 2 * i - 1 + a;
            yy = yy < 0 ? 0 : (yy > oh - 1 ? oh - 1 : yy);
            const float* rowp = gp + (int64_t)yy * ow;
            mid[a] = *reinterpret_cast<const float2*>(rowp + 2 * j);
            // the outer columns of the wave's end lanes (clamped to the row: reads the lane's own middle column at the border)
            left[a] = (lane == 0) ? rowp[j > 0 ? 2 * j - 1 : 0] : 0.0f;
            right[a] = (lane == kWave - 1) ? rowp[j < p.w - 1 ? 2 * j + 2 : ow - 1] : 0.0f;
        }
        float acc = 0.0f;
        const float wgt[4] = {0.25f, 0.75f, 0.75f, 0.25f};
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            const float from_prev = __shfl(mid[a].y, lane > 0 ? lane - 1 : 0, kWave);
            const float from_next = __shfl(mid[a].x, lane < kWave - 1 ? lane + 1 : kWave - 1, kWave);
            const float l = (j == 0) ? mid[a].x : (lane == 0 ? left[a] : from_prev);
            const float r = (j == p.w - 1) ? mid[a].y : (lane == kWave - 1 ? right[a] : from_next);
            float row = 0.0f;
            row = fmaf(wgt[0], l, row);
            row = fmaf(wgt[1], mid[a].x, row);
            row = fmaf(wgt[2], mid[a].y, row);
            row = fmaf(wgt[3], r, row);
            acc = fmaf(wgt[a], row, acc);
        }
        if (live) gx[idx] = p.alpha * acc;
    }
}

}  // namespace
}  // namespace sae

using namespace sae;

static int check_up2(const char* who, const void* a, const void* b, int64_t planes, int64_t h, int64_t w) {
    if (planes < 0 || h < 1 || w < 1 || h > (1 << 14) || w > (1 << 14))
        return fail(SAE_EINVAL, "%s: bad plane size", who);
    if (planes > 0 && (!a || !b)) return fail(SAE_EINVAL, "%s: null tensor", who);
    return SAE_OK;
}

extern "C" int sae_upsample2x_bilinear_add_f32(const float* x, const float* res, float* y, int64_t planes,
                                               int64_t h, int64_t w, float alpha, sae_stream_t stream) {
    sae::clear_stale_error();
    const int rc = check_up2("sae_upsample2x_bilinear_add_f32", x, y, planes, h, w);
    if (rc != SAE_OK || planes == 0) return rc;
    Up2Params p{planes, (int)h, (int)w, alpha};
    int64_t blocks = ceil_div64(planes * h * w, kBlock);
    if (blocks > 32768) blocks = 32768;
    hipLaunchKernelGGL(upsample2x_add_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, res, y, p);
    return check_launch("sae_upsample2x_bilinear_add_f32");
}

extern "C" int sae_upsample2x_bilinear_bwd_f32(const float* gy, float* gx, int64_t planes, int64_t h, int64_t w,
                                               float alpha, sae_stream_t stream) {
    sae::clear_stale_error();
    const int rc = check_up2("sae_upsample2x_bilinear_bwd_f32", gy, gx, planes, h, w);
    if (rc != SAE_OK || planes == 0) return rc;
    Up2Params p{planes, (int)h, (int)w, alpha};
    const int64_t total = planes * h * w;
    int64_t blocks = ceil_div64(total, kBlock);
    if (blocks > 32768) blocks = 32768;
    if (total + blocks * kBlock < ((int64_t)1 << 31))
        hipLaunchKernelGGL((upsample2x_bwd_kernel<uint32_t>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, gy, gx,
                           p, (uint32_t)total);
    else
        hipLaunchKernelGGL((upsample2x_bwd_kernel<int64_t>), dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, gy, gx, p,
                           (int64_t)total);
    return check_launch("sae_upsample2x_bilinear_bwd_f32");
}
