// Random-crop sampler of the patch discriminator.
//
// Replaces util.apply_random_crop's F.grid_sample call (reference: util/util.py:323-343, called from
// swapping_autoencoder_model.py:84-93): every crop is an axis-aligned, optionally mirrored window of the image,
// resampled bilinearly (align_corners = False, zero padding) to size x size.  The reference expands the image
// num_crops times and builds a [B*crops, size, size, 2] grid; here the kernel reads the source image directly
// from five numbers per crop (flip, scale x/y, offset x/y) and the shared linspace(-1, 1, size).
//
// Forward: one thread per output pixel, all channels (same arithmetic as grid_sampler: unnormalise
// ((g + 1) * W - 1) / 2, corner weights (x1 - ix)(y1 - iy) ..., corners accumulated in nw, ne, sw, se order).
// Backward w.r.t. the image: a GATHER — one thread per image pixel walks the few output pixels of each crop
// whose bilinear footprint touches it (the map is separable and monotonic, so they form a small rectangle),
// re-deriving their coordinates with the forward's formula.  Deterministic, no atomics (ATen scatters with
// atomicAdd).
#include "sae_common.h"

// The sampling coordinates are formed exactly as ATen's CPU grid_sampler forms them (separately rounded
// multiply / add steps): hipcc's default contraction into FMAs moves a coordinate by one ulp of ~256, i.e. the
// bilinear weights by ~1e-5.
#pragma clang fp contract(off)

namespace sae {
namespace {

struct CropGeom {
    int images, channels, h, w, crops, size;
};

// source coordinate of output index i: grid value g = lin[i] * mul + off, then grid_sampler's unnormalise
__device__ __forceinline__ float crop_coord(float lin, float mul, float off, int extent) {
    // plain operators: the `fp contract(off)` pragma above governs them (intrinsics such as __fmul_rn are inlined
    // library code the pragma does not reach, and were contracted into FMAs)
    const float g = lin * mul + off;
    return ((g + 1.0f) * (float)extent - 1.0f) * 0.5f;
}

__global__ __launch_bounds__(kBlock) void random_crop_kernel(const float* __restrict__ x,
                                                             const float* __restrict__ params,
                                                             const float* __restrict__ lin, float* __restrict__ y,
                                                             const CropGeom q) {
    const int64_t total = (int64_t)q.images * q.crops * q.size * q.size;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int c = (int)(i % q.size);
        const int r = (int)((i / q.size) % q.size);
        const int64_t k = i / ((int64_t)q.size * q.size);
        const int b = (int)(k / q.crops);
        const float* pr = params + 5 * k;
        const float ix = crop_coord(lin[c] * pr[0], pr[1], pr[3], q.w);
        const float iy = crop_coord(lin[r], pr[2], pr[4], q.h);
        const float fx = floorf(ix), fy = floorf(iy);
        const int x0 = (int)fx, y0 = (int)fy;
        const float wx1 = ix - fx, wx0 = (fx + 1.0f) - ix;
        const float wy1 = iy - fy, wy0 = (fy + 1.0f) - iy;
        const bool vx0 = x0 >= 0 && x0 < q.w, vx1 = x0 + 1 >= 0 && x0 + 1 < q.w;
        const bool vy0 = y0 >= 0 && y0 < q.h, vy1 = y0 + 1 >= 0 && y0 + 1 < q.h;
        const float nw = wx0 * wy0, ne = wx1 * wy0, sw = wx0 * wy1, se = wx1 * wy1;
        for (int ch = 0; ch < q.channels; ++ch) {
            const float* xp = x + ((int64_t)b * q.channels + ch) * q.h * q.w;
            float acc = 0.0f;
            if (vy0 && vx0) acc += xp[(int64_t)y0 * q.w + x0] * nw;
            if (vy0 && vx1) acc += xp[(int64_t)y0 * q.w + x0 + 1] * ne;
            if (vy1 && vx0) acc += xp[(int64_t)(y0 + 1) * q.w + x0] * sw;
            if (vy1 && vx1) acc += xp[(int64_t)(y0 + 1) * q.w + x0 + 1] * se;
            y[((k * q.channels + ch) * q.size + r) * q.size + c] = acc;
        }
    }
}

// output indices i whose source coordinate lies in (t - 1, t + 1): candidates from the inverse map, widened by
// two, each confirmed with the forward's own arithmetic
__device__ __forceinline__ void crop_range(float t, float mul, float off, int extent, int size, int* lo, int* hi) {
    // coord(i) = ((lin(i) * mul + off + 1) * extent - 1) / 2, lin(i) = -1 + 2 i / (size - 1)
    const float gl = (2.0f * (t - 1.0f) + 1.0f) / extent - 1.0f, gh = (2.0f * (t + 1.0f) + 1.0f) / extent - 1.0f;
    float a = (gl - off) / mul, b = (gh - off) / mul;
    if (a > b) { const float s = a; a = b; b = s; }
    const float ia = (a + 1.0f) * 0.5f * (size - 1), ib = (b + 1.0f) * 0.5f * (size - 1);
    int l = (int)floorf(ia) - 2, h = (int)ceilf(ib) + 2;
    if (l < 0) l = 0;
    if (h > size - 1) h = size - 1;
    *lo = l; *hi = h;
}

// does the bilinear footprint of the output index with linspace value `lin` cover image coordinate t ?
__device__ __forceinline__ bool crop_touches(float lin, float mul, float off, int extent, int t) {
    const int i0 = (int)floorf(crop_coord(lin, mul, off, extent));
    return i0 == t || i0 + 1 == t;
}

__global__ __launch_bounds__(kBlock) void random_crop_bwd_kernel(const float* __restrict__ gy,
                                                                 const float* __restrict__ params,
                                                                 const float* __restrict__ lin,
                                                                 float* __restrict__ gx, const CropGeom q) {
    // The crop geometry (candidate ranges, bilinear weights) depends on the pixel and the crop only: it is worked out once
    // and applied to up to kCropCh channels at a time (images have 3); per channel the terms are added in the same
    // (crop, row, column) order as before, so the result is unchanged.
    constexpr int kCropCh = 4;
    const int64_t total = (int64_t)q.images * q.h * q.w;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total; i += (int64_t)gridDim.x * kBlock) {
        const int px = (int)(i % q.w);
        const int py = (int)((i / q.w) % q.h);
        const int b = (int)(i / ((int64_t)q.w * q.h));
        for (int ch0 = 0; ch0 < q.channels; ch0 += kCropCh) {
            const int nch = q.channels - ch0 < kCropCh ? q.channels - ch0 : kCropCh;
            float acc[kCropCh] = {0.0f, 0.0f, 0.0f, 0.0f};
            for (int kc = 0; kc < q.crops; ++kc) {
                const int64_t k = (int64_t)b * q.crops + kc;
                const float* pr = params + 5 * k;
                // the window of the crop in the image, from the source coordinates of its first and last output index (the map is
                // linear in the index): a pixel more than one column / row outside it (plus one of slack for the rounding of the
                // ends) receives nothing from this crop.  The windows cover 2 - 6 % of the image, so this skips the range
                // inversion below (four divisions) for almost every (pixel, crop) pair.
                const float xa = crop_coord(lin[0] * pr[0], pr[1], pr[3], q.w), xb = crop_coord(lin[q.size - 1] * pr[0], pr[1], pr[3], q.w);
                if ((float)px < floorf(fminf(xa, xb)) - 1.0f || (float)px > floorf(fmaxf(xa, xb)) + 2.0f) continue;
                const float ya = crop_coord(lin[0], pr[2], pr[4], q.h), yb = crop_coord(lin[q.size - 1], pr[2], pr[4], q.h);
                if ((float)py < floorf(fminf(ya, yb)) - 1.0f || (float)py > floorf(fmaxf(ya, yb)) + 2.0f) continue;
                int c_lo, c_hi, r_lo, r_hi;
                crop_range((float)px, pr[0] * pr[1], pr[3], q.w, q.size, &c_lo, &c_hi);
                if (c_lo > c_hi) continue;
                crop_range((float)py, pr[2], pr[4], q.h, q.size, &r_lo, &r_hi);
                // the candidate ranges are the inverse map widened by two on each side; the map is monotonic, so the outputs
                // that really touch the pixel are contiguous: trim the ends with the forward's own arithmetic before the
                // double loop (it ran over ~100 candidate pairs for ~30 contributing ones, on the few waves whose pixels lie
                // inside a crop window -- the kernel's critical path: 0.23 ms for 13 MB)
                while (c_lo <= c_hi && !crop_touches(lin[c_lo] * pr[0], pr[1], pr[3], q.w, px)) ++c_lo;
                while (c_hi >= c_lo && !crop_touches(lin[c_hi] * pr[0], pr[1], pr[3], q.w, px)) --c_hi;
                if (c_lo > c_hi) continue;
                while (r_lo <= r_hi && !crop_touches(lin[r_lo], pr[2], pr[4], q.h, py)) ++r_lo;
                while (r_hi >= r_lo && !crop_touches(lin[r_hi], pr[2], pr[4], q.h, py)) --r_hi;
                const int64_t plane = (int64_t)q.size * q.size;
                const float* gp = gy + (k * q.channels + ch0) * plane;
                for (int r = r_lo; r <= r_hi; ++r) {
                    const float iy = crop_coord(lin[r], pr[2], pr[4], q.h);
                    const float fy = floorf(iy);
                    const int y0 = (int)fy;
                    float wy;
                    if (y0 == py) wy = (fy + 1.0f) - iy;
                    else if (y0 + 1 == py) wy = iy - fy;
                    else continue;
                    for (int c = c_lo; c <= c_hi; ++c) {
                        const float ix = crop_coord(lin[c] * pr[0], pr[1], pr[3], q.w);
                        const float fx = floorf(ix);
                        const int x0 = (int)fx;
                        float wx;
                        if (x0 == px) wx = (fx + 1.0f) - ix;
                        else if (x0 + 1 == px) wx = ix - fx;
                        else continue;
                        const float wgt = wx * wy;
#pragma unroll
                        for (int e = 0; e < kCropCh; ++e)
                            if (e < nch) acc[e] += gp[e * plane + r * q.size + c] * wgt;
                    }
                }
            }
#pragma unroll
            for (int e = 0; e < kCropCh; ++e)
                if (e < nch) gx[((int64_t)b * q.channels + ch0 + e) * q.h * q.w + (int64_t)py * q.w + px] = acc[e];
        }
    }
}

bool crop_ok(int64_t images, int64_t channels, int64_t h, int64_t w, int64_t crops, int64_t size) {
    return images >= 0 && channels >= 1 && h >= 1 && w >= 1 && crops >= 1 && size >= 2 && h < (1 << 15) && w < (1 << 15) &&
           size < (1 << 15) && channels < (1 << 20) && images * crops < ((int64_t)1 << 31);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_random_crop_f32(const float* x, const float* params, const float* lin, float* y, int64_t images,
                                   int64_t channels, int64_t h, int64_t w, int64_t crops_per_image, int64_t size,
                                   sae_stream_t stream) {
    sae::clear_stale_error();
    if (!crop_ok(images, channels, h, w, crops_per_image, size)) return fail(SAE_EINVAL, "sae_random_crop_f32: bad geometry");
    if (images == 0) return SAE_OK;
    if (!x || !params || !lin || !y) return fail(SAE_EINVAL, "sae_random_crop_f32: null tensor");
    const CropGeom q{(int)images, (int)channels, (int)h, (int)w, (int)crops_per_image, (int)size};
    int64_t blocks = ceil_div64(images * crops_per_image * size * size, kBlock);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(random_crop_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, x, params, lin, y, q);
    return check_launch("sae_random_crop_f32");
}

extern "C" int sae_random_crop_bwd_f32(const float* gy, const float* params, const float* lin, float* gx, int64_t images,
                                       int64_t channels, int64_t h, int64_t w, int64_t crops_per_image, int64_t size,
                                       sae_stream_t stream) {
    sae::clear_stale_error();
    if (!crop_ok(images, channels, h, w, crops_per_image, size)) return fail(SAE_EINVAL, "sae_random_crop_bwd_f32: bad geometry");
    if (images == 0) return SAE_OK;
    if (!gy || !params || !lin || !gx) return fail(SAE_EINVAL, "sae_random_crop_bwd_f32: null tensor");
    const CropGeom q{(int)images, (int)channels, (int)h, (int)w, (int)crops_per_image, (int)size};
    int64_t blocks = ceil_div64(images * h * w, kBlock);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(random_crop_bwd_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, (hipStream_t)stream, gy, params, lin,
                       gx, q);
    return check_launch("sae_random_crop_bwd_f32");
}
