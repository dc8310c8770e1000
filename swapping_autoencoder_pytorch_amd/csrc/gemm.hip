// Strided fp32 GEMM on the matrix cores for the EqualLinear layers
// (reference: F.linear at models/networks/stylegan2_layers.py:177,186 and its ATen backward).
//
//   C[i*ldc + j] = alpha * sum_k A[i*a_si + k*a_sk] * B[k*b_sk + j*b_sj]  (+ bias[j])
//
// One kernel serves forward (A = x, B = W^T), dgrad (A = gy, B = W) and wgrad (A = gy^T, B = x)
// through the strides.  These layers are < 1 % of the step's FLOPs and are
// weight-bandwidth / launch bound: the design goal is "no pathologies", not peak MFMA rate.
//   * 64x64 workgroup tile, 4 waves of 32x32 (v_mfma_f32_32x32x2_f32), K chunks of 32 staged
//     through LDS with whichever of (row, k) is unit-stride in memory mapped onto the lanes;
//   * skinny problems (few 64x64 tiles, long K - the batch-16 style/modulation projections) split
//     K across the 4 waves of a workgroup on a 32x32 tile and reduce through LDS, so a
//     [16 x 2048] x [2048 x 512] product still spreads over 16 workgroups x 4 waves.
#include "sae_common.h"

namespace sae {
namespace {

constexpr int kKc = 32;   // K chunk

struct GemmParams {
    int64_t m, n, k;
    int64_t a_si, a_sk, b_sk, b_sj, ldc;
    float alpha;
};

// stage a [rows x kKc] panel of a strided matrix into LDS as dst[kk][row] (row stride LD)
template <int ROWS, int LD, int NTHREADS>
__device__ __forceinline__ void stage_panel(float* dst, const float* __restrict__ src, int64_t row0, int64_t rows,
                                            int64_t k0, int64_t kmax, int64_t s_row, int64_t s_k, int t) {
    constexpr int NE = ROWS * kKc / NTHREADS;     // loads per thread, all issued before the LDS writes
    static_assert(ROWS * kKc % NTHREADS == 0, "panel must divide evenly");
    // 16-byte loads whenever the unit-stride axis allows (panel fully inside the matrix, strides and base a multiple of four
    // floats): a quarter of the vector-memory instructions -- the skinny products of the linear layers issue two loads per
    // MFMA otherwise and are bound by exactly that
    if constexpr (NE % 4 == 0) {
        constexpr int NQ = NE / 4;
        const bool inside = row0 + ROWS <= rows && k0 + kKc <= kmax && (reinterpret_cast<uintptr_t>(src) & 15) == 0;
        if (inside && s_k == 1 && (s_row & 3) == 0 && (k0 & 3) == 0) {          // k contiguous: quads of four k
            f32x4 q[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int e = t + i * NTHREADS;
                const int r = e / (kKc / 4), kq = e - r * (kKc / 4);
                q[i] = *reinterpret_cast<const f32x4*>(src + (row0 + r) * s_row + k0 + 4 * kq);
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int e = t + i * NTHREADS;
                const int r = e / (kKc / 4), kq = e - r * (kKc / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[(4 * kq + j) * LD + r] = q[i][j];
            }
            return;
        }
        if (inside && s_row == 1 && (s_k & 3) == 0 && (row0 & 3) == 0) {        // rows contiguous: quads of four rows
            f32x4 q[NQ];
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int e = t + i * NTHREADS;
                const int kk = e / (ROWS / 4), rq = e - kk * (ROWS / 4);
                q[i] = *reinterpret_cast<const f32x4*>(src + (row0 + 4 * rq) + (k0 + kk) * s_k);
            }
#pragma unroll
            for (int i = 0; i < NQ; ++i) {
                const int e = t + i * NTHREADS;
                const int kk = e / (ROWS / 4), rq = e - kk * (ROWS / 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) dst[kk * LD + 4 * rq + j] = q[i][j];
            }
            return;
        }
    }
    float v[NE];
    if (s_k == 1) {   // k contiguous in memory: lanes walk k
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = t + i * NTHREADS;
            const int r = e / kKc, kk = e - r * kKc;
            v[i] = (row0 + r < rows && k0 + kk < kmax) ? src[(row0 + r) * s_row + (k0 + kk)] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = t + i * NTHREADS;
            const int r = e / kKc, kk = e - r * kKc;
            dst[kk * LD + r] = v[i];
        }
    } else {          // rows contiguous (or generic): lanes walk rows
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = t + i * NTHREADS;
            const int kk = e / ROWS, r = e - kk * ROWS;
            v[i] = (row0 + r < rows && k0 + kk < kmax) ? src[(row0 + r) * s_row + (k0 + kk) * s_k] : 0.0f;
        }
#pragma unroll
        for (int i = 0; i < NE; ++i) {
            const int e = t + i * NTHREADS;
            const int kk = e / ROWS, r = e - kk * ROWS;
            dst[kk * LD + r] = v[i];
        }
    }
}

__global__ __launch_bounds__(kBlock) void gemm64_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                        const float* __restrict__ bias, float* __restrict__ c,
                                                        const GemmParams p) {
    constexpr int BT = 64, LD = BT + 1;
    __shared__ float As[kKc * LD];
    __shared__ float Bs[kKc * LD];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 1, wn = wid & 1;
    const int64_t i0 = (int64_t)blockIdx.y * BT, j0 = (int64_t)blockIdx.x * BT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    for (int64_t k0 = 0; k0 < p.k; k0 += kKc) {
        __syncthreads();
        stage_panel<BT, LD, kBlock>(As, a, i0, p.m, k0, p.k, p.a_si, p.a_sk, tid);
        stage_panel<BT, LD, kBlock>(Bs, b, j0, p.n, k0, p.k, p.b_sj, p.b_sk, tid);
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < kKc / 2; ++kp) {
            const float av = As[(2 * kp + half) * LD + wm * 32 + l31];
            const float bv = Bs[(2 * kp + half) * LD + wn * 32 + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
    }
    const int64_t j = j0 + wn * 32 + l31;
    if (j < p.n) {
        const float bj = bias ? bias[j] : 0.0f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int64_t i = i0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            if (i < p.m) c[i * p.ldc + j] = p.alpha * acc[r] + bj;
        }
    }
}

// 32x32 output tile per workgroup; the 4 waves take K chunks round-robin and reduce through LDS.
__global__ __launch_bounds__(kBlock) void gemm32_splitk_kernel(const float* __restrict__ a,
                                                               const float* __restrict__ b,
                                                               const float* __restrict__ bias,
                                                               float* __restrict__ c, const GemmParams p) {
    constexpr int BT = 32, LD = BT + 1;
    __shared__ float As[4][kKc * LD];
    __shared__ float Bs[4][kKc * LD];
    __shared__ float red[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.y * BT, j0 = (int64_t)blockIdx.x * BT;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int64_t nchunks = ceil_div64(p.k, kKc);
    // every wave runs the same number of iterations so the (wave-private) staging never diverges
    const int64_t iters = ceil_div64(nchunks, 4);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t k0 = (it * 4 + wid) * kKc;   // may be >= k: stages zeros
        stage_panel<BT, LD, kWave>(As[wid], a, i0, p.m, k0, p.k, p.a_si, p.a_sk, lane);
        stage_panel<BT, LD, kWave>(Bs[wid], b, j0, p.n, k0, p.k, p.b_sj, p.b_sk, lane);
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < kKc / 2; ++kp) {
            const float av = As[wid][(2 * kp + half) * LD + l31];
            const float bv = Bs[wid][(2 * kp + half) * LD + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = acc[r];
    __syncthreads();
    for (int e = tid; e < 32 * 32; e += kBlock) {
        const int i = e >> 5, j = e & 31;
        const float v = (red[0][i * 33 + j] + red[1][i * 33 + j]) + (red[2][i * 33 + j] + red[3][i * 33 + j]);
        if (i0 + i < p.m && j0 + j < p.n)
            c[(i0 + i) * p.ldc + (j0 + j)] = p.alpha * v + (bias ? bias[j0 + j] : 0.0f);
    }
}

// K split ACROSS workgroups (sae_gemm_ws_f32): workgroup (x, y, z) multiplies the 32 x 32 tile (y, x) over K slice z -- its 4 waves
// take the slice's chunks round-robin and meet in LDS, as above -- and writes the UNSCALED partial tile to ws[z][m][n]; the
// reduction kernel adds the slices in order (fixed summation order: bit-reproducible), scales and adds the bias.  The skinny
// products of the step (batch 16 - 128 rows against 2048 x 2048 ... 8192 x 512 weights) launched 16 - 256 workgroups and read
// their weight once at 0.4 - 1.6 TB/s; split into ~768 workgroups they are a weight-bandwidth problem again.
__global__ __launch_bounds__(kBlock) void gemm32_slice_kernel(const float* __restrict__ a, const float* __restrict__ b,
                                                              float* __restrict__ ws, const GemmParams p, int64_t chunks_per_slice) {
    constexpr int BT = 32, LD = BT + 1;
    __shared__ float As[4][kKc * LD];
    __shared__ float Bs[4][kKc * LD];
    __shared__ float red[4][32 * 33];
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int64_t i0 = (int64_t)blockIdx.y * BT, j0 = (int64_t)blockIdx.x * BT;
    const int64_t c_begin = (int64_t)blockIdx.z * chunks_per_slice;
    const int64_t nchunks = ceil_div64(p.k, kKc);
    const int64_t c_end = c_begin + chunks_per_slice < nchunks ? c_begin + chunks_per_slice : nchunks;
    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.0f;
    const int64_t iters = ceil_div64(c_end - c_begin, 4);
    for (int64_t it = 0; it < iters; ++it) {
        const int64_t ck = c_begin + it * 4 + wid;
        // a chunk beyond the slice stages zeros (kmax = the slice's end), so every wave runs the same number of iterations
        const int64_t k0 = ck * kKc;
        const int64_t kmax = c_end * kKc < p.k ? c_end * kKc : p.k;
        stage_panel<BT, LD, kWave>(As[wid], a, i0, p.m, k0, ck < c_end ? kmax : 0, p.a_si, p.a_sk, lane);
        stage_panel<BT, LD, kWave>(Bs[wid], b, j0, p.n, k0, ck < c_end ? kmax : 0, p.b_sj, p.b_sk, lane);
        __syncthreads();
#pragma unroll
        for (int kp = 0; kp < kKc / 2; ++kp) {
            const float av = As[wid][(2 * kp + half) * LD + l31];
            const float bv = Bs[wid][(2 * kp + half) * LD + l31];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv, acc, 0, 0, 0);
        }
        __syncthreads();
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) red[wid][((r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = acc[r];
    __syncthreads();
    float* out = ws + (int64_t)blockIdx.z * p.m * p.n;
    for (int e = tid; e < 32 * 32; e += kBlock) {
        const int i = e >> 5, j = e & 31;
        const float v = (red[0][i * 33 + j] + red[1][i * 33 + j]) + (red[2][i * 33 + j] + red[3][i * 33 + j]);
        if (i0 + i < p.m && j0 + j < p.n) out[(i0 + i) * p.n + (j0 + j)] = v;
    }
}

__global__ __launch_bounds__(kBlock) void gemm_slices_reduce_kernel(const float* __restrict__ ws, const float* __restrict__ bias,
                                                                    float* __restrict__ c, int64_t m, int64_t n, int64_t ldc,
                                                                    int slices, float alpha) {
    const int64_t e = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (e >= m * n) return;
    const int64_t i = e / n, j = e - i * n;
    float v = ws[e];
    for (int s = 1; s < slices; ++s) v += ws[(int64_t)s * m * n + e];
    c[i * ldc + j] = alpha * v + (bias ? bias[j] : 0.0f);
}

// K slices of the split form (0: the one-launch kernels are the better choice)
inline int gemm_slices(int64_t m, int64_t n, int64_t k) {
    const int64_t tiles64 = ceil_div64(m, 64) * ceil_div64(n, 64);
    if (!(tiles64 < 128 && k >= 256)) return 0;                 // sae_gemm_f32's own rule for the 32 x 32 split-K kernel
    const int64_t tiles32 = ceil_div64(m, 32) * ceil_div64(n, 32);
    // from 192 tiles on the one-launch kernel already fills the chip and the slices + reduction only add traffic (measured in the
    // step, round 6: [128 x 2048] x [2048 x 2048] 35 -> 45 us with three slices; the deep-K shapes under that count gain:
    // [24 x 8192] x [8192 x 512] 142 -> 40 us, [16 x 4376] x [4376 x 2048] 84 -> 37)
    if (tiles32 >= 192) return 0;
    const int64_t nchunks = ceil_div64(k, kKc);
    int64_t s = ceil_div64(768, tiles32);                       // ~3 workgroups per CU
    const int64_t most = nchunks / 4 > 0 ? nchunks / 4 : 1;     // at least one chunk per wave and slice
    if (s > most) s = most;
    return s >= 2 ? (int)s : 0;
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int64_t sae_gemm_workspace(int64_t m, int64_t n, int64_t k) {
    if (m < 1 || n < 1 || k < 1) return 0;
    return (int64_t)gemm_slices(m, n, k) * m * n;
}

extern "C" int sae_gemm_ws_f32(const float* a, const float* b, const float* bias, float* c, int64_t m, int64_t n, int64_t k,
                               int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc, float alpha, float* workspace,
                               int64_t workspace_floats, sae_stream_t stream) {
    const int slices = (m > 0 && n > 0 && k > 0) ? gemm_slices(m, n, k) : 0;
    if (slices == 0) return sae_gemm_f32(a, b, bias, c, m, n, k, a_si, a_sk, b_sk, b_sj, ldc, alpha, stream);
    sae::clear_stale_error();
    if (!c || !a || !b) return fail(SAE_EINVAL, "sae_gemm_ws_f32: null matrix");
    if (ldc < n) return fail(SAE_EINVAL, "sae_gemm_ws_f32: ldc < n");
    if (!workspace || workspace_floats < (int64_t)slices * m * n)
        return fail(SAE_EWORKSPACE, "sae_gemm_ws_f32: workspace %lld < %lld floats (sae_gemm_workspace)", (long long)workspace_floats,
                    (long long)((int64_t)slices * m * n));
    GemmParams p{m, n, k, a_si, a_sk, b_sk, b_sj, ldc, alpha};
    hipStream_t s = (hipStream_t)stream;
    const int64_t cps = ceil_div64(ceil_div64(k, kKc), slices);
    const dim3 grid((unsigned)ceil_div64(n, 32), (unsigned)ceil_div64(m, 32), (unsigned)slices);
    hipLaunchKernelGGL(gemm32_slice_kernel, grid, dim3(kBlock), 0, s, a, b, workspace, p, cps);
    int rc = check_launch("sae_gemm_ws_f32");
    if (rc != SAE_OK) return rc;
    hipLaunchKernelGGL(gemm_slices_reduce_kernel, dim3((unsigned)ceil_div64(m * n, kBlock)), dim3(kBlock), 0, s,
                       (const float*)workspace, bias, c, m, n, ldc, slices, alpha);
    return check_launch("sae_gemm_ws_f32 (reduce)");
}

extern "C" int sae_gemm_f32(const float* a, const float* b, const float* bias, float* c, int64_t m, int64_t n,
                            int64_t k, int64_t a_si, int64_t a_sk, int64_t b_sk, int64_t b_sj, int64_t ldc,
                            float alpha, sae_stream_t stream) {
    sae::clear_stale_error();
    if (m < 0 || n < 0 || k < 0) return fail(SAE_EINVAL, "sae_gemm_f32: negative size");
    if (m == 0 || n == 0) return SAE_OK;
    if (!c || (k > 0 && (!a || !b))) return fail(SAE_EINVAL, "sae_gemm_f32: null matrix");
    if (ldc < n) return fail(SAE_EINVAL, "sae_gemm_f32: ldc < n");
    GemmParams p{m, n, k, a_si, a_sk, b_sk, b_sj, ldc, alpha};
    hipStream_t s = (hipStream_t)stream;
    const int64_t tiles64 = ceil_div64(m, 64) * ceil_div64(n, 64);
    if (tiles64 < 128 && k >= 256) {
        const dim3 grid((unsigned)ceil_div64(n, 32), (unsigned)ceil_div64(m, 32));
        hipLaunchKernelGGL(gemm32_splitk_kernel, grid, dim3(kBlock), 0, s, a, b, bias, c, p);
    } else {
        const dim3 grid((unsigned)ceil_div64(n, 64), (unsigned)ceil_div64(m, 64));
        hipLaunchKernelGGL(gemm64_kernel, grid, dim3(kBlock), 0, s, a, b, bias, c, p);
    }
    return check_launch("sae_gemm_f32");
}
