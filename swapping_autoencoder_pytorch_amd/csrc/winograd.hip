// Winograd F(2x2, 3x3) for the 3x3 stride-1 convolutions (pad 1 or 0) of the wide layers (the transforms; the 16 products of the
// transform domain are 1x1 convolutions on the MFMA gather of conv2d.hip).
//
// The reference runs these layers through F.conv2d (models/networks/stylegan2_layers.py:136,315), i.e. whatever algorithm
// cuDNN / MIOpen picks -- for 3x3 fp32 with 256+ channels that is a Winograd variant.  A 2x2 output tile needs 16
// multiplications per (output channel, input channel) instead of 36:
//
//     Y = A^T [ (G g G^T) o (B^T d B) ] A        g: 3x3 filter, d: 4x4 input tile (stride 2 between tiles), Y: 2x2 outputs
//     B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]    G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]    A^T = [1 1 1 0; 0 1 -1 -1]
//
// Unfused, three steps, each a plain tensor in HBM:
//     V[xi][n][c][t]  = (B^T d B)[xi]              sae_wino_input_f32    (x read once, V = 4x its size written)
//     M[xi][n][m][t]  = sum_c U[xi][m][c] V[xi][n][c][t]      16 launches of the 1x1 gather (K = C, pixels = tiles)
//     y[n][m][2ty+a][2tx+b] = (A^T M A)[a][b]     sae_wino_output_f32   (+ optional bias + leaky-ReLU, as the fused epilogue)
// with U[xi][m][c] = alpha (G g G^T)[xi] from sae_wino_weights_f32 (flip = 1: the filter of the data gradient, taps reversed;
// the caller swaps the roles of the two channel strides).  The matrix work falls by 2.25; the price is moving 4x the
// activation through HBM twice, so it pays where a layer's FLOP per activation byte are high: 256 channels and up on maps up
// to 64 wide (estimate from the per-shape ledger: HISTORY.md 4.0f).  Exact-fp32 arithmetic throughout; the result differs from
// the direct kernels' by rounding (different association), ~1e-6 relative -- inside the per-op tolerance of 1e-4
// (BASELINE.json) and of the kernel tests (2e-5), but NOT bit-identical to them.
//
// Transform kernels: one thread per (plane, tile); consecutive threads = consecutive tiles of a row, so each of the 16
// transform-domain planes is written / read in coalesced runs and the 2x2 outputs leave as 8-byte stores.
#include "sae_common.h"

// SAE_TRACE_DISPATCH=1 (tuning builds): one stderr line per launch decision the tests want to see
#define SAE_WINO_TRACE(what)                                                   \
    do {                                                                       \
        static const int trace_knob = tuning_knob("SAE_TRACE_DISPATCH", 0);    \
        if (trace_knob) fprintf(stderr, "sae-dispatch wino %s\n", what);       \
    } while (0)

namespace sae {
namespace {

// U[xi][m][c], xi = 4 a + b:  (G g G^T)[a][b]
__global__ __launch_bounds__(kBlock) void wino_weight_kernel(const float* __restrict__ w, float* __restrict__ U, int M, int C,
                                                             int64_t sm, int64_t sc, int flip, float alpha,
                                                             const float* __restrict__ rs_m, const float* __restrict__ rs_c) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)M * C) return;
    const int m = (int)(i / C), c = (int)(i - (int64_t)m * C);
    const float* wp = w + m * sm + c * sc;
    float g[3][3];
#pragma unroll
    for (int t = 0; t < 9; ++t) {           // alpha * w, then the row factor, then the column factor (conv_wprep_kernel's order)
        float v = alpha * wp[flip ? 8 - t : t];
        if (rs_m) v *= rs_m[m];
        if (rs_c) v *= rs_c[c];
        g[t / 3][t % 3] = v;
    }
    float r[4][3];      // G g
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        r[0][k] = g[0][k];
        r[1][k] = 0.5f * ((g[0][k] + g[2][k]) + g[1][k]);
        r[2][k] = 0.5f * ((g[0][k] + g[2][k]) - g[1][k]);
        r[3][k] = g[2][k];
    }
    const int64_t plane = (int64_t)M * C;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const float u0 = r[a][0];
        const float u1 = 0.5f * ((r[a][0] + r[a][2]) + r[a][1]);
        const float u2 = 0.5f * ((r[a][0] + r[a][2]) - r[a][1]);
        const float u3 = r[a][2];
        U[(4 * a + 0) * plane + i] = u0;
        U[(4 * a + 1) * plane + i] = u1;
        U[(4 * a + 2) * plane + i] = u2;
        U[(4 * a + 3) * plane + i] = u3;
    }
}

// x: [planes][H][W], zero padding `pad` (0 valid, 1 same, 2 full: the data gradient of a valid convolution) ->
// V: [16][planes][TH][TW], TH = (H + 2 pad - 2) / 2 tiles of 2x2 outputs.  scale: per-plane factor or null
// (the style modulation of a ModulatedConv2d input, stylegan2_layers.py:280-286, applied on the way).
__global__ __launch_bounds__(kBlock) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                            const float* __restrict__ scale, int64_t planes, int H, int W,
                                                            int pad) {
    const int TH = (H + 2 * pad - 2) >> 1, TW = (W + 2 * pad - 2) >> 1;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * T) return;
    const int64_t pl = i / T;
    const int t = (int)(i - pl * T);
    const int ty = t / TW, tx = t - ty * TW;
    const float* xp = x + pl * H * W;
    const float s = scale ? scale[pl] : 1.0f;
    float d[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - pad + r;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int ix = 2 * tx - pad + q;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float v = xp[in ? (int64_t)iy * W + ix : 0];
            d[r][q] = in ? v * s : 0.0f;
        }
    }
    float e[4][4];      // B^T d
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        e[0][q] = d[0][q] - d[2][q];
        e[1][q] = d[1][q] + d[2][q];
        e[2][q] = d[2][q] - d[1][q];
        e[3][q] = d[1][q] - d[3][q];
    }
    const int64_t plane = planes * T;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        V[(4 * a + 0) * plane + i] = e[a][0] - e[a][2];
        V[(4 * a + 1) * plane + i] = e[a][1] + e[a][2];
        V[(4 * a + 2) * plane + i] = e[a][2] - e[a][1];
        V[(4 * a + 3) * plane + i] = e[a][1] - e[a][3];
    }
}

// Md: [16][planes][TH][TW] -> y: [planes][H][W], times plane_scale[plane] if given; act != 0:
// y = lrelu((Y + noise_w[0] * noise[plane / channels][pixel]) + bias[plane % channels]) * act_scale  (noise, bias optional)
__global__ __launch_bounds__(kBlock) void wino_output_kernel(const float* __restrict__ Md, float* __restrict__ y,
                                                             const float* __restrict__ bias, int64_t planes, int channels,
                                                             int H, int W, int act, float slope, float act_scale,
                                                             const float* __restrict__ plane_scale,
                                                             const float* __restrict__ noise,
                                                             const float* __restrict__ noise_w) {
    const int TH = H >> 1, TW = W >> 1;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * T) return;
    const int64_t pl = i / T;
    const int t = (int)(i - pl * T);
    const int ty = t / TW, tx = t - ty * TW;
    const int64_t plane = planes * T;
    float m[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) m[a][b] = Md[(4 * a + b) * plane + i];
    float r[2][4];      // A^T m
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r[0][b] = (m[0][b] + m[1][b]) + m[2][b];
        r[1][b] = (m[1][b] - m[2][b]) - m[3][b];
    }
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float bv = (act && bias) ? bias[pl % channels] : 0.0f;
    const float ps = plane_scale ? plane_scale[pl] : 1.0f;
    const float nwv = (act && noise) ? noise_w[0] : 0.0f;
    const float* zp = noise ? noise + (pl / channels) * H * W + (int64_t)(2 * ty) * W + 2 * tx : nullptr;
    float* yp = y + pl * H * W + (int64_t)(2 * ty) * W + 2 * tx;
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float o0 = (r[a][0] + r[a][1]) + r[a][2];
        float o1 = (r[a][1] - r[a][2]) - r[a][3];
        if (plane_scale) { o0 *= ps; o1 *= ps; }
        if (act) {
            if (noise) {      // (image + weight * noise) + bias, the reference's association (stylegan2_layers.py:340-351)
                o0 = o0 + nwv * zp[(int64_t)a * W];
                o1 = o1 + nwv * zp[(int64_t)a * W + 1];
            }
            o0 += bv; o1 += bv;
            o0 = ((o0 > 0.0f) ? o0 : o0 * slope) * act_scale;
            o1 = ((o1 > 0.0f) ? o1 : o1 * slope) * act_scale;
        }
        *reinterpret_cast<f32x2*>(yp + (int64_t)a * W) = f32x2{o0, o1};      // 2 tx is even and W is even: 8-byte aligned
    }
}

// ---- weight gradient on the same 16 points:  gw = G^T [ sum_tiles (A e A^T) o (B^T d B) ] G  with e the 2x2 tile of the
// output gradient (the transposition of the forward algorithm: sum_j e_j y_j read as a form in g).
// gy: [planes][H][W] -> E: [16][planes][TH][TW]; A = [1 0; 1 1; 1 -1; 0 -1]
__global__ __launch_bounds__(kBlock) void wino_gy_kernel(const float* __restrict__ gy, float* __restrict__ E,
                                                         const float* __restrict__ scale, int64_t planes, int H, int W) {
    const int TH = H >> 1, TW = W >> 1;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * T) return;
    const int64_t pl = i / T;
    const int t = (int)(i - pl * T);
    const int ty = t / TW, tx = t - ty * TW;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    const float* gp = gy + pl * H * W + (int64_t)(2 * ty) * W + 2 * tx;
    const float s = scale ? scale[pl] : 1.0f;
    const f32x2 e0 = *reinterpret_cast<const f32x2*>(gp) * s;
    const f32x2 e1 = *reinterpret_cast<const f32x2*>(gp + W) * s;
    float r[4][2];      // A e
    r[0][0] = e0[0];         r[0][1] = e0[1];
    r[1][0] = e0[0] + e1[0]; r[1][1] = e0[1] + e1[1];
    r[2][0] = e0[0] - e1[0]; r[2][1] = e0[1] - e1[1];
    r[3][0] = -e1[0];        r[3][1] = -e1[1];
    const int64_t plane = planes * T;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        E[(4 * a + 0) * plane + i] = r[a][0];
        E[(4 * a + 1) * plane + i] = r[a][0] + r[a][1];
        E[(4 * a + 2) * plane + i] = r[a][0] - r[a][1];
        E[(4 * a + 3) * plane + i] = -r[a][1];
    }
}

// gU: [16][M][C] -> gw[m * sm + c * sc + tap] = alpha * (G^T gU G)[tap];  G^T = [1 .5 .5 0; 0 .5 -.5 0; 0 .5 .5 1]
__global__ __launch_bounds__(kBlock) void wino_wgrad_output_kernel(const float* __restrict__ gU, float* __restrict__ gw, int M,
                                                                   int C, int64_t sm, int64_t sc, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)M * C) return;
    const int m = (int)(i / C), c = (int)(i - (int64_t)m * C);
    const int64_t plane = (int64_t)M * C;
    float u[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) u[a][b] = gU[(4 * a + b) * plane + i];
    float r[3][4];      // G^T u
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        r[0][b] = u[0][b] + 0.5f * (u[1][b] + u[2][b]);
        r[1][b] = 0.5f * (u[1][b] - u[2][b]);
        r[2][b] = 0.5f * (u[1][b] + u[2][b]) + u[3][b];
    }
    float* wp = gw + m * sm + c * sc;
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        wp[3 * k + 0] = alpha * (r[k][0] + 0.5f * (r[k][1] + r[k][2]));
        wp[3 * k + 1] = alpha * (0.5f * (r[k][1] - r[k][2]));
        wp[3 * k + 2] = alpha * (0.5f * (r[k][1] + r[k][2]) + r[k][3]);
    }
}

// ---- four tiles per thread (tiles_w a multiple of 4, 16-byte aligned rows): the same arithmetic per tile, 16-byte accesses.
// A wave64 vector-memory instruction costs the CU's address path ~28 cycles whatever its width (HISTORY.md 4.0b): one tile per
// thread is 32 - 34 such instructions per 64 tiles (320 B of HBM traffic each) and bound by their issue near 3.5 TB/s; four
// tiles per thread need 8 per 64 tiles.

// pad 1 only (the window of four tiles starts one column left of an aligned quad; pad 0 / 2 windows start on odd pairs)
__global__ __launch_bounds__(kBlock) void wino_input4_kernel(const float* __restrict__ x, float* __restrict__ V,
                                                             const float* __restrict__ scale, int64_t planes, int H, int W) {
    const int TH = H >> 1, TW = W >> 1, TQ = TW >> 2;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * TH * TQ) return;
    const int64_t pl = i / ((int64_t)TH * TQ);
    const int rem = (int)(i - pl * TH * TQ);
    const int ty = rem / TQ, t4 = rem - ty * TQ;
    const float* xp = x + pl * H * W;
    const float s = scale ? scale[pl] : 1.0f;
    float win[4][16];       // rows 2 ty - 1 ... 2 ty + 2, columns 8 t4 - 4 ... 8 t4 + 11
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = 2 * ty - 1 + r;
        const bool row_in = iy >= 0 && iy < H;
#pragma unroll
        for (int q4 = 0; q4 < 4; ++q4) {
            const int c0 = 8 * t4 - 4 + 4 * q4;
            const bool in = row_in && c0 >= 0 && c0 < W;          // W is a multiple of 4: a quad is inside or outside as a whole
            const f32x4 v = *reinterpret_cast<const f32x4*>(xp + (in ? (int64_t)iy * W + c0 : 0));
#pragma unroll
            for (int e = 0; e < 4; ++e) win[r][4 * q4 + e] = in ? v[e] * s : 0.0f;
        }
    }
    f32x4 out[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {            // tile 4 t4 + j: window columns 3 + 2 j ... 6 + 2 j
        float e[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int c = 3 + 2 * j + q;
            e[0][q] = win[0][c] - win[2][c];
            e[1][q] = win[1][c] + win[2][c];
            e[2][q] = win[2][c] - win[1][c];
            e[3][q] = win[1][c] - win[3][c];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            out[4 * a + 0][j] = e[a][0] - e[a][2];
            out[4 * a + 1][j] = e[a][1] + e[a][2];
            out[4 * a + 2][j] = e[a][2] - e[a][1];
            out[4 * a + 3][j] = e[a][1] - e[a][3];
        }
    }
    const int64_t plane = planes * T;
    const int64_t o = pl * T + (int64_t)ty * TW + 4 * t4;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(V + xi * plane + o) = out[xi];
}

__global__ __launch_bounds__(kBlock) void wino_gy4_kernel(const float* __restrict__ gy, float* __restrict__ E,
                                                          const float* __restrict__ scale, int64_t planes, int H, int W) {
    const int TH = H >> 1, TW = W >> 1, TQ = TW >> 2;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * TH * TQ) return;
    const int64_t pl = i / ((int64_t)TH * TQ);
    const int rem = (int)(i - pl * TH * TQ);
    const int ty = rem / TQ, t4 = rem - ty * TQ;
    const float* gp = gy + pl * H * W + (int64_t)(2 * ty) * W + 8 * t4;
    const float s = scale ? scale[pl] : 1.0f;
    float row[2][8];
#pragma unroll
    for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int h4 = 0; h4 < 2; ++h4) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(gp + (int64_t)r * W + 4 * h4) * s;
#pragma unroll
            for (int e = 0; e < 4; ++e) row[r][4 * h4 + e] = v[e];
        }
    f32x4 out[16];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const float e00 = row[0][2 * j], e01 = row[0][2 * j + 1], e10 = row[1][2 * j], e11 = row[1][2 * j + 1];
        float r[4][2];      // A e
        r[0][0] = e00;       r[0][1] = e01;
        r[1][0] = e00 + e10; r[1][1] = e01 + e11;
        r[2][0] = e00 - e10; r[2][1] = e01 - e11;
        r[3][0] = -e10;      r[3][1] = -e11;
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            out[4 * a + 0][j] = r[a][0];
            out[4 * a + 1][j] = r[a][0] + r[a][1];
            out[4 * a + 2][j] = r[a][0] - r[a][1];
            out[4 * a + 3][j] = -r[a][1];
        }
    }
    const int64_t plane = planes * T;
    const int64_t o = pl * T + (int64_t)ty * TW + 4 * t4;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) *reinterpret_cast<f32x4*>(E + xi * plane + o) = out[xi];
}

__global__ __launch_bounds__(kBlock) void wino_output4_kernel(const float* __restrict__ Md, float* __restrict__ y,
                                                              const float* __restrict__ bias, int64_t planes, int channels,
                                                              int H, int W, int act, float slope, float act_scale,
                                                              const float* __restrict__ plane_scale,
                                                              const float* __restrict__ noise,
                                                              const float* __restrict__ noise_w) {
    const int TH = H >> 1, TW = W >> 1, TQ = TW >> 2;
    const int64_t T = (int64_t)TH * TW;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= planes * TH * TQ) return;
    const int64_t pl = i / ((int64_t)TH * TQ);
    const int rem = (int)(i - pl * TH * TQ);
    const int ty = rem / TQ, t4 = rem - ty * TQ;
    const int64_t plane = planes * T;
    const int64_t o = pl * T + (int64_t)ty * TW + 4 * t4;
    f32x4 mv[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) mv[xi] = *reinterpret_cast<const f32x4*>(Md + xi * plane + o);
    const float bv = (act && bias) ? bias[pl % channels] : 0.0f;
    const float ps = plane_scale ? plane_scale[pl] : 1.0f;
    const float nwv = (act && noise) ? noise_w[0] : 0.0f;
    const float* zp = noise ? noise + (pl / channels) * H * W + (int64_t)(2 * ty) * W + 8 * t4 : nullptr;
    float* yp = y + pl * H * W + (int64_t)(2 * ty) * W + 8 * t4;
    float res[2][8];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        float r[2][4];      // A^T m
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            r[0][b] = (mv[0 + b][j] + mv[4 + b][j]) + mv[8 + b][j];
            r[1][b] = (mv[4 + b][j] - mv[8 + b][j]) - mv[12 + b][j];
        }
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            res[a][2 * j] = (r[a][0] + r[a][1]) + r[a][2];
            res[a][2 * j + 1] = (r[a][1] - r[a][2]) - r[a][3];
        }
    }
#pragma unroll
    for (int a = 0; a < 2; ++a) {
        float z[8];
        if (act && noise) {
#pragma unroll
            for (int h4 = 0; h4 < 2; ++h4) {
                const f32x4 v = *reinterpret_cast<const f32x4*>(zp + (int64_t)a * W + 4 * h4);
#pragma unroll
                for (int e = 0; e < 4; ++e) z[4 * h4 + e] = v[e];
            }
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            float v = res[a][k];
            if (plane_scale) v *= ps;
            if (act) {
                if (noise) v = v + nwv * z[k];
                v += bv;
                v = ((v > 0.0f) ? v : v * slope) * act_scale;
            }
            res[a][k] = v;
        }
        *reinterpret_cast<f32x4*>(yp + (int64_t)a * W) = f32x4{res[a][0], res[a][1], res[a][2], res[a][3]};
        *reinterpret_cast<f32x4*>(yp + (int64_t)a * W + 4) = f32x4{res[a][4], res[a][5], res[a][6], res[a][7]};
    }
}

inline unsigned blocks_for(int64_t work) {
    const int64_t b = ceil_div64(work > 0 ? work : 1, kBlock);
    return (unsigned)(b > 2147483647 ? 2147483647 : b);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* u, int64_t m,
                                    int64_t c, int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha,
                                    sae_stream_t stream) {
    sae::clear_stale_error();
    if (m < 1 || c < 1 || m * c >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_wino_weights_f32: bad shape");
    if (!w || !u) return fail(SAE_EINVAL, "sae_wino_weights_f32: null tensor");
    hipLaunchKernelGGL(wino_weight_kernel, dim3(blocks_for(m * c)), dim3(kBlock), 0, (hipStream_t)stream, w, u, (int)m, (int)c,
                       w_stride_m, w_stride_c, flip ? 1 : 0, alpha, row_scale, col_scale);
    return check_launch("sae_wino_weights_f32");
}

extern "C" int sae_wino_input_f32(const float* x, const float* plane_scale, float* v, int64_t planes, int64_t h, int64_t w,
                                  int32_t pad, sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || pad < 0 || pad > 2 || h < 1 || w < 1 || (h & 1) || (w & 1) || h + 2 * pad < 4 || w + 2 * pad < 4 ||
        h >= 32768 || w >= 32768)
        return fail(SAE_EINVAL, "sae_wino_input_f32: the map must have even sides (2x2 output tiles) and pad 0, 1 or 2, got "
                                "%lld x %lld pad %d", (long long)h, (long long)w, (int)pad);
    if (planes == 0) return SAE_OK;
    if (!x || !v) return fail(SAE_EINVAL, "sae_wino_input_f32: null tensor");
    const int64_t tiles = ((h + 2 * pad - 2) / 2) * ((w + 2 * pad - 2) / 2);
    static const int vec_knob = tuning_knob("SAE_WINO_VEC", 1);
    if (vec_knob && pad == 1 && w % 8 == 0 && aligned16(x) && aligned16(v)) {         // four tiles per thread, 16-byte accesses
        SAE_WINO_TRACE("input4");
        hipLaunchKernelGGL(wino_input4_kernel, dim3(blocks_for(planes * (h / 2) * (w / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                           x, v, plane_scale, planes, (int)h, (int)w);
        return check_launch("sae_wino_input_f32");
    }
    hipLaunchKernelGGL(wino_input_kernel, dim3(blocks_for(planes * tiles)), dim3(kBlock), 0, (hipStream_t)stream, x,
                       v, plane_scale, planes, (int)h, (int)w, (int)pad);
    return check_launch("sae_wino_input_f32");
}

extern "C" int sae_wino_output_f32(const float* md, const float* plane_scale, const float* noise, const float* noise_weight,
                                   const float* bias, float* y, int64_t planes, int64_t channels, int64_t h, int64_t w,
                                   int32_t act, float slope, float act_scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || channels < 1 || h < 2 || w < 2 || (h & 1) || (w & 1) || h >= 32768 || w >= 32768 ||
        channels >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_wino_output_f32: bad shape");
    if (planes == 0) return SAE_OK;
    if (!md || !y) return fail(SAE_EINVAL, "sae_wino_output_f32: null tensor");
    if ((reinterpret_cast<uintptr_t>(y) & 7) != 0) return fail(SAE_EINVAL, "sae_wino_output_f32: y must be 8-byte aligned");
    if (noise && (!act || !noise_weight || planes % channels != 0))
        return fail(SAE_EINVAL, "sae_wino_output_f32: the noise term belongs to the activation epilogue (act != 0, noise_weight, "
                                "planes a multiple of channels)");
    static const int vec_knob = tuning_knob("SAE_WINO_VEC", 1);
    if (vec_knob && w % 8 == 0 && aligned16(md) && aligned16(y) && (!noise || aligned16(noise))) {
        SAE_WINO_TRACE("output4");
        hipLaunchKernelGGL(wino_output4_kernel, dim3(blocks_for(planes * (h / 2) * (w / 8))), dim3(kBlock), 0, (hipStream_t)stream,
                           md, y, bias, planes, (int)channels, (int)h, (int)w, act ? 1 : 0, slope, act_scale, plane_scale, noise,
                           noise_weight);
        return check_launch("sae_wino_output_f32");
    }
    hipLaunchKernelGGL(wino_output_kernel, dim3(blocks_for(planes * (h / 2) * (w / 2))), dim3(kBlock), 0, (hipStream_t)stream,
                       md, y, bias, planes, (int)channels, (int)h, (int)w, act ? 1 : 0, slope, act_scale, plane_scale, noise,
                       noise_weight);
    return check_launch("sae_wino_output_f32");
}

extern "C" int sae_wino_gy_f32(const float* gy, const float* plane_scale, float* e, int64_t planes, int64_t h, int64_t w,
                               sae_stream_t stream) {
    sae::clear_stale_error();
    if (planes < 0 || h < 2 || w < 2 || (h & 1) || (w & 1) || h >= 32768 || w >= 32768)
        return fail(SAE_EINVAL, "sae_wino_gy_f32: the map must have even sides (2x2 output tiles), got %lld x %lld", (long long)h,
                    (long long)w);
    if (planes == 0) return SAE_OK;
    if (!gy || !e) return fail(SAE_EINVAL, "sae_wino_gy_f32: null tensor");
    if ((reinterpret_cast<uintptr_t>(gy) & 7) != 0) return fail(SAE_EINVAL, "sae_wino_gy_f32: gy must be 8-byte aligned");
    static const int vec_knob = tuning_knob("SAE_WINO_VEC", 1);
    if (vec_knob && w % 8 == 0 && aligned16(gy) && aligned16(e)) {
        SAE_WINO_TRACE("gy4");
        hipLaunchKernelGGL(wino_gy4_kernel, dim3(blocks_for(planes * (h / 2) * (w / 8))), dim3(kBlock), 0, (hipStream_t)stream, gy, e,
                           plane_scale, planes, (int)h, (int)w);
        return check_launch("sae_wino_gy_f32");
    }
    hipLaunchKernelGGL(wino_gy_kernel, dim3(blocks_for(planes * (h / 2) * (w / 2))), dim3(kBlock), 0, (hipStream_t)stream, gy, e,
                       plane_scale, planes, (int)h, (int)w);
    return check_launch("sae_wino_gy_f32");
}

extern "C" int sae_wino_wgrad_output_f32(const float* gu, float* gw, int64_t m, int64_t c, int64_t w_stride_m, int64_t w_stride_c,
                                         float alpha, sae_stream_t stream) {
    sae::clear_stale_error();
    if (m < 1 || c < 1 || m * c >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_wino_wgrad_output_f32: bad shape");
    if (!gu || !gw) return fail(SAE_EINVAL, "sae_wino_wgrad_output_f32: null tensor");
    hipLaunchKernelGGL(wino_wgrad_output_kernel, dim3(blocks_for(m * c)), dim3(kBlock), 0, (hipStream_t)stream, gu, gw, (int)m,
                       (int)c, w_stride_m, w_stride_c, alpha);
    return check_launch("sae_wino_wgrad_output_f32");
}
