// Fused Winograd F(2x2, 3x3) convolution for the 3x3 stride-1 layers: input transform, the sixteen transform-domain products and
// the output transform (+ the layer's epilogue) in ONE kernel -- nothing but x, the transformed weights and y touches HBM.
//
// The reference runs these layers through F.conv2d (models/networks/stylegan2_layers.py:136,315); the unfused route of
// winograd.hip (sae_wino_input / gemm / output) pays for its 2.25x fewer multiplications with two passes over 4x the activation,
// which is why it only wins from 256 channels up (profiles/r5_winograd_first_session.txt).  Here a workgroup owns
// 64 output channels x 64 output tiles (2x2 pixels each) for ALL sixteen points of the transform domain:
//
//     acc[xi][m][t] += sum_{c in chunk} U[xi][m][c] * V[xi][c][t]          16 x (32x32x2 fp32 MFMA) per k-pair and wave
//     V[xi][c][t] = (B^T d B)[xi] of the 4x4 patch d of tile t, channel c    formed in registers from x, written to LDS
//     U[xi][m][c] = alpha (G g G^T)[xi]                                      prepared once per weight update (sae_wino_fused_weights_f32)
//     y[2x2 of t] = A^T acc[.][m][t] A (+ noise, bias, leaky ReLU)           lane-local: a lane's sixteen accumulators of one (m, t)
//
// Four waves = 2 (halves of the 64 channels) x 2 (halves of the 64 tiles); each wave keeps 16 points x 16 registers = 256
// accumulators (the AGPR half of the 512-entry file: one wave per SIMD) and per chunk of 8 input channels issues 64 MFMAs against
// 48 LDS reads (U as one 16-byte read per point, V as two 8-byte reads), 16 LDS writes and 16 global 16-byte loads.
// LDS: two stages x (U 32 KB + V 32 KB) = 128 KB; one barrier per chunk.
//
// Data gradient of the same layers: the same kernel on the output gradient with the flipped / transposed filter (flip = 1 in the
// weight preparation) and padding 2 - pad.
#include "sae_common.h"

#include <type_traits>

namespace sae {
namespace {

constexpr int kWfM = 64;       // output channels per workgroup
constexpr int kWfT = 64;       // 2x2 output tiles per workgroup
constexpr int kWfCK = 8;       // input channels per chunk
constexpr int kWfStage = 16 * 2 * 64 * 4;     // floats of one operand stage: [xi][half][row][4]

struct WinoFusedParams {
    int N, C, H, W;            // input [N][C][H][W]
    int M, OH, OW;             // output [N][M][OH][OW], OH = H + 2 pad - 2
    int pad;
    int TH, TW;                // tiles per image
    int bw_log2, bh_log2;      // tile block = BN x BH x BW, BN * BH * BW = 64
    int blocks_x, blocks_y;    // tile blocks per image row / column
    int chunks;                // ceil(C / 8)
    const float* x_scale;      // [N * C] or null: the style modulation of the input
    const float* out_scale;    // [N * M] or null
    const float* noise;        // [N][OH][OW] or null
    const float* noise_w;      // [1]
    const float* bias;         // [M] or null
    int act;
    float slope, act_scale;
};

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned 16-byte global access

// Uf[mb][chunk][xi][half][ml][s] = U[xi][m = 64 mb + ml][c = 8 chunk + 4 half + s], zero beyond M / C: one chunk of one channel
// block is 32 KB contiguous in exactly the order the kernel keeps it in LDS.
__global__ __launch_bounds__(kBlock) void wino_fused_wprep_kernel(const float* __restrict__ w, float* __restrict__ Uf, int M, int C,
                                                                  int chunks, int64_t sm, int64_t sc, int flip, float alpha,
                                                                  const float* __restrict__ rs_m, const float* __restrict__ rs_c) {
    const int Cp = chunks * kWfCK;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t Mp = (int64_t)((M + kWfM - 1) / kWfM) * kWfM;
    if (i >= Mp * Cp) return;
    // consecutive threads: s fastest, then ml, then half, then chunk, then mb -> 16-byte runs per point
    const int s = (int)(i & 3);
    const int ml = (int)((i >> 2) & 63);
    const int hf = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int chunk = (int)(rest % chunks), mb = (int)(rest / chunks);
    const int m = mb * kWfM + ml, c = chunk * kWfCK + 4 * hf + s;
    float u[16];
    if (m < M && c < C) {
        const float* wp = w + m * sm + c * sc;
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {           // alpha * w, then the row factor, then the column factor (wino_weight_kernel's order)
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
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            u[4 * a + 0] = r[a][0];
            u[4 * a + 1] = 0.5f * ((r[a][0] + r[a][2]) + r[a][1]);
            u[4 * a + 2] = 0.5f * ((r[a][0] + r[a][2]) - r[a][1]);
            u[4 * a + 3] = r[a][2];
        }
    } else {
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) u[xi] = 0.0f;
    }
    float* dst = Uf + ((int64_t)mb * chunks + chunk) * kWfStage + (hf * 64 + ml) * 4 + s;
#pragma unroll
    for (int xi = 0; xi < 16; ++xi) dst[xi * 512] = u[xi];
}

template <bool ACT, bool PAD2, bool XS>
__global__ __launch_bounds__(kBlock, 1) void wino_fused_kernel(const float* __restrict__ x, const float* __restrict__ Uf,
                                                               float* __restrict__ y, const WinoFusedParams p) {
    __shared__ float Us[2][kWfStage];      // [xi][half][m][s]: channel 4 half + s of the chunk
    __shared__ float Vs[2][kWfStage];      // [xi][half][s >> 1][t][s & 1]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 1, wt = wid & 1;

    const int BW = 1 << p.bw_log2, BH = 1 << p.bh_log2;
    const int bshift = p.bw_log2 + p.bh_log2;
    const int BN = kWfT >> bshift;
    int b = blockIdx.x;
    const int bx = b % p.blocks_x;
    b /= p.blocks_x;
    const int by = b % p.blocks_y;
    const int bn = b / p.blocks_y;
    const int mb = blockIdx.y;

    // ---- the thread's role in staging: tile `lane` of the block, channels 2 wid and 2 wid + 1 of a chunk.  Branch-free: every
    // row of the 4x4 patch is ONE 16-byte load (4-byte aligned) from a window clamped into the row, shifted into place and
    // zeroed outside the image by selects, so that the loop body is one basic block the scheduler can interleave with the MFMAs.
    const int s_tx = bx * BW + (lane & (BW - 1));
    const int s_ty = by * BH + ((lane >> p.bw_log2) & (BH - 1));
    const int s_n = bn * BN + (lane >> bshift);
    const bool s_valid = s_tx < p.TW && s_ty < p.TH && s_n < p.N;
    const int iy0 = 2 * s_ty - p.pad, ix0 = 2 * s_tx - p.pad;
    int cx = ix0 < 0 ? 0 : ix0;
    if (cx > p.W - 4) cx = p.W - 4;
    const int shift = s_valid ? ix0 - cx : 0;          // -2 .. 2: wanted element q is loaded element q + shift
    const int64_t HW = (int64_t)p.H * p.W;
    const float* xt = x + ((int64_t)(s_valid ? s_n : 0) * p.C) * HW + (s_valid ? cx : 0);
    int rowoff[4];
    bool rowok[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = iy0 + r;
        rowok[r] = s_valid && iy >= 0 && iy < p.H;
        rowoff[r] = (rowok[r] ? iy : 0) * p.W;
    }

    f32x4 dreg[2][4];          // the two channels' 4x4 patches of the NEXT chunk (raw windows, then the patches themselves)
    f32x4 ureg[8];             // this thread's 8 quads of the next chunk's weights
    f32x2 vreg[16];            // B^T d B of both channels, as written to LDS
    float xsc[2] = {1.0f, 1.0f};

    // ---- the staging of a chunk, in pieces the main loop spreads over the sixteen MFMA groups of the previous chunk
    auto load_x = [&](int chunk, int c2) {                     // 4 global 16-byte loads
        int ch = chunk * kWfCK + 2 * wid + c2;                 // wave-uniform
        if (ch > p.C - 1) ch = p.C - 1;                        // (a channel beyond C: its patch is zeroed in `window`)
        const float* xp = xt + (int64_t)ch * HW;
#pragma unroll
        for (int r = 0; r < 4; ++r) dreg[c2][r] = *reinterpret_cast<const f32x4u*>(xp + rowoff[r]);
        if (XS) xsc[c2] = p.x_scale[(int64_t)(s_valid ? s_n : 0) * p.C + ch];
    };
    auto load_u = [&](int chunk, int lo) {                     // 4 global 16-byte loads
        const f32x4* up = reinterpret_cast<const f32x4*>(Uf + ((int64_t)mb * p.chunks + chunk) * kWfStage) + tid;
#pragma unroll
        for (int j = lo; j < lo + 4; ++j) ureg[j] = up[j * kBlock];
    };
    auto window = [&](int chunk, int c2, int r) {              // row r of the patch out of the loaded window: ~10 VALU
        const bool ok = rowok[r] && (chunk * kWfCK + 2 * wid + c2 < p.C);
        const f32x4 l = dreg[c2][r];
        f32x4 w;
        if (PAD2) {                // windows of the data gradient of a valid layer: shift -2 .. 2 (sequential selects)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                float t = 0.0f;
#pragma unroll
                for (int k = -2; k <= 2; ++k)
                    if (q + k >= 0 && q + k < 4) t = (shift == k) ? l[q + k] : t;
                w[q] = t;
            }
        } else {                   // shift -1 .. 1
            w[0] = shift == 0 ? l[0] : shift > 0 ? l[1] : 0.0f;
            w[1] = shift == 0 ? l[1] : shift > 0 ? l[2] : l[0];
            w[2] = shift == 0 ? l[2] : shift > 0 ? l[3] : l[1];
            w[3] = shift == 0 ? l[3] : shift > 0 ? 0.0f : l[2];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) w[q] = ok ? w[q] : 0.0f;
        dreg[c2][r] = XS ? w * xsc[c2] : w;
    };
    auto transform = [&](int c2) {                             // B^T d B of one channel: 32 VALU
        f32x4 e[4];
        e[0] = dreg[c2][0] - dreg[c2][2];
        e[1] = dreg[c2][1] + dreg[c2][2];
        e[2] = dreg[c2][2] - dreg[c2][1];
        e[3] = dreg[c2][1] - dreg[c2][3];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vreg[4 * a + 0][c2] = e[a][0] - e[a][2];
            vreg[4 * a + 1][c2] = e[a][1] + e[a][2];
            vreg[4 * a + 2][c2] = e[a][2] - e[a][1];
            vreg[4 * a + 3][c2] = e[a][1] - e[a][3];
        }
    };
    // channel 2 wid + c2 of the chunk = (half = wid >> 1, s = 2 (wid & 1) + c2): both channels share one 8-byte slot
    auto write_v = [&](int buf, int lo) {                      // 8 LDS 8-byte writes
        f32x2* vd = reinterpret_cast<f32x2*>(Vs[buf]) + (((wid >> 1) * 2 + (wid & 1)) * 64 + lane);
#pragma unroll
        for (int xi = lo; xi < lo + 8; ++xi) vd[xi * 256] = vreg[xi];
    };
    auto write_u = [&](int buf, int lo) {                      // 4 LDS 16-byte writes
        f32x4* ud = reinterpret_cast<f32x4*>(Us[buf]) + tid;
#pragma unroll
        for (int j = lo; j < lo + 4; ++j) ud[j * kBlock] = ureg[j];
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.0f;

    // One pass over the staged chunk `cur`: sixteen groups of four MFMAs (one point each); the operands of point xi + 1 are
    // read while the MFMAs of point xi run, and with STAGE the pieces of the next chunk's staging ride in the groups' shadows --
    // x first (its windows are needed by group 6), the weights last (written by groups 14, 15).  sched_barrier keeps the
    // pieces in their groups; inside a group the scheduler is free.
    auto pass = [&](auto stage_tag, int cur, int chunk) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        const f32x4* ua = reinterpret_cast<const f32x4*>(Us[cur]) + (half * 64 + wm * 32 + l31);
        const f32x2* vb = reinterpret_cast<const f32x2*>(Vs[cur]) + (half * 128 + wt * 32 + l31);
        f32x4 a = ua[0];
        f32x2 b01 = vb[0], b23 = vb[64];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            f32x4 an = a;
            f32x2 b01n = b01, b23n = b23;
            if (xi < 15) {
                an = ua[(xi + 1) * 128];
                b01n = vb[(xi + 1) * 256];
                b23n = vb[(xi + 1) * 256 + 64];
            }
            if (STAGE) {
                if (xi == 0) load_x(chunk + 1, 0);
                if (xi == 1) load_x(chunk + 1, 1);
                if (xi == 2) load_u(chunk + 1, 0);
                if (xi == 3) load_u(chunk + 1, 4);
            }
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b01[0], acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[1], b01[1], acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[2], b23[0], acc[xi], 0, 0, 0);
            acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[3], b23[1], acc[xi], 0, 0, 0);
            if (STAGE) {
                if (xi >= 6 && xi <= 9) {
                    window(chunk + 1, (xi - 6) >> 1, 2 * ((xi - 6) & 1));
                    window(chunk + 1, (xi - 6) >> 1, 2 * ((xi - 6) & 1) + 1);
                }
                if (xi == 10) transform(0);
                if (xi == 11) transform(1);
                if (xi == 12) write_v(cur ^ 1, 0);
                if (xi == 13) write_v(cur ^ 1, 8);
                if (xi == 14) write_u(cur ^ 1, 0);
                if (xi == 15) write_u(cur ^ 1, 4);
            }
            a = an;
            b01 = b01n;
            b23 = b23n;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    // prologue: chunk 0
    load_x(0, 0);
    load_x(0, 1);
    load_u(0, 0);
    load_u(0, 4);
#pragma unroll
    for (int c2 = 0; c2 < 2; ++c2) {
#pragma unroll
        for (int r = 0; r < 4; ++r) window(0, c2, r);
        transform(c2);
    }
    write_v(0, 0);
    write_v(0, 8);
    write_u(0, 0);
    write_u(0, 4);
    __syncthreads();

    int cur = 0;
    for (int chunk = 0; chunk + 1 < p.chunks; ++chunk) {
        pass(std::true_type{}, cur, chunk);
        __syncthreads();
        cur ^= 1;
    }
    pass(std::false_type{}, cur, 0);

    // ---- output transform, lane-local: acc[4 a + b][r] of (m = ... r ..., tile = wt * 32 + l31)
    const int ot = wt * 32 + l31;
    const int o_tx = bx * BW + (ot & (BW - 1));
    const int o_ty = by * BH + ((ot >> p.bw_log2) & (BH - 1));
    const int o_n = bn * BN + (ot >> bshift);
    if (!(o_tx < p.TW && o_ty < p.TH && o_n < p.N)) return;
    const int64_t OHW = (int64_t)p.OH * p.OW;
    const int64_t pix = (int64_t)(2 * o_ty) * p.OW + 2 * o_tx;
    float nz[2][2] = {{0.0f, 0.0f}, {0.0f, 0.0f}};
    if (ACT && p.noise) {
        const float nw = p.noise_w[0];
        const float* zp = p.noise + (int64_t)o_n * OHW + pix;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            nz[a][0] = nw * zp[(int64_t)a * p.OW];
            nz[a][1] = nw * zp[(int64_t)a * p.OW + 1];
        }
    }
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb * kWfM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (m >= p.M) continue;
        float t[2][4];      // A^T acc
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            t[0][bq] = (acc[bq][r] + acc[4 + bq][r]) + acc[8 + bq][r];
            t[1][bq] = (acc[4 + bq][r] - acc[8 + bq][r]) - acc[12 + bq][r];
        }
        const float ps = p.out_scale ? p.out_scale[(int64_t)o_n * p.M + m] : 1.0f;
        const float bv = (ACT && p.bias) ? p.bias[m] : 0.0f;
        float* yp = y + ((int64_t)o_n * p.M + m) * OHW + pix;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            float o0 = (t[a][0] + t[a][1]) + t[a][2];
            float o1 = (t[a][1] - t[a][2]) - t[a][3];
            if (p.out_scale) { o0 *= ps; o1 *= ps; }
            if (ACT) {
                if (p.noise) { o0 = o0 + nz[a][0]; o1 = o1 + nz[a][1]; }       // (image + weight * noise) + bias
                o0 += bv; o1 += bv;
                o0 = ((o0 > 0.0f) ? o0 : o0 * p.slope) * p.act_scale;
                o1 = ((o1 > 0.0f) ? o1 : o1 * p.slope) * p.act_scale;
            }
            *reinterpret_cast<f32x2*>(yp + (int64_t)a * p.OW) = f32x2{o0, o1};      // 2 tx and OW are even: 8-byte aligned
        }
    }
}

// ------------------------------------------------------------------------------------------------------------------------------
// Weight gradient of the same layers in ONE kernel (+ a fixed-order reduction over pixel slices):
//     gU[xi][m][c] = sum over images and tiles of (A e A^T)[xi] * (B^T d B)[xi]        e: 2x2 tile of gy[m], d: 4x4 patch of x[c]
//     gw[m][c][3x3] = alpha G^T gU G
// A workgroup owns 64 gradient channels (m) x 64 input channels (c) for all sixteen points and walks a slice of the pixels in
// chunks of 8 tiles (8 consecutive tiles of one tile row: tiles_w % 8 == 0); the contraction index of the MFMAs is the tile.
// Both operands are transformed in registers on the way from HBM to LDS: thread (channel, tile pair) forms A e A^T of two tiles
// of its gy channel (two 16-byte loads) and B^T d B of two tiles of its x channel (a 4 x 6 window: eight loads), 32 8-byte LDS
// writes; the MFMA side reads both operands as one 16-byte read per point.  G^T gU G is lane-local at the end (a lane holds the
// sixteen points of its (m, c) pairs); slices are summed by wino_fused_wgrad_reduce_kernel in slice order (deterministic).
struct WinoWgradParams {
    int N, C, H, W;            // x  [N][C][H][W]
    int M, OH, OW;             // gy [N][M][OH][OW]
    int pad;
    int TH, TW;                // tiles per image; TW % 8 == 0
    int cpr;                   // chunks per tile row = TW / 8
    int chunks, chunks_per_slice;
    int Mp, Cp;                // slab dims (M, C padded to 64)
    const float* x_scale;      // [N * C] or null
    const float* y_scale;      // [N * M] or null
};

typedef float f32x2u __attribute__((ext_vector_type(2), aligned(4)));

template <bool MOD>
__global__ __launch_bounds__(kBlock, 1) void wino_fused_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                     float* __restrict__ slab, const WinoWgradParams p) {
    __shared__ float Es[2][kWfStage];      // [xi][half][m][s]: tile 4 half + s of the chunk
    __shared__ float Vs[2][kWfStage];      // [xi][half][c][s]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 1, wt = wid & 1;
    const int cb = blockIdx.x, mb = blockIdx.y, slice = blockIdx.z;

    // ---- staging role: channel `ch` of both 64-channel blocks, tiles 4 hf + 2 q1 and + 1 of a chunk (hf wave-uniform)
    const int q1 = lane & 1;
    const int ch = (wid & 1) * 32 + (lane >> 1);
    const int hf = wid >> 1;
    const int kk0 = 4 * hf + 2 * q1;
    const bool m_ok = mb * kWfM + ch < p.M, c_ok = cb * kWfM + ch < p.C;
    const int m_ch = m_ok ? mb * kWfM + ch : p.M - 1;
    const int c_ch = c_ok ? cb * kWfM + ch : p.C - 1;
    const int64_t HW = (int64_t)p.H * p.W, OHW = (int64_t)p.OH * p.OW;
    const float* px = (MOD && p.x_scale) ? p.x_scale : x;          // (a valid address either way: the loop body stays branch-free)
    const float* py = (MOD && p.y_scale) ? p.y_scale : gy;

    const int ch_begin = slice * p.chunks_per_slice;
    int ch_end = ch_begin + p.chunks_per_slice;
    if (ch_end > p.chunks) ch_end = p.chunks;
    // position of the chunk being LOADED: image n, tile row ty, chunk txb of the row
    int ld_txb = ch_begin % p.cpr;
    int ld_ty = (ch_begin / p.cpr) % p.TH;
    int ld_n = ch_begin / (p.cpr * p.TH);

    f32x4 greg[2];             // gy rows 2 ty, 2 ty + 1: two tiles x two columns
    f32x4 xr4[4];              // window rows: columns 0 .. 3
    f32x2 xr2[4];              //              columns 4, 5
    int rowmask = 0, shift = 0;
    float sx = 1.0f, sy = 1.0f;
    f32x2 ev[16], vv[16];      // A e A^T and B^T d B of the two tiles, as written to LDS
    float d6[4][6];            // the 4 x 6 patch of the two tiles

    auto load_gy = [&]() {
        const int tx = ld_txb * 8 + kk0;
        const float* gp = gy + ((int64_t)ld_n * p.M + m_ch) * OHW + (int64_t)(2 * ld_ty) * p.OW + 2 * tx;
        greg[0] = *reinterpret_cast<const f32x4*>(gp);
        greg[1] = *reinterpret_cast<const f32x4*>(gp + p.OW);
        if (MOD) {
            sx = px[(int64_t)ld_n * p.C + c_ch];
            sy = py[(int64_t)ld_n * p.M + m_ch];
            sx = p.x_scale ? sx : 1.0f;
            sy = p.y_scale ? sy : 1.0f;
        }
    };
    auto load_x = [&](int r0) {            // window rows r0, r0 + 1
        const int tx = ld_txb * 8 + kk0;
        const int ix0 = 2 * tx - p.pad;
        int cx = ix0 < 0 ? 0 : ix0;
        if (cx > p.W - 6) cx = p.W - 6;
        if (r0 == 0) { shift = ix0 - cx; rowmask = 0; }
        const float* xp = x + ((int64_t)ld_n * p.C + c_ch) * HW + cx;
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) {
            const int iy = 2 * ld_ty - p.pad + r;
            const bool ok = iy >= 0 && iy < p.H;
            rowmask |= ok ? (1 << r) : 0;
            const float* rp = xp + (int64_t)(ok ? iy : 0) * p.W;
            xr4[r] = *reinterpret_cast<const f32x4u*>(rp);
            xr2[r] = *reinterpret_cast<const f32x2u*>(rp + 4);
        }
    };
    auto advance = [&]() {
        ++ld_txb;
        if (ld_txb == p.cpr) {
            ld_txb = 0;
            ++ld_ty;
            if (ld_ty == p.TH) { ld_ty = 0; ++ld_n; }
        }
    };
    auto transform_e = [&]() {             // A e A^T of both tiles
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float e00 = greg[0][2 * t], e01 = greg[0][2 * t + 1], e10 = greg[1][2 * t], e11 = greg[1][2 * t + 1];
            if (MOD) { e00 *= sy; e01 *= sy; e10 *= sy; e11 *= sy; }
            if (!m_ok) { e00 = 0.0f; e01 = 0.0f; e10 = 0.0f; e11 = 0.0f; }
            float r[4][2];
            r[0][0] = e00;       r[0][1] = e01;
            r[1][0] = e00 + e10; r[1][1] = e01 + e11;
            r[2][0] = e00 - e10; r[2][1] = e01 - e11;
            r[3][0] = -e10;      r[3][1] = -e11;
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                ev[4 * a + 0][t] = r[a][0];
                ev[4 * a + 1][t] = r[a][0] + r[a][1];
                ev[4 * a + 2][t] = r[a][0] - r[a][1];
                ev[4 * a + 3][t] = -r[a][1];
            }
        }
    };
    auto window = [&](int r) {             // row r of the 4 x 6 patch out of the loaded window (shift -1 .. 1)
        const bool ok = c_ok && ((rowmask >> r) & 1);
        float l[6] = {xr4[r][0], xr4[r][1], xr4[r][2], xr4[r][3], xr2[r][0], xr2[r][1]};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            const float lo = q > 0 ? l[q - 1] : 0.0f, hi = q < 5 ? l[q + 1] : 0.0f;
            float v = shift == 0 ? l[q] : shift > 0 ? hi : lo;
            v = ok ? v : 0.0f;
            d6[r][q] = MOD ? v * sx : v;
        }
    };
    auto transform_v = [&](int t) {        // B^T d B of tile t (columns 2 t .. 2 t + 3 of the patch)
        float e[4][4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            e[0][q] = d6[0][2 * t + q] - d6[2][2 * t + q];
            e[1][q] = d6[1][2 * t + q] + d6[2][2 * t + q];
            e[2][q] = d6[2][2 * t + q] - d6[1][2 * t + q];
            e[3][q] = d6[1][2 * t + q] - d6[3][2 * t + q];
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vv[4 * a + 0][t] = e[a][0] - e[a][2];
            vv[4 * a + 1][t] = e[a][1] + e[a][2];
            vv[4 * a + 2][t] = e[a][2] - e[a][1];
            vv[4 * a + 3][t] = e[a][1] - e[a][3];
        }
    };
    auto write_e = [&](int buf, int lo) {
        f32x2* d = reinterpret_cast<f32x2*>(Es[buf]) + ((hf * 64 + ch) * 2 + q1);
#pragma unroll
        for (int xi = lo; xi < lo + 8; ++xi) d[xi * 256] = ev[xi];
    };
    auto write_v = [&](int buf, int lo) {
        f32x2* d = reinterpret_cast<f32x2*>(Vs[buf]) + ((hf * 64 + ch) * 2 + q1);
#pragma unroll
        for (int xi = lo; xi < lo + 8; ++xi) d[xi * 256] = vv[xi];
    };

    f32x16 acc[16];
#pragma unroll
    for (int xi = 0; xi < 16; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.0f;

    auto pass = [&](auto stage_tag, int cur) {
        constexpr bool STAGE = decltype(stage_tag)::value;
        const f32x4* ea = reinterpret_cast<const f32x4*>(Es[cur]) + (half * 64 + wm * 32 + l31);
        const f32x4* vb = reinterpret_cast<const f32x4*>(Vs[cur]) + (half * 64 + wt * 32 + l31);
        f32x4 a = ea[0], b = vb[0];
#pragma unroll
        for (int xi = 0; xi < 16; ++xi) {
            f32x4 an = a, bn = b;
            if (xi < 15) {
                an = ea[(xi + 1) * 128];
                bn = vb[(xi + 1) * 128];
            }
            if (STAGE) {
                if (xi == 0) load_gy();
                if (xi == 1) load_x(0);
                if (xi == 2) { load_x(2); advance(); }
            }
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], b[s4], acc[xi], 0, 0, 0);
            if (STAGE) {
                if (xi == 5) transform_e();
                if (xi >= 6 && xi <= 9) window(xi - 6);
                if (xi == 10) transform_v(0);
                if (xi == 11) transform_v(1);
                if (xi == 12) write_e(cur ^ 1, 0);
                if (xi == 13) write_e(cur ^ 1, 8);
                if (xi == 14) write_v(cur ^ 1, 0);
                if (xi == 15) write_v(cur ^ 1, 8);
            }
            a = an;
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    if (ch_begin < ch_end) {
        load_gy();
        load_x(0);
        load_x(2);
        advance();
        transform_e();
#pragma unroll
        for (int r = 0; r < 4; ++r) window(r);
        transform_v(0);
        transform_v(1);
        write_e(0, 0);
        write_e(0, 8);
        write_v(0, 0);
        write_v(0, 8);
        __syncthreads();
        int cur = 0;
        for (int chunk = ch_begin; chunk + 1 < ch_end; ++chunk) {
            pass(std::true_type{}, cur);
            __syncthreads();
            cur ^= 1;
        }
        pass(std::false_type{}, cur);
    }

    // ---- G^T gU G, lane-local: acc[4 a + b][r] of (m = wm * 32 + row(r, half), c = wt * 32 + l31) -> slab[slice][m][tap][c]
    const int c = cb * kWfM + wt * 32 + l31;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int m = mb * kWfM + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        float t[3][4];      // G^T u
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            t[0][bq] = acc[bq][r] + 0.5f * (acc[4 + bq][r] + acc[8 + bq][r]);
            t[1][bq] = 0.5f * (acc[4 + bq][r] - acc[8 + bq][r]);
            t[2][bq] = 0.5f * (acc[4 + bq][r] + acc[8 + bq][r]) + acc[12 + bq][r];
        }
        float* sp = slab + (((int64_t)slice * p.Mp + m) * 9) * p.Cp + c;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            sp[(int64_t)(3 * k + 0) * p.Cp] = t[k][0] + 0.5f * (t[k][1] + t[k][2]);
            sp[(int64_t)(3 * k + 1) * p.Cp] = 0.5f * (t[k][1] - t[k][2]);
            sp[(int64_t)(3 * k + 2) * p.Cp] = 0.5f * (t[k][1] + t[k][2]) + t[k][3];
        }
    }
}

// gw[m * sm + c * sc + tap] = alpha * sum over slices (in slice order) of slab[slice][m][tap][c]
__global__ __launch_bounds__(kBlock) void wino_fused_wgrad_reduce_kernel(const float* __restrict__ slab, float* __restrict__ gw,
                                                                         int M, int C, int Mp, int Cp, int slices, int64_t sm,
                                                                         int64_t sc, float alpha) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)M * 9 * C) return;
    const int c = (int)(i % C);
    const int tap = (int)((i / C) % 9);
    const int m = (int)(i / ((int64_t)C * 9));
    const float* sp = slab + ((int64_t)m * 9 + tap) * Cp + c;
    const int64_t stride = (int64_t)Mp * 9 * Cp;
    float acc = 0.0f;
    for (int s = 0; s < slices; ++s) acc += sp[s * stride];
    gw[m * sm + c * sc + tap] = alpha * acc;
}

// pixel slices of the fused weight gradient: workgroups = 64 x 64 channel blocks x slices in whole rounds of 256 where the
// pixel count allows (a slice needs at least 8 chunks to pay for its epilogue)
inline int wgrad_slices(int64_t mbs, int64_t cbs, int64_t chunks) {
    const int64_t blocks = mbs * cbs;
    int64_t unit = 256;
    for (int64_t g = blocks; (g & 1) == 0 && unit > 1; g >>= 1) unit >>= 1;      // 256 / gcd(256, blocks)
    int64_t slices = (1536 + blocks - 1) / blocks;
    slices = (slices + unit - 1) / unit * unit;
    if (slices > 256) slices = 256;
    const int64_t cap = chunks / 8 > 0 ? chunks / 8 : 1;
    if (slices > cap) slices = cap;
    return (int)slices;
}

inline int64_t fused_weight_floats(int64_t m, int64_t c) {
    return ceil_div64(m, kWfM) * ceil_div64(c, kWfCK) * kWfStage;
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int64_t sae_wino_fused_weights_floats(int64_t m, int64_t c) {
    if (m < 1 || c < 1) return 0;
    return fused_weight_floats(m, c);
}

extern "C" int sae_wino_fused_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t m,
                                          int64_t c, int64_t w_stride_m, int64_t w_stride_c, int32_t flip, float alpha,
                                          sae_stream_t stream) {
    sae::clear_stale_error();
    if (m < 1 || c < 1 || fused_weight_floats(m, c) >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_wino_fused_weights_f32: bad shape");
    if (!w || !uf) return fail(SAE_EINVAL, "sae_wino_fused_weights_f32: null tensor");
    if (!aligned16(uf)) return fail(SAE_EINVAL, "sae_wino_fused_weights_f32: uf must be 16-byte aligned");
    const int chunks = (int)ceil_div64(c, kWfCK);
    const int64_t work = ceil_div64(m, kWfM) * kWfM * chunks * kWfCK;
    hipLaunchKernelGGL(wino_fused_wprep_kernel, dim3((unsigned)ceil_div64(work, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, w,
                       uf, (int)m, (int)c, chunks, w_stride_m, w_stride_c, flip ? 1 : 0, alpha, row_scale, col_scale);
    return check_launch("sae_wino_fused_weights_f32");
}

extern "C" int sae_wino_fused_conv_f32(const float* x, const float* x_scale, const float* uf, const float* out_scale,
                                       const float* noise, const float* noise_weight, const float* bias, float* y, int64_t n,
                                       int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad, int32_t act, float slope,
                                       float act_scale, sae_stream_t stream) {
    sae::clear_stale_error();
    if (n < 0 || c < 1 || m < 1 || pad < 0 || pad > 2 || h < 1 || w < 1 || (h & 1) || (w & 1) || h + 2 * pad < 4 || w + 2 * pad < 4 ||
        h >= 32768 || w >= 32768 || c >= (1 << 24) || m >= (1 << 24) || n >= (1 << 24))
        return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: the map must have even sides (2x2 output tiles) and pad 0, 1 or 2, got "
                                "%lld x %lld pad %d", (long long)h, (long long)w, (int)pad);
    if (n == 0) return SAE_OK;
    if (!x || !uf || !y) return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: null tensor");
    if (w < 4) return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: rows of at least 4 floats (the patch rows are 16-byte loads), got %lld", (long long)w);
    if (!aligned16(uf) || (reinterpret_cast<uintptr_t>(y) & 7) != 0)
        return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: uf must be 16-byte and y 8-byte aligned");
    if (noise && (!act || !noise_weight))
        return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: the noise term belongs to the activation epilogue (act != 0, noise_weight)");
    WinoFusedParams p;
    p.N = (int)n; p.C = (int)c; p.H = (int)h; p.W = (int)w; p.M = (int)m;
    p.pad = pad;
    p.OH = (int)h + 2 * pad - 2; p.OW = (int)w + 2 * pad - 2;
    p.TH = p.OH / 2; p.TW = p.OW / 2;
    int bw = ilog2_ceil(p.TW);
    if (bw > 4) bw = 4;                         // at most 16 tiles of a row: a wave's half block writes 128-byte runs
    int bh = ilog2_ceil(p.TH);
    if (bw + bh > 6) bh = 6 - bw;
    p.bw_log2 = bw; p.bh_log2 = bh;
    const int BN = kWfT >> (bw + bh);
    p.blocks_x = ceil_div(p.TW, 1 << bw);
    p.blocks_y = ceil_div(p.TH, 1 << bh);
    const int64_t blocks = (int64_t)p.blocks_x * p.blocks_y * ceil_div64(n, BN);
    if (blocks >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: too many tile blocks");
    p.chunks = (int)ceil_div64(c, kWfCK);
    p.x_scale = x_scale; p.out_scale = out_scale; p.noise = noise; p.noise_w = noise_weight; p.bias = bias;
    p.act = act ? 1 : 0; p.slope = slope; p.act_scale = act_scale;
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div64(m, kWfM));
    const hipStream_t st = (hipStream_t)stream;
#define SAE_WF_LAUNCH(A, P2, X) hipLaunchKernelGGL((wino_fused_kernel<A, P2, X>), grid, dim3(kBlock), 0, st, x, uf, y, p)
    const int variant = (act ? 4 : 0) | (pad == 2 ? 2 : 0) | (x_scale ? 1 : 0);
    switch (variant) {
        case 0: SAE_WF_LAUNCH(false, false, false); break;
        case 1: SAE_WF_LAUNCH(false, false, true); break;
        case 2: SAE_WF_LAUNCH(false, true, false); break;
        case 3: SAE_WF_LAUNCH(false, true, true); break;
        case 4: SAE_WF_LAUNCH(true, false, false); break;
        case 5: SAE_WF_LAUNCH(true, false, true); break;
        case 6: SAE_WF_LAUNCH(true, true, false); break;
        default: SAE_WF_LAUNCH(true, true, true); break;
    }
#undef SAE_WF_LAUNCH
    return check_launch("sae_wino_fused_conv_f32");
}

extern "C" int64_t sae_wino_fused_wgrad_workspace(int64_t n, int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad) {
    if (n < 1 || c < 1 || m < 1 || h < 2 || w < 2 || pad < 0 || pad > 1) return 0;
    const int64_t th = (h + 2 * pad - 2) / 2, tw = (w + 2 * pad - 2) / 2;
    if (tw % 8 != 0) return 0;
    const int64_t mbs = ceil_div64(m, kWfM), cbs = ceil_div64(c, kWfM);
    return (int64_t)wgrad_slices(mbs, cbs, n * th * (tw / 8)) * mbs * kWfM * 9 * cbs * kWfM;
}

extern "C" int sae_wino_fused_wgrad_f32(const float* x, const float* x_scale, const float* gy, const float* y_scale, float* gw,
                                        int64_t n, int64_t c, int64_t m, int64_t h, int64_t w, int32_t pad, int64_t w_stride_m,
                                        int64_t w_stride_c, float alpha, float* workspace, int64_t workspace_floats,
                                        sae_stream_t stream) {
    sae::clear_stale_error();
    if (n < 1 || c < 1 || m < 1 || pad < 0 || pad > 1 || (h & 1) || (w & 1) || h + 2 * pad < 4 || h >= 32768 || w >= 32768 ||
        c >= (1 << 24) || m >= (1 << 24) || n >= (1 << 24))
        return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: bad shape");
    const int64_t oh = h + 2 * pad - 2, ow = w + 2 * pad - 2, th = oh / 2, tw = ow / 2;
    if (tw < 8 || tw % 8 != 0)
        return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: output rows of a multiple of 16 pixels (chunks of 8 tiles), got %lld",
                    (long long)ow);
    if (!x || !gy || !gw || !workspace) return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: null tensor");
    if (!aligned16(gy)) return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: gy must be 16-byte aligned");
    const int64_t need = sae_wino_fused_wgrad_workspace(n, c, m, h, w, pad);
    if (workspace_floats < need)
        return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: workspace of %lld floats, %lld needed", (long long)workspace_floats,
                    (long long)need);
    WinoWgradParams p;
    p.N = (int)n; p.C = (int)c; p.H = (int)h; p.W = (int)w; p.M = (int)m; p.OH = (int)oh; p.OW = (int)ow; p.pad = pad;
    p.TH = (int)th; p.TW = (int)tw; p.cpr = (int)(tw / 8);
    const int64_t chunks = n * th * p.cpr;
    if (chunks >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: too many tiles");
    p.chunks = (int)chunks;
    const int64_t mbs = ceil_div64(m, kWfM), cbs = ceil_div64(c, kWfM);
    const int slices = wgrad_slices(mbs, cbs, chunks);
    p.chunks_per_slice = (int)ceil_div64(chunks, slices);
    p.Mp = (int)(mbs * kWfM); p.Cp = (int)(cbs * kWfM);
    p.x_scale = x_scale; p.y_scale = y_scale;
    const dim3 grid((unsigned)cbs, (unsigned)mbs, (unsigned)slices);
    const hipStream_t st = (hipStream_t)stream;
    if (x_scale || y_scale)
        hipLaunchKernelGGL(wino_fused_wgrad_kernel<true>, grid, dim3(kBlock), 0, st, x, gy, workspace, p);
    else
        hipLaunchKernelGGL(wino_fused_wgrad_kernel<false>, grid, dim3(kBlock), 0, st, x, gy, workspace, p);
    int rc = check_launch("sae_wino_fused_wgrad_f32");
    if (rc != SAE_OK) return rc;
    hipLaunchKernelGGL(wino_fused_wgrad_reduce_kernel, dim3((unsigned)ceil_div64(m * 9 * c, kBlock)), dim3(kBlock), 0, st, workspace,
                       gw, (int)m, (int)c, p.Mp, p.Cp, slices, w_stride_m, w_stride_c, alpha);
    return check_launch("sae_wino_fused_wgrad_f32 (reduce)");
}
