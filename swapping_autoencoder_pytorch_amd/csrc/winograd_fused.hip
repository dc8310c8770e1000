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
// LDS: two stages x (U 32 KB + V 32 KB) = 128 KB; one barrier per chunk.  The patches of a chunk are loaded TWO chunks ahead
// (they sit in registers for a whole pass before their transform: the loads miss the L2 for every first channel block that
// touches them) and no MFMA group carries more than two loads per wave: the four waves' loads queue up in the CU's one
// texture addresser, and a wave that cannot issue its load cannot issue the MFMAs behind it either (round 6: +5 - 8 % on the
// 128 / 256-channel layers, profiles/r6_ab_wino_deep_prefetch.txt).
//
// Data gradient of the same layers: the same kernel on the output gradient with the flipped / transposed filter (flip = 1 in the
// weight preparation) and padding 2 - pad.
#include "sae_common.h"

#include <type_traits>

namespace sae {
namespace {

// The input transform works on packed pairs -- v_pk_add_f32 with op_sel / neg modifiers (sae_common.h: pk_sub / pk_c01 / pk_c23), 56
// instead of 84 vector-ALU instructions per thread and chunk -- with V kept in LDS as pairs of POINTS and the MFMA loop walking the
// points two at a time (+1.3 % on the 128 / 256-channel layers; the scalar form and the packed form of the weight gradient's x
// side, which measured 4 - 7 % SLOWER, are recorded in profiles/r5_ab_wino_fused_pk.txt and tools/archive/variants/).

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
    unsigned x_bytes;          // bytes of x (< 2^31: per-lane byte offsets are 32-bit, 0x80000000 = "outside": reads as zero)
    const float* x_scale;      // [N * C] or null: the style modulation of the input
    const float* out_scale;    // [N * M] or null
    const float* noise;        // [N][OH][OW] or null
    const float* noise_w;      // [1]
    const float* bias;         // [M] or null
    int act;
    float slope, act_scale;
};

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

// XS: 0 = no input scale, 1 = x_scale[n, c] fetched per lane, 2 = every tile of a block lies in ONE image (BN == 1: the maps with
// at least 64 tiles -- every modulated layer of the generator): the factor is wave-uniform and comes through a scalar load (two
// vector loads and their address arithmetic less per chunk)
template <bool ACT, bool PAD2, int XS>
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
    // Workgroup order.  Ids go round the 8 XCDs (id % 8), each with its own L2, in dispatch order.  Re-labelled so that an XCD
    // walks a CONTIGUOUS eighth of the tile blocks (neighbouring blocks share the halo rows / columns of their patches: fetched
    // once per L2 instead of once per XCD) and all XCDs work on the same channel block at a time (its 0.5 - 2 MB of prepared
    // weights stay in every L2 while the activations stream through).  Needs tile blocks % 8 == 0; otherwise launch order.
    int b = blockIdx.x, mb = blockIdx.y;
    if ((gridDim.x & 7) == 0) {
        const int lin = blockIdx.x + gridDim.x * blockIdx.y;
        const int per = gridDim.x >> 3;
        const int j = lin >> 3;
        mb = j / per;
        b = (lin & 7) * per + (j - mb * per);
    }
    const bool first_block = b == 0;
    const int bx = b % p.blocks_x;
    b /= p.blocks_x;
    const int by = b % p.blocks_y;
    const int bn = b / p.blocks_y;

    // ---- the thread's role in staging: tile `lane` of the block, channels 2 wid and 2 wid + 1 of a chunk.  Every row of the 4x4
    // patch is ONE 16-byte buffer load whose per-lane offset never changes (image + row + column of the tile; the channel rides
    // in the scalar offset): rows outside the image, and tiles outside the map, carry an out-of-range offset and read as zeros
    // (the descriptor's range check), a window starting left of the tensor wraps to an out-of-range offset too.  What is left
    // for the ALU is the zeroing of patch COLUMNS outside the image (the neighbouring row's data): blocks that touch the left /
    // right border only.  On this chip every VALU instruction of a wave costs its SIMD ~7 cycles of fp32-MFMA issue
    // (profiles/r5_pmc_wino_fused.txt), so the staging is built to need as few as possible.
    const int s_tx = bx * BW + (lane & (BW - 1));
    const int s_ty = by * BH + ((lane >> p.bw_log2) & (BH - 1));
    const int s_n = bn * BN + (lane >> bshift);
    const bool s_valid = s_tx < p.TW && s_ty < p.TH && s_n < p.N;
    [[maybe_unused]] const int n_u = __builtin_amdgcn_readfirstlane(bn * BN);        // XS == 2 (BN == 1): the block's image
    const int iy0 = 2 * s_ty - p.pad, ix0 = 2 * s_tx - p.pad;
    const int64_t HW = (int64_t)p.H * p.W;
    // The block that holds the tensor's first row (block 0) cannot start a window one float left of it: an offset of -4 wraps to
    // 4 GiB - 4 and the WHOLE 16-byte load reads as zeros.  Its windows are clamped into the row and shifted into place by
    // selects (mode 2: one workgroup per channel block); every other block takes the unclamped window (modes 0 / 1).
    int cx = ix0;
    if (first_block) {
        cx = ix0 < 0 ? 0 : ix0;
        if (cx > p.W - 4) cx = p.W - 4;
    }
    const int shift = ix0 - cx;            // mode 2: wanted element q is loaded element q + shift (-2 .. 2)
    unsigned rowv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int iy = iy0 + r;
        const bool ok = s_valid && iy >= 0 && iy < p.H;
        rowv[r] = ok ? (unsigned)(((int64_t)s_n * p.C * HW + (int64_t)iy * p.W + cx) * 4) : 0x80000000u;
    }
    bool colok[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) colok[q] = ix0 + q >= 0 && ix0 + q < p.W;
    // does any tile of this block have a patch column outside the image?  (uniform)
    const int tx_lo = bx * BW, tx_hi = (bx + 1) * BW - 1;
    const bool x_interior = 2 * tx_lo - p.pad >= 0 && 2 * tx_hi - p.pad + 3 < p.W && tx_hi < p.TW;
    const unsigned plane_bytes = (unsigned)(HW * 4);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Uf + (int64_t)mb * p.chunks * kWfStage), 0, (unsigned)(p.chunks * kWfStage * 4), 0x00020000);
    const unsigned uoff = tid * 16;

    f32x4 dreg[2][4];          // the two channels' 4x4 patches of the NEXT chunk
    f32x4 ureg[8];             // this thread's 8 quads of the next chunk's weights
    f32x2 vreg[16];            // B^T d B of both channels, as written to LDS
    float xsc[2] = {1.0f, 1.0f};

    // ---- the staging of a chunk, in pieces the main loop spreads over the sixteen MFMA groups of the previous chunk
    auto load_x = [&](int chunk, int c2) {                     // 4 buffer loads, no vector ALU
        int ch = chunk * kWfCK + 2 * wid + c2;                 // wave-uniform
        if (ch > p.C - 1) ch = p.C - 1;                        // (a channel beyond C meets zero weights)
        const unsigned soff = (unsigned)ch * plane_bytes;
        // (the channel plane rides in the SCALAR offset.  gfx950 range-checks a raw buffer access as offset >= num_records -
        // soffset -- measured in round 6: with num_records reduced by soff on top, the last image's planes read as zeros -- so a
        // right-border window in the tensor's last row cannot reach past x for any channel)
#pragma unroll
        for (int r = 0; r < 4; ++r) dreg[c2][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, rowv[r], soff, 0));
        if (XS == 2) xsc[c2] = p.x_scale[(int64_t)n_u * p.C + ch];
        else if (XS) xsc[c2] = p.x_scale[(int64_t)(s_valid ? s_n : 0) * p.C + ch];
    };
    auto load_u = [&](int chunk, int lo) {                     // 4 buffer loads, no vector ALU
#pragma unroll
        for (int j = lo; j < lo + 4; ++j)
            ureg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uoff, (unsigned)(chunk * kWfStage * 4 + j * 4096), 0));
    };
    auto transform = [&](auto mode_tag, int c2) {              // B^T d B of one channel
        constexpr int MODE = decltype(mode_tag)::value;        // 0: all columns inside, 1: zero the outside columns, 2: shifted windows
        f32x4 d[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) d[r] = XS ? dreg[c2][r] * xsc[c2] : dreg[c2][r];
        f32x4 e[4];
        e[0] = d[0] - d[2];
        e[1] = d[1] + d[2];
        e[2] = d[2] - d[1];
        e[3] = d[1] - d[3];
        if (MODE == 2) {           // (the row transform acts column by column, so the shift can follow it)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const f32x4 l = e[a];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = 0.0f;
#pragma unroll
                    for (int k = -2; k <= 2; ++k)
                        if (q + k >= 0 && q + k < 4) t = (shift == k) ? l[q + k] : t;
                    e[a][q] = colok[q] ? t : 0.0f;
                }
            }
        }
        if (MODE == 1) {           // columns of the patch outside the image hold the neighbouring row's data: zero them
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                e[a][0] = colok[0] ? e[a][0] : 0.0f;
                e[a][3] = colok[3] ? e[a][3] : 0.0f;
                if (PAD2) {
                    e[a][1] = colok[1] ? e[a][1] : 0.0f;
                    e[a][2] = colok[2] ? e[a][2] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            vreg[4 * a + 0][c2] = e[a][0] - e[a][2];
            vreg[4 * a + 1][c2] = e[a][1] + e[a][2];
            vreg[4 * a + 2][c2] = e[a][2] - e[a][1];
            vreg[4 * a + 3][c2] = e[a][1] - e[a][3];
        }
    };
    // packed form: pv[c2][a][bp] = (V[4 a + 2 bp], V[4 a + 2 bp + 1]) of channel c2; LDS: Vs[a][bp][half][s][t] pairs, channel
    // 2 wid + c2 = (half = wid >> 1, s = 2 (wid & 1) + c2)
    f32x2 pv[2][4][2];
    auto transform_pk = [&](auto mode_tag, int c2) {
        constexpr int MODE = decltype(mode_tag)::value;
        f32x2 dl[4], dh[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const f32x4 d = XS ? dreg[c2][r] * xsc[c2] : dreg[c2][r];
            dl[r] = f32x2{d[0], d[1]};
            dh[r] = f32x2{d[2], d[3]};
        }
        f32x2 el[4], eh[4];
        el[0] = pk_sub(dl[0], dl[2]); eh[0] = pk_sub(dh[0], dh[2]);
        el[1] = dl[1] + dl[2];        eh[1] = dh[1] + dh[2];
        el[2] = pk_sub(dl[2], dl[1]); eh[2] = pk_sub(dh[2], dh[1]);
        el[3] = pk_sub(dl[1], dl[3]); eh[3] = pk_sub(dh[1], dh[3]);
        if (MODE == 2) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                const float l[4] = {el[a][0], el[a][1], eh[a][0], eh[a][1]};
                float o[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float t = 0.0f;
#pragma unroll
                    for (int k = -2; k <= 2; ++k)
                        if (q + k >= 0 && q + k < 4) t = (shift == k) ? l[q + k] : t;
                    o[q] = colok[q] ? t : 0.0f;
                }
                el[a] = f32x2{o[0], o[1]};
                eh[a] = f32x2{o[2], o[3]};
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                el[a][0] = colok[0] ? el[a][0] : 0.0f;
                eh[a][1] = colok[3] ? eh[a][1] : 0.0f;
                if (PAD2) {
                    el[a][1] = colok[1] ? el[a][1] : 0.0f;
                    eh[a][0] = colok[2] ? eh[a][0] : 0.0f;
                }
            }
        }
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            pv[c2][a][0] = pk_c01(el[a], eh[a]);
            pv[c2][a][1] = pk_c23(el[a], eh[a]);
        }
    };
    auto write_v = [&](int buf, int lo) {                      // 8 LDS 8-byte writes: pairs lo .. lo + 3 of both channels
        f32x2* vd = reinterpret_cast<f32x2*>(Vs[buf]) + ((wid >> 1) * 4 + 2 * (wid & 1)) * 64 + lane;
#pragma unroll
        for (int pp = lo / 2; pp < lo / 2 + 4; ++pp)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) vd[pp * 512 + c2 * 64] = pv[c2][pp >> 1][pp & 1];
    };
    auto write_u = [&](int buf, int lo) {                      // 4 LDS 16-byte writes
        f32x4* ud = reinterpret_cast<f32x4*>(Us[buf]) + tid;
#pragma unroll
        for (int j = lo; j < lo + 4; ++j) ud[j * kBlock] = ureg[j];
    };

    // never zeroed: the first pass of a workgroup starts every accumulator from a constant-zero C operand instead (256 accumulator
    // writes = 0.5 us per workgroup, in front of the loop with nothing to hide behind)
    f32x16 acc[16];

    auto load_x_rows = [&](int chunk, int c2, int r0) {
        int ch = chunk * kWfCK + 2 * wid + c2;
        if (ch > p.C - 1) ch = p.C - 1;
        const unsigned soff = (unsigned)ch * plane_bytes;
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) dreg[c2][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, rowv[r], soff, 0));
        if (XS == 2 && r0 == 0) xsc[c2] = p.x_scale[(int64_t)n_u * p.C + ch];
        else if (XS && r0 == 0) xsc[c2] = p.x_scale[(int64_t)(s_valid ? s_n : 0) * p.C + ch];
    };
    auto load_u2 = [&](int chunk, int j0) {
#pragma unroll
        for (int j = j0; j < j0 + 2; ++j)
            ureg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uoff, (unsigned)(chunk * kWfStage * 4 + j * 4096), 0));
    };
    // the patches of chunk c1 were loaded during the PREVIOUS pass (a whole pass of latency cover); every group carries at most
    // two loads per wave (the four waves' loads of a group queue up in the one texture addresser of the CU)
    auto pass = [&](auto mode_tag, auto first_tag, int cur, int c1, int c2n) {
        constexpr bool FIRST = decltype(first_tag)::value;
        const f32x16 zero16 = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
        const f32x4* ua = reinterpret_cast<const f32x4*>(Us[cur]) + (half * 64 + wm * 32 + l31);
        const f32x2* vb = reinterpret_cast<const f32x2*>(Vs[cur]) + (half * 256 + wt * 32 + l31);
        f32x4 a0 = ua[0], a1 = ua[128];
        f32x2 b[4] = {vb[0], vb[64], vb[128], vb[192]};
#pragma unroll
        for (int pp = 0; pp < 8; ++pp) {
            f32x4 a0n = a0, a1n = a1;
            f32x2 bn[4] = {b[0], b[1], b[2], b[3]};
            if (pp < 7) {
                a0n = ua[(2 * pp + 2) * 128];
                a1n = ua[(2 * pp + 3) * 128];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) bn[s4] = vb[(pp + 1) * 512 + s4 * 64];
            }
            if (pp < 4) load_u2(c1, 2 * pp);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], b[s4][0], (FIRST && s4 == 0) ? zero16 : acc[2 * pp], 0, 0, 0);
                acc[2 * pp + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], b[s4][1], (FIRST && s4 == 0) ? zero16 : acc[2 * pp + 1], 0, 0, 0);
            }
            if (pp == 0) transform_pk(mode_tag, 0);
            if (pp == 1) transform_pk(mode_tag, 1);
            if (pp == 2) { write_v(cur ^ 1, 0); load_x_rows(c2n, 0, 0); }
            if (pp == 3) { write_v(cur ^ 1, 8); load_x_rows(c2n, 0, 2); }
            if (pp == 4) load_x_rows(c2n, 1, 0);
            if (pp == 5) load_x_rows(c2n, 1, 2);
            if (pp == 6) write_u(cur ^ 1, 0);
            if (pp == 7) write_u(cur ^ 1, 4);
            a0 = a0n;
            a1 = a1n;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) b[s4] = bn[s4];
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    auto run = [&](auto mode_tag) {
        const int last = p.chunks - 1;
        load_x(0, 0);
        load_x(0, 1);
        load_u(0, 0);
        load_u(0, 4);
        transform_pk(mode_tag, 0);
        transform_pk(mode_tag, 1);
        load_x(last < 1 ? last : 1, 0);
        load_x(last < 1 ? last : 1, 1);
        write_v(0, 0);
        write_v(0, 8);
        write_u(0, 0);
        write_u(0, 4);
        __syncthreads();
        pass(mode_tag, std::true_type{}, 0, 1 < last ? 1 : last, 2 < last ? 2 : last);
        __syncthreads();
        int cur = 1;
        for (int chunk = 1; chunk < p.chunks; ++chunk) {
            pass(mode_tag, std::false_type{}, cur, chunk + 1 < last ? chunk + 1 : last, chunk + 2 < last ? chunk + 2 : last);
            __syncthreads();
            cur ^= 1;
        }
    };
    if (first_block)
        run(std::integral_constant<int, 2>{});
    else if (x_interior)
        run(std::integral_constant<int, 0>{});
    else
        run(std::integral_constant<int, 1>{});

    // ---- output transform, lane-local: acc[4 a + b][r] of (m = ... r ..., tile = wt * 32 + l31)
    const int ot = wt * 32 + l31;
    const int o_tx = bx * BW + (ot & (BW - 1));
    const int o_ty = by * BH + ((ot >> p.bw_log2) & (BH - 1));
    const int o_n = bn * BN + (ot >> bshift);
    if (!(o_tx < p.TW && o_ty < p.TH && o_n < p.N)) return;
    const int64_t OHW = (int64_t)p.OH * p.OW;
    const int64_t pix = (int64_t)(2 * o_ty) * p.OW + 2 * o_tx;
    // (the identities of the epilogue's optional factors -- out_scale 1, noise -0 -- leave every value as it is, signed zeros
    // included: the loop below applies them unconditionally instead of selecting per output)
    float nz[2][2] = {{-0.0f, -0.0f}, {-0.0f, -0.0f}};
    if (ACT && p.noise) {
        const float nw = p.noise_w[0];
        const float* zp = p.noise + (int64_t)o_n * OHW + pix;
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            nz[a][0] = nw * zp[(int64_t)a * p.OW];
            nz[a][1] = nw * zp[(int64_t)a * p.OW + 1];
        }
    }
    // The per-channel factors of the lane's sixteen rows are fetched in ONE go and the loop below is straight-line code (rows
    // beyond M are computed and not stored): with the loads and the `m < M` branch inside it every row waited for its own two
    // loads -- and, vmcnt counting stores as well, for the previous row's stores -- sixteen round trips per workgroup with nobody
    // else on the SIMD to fill them (round 6: profiles/r6_ab_wino_epilogue.txt)
    const int m_lane = mb * kWfM + wm * 32 + 4 * half;
    float psv[16], bvv[16];
    if (p.out_scale) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_lane + (r & 3) + 8 * (r >> 2);
            psv[r] = p.out_scale[(int64_t)o_n * p.M + (m < p.M ? m : p.M - 1)];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) psv[r] = 1.0f;
    }
    if (ACT && p.bias) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m_lane + (r & 3) + 8 * (r >> 2);
            bvv[r] = p.bias[m < p.M ? m : p.M - 1];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) bvv[r] = 0.0f;
    }
    // all of them have landed before the first store is issued: the compiler's counter bookkeeping across the sixteen conditional
    // store blocks would otherwise make every row wait for (nearly) all stores issued before it
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
    float* yp = y + ((int64_t)o_n * p.M + m_lane) * OHW + pix;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int mo = (r & 3) + 8 * (r >> 2);
        float t[2][4];      // A^T acc
#pragma unroll
        for (int bq = 0; bq < 4; ++bq) {
            t[0][bq] = (acc[bq][r] + acc[4 + bq][r]) + acc[8 + bq][r];
            t[1][bq] = (acc[4 + bq][r] - acc[8 + bq][r]) - acc[12 + bq][r];
        }
        const float ps = psv[r];
        const float bv = bvv[r];
        f32x2 o[2];
#pragma unroll
        for (int a = 0; a < 2; ++a) {
#pragma clang fp contract(off)      // product, sum, sum as the reference's separate ops round them (no fma across the factors)
            float o0 = ((t[a][0] + t[a][1]) + t[a][2]) * ps;
            float o1 = ((t[a][1] - t[a][2]) - t[a][3]) * ps;
            if (ACT) {
                o0 = (o0 + nz[a][0]) + bv;                                      // (image + weight * noise) + bias
                o1 = (o1 + nz[a][1]) + bv;
                o0 = ((o0 > 0.0f) ? o0 : o0 * p.slope) * p.act_scale;
                o1 = ((o1 > 0.0f) ? o1 : o1 * p.slope) * p.act_scale;
            }
            o[a] = f32x2{o0, o1};
        }
        if (m_lane + mo < p.M) {
            *reinterpret_cast<f32x2*>(yp) = o[0];                              // 2 tx and OW are even: 8-byte aligned
            *reinterpret_cast<f32x2*>(yp + p.OW) = o[1];
        }
        yp += ((r & 3) == 3 ? 5 : 1) * OHW;
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
    unsigned x_bytes, gy_bytes; // < 2^31: per-lane byte offsets are 32-bit, 0x80000000 = "outside": reads as zero
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

    // ---- staging role: channel `ch` of both 64-channel blocks, tiles 4 hf + 2 q1 and + 1 of a chunk (hf wave-uniform).  All
    // loads are buffer loads: the per-lane offset (channel plane + the pair's column) never changes, the chunk's position (image,
    // tile row, 8-tile group) is scalar; window rows outside the image get an out-of-range offset and read as zeros.  Channels
    // beyond M / C are clamped to the last one: their products land in slab rows / columns nobody reads.
    const int q1 = lane & 1;
    const int ch = (wid & 1) * 32 + (lane >> 1);
    const int hf = wid >> 1;
    const int kk0 = 4 * hf + 2 * q1;
    const int m_ch = mb * kWfM + ch < p.M ? mb * kWfM + ch : p.M - 1;
    const int c_ch = cb * kWfM + ch < p.C ? cb * kWfM + ch : p.C - 1;
    const int64_t HW = (int64_t)p.H * p.W, OHW = (int64_t)p.OH * p.OW;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t grs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(gy), 0, p.gy_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t sxr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MOD && p.x_scale ? p.x_scale : x), 0,
                                                                       (unsigned)(p.N * p.C * 4), 0x00020000);
    const __amdgpu_buffer_rsrc_t syr = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(MOD && p.y_scale ? p.y_scale : gy), 0,
                                                                       (unsigned)(p.N * p.M * 4), 0x00020000);
    const unsigned g_lane = (unsigned)(((int64_t)m_ch * OHW + 2 * kk0) * 4);
    const unsigned x_lane = (unsigned)(((int64_t)c_ch * HW + 2 * kk0 - p.pad) * 4);      // (-4 at most: wraps, with the scalar part, to "outside")
    const unsigned ow_bytes = (unsigned)(p.OW * 4);

    const int ch_begin = slice * p.chunks_per_slice;
    int ch_end = ch_begin + p.chunks_per_slice;
    if (ch_end > p.chunks) ch_end = p.chunks;
    // position of the chunk being LOADED: image n, tile row ty, chunk txb of the row -- and the scalar byte offsets that go with
    // it, kept RUNNING (+ 64 bytes per chunk, a row / image step at the wraps): recomputed from (n, ty, txb) per chunk they were
    // 120 scalar instructions per 64 MFMAs, and every instruction of the wave takes an issue slot from the matrix pipe
    int ld_txb = ch_begin % p.cpr;
    int ld_ty = (ch_begin / p.cpr) % p.TH;
    int ld_n = ch_begin / (p.cpr * p.TH);
    unsigned g_soff = 0, x_soff = 0, up_row[4] = {0, 0, 0, 0};
    auto place = [&]() {                   // offsets of (ld_n, ld_ty, ld_txb) from scratch: slice start and image wraps
        g_soff = (unsigned)((((int64_t)ld_n * p.M) * OHW + (int64_t)(2 * ld_ty) * p.OW + 16 * ld_txb) * 4);
        x_soff = (unsigned)((((int64_t)ld_n * p.C) * HW + 16 * ld_txb) * 4);         // image + column; the row is in up_row
    };
    auto rows = [&]() {                    // per tile row: the four window rows, "outside" for those not in the image (uniform)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int iy = 2 * ld_ty - p.pad + r;
            up_row[r] = (iy >= 0 && iy < p.H) ? (unsigned)(iy * p.W * 4) : 0x80000000u;
        }
    };
    place();
    rows();

    f32x4 greg[2];             // gy rows 2 ty, 2 ty + 1: two tiles x two columns
    f32x4 xr4[4];              // window rows: columns 0 .. 3
    f32x2 xr2[4];              //              columns 4, 5
    bool zl = false, zr = false, edge = false;      // the pair's window sticks out of the image on the left / right; the chunk has such a pair
    float sx = 1.0f, sy = 1.0f;
    f32x2 ev[16], vv[16];      // A e A^T and B^T d B of the two tiles, as written to LDS

    auto load_gy = [&]() {
        greg[0] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, g_lane, g_soff, 0));
        greg[1] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(grs, g_lane + ow_bytes, g_soff, 0));
    };
    auto load_scales = [&]() {             // the two style factors of the chunk being loaded (its image)
        if (MOD) {
            sx = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(sxr, (unsigned)(c_ch * 4), (unsigned)(ld_n * p.C * 4), 0));
            sy = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(syr, (unsigned)(m_ch * 4), (unsigned)(ld_n * p.M * 4), 0));
            sx = p.x_scale ? sx : 1.0f;
            sy = p.y_scale ? sy : 1.0f;
        }
    };
    auto load_x = [&](int r0) {            // window rows r0, r0 + 1
        if (r0 == 0) {
            edge = p.pad == 1 && (ld_txb == 0 || ld_txb == p.cpr - 1);      // uniform: this chunk touches the left / right border
            zl = p.pad == 1 && ld_txb == 0 && kk0 == 0;
            zr = p.pad == 1 && ld_txb == p.cpr - 1 && kk0 == 6;
        }
        // the pair at the left border starts its window AT column 0 (one float further left would be offset -4 in the tensor's
        // first row: it wraps and the whole load reads as zeros) and is shifted into place in transform_v.  A window row
        // outside the image adds 0x80000000 to a sum that is non-negative and below 2 GiB: out of range.
        const unsigned vl = x_lane + (zl ? 4u : 0u);
#pragma unroll
        for (int r = r0; r < r0 + 2; ++r) {
            const unsigned v = vl + x_soff + up_row[r];
            xr4[r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, v, 0, 0));
            xr2[r] = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(xrs, v + 16, 0, 0));
        }
    };
    auto advance = [&]() {
        ++ld_txb;
        g_soff += 64;
        x_soff += 64;
        if (ld_txb == p.cpr) {
            ld_txb = 0;
            ++ld_ty;
            if (ld_ty == p.TH) {
                ld_ty = 0;
                ++ld_n;
                place();
            } else {
                g_soff += (unsigned)((2 * p.OW - 16 * p.cpr) * 4);
                x_soff -= (unsigned)(16 * p.cpr * 4);
            }
            rows();
        }
    };
    auto transform_e = [&]() {             // A e A^T of both tiles
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float e00 = greg[0][2 * t], e01 = greg[0][2 * t + 1], e10 = greg[1][2 * t], e11 = greg[1][2 * t + 1];
            if (MOD) { e00 *= sy; e01 *= sy; e10 *= sy; e11 *= sy; }
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
    auto transform_v = [&]() {             // B^T d B of both tiles (columns 0 .. 3 and 2 .. 5 of the 4 x 6 window)
        float e[4][6];
#pragma unroll
        for (int q = 0; q < 6; ++q) {
            float d0 = q < 4 ? xr4[0][q] : xr2[0][q - 4], d1 = q < 4 ? xr4[1][q] : xr2[1][q - 4];
            float d2 = q < 4 ? xr4[2][q] : xr2[2][q - 4], d3 = q < 4 ? xr4[3][q] : xr2[3][q - 4];
            if (MOD) { d0 *= sx; d1 *= sx; d2 *= sx; d3 *= sx; }
            e[0][q] = d0 - d2;
            e[1][q] = d1 + d2;
            e[2][q] = d2 - d1;
            e[3][q] = d1 - d3;
        }
        if (edge) {                        // (uniform branch: chunks at the left / right border of the map only)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                // left: the window was loaded from column 0 -> shift right by one, column -1 is padding;
                // right: the last column holds the next row's first element -> padding
                const float l0 = e[a][0], l1 = e[a][1], l2 = e[a][2], l3 = e[a][3], l4 = e[a][4];
                e[a][0] = zl ? 0.0f : l0;
                e[a][1] = zl ? l0 : l1;
                e[a][2] = zl ? l1 : l2;
                e[a][3] = zl ? l2 : l3;
                e[a][4] = zl ? l3 : l4;
                e[a][5] = zl ? l4 : (zr ? 0.0f : e[a][5]);
            }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int a = 0; a < 4; ++a) {
                vv[4 * a + 0][t] = e[a][2 * t + 0] - e[a][2 * t + 2];
                vv[4 * a + 1][t] = e[a][2 * t + 1] + e[a][2 * t + 2];
                vv[4 * a + 2][t] = e[a][2 * t + 2] - e[a][2 * t + 1];
                vv[4 * a + 3][t] = e[a][2 * t + 1] - e[a][2 * t + 3];
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

    // the operands of chunk + 1 were loaded during the PREVIOUS pass; they are transformed first, and the loads of chunk + 2
    // follow as soon as their registers are free
    auto pass = [&](int cur, bool more) {
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
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) acc[xi] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s4], b[s4], acc[xi], 0, 0, 0);
            // (this order is measured: profiles/r6_ab_wino_deep_prefetch.txt -- with the gy loads between the two transforms the
            // kernel is 5 - 6 % faster than with both transforms first)
            if (xi == 0) transform_e();
            if (xi == 1) load_gy();
            if (xi == 3) transform_v();
            if (xi == 4) {
                load_scales();               // (after BOTH transforms: they use the factors of the chunk loaded a pass ago)
                load_x(0);
            }
            if (xi == 5) {
                load_x(2);
                if (more) advance();
            }
            if (xi == 8) write_e(cur ^ 1, 0);
            if (xi == 9) write_e(cur ^ 1, 8);
            if (xi == 10) write_v(cur ^ 1, 0);
            if (xi == 11) write_v(cur ^ 1, 8);
            a = an;
            b = bn;
            __builtin_amdgcn_sched_barrier(0);
        }
    };
    if (ch_begin < ch_end) {
        load_gy();
        load_scales();
        load_x(0);
        load_x(2);
        transform_e();
        transform_v();
        write_e(0, 0);
        write_e(0, 8);
        write_v(0, 0);
        write_v(0, 8);
        if (ch_begin + 1 < ch_end) advance();
        load_gy();
        load_scales();
        load_x(0);
        load_x(2);
        if (ch_begin + 2 < ch_end) advance();
        __syncthreads();
        int cur = 0;
        for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
            pass(cur, chunk + 3 < ch_end);
            __syncthreads();
            cur ^= 1;
        }
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
    // eight loads in flight, added in slice order (a chain of `slices` dependent loads per thread ran at 1.4 TB/s)
    float acc = 0.0f;
    int s = 0;
    for (; s + 8 <= slices; s += 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) v[u] = sp[(s + u) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; s < slices; ++s) acc += sp[s * stride];
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
    if (n * c * h * w * 4 >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_wino_fused_conv_f32: x of %lld bytes; the kernel addresses it with 32-bit byte offsets (< 2 GiB)",
                    (long long)(n * c * h * w * 4));
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
    p.x_bytes = (unsigned)(n * c * h * w * 4);
    p.x_scale = x_scale; p.out_scale = out_scale; p.noise = noise; p.noise_w = noise_weight; p.bias = bias;
    p.act = act ? 1 : 0; p.slope = slope; p.act_scale = act_scale;
    const dim3 grid((unsigned)blocks, (unsigned)ceil_div64(m, kWfM));
    const hipStream_t st = (hipStream_t)stream;
#define SAE_WF_LAUNCH(A, P2, X) hipLaunchKernelGGL((wino_fused_kernel<A, P2, X>), grid, dim3(kBlock), 0, st, x, uf, y, p)
    const int xs = x_scale ? (BN == 1 ? 2 : 1) : 0;
    const int variant = (act ? 6 : 0) + (pad == 2 ? 3 : 0) + xs;
    switch (variant) {
        case 0: SAE_WF_LAUNCH(false, false, 0); break;
        case 1: SAE_WF_LAUNCH(false, false, 1); break;
        case 2: SAE_WF_LAUNCH(false, false, 2); break;
        case 3: SAE_WF_LAUNCH(false, true, 0); break;
        case 4: SAE_WF_LAUNCH(false, true, 1); break;
        case 5: SAE_WF_LAUNCH(false, true, 2); break;
        case 6: SAE_WF_LAUNCH(true, false, 0); break;
        case 7: SAE_WF_LAUNCH(true, false, 1); break;
        case 8: SAE_WF_LAUNCH(true, false, 2); break;
        case 9: SAE_WF_LAUNCH(true, true, 0); break;
        case 10: SAE_WF_LAUNCH(true, true, 1); break;
        default: SAE_WF_LAUNCH(true, true, 2); break;
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
    if (n * c * h * w * 4 >= ((int64_t)1 << 31) || n * m * oh * ow * 4 >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_wino_fused_wgrad_f32: the kernel addresses x and gy with 32-bit byte offsets (< 2 GiB each)");
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
    p.x_bytes = (unsigned)(n * c * h * w * 4); p.gy_bytes = (unsigned)(n * m * oh * ow * 4);
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
