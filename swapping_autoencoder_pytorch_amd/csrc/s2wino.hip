// The 3x3 STRIDE-2 convolution family on its polyphase minimal-filtering form: 25 instead of 36 multiplications per 2x2 of
// low-resolution positions, input / output transforms in registers, ONE kernel per operation.
//
// The reference reaches these layers as Blur -> F.conv2d(stride 2) (ConvLayer(downsample=True),
// models/networks/stylegan2_layers.py:627-648) and F.conv_transpose2d(stride 2) -> Blur (ModulatedConv2d(upsample=True), :296-309):
// the large side of the pair is always a (2H+1) x (2W+1) map, the small side H x W.  Along one axis, with taps w0 w1 w2,
//
//     y[o] = w0 x[2o] + w1 x[2o+1] + w2 x[2o+2]                     forward
//     dx[2m] = w0 g[m] + w2 g[m-1],   dx[2m+1] = w1 g[m]            data gradient / transposed convolution
//
// The EVEN positions of the large side meet a 2-tap filter (w0, w2), the ODD positions a 1-tap filter (w1).  Two outputs of a
// 2-tap filter take 3 multiplications (F(2,2): (a-b) w2, b (w0+w2), (c-b) w0), two outputs of a 1-tap filter take 2: 5 instead
// of 6 per pair and axis, 25 instead of 36 in the plane.  In the plane the 25 products fall into four CLASSES by the parity of
// the large-side row / column: EE 3x3 = 9 points, EO 3x2 = 6, OE 2x3 = 6, OO 2x2 = 4.
//
// Data gradient (s2w_dgrad_kernel): the four classes write DISJOINT outputs (the parity classes of dx), so a workgroup takes
// one half of the points -- type A = EE + OO (13 points), type B = EO + OE (12) -- for 64 output channels x 64 tiles; a tile is
// the 3x3 patch g[2ty-1 .. 2ty+1][2tx-1 .. 2tx+1] -> the 4x4 block dx[4ty .. 4ty+3][4tx .. 4tx+3].  The pipeline is
// winograd_fused.hip's: per chunk of 8 contraction channels a wave issues 4 MFMAs (32x32x2 fp32) per point against operands
// staged through two LDS stages with ONE barrier per chunk; the patches are loaded two chunks ahead and transformed at the
// start of the pass that follows, and no MFMA group carries more than two loads per wave.  The (2H+1)-th row / column of dx
// (tiles ty = H/2 or tx = W/2: one extra tile row and column whose patches are mostly padding) is a linear list of STRIP tiles
// after the main H/2 x W/2 region, so the main region keeps power-of-two tile blocks.
//
// Measured (profiles/r6_ab_s2wino_dgrad.txt): 1.15 - 1.25x the direct kernel on the 2^k + 1 grids up to 65 wide with >= 512
// contraction channels (the direct kernel's tiles quantise badly there), parity or behind on the 129 / 257-wide maps: the
// 1.44x fewer MFMAs are paid for with 13 / 12 points per staged patch instead of the 16 of F(2x2,3x3) and a checkerboard of
// 4-byte stores.  The split by ROW parity (15 / 10 points, whole 16-byte runs per lane) fixed the stores and lost as much to
// the short passes of its 10-point type (tools/archive/variants/src/s2wino_row_parity_split_with_experiment_switches.hip).  The
// route (stylegan2_op/winograd.py) takes this kernel only where it wins.
#include "sae_common.h"

#include <type_traits>

namespace sae {
namespace {

constexpr int kS2Q = 64;        // output channels per workgroup
constexpr int kS2T = 64;        // tiles per workgroup
constexpr int kS2CK = 8;        // contraction channels per chunk

template <int TYPE>
struct S2wPoints {
    static constexpr int NP = TYPE == 0 ? 13 : 12;      // points of the type
    static constexpr int NPAIR = TYPE == 0 ? 7 : 6;     // pairs of points (the LDS words of V hold two points)
    static constexpr int NPS = 2 * NPAIR;               // point slots of a weight stage
};
constexpr int kS2Slots0 = 14, kS2Slots1 = 12;
constexpr int kS2SlotsAll = kS2Slots0 + kS2Slots1;

struct S2wParams {
    int N, K, H, W;            // small side  [N][K][H][W], H and W even
    int Q, OH, OW;             // large side  [N][Q][2H+1][2W+1]
    int MH, MW;                // main tiles per image: H / 2, W / 2
    int bw_log2, bh_log2;      // main tile block = BN x BH x BW, BN * BH * BW = 64
    int blocks_x, blocks_y;
    int main_blocks, blocks;   // tile blocks: main region, main + strips
    int per;                   // tile blocks per XCD: ceil(blocks / 8)
    int qbs;                   // 64-channel output blocks
    int chunks;                // ceil(K / 8)
    unsigned in_bytes;         // bytes of the small-side tensor (< 2^31)
    const float* in_scale;     // [N * K] or null
    const float* out_scale;    // [N * Q] or null
};

// tile `idx` of tile block `b` (b is wave-uniform)
__device__ __forceinline__ bool s2w_tile(const S2wParams& p, int b, int idx, int& n, int& ty, int& tx) {
    if (b < p.main_blocks) {
        const int BW = 1 << p.bw_log2, BH = 1 << p.bh_log2;
        const int bshift = p.bw_log2 + p.bh_log2;
        const int bx = b % p.blocks_x;
        const int t = b / p.blocks_x;
        const int by = t % p.blocks_y;
        const int bn = t / p.blocks_y;
        tx = bx * BW + (idx & (BW - 1));
        ty = by * BH + ((idx >> p.bw_log2) & (BH - 1));
        n = bn * (kS2T >> bshift) + (idx >> bshift);
        return tx < p.MW && ty < p.MH && n < p.N;
    }
    const int S = p.MW + p.MH + 1;                      // strip tiles per image: the row ty = MH, then the column tx = MW
    const int s = (b - p.main_blocks) * kS2T + idx;
    n = s / S;
    const int r = s - n * S;
    if (r <= p.MW) {
        ty = p.MH;
        tx = r;
    } else {
        tx = p.MW;
        ty = r - p.MW - 1;
    }
    return n < p.N;
}

// Uf: type A (EE + OO) then type B (EO + OE); within a type [qb][chunk][slot][half][ql][s] = U[slot][q = 64 qb + ql][k = 8 chunk +
// 4 half + s], zero beyond Q / K and in type A's fourteenth slot: one chunk of one channel block is contiguous in the order the
// kernel keeps it in LDS.  g[ky][kx] are the taps as the PRODUCT sees them (flip = 1: the data gradient meets the reversed
// filter); along an axis the even class multiplies by (g0, g0 + g2, g2) and the odd class by g1.
//   type A: EE point 3 i + j, OO point 9 + 2 i' + j'     (i, j: even-class row / column factor; i', j': odd row / column)
//   type B: EO point 2 i + j', OE point 6 + 3 i' + j
__global__ __launch_bounds__(kBlock) void s2w_wprep_kernel(const float* __restrict__ w, float* __restrict__ Uf, int Q, int K, int chunks,
                                                           int64_t sq, int64_t sk, int flip, float alpha,
                                                           const float* __restrict__ rs_q, const float* __restrict__ rs_k) {
    const int Kp = chunks * kS2CK;
    const int qbs = (Q + kS2Q - 1) / kS2Q;
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)qbs * kS2Q * Kp) return;
    const int s = (int)(i & 3);
    const int ql = (int)((i >> 2) & 63);
    const int hf = (int)((i >> 8) & 1);
    const int64_t rest = i >> 9;
    const int chunk = (int)(rest % chunks), qb = (int)(rest / chunks);
    const int q = qb * kS2Q + ql, k = chunk * kS2CK + 4 * hf + s;
    float ua[kS2Slots0], ub[kS2Slots1];
#pragma unroll
    for (int t = 0; t < kS2Slots0; ++t) ua[t] = 0.0f;
#pragma unroll
    for (int t = 0; t < kS2Slots1; ++t) ub[t] = 0.0f;
    if (q < Q && k < K) {
        const float* wp = w + q * sq + k * sk;
        float g[3][3];
#pragma unroll
        for (int t = 0; t < 9; ++t) {
            float v = alpha * wp[flip ? 8 - t : t];
            if (rs_q) v *= rs_q[q];
            if (rs_k) v *= rs_k[k];
            g[t / 3][t % 3] = v;
        }
        float r[3][3];      // even-class factor along the rows
#pragma unroll
        for (int kx = 0; kx < 3; ++kx) {
            r[0][kx] = g[0][kx];
            r[1][kx] = g[0][kx] + g[2][kx];
            r[2][kx] = g[2][kx];
        }
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            ua[3 * a + 0] = r[a][0];                    // EE
            ua[3 * a + 1] = r[a][0] + r[a][2];
            ua[3 * a + 2] = r[a][2];
            ub[2 * a + 0] = r[a][1];                    // EO
            ub[2 * a + 1] = r[a][1];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) ua[9 + t] = g[1][1];      // OO
#pragma unroll
        for (int a = 0; a < 2; ++a) {                   // OE
            ub[6 + 3 * a + 0] = g[1][0];
            ub[6 + 3 * a + 1] = g[1][0] + g[1][2];
            ub[6 + 3 * a + 2] = g[1][2];
        }
    }
    const int64_t within = (hf * 64 + ql) * 4 + s;
    float* da = Uf + ((int64_t)qb * chunks + chunk) * (kS2Slots0 * 512) + within;
    float* db = Uf + (int64_t)qbs * chunks * (kS2Slots0 * 512) + ((int64_t)qb * chunks + chunk) * (kS2Slots1 * 512) + within;
#pragma unroll
    for (int t = 0; t < kS2Slots0; ++t) da[t * 512] = ua[t];
#pragma unroll
    for (int t = 0; t < kS2Slots1; ++t) db[t * 512] = ub[t];
}

template <int TYPE, bool XS>
__device__ __forceinline__ void s2w_dgrad_body(const float* __restrict__ g, const float* __restrict__ Ut, float* __restrict__ dx,
                                               const S2wParams& p, const int b, const int qb, float* __restrict__ UsBase,
                                               float* __restrict__ VsBase) {
    using P = S2wPoints<TYPE>;
    constexpr int NP = P::NP, NPAIR = P::NPAIR, NPS = P::NPS;
    constexpr int kUStage = NPS * 512;         // floats of a weight stage: [slot][half][q][s]
    constexpr int kVStage = NPAIR * 1024;      // floats of an input stage: [pair][half][s][t][2]
    constexpr int UQ = NPS / 2;                // 16-byte quads of the weight stage per thread

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid >> 1, wt = wid & 1;
    const bool first_block = b == 0;

    // ---- staging role: tile `lane` of the block, channels 2 wid and 2 wid + 1 of a chunk.  Every row of the 3x3 patch is ONE
    // 16-byte buffer load (columns 2tx-1 .. 2tx+2, the last one unused) whose per-lane offset never changes; the channel rides
    // in the scalar offset; rows outside the map and tiles outside the problem carry an out-of-range offset and read as zeros.
    int s_n, s_ty, s_tx;
    const bool s_valid = s2w_tile(p, b, lane, s_n, s_ty, s_tx);
    const int iy0 = 2 * s_ty - 1, ix0 = 2 * s_tx - 1;
    const int64_t HW = (int64_t)p.H * p.W;
    int cx = ix0;
    if (first_block) {              // (a window one float left of the tensor would wrap to an out-of-range offset: winograd_fused.hip)
        cx = ix0 < 0 ? 0 : ix0;
        if (cx > p.W - 4) cx = p.W - 4;
    }
    const int shift = ix0 - cx;
    unsigned rowv[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int iy = iy0 + r;
        const bool ok = s_valid && iy >= 0 && iy < p.H;
        rowv[r] = ok ? (unsigned)(((int64_t)s_n * p.K * HW + (int64_t)iy * p.W + cx) * 4) : 0x80000000u;
    }
    bool colok[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) colok[q] = ix0 + q >= 0 && ix0 + q < p.W;
    // uniform: no tile of the block has a patch column outside the map (main blocks right of the first block column)
    const bool x_interior = b < p.main_blocks && (b % p.blocks_x) != 0;
    const unsigned plane_bytes = (unsigned)(HW * 4);
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(g), 0, p.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t urs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(Ut + (int64_t)qb * p.chunks * kUStage), 0, (unsigned)(p.chunks * kUStage * 4), 0x00020000);
    const unsigned uoff = tid * 16;
    const int last = p.chunks - 1;

    f32x4 dreg[2][3];
    f32x4 ureg[UQ];
    f32x2 pv[2][NPAIR];
    float xsc[2] = {1.0f, 1.0f};

    auto load_x_row = [&](int chunk, int c2, int r) {          // one buffer load, no vector ALU
        int ch = chunk * kS2CK + 2 * wid + c2;
        if (ch > p.K - 1) ch = p.K - 1;                        // (a channel beyond K meets zero weights)
        const unsigned soff = (unsigned)ch * plane_bytes;
        dreg[c2][r] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, rowv[r], soff, 0));
        if (XS && r == 0) xsc[c2] = p.in_scale[(int64_t)(s_valid ? s_n : 0) * p.K + ch];
    };
    auto load_u1 = [&](int chunk, int j) {
        ureg[j] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(urs, uoff, (unsigned)(chunk * kUStage * 4 + j * 4096), 0));
    };
    auto transform = [&](auto mode_tag, int c2) {
        constexpr int MODE = decltype(mode_tag)::value;        // 0: all columns inside, 1: zero the outside columns, 2: shifted windows
        f32x4 d[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) d[r] = XS ? dreg[c2][r] * xsc[c2] : dreg[c2][r];
        if (MODE == 2) {
#pragma unroll
            for (int r = 0; r < 3; ++r) {
                const f32x4 l = d[r];
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    float t = 0.0f;
#pragma unroll
                    for (int k = -2; k <= 2; ++k)
                        if (q + k >= 0 && q + k < 4) t = (shift == k) ? l[q + k] : t;
                    d[r][q] = colok[q] ? t : 0.0f;
                }
            }
        }
        if (MODE == 1) {
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) d[r][q] = colok[q] ? d[r][q] : 0.0f;
        }
        const f32x4 e0 = d[0] - d[1], e2 = d[2] - d[1];       // even class along the rows: (a - b, b, c - b)
        if (TYPE == 0) {
            // EE: point 3 i + j = (row factor i) x (column factor j);  OO: 9 + 2 i' + j' = g[1 + i'][1 + j']
            pv[c2][0] = f32x2{e0[0] - e0[1], e0[1]};
            pv[c2][1] = f32x2{e0[2] - e0[1], d[1][0] - d[1][1]};
            pv[c2][2] = f32x2{d[1][1], d[1][2] - d[1][1]};
            pv[c2][3] = f32x2{e2[0] - e2[1], e2[1]};
            pv[c2][4] = f32x2{e2[2] - e2[1], d[1][1]};
            pv[c2][5] = f32x2{d[1][2], d[2][1]};
            pv[c2][6 < NPAIR ? 6 : 0] = f32x2{d[2][2], 0.0f};
        } else {
            // EO: point 2 i + j' = (row factor i) x column 1 + j';  OE: 6 + 3 i' + j = row 1 + i' x (column factor j)
            pv[c2][0] = f32x2{e0[1], e0[2]};
            pv[c2][1] = f32x2{d[1][1], d[1][2]};
            pv[c2][2] = f32x2{e2[1], e2[2]};
            pv[c2][3] = f32x2{d[1][0] - d[1][1], d[1][1]};
            pv[c2][4] = f32x2{d[1][2] - d[1][1], d[2][0] - d[2][1]};
            pv[c2][5] = f32x2{d[2][1], d[2][2] - d[2][1]};
        }
    };
    auto write_v = [&](int buf, int lo, int hi) {
        f32x2* vd = reinterpret_cast<f32x2*>(VsBase + buf * kVStage) + ((wid >> 1) * 4 + 2 * (wid & 1)) * 64 + lane;
#pragma unroll
        for (int pp = lo; pp < hi; ++pp)
#pragma unroll
            for (int c2 = 0; c2 < 2; ++c2) vd[pp * 512 + c2 * 64] = pv[c2][pp];
    };
    auto write_u = [&](int buf) {
        f32x4* ud = reinterpret_cast<f32x4*>(UsBase + buf * kUStage) + tid;
#pragma unroll
        for (int j = 0; j < UQ; ++j) ud[j * kBlock] = ureg[j];
    };

    f32x16 acc[NP];
#pragma unroll
    for (int xi = 0; xi < NP; ++xi)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[xi][r] = 0.0f;

    // One pass over the staged chunk `cur`: NPAIR groups of eight MFMAs (a pair of points each; type A's last group holds one
    // point), the operands of the next group read while this one runs.  Riding in the groups: the weights of chunk c1 (two quads
    // per group from group 0 on, written to LDS in the last one), the transform of chunk c1's patches (groups 0 / 1: they were
    // loaded during the PREVIOUS pass) and their LDS writes, and the six patch-row loads of chunk c2n = c1 + 1, one per group.
    auto pass = [&](auto mode_tag, int cur, int c1, int c2n) {
        const f32x4* ua = reinterpret_cast<const f32x4*>(UsBase + cur * kUStage) + (half * 64 + wm * 32 + l31);
        const f32x2* vb = reinterpret_cast<const f32x2*>(VsBase + cur * kVStage) + (half * 256 + wt * 32 + l31);
        f32x4 a0 = ua[0], a1 = ua[128];
        f32x2 bq[4] = {vb[0], vb[64], vb[128], vb[192]};
#pragma unroll
        for (int pp = 0; pp < NPAIR; ++pp) {
            const bool two = 2 * pp + 1 < NP;
            f32x4 a0n = a0, a1n = a1;
            f32x2 bn[4] = {bq[0], bq[1], bq[2], bq[3]};
            if (pp + 1 < NPAIR) {
                a0n = ua[(2 * pp + 2) * 128];
                if (2 * pp + 3 < NP) a1n = ua[(2 * pp + 3) * 128];
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) bn[s4] = vb[(pp + 1) * 512 + s4 * 64];
            }
#pragma unroll
            for (int j = 2 * pp; j < 2 * pp + 2; ++j)
                if (j < UQ) load_u1(c1, j);
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) {
                acc[2 * pp] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0[s4], bq[s4][0], acc[2 * pp], 0, 0, 0);
                if (two) acc[2 * pp + (two ? 1 : 0)] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1[s4], bq[s4][1], acc[2 * pp + (two ? 1 : 0)], 0, 0, 0);
            }
            if (pp == 0) transform(mode_tag, 0);
            if (pp == 1) transform(mode_tag, 1);
#pragma unroll
            for (int li = 0; li < 6; ++li)                    // row load li rides in group 1 + li (type B: the last two share group 5)
                if (1 + (li < NPAIR - 1 ? li : NPAIR - 2) == pp) load_x_row(c2n, li / 3, li % 3);
            if (pp == 2) write_v(cur ^ 1, 0, NPAIR / 2);
            if (pp == 3) write_v(cur ^ 1, NPAIR / 2, NPAIR);
            if (pp == NPAIR - 1) write_u(cur ^ 1);
            a0 = a0n;
            a1 = a1n;
#pragma unroll
            for (int s4 = 0; s4 < 4; ++s4) bq[s4] = bn[s4];
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    auto run = [&](auto mode_tag) {
#pragma unroll
        for (int li = 0; li < 6; ++li) load_x_row(0, li / 3, li % 3);
#pragma unroll
        for (int j = 0; j < UQ; ++j) load_u1(0, j);
        transform(mode_tag, 0);
        transform(mode_tag, 1);
#pragma unroll
        for (int li = 0; li < 6; ++li) load_x_row(last < 1 ? last : 1, li / 3, li % 3);
        write_v(0, 0, NPAIR);
        write_u(0);
        __syncthreads();
        int cur = 0;
        for (int chunk = 0; chunk < p.chunks; ++chunk) {
            // (beyond the last chunk the staging repeats the last one into the buffer nobody reads: one instantiation of the pass)
            pass(mode_tag, cur, chunk + 1 < last ? chunk + 1 : last, chunk + 2 < last ? chunk + 2 : last);
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

    // ---- output transform, lane-local: acc[point][r] of (q = ... r ..., tile = wt * 32 + l31) -> this type's eight elements of the
    // tile's 4x4 block: v[k] at (row, column) = (RO[k], CO[k])
    int o_n, o_ty, o_tx;
    if (!s2w_tile(p, b, wt * 32 + l31, o_n, o_ty, o_tx)) return;
    const int64_t OHW = (int64_t)p.OH * p.OW;
    const int oy = 4 * o_ty, ox = 4 * o_tx;
    float* ybase = dx + (int64_t)o_n * p.Q * OHW + (int64_t)oy * p.OW + ox;
    const float* osc = p.out_scale ? p.out_scale + (int64_t)o_n * p.Q : nullptr;
    constexpr int ROA[8] = {0, 0, 2, 2, 1, 1, 3, 3}, COA[8] = {0, 2, 0, 2, 1, 3, 1, 3};
    constexpr int ROB[8] = {0, 0, 2, 2, 1, 1, 3, 3}, COB[8] = {1, 3, 1, 3, 0, 2, 0, 2};
    auto values_of = [&](int r, float (&v)[8]) {
        if (TYPE == 0) {
            float t0[3], t1[3];
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                t0[j] = acc[j][r] + acc[3 + j][r];
                t1[j] = acc[3 + j][r] + acc[6 + j][r];
            }
            v[0] = t0[0] + t0[1]; v[1] = t0[1] + t0[2]; v[2] = t1[0] + t1[1]; v[3] = t1[1] + t1[2];
            v[4] = acc[9][r]; v[5] = acc[10][r]; v[6] = acc[11][r]; v[7] = acc[12][r];
        } else {
            v[0] = acc[0][r] + acc[2][r]; v[1] = acc[1][r] + acc[3][r];
            v[2] = acc[2][r] + acc[4][r]; v[3] = acc[3][r] + acc[5][r];
            v[4] = acc[6][r] + acc[7][r]; v[5] = acc[7][r] + acc[8][r];
            v[6] = acc[9][r] + acc[10][r]; v[7] = acc[10][r] + acc[11][r];
        }
    };
    // uniform: every tile of the block has its whole 4x4 inside dx and every channel of the block exists
    const bool whole = b < p.main_blocks && (qb + 1) * kS2Q <= p.Q;
    // the lane's sixteen out_scale factors in one go, landed before the first store (inside the store loops every row waited for
    // its own load and -- vmcnt counts stores too -- for the previous row's stores: csrc/winograd_fused.hip's epilogue)
    float psv[16];
    if (osc) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qb * kS2Q + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            psv[r] = osc[q < p.Q ? q : p.Q - 1];
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) psv[r] = 1.0f;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0)
    if (whole) {
        int off[8];
#pragma unroll
        for (int k = 0; k < 8; ++k) off[k] = (TYPE == 0 ? ROA[k] : ROB[k]) * p.OW + (TYPE == 0 ? COA[k] : COB[k]);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qb * kS2Q + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v[8];
            values_of(r, v);
            const float ps = psv[r];
            float* yp = ybase + (int64_t)q * OHW;
#pragma unroll
            for (int k = 0; k < 8; ++k) yp[off[k]] = v[k] * ps;
        }
    } else {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int q = qb * kS2Q + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            float v[8];
            values_of(r, v);
            const float ps = psv[r];
            float* yp = ybase + (int64_t)q * OHW;
#pragma unroll
            for (int k = 0; k < 8; ++k) {
                const int ro = TYPE == 0 ? ROA[k] : ROB[k], co = TYPE == 0 ? COA[k] : COB[k];
                if (q < p.Q && oy + ro < p.OH && ox + co < p.OW) yp[(int64_t)ro * p.OW + co] = v[k] * ps;
            }
        }
    }
}

// Workgroup order: ids go round the 8 XCDs (id % 8), each with its own L2.  An XCD walks a contiguous eighth of the tile blocks,
// type A and type B of a tile block back to back (they read the same patches and write the two halves of the same output
// lines), all XCDs on the same channel block at a time (its prepared weights stay in every L2).
template <bool XS>
__global__ __launch_bounds__(kBlock, 1) void s2w_dgrad_kernel(const float* __restrict__ g, const float* __restrict__ Uf,
                                                              float* __restrict__ dx, const S2wParams p) {
    __shared__ float Us[2 * kS2Slots0 * 512];
    __shared__ float Vs[2 * (kS2Slots0 / 2) * 1024];
    const int lin = blockIdx.x;
    const int xcd = lin & 7, j = lin >> 3;
    const int type = j & 1, jj = j >> 1;
    const int qb = jj / p.per;
    const int b = xcd * p.per + (jj - qb * p.per);
    if (b >= p.blocks) return;
    if (type == 0)
        s2w_dgrad_body<0, XS>(g, Uf, dx, p, b, qb, Us, Vs);
    else
        s2w_dgrad_body<1, XS>(g, Uf + (int64_t)p.qbs * p.chunks * (kS2Slots0 * 512), dx, p, b, qb, Us, Vs);
}

inline int64_t s2w_weight_floats(int64_t q, int64_t k) {
    return ceil_div64(q, kS2Q) * ceil_div64(k, kS2CK) * (kS2SlotsAll * 512);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int64_t sae_s2wino_weights_floats(int64_t cout, int64_t cin) {
    if (cout < 1 || cin < 1) return 0;
    return s2w_weight_floats(cout, cin);
}

extern "C" int sae_s2wino_weights_f32(const float* w, const float* row_scale, const float* col_scale, float* uf, int64_t cout,
                                      int64_t cin, int64_t w_stride_out, int64_t w_stride_in, int32_t flip, float alpha,
                                      sae_stream_t stream) {
    sae::clear_stale_error();
    if (cout < 1 || cin < 1 || s2w_weight_floats(cout, cin) >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_s2wino_weights_f32: bad shape");
    if (!w || !uf) return fail(SAE_EINVAL, "sae_s2wino_weights_f32: null tensor");
    if (!aligned16(uf)) return fail(SAE_EINVAL, "sae_s2wino_weights_f32: uf must be 16-byte aligned");
    const int chunks = (int)ceil_div64(cin, kS2CK);
    const int64_t work = ceil_div64(cout, kS2Q) * kS2Q * chunks * kS2CK;
    hipLaunchKernelGGL(s2w_wprep_kernel, dim3((unsigned)ceil_div64(work, kBlock)), dim3(kBlock), 0, (hipStream_t)stream, w, uf,
                       (int)cout, (int)cin, chunks, w_stride_out, w_stride_in, flip ? 1 : 0, alpha, row_scale, col_scale);
    return check_launch("sae_s2wino_weights_f32");
}

extern "C" int sae_s2wino_dgrad_f32(const float* g, const float* g_scale, const float* uf, const float* out_scale, float* dx,
                                    int64_t n, int64_t cin, int64_t cout, int64_t h, int64_t w, sae_stream_t stream) {
    sae::clear_stale_error();
    if (n < 0 || cin < 1 || cout < 1 || h < 2 || w < 4 || (h & 1) || (w & 1) || h >= 16384 || w >= 16384 || cin >= (1 << 24) ||
        cout >= (1 << 24) || n >= (1 << 24))
        return fail(SAE_EINVAL, "sae_s2wino_dgrad_f32: the small side must have even sides, rows of at least 4 floats, got %lld x %lld",
                    (long long)h, (long long)w);
    if (n == 0) return SAE_OK;
    if (!g || !uf || !dx) return fail(SAE_EINVAL, "sae_s2wino_dgrad_f32: null tensor");
    if (n * cin * h * w * 4 >= ((int64_t)1 << 31))
        return fail(SAE_EINVAL, "sae_s2wino_dgrad_f32: input of %lld bytes; the kernel addresses it with 32-bit byte offsets (< 2 GiB)",
                    (long long)(n * cin * h * w * 4));
    if (!aligned16(uf)) return fail(SAE_EINVAL, "sae_s2wino_dgrad_f32: uf must be 16-byte aligned");
    S2wParams p;
    p.N = (int)n; p.K = (int)cin; p.H = (int)h; p.W = (int)w; p.Q = (int)cout;
    p.OH = 2 * (int)h + 1; p.OW = 2 * (int)w + 1;
    p.MH = (int)h / 2; p.MW = (int)w / 2;
    int bw = ilog2_ceil(p.MW);
    if (bw > 4) bw = 4;
    int bh = ilog2_ceil(p.MH);
    if (bw + bh > 6) bh = 6 - bw;
    p.bw_log2 = bw; p.bh_log2 = bh;
    const int BN = kS2T >> (bw + bh);
    p.blocks_x = ceil_div(p.MW, 1 << bw);
    p.blocks_y = ceil_div(p.MH, 1 << bh);
    const int64_t main_blocks = (int64_t)p.blocks_x * p.blocks_y * ceil_div64(n, BN);
    const int64_t strip_blocks = ceil_div64(n * (p.MW + p.MH + 1), kS2T);
    const int64_t qbs = ceil_div64(cout, kS2Q);
    const int64_t per = ceil_div64(main_blocks + strip_blocks, 8);
    if (per * 8 * 2 * qbs >= ((int64_t)1 << 31)) return fail(SAE_EINVAL, "sae_s2wino_dgrad_f32: too many tile blocks");
    p.main_blocks = (int)main_blocks; p.blocks = (int)(main_blocks + strip_blocks); p.per = (int)per; p.qbs = (int)qbs;
    p.chunks = (int)ceil_div64(cin, kS2CK);
    p.in_bytes = (unsigned)(n * cin * h * w * 4);
    p.in_scale = g_scale; p.out_scale = out_scale;
    const dim3 grid((unsigned)(per * 8 * 2 * qbs));
    const hipStream_t st = (hipStream_t)stream;
    if (g_scale)
        hipLaunchKernelGGL(s2w_dgrad_kernel<true>, grid, dim3(kBlock), 0, st, g, uf, dx, p);
    else
        hipLaunchKernelGGL(s2w_dgrad_kernel<false>, grid, dim3(kBlock), 0, st, g, uf, dx, p);
    return check_launch("sae_s2wino_dgrad_f32");
}
