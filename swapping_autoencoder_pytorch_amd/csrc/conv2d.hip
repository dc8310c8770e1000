// Dense conv2d family (forward / dgrad / wgrad, incl. the stride-2 transposed form) as implicit
// GEMMs on the gfx950 fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact fp32, 157 TFLOP/s peak).
//
// Replaces the ATen calls F.conv2d / F.conv_transpose2d and their backward at
// models/networks/stylegan2_layers.py:136,175,182,306,315,321 (EqualConv2d, EqualLinear on 4-D
// input, ModulatedConv2d).  The reference's "modulated conv" is a dense conv with batch-shared
// weights, so batch folds into the GEMM N dimension here instead of a
// groups = batch grouped conv.
//
// GEMM view (NCHW, W contiguous):  D[m][pixel] = sum_{c,tap} A[m][(c,tap)] * B[(c,tap)][pixel]
//   A = weights, re-laid once per call by conv_wprep_kernel into wp[tap][c][m] (m contiguous,
//       zero padded to the tile, equalised-lr scale `alpha` and the dgrad flip/transposition
//       folded in), staged to LDS as As[tap][c][m] with 16-byte global loads;
//   B = never materialised: for a tile of 32*NI*WN output pixels the input PATCH (tile + halo,
//       zero filled outside the image) of CK channels is staged to LDS once and each of the
//       KS*KS taps reads it at a shifted offset, so every staged input element feeds 9 MFMAs
//       rows (3x3) and HBM/L2 traffic for B is the bare input, not 9x im2col;
//   MFMA operands (one fp32 VGPR each): lane l supplies A[i = l&31][k = l>>5] and
//       B[k = l>>5][j = l&31]; the two k of one instruction are two input channels at the same
//       tap, so the 32 lanes of a half-wave read 32 consecutive pixels (conflict-free
//       ds_read_b32; for stride 2 the patch columns are stored de-interleaved even|odd so the
//       stride-2 pixel walk is still unit-stride in LDS).
// A workgroup is 4 waves (one per SIMD); a wave owns MI x NI 32x32 accumulator tiles.  The next
// K-chunk's global loads are issued before the current chunk's MFMAs (register staging, write
// after the barrier) so HBM/L2 latency hides under the matrix pipe.
//
// Stride-2 dgrad / transposed conv ("tr" kernel): output pixels are split by parity class
// (oy&1, ox&1); class (ey,ex) only receives taps with ky = ey, kx = ex (mod 2) -> 4/2/2/1 taps,
// no multiplications by inserted zeros.  A wave's four N-tiles are the four classes of the
// same 32 half-resolution positions, so every tap issues exactly one useful MFMA per M-tile.
//
// wgrad: D[m][(c,tap)] = sum_pixels gy[m][pixel] * x[c][pixel shifted by tap]; K = pixels is
// huge and M x C*taps small, so the pixel range is split over workgroups (split-K), each
// leaving an fp32 slab; a fixed-order second kernel sums the slabs, applies alpha and scatters
// to the parameter layout (deterministic, no atomics).
#include "sae_common.h"

#include <cstdlib>
#include <type_traits>

namespace sae {
namespace {

// (tuning_knob: sae_common.h)
// SAE_TRACE_DISPATCH=1 (tuning builds): one stderr line per launch decision the tests want to see
#define SAE_TRACE(...)                                                         \
    do {                                                                       \
        static const int trace_knob = tuning_knob("SAE_TRACE_DISPATCH", 0);    \
        if (trace_knob) { fprintf(stderr, "sae-dispatch " __VA_ARGS__); fputc('\n', stderr); } \
    } while (0)

#include "conv2d_gather.inc"

#ifdef SAE_TUNING
#include "tuning/conv2d_ws.inc"
#endif

#include "conv2d_bx.inc"

#include "conv2d_bx8.inc"

#ifdef SAE_TUNING
#include "tuning/conv2d_f8.inc"
#endif

#include "conv2d_transposed.inc"

#include "conv2d_wgrad.inc"

#ifndef SAE_IGEMM_SPLIT_MODEL_DEFAULT
#define SAE_IGEMM_SPLIT_MODEL_DEFAULT 1
#endif
// Launch plan of a forward-type gather: tile shape, K split and workspace layout
// [ wp : taps*Cp*Mp ][ slabs : ksplit * round4(N*M*OH*OW) ]  (slabs only when ksplit > 1).
struct GatherPlan {
    FwdShape sh; int Mp, Cp, taps; int tw_log2, th_log2, tiles_x, tiles_y, tiles_n; int ksplit, cps;
    bool bx;   // bf16-split arithmetic (3x3 stride 1, 128x128 tile)
    bool bx8;  // ... on the 8-wave LDS-DMA kernel (128 x 256-pixel tile)
    bool f8;   // exact fp32 on the 8-wave LDS-DMA kernel (conv_igemm_f8_kernel)
    int64_t wp_floats, out_floats4, ws_floats;
};
GatherPlan gather_plan(int N, int cin, int mout, int OH, int OW, int ks, int stride, bool scatter) {
    GatherPlan g{};
    g.sh = fwd_shape(mout, ks, stride, OW);
    g.Mp = round_up(mout, g.sh.bm);
    g.Cp = round_up(cin, g.sh.ck);
    g.taps = ks * ks;
    g.bx8 = false;
    static const int bx8_knob = tuning_knob("SAE_BX8", 1);
    if (bx8_knob && conv_math() == 1 && ks == 3 && stride == 1 && g.sh.cfg == 0) {
        int twl, thl;
        pick_tile(256, OH, OW, 32, &twl, &thl);
        const int tw8 = 1 << twl, th8 = 1 << thl, tn8 = 256 / (tw8 * th8);
        if (tn8 * (th8 + 2) * (tw8 + 2) <= 512) { g.bx8 = true; g.sh.bn = 256; }   // one patch position per thread
    }
    g.f8 = false;
    // measured on MI355X (same box, tools/kb_subset.py): 121.9 vs 125.8 TFLOP/s at 128 -> 128 @256^2 B=16 and 126.4 vs 130.0
    // at 512 -> 512 @64^2 for this kernel vs the 4-wave register-staged one: one workgroup per CU loses the overlap two
    // independent workgroups give (their MFMAs are what covers each other's staging phases).  Off by default.
#ifdef SAE_TUNING      // the product build does not contain the kernel
    static const int f8_knob = tuning_knob("SAE_F8", 0);
#else
    constexpr int f8_knob = 0;
#endif
    if (f8_knob && conv_math() == 0 && ks == 3 && stride == 1 && g.sh.cfg == 0 && !scatter) {
        int twl, thl;
        pick_tile(256, OH, OW, 32, &twl, &thl);
        const int tw8 = 1 << twl, th8 = 1 << thl, tn8 = 256 / (tw8 * th8);
        // one patch position per thread, and enough 256-pixel tiles to fill the chip once (small layers keep the
        // 4-wave kernel with its split-K path)
        const int64_t tiles = (int64_t)ceil_div(OW, tw8) * ceil_div(OH, th8) * ceil_div(N, tn8) * (g.Mp / 128);
        static const int f8_min_tiles = tuning_knob("SAE_F8_MIN_TILES", 256);   // tests: 1
        if (tn8 * (th8 + 2) * (tw8 + 2) <= 512 && tiles >= f8_min_tiles) { g.f8 = true; g.sh.bn = 256; }
    }
    pick_tile(g.sh.bn, OH, OW, 32, &g.tw_log2, &g.th_log2);
    const int tw = 1 << g.tw_log2, th = 1 << g.th_log2, tn = g.sh.bn / (tw * th);
    g.tiles_x = ceil_div(OW, tw);
    g.tiles_y = ceil_div(OH, th);
    g.tiles_n = ceil_div(N, tn);
    const int blocks = g.tiles_x * g.tiles_y * g.tiles_n * (g.Mp / g.sh.bm);
    const int nchunks = g.Cp / g.sh.ck;
    g.ksplit = 1;
    g.cps = nchunks;
    // few workgroups and a long K loop (the 4x4..16x16 tails of D / Dpatch): split the input
    // channels over blockIdx.z so the launch still fills the 256 CUs
    if (!scatter && blocks < 192 && nchunks >= 8 && ((int64_t)N * mout * OH * OW) % 4 == 0) {
        int k = 512 / blocks;
        if (k > 8) k = 8;
        if (k > nchunks / 4) k = nchunks / 4;
        if (k >= 2) {
            g.cps = ceil_div(nchunks, k);
            g.ksplit = ceil_div(nchunks, g.cps);
        }
    }
    // Mid-size launches of the exact-fp32 3x3 kernels: a few ROUNDS of workgroups on the 512 slots of the chip (two per CU), the
    // last one mostly empty -- 512 -> 512 @32^2 with 24 images is 768 workgroups = two rounds for 1.5 rounds of work (0.67 of peak
    // where the same layer with 16 images, exactly one round, reaches 0.83).  Splitting K in two makes it three rounds of half
    // the length.  Priced in microseconds: a round of one chunk ~ 7.7 us (144 MFMAs x 64 cycles for each of the two waves of a
    // SIMD), the slabs and their fixed-order reduction ~ (2 k + 1) x output bytes at 4 TB/s + a launch.
    static const int rounds_knob = tuning_knob("SAE_IGEMM_ROUND_SPLIT", 1);
    if (rounds_knob && g.ksplit == 1 && !scatter && conv_math() == 0 && ks == 3 && blocks >= 192 && blocks < 4096 && nchunks >= 16 &&
        ((int64_t)N * mout * OH * OW) % 4 == 0) {
        const double out_bytes = 4.0 * (double)N * mout * OH * OW;
        auto cost = [&](int k) {
            const double rounds = (double)ceil_div(blocks * k, 512);
            return rounds * ceil_div(nchunks, k) * 7.7 + (k > 1 ? (2 * k + 1) * out_bytes / 4e6 + 5.0 : 0.0);
        };
        int best = 1;
        for (int k = 2; k <= 4; ++k)
            if (cost(k) < cost(best)) best = k;
        if (best > 1 && cost(best) < 0.93 * cost(1)) {
            g.cps = ceil_div(nchunks, best);
            g.ksplit = ceil_div(nchunks, g.cps);
        }
    }
    // One price list for both rules above (exact-fp32 3x3, fewer than 4096 workgroups): k = 1 ... 16 slices, a last round with at
    // most one workgroup per CU priced at 0.6 of a round (a workgroup that has its CU to itself runs that much faster: 512 -> 512
    // @32^2 with 8 images, 256 workgroups, reaches 0.76 of peak where two per CU share 0.83)
    // Same box, tools/ab_conv.py (profiles/r4_ab_plan_models.txt): launches of fewer than 192 workgroups gain (512 -> 512 @8^2, 24
    // images: fwd 69 -> 78, dgrad 78 -> 88 TFLOP/s; @4^2, 40 images: 44 -> 49, 51 -> 59), the 192-workgroup ones (k 2 -> 4) lose 1 - 2 %:
    // the list replaces the first rule only.  (knob 2: both)
    static const int model_knob = tuning_knob("SAE_IGEMM_SPLIT_MODEL", SAE_IGEMM_SPLIT_MODEL_DEFAULT);
    if (model_knob && !scatter && conv_math() == 0 && ks == 3 && blocks < (model_knob == 2 ? 4096 : 192) && nchunks >= 8 &&
        ((int64_t)N * mout * OH * OW) % 4 == 0) {
        const double out_bytes = 4.0 * (double)N * mout * OH * OW;
        auto cost = [&](int k) {
            const int cps = ceil_div(nchunks, k), eff = ceil_div(nchunks, cps);
            const int64_t wgs = (int64_t)blocks * eff, full = wgs / 512, rem = wgs - full * 512;
            const double rounds = (double)full + (rem > 0 ? (rem <= 256 ? 0.6 : 1.0) : 0.0);
            return rounds * cps * 9.3 + (eff > 1 ? (2 * eff + 1) * out_bytes / 4e6 + 5.0 : 0.0);
        };
        int best = g.ksplit;
        const int kmax = nchunks / 4 < 16 ? nchunks / 4 : 16;
        for (int k = 1; k <= kmax; ++k)
            if (cost(k) < cost(best)) best = k;
        if (best != g.ksplit && cost(best) < 0.93 * cost(g.ksplit)) {
            g.cps = ceil_div(nchunks, best);
            g.ksplit = ceil_div(nchunks, g.cps);
        }
    }
    g.wp_floats = (int64_t)g.taps * g.Cp * g.Mp;
    // bf16x6: the 128-row tile (8-wave or 4-wave kernel) and, on the 4-wave kernel, the 64- and 32-row tiles of the
    // narrow layers (cfg 1: 64 x 256, cfg 4: 32 x 256)
    static const int bx_narrow_knob = tuning_knob("SAE_BX_NARROW", 1);
    g.bx = conv_math() == 1 && ks == 3 && stride == 1 &&
           (g.sh.cfg == 0 || (bx_narrow_knob && (g.sh.cfg == 1 || g.sh.cfg == 4)));
    if (conv_math() == 1 && ks == 3 && stride == 2 && g.sh.cfg == 6) g.bx = true;
    if (g.bx) g.wp_floats = (int64_t)27 * g.Mp * (g.Cp / 8) * 4;   // 16-byte cells: [tap][split][m] per 8 channels
    g.out_floats4 = ((int64_t)N * mout * OH * OW + 3) / 4 * 4;
    g.ws_floats = g.wp_floats + (g.ksplit > 1 ? g.ksplit * g.out_floats4 : 0);
    return g;
}

// ---- forward-type launch (regular gather) ------------------------------------------------------
template <int KS, int S>
int launch_igemm(const float* x, const float* wp, float* y, IgemmParams p, const GatherPlan& g, hipStream_t s) {
    const FwdShape& sh = g.sh;
    p.tw_log2 = g.tw_log2; p.th_log2 = g.th_log2;
    p.tiles_x = g.tiles_x; p.tiles_y = g.tiles_y; p.tiles_n = g.tiles_n;
    p.chunks_per_split = g.cps;
    const int tw = 1 << p.tw_log2, th = 1 << p.th_log2, tn = sh.bn / (tw * th);
    const int ph = (KS == 1) ? th : (th - 1) * S + KS, pw = (KS == 1) ? tw : (tw - 1) * S + KS;
    const int cap = (KS == 1) ? sh.bn : (S == 1 ? (9 * sh.bn) / 4 : (41 * sh.bn) / 8);
    if (tn * ph * pw > cap) return fail(SAE_EINVAL, "conv igemm: patch %d exceeds LDS cap %d", tn * ph * pw, cap);
    const dim3 grid((unsigned)(p.tiles_x * p.tiles_y * p.tiles_n), (unsigned)(p.Mp / sh.bm), (unsigned)(p.batch ? p.batch : g.ksplit));
    constexpr int CK = (KS == 1) ? 32 : 8;
    constexpr int CK2 = (KS == 1) ? 16 : 8;
    if constexpr (KS == 3 && S == 1) {
#ifdef SAE_TUNING
        if (g.f8) {
            if (p.in_scale) hipLaunchKernelGGL((conv_igemm_f8_kernel<2, 2, 2, 4, true>), grid, dim3(kBlock8), 0, s, x, wp, y, p);
            else hipLaunchKernelGGL((conv_igemm_f8_kernel<2, 2, 2, 4, false>), grid, dim3(kBlock8), 0, s, x, wp, y, p);
            return SAE_OK;
        }
#endif
        if (g.bx8) {
            hipLaunchKernelGGL((conv_igemm_bx8_kernel<2, 2, 2, 4>), grid, dim3(kBlock8), 0, s, x,
                               reinterpret_cast<const u32x4*>(wp), y, p);
            return SAE_OK;
        }
        if (g.bx) {
            const u32x4* wb = reinterpret_cast<const u32x4*>(wp);
            if (sh.cfg == 1) hipLaunchKernelGGL((conv_igemm_bx_kernel<1, 2, 2, 1, 4>), grid, dim3(kBlock), 0, s, x, wb, y, p);
            else if (sh.cfg == 4) hipLaunchKernelGGL((conv_igemm_bx_kernel<1, 1, 2, 1, 4>), grid, dim3(kBlock), 0, s, x, wb, y, p);
            else hipLaunchKernelGGL((conv_igemm_bx_kernel<1, 2, 2, 2, 2>), grid, dim3(kBlock), 0, s, x, wb, y, p);
            return SAE_OK;
        }
    }
    if constexpr (KS == 3 && S == 2) {
        if (g.bx) {     // 64 x 128 tile (cfg 6): the stride-2 patch is four times the pixels, LDS allows no more
            hipLaunchKernelGGL((conv_igemm_bx_kernel<2, 1, 2, 2, 2>), grid, dim3(kBlock), 0, s, x,
                               reinterpret_cast<const u32x4*>(wp), y, p);
            return SAE_OK;
        }
    }
    // 1x1 stride 1, pad 0 with quad staging: the tile's own pixels, rows a multiple of 16 bytes (tw >= 4 always is)
    static const int quad1_knob = tuning_knob("SAE_IGEMM_QUAD", 1);
    const bool quad1 = quad1_knob && KS == 1 && S == 1 && p.pad == 0 && p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0;
    // 3x3 stride 1 on the 256-pixel tiles (64 x 256, 32 x 256): quad staging when the widened patch fits 128 quads
    const bool quad3w = quad1_knob && KS == 3 && S == 1 && sh.bn == 256 && p.W % 4 == 0 && p.pad <= 4 &&
                        (reinterpret_cast<uintptr_t>(x) & 15) == 0 && tn * ph * ((tw + 8) / 4) <= 128;
    switch (sh.cfg) {
#define SAE_IGEMM(...)                                                                                          \
    do {                                                                                                        \
        if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<__VA_ARGS__, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);  \
        else hipLaunchKernelGGL((conv_igemm_kernel<__VA_ARGS__, false>), grid, dim3(kBlock), 0, s, x, wp, y, p);            \
    } while (0)
        case 4:
            if constexpr (KS == 3 && S == 1) {
                if (quad3w) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 1, 2, 1, 4, 8, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 1, 2, 1, 4, 8, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
                SAE_IGEMM(3, 1, 1, 2, 1, 4, 8);
                break;
            } else {
                return fail(SAE_EINVAL, "conv igemm: 32x256 tile is 3x3 stride-1 only");
            }
        case 3:
            if constexpr (KS == 3 && S == 1) {
                SAE_IGEMM(3, 1, 2, 4, 2, 2, 8);
                break;
            } else {
                return fail(SAE_EINVAL, "conv igemm: 128x256 tile is 3x3 stride-1 only");
            }
        case 0:
            if constexpr (KS == 3 && S == 1) {
                // quad staging (see conv_igemm_kernel): rows a multiple of 16 bytes, 16-byte aligned tensor, and the
                // widened patch within 64 quads per channel (always for tiles >= 16 wide)
                static const int quad_knob = tuning_knob("SAE_IGEMM_QUAD", 1);
                const int qn = tn * ph * ((tw + 8) / 4);
                if (quad_knob && p.W % 4 == 0 && (reinterpret_cast<uintptr_t>(x) & 15) == 0 && qn <= sh.bn / 2 && p.pad <= 4) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 2, 2, 2, 2, 8, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 2, 2, 2, 2, 8, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
            }
            if constexpr (KS == 3 && S == 2) {
                // stride 2, pad 0: 4-byte aligned quads from column 2 x0 on, (2 tw + 4) / 4 per row
                const int qn2 = tn * ph * ((pw + 3) / 4);
                if (quad1_knob && p.pad == 0 && p.W >= 4 && qn2 <= 5 * sh.bn / 4) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<3, 2, 2, 2, 2, 2, 8, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<3, 2, 2, 2, 2, 2, 8, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
            }
            if constexpr (KS == 1 && S == 1) {
                if (quad1) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<1, 1, 2, 2, 2, 2, 32, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<1, 1, 2, 2, 2, 2, 32, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
            }
            SAE_IGEMM(KS, S, 2, 2, 2, 2, CK);
            break;
        case 1:
            if constexpr (KS == 3 && S == 1) {
#ifdef SAE_TUNING      // the product build does not contain the kernel (a recorded experiment, 4 - 5 % slower)
                static const int ws_knob = tuning_knob("SAE_WS", 0);
                if (quad3w && ws_knob && !p.in_scale && p.vec_store && g.ksplit == 1 && !p.residual && !p.noise) {
                    SAE_TRACE("ws 64x256 tiles=%u x %u chunks=%d", grid.x, grid.y, p.Cp / 8);
                    hipLaunchKernelGGL((conv_igemm_ws_kernel<2, 2, 1, 4>), grid, dim3(kBlockWs), 0, s, x, wp, y, p);
                    break;
                }
#endif
                if (quad3w) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 2, 2, 1, 4, 8, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<3, 1, 2, 2, 1, 4, 8, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
            }
            if constexpr (KS == 1 && S == 1) {
                if (quad1) {
                    if (p.in_scale) hipLaunchKernelGGL((conv_igemm_kernel<1, 1, 2, 2, 1, 4, 32, true, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    else hipLaunchKernelGGL((conv_igemm_kernel<1, 1, 2, 2, 1, 4, 32, false, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);
                    break;
                }
            }
            SAE_IGEMM(KS, S, 2, 2, 1, 4, CK);
            break;
        default:
            if constexpr (S == 1)
                SAE_IGEMM(KS, 1, 1, 4, 1, 4, CK2);
            else
                return fail(SAE_EINVAL, "conv igemm: 32x512 tile is stride-1 only");
            break;
#undef SAE_IGEMM
    }
    return SAE_OK;
}

// per-channel weight factors of a launch (either may be null): rs_m along its output axis, rs_c along its contraction axis
struct WScale { const float* rs_m; const float* rs_c; };

int run_wprep(const float* w, float* wp, int M, int C, int Mp, int Cp, int taps, int64_t sm, int64_t sc, int flip,
              float alpha, hipStream_t s, WScale ws = WScale{nullptr, nullptr}) {
    const int64_t total = (int64_t)taps * Cp * Mp;
    int64_t blocks = ceil_div64(total, kBlock);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_wprep_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, w, wp, M, C, Mp, Cp, taps, sm,
                       sc, flip, alpha, ws.rs_m, ws.rs_c);
    return SAE_OK;
}

int run_wprep_bx(const float* w, float* wpb, int M, int C, int Mp, int Cp, int BM, int64_t sm, int64_t sc, int flip,
                 float alpha, hipStream_t s, WScale ws = WScale{nullptr, nullptr}) {
    const int64_t total = (int64_t)Mp * (Cp / 8) * 9;
    int64_t blocks = ceil_div64(total, kBlock);
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_wprep_bx_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, w,
                       reinterpret_cast<u32x4*>(wpb), M, C, Mp, Cp, BM, sm, sc, flip, alpha, ws.rs_m, ws.rs_c);
    return SAE_OK;
}

// ---- prepared weights (sae_conv2d_desc::prepped, sae_conv2d_wprep_*) -------------------------------------------------------
// A launch re-lays its weights (run_wprep / run_wprep_bx) into the head of its workspace.  The layout depends on the launch's
// tile shape, not on the batch, and the weights change once per optimiser step, so a caller may keep the re-laid copy:
//   sae_conv2d_wprep_f32 writes it into a caller-owned buffer, later calls hand it back through the descriptor.
// The request of the call at hand lives in a thread-local record set by the entry point (the calls are re-entrant per thread).
struct PrepRequest {
    const float* use;        // prepared weights to use instead of re-laying (null: re-lay into the workspace)
    int64_t use_floats;
    int64_t use_layout;      // sae_conv2d_wprep_layout() of the launch they were prepared for
    float* make;             // wprep-only call: write the layout here and launch nothing else
    int64_t make_floats;
    int64_t* want_floats;    // query: floats of the layout (0 when the path has none) ...
    int64_t* want_layout;    // ... and its identity
};
thread_local PrepRequest t_prep = {nullptr, 0, 0, nullptr, 0, nullptr, nullptr};
struct PrepGuard {
    PrepRequest saved;
    explicit PrepGuard(const PrepRequest& r) : saved(t_prep) { t_prep = r; }
    ~PrepGuard() { t_prep = saved; }
};
inline PrepRequest prep_from_desc(const sae_conv2d_desc* d) {
    return PrepRequest{d ? d->prepped : nullptr, d ? d->prepped_floats : 0, d ? d->prepped_layout : 0, nullptr, 0, nullptr, nullptr};
}
inline int64_t layout_id(int kind, int Mp, int Cp, int taps, int flip, int64_t sm, int64_t sc, int bm) {
    uint64_t h = 1469598103934665603ull;
    const int64_t v[8] = {kind, Mp, Cp, taps, flip, sm, sc, bm};
    for (int i = 0; i < 8; ++i) { h ^= (uint64_t)v[i]; h *= 1099511628211ull; }
    return (int64_t)(h & 0x7fffffffffffffffull) | 1;      // never 0 (= "this path has no layout")
}
// The weights of a launch: re-laid into `ws` unless the caller prepared them.  *done = true: the request was a query or a
// wprep-only call and has been served (the caller returns SAE_OK without launching).  kind 0: fp32 layout [tap][Cp][Mp],
// 1: bf16x6 cells.
int stage_weights(const float* w, float* ws, int M, int C, int Mp, int Cp, int taps, int64_t sm, int64_t sc, int flip, float alpha,
                  hipStream_t s, WScale wsc, bool bx, int bm, const float** wp, bool* done) {
    const int64_t need = bx ? (int64_t)27 * Mp * (Cp / 8) * 4 : (int64_t)taps * Cp * Mp;
    const int64_t layout = layout_id(bx ? 1 : 0, Mp, Cp, taps, flip, sm, sc, bx ? bm : 0);
    *done = false;
    if (t_prep.want_floats) {
        *t_prep.want_floats = need;
        if (t_prep.want_layout) *t_prep.want_layout = layout;
        *done = true;
        return SAE_OK;
    }
    float* dst = ws;
    if (t_prep.make) {
        if (t_prep.make_floats != need)
            return fail(SAE_EINVAL, "sae_conv2d_wprep_f32: buffer of %lld floats, the layout has %lld", (long long)t_prep.make_floats,
                        (long long)need);
        dst = t_prep.make;
        *done = true;
    } else if (t_prep.use && t_prep.use_layout == layout && t_prep.use_floats == need) {
        *wp = t_prep.use;
        return SAE_OK;
    }
    // (prepared weights of ANOTHER layout -- the launch took another kernel than the query assumed, e.g. because of the
    // alignment of this call's tensors -- are ignored: the weights are re-laid into the workspace as without them)
    if (bx) run_wprep_bx(w, dst, M, C, Mp, Cp, bm, sm, sc, flip, alpha, s, wsc);
    else run_wprep(w, dst, M, C, Mp, Cp, taps, sm, sc, flip, alpha, s, wsc);
    *wp = dst;
    return SAE_OK;
}

// ------------------------------------------------------------------------------------------
// 1x1 stride-1 pad-0 convolutions with at most four channels on one side (FromRGB 3 -> 128, ToRGB 128 -> 3, their
// input gradients): planes in, planes out, no matrix cores.  Through the 128-row gather tile FromRGB ran at 2.5 TFLOP/s
// = 1.7 TB/s of output (0.8 ms at n = 40); as a stream it is bound by the 1.3 GB it writes.
// A thread owns four consecutive pixels of one image; weights (with alpha and the optional per-channel factors folded
// in, as conv_wprep_kernel does) are wave-uniform scalar loads.
//   THIN_IN : C <= 4, any M: the C input quads stay in registers while M output quads are produced
//   !THIN_IN: M <= 4, any C: M accumulator quads over a loop of C input quads
// ------------------------------------------------------------------------------------------
struct ThinParams {
    int N, C, M;
    int64_t HW;
    int64_t sm, sc;
    float alpha;
    const float* rs_m; const float* rs_c;      // weight factors (or null)
    const float* in_scale;                     // [N][C] activation factors (or null)
    const float* bias; int act; float slope, scale;
};

template <bool THIN_IN>
__global__ __launch_bounds__(kBlock) void conv1x1_thin_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                              float* __restrict__ y, const ThinParams p) {
    const int n = blockIdx.y;
    const int64_t q = (int64_t)blockIdx.x * kBlock + threadIdx.x;      // quad of pixels
    if (4 * q >= p.HW) return;
    const float* xn = x + (int64_t)n * p.C * p.HW + 4 * q;
    float* yn = y + (int64_t)n * p.M * p.HW + 4 * q;
    auto weight = [&](int m, int c) {
        float v = p.alpha * w[m * p.sm + c * p.sc];
        if (p.rs_m) v *= p.rs_m[m];
        if (p.rs_c) v *= p.rs_c[c];
        return v;
    };
    auto finish = [&](f32x4 o, int m) {
        if (p.act) {
            const float bv = p.bias ? p.bias[m] : 0.0f;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float t = o[e] + bv;
                o[e] = ((t > 0.0f) ? t : t * p.slope) * p.scale;
            }
        }
        *reinterpret_cast<f32x4*>(yn + (int64_t)m * p.HW) = o;
    };
    if constexpr (THIN_IN) {
        f32x4 xv[4];
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            xv[c] = (c < p.C) ? *reinterpret_cast<const f32x4*>(xn + (int64_t)c * p.HW) : f32x4{0.0f, 0.0f, 0.0f, 0.0f};
            if (p.in_scale && c < p.C) xv[c] *= p.in_scale[n * p.C + c];
        }
        for (int m = 0; m < p.M; ++m) {
            f32x4 o = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
            for (int c = 0; c < 4; ++c)
                if (c < p.C) o += weight(m, c) * xv[c];
            finish(o, m);
        }
    } else {
        f32x4 acc[4];
#pragma unroll
        for (int m = 0; m < 4; ++m) acc[m] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll 8
        for (int c = 0; c < p.C; ++c) {
            f32x4 xv = *reinterpret_cast<const f32x4*>(xn + (int64_t)c * p.HW);
            if (p.in_scale) xv *= p.in_scale[n * p.C + c];
#pragma unroll
            for (int m = 0; m < 4; ++m)
                if (m < p.M) acc[m] += weight(m, c) * xv;
        }
#pragma unroll
        for (int m = 0; m < 4; ++m)
            if (m < p.M) finish(acc[m], m);
    }
}

// forward-type gather producing `mout` channels from `cin` channels
struct Epilogue {
    const float* bias; int act; float slope, scale;
    const float* residual = nullptr; float res_scale = 1.0f;
    const float* noise = nullptr; const float* noise_w = nullptr;     // IgemmParams::noise (modulated forward only)
};

int run_gather(const float* x, const float* w, float* y, float* ws, int64_t ws_floats, int N, int cin, int H, int W,
               int mout, int OH, int OW, int YH, int YW, int oys, int oxs, int ks, int stride, int pad, int64_t sm,
               int64_t sc, int flip, float alpha, hipStream_t s, Epilogue ep = Epilogue{nullptr, 0, 0.0f, 1.0f},
               const float* in_scale = nullptr, WScale wsc = WScale{nullptr, nullptr}) {
    const GatherPlan g = gather_plan(N, cin, mout, OH, OW, ks, stride, oys != 1 || oxs != 1);
    const bool prep_only = t_prep.make || t_prep.want_floats;
    if (!prep_only && (!ws || ws_floats < g.ws_floats))
        return fail(SAE_EWORKSPACE, "conv2d: workspace %lld < %lld floats", (long long)ws_floats, (long long)g.ws_floats);
    // thin 1x1 layers (at most four channels on one side, e.g. FromRGB / ToRGB): streamed, see conv1x1_thin_kernel
    static const int thin_knob = tuning_knob("SAE_CONV_THIN", 1);
    if (ep.residual && (oys != 1 || oxs != 1 || YH != OH || YW != OW || g.bx || (reinterpret_cast<uintptr_t>(ep.residual) & 15) != 0))
        return fail(SAE_EINVAL, "conv2d: the fused residual needs a dense, 16-byte aligned output-shaped tensor and the "
                                "exact-fp32 kernels");
    if (ep.noise && (!in_scale || !ep.act || g.bx || oys != 1 || oxs != 1 || YH != OH || YW != OW))
        return fail(SAE_EINVAL, "conv2d: the fused noise needs the modulated exact-fp32 forward with its activation epilogue");
    if (thin_knob && !ep.residual && !ep.noise && ks == 1 && stride == 1 && pad == 0 && oys == 1 && oxs == 1 && (cin <= 4 || mout <= 4) &&
        H == OH && W == OW && YH == OH && YW == OW && ((int64_t)H * W) % 4 == 0 &&
        ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(y)) & 15) == 0 && N <= 65535) {
        if (prep_only) {      // the streaming kernels read the parameter layout: nothing to prepare
            if (t_prep.want_floats) { *t_prep.want_floats = 0; if (t_prep.want_layout) *t_prep.want_layout = 0; return SAE_OK; }
            return fail(SAE_EINVAL, "sae_conv2d_wprep_f32: this launch has no weight layout (sae_conv2d_wprep_floats returns 0)");
        }
        ThinParams t{};
        t.N = N; t.C = cin; t.M = mout; t.HW = (int64_t)H * W; t.sm = sm; t.sc = sc; t.alpha = alpha;
        t.rs_m = wsc.rs_m; t.rs_c = wsc.rs_c; t.in_scale = in_scale;
        t.bias = ep.bias; t.act = ep.act; t.slope = ep.slope; t.scale = ep.scale;
        const dim3 grid((unsigned)ceil_div64(t.HW / 4, kBlock), (unsigned)N);
        if (cin <= 4) hipLaunchKernelGGL((conv1x1_thin_kernel<true>), grid, dim3(kBlock), 0, s, x, w, y, t);
        else hipLaunchKernelGGL((conv1x1_thin_kernel<false>), grid, dim3(kBlock), 0, s, x, w, y, t);
        return SAE_OK;
    }
    if (in_scale && g.bx)
        return fail(SAE_EINVAL, "modulated conv: the activation factors are staged by the exact-fp32 kernels only "
                                "(SAE_CONV_MATH_F32); under bf16x6 modulate the activation before the call");
    const float* wp = ws;
    {
        bool done;
        const int rc = stage_weights(w, ws, mout, cin, g.Mp, g.Cp, g.taps, sm, sc, flip, alpha, s, wsc, g.bx, g.sh.bm, &wp, &done);
        if (rc != SAE_OK || done) return rc;
    }
    IgemmParams p{};
    p.N = N; p.C = cin; p.H = H; p.W = W; p.M = mout; p.OH = OH; p.OW = OW; p.YH = YH; p.YW = YW;
    p.oys = oys; p.oxs = oxs; p.Cp = g.Cp; p.Mp = g.Mp; p.pad = pad;
    p.slab_stride = g.out_floats4;
    p.in_scale = in_scale;
    static const int xcd_knob = tuning_knob("SAE_XCD_ORDER", 1);
    p.xcd_order = xcd_knob;
    if (g.ksplit == 1) {
        p.bias = ep.bias; p.act = ep.act; p.act_slope = ep.slope; p.act_scale = ep.scale;
        p.residual = ep.residual; p.res_scale = ep.res_scale;
        p.noise = ep.noise; p.noise_w = ep.noise_w;
    }
    float* out = (g.ksplit > 1) ? ws + g.wp_floats : y;
    static const int vec_knob = tuning_knob("SAE_IGEMM_VEC_STORE", 1);
    // Round 2 measured +2.5 % with K loops of 64 chunks and -1 % with 16 or 32 on the 128 x 128 tile and enabled it for long
    // loops (and fused bias + leaky-ReLU) only.  With the epilogue buffer spanning both staging areas (so that the 64 x 256
    // tile and the 1x1 kernels have it at all) it wins for every K length (same-box A/B, tools/ab_conv.py tuning with
    // SAE_IGEMM_VEC_STORE=1|3): 128 -> 128 @256^2 129.1 -> 131.0, 256 @128^2 134.4 -> 135.9, 1x1 128 -> 256 @256^2 83.6 -> 90.5
    // TFLOP/s.  (knob 3 = the round-2 rule)
    p.vec_store = vec_knob && oys == 1 && oxs == 1 && OW % 4 == 0 && YW % 4 == 0 && (reinterpret_cast<uintptr_t>(out) & 15) == 0 &&
                  (vec_knob != 3 || g.cps >= 48 || ep.act || ep.residual);
    int rc;
    if (ks == 3 && stride == 1) rc = launch_igemm<3, 1>(x, wp, out, p, g, s);
    else if (ks == 3) rc = launch_igemm<3, 2>(x, wp, out, p, g, s);
    else if (stride == 1) rc = launch_igemm<1, 1>(x, wp, out, p, g, s);
    else rc = launch_igemm<1, 2>(x, wp, out, p, g, s);
    if (rc != SAE_OK) return rc;
    if (g.ksplit > 1) {
        // y is exactly N*mout*OH*OW floats; the slabs are padded to a multiple of 4, y may not be
        const int64_t numel = (int64_t)N * mout * OH * OW;
        const int64_t n4 = numel / 4;
        int64_t blocks = ceil_div64(n4 > 0 ? n4 : 1, kBlock);
        if (blocks > 4096) blocks = 4096;
        if (n4 > 0)
            hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s,
                               (const float*)out, y, n4, g.out_floats4, g.ksplit, ep.bias, ep.act, ep.slope, ep.scale,
                               OH * OW, mout, ep.residual, ep.res_scale, ep.noise, ep.noise_w);
    }
    return SAE_OK;
}

int64_t gather_ws(int N, int cin, int mout, int OH, int OW, int ks, int stride, bool scatter) {
    return gather_plan(N, cin, mout, OH, OW, ks, stride, scatter).ws_floats;
}

// ---- tr2 (conv_igemm_tr2_kernel): exact fp32, input rows a multiple of four floats wide and 16-byte aligned
#ifndef SAE_TR2_DEFAULT
#define SAE_TR2_DEFAULT 1
#endif
#ifndef SAE_TR2_MINW_DEFAULT
#define SAE_TR2_MINW_DEFAULT 32
#endif
#ifndef SAE_TR2_FLAT_CK
#define SAE_TR2_FLAT_CK 16      // channel chunk of the flat 64-row tile (its 16-channel form sits at the 256-register ceiling)
#endif
struct Tr2Shape { int mi, bm, ck; };
Tr2Shape tr2_shape(int mout) {
    static const int knob = tuning_knob("SAE_TR2", SAE_TR2_DEFAULT);      // 0: conv_igemm_tr_kernel, 2: 8-channel chunks
    Tr2Shape s{};
    s.mi = mout > 32 ? 2 : 1;
    s.bm = 32 * s.mi;
    // 16-channel chunks for the 64-row tile (117.6 / 113.9 / 101.1 TFLOP/s on the three D shapes vs 118.0 / 112.3 / 99.3 with
    // 8); the 32-row tile of the narrow layers keeps 8 (three workgroups per CU: 92.6 vs 84.4 at 64 -> 32 @129, B = 128)
    s.ck = (knob == 2 || (knob == 1 && s.mi == 1)) ? 8 : 16;
    return s;
}
bool tr2_eligible(const float* x, int N, int cin, int IH, int IW, int mout, bool modulated) {
    static const int knob = tuning_knob("SAE_TR2", SAE_TR2_DEFAULT);
    if (!knob || conv_math() != 0) return false;
    if (IW % 4 != 0 || IW < 8 || !aligned16(x)) return false;
    // in-step by shape (profiles/r3_tr2_by_shape.txt): 32-wide inputs and up gain (n = 16 modulated 128 -> 256 @257: 100.6 ->
    // 114.9 TFLOP/s; 512 -> 512 @65: 84.3 -> 89.7), the 16- and 8-wide ones lose to the free-form tiles of
    // conv_igemm_tr_kernel (their 2^k + 1 grids are mostly strip), except modulated ones (factors from LDS, not per slot)
    // ... and 16-wide ones (17 x 17 grids) on FLAT tiles when the launch has a full round of workgroups: Dpatch 128 -> 256 @33,
    // B = 128: 60 -> 75 TFLOP/s; with 16 images (384 workgroups) the K split of conv_igemm_tr_kernel stays ahead, 60 vs 59
    static const int minw_knob = tuning_knob("SAE_TR2_MINW", SAE_TR2_MINW_DEFAULT);
    const bool flat16 = IW == 16 && IH >= 8 && (int64_t)N * ceil_div((IH + 1) * (IW + 1), 128) * ceil_div(mout, 64) >= 512;
    if (knob == 1 && IW < (modulated ? 16 : minw_knob) && !flat16) return false;
    if (modulated && cin > kTr2MaxC) return false;
    // 32-bit byte offsets inside a tile's images (at most 128 of them)
    if ((int64_t)(N < 128 ? N : 128) * cin * IH * IW * 4 >= ((int64_t)1 << 31)) return false;
    (void)mout;
    return true;
}
// tile of a region: TN x TH x TW positions, TW a multiple of 4 or the whole (narrow) region; fewest wave-tiles
bool tr2_pick_tile(int qw, int qh, int N, bool one_image, TrRegion& g) {
    double best = -1.0;
    for (int tw = 128; tw >= 1; --tw) {
        if (tw % 4 != 0 && !(tw == qw && qw < 32)) continue;
        const int rs = ((tw + 3) & ~3) + 4;
        for (int th = 1; th * tw <= 128; ++th) {
            if (th > qh && th > 1) break;
            int tn = 128 / (tw * th);
            if (tn > N) tn = N;
            if (one_image) tn = 1;
            while (tn > 1 && tn * (th + 1) * rs > kTr2Cap) --tn;
            if (tn * (th + 1) * rs > kTr2Cap) continue;
            const int waves = ceil_div(tw * th * tn, 32);
            const double cost = (double)ceil_div(qw, tw) * ceil_div(qh, th) * ceil_div(N, tn) * (1.5 + waves);
            if (best < 0 || cost < best) { best = cost; g.tw = tw; g.th = th; g.tn = tn; }
        }
    }
    return best >= 0;
}
int run_tr2(const float* x, const float* w, float* y, float* ws, int64_t ws_floats, int N, int cin, int IH, int IW,
            int mout, int OH, int OW, int pad, int64_t sm, int64_t sc, float alpha, hipStream_t s,
            const float* in_scale, WScale wsc) {
    const Tr2Shape sh = tr2_shape(mout);
    const int Mp = round_up(mout, sh.bm), Cp = round_up(cin, sh.ck);
    const int64_t need = (int64_t)9 * Cp * Mp;
    if (!(t_prep.make || t_prep.want_floats) && (!ws || ws_floats < need))
        return fail(SAE_EWORKSPACE, "conv2d: workspace %lld < %lld floats", (long long)ws_floats, (long long)need);
    const float* wp = ws;
    {
        bool done;
        const int rc = stage_weights(w, ws, mout, cin, Mp, Cp, 9, sm, sc, 0, alpha, s, wsc, false, sh.bm, &wp, &done);
        if (rc != SAE_OK || done) return rc;
    }
    TrParams p{};
    p.in_scale = in_scale;
    p.N = N; p.C = cin; p.IH = IH; p.IW = IW; p.M = mout; p.OH = OH; p.OW = OW; p.Cp = Cp; p.Mp = Mp; p.pad = pad;
    const int QH = (OH + pad - 1) / 2 + 1, QW = (OW + pad - 1) / 2 + 1;
    // small odd grids (17 ... 65 wide): flat tiles, runs of 128 positions of the whole grid (see the kernel)
    static const int flat_knob = tuning_knob("SAE_TR2_FLAT", 1);
    const int flat_rows = (QW + 126) / QW + 1;                         // rows a run of 128 positions can touch
    const int flat_rs = ((QW + 3) & ~3) + 4;
    // (33- and 65-wide grids measured SLOWER flat than as main region + strips: 107.5 vs 112.1 TFLOP/s at 256 -> 512 @129, 95.5 vs
    // 100.1 at 512 -> 512 @65 -- a run of 128 positions of a 65-wide grid stages 4 x 72 floats per channel where a 4 x 32 tile
    // stages 5 x 36, and its 8-byte stores are twice as many; profiles/r4_ab_tr2_flat.txt.  SAE_TR2_FLAT=2 takes them anyway.)
    const bool flat = flat_knob && (QW & 1) && QW >= 17 && QW <= (flat_knob == 2 ? 72 : 20) && QH * QW >= 128 &&
                      (flat_rows + 1) * flat_rs <= kTr2FlatCap;
    if (flat) {
        for (int r = 0; r < 3; ++r) { p.reg[r] = TrRegion{}; p.reg[r].tw = p.reg[r].th = p.reg[r].tn = p.reg[r].tiles_x = p.reg[r].tiles_y = p.reg[r].tiles_n = 1; }
        TrRegion& g = p.reg[0];
        g.QH = QH; g.QW = QW; g.tw = QW; g.th = flat_rows; g.tn = 1; g.flat = 1;
        g.tiles_x = ceil_div(QH * QW, 128); g.tiles_y = 1; g.tiles_n = N;
        g.blocks = g.tiles_x * N;
        p.mtiles = Mp / sh.bm;
        p.main_items = g.blocks * p.mtiles;
        p.strip_items = 0;
        SAE_TRACE("tr2 flat mi=%d ck=%d mod=%d tiles=%d rows=%d", sh.mi, sh.ck, in_scale ? 1 : 0, g.blocks, flat_rows);
        const dim3 fgrid((unsigned)(8 * ceil_div(p.main_items, 8)));
#define SAE_TR2F(MI_, CK_)                                                                                                   \
    do {                                                                                                                     \
        if (in_scale) hipLaunchKernelGGL((conv_igemm_tr2_kernel<MI_, CK_, true, true>), fgrid, dim3(kBlock), 0, s, x, wp, y, p);  \
        else hipLaunchKernelGGL((conv_igemm_tr2_kernel<MI_, CK_, false, true>), fgrid, dim3(kBlock), 0, s, x, wp, y, p);          \
    } while (0)
        if (sh.mi == 2 && sh.ck == 16 && SAE_TR2_FLAT_CK == 16) SAE_TR2F(2, 16);
        else if (sh.mi == 2) SAE_TR2F(2, 8);
        else if (sh.ck == 16) SAE_TR2F(1, 16);
        else SAE_TR2F(1, 8);
#undef SAE_TR2F
        return SAE_OK;
    }
    // main region (sides a multiple of 4) + right / bottom strips, as in run_tr
    auto main_side = [](int q, int align) {
        int t = 32;
        while (t > q) t >>= 1;
        const int m = (q / t) * t;
        return (q - m > 0 && (q - m) * 4 <= t && m > 0 && m % align == 0) ? m : q;    // split only a thin remainder
    };
    const int QHm = main_side(QH, 1), QWm = main_side(QW, 4);   // the right strip starts on a quad boundary
    struct Region { int y0, y1, x0, x1; };
    Region regions[3];
    int nreg = 0;
    regions[nreg++] = {0, QHm, 0, QWm};
    if (QWm < QW) regions[nreg++] = {0, QH, QWm, QW};
    if (QHm < QH) regions[nreg++] = {QHm, QH, 0, QWm};
    int total = 0;
    for (int r = 0; r < 3; ++r) {
        TrRegion& g = p.reg[r];
        g = TrRegion{};
        if (r >= nreg) { g.tw = g.th = g.tn = g.tiles_x = g.tiles_y = g.tiles_n = 1; continue; }
        const int qh = regions[r].y1 - regions[r].y0, qw = regions[r].x1 - regions[r].x0;
        g.qy_base = regions[r].y0; g.qx_base = regions[r].x0; g.QH = regions[r].y1; g.QW = regions[r].x1;
        if (!tr2_pick_tile(qw, qh, N, in_scale != nullptr, g)) return fail(SAE_EINVAL, "conv tr2: no tile fits the LDS patch cap");
        g.tiles_x = ceil_div(qw, g.tw);
        g.tiles_y = ceil_div(qh, g.th);
        g.tiles_n = ceil_div(N, g.tn);
        g.blocks = g.tiles_x * g.tiles_y * g.tiles_n;
        total += g.blocks;
    }
    p.mtiles = Mp / sh.bm;
    p.main_items = p.reg[0].blocks * p.mtiles;
    p.strip_items = (total - p.reg[0].blocks) * p.mtiles;
    SAE_TRACE("tr2 mi=%d ck=%d mod=%d tiles=%d main=%dx%dx%d", sh.mi, sh.ck, in_scale ? 1 : 0, total, p.reg[0].tn, p.reg[0].th,
              p.reg[0].tw);
    const dim3 grid((unsigned)(8 * (ceil_div(p.main_items, 8) + ceil_div(p.strip_items, 8))));
#define SAE_TR2(MI_, CK_)                                                                                              \
    do {                                                                                                               \
        if (in_scale) hipLaunchKernelGGL((conv_igemm_tr2_kernel<MI_, CK_, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);   \
        else hipLaunchKernelGGL((conv_igemm_tr2_kernel<MI_, CK_, false>), grid, dim3(kBlock), 0, s, x, wp, y, p);           \
    } while (0)
    if (sh.mi == 2 && sh.ck == 16) SAE_TR2(2, 16);
    else if (sh.mi == 2) SAE_TR2(2, 8);
    else if (sh.ck == 16) SAE_TR2(1, 16);
    else SAE_TR2(1, 8);
#undef SAE_TR2
    return SAE_OK;
}

// q-grid decomposition of conv_igemm_tr_kernel: fills p.reg[], returns the number of q tiles (-1: no tile fits)
int tr_plan_regions(TrParams& p, int N, int OH, int OW, int pad, int bq) {
    const int QH = (OH + pad - 1) / 2 + 1, QW = (OW + pad - 1) / 2 + 1;
    // The transposed problems of this network have 2^k + 1 wide q grids (129, 65, 33, ...): one
    // launch with 32-wide tiles would spend 25 % (129) to 94 % (33) of its MFMAs on padding.  The
    // grid is cut into a main region whose sides are multiples of the natural tile side plus thin
    // right / bottom strips, each launched with its own best tile shape.  For the smallest grids (5 x 5, 9 x 9, 17 x 17:
    // the 9 x 9 ... 33 x 33 maps of D / Dpatch) the strips are a large share of the work and free-form tiles over the WHOLE
    // grid (e.g. five 5 x 5 grids per 128-position tile) can cost less: both decompositions are priced, the cheaper one runs.
    auto main_side = [](int q) {
        int t = 32;
        while (t > q) t >>= 1;
        const int m = (q / t) * t;
        return (q - m > 0 && (q - m) * 4 <= t && m > 0) ? m : q;    // split only a thin remainder
    };
    struct Region { int y0, y1, x0, x1; };
    // tile = tn x th x tw q-positions: fewest workgroups (each costs bq lanes of MFMA work), with a
    // penalty for narrow rows (short global-memory runs: a 9-wide tile measured no faster than a
    // 32-wide one with 12 % more workgroups)
    auto plan = [&](const Region* regions, int nreg, TrRegion* out, double* cost_out) {
        int total_blocks = 0;
        double total_cost = 0.0;
        for (int r = 0; r < 3; ++r) {
            TrRegion& g = out[r];
            g = TrRegion{};
            if (r >= nreg) { g.tw = g.th = g.tn = g.tiles_x = g.tiles_y = g.tiles_n = 1; continue; }   // empty: blocks = 0
            const int qh = regions[r].y1 - regions[r].y0, qw = regions[r].x1 - regions[r].x0;
            g.qy_base = regions[r].y0; g.qx_base = regions[r].x0; g.QH = regions[r].y1; g.QW = regions[r].x1;
            const int cap = (25 * bq) / 16;
            double best = -1.0;
            for (int tw = 1; tw <= 32; ++tw)
                for (int th = 1; th * tw <= bq; ++th) {
                    if (tw < 4 && tw < qw) continue;               // narrow tiles only for narrow strips
                    int tn = bq / (tw * th);
                    if (tn > N) tn = N;
                    if (tn * (th + 1) * (tw + 1) > cap) continue;
                    const double cost = (double)ceil_div(qw, tw) * ceil_div(qh, th) * ceil_div(N, tn) * (1.0 + 8.0 / (tw < qw ? tw : 32));
                    if (best < 0 || cost < best) {
                        best = cost; g.tw = tw; g.th = th; g.tn = tn;
                    }
                }
            if (best < 0) return -1;
            g.tiles_x = ceil_div(qw, g.tw);
            g.tiles_y = ceil_div(qh, g.th);
            g.tiles_n = ceil_div(N, g.tn);
            g.blocks = g.tiles_x * g.tiles_y * g.tiles_n;
            total_blocks += g.blocks;
            total_cost += best;
        }
        *cost_out = total_cost;
        return total_blocks;
    };
    const int QHm = main_side(QH), QWm = main_side(QW);
    Region split[3];
    int nsplit = 0;
    split[nsplit++] = {0, QHm, 0, QWm};
    if (QWm < QW) split[nsplit++] = {0, QH, QWm, QW};          // right strip (full height)
    if (QHm < QH) split[nsplit++] = {QHm, QH, 0, QWm};          // bottom strip
    double cost_split = 0.0;
    const int blocks_split = plan(split, nsplit, p.reg, &cost_split);
    static const int whole_knob = tuning_knob("SAE_TR_WHOLE", 1);
    if (nsplit > 1 && whole_knob && QH <= 17 && QW <= 17) {
        const Region whole[1] = {{0, QH, 0, QW}};
        TrRegion alt[3];
        double cost_whole = 0.0;
        const int blocks_whole = plan(whole, 1, alt, &cost_whole);
        if (blocks_whole > 0 && (blocks_split < 0 || cost_whole < cost_split || whole_knob == 2)) {
            for (int r = 0; r < 3; ++r) p.reg[r] = alt[r];
            return blocks_whole;
        }
    }
    return blocks_split;
}

// K split of conv_igemm_tr_kernel (exact fp32 only): like the forward gather's, for launches that leave most CUs idle
struct TrSplit { int ksplit, cps; int64_t out_floats4; };
TrSplit tr_split(int total_blocks, int mtiles, int nchunks, int64_t numel, bool fp32_kernel) {
    TrSplit t{1, nchunks, (numel + 3) / 4 * 4};
    const int blocks = total_blocks * mtiles;
    static const int split_knob = tuning_knob("SAE_TR_SPLITK", 1);     // 0: never (bit-identity comparisons against tr2)
    if (split_knob && fp32_kernel && blocks < 192 && nchunks >= 16 && numel % 4 == 0) {
        int k = 512 / (blocks > 0 ? blocks : 1);
        if (k > 8) k = 8;
        if (k > nchunks / 8) k = nchunks / 8;
        if (k >= 2) {
            t.cps = ceil_div(nchunks, k);
            t.ksplit = ceil_div(nchunks, t.cps);
        }
    }
    return t;
}

// stride-2 3x3 transposed gather producing `mout` channels (the large image) from `cin` channels
int run_tr(const float* x, const float* w, float* y, float* ws, int64_t ws_floats, int N, int cin, int IH, int IW,
           int mout, int OH, int OW, int pad, int64_t sm, int64_t sc, float alpha, hipStream_t s,
           const float* in_scale = nullptr, WScale wsc = WScale{nullptr, nullptr}) {
    if (tr2_eligible(x, N, cin, IH, IW, mout, in_scale != nullptr))
        return run_tr2(x, w, y, ws, ws_floats, N, cin, IH, IW, mout, OH, OW, pad, sm, sc, alpha, s, in_scale, wsc);
    const TrShape sh = tr_shape(mout);
    constexpr int CK = 8;
    const int Mp = round_up(mout, sh.bm), Cp = round_up(cin, sh.ck);
    const bool bx = conv_math() == 1 && sh.cfg == 0;
    const int64_t need = bx ? (int64_t)27 * Mp * (Cp / 8) * 4 : (int64_t)9 * Cp * Mp;
    if (!(t_prep.make || t_prep.want_floats) && (!ws || ws_floats < need))
        return fail(SAE_EWORKSPACE, "conv2d: workspace %lld < %lld floats", (long long)ws_floats, (long long)need);
    if (in_scale && bx)
        return fail(SAE_EINVAL, "modulated conv: the activation factors are staged by the exact-fp32 kernels only "
                                "(SAE_CONV_MATH_F32); under bf16x6 modulate the activation before the call");
    const float* wp = ws;
    {
        bool done;
        const int rc = stage_weights(w, ws, mout, cin, Mp, Cp, 9, sm, sc, 0, alpha, s, wsc, bx, sh.bm, &wp, &done);
        if (rc != SAE_OK || done) return rc;
    }
    TrParams p{};
    p.in_scale = in_scale;
    p.N = N; p.C = cin; p.IH = IH; p.IW = IW; p.M = mout; p.OH = OH; p.OW = OW; p.Cp = Cp; p.Mp = Mp; p.pad = pad;
    static const int nostore_knob = tuning_knob("SAE_TR_NOSTORE", 0);
    p.debug_skip_store = nostore_knob;
    const int total_blocks = tr_plan_regions(p, N, OH, OW, pad, sh.bq);
    if (total_blocks < 0) return fail(SAE_EINVAL, "conv tr: no tile fits the LDS patch cap");
    const TrSplit sp = tr_split(total_blocks, Mp / sh.bm, Cp / sh.ck, (int64_t)N * mout * OH * OW, !bx);
    if (sp.ksplit > 1 && ws_floats < need + (int64_t)sp.ksplit * sp.out_floats4)
        return fail(SAE_EWORKSPACE, "conv2d: workspace %lld < %lld floats", (long long)ws_floats,
                    (long long)(need + (int64_t)sp.ksplit * sp.out_floats4));
    p.chunks_per_split = sp.cps;
    p.slab_stride = sp.ksplit > 1 ? sp.out_floats4 : 0;
    SAE_TRACE("tr cfg=%d tiles=%d ksplit=%d", sh.cfg, total_blocks, sp.ksplit);
    float* const y_final = y;
    if (sp.ksplit > 1) y = ws + need;
    {
        const dim3 grid((unsigned)total_blocks, (unsigned)(Mp / sh.bm), (unsigned)sp.ksplit);
        if (bx) {
            hipLaunchKernelGGL((conv_igemm_tr_bx_kernel<2, 2, 2>), grid, dim3(kBlock), 0, s, x,
                               reinterpret_cast<const u32x4*>(wp), y, p);
            return SAE_OK;
        }
#define SAE_TR(...)                                                                                             \
    do {                                                                                                        \
        if (p.in_scale) hipLaunchKernelGGL((conv_igemm_tr_kernel<__VA_ARGS__, true>), grid, dim3(kBlock), 0, s, x, wp, y, p);   \
        else hipLaunchKernelGGL((conv_igemm_tr_kernel<__VA_ARGS__, false>), grid, dim3(kBlock), 0, s, x, wp, y, p);             \
    } while (0)
        switch (sh.cfg) {
            case 3: SAE_TR(2, 1, 4, 16); break;
            case 0: SAE_TR(2, 2, 2, CK); break;
            case 4:
                if (p.in_scale) hipLaunchKernelGGL((conv_igemm_tr_kernel<2, 2, 4, CK, true>), grid, dim3(512), 0, s, x, wp, y, p);
                else hipLaunchKernelGGL((conv_igemm_tr_kernel<2, 2, 4, CK, false>), grid, dim3(512), 0, s, x, wp, y, p);
                break;
            case 1: SAE_TR(2, 1, 4, CK); break;
            default: SAE_TR(1, 1, 4, CK); break;
        }
#undef SAE_TR
    }
    if (sp.ksplit > 1) {
        const int64_t n4 = (int64_t)N * mout * OH * OW / 4;
        int64_t blocks = ceil_div64(n4, kBlock);
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_splitk_reduce_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const float*)y, y_final, n4,
                           sp.out_floats4, sp.ksplit, (const float*)nullptr, 0, 0.0f, 1.0f, OH * OW, mout,
                           (const float*)nullptr, 1.0f, (const float*)nullptr, (const float*)nullptr);
    }
    return SAE_OK;
}

int64_t tr_ws(int N, int cin, int mout, int OH, int OW, int pad) {
    const TrShape sh = tr_shape(mout);
    if (conv_math() == 1 && sh.cfg == 0) return (int64_t)27 * round_up(mout, sh.bm) * (round_up(cin, 8) / 8) * 4;
    // whichever of the two fp32 kernels the launch takes (tr2 needs the pointer alignment to decide)
    const Tr2Shape s2 = tr2_shape(mout);
    int64_t a = (int64_t)9 * round_up(cin, sh.ck) * round_up(mout, sh.bm);
    const int64_t b = (int64_t)9 * round_up(cin, s2.ck) * round_up(mout, s2.bm);
    TrParams p{};
    const int total_blocks = tr_plan_regions(p, N, OH, OW, pad, sh.bq);
    if (total_blocks > 0) {
        const TrSplit sp = tr_split(total_blocks, round_up(mout, sh.bm) / sh.bm, round_up(cin, sh.ck) / sh.ck,
                                    (int64_t)N * mout * OH * OW, true);
        if (sp.ksplit > 1) a += (int64_t)sp.ksplit * sp.out_floats4;
    }
    return a > b ? a : b;
}

template <int KS, int S, int TA, int TB, int WA, int WB, int MODE, bool WQ = false>
void launch_wgrad(const float* x, const float* gy, float* slab, const WgradParams& p, const WgPlan& w, hipStream_t s) {
    const dim3 grid((unsigned)(w.Bp / w.sh.bb), (unsigned)(w.Ap / w.sh.ba), (unsigned)w.slices);
    if constexpr (WQ) {
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, TA, TB, WA, WB, MODE, false, true>), grid, dim3(kBlock), 0, s, x, gy, slab, p);
        return;
    }
    if (p.l_scale || p.s_scale)
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, TA, TB, WA, WB, MODE, true>), grid, dim3(kBlock), 0, s, x, gy, slab, p);
    else
        hipLaunchKernelGGL((conv_wgrad_kernel<KS, S, TA, TB, WA, WB, MODE, false>), grid, dim3(kBlock), 0, s, x, gy, slab, p);
}

}  // namespace
}  // namespace sae

using namespace sae;

namespace {
// conv_wgrad16_kernel: the shape half of its eligibility test (the launch adds pointer alignment and "no per-element factors")
bool wg16_shape_ok(const sae_conv2d_desc* d) {
    static const int knob = tuning_knob("SAE_WGRAD16", 1);
    if (!knob || conv_math() != 0 || d->kh != 3 || d->kw != 3 || d->n < 1) return false;
    if (wg_shape((int)d->m, (int)d->c, d->kh, d->stride).mode != 0) return false;
    int tw_log2, th_log2;
    pick_tile(kWgPix, (int)d->oh, (int)d->ow, 32, &tw_log2, &th_log2);
    if ((kWgPix >> (tw_log2 + th_log2)) != 1 || d->ow % 4 != 0) return false;          // one image per 64-pixel chunk, gy quads
    if (d->stride == 1) return d->w % 4 == 0 && d->pad <= 4;
    // stride 2: only where the 128 a x 32 b tile of the first-generation kernel pads the gradient's channels to twice their
    // number (Dpatch 32 -> 64 @129^2, B = 128: 55 -> 94 TFLOP/s); on wide layers its larger tile amortises the stride-2
    // patch (5.3 staged floats per output pixel and channel) better: 114 vs 111 TFLOP/s (profiles/r4_ab_wgrad16.txt)
    static const int s2_knob = tuning_knob("SAE_WGRAD16_S2", 64);
    return d->stride == 2 && d->pad == 0 && d->w >= 4 && d->m <= s2_knob;
}

// the streaming weight gradient of thin 1x1 layers (conv1x1_thin_wgrad_kernel): shape test and launch geometry
struct ThinWgPlan { bool ok; int cb, cs, cbp, split, qps; bool big_is_m; int64_t ws_floats; };
ThinWgPlan thin_wg_plan(const sae_conv2d_desc* d) {
    ThinWgPlan t{};
    static const int knob = tuning_knob("SAE_WGRAD_THIN", 1);
    const int64_t hw = d->h * d->w;
    t.ok = knob && d->kh == 1 && d->kw == 1 && d->stride == 1 && d->pad == 0 && (d->m <= 4 || d->c <= 4) && hw % 4 == 0 && hw >= 1024 &&
           d->n >= 1 && d->n <= 65535;
    if (!t.ok) return t;
    t.big_is_m = d->c <= 4;                       // FromRGB: the gradient tensor is the big one
    t.cb = (int)(t.big_is_m ? d->m : d->c);
    t.cs = (int)(t.big_is_m ? d->c : d->m);
    t.cbp = (t.cb + 3) / 4 * 4;
    const int64_t quads = hw / 4;
    int64_t split = ceil_div64(2048, (int64_t)(t.cbp / 4) * d->n);     // ~2048 workgroups
    if (split > quads / (2 * kBlock)) split = quads / (2 * kBlock);      // at least two rounds of quads per thread
    if (split < 1) split = 1;
    if (split > 64) split = 64;
    t.qps = (int)ceil_div64(quads, split);
    t.split = (int)ceil_div64(quads, t.qps);
    t.ws_floats = (int64_t)d->n * t.split * t.cbp * 4;
    return t;
}
}  // namespace

extern "C" int64_t sae_conv2d_workspace(const sae_conv2d_desc* d, int32_t op) {
    if (!desc_ok(d, "sae_conv2d_workspace")) return 0;
    switch (op) {
        case SAE_CONV_FWD:
            return gather_ws((int)d->n, (int)d->c, (int)d->m, (int)d->oh, (int)d->ow, d->kh, d->stride, false);
        case SAE_CONV_DGRAD:
            if (d->stride == 1) return gather_ws((int)d->n, (int)d->m, (int)d->c, (int)d->h, (int)d->w, d->kh, 1, false);
            if (d->kh == 1) return gather_ws((int)d->n, (int)d->m, (int)d->c, (int)d->oh, (int)d->ow, 1, 1, true);
            return tr_ws((int)d->n, (int)d->m, (int)d->c, (int)d->h, (int)d->w, d->pad);
        case SAE_CONV_WGRAD: {
            const WgPlan w = wg_plan(d);
            int64_t need = (int64_t)w.slices * w.taps * w.Ap * w.Bp;
            if (wg16_shape_ok(d)) {                     // (taken only for suitably aligned tensors: room for either plan)
                const WgPlan w16 = wg_plan(d, true);
                const int64_t need16 = (int64_t)w16.slices * w16.taps * w16.Ap * w16.Bp;
                if (need16 > need) need = need16;
            }
            const ThinWgPlan t = thin_wg_plan(d);       // (taken only for 16-byte aligned tensors: room for either path)
            return (t.ok && t.ws_floats > need) ? t.ws_floats : need;
        }
        default: return 0;
    }
}

namespace {
const sae_conv2d_mod kNoMod = {nullptr, nullptr, nullptr, nullptr};

int conv_fwd_impl(const char* who, const float* x, const float* w, float* y, const sae_conv2d_desc* d,
                  const sae_conv2d_mod& mod, float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    if (!desc_ok(d, who)) return SAE_EINVAL;
    if (d->n == 0) return SAE_OK;
    if (!x || !w || !y) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (mod.y_scale) return fail(SAE_EINVAL, "%s: y_scale has no meaning for the forward operation", who);
    hipStream_t s = (hipStream_t)stream;
    const PrepGuard prep(t_prep.make || t_prep.want_floats ? t_prep : prep_from_desc(d));
    int rc = run_gather(x, w, y, workspace, workspace_floats, (int)d->n, (int)d->c, (int)d->h, (int)d->w, (int)d->m,
                        (int)d->oh, (int)d->ow, (int)d->oh, (int)d->ow, 1, 1, d->kh, d->stride, d->pad,
                        d->w_stride_m, d->w_stride_c, 0, alpha, s, Epilogue{nullptr, 0, 0.0f, 1.0f}, mod.x_scale,
                        WScale{mod.wm_scale, mod.wc_scale});
    if (rc != SAE_OK) return rc;
    return check_launch(who);
}
}  // namespace

extern "C" int sae_conv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d, float alpha,
                                  float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_fwd_impl("sae_conv2d_fwd_f32", x, w, y, d, kNoMod, alpha, workspace, workspace_floats, stream);
}

// ---- the sixteen transform-domain products of the Winograd route (csrc/winograd.hip): M[xi] = U[xi] V[xi], each a 1x1
// convolution of `tiles_h x tiles_w` "pixels" -- ONE batched launch of the 1x1 gather (blockIdx.z = xi, IgemmParams::batch)
// behind ONE launch that lays the sixteen [m][c] weight slices out as the gather reads them ([c][Mp], zero padded).
namespace sae {
namespace {
// u: [16][M][C] -> wp: [16][Cp][Mp]
__global__ __launch_bounds__(kBlock) void wino_wprep_kernel(const float* __restrict__ u, float* __restrict__ wp, int M, int C,
                                                            int Mp, int Cp) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= (int64_t)Cp * Mp) return;
    const int c = (int)(i / Mp), m = (int)(i - (int64_t)c * Mp);
    const int xi = blockIdx.y;
    wp[(int64_t)xi * Cp * Mp + i] = (m < M && c < C) ? u[((int64_t)xi * M + m) * C + c] : 0.0f;
}
GatherPlan wino_gemm_plan(int n, int c, int m, int th, int tw) {
    GatherPlan g = gather_plan(n, c, m, th, tw, 1, 1, false);
    g.ksplit = 1;                                       // sixteen times the workgroups of one product: no K split
    g.cps = g.Cp / g.sh.ck;
    g.ws_floats = 16 * g.wp_floats;
    return g;
}
bool wino_gemm_shape_ok(int64_t n, int64_t c, int64_t m, int64_t th, int64_t tw, const char* who) {
    if (n < 0 || c < 1 || m < 1 || th < 1 || tw < 1 || n >= 65536 || c >= 65536 || m >= 65536 || th >= 32768 || tw >= 32768 ||
        n * c * th * tw >= ((int64_t)1 << 31) || n * m * th * tw >= ((int64_t)1 << 31)) {
        fail(SAE_EINVAL, "%s: bad shape", who);
        return false;
    }
    return true;
}
}  // namespace
}  // namespace sae

extern "C" int64_t sae_wino_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w) {
    if (!wino_gemm_shape_ok(n, c, m, tiles_h, tiles_w, "sae_wino_gemm_workspace") || n == 0) return 0;
    return wino_gemm_plan((int)n, (int)c, (int)m, (int)tiles_h, (int)tiles_w).ws_floats;
}

extern "C" int sae_wino_gemm_f32(const float* v, const float* u, float* md, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                                 int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    const char* who = "sae_wino_gemm_f32";
    if (!wino_gemm_shape_ok(n, c, m, tiles_h, tiles_w, who)) return SAE_EINVAL;
    if (n == 0) return SAE_OK;
    if (!v || !u || !md) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (conv_math() != 0) return fail(SAE_EINVAL, "%s: exact-fp32 arithmetic only (SAE_CONV_MATH_F32)", who);
    const int N = (int)n, C = (int)c, M = (int)m, TH = (int)tiles_h, TW = (int)tiles_w;
    const GatherPlan g = wino_gemm_plan(N, C, M, TH, TW);
    if (!workspace || workspace_floats < g.ws_floats)
        return fail(SAE_EWORKSPACE, "%s: workspace %lld < %lld floats", who, (long long)workspace_floats, (long long)g.ws_floats);
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(wino_wprep_kernel, dim3((unsigned)ceil_div64((int64_t)g.Cp * g.Mp, kBlock), 16), dim3(kBlock), 0, s, u,
                       workspace, M, C, g.Mp, g.Cp);
    IgemmParams p{};
    p.N = N; p.C = C; p.H = TH; p.W = TW; p.M = M; p.OH = TH; p.OW = TW; p.YH = TH; p.YW = TW;
    p.oys = 1; p.oxs = 1; p.Cp = g.Cp; p.Mp = g.Mp; p.pad = 0;
    const int64_t tiles = (int64_t)TH * TW;
    p.batch = 16; p.batch_x = (int64_t)N * C * tiles; p.batch_w = g.wp_floats; p.slab_stride = (int64_t)N * M * tiles;
    static const int xcd_knob = tuning_knob("SAE_XCD_ORDER", 1);
    p.xcd_order = xcd_knob;
    p.vec_store = TW % 4 == 0 && (reinterpret_cast<uintptr_t>(md) & 15) == 0;
    const int rc = launch_igemm<1, 1>(v, workspace, md, p, g, s);
    if (rc != SAE_OK) return rc;
    return check_launch(who);
}

extern "C" int sae_modconv2d_fwd_f32(const float* x, const float* w, float* y, const sae_conv2d_desc* d,
                                     const sae_conv2d_mod* mod, float alpha, float* workspace, int64_t workspace_floats,
                                     sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_fwd_impl("sae_modconv2d_fwd_f32", x, w, y, d, mod ? *mod : kNoMod, alpha, workspace, workspace_floats, stream);
}

extern "C" int sae_modconv2d_fwd_noise_bias_act_f32(const float* x, const float* w, const float* noise, const float* noise_weight,
                                                    const float* bias, float* y, const sae_conv2d_desc* d, const sae_conv2d_mod* mod,
                                                    float alpha, float act_slope, float act_scale, float* workspace,
                                                    int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    const char* who = "sae_modconv2d_fwd_noise_bias_act_f32";
    if (!desc_ok(d, who)) return SAE_EINVAL;
    if (d->n == 0) return SAE_OK;
    if (!x || !w || !y || !mod || !mod->x_scale) return fail(SAE_EINVAL, "%s: null tensor (x, w, y and mod->x_scale are required)", who);
    if (mod->y_scale) return fail(SAE_EINVAL, "%s: y_scale has no meaning for the forward operation", who);
    if (noise && !noise_weight) return fail(SAE_EINVAL, "%s: noise needs noise_weight", who);
    if (d->stride != 1) return fail(SAE_EINVAL, "%s: stride 1 only (StyledConv's plain form)", who);
    hipStream_t s = (hipStream_t)stream;
    Epilogue ep{bias, 1, act_slope, act_scale};
    ep.noise = noise; ep.noise_w = noise ? noise_weight : nullptr;
    const PrepGuard prep(prep_from_desc(d));
    int rc = run_gather(x, w, y, workspace, workspace_floats, (int)d->n, (int)d->c, (int)d->h, (int)d->w, (int)d->m,
                        (int)d->oh, (int)d->ow, (int)d->oh, (int)d->ow, 1, 1, d->kh, d->stride, d->pad,
                        d->w_stride_m, d->w_stride_c, 0, alpha, s, ep, mod->x_scale, WScale{mod->wm_scale, mod->wc_scale});
    if (rc != SAE_OK) return rc;
    return check_launch(who);
}

extern "C" int sae_conv2d_fwd_bias_act_f32(const float* x, const float* w, const float* bias, float* y,
                                           const sae_conv2d_desc* d, float alpha, float act_slope, float act_scale,
                                           float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!desc_ok(d, "sae_conv2d_fwd_bias_act_f32")) return SAE_EINVAL;
    if (d->n == 0) return SAE_OK;
    if (!x || !w || !y) return fail(SAE_EINVAL, "sae_conv2d_fwd_bias_act_f32: null tensor");
    hipStream_t s = (hipStream_t)stream;
    const PrepGuard prep(prep_from_desc(d));
    int rc = run_gather(x, w, y, workspace, workspace_floats, (int)d->n, (int)d->c, (int)d->h, (int)d->w, (int)d->m,
                        (int)d->oh, (int)d->ow, (int)d->oh, (int)d->ow, 1, 1, d->kh, d->stride, d->pad,
                        d->w_stride_m, d->w_stride_c, 0, alpha, s, Epilogue{bias, 1, act_slope, act_scale});
    if (rc != SAE_OK) return rc;
    return check_launch("sae_conv2d_fwd_bias_act_f32");
}

extern "C" int sae_conv2d_fwd_residual_f32(const float* x, const float* w, const float* residual, float* y,
                                           const sae_conv2d_desc* d, float alpha, float res_scale, float* workspace,
                                           int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!desc_ok(d, "sae_conv2d_fwd_residual_f32")) return SAE_EINVAL;
    if (d->n == 0) return SAE_OK;
    if (!x || !w || !y || !residual) return fail(SAE_EINVAL, "sae_conv2d_fwd_residual_f32: null tensor");
    if (d->kh != 1)
        return fail(SAE_EINVAL, "sae_conv2d_fwd_residual_f32: 1x1 convolutions only (the skip path of a ResBlock)");
    hipStream_t s = (hipStream_t)stream;
    Epilogue ep{nullptr, 0, 0.0f, 1.0f};
    ep.residual = residual; ep.res_scale = res_scale;
    const PrepGuard prep(prep_from_desc(d));
    int rc = run_gather(x, w, y, workspace, workspace_floats, (int)d->n, (int)d->c, (int)d->h, (int)d->w, (int)d->m,
                        (int)d->oh, (int)d->ow, (int)d->oh, (int)d->ow, 1, 1, d->kh, d->stride, d->pad,
                        d->w_stride_m, d->w_stride_c, 0, alpha, s, ep);
    if (rc != SAE_OK) return rc;
    return check_launch("sae_conv2d_fwd_residual_f32");
}

namespace {
int conv_dgrad_impl(const char* who, const float* gy, const float* w, float* gx, const sae_conv2d_desc* d,
                    const sae_conv2d_mod& mod, float alpha, float* workspace, int64_t workspace_floats,
                    sae_stream_t stream) {
    if (!desc_ok(d, who)) return SAE_EINVAL;
    if (d->n == 0) return SAE_OK;
    if (!gy || !w || !gx) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (mod.x_scale) return fail(SAE_EINVAL, "%s: x_scale has no meaning for the data gradient", who);
    hipStream_t s = (hipStream_t)stream;
    const PrepGuard prep(t_prep.make || t_prep.want_floats ? t_prep : prep_from_desc(d));
    // the launch produces the c axis and contracts over the m axis of the descriptor
    const WScale wsc{mod.wc_scale, mod.wm_scale};
    const Epilogue none{nullptr, 0, 0.0f, 1.0f};
    int rc;
    if (d->stride == 1) {
        // gx = full correlation of gy with the flipped, channel-transposed taps: pad' = k - 1 - pad
        rc = run_gather(gy, w, gx, workspace, workspace_floats, (int)d->n, (int)d->m, (int)d->oh, (int)d->ow,
                        (int)d->c, (int)d->h, (int)d->w, (int)d->h, (int)d->w, 1, 1, d->kh, 1, d->kh - 1 - d->pad,
                        d->w_stride_c, d->w_stride_m, 1, alpha, s, none, mod.y_scale, wsc);
    } else if (d->kh == 1) {
        // 1x1 stride 2 (pad 0): gx[2oy][2ox] = W^T gy, every other position is zero
        if (!(t_prep.make || t_prep.want_floats)) hipMemsetAsync(gx, 0, sizeof(float) * (size_t)(d->n * d->c * d->h * d->w), s);
        rc = run_gather(gy, w, gx, workspace, workspace_floats, (int)d->n, (int)d->m, (int)d->oh, (int)d->ow,
                        (int)d->c, (int)d->oh, (int)d->ow, (int)d->h, (int)d->w, 2, 2, 1, 1, 0, d->w_stride_c,
                        d->w_stride_m, 0, alpha, s, none, mod.y_scale, wsc);
    } else {
        rc = run_tr(gy, w, gx, workspace, workspace_floats, (int)d->n, (int)d->m, (int)d->oh, (int)d->ow, (int)d->c,
                    (int)d->h, (int)d->w, d->pad, d->w_stride_c, d->w_stride_m, alpha, s, mod.y_scale, wsc);
    }
    if (rc != SAE_OK) return rc;
    return check_launch(who);
}
}  // namespace

extern "C" int sae_conv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d,
                                    float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_dgrad_impl("sae_conv2d_dgrad_f32", gy, w, gx, d, kNoMod, alpha, workspace, workspace_floats, stream);
}

extern "C" int sae_modconv2d_dgrad_f32(const float* gy, const float* w, float* gx, const sae_conv2d_desc* d,
                                       const sae_conv2d_mod* mod, float alpha, float* workspace,
                                       int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_dgrad_impl("sae_modconv2d_dgrad_f32", gy, w, gx, d, mod ? *mod : kNoMod, alpha, workspace, workspace_floats,
                           stream);
}

namespace {
int conv_wgrad_impl(const char* who, const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                    const sae_conv2d_mod& mod, float alpha, float* workspace, int64_t workspace_floats,
                    sae_stream_t stream);
}  // namespace

extern "C" int sae_conv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                                    float alpha, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_wgrad_impl("sae_conv2d_wgrad_f32", x, gy, gw, d, kNoMod, alpha, workspace, workspace_floats, stream);
}

// ---- the sixteen products of the Winograd weight gradient: gU[xi][m][c] = sum over images and tiles of E[xi][n][m][t] V[xi][n][c][t],
// each the weight gradient of a 1x1 convolution (K = n * tiles): sixteen launches of the 1x1 weight-gradient kernel + its reduction.
namespace sae {
namespace {
sae_conv2d_desc wino_wgrad_desc(int64_t n, int64_t c, int64_t m, int64_t th, int64_t tw) {
    sae_conv2d_desc d{};
    d.n = n; d.c = c; d.h = th; d.w = tw; d.m = m; d.oh = th; d.ow = tw;
    d.kh = d.kw = 1; d.stride = 1; d.pad = 0;
    d.w_stride_m = c; d.w_stride_c = 1;
    return d;
}
}  // namespace
}  // namespace sae

extern "C" int64_t sae_wino_wgrad_gemm_workspace(int64_t n, int64_t c, int64_t m, int64_t tiles_h, int64_t tiles_w) {
    if (!wino_gemm_shape_ok(n, c, m, tiles_h, tiles_w, "sae_wino_wgrad_gemm_workspace") || n == 0) return 0;
    const sae_conv2d_desc d = wino_wgrad_desc(n, c, m, tiles_h, tiles_w);
    return sae_conv2d_workspace(&d, SAE_CONV_WGRAD);
}

extern "C" int sae_wino_wgrad_gemm_f32(const float* v, const float* e, float* gu, int64_t n, int64_t c, int64_t m, int64_t tiles_h,
                                       int64_t tiles_w, float* workspace, int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    const char* who = "sae_wino_wgrad_gemm_f32";
    if (!wino_gemm_shape_ok(n, c, m, tiles_h, tiles_w, who)) return SAE_EINVAL;
    if (!gu) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (n > 0 && (!v || !e)) return fail(SAE_EINVAL, "%s: null tensor", who);
    const sae_conv2d_desc d = wino_wgrad_desc(n, c, m, tiles_h, tiles_w);
    const int64_t tiles = tiles_h * tiles_w;
    for (int xi = 0; xi < 16; ++xi) {
        const int rc = conv_wgrad_impl(who, v + xi * n * c * tiles, e + xi * n * m * tiles, gu + xi * m * c, &d, kNoMod, 1.0f,
                                       workspace, workspace_floats, stream);
        if (rc != SAE_OK) return rc;
    }
    return SAE_OK;
}

extern "C" int sae_modconv2d_wgrad_f32(const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                                       const sae_conv2d_mod* mod, float alpha, float* workspace,
                                       int64_t workspace_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    return conv_wgrad_impl("sae_modconv2d_wgrad_f32", x, gy, gw, d, mod ? *mod : kNoMod, alpha, workspace, workspace_floats,
                           stream);
}

namespace {
int conv_wgrad_impl(const char* who, const float* x, const float* gy, float* gw, const sae_conv2d_desc* d,
                    const sae_conv2d_mod& mod, float alpha, float* workspace, int64_t workspace_floats,
                    sae_stream_t stream) {
    if (!desc_ok(d, who)) return SAE_EINVAL;
    if (!gw) return fail(SAE_EINVAL, "%s: null gw", who);
    if (d->n > 0 && (!x || !gy)) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (mod.wm_scale || mod.wc_scale) return fail(SAE_EINVAL, "%s: weight factors have no meaning for the weight gradient", who);
    hipStream_t s = (hipStream_t)stream;
    {
        const ThinWgPlan t = thin_wg_plan(d);
        if (t.ok && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0) {
            if (!workspace || workspace_floats < t.ws_floats)
                return fail(SAE_EWORKSPACE, "%s: workspace %lld < %lld floats", who, (long long)workspace_floats, (long long)t.ws_floats);
            ThinWgParams q{};
            q.N = (int)d->n; q.CB = t.cb; q.CS = t.cs; q.split = t.split; q.HW = d->h * d->w; q.quads_per_slice = t.qps;
            q.big_scale = t.big_is_m ? mod.y_scale : mod.x_scale;
            q.small_scale = t.big_is_m ? mod.x_scale : mod.y_scale;
            SAE_TRACE("wgrad thin 1x1: big %d small %d split %d", t.cb, t.cs, t.split);
            hipLaunchKernelGGL(conv1x1_thin_wgrad_kernel, dim3((unsigned)(t.cbp / 4), (unsigned)d->n, (unsigned)t.split), dim3(kBlock), 0,
                               s, t.big_is_m ? gy : x, t.big_is_m ? x : gy, workspace, q);
            hipLaunchKernelGGL(conv1x1_thin_wgrad_reduce_kernel, dim3((unsigned)ceil_div(t.cb * t.cs, kBlock / kWave)), dim3(kBlock), 0, s,
                               (const float*)workspace, gw, t.cb, t.cbp, t.cs, (int)d->n * t.split, t.big_is_m ? 1 : 0, d->w_stride_m,
                               d->w_stride_c, alpha);
            return check_launch(who);
        }
    }
    // second-generation kernel (conv_wgrad16_kernel): 3 x 3, exact fp32, one image per chunk, quad-addressable rows, operand
    // factors (if any) applied per K slice in the reduction
    bool use16 = wg16_shape_ok(d) && (reinterpret_cast<uintptr_t>(gy) & 15) == 0 &&
                 (d->stride == 2 || (reinterpret_cast<uintptr_t>(x) & 15) == 0);
    if (use16 && (mod.x_scale || mod.y_scale)) {
        const WgPlan w16 = wg_plan(d, true);
        const int cpi = w16.tiles_x * w16.tiles_y;
        use16 = w16.cps <= cpi && cpi % w16.cps == 0;
    }
    const WgPlan w = wg_plan(d, use16);
    if ((mod.x_scale || mod.y_scale) && w.bx)
        return fail(SAE_EINVAL, "modulated conv: the activation factors are staged by the exact-fp32 kernels only "
                                "(SAE_CONV_MATH_F32); under bf16x6 modulate the activation before the call");
    const int64_t need = (int64_t)w.slices * w.taps * w.Ap * w.Bp;
    if (!workspace || workspace_floats < need)
        return fail(SAE_EWORKSPACE, "%s: workspace %lld < %lld floats", who, (long long)workspace_floats, (long long)need);
    WgradParams p{};
    p.N = (int)d->n; p.C = (int)d->c; p.H = (int)d->h; p.W = (int)d->w; p.M = (int)d->m; p.OH = (int)d->oh;
    p.OW = (int)d->ow; p.pad = d->pad; p.tw_log2 = w.tw_log2; p.th_log2 = w.th_log2; p.tiles_x = w.tiles_x;
    p.tiles_y = w.tiles_y; p.tiles_n = w.tiles_n; p.chunks = w.chunks; p.chunks_per_slice = w.cps; p.Ap = w.Ap;
    p.Bp = w.Bp;
    // operand modulation: per K-slice in the reduction when every slice lies inside one image (large images: chunks
    // of 64 pixels, one image per chunk, and a slice length that divides the chunks of an image), else per staged
    // element in the MOD instantiation of the main kernel
    int slices_per_image = 0;
    {
        const int tn = kWgPix >> (w.tw_log2 + w.th_log2);
        const int cpi = w.tiles_x * w.tiles_y;
        if ((mod.x_scale || mod.y_scale) && !w.bx && tn == 1 && w.cps <= cpi && cpi % w.cps == 0)
            slices_per_image = cpi / w.cps;
    }
    static const int xcd_knob = tuning_knob("SAE_XCD_ORDER", 1);
    p.xcd_order = xcd_knob;
    p.l_scale = slices_per_image ? nullptr : mod.x_scale;
    p.s_scale = slices_per_image ? nullptr : mod.y_scale;
    if (!w.bx) {
        const int tw = 1 << w.tw_log2, th = 1 << w.th_log2, tn = kWgPix / (tw * th);
        const int ph = (d->kh == 1) ? th : (th - 1) * d->stride + d->kh;
        const int pw = (d->kh == 1) ? tw : (tw - 1) * d->stride + d->kh;
        const int cap = (d->kh == 1) ? 65 : (d->stride == 1 ? 145 : 325);
        if (tn * ph * pw > cap) return fail(SAE_EINVAL, "%s: patch exceeds LDS cap", who);
    }
    // quad staging (see conv_wgrad_kernel): stride 1, one image per 64-pixel chunk, rows of both tensors a multiple of 16
    // bytes, 16-byte aligned tensors, factors (if any) applied per K-slice in the reduction
    static const int wq_knob = tuning_knob("SAE_WGRAD_QUAD", 1);
    const bool wq = wq_knob && !w.bx && w.sh.mode == 0 && d->stride == 1 && (kWgPix >> (w.tw_log2 + w.th_log2)) == 1 &&
                    d->ow % 4 == 0 && d->w % 4 == 0 && ((reinterpret_cast<uintptr_t>(x) | reinterpret_cast<uintptr_t>(gy)) & 15) == 0 &&
                    !p.l_scale && !p.s_scale && (d->kh == 3 ? d->pad <= 4 : d->pad == 0);
    // ... stride 2 (3x3, pad 0): gy rows a multiple of 16 bytes, x rows of any width >= 4 (4-byte aligned quads)
    const bool wq2 = wq_knob && !w.bx && w.sh.mode == 0 && d->kh == 3 && d->stride == 2 && d->pad == 0 &&
                     (kWgPix >> (w.tw_log2 + w.th_log2)) == 1 && d->ow % 4 == 0 && d->w >= 4 &&
                     (reinterpret_cast<uintptr_t>(gy) & 15) == 0 && !p.l_scale && !p.s_scale;
    if (use16) {
        if (p.l_scale || p.s_scale) return fail(SAE_EINVAL, "%s: internal error: wg16 with per-element factors", who);
        const dim3 grid((unsigned)(w.Bp / w.sh.bb), (unsigned)(w.Ap / w.sh.ba), (unsigned)w.slices);
        SAE_TRACE("wgrad wg16 s%d: %d x %d tiles, %d slices of %d chunks", d->stride, w.Ap / w.sh.ba, w.Bp / w.sh.bb, w.slices, w.cps);
        if (d->stride == 1) hipLaunchKernelGGL((conv_wgrad16_kernel<1>), grid, dim3(kBlock), 0, s, x, gy, workspace, p);
        else hipLaunchKernelGGL((conv_wgrad16_kernel<2>), grid, dim3(kBlock), 0, s, x, gy, workspace, p);
    } else if (d->n > 0 && w.bx) {
        WgBxParams q{};
        q.N = p.N; q.C = p.C; q.H = p.H; q.W = p.W; q.M = p.M; q.OH = p.OH; q.OW = p.OW; q.pad = p.pad;
        q.tw8_log2 = w.tw8_log2; q.th_log2 = w.th_log2; q.tiles_x = w.tiles_x; q.tiles_y = w.tiles_y;
        q.tiles_n = w.tiles_n; q.chunks = w.chunks; q.chunks_per_slice = w.cps; q.Ap = w.Ap; q.Bp = w.Bp;
        const dim3 grid((unsigned)(w.Bp / w.sh.bb), (unsigned)(w.Ap / w.sh.ba), (unsigned)w.slices);
        if (d->stride == 1)
            hipLaunchKernelGGL((conv_wgrad_bx_kernel<1, 2, 2>), grid, dim3(kBlock), 0, s, x, gy, workspace, q);
        else
            hipLaunchKernelGGL((conv_wgrad_bx_kernel<2, 4, 1>), grid, dim3(kBlock), 0, s, x, gy, workspace, q);
    } else if (d->n > 0) {
        if (w.sh.mode == 2) {
            if (d->kh == 3 && d->stride == 1) launch_wgrad<3, 1, 1, 1, 1, 1, 2>(x, gy, workspace, p, w, s);
            else if (d->kh == 3) launch_wgrad<3, 2, 1, 1, 1, 1, 2>(x, gy, workspace, p, w, s);
            else if (d->stride == 1) launch_wgrad<1, 1, 1, 1, 1, 1, 2>(x, gy, workspace, p, w, s);
            else launch_wgrad<1, 2, 1, 1, 1, 1, 2>(x, gy, workspace, p, w, s);
        } else if (w.sh.mode == 1) {
            launch_wgrad<3, 1, 1, 1, 1, 1, 1>(x, gy, workspace, p, w, s);
        } else if (d->kh == 3 && d->stride == 1) {
            if (wq) launch_wgrad<3, 1, 1, 1, 2, 2, 0, true>(x, gy, workspace, p, w, s);
            else launch_wgrad<3, 1, 1, 1, 2, 2, 0>(x, gy, workspace, p, w, s);
        }
        else if (d->kh == 3) {
            // operand double buffer: 497 of 512 registers, no spill; 77.7 vs 68.6 TFLOP/s measured
            static const int db_knob = tuning_knob("SAE_WGRAD_S2_DB", 1);
            if (wq2) launch_wgrad<3, 2, 1, 1, 4, 1, 4, true>(x, gy, workspace, p, w, s);
            else if (db_knob) launch_wgrad<3, 2, 1, 1, 4, 1, 4>(x, gy, workspace, p, w, s);
            else launch_wgrad<3, 2, 1, 1, 4, 1, 0>(x, gy, workspace, p, w, s);
        }
        else if (d->stride == 1) {
            if (wq) launch_wgrad<1, 1, 2, 2, 2, 2, 0, true>(x, gy, workspace, p, w, s);
            else launch_wgrad<1, 1, 2, 2, 2, 2, 0>(x, gy, workspace, p, w, s);
        }
        else launch_wgrad<1, 2, 2, 2, 2, 2, 0>(x, gy, workspace, p, w, s);
    }
    const int64_t total = (int64_t)w.taps * d->m * d->c;
    int64_t blocks = ceil_div64(total * 4, kBlock);      // four lanes per output element
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(conv_wgrad_reduce_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, (const float*)workspace,
                       gw, (int)d->m, (int)d->c, w.Ap, w.Bp, w.taps, d->n > 0 ? w.slices : 0, d->w_stride_m,
                       d->w_stride_c, alpha, slices_per_image ? mod.x_scale : nullptr,
                       slices_per_image ? mod.y_scale : nullptr, slices_per_image);
    return check_launch(who);
}
}  // namespace

namespace {
int wprep_dispatch(const char* who, const float* w, const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, float alpha,
                   const PrepRequest& req, sae_stream_t stream) {
    if (op != SAE_CONV_FWD && op != SAE_CONV_DGRAD) return fail(SAE_EINVAL, "%s: op must be SAE_CONV_FWD or SAE_CONV_DGRAD", who);
    const PrepGuard guard(req);
    // the activations are not touched by a query / wprep-only call; the dispatch only looks at their alignment
    float* const stand_in = reinterpret_cast<float*>((uintptr_t)4096);
    const sae_conv2d_mod& m = mod ? *mod : kNoMod;
    if (op == SAE_CONV_FWD) return conv_fwd_impl(who, stand_in, w ? w : stand_in, stand_in, d, m, alpha, nullptr, 0, stream);
    return conv_dgrad_impl(who, stand_in, w ? w : stand_in, stand_in, d, m, alpha, nullptr, 0, stream);
}
}  // namespace

extern "C" int sae_conv2d_wprep_query(const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, int64_t* floats,
                                      int64_t* layout) {
    sae::clear_stale_error();
    if (!floats || !layout) return fail(SAE_EINVAL, "sae_conv2d_wprep_query: null result pointer");
    *floats = 0; *layout = 0;
    if (op == SAE_CONV_WGRAD) return SAE_OK;      // the weight gradient reads no weights
    return wprep_dispatch("sae_conv2d_wprep_query", nullptr, d, mod, op, 1.0f, PrepRequest{nullptr, 0, 0, nullptr, 0, floats, layout}, nullptr);
}

extern "C" int sae_conv2d_wprep_f32(const float* w, const sae_conv2d_desc* d, const sae_conv2d_mod* mod, int32_t op, float alpha,
                                    float* out, int64_t out_floats, sae_stream_t stream) {
    sae::clear_stale_error();
    if (!w || !out || out_floats < 1) return fail(SAE_EINVAL, "sae_conv2d_wprep_f32: null tensor");
    return wprep_dispatch("sae_conv2d_wprep_f32", w, d, mod, op, alpha, PrepRequest{nullptr, 0, 0, out, out_floats, nullptr, nullptr}, stream);
}

#ifdef SAE_CLOCK_PROBE
// profiling variant only: read (and optionally clear) the ten counters of g_clock_probe
extern "C" int sae_debug_clock_probe(unsigned long long* host3, int reset) {
    if (hipMemcpyFromSymbol(host3, HIP_SYMBOL(g_clock_probe), 80) != hipSuccess) return 1;
    if (reset) {
        const unsigned long long z[16] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_clock_probe), z, 128) != hipSuccess) return 1;
    }
    return 0;
}
#endif
