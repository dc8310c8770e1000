// K1 — upfirdn2d: zero-insertion upsample -> pad/crop -> 2-D FIR -> decimate.
//
// Replaces upfirdn2d_op.upfirdn2d (reference: models/networks/stylegan2_op/upfirdn2d.cpp:12-19,
// upfirdn2d_kernel.cu:52-272).  The arithmetic per output follows upfirdn2d_kernel.cu:114-129
// (taps flipped as in :71-81, out-of-range input reads as zero :98-104, output size :167-168),
// with the same accumulation order (tap rows outer, tap columns inner, one fp32 chain).
//
// Two kernels:
//  * blur_kernel<KH,KW,TW,RB> — the train path.  Every upfirdn2d call issued by train.py has
//    up = down = 1: a pure FIR blur with <= 4x4 taps, HBM-bound at
//    4*(numel_in + numel_out) bytes.  Design for gfx950: a wave's 64 lanes own 64 CONSECUTIVE
//    output columns, so every global load/store instruction of a wave is one contiguous
//    256-byte run regardless of the odd row lengths (257, 255, 129 ...) the blurs produce, which
//    rule out 16-byte accesses; each thread register-blocks RB consecutive output rows of its
//    column, so one LDS read feeds up to KH outputs ((RB+KH-1)*KW reads per RB outputs, 4.75 per
//    output at RB = 16 instead of 16).  The input strip of every thread row (RB+KH-1 rows by
//    TW+KW-1 columns, zero-filled outside the image) is staged through LDS once.  Small planes
//    (the 4x4 ... 32x32 tails of D / Dpatch with up to 49152 planes) are packed many
//    strips-per-block (TW = 8/16/32) instead of one mostly idle block per plane as in the CUDA
//    kernel (fixed 16x64 tile, upfirdn2d_kernel.cu:177-222).
//  * upfirdn2d_generic_kernel — any up/down/taps/minor (API parity: Upsample/Downsample,
//    stylegan2_layers.py:38-87).  The reference returns uninitialised memory when no template
//    matches (upfirdn2d_kernel.cu:172-268); here every valid argument set computes.
#include "sae_common.h"

#include <cstdlib>

namespace sae {
namespace {

__device__ __forceinline__ int floor_div_i(int a, int b) {  // upfirdn2d_kernel.cu:18-26
    int c = a / b;
    if (c * b > a) c--;
    return c;
}

// Elementwise work that FOLLOWS a K1 call in the backward pass of a ResBlock, done on the way out (EPI instantiations,
// sae_upfirdn2d_epilogue_f32): the accumulation of a forked gradient (y += v, autograd's add) and / or the K2 backward
// (fused_act.py:32-41) v = (act_ref > 0 ? v : slope v) scale with its bias-gradient partial sums -- the tensor between
// the two ops is never written to or re-read from HBM.
struct K1Epilogue {
    const float* act_ref;    // output-shaped saved activation, or null
    float* partial;          // [channels][outer * q] per-strip partial sums of the activated gradient (act_ref only)
    float slope, scale;
    int accumulate;
    int channels;            // channel of plane p = p % channels
    int64_t q_per_channel;   // outer * (row groups per plane * x tiles)
    // FORWARD activation on the way out (sae_upfirdn2d_noise_bias_act_f32: the blur that ends StyledConv's upsampling conv, followed by
    // NoiseInjection + FusedLeakyReLU, stylegan2_layers.py:398-405): y = lrelu((v + fwd_noise_w[0] * noise[n][pixel]) + bias[c], slope) * scale
    int fwd_act;
    const float* fwd_noise;    // [outer][out_h * out_w] or null
    const float* fwd_noise_w;  // one float on the device
    const float* fwd_bias;     // [channels] or null
};

// The epilogue's own operands (the old value of y, the activation reference) are fetched together with the strip, long
// before the FIR needs them: issued after the FIR they were a second, fully exposed round trip per block.
// ACC = false: instantiations that never accumulate (the blur + activation-backward pass of the ResBlock) do not carry the
// `old` registers -- with the taps in scalar registers that takes the 64 x 8 blur from 84 to 60 VGPRs, 5 to 8 waves per SIMD.
template <int RB, bool ACC = true>
struct K1Operands { float old[ACC ? RB : 1], ref[RB]; };

template <int RB, bool ACC = true>
__device__ __forceinline__ void k1_prefetch(const K1Epilogue& e, K1Operands<RB, ACC>& q, const float* yp, bool col_ok,
                                            int oy0, int out_h, int out_w, int ox, int64_t plane, int64_t plane_elems) {
    const float* rp = e.act_ref ? e.act_ref + plane * plane_elems
                                : (e.fwd_noise ? e.fwd_noise + (plane / e.channels) * plane_elems : yp);
    const bool want_ref = e.act_ref || e.fwd_noise;
#pragma unroll
    for (int o = 0; o < RB; ++o) {
        const int oy = oy0 + o;
        const bool ok = col_ok && oy < out_h;
        const int idx = ok ? oy * out_w + ox : 0;       // branch-free (see blur_kernel): element 0 otherwise; a plane has < 2^31 elements
        if constexpr (ACC) q.old[o] = e.accumulate ? yp[idx] : 0.0f;
        q.ref[o] = want_ref ? rp[idx] : 1.0f;
    }
}

// v: the thread's RB outputs of column ox, rows oy0 ...; writes y and (act_ref) the strip's partial bias-gradient sum
template <int TW, int RB, bool ACC = true>
__device__ __forceinline__ void k1_epilogue(const K1Epilogue& e, const K1Operands<RB, ACC>& q, float (&v)[RB], float* yp,
                                            bool col_ok, int oy0, int out_h, int out_w, int ox, int64_t plane, int strip_q,
                                            int q_per_plane, bool live, int tx) {
    float bsum = 0.0f;
    float fwd_nw = 0.0f, fwd_b = 0.0f;
    if (e.fwd_act) {
        if (e.fwd_noise) fwd_nw = e.fwd_noise_w[0];
        if (e.fwd_bias) fwd_b = e.fwd_bias[plane % e.channels];
    }
    // The prefetched operands (k1_prefetch) have landed before the first row is stored: the compiler cannot carry its vmcnt
    // bookkeeping across the RB skippable row blocks below and put `s_waitcnt vmcnt(0)` into every one of them -- and vmcnt counts
    // stores, so every row waited for the previous row's store to reach the L2: one store in flight per wave (round 6, read off
    // the ISA; profiles/r6_ab_k1_epilogue_wait.txt)
    __builtin_amdgcn_s_waitcnt(0x0F70);         // vmcnt(0), once
    if (col_ok) {
#pragma unroll
        for (int o = 0; o < RB; ++o) {
            const int oy = oy0 + o;
            if (oy < out_h) {
                float t = v[o];
                if constexpr (ACC) {
                    if (e.accumulate) t += q.old[o];
                }
                if (e.act_ref) {
                    t = ((q.ref[o] > 0.0f) ? t : t * e.slope) * e.scale;
                    bsum += t;
                } else if (e.fwd_act) {
                    if (e.fwd_noise) t = t + fwd_nw * q.ref[o];       // (image + weight * noise) + bias, :340-351
                    t = t + fwd_b;
                    t = ((t > 0.0f) ? t : t * e.slope) * e.scale;
                }
                yp[oy * out_w + ox] = t;          // (32-bit index: a plane has < 2^31 elements)
            }
        }
    }
    if (e.act_ref) {       // strip sum over its TW lanes (consecutive lanes of one wave), fixed order
#pragma unroll
        for (int m = TW / 2; m >= 1; m >>= 1) bsum += __shfl_xor(bsum, m, 64);
        if (live && tx == 0) {
            const int64_t n = plane / e.channels, c = plane - n * e.channels;
            e.partial[c * e.q_per_channel + n * q_per_plane + strip_q] = bsum;
        }
    }
}

// XCD-aware workgroup order.  Consecutive workgroup ids are dealt round-robin to the 8 XCDs, each with its own L2; the
// strips of neighbouring workgroups share halo rows (3 of 11 staged rows at 8 rows per thread) and the 128-byte lines that
// straddle the 64-column tile boundaries of the 2^k + 1 wide rows, so in launch order every XCD fetched its own copy: 1102 MB
// read from HBM for 537 MB of input (blur 256^2 -> 257^2, profiles/r3_pmc_f32.txt), i.e. the kernel sat at the HBM roof with
// 1.6x its algorithmic traffic.  XCD k now takes the k-th contiguous eighth of the workgroup list.
// (32-bit throughout: the id is below gridDim.x, and the row-group / plane indices derived from it are below 2^31 -- checked on
// the host, k1_groups_fit -- so that none of the divisions below is a 64-bit one: ~60 instructions each in front of 128 FMAs)
__device__ __forceinline__ unsigned k1_block_id() {
    const unsigned per = gridDim.x >> 3;         // the host rounds the grid up to a multiple of 8
    return (blockIdx.x & 7u) * per + (blockIdx.x >> 3);
}

struct BlurParams {
    int64_t planes;
    int in_h, in_w, out_h, out_w;
    int pad_x0, pad_y0;
    int kh, kw;          // actual tap counts (<= template KH/KW); taps beyond read as zero
    int groups_per_plane;  // ceil(out_h / RB)
    int x_tiles;           // ceil(out_w / TW)
    int64_t groups;        // planes * groups_per_plane
    int64_t blocks;        // workgroups that have work (the grid is rounded up to a multiple of 8, see k1_block_id)
    int main_per_unit;     // blur_tail_kernel: 64-column workgroups per unit of 32 row groups ((x_tiles - 1) * 8)
};

// One strip: RB output rows by TW output columns starting at column ox0 of row group g (global index over all planes), staged
// through the thread row's LDS strip `sp`.  `xt` is the strip's x-tile index in the epilogue's partial-sum layout.
template <int KH, int KW, int TW, int RB, bool EPI, bool ACC = true>
__device__ __forceinline__ void blur_strip(const float* __restrict__ x, float* __restrict__ y, const BlurParams& p,
                                           const K1Epilogue& e, float* sp, const float* taps, const int tx, const unsigned g_in,
                                           const int xt, const int ox0) {
    constexpr int SR = RB + KH - 1;          // staged rows per strip
    constexpr int SW = TW + KW - 1;          // staged columns per strip
    constexpr int SWP = SW | 1;              // odd row stride
    // With 64-column strips a wave IS one thread row: the row group, and with it the plane, the strip's rows and their validity,
    // are wave-uniform -- said so, all of that arithmetic runs on the scalar unit and the row of a load rides in its scalar offset
    const unsigned g = TW == 64 ? (unsigned)__builtin_amdgcn_readfirstlane((int)g_in) : g_in;
    const bool live = g < (unsigned)p.groups;
    const unsigned plane_u = live ? g / (unsigned)p.groups_per_plane : 0u;
    const int64_t plane = plane_u;
    const int oy0 = live ? (int)(g - plane_u * (unsigned)p.groups_per_plane) * RB : 0;
    const int iy0 = oy0 - p.pad_y0;  // input row of staged row 0
    const int ix0 = ox0 - p.pad_x0;  // input column of staged column 0

    // The strip is fetched through range-checked buffer loads with 32-bit per-lane offsets relative to the first plane the wave
    // touches: an element outside the image carries an out-of-range offset and reads as zero (no select), and the address of a
    // row is one 32-bit add.  With 64-bit indexing (a v_mad_u64_u32 and a 64-bit add per load and store, selects for the zero
    // padding) the kernel issued ~7 instructions per FMA and was bound by instruction issue, not by HBM: the form with an
    // epilogue moved 50 % more bytes in the same time (round 6, HISTORY 4.0i)
    const int64_t hw_in = (int64_t)p.in_h * p.in_w;
    const int64_t plane0 = __builtin_amdgcn_readfirstlane((int)plane);
    int64_t span = (p.planes - plane0) * hw_in * 4;
    if (span > 0x7fffffff) span = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + plane0 * hw_in), 0, (unsigned)span,
                                                                       0x00020000);
    const int rel = (int)((plane - plane0) * hw_in) + iy0 * p.in_w + ix0;       // element offset of staged (row 0, column 0)
    // all global loads of the strip are issued back to back into registers, then written to LDS: one load in flight
    // per wave would leave HBM latency fully exposed.  The strip is fetched ROW-wise: lane tx takes column tx of each of
    // the SR rows (one contiguous TW * 4-byte run per instruction, no per-element index division — the generic
    // e -> (e / SW, e % SW) walk cost ~30 VALU instructions per output, more than the 16 FMAs of the filter), and the
    // KW - 1 halo columns of all rows are gathered by NX more loads.
    constexpr int XW = KW - 1;                               // halo columns right of the TW body
    constexpr int NX = (XW * SR + TW - 1) / TW;              // loads per thread for them (1 at TW = 64)
    float body[SR];
    float halo[NX > 0 ? NX : 1];
    const int ixb = ix0 + tx;
    const bool col_ok = live && ixb >= 0 && ixb < p.in_w;
    [[maybe_unused]] const unsigned vcol = col_ok ? (unsigned)(ixb * 4) : 0x80000000u;   // TW == 64: the lane's column, once
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int iy = iy0 + r;
        // branch-free: a load inside a divergent branch makes hipcc drain vmcnt at the join, which left two
        // loads in flight per wave; out-of-image elements carry an out-of-range offset and read as zero
        if constexpr (TW == 64) {
            const bool row_ok = iy >= 0 && iy < p.in_h;                                  // scalar
            const unsigned soff = row_ok ? (unsigned)(((int)((plane - plane0) * hw_in) + iy * p.in_w) * 4) : 0u;
            body[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, row_ok ? vcol : 0x80000000u, soff, 0));
        } else {
            const bool ok = col_ok && iy >= 0 && iy < p.in_h;
            const unsigned off = ok ? (unsigned)((rel + r * p.in_w + tx) * 4) : 0x80000000u;
            body[r] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tx + i * TW;
        const int r = e / (XW > 0 ? XW : 1), c = TW + e - r * XW;
        const int iy = iy0 + r, ix = ix0 + c;
        const bool ok = live && e < XW * SR && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        const unsigned off = ok ? (unsigned)((rel + r * p.in_w + c) * 4) : 0x80000000u;
        halo[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
    }
    [[maybe_unused]] K1Operands<EPI ? RB : 1, ACC> eq;
    if constexpr (EPI)
        k1_prefetch<RB, ACC>(e, eq, y + plane * (int64_t)p.out_h * p.out_w, live && ox0 + tx < p.out_w, oy0, p.out_h, p.out_w,
                        ox0 + tx, plane, (int64_t)p.out_h * p.out_w);
#pragma unroll
    for (int r = 0; r < SR; ++r) sp[r * SWP + tx] = body[r];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tx + i * TW;
        const int r = e / (XW > 0 ? XW : 1), c = TW + e - r * XW;
        if (e < XW * SR) sp[r * SWP + c] = halo[i];
    }
    __syncthreads();

    float tap[KH][KW];
#pragma unroll
    for (int a = 0; a < KH; ++a)
#pragma unroll
        for (int b = 0; b < KW; ++b)      // wave-uniform: kept in scalar registers (16 VGPRs less at 4 x 4 taps)
            tap[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, taps[a * KW + b])));

    // (measured and dropped, same-box A/B with tools/kb_k1.py: the rows accumulated in pairs as packed fp32, v_pk_fma_f32
    // with the staged value broadcast against a pair of taps held in scalar registers -- 20 instead of 32 vector
    // instructions per pair of outputs and 54 instead of 67 VGPRs, yet 3.13 vs 3.46 TB/s at 256^2 and 4.10 vs 4.33 at
    // 257 -> 256: the kernel is not bound by its FMA issue)
    float acc[RB];
#pragma unroll
    for (int o = 0; o < RB; ++o) acc[o] = 0.0f;

#pragma unroll
    for (int r = 0; r < SR; ++r) {
        float v[KW];
#pragma unroll
        for (int c = 0; c < KW; ++c) v[c] = sp[r * SWP + tx + c];
#pragma unroll
        for (int ky = 0; ky < KH; ++ky) {
            const int o = r - ky;  // output row fed by staged row r through tap row ky
            if (o >= 0 && o < RB) {
#pragma unroll
                for (int c = 0; c < KW; ++c) acc[o] = fmaf(v[c], tap[ky][c], acc[o]);
            }
        }
    }

    const int ox = ox0 + tx;
    if constexpr (EPI) {
        const int gi = (int)(g - plane_u * (unsigned)p.groups_per_plane);
        k1_epilogue<TW, RB, ACC>(e, eq, acc, y + plane * (int64_t)p.out_h * p.out_w, live && ox < p.out_w, oy0, p.out_h, p.out_w, ox,
                            plane, gi * p.x_tiles + xt, p.groups_per_plane * p.x_tiles, live, tx);
        return;
    }
    if (live && ox < p.out_w) {
        float* yb = y + (plane * (int64_t)p.out_h + oy0) * p.out_w + ox;       // one 64-bit address; rows are 32-bit steps from it
#pragma unroll
        for (int o = 0; o < RB; ++o)
            if (oy0 + o < p.out_h) yb[o * p.out_w] = acc[o];
    }
}

template <int KH, int KW, int TW, int RB, bool EPI = false, bool ACC = true>
__global__ __launch_bounds__(kBlock) void blur_kernel(const float* __restrict__ x,
                                                      const float* __restrict__ k,
                                                      float* __restrict__ y, const BlurParams p, const K1Epilogue e) {
    constexpr int NR = kBlock / TW;          // thread rows per block
    constexpr int SR = RB + KH - 1;
    constexpr int SWP = (TW + KW - 1) | 1;
    __shared__ float strip[NR][SR * SWP];
    __shared__ float taps[KH * KW];

    const int tx = threadIdx.x % TW;
    const int tr = threadIdx.x / TW;

    // flipped taps, zero padded to the template size (upfirdn2d_kernel.cu:71-81)
    if (threadIdx.x < KH * KW) {
        const int ky = threadIdx.x / KW, kx = threadIdx.x % KW;
        float v = 0.0f;
        if (ky < p.kh && kx < p.kw) v = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
        taps[threadIdx.x] = v;
    }

    const unsigned bid = k1_block_id();
    if (bid >= (unsigned)p.blocks) return;          // (whole workgroup: the grid's padding to a multiple of 8)
    const unsigned bq = bid / (unsigned)p.x_tiles;
    const int xt = (int)(bid - bq * (unsigned)p.x_tiles);
    const unsigned g = bq * NR + tr;                // strip (row group) handled by this thread row
    blur_strip<KH, KW, TW, RB, EPI, ACC>(x, y, p, e, strip[tr], taps, tx, g, xt, xt * TW);
}

// The 2^k + 1 wide outputs of the train step (65, 129, 257: every blur in front of a stride-2 conv) leave ONE column for the
// last 64-column tile: a fifth, a third, half of the wavefronts ran with one live lane, and the kernel's rate followed the
// live fraction (4.9 TB/s at 256 columns, 4.1 - 4.4 at 257, 3.7 at 129, 2.6 at 65: gpurun r3C by-shape ledger).  Here the
// remainder columns (1 ... 8) of 32 row groups are packed into one workgroup of 8-lane thread rows, placed right behind the 64-column
// workgroups of the same row groups (it re-reads the last columns of their input from the same XCD's L2):
//   unit u = [ main_per_unit workgroups of 4 row groups x 64 columns | 1 workgroup of 32 row groups x 8 columns ]
// The partial-sum layout of the epilogue is unchanged (the remainder is x tile x_tiles - 1).
template <int KH, int KW, int RB, bool EPI = false, bool ACC = true>
__global__ __launch_bounds__(kBlock) void blur_tail_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                           float* __restrict__ y, const BlurParams p, const K1Epilogue e) {
    constexpr int SR = RB + KH - 1;
    constexpr int SWP_M = (64 + KW - 1) | 1, SWP_T = (8 + KW - 1) | 1;
    constexpr int LDS_M = (kBlock / 64) * SR * SWP_M, LDS_T = (kBlock / 8) * SR * SWP_T;
    __shared__ float strips[LDS_M > LDS_T ? LDS_M : LDS_T];
    __shared__ float taps[KH * KW];
    if (threadIdx.x < KH * KW) {
        const int ky = threadIdx.x / KW, kx = threadIdx.x % KW;
        float v = 0.0f;
        if (ky < p.kh && kx < p.kw) v = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
        taps[threadIdx.x] = v;
    }
    const unsigned bid = k1_block_id();
    if (bid >= (unsigned)p.blocks) return;
    const unsigned unit = bid / (unsigned)(p.main_per_unit + 1);
    const int r = (int)(bid - unit * (unsigned)(p.main_per_unit + 1));
    if (r < p.main_per_unit) {
        const int xm = p.x_tiles - 1;
        const int rq = r / xm;
        const int xt = r - rq * xm;
        const int tr = threadIdx.x / 64;
        const unsigned g = (unit * 8 + rq) * 4 + tr;
        blur_strip<KH, KW, 64, RB, EPI, ACC>(x, y, p, e, strips + tr * SR * SWP_M, taps, threadIdx.x % 64, g, xt, xt * 64);
    } else {
        const int tr = threadIdx.x / 8;
        blur_strip<KH, KW, 8, RB, EPI, ACC>(x, y, p, e, strips + tr * SR * SWP_T, taps, threadIdx.x % 8, unit * 32 + tr, p.x_tiles - 1,
                                       (p.x_tiles - 1) * 64);
    }
}

// ------------------------------------------------------------------------------------------
// Streaming blur (round 6): the up = down = 1 FIR of <= 4 x 4 taps on planes at least 32 floats wide, WITHOUT the LDS strip.
//
// blur_strip moves one dword per lane and instruction (a wave = one 256-byte run) and re-reads 3 of every 11 staged rows; the
// big planes of the step ran it at 3.3 - 4.5 TB/s where the elementwise kernels, which move 16 bytes per lane, reach 5.5.  Here
// a thread owns FOUR consecutive output columns of a vertical strip of `rb` output rows (32 or so: 3 halo rows re-read per
// strip) and walks down the strip with the last KH input rows' contributions in registers:
//   * per input row two 16-byte buffer loads fetch the 8 floats [4 j - pad_x0, 4 j - pad_x0 + 8) its four outputs draw on (only
//     4-byte aligned -- rows are 2^k or 2^k + 1 floats -- which gfx950's vector memory takes); neighbouring lanes overlap by
//     4 floats, which the L1 absorbs; the loads of the NEXT four rows are in flight while four rows are consumed;
//   * the descriptor starts at the wave's first plane, the per-lane offset is relative to it, and the range check returns zeros
//     before the tensor's first and after its last float; window columns left or right of the ROW (the zero padding of
//     upfirdn2d_kernel.cu:98-104, which the check cannot see: they are the neighbouring row's data) are zeroed by a
//     loop-invariant per-lane mask, rows above / below the plane by an out-of-range offset;
//   * four accumulator rows (a ring over output rows): input row i feeds outputs i - ky through tap row ky, in exactly the
//     order blur_strip uses (tap rows outer, tap columns inner, one fp32 fma chain per output): BIT-IDENTICAL results;
//   * the forward epilogue of StyledConv's upsampling form (noise, bias, leaky ReLU: K1Epilogue::fwd_act) on the way out;
//     each finished row leaves as one 16-byte store per lane.
struct StreamParams {
    int64_t planes, slots;     // slots = planes * strips
    int in_h, in_w, out_h, out_w;
    int pad_x0, pad_y0;
    int kh, kw;
    int lpr;                   // lanes per output row = ceil(out_w / 4)
    int rb, strips;            // output rows per strip (a multiple of 4), strips per plane
};

typedef float f32x4a4 __attribute__((ext_vector_type(4), aligned(4)));

template <bool FWD_ACT>
__global__ __launch_bounds__(kBlock) void blur_stream_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                             float* __restrict__ y, const StreamParams p, const K1Epilogue e) {
    const int64_t gt = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    const int64_t slot_raw = gt / p.lpr;
    const int j = (int)(gt - slot_raw * p.lpr);
    const bool live = slot_raw < p.slots;
    const int64_t slot = live ? slot_raw : p.slots - 1;
    const int64_t plane = slot / p.strips;
    const int oy0 = (int)(slot - plane * p.strips) * p.rb;
    const int ox0 = 4 * j;

    // flipped taps, zero beyond the actual counts (upfirdn2d_kernel.cu:71-81); uniform -> scalar registers
    float tap[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const bool has = a < p.kh && b < p.kw;
            const float t = k[has ? (p.kh - 1 - a) * p.kw + (p.kw - 1 - b) : 0];
            tap[a][b] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, has ? t : 0.0f)));
        }

    // descriptor: from the first plane any lane of this wave touches to the end of the tensor (at most 2 GiB - 1)
    const int64_t hw = (int64_t)p.in_h * p.in_w;
    const int64_t plane0 = __builtin_amdgcn_readfirstlane((int)plane);
    int64_t span = (p.planes - plane0) * hw * 4;
    if (span > 0x7fffffff) span = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + plane0 * hw), 0, (unsigned)span, 0x00020000);
    // The 8-float window of the lane starts at column c0 = ox0 - pad_x0; the lanes at the left border (c0 < 0) fetch theirs from
    // column 0 instead and move it into place (`shift` columns) -- an offset left of the tensor's first float would have to rely
    // on how the hardware wraps a negative offset.  Every offset below is >= 0 for a row inside the plane.
    const int c0 = ox0 - p.pad_x0;
    const int shift = c0 < 0 ? -c0 : 0;                       // 0 ... 3 (pad_x0 <= kw - 1 <= 3)
    // per-row offset = this + i * row_bytes (i = 0: input row oy0 - pad_y0, possibly above the plane: masked below)
    const int64_t rel = ((plane - plane0) * p.in_h + (oy0 - p.pad_y0)) * (int64_t)p.in_w + c0 + shift;
    const int rel_bytes = (int)(rel * 4);
    const int row_bytes = p.in_w * 4;
    bool colok[8];
#pragma unroll
    for (int q = 0; q < 8; ++q) colok[q] = c0 + q >= 0 && c0 + q < p.in_w;

    auto load_row = [&](int i, f32x4& lo, f32x4& hi) {        // input row oy0 - pad_y0 + i
        const int iy = oy0 - p.pad_y0 + i;
        const bool ok = live && iy >= 0 && iy < p.in_h && i < p.rb + 3;
        const unsigned v = ok ? (unsigned)(rel_bytes + i * row_bytes) : 0x80000000u;
        lo = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, v, 0, 0));
        hi = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(xrs, v + 16u, 0, 0));
    };

    float acc[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) acc[a][b] = 0.0f;

    [[maybe_unused]] float fwd_nw = 0.0f, fwd_b = 0.0f;
    [[maybe_unused]] const float* zp = nullptr;
    if constexpr (FWD_ACT) {
        if (e.fwd_noise) {
            fwd_nw = e.fwd_noise_w[0];
            zp = e.fwd_noise + (plane / e.channels) * (int64_t)p.out_h * p.out_w;
        }
        if (e.fwd_bias) fwd_b = e.fwd_bias[plane % e.channels];
    }
    float* yp = y + plane * (int64_t)p.out_h * p.out_w;
    const bool full = ox0 + 3 < p.out_w;          // all four columns exist: one 16-byte store
    // The noise quads of an iteration's four output rows are requested together at the top of the iteration, BEFORE the next rows
    // of x, as range-checked buffer loads over the whole noise tensor (a row outside the strip reads zeros; a quad at a row's end
    // reaches into the next row or past the tensor: values nobody uses) -- fetched inside each row's store block they were the
    // NEWEST load in flight, and waiting for the newest load drains everything before it: the eight prefetched rows of x and the
    // previous row's store, four times per iteration (round 6, read off the ISA)
    [[maybe_unused]] __amdgpu_buffer_rsrc_t zrs = xrs;
    [[maybe_unused]] int zrel = 0;
    if constexpr (FWD_ACT) {
        if (zp) {
            int64_t zbytes = (p.planes / e.channels) * (int64_t)p.out_h * p.out_w * 4;
            if (zbytes > 0x7fffffff) zbytes = 0x7fffffff;
            zrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(e.fwd_noise), 0, (unsigned)zbytes, 0x00020000);
            zrel = (int)((((plane / e.channels) * p.out_h + oy0) * (int64_t)p.out_w + ox0) * 4);
        }
    }
    [[maybe_unused]] auto load_z = [&](int o) {                // the quad of output row oy0 + o
        const bool ok = live && o >= 0 && o < p.rb && oy0 + o < p.out_h;
        const unsigned v = ok ? (unsigned)(zrel + o * p.out_w * 4) : 0x80000000u;
        return __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(zrs, v, 0, 0));
    };

    f32x4 nlo[4], nhi[4];
#pragma unroll
    for (int s4 = 0; s4 < 4; ++s4) load_row(s4, nlo[s4], nhi[s4]);
    for (int i0 = 0; i0 < p.rb + 3; i0 += 4) {
        float v[4][8];
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            float w[8];
#pragma unroll
            for (int q = 0; q < 8; ++q) w[q] = q < 4 ? nlo[s4][q] : nhi[s4][q - 4];
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                // window element q is fetched element q - shift
                float t = w[q];
                if (q >= 1) t = shift == 1 ? w[q - 1] : t;
                if (q >= 2) t = shift == 2 ? w[q - 2] : t;
                if (q >= 3) t = shift == 3 ? w[q - 3] : t;
                v[s4][q] = colok[q] ? t : 0.0f;
            }
        }
        [[maybe_unused]] f32x4 zq[4];
        if constexpr (FWD_ACT) {
            if (zp) {
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) zq[s4] = load_z(i0 + s4 - 3);
            }
        }
        // (always issued: beyond the strip's last row load_row masks the offset and the load reads zeros -- an `if` around the
        // prefetch would hide from the compiler how many loads are in flight behind the noise quads)
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) load_row(i0 + 4 + s4, nlo[s4], nhi[s4]);
        if constexpr (FWD_ACT) {
            // the quads have landed (the compiler waits for exactly them in front of this empty asm: the x rows behind them stay
            // in flight) -- the skippable store blocks below then need no wait of their own
            if (zp) asm volatile("" : "+v"(zq[0]), "+v"(zq[1]), "+v"(zq[2]), "+v"(zq[3]));
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            // input row i = i0 + s4 feeds output row i - ky through tap row ky; ring slot of output row o is o & 3
#pragma unroll
            for (int ky = 0; ky < 4; ++ky) {
                const int rs = (s4 - ky + 4) & 3;
#pragma unroll
                for (int xo = 0; xo < 4; ++xo)
#pragma unroll
                    for (int c = 0; c < 4; ++c) acc[rs][xo] = fmaf(v[s4][xo + c], tap[ky][c], acc[rs][xo]);
            }
            // output row o = i - 3 is complete (its last contribution came through tap row 3)
            const int o = i0 + s4 - 3;
            const int rs = (s4 + 1) & 3;            // = (s4 - 3 + 4) & 3
            const int oy = oy0 + o;
            if (live && o >= 0 && o < p.rb && oy < p.out_h) {
                float t[4];
#pragma unroll
                for (int xo = 0; xo < 4; ++xo) t[xo] = acc[rs][xo];
                if constexpr (FWD_ACT) {
                    float z[4] = {0.0f, 0.0f, 0.0f, 0.0f};
                    if (zp) {
#pragma unroll
                        for (int xo = 0; xo < 4; ++xo) z[xo] = zq[s4][xo];
                    }
#pragma unroll
                    for (int xo = 0; xo < 4; ++xo) {
                        float u = t[xo];
                        if (zp) u = u + fwd_nw * z[xo];             // (image + weight * noise) + bias, stylegan2_layers.py:340-351
                        u = u + fwd_b;
                        t[xo] = ((u > 0.0f) ? u : u * e.slope) * e.scale;
                    }
                }
                float* yr = yp + (int64_t)oy * p.out_w + ox0;
                if (full) {
                    *reinterpret_cast<f32x4a4*>(yr) = f32x4a4{t[0], t[1], t[2], t[3]};
                } else {
#pragma unroll
                    for (int xo = 0; xo < 4; ++xo)
                        if (ox0 + xo < p.out_w) yr[xo] = t[xo];
                }
            }
#pragma unroll
            for (int xo = 0; xo < 4; ++xo) acc[rs][xo] = 0.0f;
        }
    }
}

// The planes the streaming kernel takes -- where the same-box A/B says it wins (tools/ab_k1_stream.py ->
// profiles/r6_ab_k1_stream.txt, TB/s of algorithmic bytes, strip kernel -> streaming kernel):
//   blur + noise + bias + leaky-ReLU forward (2^k + 1 -> 2^k wide): 257: 3.4 -> 4.4, 129: 3.5 -> 4.4 / 3.9 -> 4.5, 513: 3.5 -> 4.3;
//     65: 3.9 -> 4.1 (even), 33: 2.9 -> 2.6 (loses)                                             => output rows of 96 floats and up
//   plain 4 x 4 blur (2^k -> 2^k + 1 wide): 32: 2.6 -> 3.2 / 2.9 -> 4.2; 64 ... 1024: 4.0 - 4.5 -> 3.9 - 4.2 (loses 2 - 9 %: its
//     16-byte stores start on 4-byte boundaries of the 2^k + 1 wide rows)                          => the 32-wide planes only
//   3-tap blurs: 4.8 - 5.7 -> 4.2 (the strip kernel stages 10 rows for 8 there)                    => never
// SAE_K1_STREAM (tuning builds): 0 = never, 1 = this rule, 2 = every plane the kernel can take (the bit-identity tests)
inline bool blur_streams(const BlurParams& p, bool fwd_act) {
    const int knob = tuning_knob("SAE_K1_STREAM", 1);
    const bool can = p.kh <= 4 && p.kw <= 4 && p.kh >= 2 && p.in_w >= 32 && p.out_w >= 32 && p.out_h >= 16;
    if (knob == 0 || !can) return false;
    if (knob == 2) return true;
    if (p.kh != 4 || p.kw != 4) return false;
    return fwd_act ? p.out_w >= 96 : (p.in_w == 32 && p.in_h == 32);
}

template <bool FWD_ACT>
void launch_blur_stream(const float* x, const float* k, float* y, const BlurParams& b, hipStream_t s, const K1Epilogue& e) {
    StreamParams p{};
    p.planes = b.planes;
    p.in_h = b.in_h; p.in_w = b.in_w; p.out_h = b.out_h; p.out_w = b.out_w;
    p.pad_x0 = b.pad_x0; p.pad_y0 = b.pad_y0; p.kh = b.kh; p.kw = b.kw;
    p.lpr = ceil_div(b.out_w, 4);
    // about 32 output rows per strip, strips of equal height (a 257-row plane: 8 strips of 36 / 5 rows, not 8 x 32 + 1)
    const int nstrips = b.out_h >= 48 ? (b.out_h + 16) / 32 : 1;
    p.rb = ceil_div(ceil_div(b.out_h, nstrips), 4) * 4;
    p.strips = ceil_div(b.out_h, p.rb);
    p.slots = p.planes * p.strips;
    const int64_t blocks = ceil_div64(p.slots * p.lpr, kBlock);
    hipLaunchKernelGGL((blur_stream_kernel<FWD_ACT>), dim3((unsigned)blocks), dim3(kBlock), 0, s, x, k, y, p, e);
}

struct GenericParams {
    int64_t major, minor;
    int in_h, in_w, out_h, out_w;
    int kh, kw, up_x, up_y, down_x, down_y, pad_x0, pad_y0;
    int64_t total;  // major * out_h * out_w * minor
};

__global__ __launch_bounds__(kBlock) void upfirdn2d_generic_kernel(const float* __restrict__ x,
                                                                   const float* __restrict__ k,
                                                                   float* __restrict__ y,
                                                                   const GenericParams p) {
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < p.total;
         i += (int64_t)gridDim.x * kBlock) {
        int64_t t = i;
        const int64_t mn = t % p.minor; t /= p.minor;
        const int ox = (int)(t % p.out_w); t /= p.out_w;
        const int oy = (int)(t % p.out_h);
        const int64_t mj = t / p.out_h;
        const int mid_x = ox * p.down_x + p.up_x - 1 - p.pad_x0;
        const int mid_y = oy * p.down_y + p.up_y - 1 - p.pad_y0;
        const int in_x = floor_div_i(mid_x, p.up_x);
        const int in_y = floor_div_i(mid_y, p.up_y);
        const int kx0 = (in_x + 1) * p.up_x - mid_x - 1;
        const int ky0 = (in_y + 1) * p.up_y - mid_y - 1;
        float v = 0.0f;
        for (int yy = 0; ky0 + yy * p.up_y < p.kh; ++yy) {
            const int iy = in_y + yy;
            if (iy < 0 || iy >= p.in_h) continue;
            const int ky = ky0 + yy * p.up_y;
            for (int xx = 0; kx0 + xx * p.up_x < p.kw; ++xx) {
                const int ix = in_x + xx;
                if (ix < 0 || ix >= p.in_w) continue;
                const int kx = kx0 + xx * p.up_x;
                const float tapv = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
                v = fmaf(x[((mj * p.in_h + iy) * p.in_w + ix) * p.minor + mn], tapv, v);
            }
        }
        y[i] = v;
    }
}

// x2 decimation (up 1, down 2) and x2 zero-insertion upsampling (up 2, down 1) with <= 4x4 taps:
// the pair that appears when the skip path of a downsampling block decimates inside the FIR
// (forward = down 2, backward = up 2).  Same layout as the blur kernel: lanes own consecutive
// OUTPUT columns, the input strip a thread row needs is staged to LDS once, per-output arithmetic
// is upfirdn2d_kernel.cu:114-129 with compile-time up/down.
struct UpDownParams {
    int64_t planes;
    int in_h, in_w, out_h, out_w;
    int pad_x0, pad_y0;
    int kh, kw;
    int groups_per_plane, x_tiles;
    int64_t groups;
    int64_t blocks;
};

template <int UP, int DOWN, int KH, int KW, int TW, int RB, bool EPI = false>
__global__ __launch_bounds__(kBlock) void updown_kernel(const float* __restrict__ x, const float* __restrict__ k,
                                                        float* __restrict__ y, const UpDownParams p, const K1Epilogue e) {
    constexpr int NR = kBlock / TW;
    constexpr int SR = ((RB - 1) * DOWN + KH - 1) / UP + 1;     // upfirdn2d_kernel.cu:54-55
    constexpr int SW = ((TW - 1) * DOWN + KW - 1) / UP + 1;
    constexpr int SWP = SW | 1;
    __shared__ float strip[NR][SR * SWP];
    __shared__ float taps[KH * KW];

    const int tx = threadIdx.x % TW;
    const int tr = threadIdx.x / TW;
    if (threadIdx.x < KH * KW) {
        const int ky = threadIdx.x / KW, kx = threadIdx.x % KW;
        float v = 0.0f;
        if (ky < p.kh && kx < p.kw) v = k[(p.kh - 1 - ky) * p.kw + (p.kw - 1 - kx)];
        taps[threadIdx.x] = v;
    }
    const unsigned bid = k1_block_id();
    if (bid >= (unsigned)p.blocks) return;
    const unsigned bq = bid / (unsigned)p.x_tiles;
    const int xt = (int)(bid - bq * (unsigned)p.x_tiles);
    const unsigned g = bq * NR + tr;
    const bool live = g < (unsigned)p.groups;
    const unsigned plane_u = live ? g / (unsigned)p.groups_per_plane : 0u;
    const int64_t plane = plane_u;
    const int oy0 = live ? (int)(g - plane_u * (unsigned)p.groups_per_plane) * RB : 0;
    const int ox0 = xt * TW;
    const int tile_mid_y = oy0 * DOWN + UP - 1 - p.pad_y0;
    const int tile_mid_x = ox0 * DOWN + UP - 1 - p.pad_x0;
    const int tile_in_y = floor_div_i(tile_mid_y, UP);
    const int tile_in_x = floor_div_i(tile_mid_x, UP);

    // (range-checked buffer loads with 32-bit offsets relative to the wave's first plane, as in blur_strip)
    const int64_t hw_in = (int64_t)p.in_h * p.in_w;
    const int64_t plane0 = __builtin_amdgcn_readfirstlane((int)plane);
    int64_t span = (p.planes - plane0) * hw_in * 4;
    if (span > 0x7fffffff) span = 0x7fffffff;
    const __amdgpu_buffer_rsrc_t xrs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x + plane0 * hw_in), 0, (unsigned)span,
                                                                       0x00020000);
    [[maybe_unused]] const int rel = (int)((plane - plane0) * hw_in) + tile_in_y * p.in_w + tile_in_x;   // element offset of staged (0, 0)
    [[maybe_unused]] const float* xg = x + plane * hw_in;
    float* sp = strip[tr];
    // row-wise staging as in blur_kernel: CB full TW-wide column blocks per row (lane tx takes column tx + b * TW),
    // the SW - CB * TW remaining columns of all rows gathered by NX more loads; no per-element index division
    constexpr int CB = SW / TW;
    constexpr int XW = SW - CB * TW;
    constexpr int NX = (XW * SR + TW - 1) / TW;
    float body[SR][CB > 0 ? CB : 1];
    float halo[NX > 0 ? NX : 1];
#pragma unroll
    for (int r = 0; r < SR; ++r) {
        const int iy = tile_in_y + r;
        const bool row_ok = live && iy >= 0 && iy < p.in_h;
#pragma unroll
        for (int b = 0; b < CB; ++b) {
            const int ix = tile_in_x + tx + b * TW;
            const bool ok = row_ok && ix >= 0 && ix < p.in_w;   // branch-free, see blur_kernel
            if constexpr (DOWN == 1) {
                const unsigned off = ok ? (unsigned)((rel + r * p.in_w + tx + b * TW) * 4) : 0x80000000u;
                body[r][b] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
            } else {       // (the decimating form measured 2 % slower on buffer loads: 5.78 -> 5.66 TB/s)
                const float v = xg[ok ? (int64_t)iy * p.in_w + ix : 0];
                body[r][b] = ok ? v : 0.0f;
            }
        }
    }
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tx + i * TW;
        const int r = e / (XW > 0 ? XW : 1), c = CB * TW + e - r * XW;
        const int iy = tile_in_y + r, ix = tile_in_x + c;
        const bool ok = live && e < XW * SR && iy >= 0 && iy < p.in_h && ix >= 0 && ix < p.in_w;
        if constexpr (DOWN == 1) {
            const unsigned off = ok ? (unsigned)((rel + r * p.in_w + c) * 4) : 0x80000000u;
            halo[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(xrs, off, 0, 0));
        } else {
            const float v = xg[ok ? (int64_t)iy * p.in_w + ix : 0];
            halo[i] = ok ? v : 0.0f;
        }
    }
    [[maybe_unused]] K1Operands<EPI ? RB : 1> eq;
    if constexpr (EPI)
        k1_prefetch<RB>(e, eq, y + plane * (int64_t)p.out_h * p.out_w, live && ox0 + tx < p.out_w, oy0, p.out_h, p.out_w,
                        ox0 + tx, plane, (int64_t)p.out_h * p.out_w);
#pragma unroll
    for (int r = 0; r < SR; ++r)
#pragma unroll
        for (int b = 0; b < CB; ++b) sp[r * SWP + tx + b * TW] = body[r][b];
#pragma unroll
    for (int i = 0; i < NX; ++i) {
        const int e = tx + i * TW;
        const int r = e / (XW > 0 ? XW : 1), c = CB * TW + e - r * XW;
        if (e < XW * SR) sp[r * SWP + c] = halo[i];
    }
    __syncthreads();

    const int mid_x = tile_mid_x + tx * DOWN;
    const int in_x = floor_div_i(mid_x, UP);
    const int rel_x = in_x - tile_in_x;
    const int kx0 = (in_x + 1) * UP - mid_x - 1;
    const int ox = ox0 + tx;
    float* yp = y + plane * (int64_t)p.out_h * p.out_w;
    [[maybe_unused]] float vo[RB];
    if constexpr (UP == 2 && DOWN == 1 && KH == 4 && KW == 4) {
        // Zero-insert x2 with compile-time tap phases.  Output row oy0 + o reads staged rows (o + q) >> 1 + yy through tap rows
        // 1 - ((o + q) & 1) + 2 yy, q = tile_mid_y & 1 (uniform per thread row); the same along x with the lane's own parity.
        // The lane's eight taps (its column phase, all four rows) are read from LDS once, the two row phases are told apart by ONE
        // select per tap on q, and every LDS address is a per-thread base + a compile-time offset: the generic loop below spends
        // ~100 vector instructions per output on run-time tap indices and predicates (870 for 54 FMAs in the epilogue form), and
        // these kernels are bound by what they issue.  Same taps, same order (tap rows outer, columns inner): bit-identical.
        const int q = tile_mid_y & 1;
        const float* tp = taps + kx0;
        float te[2][2], to[2][2];                    // taps of the even / odd output rows of the strip: [yy][xx]
#pragma unroll
        for (int yy = 0; yy < 2; ++yy)
#pragma unroll
            for (int xx = 0; xx < 2; ++xx) {
                const float t1 = tp[(1 + 2 * yy) * KW + 2 * xx], t0 = tp[(2 * yy) * KW + 2 * xx];   // tap rows 1 + 2 yy / 2 yy
                te[yy][xx] = q ? t0 : t1;            // even o: tap row 1 - (q & 1) + 2 yy
                to[yy][xx] = q ? t1 : t0;
            }
        const float* se = sp + rel_x;                // even o: staged row o / 2 + yy
        const float* so = sp + rel_x + q * SWP;      // odd o: staged row (o - 1) / 2 + q + yy
#pragma unroll
        for (int o = 0; o < RB; ++o) {
            const float* sr = (o & 1) ? so + (o >> 1) * SWP : se + (o >> 1) * SWP;
            float v = 0.0f;
#pragma unroll
            for (int yy = 0; yy < 2; ++yy)
#pragma unroll
                for (int xx = 0; xx < 2; ++xx) v = fmaf(sr[yy * SWP + xx], (o & 1) ? to[yy][xx] : te[yy][xx], v);
            if constexpr (EPI) vo[o] = v;
            else if (live && ox < p.out_w && oy0 + o < p.out_h) yp[(int64_t)(oy0 + o) * p.out_w + ox] = v;
        }
    } else {
#pragma unroll
    for (int o = 0; o < RB; ++o) {
        const int mid_y = tile_mid_y + o * DOWN;
        const int in_y = floor_div_i(mid_y, UP);
        const int rel_y = in_y - tile_in_y;
        const int ky0 = (in_y + 1) * UP - mid_y - 1;
        float v = 0.0f;
#pragma unroll
        for (int yy = 0; yy < (KH + UP - 1) / UP; ++yy)
#pragma unroll
            for (int xx = 0; xx < (KW + UP - 1) / UP; ++xx) {
                const int ky = ky0 + yy * UP, kx = kx0 + xx * UP;
                if (ky < KH && kx < KW) v = fmaf(sp[(rel_y + yy) * SWP + rel_x + xx], taps[ky * KW + kx], v);
            }
        const int oy = oy0 + o;
        if constexpr (EPI) vo[o] = v;
        else if (live && ox < p.out_w && oy < p.out_h) yp[(int64_t)oy * p.out_w + ox] = v;
    }
    }
    if constexpr (EPI) {
        const int gi = (int)(g - plane_u * (unsigned)p.groups_per_plane);
        k1_epilogue<TW, RB>(e, eq, vo, yp, live && ox < p.out_w, oy0, p.out_h, p.out_w, ox, plane, gi * p.x_tiles + xt,
                            p.groups_per_plane * p.x_tiles, live, tx);
    }
}

template <int UP, int DOWN, int TW, int RB, bool EPI = false>
void launch_updown(const float* x, const float* k, float* y, UpDownParams p, hipStream_t s, K1Epilogue e = K1Epilogue{}) {
    constexpr int NR = kBlock / TW;
    p.groups_per_plane = ceil_div(p.out_h, RB);
    p.x_tiles = ceil_div(p.out_w, TW);
    p.groups = p.planes * p.groups_per_plane;
    p.blocks = ceil_div64(p.groups, NR) * p.x_tiles;
    hipLaunchKernelGGL((updown_kernel<UP, DOWN, 4, 4, TW, RB, EPI>), dim3((unsigned)((p.blocks + 7) / 8 * 8)), dim3(kBlock), 0, s,
                       x, k, y, p, e);
}

template <int KH, int KW, int TW, int RB, bool EPI = false, bool ACC = true>
void launch_blur(const float* x, const float* k, float* y, BlurParams p, hipStream_t s, K1Epilogue e = K1Epilogue{}) {
    constexpr int NR = kBlock / TW;
    p.groups_per_plane = ceil_div(p.out_h, RB);
    p.x_tiles = ceil_div(p.out_w, TW);
    p.groups = p.planes * p.groups_per_plane;
    p.blocks = ceil_div64(p.groups, NR) * p.x_tiles;
    hipLaunchKernelGGL((blur_kernel<KH, KW, TW, RB, EPI, ACC>), dim3((unsigned)((p.blocks + 7) / 8 * 8)), dim3(kBlock), 0, s, x, k, y,
                       p, e);
}

template <int KH, int KW, int RB, bool EPI = false, bool ACC = true>
void launch_blur_tail(const float* x, const float* k, float* y, BlurParams p, hipStream_t s, K1Epilogue e = K1Epilogue{}) {
    p.groups_per_plane = ceil_div(p.out_h, RB);
    p.x_tiles = ceil_div(p.out_w, 64);
    p.groups = p.planes * p.groups_per_plane;
    p.main_per_unit = (p.x_tiles - 1) * 8;
    p.blocks = ceil_div64(p.groups, 32) * (p.main_per_unit + 1);
    hipLaunchKernelGGL((blur_tail_kernel<KH, KW, RB, EPI, ACC>), dim3((unsigned)((p.blocks + 7) / 8 * 8)), dim3(kBlock), 0, s, x, k, y,
                       p, e);
}

// the planes kernels index row groups in 32 bits (k1_block_id): planes x ceil(out_h / 4) row groups at most
inline bool k1_groups_fit(int64_t planes, int64_t out_h) { return planes * ((out_h + 3) / 4) < ((int64_t)1 << 31); }
// ... and elements inside a plane, and the byte offsets of a wave's buffer loads (relative to the first of the at most 8 planes a
// wave of 8-column thread rows touches), in 32 bits
inline bool k1_plane_fits(int64_t in_h, int64_t in_w, int64_t out_h, int64_t out_w) {
    return in_h * in_w <= ((int64_t)1 << 24) && out_h * out_w <= ((int64_t)1 << 28);
}

// 64-column tiles with a remainder of 1 ... 8 columns
inline bool blur_has_tail(int out_w) { return out_w > 64 && out_w % 64 >= 1 && out_w % 64 <= 8; }

// tile of the planes kernels: width from the output width, rows per thread from the output height (blur: 8 rows per thread
// on wide planes, 67 VGPRs -> 7 waves/SIMD, measured 3.5-4.0 TB/s vs 2.9-3.5 with 16 rows)
struct K1Tile { int tw, rb; };
inline K1Tile blur_tile(int out_w, int out_h) {
    if (out_w > 32) return {64, out_h >= 16 ? 8 : 4};
    if (out_w > 16) return {32, out_h >= 16 ? 16 : 4};
    if (out_w > 8) return {16, out_h >= 16 ? 16 : 4};
    return {8, 4};
}
inline K1Tile updown_tile(int out_w) { return out_w > 16 ? K1Tile{64, 8} : K1Tile{16, 4}; }

template <bool EPI>
void dispatch_blur44(const float* x, const float* k, float* y, const BlurParams& p, hipStream_t s, const K1Epilogue& e) {
    const K1Tile t = blur_tile(p.out_w, p.out_h);
    const bool tail = t.tw == 64 && t.rb == 8 && blur_has_tail(p.out_w) && tuning_knob("SAE_K1_TAIL", 1);
    if (EPI && !e.accumulate && t.tw == 64 && t.rb == 8) {       // the train step's blur + activation backward: no `old` operand
        if (tail) launch_blur_tail<4, 4, 8, EPI, false>(x, k, y, p, s, e);
        else launch_blur<4, 4, 64, 8, EPI, false>(x, k, y, p, s, e);
    } else if (tail) launch_blur_tail<4, 4, 8, EPI>(x, k, y, p, s, e);
    else if (t.tw == 64 && t.rb == 8) launch_blur<4, 4, 64, 8, EPI>(x, k, y, p, s, e);
    else if (t.tw == 64) launch_blur<4, 4, 64, 4, EPI>(x, k, y, p, s, e);
    else if (t.tw == 32 && t.rb == 16) launch_blur<4, 4, 32, 16, EPI>(x, k, y, p, s, e);
    else if (t.tw == 32) launch_blur<4, 4, 32, 4, EPI>(x, k, y, p, s, e);
    else if (t.tw == 16 && t.rb == 16) launch_blur<4, 4, 16, 16, EPI>(x, k, y, p, s, e);
    else if (t.tw == 16) launch_blur<4, 4, 16, 4, EPI>(x, k, y, p, s, e);
    else launch_blur<4, 4, 8, 4, EPI>(x, k, y, p, s, e);
}

template <int KH, int KW>
void dispatch_blur(const float* x, const float* k, float* y, const BlurParams& p, hipStream_t s) {
    const K1Tile t = blur_tile(p.out_w, p.out_h);
    if (KH > 1 && t.tw == 64 && t.rb == 8 && blur_has_tail(p.out_w) && tuning_knob("SAE_K1_TAIL", 1)) launch_blur_tail<KH, KW, 8>(x, k, y, p, s);
    else if (t.tw == 64 && t.rb == 8) launch_blur<KH, KW, 64, 8>(x, k, y, p, s);
    else if (t.tw == 64) launch_blur<KH, KW, 64, 4>(x, k, y, p, s);
    else if (t.tw == 32 && t.rb == 16) launch_blur<KH, KW, 32, 16>(x, k, y, p, s);
    else if (t.tw == 32) launch_blur<KH, KW, 32, 4>(x, k, y, p, s);
    else if (t.tw == 16 && t.rb == 16) launch_blur<KH, KW, 16, 16>(x, k, y, p, s);
    else if (t.tw == 16) launch_blur<KH, KW, 16, 4>(x, k, y, p, s);
    else launch_blur<KH, KW, 8, 4>(x, k, y, p, s);
}

// second stage of the fused bias gradient: gb[c] = sum_q partial[c][q], one workgroup per channel, fixed order.  q runs to a
// few thousand here (outer * strips per plane): one wave per channel walked them as a chain of q / 64 dependent loads, 50 us per
// call and a fifth of the fused blur's time in the step (profiles/r3_step_church256_b16_f32_kernel_trace.txt); thread t now
// sums q = t, t + 256, ... in four independent chains.
__global__ __launch_bounds__(kBlock) void k1_bias_finalize_kernel(const float* __restrict__ partial, float* __restrict__ gb,
                                                                  int64_t channels, int64_t q_count) {
    __shared__ float red[kBlock / kWave];
    const float* pc = partial + (int64_t)blockIdx.x * q_count;
    float a0 = 0.0f, a1 = 0.0f, a2 = 0.0f, a3 = 0.0f;
    int64_t q = threadIdx.x;
    for (; q + 3 * kBlock < q_count; q += 4 * kBlock) {
        a0 += pc[q];
        a1 += pc[q + kBlock];
        a2 += pc[q + 2 * kBlock];
        a3 += pc[q + 3 * kBlock];
    }
    for (; q < q_count; q += kBlock) a0 += pc[q];
    float acc = (a0 + a1) + (a2 + a3);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) t += red[w];
        gb[blockIdx.x] = t;
    }
}

// partial sums per channel of an epilogue launch: outer * row groups per plane * x tiles
inline int64_t k1_q_per_plane(int out_h, int out_w, bool x2) {
    const K1Tile t = x2 ? updown_tile(out_w) : blur_tile(out_w, out_h);
    return (int64_t)ceil_div(out_h, t.rb) * ceil_div(out_w, t.tw);
}

}  // namespace
}  // namespace sae

using namespace sae;

extern "C" int sae_upfirdn2d_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h,
                                 int64_t in_w, int64_t minor, int32_t kh, int32_t kw, int32_t up_x,
                                 int32_t up_y, int32_t down_x, int32_t down_y, int32_t pad_x0,
                                 int32_t pad_x1, int32_t pad_y0, int32_t pad_y1, sae_stream_t stream) {
    sae::clear_stale_error();
    if (up_x < 1 || up_y < 1 || down_x < 1 || down_y < 1 || kh < 1 || kw < 1 || (int64_t)kh * kw > 1024)
        return fail(SAE_EINVAL, "sae_upfirdn2d_f32: bad up/down/taps (%d,%d,%d,%d,%dx%d)", up_x, up_y, down_x,
                    down_y, kh, kw);
    if (major < 0 || minor < 1 || in_h < 1 || in_w < 1)
        return fail(SAE_EINVAL, "sae_upfirdn2d_f32: bad tensor size");
    if (in_h > (1 << 24) || in_w > (1 << 24)) return fail(SAE_EINVAL, "sae_upfirdn2d_f32: plane too large");
    const int64_t out_h = (in_h * up_y + pad_y0 + pad_y1 - kh + down_y) / down_y;
    const int64_t out_w = (in_w * up_x + pad_x0 + pad_x1 - kw + down_x) / down_x;
    if (out_h < 1 || out_w < 1)
        return fail(SAE_EINVAL, "sae_upfirdn2d_f32: empty output (%lld x %lld)", (long long)out_h, (long long)out_w);
    if (major == 0) return SAE_OK;
    if (!x || !k || !y) return fail(SAE_EINVAL, "sae_upfirdn2d_f32: null tensor");
    hipStream_t s = (hipStream_t)stream;

    // (planes beyond the 32-bit indexing of the planes kernels -- 4096 x 4096 inputs -- take the generic kernel)
    const bool planes_ok = minor == 1 && k1_groups_fit(major, out_h) && k1_plane_fits(in_h, in_w, out_h, out_w);
    const bool is_blur = planes_ok && up_x == 1 && up_y == 1 && down_x == 1 && down_y == 1 && kh <= 4 && kw <= 4;
    if (is_blur) {
        BlurParams p{};
        p.planes = major;
        p.in_h = (int)in_h; p.in_w = (int)in_w; p.out_h = (int)out_h; p.out_w = (int)out_w;
        p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
        if (blur_streams(p, false)) launch_blur_stream<false>(x, k, y, p, s, K1Epilogue{});
        else if (kh == 1 && kw == 1) dispatch_blur<1, 1>(x, k, y, p, s);
        else if (kh <= 3 && kw <= 3) dispatch_blur<3, 3>(x, k, y, p, s);
        else dispatch_blur<4, 4>(x, k, y, p, s);
        return check_launch("sae_upfirdn2d_f32(blur)");
    }
    const bool is_x2 = planes_ok && kh <= 4 && kw <= 4 && up_x == up_y && down_x == down_y &&
                       ((up_x == 1 && down_x == 2) || (up_x == 2 && down_x == 1));
    if (is_x2) {
        UpDownParams p{};
        p.planes = major;
        p.in_h = (int)in_h; p.in_w = (int)in_w; p.out_h = (int)out_h; p.out_w = (int)out_w;
        p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
        const bool wide = updown_tile((int)out_w).tw == 64;
        if (down_x == 2) {
            if (wide) launch_updown<1, 2, 64, 8>(x, k, y, p, s); else launch_updown<1, 2, 16, 4>(x, k, y, p, s);
        } else {
            if (wide) launch_updown<2, 1, 64, 8>(x, k, y, p, s); else launch_updown<2, 1, 16, 4>(x, k, y, p, s);
        }
        return check_launch("sae_upfirdn2d_f32(x2)");
    }
    GenericParams g{};
    g.major = major; g.minor = minor;
    g.in_h = (int)in_h; g.in_w = (int)in_w; g.out_h = (int)out_h; g.out_w = (int)out_w;
    g.kh = kh; g.kw = kw; g.up_x = up_x; g.up_y = up_y; g.down_x = down_x; g.down_y = down_y;
    g.pad_x0 = pad_x0; g.pad_y0 = pad_y0;
    g.total = major * out_h * out_w * minor;
    int64_t blocks = ceil_div64(g.total, kBlock);
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(upfirdn2d_generic_kernel, dim3((unsigned)blocks), dim3(kBlock), 0, s, x, k, y, g);
    return check_launch("sae_upfirdn2d_f32(generic)");
}

extern "C" int64_t sae_upfirdn2d_epilogue_workspace(int64_t major, int64_t out_h, int64_t out_w, int64_t channels,
                                                    int32_t up) {
    if (major < 1 || out_h < 1 || out_w < 1 || channels < 1 || major % channels != 0 || out_h > (1 << 24) || out_w > (1 << 24))
        return 0;
    return major * k1_q_per_plane((int)out_h, (int)out_w, up == 2);
}

extern "C" int sae_upfirdn2d_epilogue_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                                          int32_t kh, int32_t kw, int32_t up, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0,
                                          int32_t pad_y1, const float* act_ref, float slope, float scale, float* gb,
                                          int64_t channels, int32_t accumulate, float* workspace, int64_t workspace_floats,
                                          sae_stream_t stream) {
    sae::clear_stale_error();
    if ((up != 1 && up != 2) || kh < 1 || kw < 1 || kh > 4 || kw > 4)
        return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: up must be 1 or 2 and the taps at most 4 x 4 (up=%d, %dx%d)", up, kh, kw);
    if (major < 0 || in_h < 1 || in_w < 1 || in_h > (1 << 24) || in_w > (1 << 24))
        return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: bad tensor size");
    const int64_t out_h = in_h * up + pad_y0 + pad_y1 - kh + 1;
    const int64_t out_w = in_w * up + pad_x0 + pad_x1 - kw + 1;
    if (out_h < 1 || out_w < 1)
        return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: empty output (%lld x %lld)", (long long)out_h, (long long)out_w);
    hipStream_t s = (hipStream_t)stream;
    if (act_ref) {
        if (!gb || channels < 1 || major % channels != 0)
            return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: act_ref needs gb and major = outer * channels");
        if (major == 0) {
            hipMemsetAsync(gb, 0, sizeof(float) * (size_t)channels, s);
            return check_launch("sae_upfirdn2d_epilogue_f32(memset)");
        }
    }
    if (major == 0) return SAE_OK;
    if (!x || !k || !y) return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: null tensor");
    if (!k1_groups_fit(major, out_h) || !k1_plane_fits(in_h, in_w, out_h, out_w))
        return fail(SAE_EINVAL, "sae_upfirdn2d_epilogue_f32: more than 2^31 row groups, or a plane beyond 4096 x 4096");
    K1Epilogue e{};
    e.act_ref = act_ref; e.slope = slope; e.scale = scale; e.accumulate = accumulate ? 1 : 0;
    e.channels = act_ref ? (int)channels : 1;
    const int64_t qpp = k1_q_per_plane((int)out_h, (int)out_w, up == 2);
    if (act_ref) {
        const int64_t need = major * qpp;
        if (!workspace || workspace_floats < need)
            return fail(SAE_EWORKSPACE, "sae_upfirdn2d_epilogue_f32: workspace %lld < %lld floats", (long long)workspace_floats,
                        (long long)need);
        e.partial = workspace;
        e.q_per_channel = (major / channels) * qpp;
    }
    if (up == 1) {
        BlurParams p{};
        p.planes = major;
        p.in_h = (int)in_h; p.in_w = (int)in_w; p.out_h = (int)out_h; p.out_w = (int)out_w;
        p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
        dispatch_blur44<true>(x, k, y, p, s, e);
    } else {
        UpDownParams p{};
        p.planes = major;
        p.in_h = (int)in_h; p.in_w = (int)in_w; p.out_h = (int)out_h; p.out_w = (int)out_w;
        p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
        if (updown_tile((int)out_w).tw == 64) launch_updown<2, 1, 64, 8, true>(x, k, y, p, s, e);
        else launch_updown<2, 1, 16, 4, true>(x, k, y, p, s, e);
    }
    if (act_ref)
        hipLaunchKernelGGL(k1_bias_finalize_kernel, dim3((unsigned)channels), dim3(kBlock), 0, s, (const float*)workspace, gb,
                           channels, e.q_per_channel);
    return check_launch("sae_upfirdn2d_epilogue_f32");
}

extern "C" int sae_upfirdn2d_noise_bias_act_f32(const float* x, const float* k, float* y, int64_t major, int64_t in_h, int64_t in_w,
                                                int32_t kh, int32_t kw, int32_t pad_x0, int32_t pad_x1, int32_t pad_y0,
                                                int32_t pad_y1, const float* noise, const float* noise_weight, const float* bias,
                                                int64_t channels, float slope, float scale, sae_stream_t stream) {
    sae::clear_stale_error();
    const char* who = "sae_upfirdn2d_noise_bias_act_f32";
    if (kh < 1 || kw < 1 || kh > 4 || kw > 4) return fail(SAE_EINVAL, "%s: at most 4 x 4 taps (%dx%d)", who, kh, kw);
    if (major < 0 || in_h < 1 || in_w < 1 || in_h > (1 << 24) || in_w > (1 << 24) || channels < 1 || major % channels != 0)
        return fail(SAE_EINVAL, "%s: bad tensor size (major = outer * channels)", who);
    const int64_t out_h = in_h + pad_y0 + pad_y1 - kh + 1;
    const int64_t out_w = in_w + pad_x0 + pad_x1 - kw + 1;
    if (out_h < 1 || out_w < 1) return fail(SAE_EINVAL, "%s: empty output (%lld x %lld)", who, (long long)out_h, (long long)out_w);
    if (major == 0) return SAE_OK;
    if (!x || !k || !y || (noise && !noise_weight)) return fail(SAE_EINVAL, "%s: null tensor", who);
    if (!k1_groups_fit(major, out_h) || !k1_plane_fits(in_h, in_w, out_h, out_w))
        return fail(SAE_EINVAL, "%s: more than 2^31 row groups, or a plane beyond 4096 x 4096", who);
    K1Epilogue e{};
    e.slope = slope; e.scale = scale; e.channels = (int)channels;
    e.fwd_act = 1; e.fwd_noise = noise; e.fwd_noise_w = noise_weight; e.fwd_bias = bias;
    BlurParams p{};
    p.planes = major;
    p.in_h = (int)in_h; p.in_w = (int)in_w; p.out_h = (int)out_h; p.out_w = (int)out_w;
    p.pad_x0 = pad_x0; p.pad_y0 = pad_y0; p.kh = kh; p.kw = kw;
    if (blur_streams(p, true)) launch_blur_stream<true>(x, k, y, p, (hipStream_t)stream, e);
    else dispatch_blur44<true>(x, k, y, p, (hipStream_t)stream, e);
    return check_launch(who);
}
