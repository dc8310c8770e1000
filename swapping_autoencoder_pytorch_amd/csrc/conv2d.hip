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

// ------------------------------------------------------------------------------------------
// weight re-layout
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(kBlock) void conv_wprep_kernel(const float* __restrict__ w,
                                                            float* __restrict__ wp, int M, int C, int Mp,
                                                            int Cp, int taps, int64_t sm, int64_t sc,
                                                            int flip, float alpha,
                                                            const float* __restrict__ rs_m,
                                                            const float* __restrict__ rs_c) {
    // rs_m[m] / rs_c[c] (either may be null): per-channel factors of the staged weight along the launch's output
    // (m) or contraction (c) axis: the demodulation of ModulatedConv2d (stylegan2_layers.py:290-292) rides here
    const int64_t total = (int64_t)taps * Cp * Mp;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        const int m = (int)(i % Mp);
        const int64_t t = i / Mp;
        const int c = (int)(t % Cp);
        const int tap = (int)(t / Cp);
        float v = 0.0f;
        if (m < M && c < C) {
            v = alpha * w[m * sm + c * sc + (flip ? taps - 1 - tap : tap)];
            if (rs_m) v *= rs_m[m];
            if (rs_c) v *= rs_c[c];
        }
        wp[i] = v;
    }
}

// ------------------------------------------------------------------------------------------
// forward-type implicit GEMM (stride 1 or 2 gather)
// ------------------------------------------------------------------------------------------
// Profiling hook (tools/build_variant.sh ... -DSAE_CLOCK_PROBE, never in the product build): every workgroup of the
// three fp32 MFMA kernels adds its duration in shader cycles (s_memtime) and in constant 100 MHz ticks (s_memrealtime)
// to a device counter, from which tools/ab_conv.py --clock derives the shader clock the kernel actually ran at.
#ifdef SAE_CLOCK_PROBE
__device__ unsigned long long g_clock_probe[16];
// [0] shader cycles, [1] 100 MHz ticks, [2] workgroups; phases of conv_igemm_kernel as seen by wave 0, in shader cycles:
// [3] prologue, [4] MFMA phase, [5] wait at the barrier after it, [6] LDS stores, [7] wait at the second barrier,
// [8] global-load issue, [9] epilogue
#define SAE_CLOCK_BEGIN const unsigned long long cp_c0 = clock64(), cp_r0 = wall_clock64(); \
    unsigned long long cp_t = cp_c0, cp_ph[7] = {0, 0, 0, 0, 0, 0, 0};
#define SAE_CLOCK_PHASE(i) { const unsigned long long cp_n = clock64(); cp_ph[i] += cp_n - cp_t; cp_t = cp_n; }
#define SAE_CLOCK_END                                                   \
    if (threadIdx.x == 0) {                                             \
        atomicAdd(&g_clock_probe[0], clock64() - cp_c0);                \
        atomicAdd(&g_clock_probe[1], wall_clock64() - cp_r0);           \
        atomicAdd(&g_clock_probe[2], 1ull);                             \
        for (int cp_i = 0; cp_i < 7; ++cp_i) atomicAdd(&g_clock_probe[3 + cp_i], cp_ph[cp_i]); \
    }
#else
#define SAE_CLOCK_BEGIN
#define SAE_CLOCK_PHASE(i)
#define SAE_CLOCK_END
#endif

// LDS operand reads of the gather kernels run this many (tap, channel pair) steps ahead of the MFMAs that use them
#ifndef SAE_IGEMM_AHEAD
#define SAE_IGEMM_AHEAD 1
#endif
constexpr int kIgemmAhead = SAE_IGEMM_AHEAD;
// Workgroups per CU the 1x1 instantiations of the gather are compiled for (register budget 512 / this).  Their K loops are 4 - 16
// chunks long and the layers are close to HBM-bound (fwd + residual 128 -> 256 @128^2 moves 1.7 GB for 43 GFLOP): a third
// workgroup per CU hides more of each one's load -> store round trip than the dozen spilled registers cost -- same-box A/B
// (tools/ab_conv.py, profiles/r4_ab_1x1_occupancy.txt): fwd 84 -> 92, 90 -> 93, 102 -> 111, 91 -> 101 TFLOP/s, dgrad 96 -> 104,
// 102 -> 107, 107 -> 116, 92 -> 101; four (128 registers) spills the accumulators: 58 - 83.
#ifndef SAE_IGEMM_1X1_WAVES
#define SAE_IGEMM_1X1_WAVES 3
#endif

struct IgemmParams {
    int N, C, H, W;       // input tensor; C = contraction channels
    int M, OH, OW;        // logical output grid, M = output channels
    int YH, YW;           // allocated output plane
    int oys, oxs;         // output scatter multipliers (2 for the 1x1 stride-2 dgrad)
    int Cp, Mp;           // padded dims of wp
    int pad;              // iy = oy*S + ky - pad
    int tw_log2, th_log2; // pixel tile = TN images x TH rows x TW cols
    int tiles_x, tiles_y, tiles_n;
    int chunks_per_split;   // split-K over input channels (blockIdx.z) for launches with few tiles
    int64_t slab_stride;    // floats between the partial outputs of consecutive K slices
    // fused epilogue (forward only): y = lrelu(acc + bias[m]) * act_scale when act != 0
    const float* bias;
    float act_slope, act_scale;
    int act;
    // fused residual merge (forward only): y = (acc + residual[same index as y]) * res_scale when residual != null -- the
    // (out + skip) / sqrt(2) of a ResBlock (stylegan2_layers.py:689) done by the skip path's 1x1 conv on its way out
    const float* residual;
    float res_scale;
    // fused NoiseInjection in front of the activation (MOD instantiations, forward only, with act != 0):
    // y = lrelu((acc + noise_w[0] * noise[n][oy][ox]) + bias[m]) * act_scale -- StyledConv's conv -> noise -> FusedLeakyReLU
    // (stylegan2_layers.py:398-405) without the pass over the conv's output.  noise: [N][OH][OW], noise_w: one float on the device
    const float* noise;
    const float* noise_w;
    // style modulation of the INPUT (ModulatedConv2d, stylegan2_layers.py:280-286): when non-null, x[n][c][..] is
    // multiplied by in_scale[n * C + c] on its way into LDS, so the modulated activation never exists in HBM
    const float* in_scale;
    // always 0.  The per-channel factor of a one-image tile is wave-uniform; indexing it with (lane & zmask) keeps the
    // compiler from turning it into scalar-cache loads, whose out-of-order return shares the LDS wait counter
    // (lgkmcnt) and would serialise the ds_read pipeline of the MFMA loop
    int zmask;
    // batched 1x1 launches (the sixteen transform-domain products of the Winograd route, csrc/winograd.hip): batch > 0 makes
    // blockIdx.z a batch index instead of a K slice -- input, weights and output (slab_stride) advance by these many floats
    int batch;
    int64_t batch_x, batch_w;
    int xcd_order;          // 1: XCD-aware tile order (see conv_igemm_kernel)
    int vec_store;          // 1: LDS-transposed dwordx4 epilogue (see conv_igemm_kernel)
};

template <int KS, int S, int BN>
struct PatchCap {
    // worst case over TW,TH >= 4 (smallest tiles have the largest halo share)
    static constexpr int value = (KS == 1) ? BN : (S == 1 ? (9 * BN) / 4 : (41 * BN) / 8);
};

// MOD: the input is style-modulated while staged (IgemmParams::in_scale); a separate instantiation, so the
// un-modulated kernels of E / D / Dpatch carry no trace of it
// QUAD (stride 1, rows a multiple of four floats wide; 3x3: tiles at least 16 wide, 1x1: pad 0): the input patch is staged
// as 16-byte quads aligned in memory -- 3x3: the rows of the patch widened to [x0 - 4, x0 + TW + 4) -- CK channels' worth of quads
// dealt out over the workgroup, so a chunk takes BN/64 dwordx4 loads and as many ds_write_b128 per thread instead of
// 2.25 BN/32 dword ones.  A wave64 vector-memory instruction occupies the CU's address path for ~25-30 cycles whatever
// its width (tools/probe/mfma_clock_probe.hip, profiles/r2_phase_clock_*.txt): the staging phase of a chunk is bound by the
// NUMBER of such instructions, and the eight waves of a CU issue theirs at the same time.
template <int KS, int S, int MI, int NI, int WM, int WN, int CK, bool MOD = false, bool QUAD = false>
// (measured and dropped: __launch_bounds__(kBlock, 2) for the 64-accumulator tiles -- it brings the modulated stride-2
// instantiations, 8 registers over budget, back to two waves per SIMD with 11-16 spilled registers; their time did not move
// (5.16 ms per iteration either way) and the step got 0.7 % slower, the other instantiations' allocation changes with it)
__global__ __launch_bounds__(kBlock, (KS == 1 ? (MOD ? 2 : SAE_IGEMM_1X1_WAVES) : 1)) void conv_igemm_kernel(const float* __restrict__ x,
                                                            const float* __restrict__ wp,
                                                            float* __restrict__ y, const IgemmParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(!QUAD || S == 1 || KS == 3, "quad staging: stride 1, or 3x3 stride 2");
    // the fused residual merge (IgemmParams::residual) is compiled into the 1x1 kernels only -- the skip convs are its one
    // use, and one register more sent the stride-2 3x3 gather from two waves per SIMD to one (27.0 -> 31.2 ms per iteration)
    constexpr bool RESIDUAL = KS == 1;
    constexpr int XQ0 = (KS == 3) ? 4 : 0;      // QUAD: columns added on either side of the tile
    constexpr int T = KS * KS;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    constexpr int XCAP = PatchCap<KS, S, BN>::value;
    constexpr int PPT = QUAD ? 1 : (XCAP + kBlock - 1) / kBlock;      // patch slots per thread per channel
    constexpr int QCAP = (S == 2) ? 5 * BN / 4 : (KS == 3) ? BN / 2 : BN / 4;   // QUAD: quads per channel accepted by the host
    constexpr int QPT = QUAD ? (CK * QCAP + kBlock - 1) / kBlock : 1; // ... and quad slots per thread per chunk
    constexpr int A_VEC = T * CK * BM / 4;                  // float4 per A chunk
    constexpr int APT = (A_VEC + kBlock - 1) / kBlock;
    // a channel row of Xs is XCAP patch floats + a 64-float dump area: threads whose slot lies beyond the patch store
    // there, so no LDS store of the K loop is predicated (XROW = XCAP mod 64 keeps the bank pattern of the reads)
    constexpr int XROW = XCAP + 64;
    // one allocation: the LDS-transposed epilogue may use the whole of it (the weight tile alone is too small for the 1x1
    // kernels' 32-channel chunks)
    constexpr int AS = T * CK * BM, XS = CK * XROW;
    __shared__ __attribute__((aligned(16))) float smem[AS + XS];
    float* const As = smem;
    float* const Xs = smem + AS;
    // the tile's slice of the noise map (IgemmParams::noise): BN floats, shared by all BM rows of the tile, fetched once at the
    // start (into registers, see below) and read from LDS in the epilogue -- a global load inside the store loop would make every store wait for its own
    // round trip, and 8 - 16 quads per thread cannot be prefetched into registers (the fused residual's lesson)
    __shared__ float zs[MOD ? BN : 1];

    SAE_CLOCK_BEGIN
    const int tid = threadIdx.x;
    __builtin_assume(tid < kBlock);   // hipcc otherwise assumes up to 1024 and predicates the tail of the staging loops
    const int lane = tid & 63, wid = tid >> 6;
    const int wid_u = __builtin_amdgcn_readfirstlane(wid);   // the same value, known to be wave-uniform
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = BN >> (p.tw_log2 + p.th_log2);
    // XCD-aware tile order: consecutive workgroup ids land on consecutive XCDs (id % 8), each with its own L2.  Giving
    // XCD k the k-th CONTIGUOUS eighth of the tile list puts a tile and its spatial neighbours (which share halo rows
    // and the 128-byte lines the halo columns straddle) behind the same L2, in flight at about the same time; in plain
    // order every XCD fetched its own copy: 1.56 GB from the fabric for 0.54 GB of input (profiles/r2_pmc_f32.txt)
    // With several M tiles (gridDim.y > 1) the M tile is the FASTEST index of the re-labelled order: the workgroups that
    // read the same input patch run together behind one L2 (in launch order they are gridDim.x workgroups apart and each
    // fetched its own copy from HBM: 1138 MB read for 537 MB of input with two M tiles, profiles/r3_pmc_f32.txt).
    int bt = blockIdx.x;
    int mt = blockIdx.y;
    if (p.xcd_order) {
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.y;           // dispatch order: x fastest
            const int wk = (lin & 7) * (total >> 3) + (lin >> 3);
            mt = wk % gridDim.y;
            bt = wk / gridDim.y;
        }
    }
    const int tix = bt % p.tiles_x; bt /= p.tiles_x;
    const int tiy = bt % p.tiles_y;
    const int tin = bt / p.tiles_y;
    const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
    const int m0 = mt * BM;
    // issued here, parked in registers across the K loop (one per thread at BN = 256) and written to LDS just before the
    // epilogue: written to LDS right away, the wait for this load stood in front of the first chunk's loads of every workgroup
    constexpr int ZPT = MOD ? (BN + kBlock - 1) / kBlock : 1;
    [[maybe_unused]] float zreg[ZPT];
    if constexpr (MOD) {
        if (p.noise) {
#pragma unroll
            for (int k = 0; k < ZPT; ++k) {
                const int pp = tid + kBlock * k;
                const int px = pp & (TW - 1);
                const int py = (pp >> p.tw_log2) & (TH - 1);
                const int pn = pp >> (p.tw_log2 + p.th_log2);
                const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
                const bool ok = pp < BN && n < p.N && oy < p.OH && ox < p.OW;
                const float v = p.noise[ok ? ((int64_t)n * p.OH + oy) * p.OW + ox : 0];
                zreg[k] = ok ? v : 0.0f;
            }
        }
    }

    const int PH = (KS == 1) ? TH : (TH - 1) * S + KS;
    const int PW = (KS == 1) ? TW : (TW - 1) * S + KS;
    // QUAD, stride 2 (pad 0): rows widened to whole quads starting at column 2 x0 (4-byte aligned quads, see the wgrad
    // kernel), kept de-interleaved in LDS like the dword path: even columns, then odd columns
    const int RS = QUAD ? (S == 2 ? ((PW + 3) >> 2) << 2 : TW + 2 * XQ0) : PW;
    const int HALFW = (QUAD && S == 2) ? RS >> 1 : (PW + 1) >> 1;
    const int IP = PH * RS;
    const int CP = TN * IP;           // staged floats per channel (<= XCAP, checked on the host)
    const int HW = p.H * p.W;

    // per-thread patch slots.  pbyte: BYTE offset of the slot's input element relative to (image n0, channel 0);
    // padding / out-of-tile slots read element 0 instead (always valid memory) and are zeroed on their way into LDS
    // (pok), so that no global load of the K loop sits in a divergent branch and every load is the
    // "uniform 64-bit base + 32-bit lane offset" form with loop-invariant lane offsets.  xdst: LDS index within a
    // channel row (the dump area for slots beyond the patch).
    unsigned pbyte[PPT];
    bool pok[PPT];
    int xdst[PPT];
    bool wave_has[PPT];     // wave-uniform: some lane of this wave has a patch element in slot s (else the slot is skipped)
    int sidx[MOD ? (QUAD ? QPT : PPT) : 1];   // in_scale row of the slot's image
    // QUAD: slot k of a thread is quad j = tid + kBlock * k of the chunk: channel j / QN, quad j % QN of the patch
    // (image, row, quad column); qbyte is relative to (image n0, channel c0) and includes the channel
    unsigned qbyte[QPT];
    bool qok[QPT];
    int qdst[QPT], qch[QPT];
    [[maybe_unused]] int qdst2[(QUAD && S == 2) ? QPT : 1];   // stride 2: LDS index of the quad's odd columns
    [[maybe_unused]] int qsh[(QUAD && S == 2) ? QPT : 1];     // ... and how far the quad was moved left to stay inside its row
    if constexpr (QUAD) {
        const int RQ = RS >> 2;                 // quads per patch row
        const int QI = PH * RQ, QN = TN * QI;   // quads per image, per channel (<= QCAP, checked on the host)
#pragma unroll
        for (int kq = 0; kq < QPT; ++kq) {
            const int j = tid + kBlock * kq;
            const int ch = j / QN;
            const int q = j - ch * QN;
            const int pn = q / QI;
            const int rem = q - pn * QI;
            const int r = rem / RQ;
            const int qc = rem - r * RQ;
            const int iy = oy0 * S - p.pad + r, ix = (S == 2) ? ox0 * 2 + 4 * qc : ox0 - XQ0 + 4 * qc - (KS == 1 ? p.pad : 0);
            const bool slot = ch < CK;
            const bool in = slot && n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            qok[kq] = in;
            qch[kq] = slot ? ch : CK - 1;
            if constexpr (S == 2) {
                const int ixc = ix + 4 <= p.W ? ix : p.W - 4;      // a quad that would run past its row is fetched from W - 4
                qsh[kq] = in ? ix - ixc : 0;
                qbyte[kq] = in ? 4u * (unsigned)((pn * p.C + ch) * HW + iy * p.W + ixc) : 0u;
                const int d0 = slot ? ch * XROW + pn * IP + r * RS + 2 * qc
                                    : ((tid >> 3) & (CK - 1)) * XROW + XCAP + 2 * (tid & 7);
                qdst[kq] = d0;
                qdst2[kq] = slot ? d0 + HALFW : d0 + 32;
            } else {
                qbyte[kq] = in ? 4u * (unsigned)((pn * p.C + ch) * HW + iy * p.W + ix) : 0u;
                qdst[kq] = slot ? ch * XROW + 4 * q : ((tid >> 4) & (CK - 1)) * XROW + XCAP + 4 * (tid & 15);
            }
            if constexpr (MOD) sidx[kq] = in ? (n0 + pn) * p.C + ch : 0;
        }
        pbyte[0] = 0; pok[0] = false; xdst[0] = 0; wave_has[0] = false;
    } else {
        qbyte[0] = 0; qok[0] = false; qdst[0] = 0; qch[0] = 0;
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int e = tid + kBlock * s;
        int off = -1;
        if constexpr (MOD) sidx[s] = 0;
        if (e < CP) {
            const int pn = e / IP;
            const int rem = e - pn * IP;
            const int r = rem / RS;
            const int cl = rem - r * RS;
            int c = cl;
            if (KS != 1 && S == 2) c = (cl < HALFW) ? 2 * cl : 2 * (cl - HALFW) + 1;
            int iy, ix;
            if (KS == 1) { iy = (oy0 + r) * S - p.pad; ix = (ox0 + c) * S - p.pad; }
            else { iy = oy0 * S - p.pad + r; ix = ox0 * S - p.pad + c; }
            if (n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
                off = pn * p.C * HW + iy * p.W + ix;
                if constexpr (MOD) sidx[s] = (n0 + pn) * p.C;
            }
        }
        pok[s] = off >= 0;
        pbyte[s] = off >= 0 ? 4u * (unsigned)off : 0u;
        xdst[s] = e < CP ? e : XCAP + (tid & 63);
        // (not for the stride-2 gather: measured 116 vs 121 TFLOP/s with the skip, its third slot is the only sparse one)
        wave_has[s] = (KS == 3 && S == 2) || wid_u * kWave + kBlock * s < CP;
    }
    }
    // weight staging: float4 e4 = tid + kBlock * i of the chunk's [tap][ch][m] block; its byte offset relative to
    // (channel c0, column m0) of wp is loop-invariant
    unsigned abyte[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int e4 = tid + kBlock * i;
        const int row = e4 / (BM / 4);       // tap*CK + ch
        const int col4 = e4 - row * (BM / 4);
        const int tap = (row / CK) < T ? row / CK : 0, ch = row - (row / CK) * CK;
        abyte[i] = 4u * (unsigned)((tap * p.Cp + ch) * p.Mp + col4 * 4);
    }
    const bool one_image = TN == 1;          // the whole tile lies in image n0: one (uniform) factor per channel
    float sc[MOD ? (QUAD ? QPT : CK) : 1];   // ... prefetched with the chunk (QUAD: the factor of each quad slot)

    // per-lane LDS base of each N-tile pixel, and per-tap offsets
    int pixbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        pixbase[ni] = pn * IP + ((KS == 1) ? py * RS : py * S * RS) + px + ((QUAD && KS == 3 && S == 1) ? 4 - p.pad : 0);
    }
    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ky = t / KS, kx = t % KS;
        tapoff[t] = (KS == 1) ? 0 : ky * RS + ((S == 2) ? (kx & 1) * HALFW + (kx >> 1) : kx);
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    // (KS == 1 only: blockIdx.z as the batch index of a batched launch, see IgemmParams::batch)
    int zb = 0;
    if constexpr (KS == 1) { if (p.batch) zb = blockIdx.z; }
    const float* xb = x + (int64_t)n0 * p.C * HW + zb * p.batch_x;
    const float* const wpz = wp + zb * p.batch_w;
    float xv[QUAD ? 1 : CK][PPT] = {};
    f32x4 xq[QPT];
    f32x4 av[APT];

    auto load_chunk = [&](int c0) {
        if constexpr (QUAD) {
            const char* xc = reinterpret_cast<const char*>(xb + (int64_t)c0 * HW);
#pragma unroll
            for (int kq = 0; kq < QPT; ++kq) {
                const bool ok = qok[kq] && c0 + qch[kq] < p.C;       // channels beyond C (last chunk): element 0, zeroed later
                if constexpr (S == 2) {
                    struct __attribute__((packed, aligned(4))) U4 { f32x4 v; };     // 4-byte aligned quad
                    xq[kq] = reinterpret_cast<const U4*>(xc + (ok ? qbyte[kq] : 0u))->v;
                } else {
                    xq[kq] = *reinterpret_cast<const f32x4*>(xc + (ok ? qbyte[kq] : 0u));
                }
                if constexpr (MOD) sc[kq] = p.in_scale[ok ? sidx[kq] + c0 : 0];
            }
        } else {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (!wave_has[s]) continue;
#pragma unroll
            for (int ch = 0; ch < CK; ++ch) {
                // channels beyond C (last chunk) re-read channel C - 1 and are zeroed in store_chunk
                const int cc = (c0 + ch) < p.C ? c0 + ch : p.C - 1;
                const char* xc = reinterpret_cast<const char*>(xb + (int64_t)cc * HW);
                xv[ch][s] = *reinterpret_cast<const float*>(xc + pbyte[s]);
            }
        }
        if constexpr (MOD) {
            if (one_image) {
                const float* srow = p.in_scale + n0 * p.C + (lane & p.zmask);
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) sc[ch] = srow[(c0 + ch) < p.C ? c0 + ch : 0];
            }
        }
        }
        const char* wb = reinterpret_cast<const char*>(wpz + (int64_t)c0 * p.Mp + m0);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + kBlock * i;
            if (e4 < A_VEC) av[i] = *reinterpret_cast<const f32x4*>(wb + abyte[i]);
        }
    };
    auto store_chunk = [&](int c0) {
        if constexpr (QUAD) {
#pragma unroll
            for (int kq = 0; kq < QPT; ++kq) {
                const bool ok = qok[kq] && c0 + qch[kq] < p.C;
                f32x4 v = xq[kq];
                if constexpr (MOD) v *= sc[kq];
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
                if constexpr (S == 2) {
                    typedef float f32x2 __attribute__((ext_vector_type(2)));
                    const int sh = qsh[kq];      // fetched sh floats to the left of its place: the tail lies beyond the row
                    const f32x4 u = v;
                    if (sh == 1) v = f32x4{u[1], u[2], u[3], 0.0f};
                    if (sh == 2) v = f32x4{u[2], u[3], 0.0f, 0.0f};
                    if (sh == 3) v = f32x4{u[3], 0.0f, 0.0f, 0.0f};
                    *reinterpret_cast<f32x2*>(Xs + qdst[kq]) = f32x2{v[0], v[2]};      // even columns
                    *reinterpret_cast<f32x2*>(Xs + qdst2[kq]) = f32x2{v[1], v[3]};     // odd columns
                } else {
                    *reinterpret_cast<f32x4*>(Xs + qdst[kq]) = v;
                }
            }
        } else {
        if constexpr (MOD) {
            if (one_image) {
#pragma unroll
                for (int ch = 0; ch < CK; ++ch)
#pragma unroll
                    for (int s = 0; s < PPT; ++s) xv[ch][s] *= sc[ch];
            } else {     // small images, several per tile: the factor depends on the slot's image
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) {
                    const int cc = (c0 + ch) < p.C ? c0 + ch : 0;
#pragma unroll
                    for (int s = 0; s < PPT; ++s) xv[ch][s] *= p.in_scale[sidx[s] + cc];
                }
            }
        }
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            if (!wave_has[s]) continue;
#pragma unroll
            for (int ch = 0; ch < CK; ++ch) {
                const bool ch_ok = (c0 + ch) < p.C;
                Xs[ch * XROW + xdst[s]] = (ch_ok && pok[s]) ? xv[ch][s] : 0.0f;
            }
        }
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + kBlock * i;
            if (e4 < A_VEC) *reinterpret_cast<f32x4*>(As + e4 * 4) = av[i];
        }
    };

    const int c_begin = ((KS == 1 && p.batch) ? 0 : (int)blockIdx.z) * p.chunks_per_split * CK;
    int c_end = c_begin + p.chunks_per_split * CK;
    if (c_end > p.Cp) c_end = p.Cp;
    load_chunk(c_begin);
    SAE_CLOCK_PHASE(0)
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        __syncthreads();   // everyone finished reading the previous chunk
        SAE_CLOCK_PHASE(2)
        store_chunk(c0);
        SAE_CLOCK_PHASE(3)
        __syncthreads();
        SAE_CLOCK_PHASE(4)
        // in flight under the MFMAs below (issuing them one per MFMA step instead measured the same or slower)
        if (c0 + CK < c_end) load_chunk(c0 + CK);
        SAE_CLOCK_PHASE(5)
        // One step = one (tap, channel pair): MI x NI MFMAs on operands that were read from LDS kIgemmAhead steps
        // earlier (a ring of register sets), so a wave's ds_reads are in flight under its own MFMAs instead of being
        // waited for in front of each group.  The scheduling barriers keep the compiler from sinking the reads back
        // to their first use; the accumulation order (tap-major, then channel pair) is unchanged.
        constexpr int KK = CK / 2;
        constexpr int NSTEP = T * KK;
        constexpr int AH = kIgemmAhead, RING = AH + 1;
        float ra[RING][MI], rb[RING][NI];
        auto fetch = [&](int s, float (&a)[MI], float (&b)[NI]) {
            const int t = s / KK, ch = 2 * (s % KK) + half;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = As[(t * CK + ch) * BM + (wm * MI + mi) * 32 + l31];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = Xs[ch * XROW + pixbase[ni] + tapoff[t]];
        };
#pragma unroll
        for (int s = 0; s < AH; ++s) fetch(s, ra[s % RING], rb[s % RING]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + AH < NSTEP) fetch(s + AH, ra[(s + AH) % RING], rb[(s + AH) % RING]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % RING][mi], rb[s % RING][ni],
                                                                       acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        SAE_CLOCK_PHASE(1)
    }

    // Vector epilogue (IgemmParams::vec_store; output rows a multiple of four floats, 16-byte aligned, no scatter): the
    // accumulators go through LDS once, 32 * WM rows at a time, and leave as dwordx4 stores of four consecutive pixels --
    // 16 instead of 64 vector-memory instructions per wave and tile (the eight waves of a CU reach their epilogues
    // together, and a wave64 store occupies the address path as long as a load does).
    [[maybe_unused]] float noise_wv = 0.0f;
    if constexpr (MOD) {
        if (p.noise) {
            noise_wv = p.noise_w[0];
#pragma unroll
            for (int k = 0; k < ZPT; ++k)
                if (tid + kBlock * k < BN) zs[tid + kBlock * k] = zreg[k];
            __syncthreads();
        }
    }
    constexpr int LDC = BN + 4;
    constexpr bool VEC_OK = AS + XS >= 32 * WM * LDC;         // the staging buffers hold one pass
    if constexpr (VEC_OK) {
        if (p.vec_store) {
            constexpr int QROW = BN / 4;                        // quads per tile row
            constexpr int VPT = 32 * WM * QROW / kBlock;        // quads per thread per pass
            float* Cs = As;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                // the residual of this pass is fetched BEFORE the LDS round trip (its addresses do not depend on it): issued
                // after it, every store waited for its own load -- the 1x1 skip convs (K loops of 4-16 chunks) are all
                // epilogue and ran at 44 TFLOP/s
                [[maybe_unused]] f32x4 resq[RESIDUAL ? VPT : 1];
                if (RESIDUAL && p.residual) {
#pragma unroll
                    for (int v = 0; v < VPT; ++v) {
                        const int qi = tid + kBlock * v;
                        const int row = qi / QROW, qx = qi - row * QROW;
                        const int m = m0 + ((row >> 5) * MI + mi) * 32 + (row & 31);
                        const int pp = 4 * qx;
                        const int px = pp & (TW - 1);
                        const int py = (pp >> p.tw_log2) & (TH - 1);
                        const int pn = pp >> (p.tw_log2 + p.th_log2);
                        const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
                        const bool ok = m < p.M && n < p.N && oy < p.OH && ox < p.OW;
                        const int64_t yi = ok ? (((int64_t)n * p.M + m) * p.YH + oy) * p.YW + ox : 0;
                        resq[v] = *reinterpret_cast<const f32x4*>(p.residual + yi);
                    }
                }
                __syncthreads();       // the K loop's (or the previous pass's) readers are done with As
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                    for (int r = 0; r < 16; ++r)
                        Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + (wn * NI + ni) * 32 + l31] = acc[mi][ni][r];
                __syncthreads();
#pragma unroll
                for (int v = 0; v < VPT; ++v) {
                    const int qi = tid + kBlock * v;
                    const int row = qi / QROW, qx = qi - row * QROW;
                    const int m = m0 + ((row >> 5) * MI + mi) * 32 + (row & 31);
                    const int pp = 4 * qx;
                    const int px = pp & (TW - 1);
                    const int py = (pp >> p.tw_log2) & (TH - 1);
                    const int pn = pp >> (p.tw_log2 + p.th_log2);
                    const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
                    if (m < p.M && n < p.N && oy < p.OH && ox < p.OW) {
                        f32x4 c = *reinterpret_cast<const f32x4*>(Cs + row * LDC + 4 * qx);
                        if (p.act) {   // bias + leaky-ReLU of the following FusedLeakyReLU, fused_bias_act_kernel.cu:30,47
                            const float bv = p.bias ? p.bias[m] : 0.0f;
                            if constexpr (MOD) {
                                if (p.noise) {   // (image + weight * noise) + bias, the reference's association (:340-351)
                                    const f32x4 z = *reinterpret_cast<const f32x4*>(zs + 4 * qx);
#pragma unroll
                                    for (int e = 0; e < 4; ++e) c[e] = c[e] + noise_wv * z[e];
                                }
                            }
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float t = c[e] + bv;
                                c[e] = ((t > 0.0f) ? t : t * p.act_slope) * p.act_scale;
                            }
                        }
                        const int64_t yi = (((int64_t)n * p.M + m) * p.YH + oy) * p.YW + ox;
                        if constexpr (RESIDUAL)
                            if (p.residual) c = (c + resq[v]) * p.res_scale;
                        *reinterpret_cast<f32x4*>(y + (int64_t)blockIdx.z * p.slab_stride + yi) = c;
                    }
                }
            }
            SAE_CLOCK_PHASE(6)
            SAE_CLOCK_END
            return;
        }
    }
    // epilogue: D row = (r&3) + 8*(r>>2) + 4*half, col = l31
    float bv[MI][16];        // bias of the rows this lane holds (fetched once, not per pixel tile)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            bv[mi][r] = (p.act && p.bias && m < p.M) ? p.bias[m] : 0.0f;
        }
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
        if (n < p.N && oy < p.OH && ox < p.OW) {
            float* yb = y + (int64_t)blockIdx.z * p.slab_stride +
                        ((int64_t)n * p.M * p.YH + (int64_t)oy * p.oys) * p.YW + (int64_t)ox * p.oxs;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) {
                // the 16 residual values of this accumulator tile are fetched together, branch-free (rows beyond M re-read
                // row M - 1), before any of them is used: a load inside the store loop made every store wait for its own
                // round trip (the 1x1 skip convs ran at 40 TFLOP/s)
                [[maybe_unused]] float rv[RESIDUAL ? 16 : 1];
                if (RESIDUAL && p.residual) {     // (only with oys == oxs == 1 and no K split: y and residual share indices)
                    const float* rb = p.residual + ((int64_t)n * p.M * p.YH + oy) * p.YW + ox;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        rv[r] = rb[(int64_t)(m < p.M ? m : p.M - 1) * p.YH * p.YW];
                    }
                }
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) {
                        float v = acc[mi][ni][r];
                        if (p.act) {   // bias + leaky-ReLU of the following FusedLeakyReLU, fused_bias_act_kernel.cu:30,47
                            if constexpr (MOD) {
                                if (p.noise) v = v + noise_wv * zs[pp];
                            }
                            v += bv[mi][r];
                            v = ((v > 0.0f) ? v : v * p.act_slope) * p.act_scale;
                        }
                        if constexpr (RESIDUAL)
                            if (p.residual) v = (v + rv[r]) * p.res_scale;
                        yb[(int64_t)m * p.YH * p.YW] = v;
                    }
                }
            }
        }
    }
    SAE_CLOCK_PHASE(6)
    SAE_CLOCK_END
}

#ifdef SAE_TUNING
// ------------------------------------------------------------------------------------------
// Wave-specialised form of the quad-staged 3x3 stride-1 gather ("ws"): six waves per workgroup -- four CONSUMER waves with the
// K loop of conv_igemm_kernel and not one vector-memory instruction in it, two PRODUCER waves that do nothing but move the next
// chunk global -> LDS by DMA (global_load_lds_dwordx4) into the other of two staging buffers.  One barrier per chunk.
//
// Why: in conv_igemm_kernel every wave issues its share of the chunk's loads (9 - 11 dwordx4 per 144 MFMAs) between two
// barriers; a wave64 vector-memory instruction occupies the CU's one address path for ~28 cycles and the eight waves of a CU
// issue theirs together, so a wave spends 14 - 17 % of its time in that phase and 4 - 5 % writing LDS
// (profiles/r2_phase_clock_quad.txt); the probe kernel loses 8 % of the matrix pipe to the loads alone
// (profiles/r2_mfma_clock_probe.txt: barrier only 0.977, loads + barrier 0.899).  The 8-wave kernel of round 2
// (conv_igemm_f8_kernel) moved the weights to DMA but left every wave its input loads and ran one workgroup per CU: 3 % slower.
// Here the MFMA waves wait for nothing but the barrier, and two workgroups per CU (3 waves per SIMD, 168 registers) still
// cover each other's prologue and epilogue.
//
//   * LDS: two buffers of [A: 9 taps x 8 channels x BM][X: 8 channel rows of XCAP + 64], the layouts of conv_igemm_kernel
//     (QUAD): 2 x 38.9 KB for the 64 x 256 tile, two workgroups per CU = 156 of the 160 KB;
//   * DMA data lands lane-linear (lane l of an instruction writes 16 bytes at base + 16 l): an A instruction is 64 consecutive
//     quads of the chunk's [tap][channel][BM] block, an X instruction 64 consecutive quads of one channel's patch.  Padding
//     quads (outside the image or the batch) are MASKED lanes: the patch areas are zeroed once and those cells never written;
//     the rows of channels beyond C (last chunk) are zero-filled by the producer with LDS stores;
//   * ordering: the producer waits for its own vmcnt(0), then the barrier; a consumer that has passed the barrier reads landed
//     data.  The buffer a producer overwrites during chunk c is the one every consumer finished before the previous barrier;
//   * MFMA operand order, accumulation order and the LDS-transposed epilogue are those of conv_igemm_kernel: bit-identical
//     results (tests/test_ws_gather.py).  The producers take part in the epilogue's barriers and nothing else of it.
// Plain (not style-modulated) launches with the vector epilogue and no K split; everything else stays on conv_igemm_kernel.
//
// MEASURED (same box, tools/ab_conv.py, profiles/r4_ab_wave_specialised.txt): 128 -> 128 @256^2 125.2 vs 131.0 TFLOP/s, 256 -> 256
// @128^2 128.4 vs 134.1, 64 -> 64 @64^2 (128 images) 117.4 vs 124.1: 4 - 5 % SLOWER than conv_igemm_kernel.  (A first version that
// computed both roles' state up front needed 168 registers; six waves of that size do not fit a CU twice whatever the SIMD
// placement, one workgroup per CU ran: 116.4.  With the roles as two programs it is 95.)  The loads the MFMA waves no longer issue
// were not what held the matrix pipe at 0.83: two workgroups per CU already cover each other's staging, and this form adds a
// barrier that couples six waves per chunk.  A recorded experiment like conv_igemm_f8_kernel: compiled into tuning builds only
// (-DSAE_TUNING, SAE_WS=1); the product library does not contain it.
// ------------------------------------------------------------------------------------------
constexpr int kBlockWs = 384;
#ifndef SAE_WS_WAVES
#define SAE_WS_WAVES 3          // waves per SIMD the kernel is compiled for (512 / this registers)
#endif
template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(kBlockWs, SAE_WS_WAVES) void conv_igemm_ws_kernel(const float* __restrict__ x,
                                                                  const float* __restrict__ wp,
                                                                  float* __restrict__ y, const IgemmParams p) {
    static_assert(WM * WN == 4, "4 consumer waves per workgroup");
    constexpr int T = 9, CK = 8, NPROD = 2;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    constexpr int XCAP = PatchCap<3, 1, BN>::value;
    constexpr int QCAP = BN / 2;                               // quads per channel accepted by the host
    constexpr int XK = QCAP / kWave;                           // X DMA instructions per channel
    constexpr int XROW = XCAP + 64;
    constexpr int AS = T * CK * BM, XS = CK * XROW, BUF = AS + XS;
    constexpr int A_INSTR = AS / 4 / kWave;                    // A DMA instructions per chunk
    constexpr int APW = (A_INSTR + NPROD - 1) / NPROD;         // ... per producer wave
    static_assert((AS / 4) % kWave == 0 && QCAP % kWave == 0 && XCAP % 4 == 0 && BUF % 4 == 0, "whole wave DMAs");
    __shared__ __attribute__((aligned(16))) float smem[2 * BUF];

    const int tid = threadIdx.x;
    __builtin_assume(tid < kBlockWs);
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool producer = wid >= 4;
    const int pw = wid - 4;                                    // producer index (0, 1)
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = (wid & 3) / WN, wn = (wid & 3) % WN;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = BN >> (p.tw_log2 + p.th_log2);
    int bt = blockIdx.x;
    int mt = blockIdx.y;
    if (p.xcd_order) {                                         // see conv_igemm_kernel
        const int total = gridDim.x * gridDim.y;
        if ((total & 7) == 0) {
            const int lin = blockIdx.x + gridDim.x * blockIdx.y;
            const int wk = (lin & 7) * (total >> 3) + (lin >> 3);
            mt = wk % gridDim.y;
            bt = wk / gridDim.y;
        }
    }
    const int tix = bt % p.tiles_x; bt /= p.tiles_x;
    const int tiy = bt % p.tiles_y;
    const int tin = bt / p.tiles_y;
    const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
    const int m0 = mt * BM;

    const int PH = TH + 2;
    const int RS = TW + 8;                                     // patch row = input columns [x0 - 4, x0 + TW + 4)
    const int RQ = RS >> 2;
    const int IP = PH * RS;
    const int QI = PH * RQ, QN = TN * QI;                      // quads per image, per channel (<= QCAP, host)
    const int HW = p.H * p.W;

    // the patch areas of both buffers start as zeros: a padding quad is a masked DMA lane, its cell is never written
    for (int i = tid; i < 2 * (XS / 4); i += kBlockWs) {
        const int b = i / (XS / 4), e = i - b * (XS / 4);
        *reinterpret_cast<f32x4*>(smem + b * BUF + AS + 4 * e) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
    }

    __syncthreads();                             // the zeros are in place

    // The two roles are two straight-line programs with the same sequence of barriers (one per chunk, then two per epilogue
    // pass); nothing computed for one role is live in the other, so the register allocation is the larger of the two, not their sum.
    if (producer) {
        // where this lane's cell of each DMA instruction comes from
        unsigned qoff[XK];           // X: float offset of quad q = lane + 64 k relative to (image n0, the channel's plane)
        bool qok[XK];
        unsigned aoff[APW];          // A: float offset of cell e = 64 j + lane (j = pw + NPROD i) relative to (channel c0, column m0) of wp
#pragma unroll
        for (int k = 0; k < XK; ++k) {
            const int q = lane + kWave * k;
            const int pn = q / QI;
            const int rem = q - pn * QI;
            const int r = rem / RQ;
            const int qc = rem - r * RQ;
            const int iy = oy0 - p.pad + r, ix = ox0 - 4 + 4 * qc;
            const bool in = q < QN && n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            qok[k] = in;
            qoff[k] = in ? (unsigned)(pn * p.C * HW + iy * p.W + ix) : 0u;
        }
#pragma unroll
        for (int i = 0; i < APW; ++i) {
            const int e = (pw + NPROD * i) * kWave + lane;
            const int row = e / (BM / 4), col4 = e - row * (BM / 4);
            const int tap = (row / CK) < T ? row / CK : 0, ch = row - (row / CK) * CK;
            aoff[i] = (unsigned)((tap * p.Cp + ch) * p.Mp + col4 * 4);
        }
        const float* xb = x + (int64_t)n0 * p.C * HW;
        // iteration c0 (from -CK) stages chunk c0 + CK into the buffer the consumers are NOT reading
        int buf = 0;                             // the buffer of chunk c0
        for (int c0 = -CK; c0 < p.Cp; c0 += CK) {
            const int cn = c0 + CK;
            if (cn < p.Cp) {
                float* Ab = smem + (c0 < 0 ? 0 : buf ^ 1) * BUF;
                float* Xb = Ab + AS;
                const float* wb = wp + (int64_t)cn * p.Mp + m0;
#pragma unroll
                for (int i = 0; i < APW; ++i) {
                    const int j = pw + NPROD * i;
                    // (source and destination through local pointers of non-dependent type: with `wb + aoff[i]` as the argument --
                    // an element of an array whose size depends on the template parameters -- the HOST pass of hipcc drops the
                    // kernel's stub without a diagnostic and the library fails to load with an undefined symbol)
                    const float* src = wb + aoff[i];
                    float* dst = Ab + j * kWave * 4;
                    if (j < A_INSTR)
                        __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                }
#pragma unroll
                for (int h = 0; h < CK / NPROD; ++h) {
                    const int ch = NPROD * h + pw;
                    float* row = Xb + ch * XROW;
                    if (cn + ch < p.C) {
                        const float* xc = xb + (int64_t)(cn + ch) * HW;
#pragma unroll
                        for (int k = 0; k < XK; ++k) {
                            const float* src = xc + qoff[k];
                            float* dst = row + kWave * 4 * k;
                            if (qok[k])
                                __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)dst, 16, 0, 0);
                        }
                    } else {        // a channel beyond C (the last chunk's tail): zeros, whatever an earlier chunk left there
#pragma unroll
                        for (int k = 0; k < XK; ++k)
                            *reinterpret_cast<f32x4*>(row + 4 * (lane + kWave * k)) = f32x4{0.0f, 0.0f, 0.0f, 0.0f};
                    }
                }
            }
            __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0): this wave's DMA has landed (the barrier publishes it)
            __syncthreads();
            if (c0 >= 0) buf ^= 1;
        }
#pragma unroll
        for (int mi = 0; mi < MI; ++mi) {        // the consumers' epilogue passes
            __syncthreads();
            __syncthreads();
        }
        return;
    }

    // ---- consumer (conv_igemm_kernel, QUAD)
    int pixbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        pixbase[ni] = pn * IP + py * RS + px + 4 - p.pad;
    }
    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tapoff[t] = (t / 3) * RS + (t % 3);
    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    __syncthreads();                             // chunk 0 is staged (the producers' iteration c0 = -CK)
    int buf = 0;
    for (int c0 = 0; c0 < p.Cp; c0 += CK) {
        const float* As = smem + buf * BUF;
        const float* Xs = As + AS;
        constexpr int KK = CK / 2;
        constexpr int NSTEP = T * KK;
        constexpr int AH = kIgemmAhead, RING = AH + 1;
        float ra[RING][MI], rb[RING][NI];
        auto fetch = [&](int s, float (&a)[MI], float (&b)[NI]) {
            const int t = s / KK, ch = 2 * (s % KK) + half;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi) a[mi] = As[(t * CK + ch) * BM + (wm * MI + mi) * 32 + l31];
#pragma unroll
            for (int ni = 0; ni < NI; ++ni) b[ni] = Xs[ch * XROW + pixbase[ni] + tapoff[t]];
        };
#pragma unroll
        for (int s = 0; s < AH; ++s) fetch(s, ra[s % RING], rb[s % RING]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + AH < NSTEP) fetch(s + AH, ra[(s + AH) % RING], rb[(s + AH) % RING]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % RING][mi], rb[s % RING][ni],
                                                                       acc[mi][ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        buf ^= 1;
    }

    // ---- epilogue (conv_igemm_kernel's vector epilogue; p.vec_store is a launch condition)
    constexpr int LDC = BN + 4;
    static_assert(BUF >= 32 * WM * LDC, "one epilogue pass fits a staging buffer");
    constexpr int QROW = BN / 4;
    constexpr int VPT = 32 * WM * QROW / kBlock;
    float* Cs = smem;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi) {
        __syncthreads();
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r)
                Cs[(wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * LDC + (wn * NI + ni) * 32 + l31] = acc[mi][ni][r];
        __syncthreads();
#pragma unroll
        for (int v = 0; v < VPT; ++v) {
            const int qi = tid + kBlock * v;
            const int row = qi / QROW, qx = qi - row * QROW;
            const int m = m0 + ((row >> 5) * MI + mi) * 32 + (row & 31);
            const int pp = 4 * qx;
            const int px = pp & (TW - 1);
            const int py = (pp >> p.tw_log2) & (TH - 1);
            const int pn = pp >> (p.tw_log2 + p.th_log2);
            const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
            if (m < p.M && n < p.N && oy < p.OH && ox < p.OW) {
                f32x4 c = *reinterpret_cast<const f32x4*>(Cs + row * LDC + 4 * qx);
                if (p.act) {
                    const float bv = p.bias ? p.bias[m] : 0.0f;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float t = c[e] + bv;
                        c[e] = ((t > 0.0f) ? t : t * p.act_slope) * p.act_scale;
                    }
                }
                const int64_t yi = (((int64_t)n * p.M + m) * p.YH + oy) * p.YW + ox;
                *reinterpret_cast<f32x4*>(y + yi) = c;
            }
        }
    }
}

#endif   // SAE_TUNING

// ------------------------------------------------------------------------------------------
// "bx" arithmetic (opt-in, sae_set_conv_math(1)): fp32 products on the bf16 matrix cores.
//
// Every fp32 operand is split exactly into three bf16 pieces a = a0 + a1 + a2 (round-to-nearest
// residual chain: |a1| <= 2^-9 |a|, |a2| <= 2^-18 |a|, remainder <= 2^-27 |a|).  A product block is
// six v_mfma_f32_32x32x16_bf16 (a0b2, a2b0, a1b1, a0b1, a1b0, a0b0; the three dropped cross terms are
// <= 2^-26 |ab|) accumulated in fp32: measured error against fp64 equals the exact-fp32 MFMA chain
// (5.0e-7 vs 6.0e-7 rel-L2 at K = 1152, tools/probe/bf16x6_probe.hip) at 6/16 of its matrix-pipe time.
//
// Layout: a 16-byte LDS/global cell holds the 8 channels of one K-chunk for one (tap, split, m) of A
// or one (split, patch position) of B, i.e. exactly one lane's MFMA operand.  The 16 k of an
// instruction are 8 channels x 2 taps (lanes 0-31 tap 2g, lanes 32-63 tap 2g+1); the ninth tap pairs
// with an all-zero A cell.  Weights are split once per call by conv_wprep_bx_kernel into
// wpb[m-tile][chunk][tap][split][m][8 ch], so A staging is a linear 16-byte copy; input patch
// elements are split when they are written to LDS.
// ------------------------------------------------------------------------------------------
__device__ inline void split3_bf16(const float (&v)[8], bf16x8& s0, bf16x8& s1, bf16x8& s2) {
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const __bf16 h = (__bf16)v[e];
        float r = v[e] - (float)h;
        const __bf16 m = (__bf16)r;
        r -= (float)m;
        s0[e] = h; s1[e] = m; s2[e] = (__bf16)r;
    }
}

__global__ __launch_bounds__(kBlock) void conv_wprep_bx_kernel(const float* __restrict__ w,
                                                               u32x4* __restrict__ wpb, int M, int C, int Mp,
                                                               int Cp, int BM, int64_t sm, int64_t sc, int flip,
                                                               float alpha, const float* __restrict__ rs_m,
                                                               const float* __restrict__ rs_c) {
    const int nchunks = Cp / 8;
    const int64_t total = (int64_t)(Mp / BM) * nchunks * 9 * BM;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * kBlock) {
        const int ml = (int)(i % BM);
        int64_t t = i / BM;
        const int tap = (int)(t % 9); t /= 9;
        const int chunk = (int)(t % nchunks);
        const int mtile = (int)(t / nchunks);
        const int m = mtile * BM + ml;
        float v[8];
#pragma unroll
        for (int ch = 0; ch < 8; ++ch) {
            const int c = chunk * 8 + ch;
            v[ch] = (m < M && c < C) ? alpha * w[m * sm + c * sc + (flip ? 8 - tap : tap)] : 0.0f;
            if (m < M && c < C) {
                if (rs_m) v[ch] *= rs_m[m];
                if (rs_c) v[ch] *= rs_c[c];
            }
        }
        bf16x8 s0, s1, s2;
        split3_bf16(v, s0, s1, s2);
        u32x4* cell = wpb + (((int64_t)mtile * nchunks + chunk) * 9 + tap) * 3 * BM + ml;
        cell[0] = __builtin_bit_cast(u32x4, s0);
        cell[BM] = __builtin_bit_cast(u32x4, s1);
        cell[2 * BM] = __builtin_bit_cast(u32x4, s2);
    }
}

template <int S, int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(kBlock) void conv_igemm_bx_kernel(const float* __restrict__ x,
                                                               const u32x4* __restrict__ wpb,
                                                               float* __restrict__ y, const IgemmParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int T = 9, CK = 8;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    constexpr int XCAP = PatchCap<3, S, BN>::value;
    constexpr int PPT = (XCAP + kBlock - 1) / kBlock;      // patch positions per thread
    constexpr int A_CELLS = T * 3 * BM;                     // 16-byte cells per A chunk
    constexpr int APT = (A_CELLS + kBlock - 1) / kBlock;
    __shared__ u32x4 As[A_CELLS + 1];                       // + the all-zero cell
    __shared__ u32x4 Xs[3 * XCAP];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = BN >> (p.tw_log2 + p.th_log2);
    int bt = blockIdx.x;
    const int tix = bt % p.tiles_x; bt /= p.tiles_x;
    const int tiy = bt % p.tiles_y;
    const int tin = bt / p.tiles_y;
    const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
    const int m0 = blockIdx.y * BM;

    const int PH = (TH - 1) * S + 3, PW = (TW - 1) * S + 3;
    const int HALFW = (PW + 1) >> 1;      // stride 2: patch columns are stored de-interleaved, even | odd
    const int IP = PH * PW;
    const int CP = TN * IP;           // staged positions (<= XCAP, checked on the host)
    const int HW = p.H * p.W;

    int poff[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int e = tid + kBlock * s;
        int off = -1;
        if (e < CP) {
            const int pn = e / IP;
            const int rem = e - pn * IP;
            const int r = rem / PW;
            const int cl = rem - r * PW;
            const int c = (S == 2) ? (cl < HALFW ? 2 * cl : 2 * (cl - HALFW) + 1) : cl;
            const int iy = oy0 * S - p.pad + r, ix = ox0 * S - p.pad + c;
            if (n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                off = pn * p.C * HW + iy * p.W + ix;
        }
        poff[s] = off;
    }

    // per-lane LDS cell of each N-tile pixel; per-lane tap offset of each tap pair
    int pixbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        pixbase[ni] = pn * IP + py * S * PW + px;
    }
    int boff[5], aoff[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int t = (g < 4) ? 2 * g + half : 8;
        boff[g] = (t / 3) * PW + ((S == 2) ? ((t % 3) & 1) * HALFW + ((t % 3) >> 1) : (t % 3));
        aoff[g] = (g == 4 && half) ? A_CELLS : t * 3 * BM + wm * MI * 32 + l31;
    }

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    const u32x4* wt = wpb + (int64_t)blockIdx.y * (p.Cp / CK) * A_CELLS;
    float xv[CK][PPT];
    u32x4 av[APT];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const bool ch_ok = (c0 + ch) < p.C;
            const float* xc = xb + (int64_t)(c0 + ch) * HW;
#pragma unroll
            for (int s = 0; s < PPT; ++s) xv[ch][s] = (ch_ok && poff[s] >= 0) ? xc[poff[s]] : 0.0f;
        }
        const u32x4* wc = wt + (int64_t)(c0 / CK) * A_CELLS;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + kBlock * i;
            if (e < A_CELLS) av[i] = wc[e];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int e = tid + kBlock * s;
            if (e < CP) {
                float v[CK];
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) v[ch] = xv[ch][s];
                bf16x8 s0, s1, s2;
                split3_bf16(v, s0, s1, s2);
                Xs[e] = __builtin_bit_cast(u32x4, s0);
                Xs[XCAP + e] = __builtin_bit_cast(u32x4, s1);
                Xs[2 * XCAP + e] = __builtin_bit_cast(u32x4, s2);
            }
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + kBlock * i;
            if (e < A_CELLS) As[e] = av[i];
        }
    };

    if (tid == 0) As[A_CELLS] = u32x4{0u, 0u, 0u, 0u};
    const int c_begin = blockIdx.z * p.chunks_per_split * CK;
    int c_end = c_begin + p.chunks_per_split * CK;
    if (c_end > p.Cp) c_end = p.Cp;
    load_chunk(c_begin);
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        __syncthreads();   // everyone finished reading the previous chunk
        store_chunk();
        __syncthreads();
        if (c0 + CK < c_end) load_chunk(c0 + CK);   // in flight under the MFMAs below
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            bf16x8 a[MI][3], b[NI][3];
            // the zero cell has no split planes: its lanes read it for every split
            const int astep = (g == 4 && half) ? 0 : BM;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    a[mi][sp] = __builtin_bit_cast(bf16x8, As[aoff[g] + sp * astep + ((g == 4 && half) ? 0 : mi * 32)]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    b[ni][sp] = __builtin_bit_cast(bf16x8, Xs[sp * XCAP + pixbase[ni] + boff[g]]);
            // smallest terms first
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
            constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][TA[q]], b[ni][TB[q]],
                                                                              acc[mi][ni], 0, 0, 0);
        }
    }

    // epilogue: D row = (r&3) + 8*(r>>2) + 4*half, col = l31
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
        if (n < p.N && oy < p.OH && ox < p.OW) {
            float* yb = y + (int64_t)blockIdx.z * p.slab_stride +
                        ((int64_t)n * p.M * p.YH + (int64_t)oy * p.oys) * p.YW + (int64_t)ox * p.oxs;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) {
                        float v = acc[mi][ni][r];
                        if (p.act) {
                            if (p.bias) v += p.bias[m];
                            v = ((v > 0.0f) ? v : v * p.act_slope) * p.act_scale;
                        }
                        yb[(int64_t)m * p.YH * p.YW] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// "bx" gather, 8 waves, LDS-DMA weights (128 x 256-pixel tile, one 512-thread workgroup per CU).
// conv_igemm_bx_kernel loses ~20 % to its staging phase (with the staging removed it runs 245 TFLOP/s-eq):
// two 4-wave workgroups per CU drift into phase and then both sit in store-wait-barrier while the matrix
// pipe idles.  Here the whole CU is ONE workgroup with everything double buffered in its 160 KB of LDS:
//   * the split weights of chunk k+1 go global -> LDS by DMA (global_load_lds_dwordx4: the wpb tile is a
//     linear image of the LDS tile, lane l lands at base + 16 l), no registers, no ds_write pass;
//   * the input patch of chunk k+1 is loaded to registers (one position per thread), split and written to
//     the other Xs buffer after the MFMAs of chunk k;
//   * one barrier per chunk (s_waitcnt vmcnt(0) first: DMA data is ordered for other waves' ds_reads only
//     by the issuer's vmcnt followed by a barrier the reader has passed).
// The A tile is shared by 8 waves instead of 4, halving its L2 traffic per MFMA.
// ------------------------------------------------------------------------------------------
constexpr int kBlock8 = 512;

template <int MI, int NI, int WM, int WN>
__global__ __launch_bounds__(kBlock8) void conv_igemm_bx8_kernel(const float* __restrict__ x,
                                                                 const u32x4* __restrict__ wpb,
                                                                 float* __restrict__ y, const IgemmParams p) {
    static_assert(WM * WN == 8, "8 waves per workgroup");
    constexpr int T = 9, CK = 8;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    constexpr int XCAP = 2 * BN;                            // patch positions (host-checked): one per thread
    static_assert(XCAP == kBlock8, "one patch position per thread");
    constexpr int A_CELLS = T * 3 * BM;                     // 16-byte cells per A chunk
    constexpr int A_INSTR = A_CELLS / kWave;                // DMA instructions per chunk (64 cells each)
    static_assert(A_CELLS % kWave == 0, "A tile is a whole number of wave DMAs");
    __shared__ u32x4 As[2][A_CELLS];
    __shared__ u32x4 Xs[2][3 * XCAP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = BN >> (p.tw_log2 + p.th_log2);
    int bt = blockIdx.x;
    const int tix = bt % p.tiles_x; bt /= p.tiles_x;
    const int tiy = bt % p.tiles_y;
    const int tin = bt / p.tiles_y;
    const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
    const int m0 = blockIdx.y * BM;

    const int PH = TH + 2, PW = TW + 2;
    const int IP = PH * PW;
    const int CP = TN * IP;           // staged positions (<= XCAP, checked on the host)
    const int HW = p.H * p.W;

    int poff = -1;
    if (tid < CP) {
        const int pn = tid / IP;
        const int rem = tid - pn * IP;
        const int r = rem / PW;
        const int c = rem - r * PW;
        const int iy = oy0 - p.pad + r, ix = ox0 - p.pad + c;
        if (n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
            poff = pn * p.C * HW + iy * p.W + ix;
    }

    int pixbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        pixbase[ni] = pn * IP + py * PW + px;
    }
    int boff[5], aoff[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int t = (g < 4) ? 2 * g + half : 8;
        boff[g] = (t / 3) * PW + (t % 3);
        aoff[g] = t * 3 * BM + wm * MI * 32 + l31;
    }
    // The ninth tap has no partner inside a chunk.  Instead of pairing it with zeros (10 % of all MFMAs), the
    // lower half-wave keeps the tap-8 operands of every EVEN chunk in registers and the upper half-wave loads
    // those of the following ODD chunk: one MFMA group per chunk pair whose 16 k are 8 channels of each chunk.
    bf16x8 a8[MI][3], b8[NI][3];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) a8[mi][sp] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) b8[ni][sp] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
    bool odd = false;

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    const u32x4* wt = wpb + (int64_t)blockIdx.y * (p.Cp / CK) * A_CELLS;
    float xv[CK];

    auto dma_a = [&](int c0, int buf) {
        const u32x4* wc = wt + (int64_t)(c0 / CK) * A_CELLS + lane;
        for (int j = wid; j < A_INSTR; j += WM * WN)
            __builtin_amdgcn_global_load_lds(wc + j * kWave, (__attribute__((address_space(3))) void*)(&As[buf][j * kWave]),
                                             16, 0, 0);
    };
    auto load_x = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            // branch-free: invalid positions read element 0 of the tile's first image and are zeroed in store_x
            const bool ok = (c0 + ch) < p.C && poff >= 0;
            xv[ch] = xb[ok ? (int64_t)(c0 + ch) * HW + poff : 0];
        }
    };
    auto store_x = [&](int c0, int buf) {
        if (tid < CP) {
            float v[CK];
#pragma unroll
            for (int ch = 0; ch < CK; ++ch) v[ch] = ((c0 + ch) < p.C && poff >= 0) ? xv[ch] : 0.0f;
            bf16x8 s0, s1, s2;
            split3_bf16(v, s0, s1, s2);
            Xs[buf][tid] = __builtin_bit_cast(u32x4, s0);
            Xs[buf][XCAP + tid] = __builtin_bit_cast(u32x4, s1);
            Xs[buf][2 * XCAP + tid] = __builtin_bit_cast(u32x4, s2);
        }
    };

    const int c_begin = blockIdx.z * p.chunks_per_split * CK;
    int c_end = c_begin + p.chunks_per_split * CK;
    if (c_end > p.Cp) c_end = p.Cp;
    dma_a(c_begin, 0);
    load_x(c_begin);
    store_x(c_begin, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA has landed
    __syncthreads();
    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        const bool more = c0 + CK < c_end;
        if (more) {
            dma_a(c0 + CK, buf ^ 1);         // lands in the other buffer under the MFMAs below
            load_x(c0 + CK);
        }
        const u32x4* Ac = As[buf];
        const u32x4* Xc = Xs[buf];
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // the split + LDS write of the next patch sits in the middle of the MFMA stream (same basic block:
            // its VALU / ds_write instructions issue in the shadow of the matrix pipe)
            if (g == 3 && more) store_x(c0 + CK, buf ^ 1);
            bf16x8 a[MI][3], b[NI][3];
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) a[mi][sp] = __builtin_bit_cast(bf16x8, Ac[aoff[g] + sp * BM + mi * 32]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    b[ni][sp] = __builtin_bit_cast(bf16x8, Xc[sp * XCAP + pixbase[ni] + boff[g]]);
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][TA[q]], b[ni][TB[q]],
                                                                              acc[mi][ni], 0, 0, 0);
        }
        // tap 8: even chunk -> the lower half-wave latches its operands; odd chunk -> the upper half-wave loads
        // its own and the pair is multiplied
        if (!odd || half) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) a8[mi][sp] = __builtin_bit_cast(bf16x8, Ac[aoff[4] + sp * BM + mi * 32]);
#pragma unroll
            for (int ni = 0; ni < NI; ++ni)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    b8[ni][sp] = __builtin_bit_cast(bf16x8, Xc[sp * XCAP + pixbase[ni] + boff[4]]);
        }
        if (odd) {
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mi][TA[q]], b8[ni][TB[q]],
                                                                              acc[mi][ni], 0, 0, 0);
        }
        odd = !odd;
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) before the barrier: see the header comment
        __syncthreads();
        buf ^= 1;
    }

    if (odd) {   // odd number of chunks: the last tap-8 operands pair with zeros in the upper half-wave
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
        if (half) {
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp) a8[mi][sp] = __builtin_bit_cast(bf16x8, u32x4{0u, 0u, 0u, 0u});
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int ni = 0; ni < NI; ++ni)
                    acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a8[mi][TA[q]], b8[ni][TB[q]], acc[mi][ni],
                                                                          0, 0, 0);
    }

    // epilogue: D row = (r&3) + 8*(r>>2) + 4*half, col = l31
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
        if (n < p.N && oy < p.OH && ox < p.OW) {
            float* yb = y + (int64_t)blockIdx.z * p.slab_stride +
                        ((int64_t)n * p.M * p.YH + (int64_t)oy * p.oys) * p.YW + (int64_t)ox * p.oxs;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) {
                        float v = acc[mi][ni][r];
                        if (p.act) {
                            if (p.bias) v += p.bias[m];
                            v = ((v > 0.0f) ? v : v * p.act_slope) * p.act_scale;
                        }
                        yb[(int64_t)m * p.YH * p.YW] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// fp32 3x3 stride-1 gather on 8 waves with LDS-DMA weights: 128 x 256-pixel tile, ONE 512-thread workgroup per CU.
// conv_igemm_kernel<3,1,2,2,2,2,8> keeps the matrix pipe 80 % busy: each of its two workgroups per CU stops at two
// barriers per 8-channel chunk, writes the chunk to LDS from registers in between, and the two drift into phase.
// Here (the structure of conv_igemm_bx8_kernel, in exact fp32):
//   * the weight tile of chunk k+1 goes global -> LDS by DMA (global_load_lds_dwordx4, 16 B per lane: two rows
//     [tap][channel][128 m] of the re-laid weights per wave instruction, landing lane-linear in the other buffer):
//     no registers, no ds_write pass;
//   * the input patch of chunk k+1 (one position per thread, 8 channels) is loaded to registers before the MFMAs of
//     chunk k and written to the other buffer in the middle of them;
//   * ONE barrier per chunk (vmcnt(0) first: DMA data is ordered for other waves' ds_reads only by the issuer's
//     vmcnt followed by a barrier the reader has passed);
//   * the weight tile is shared by 8 waves instead of 4: half the L2 -> LDS weight traffic per MFMA.
// MFMA operand order, accumulation order per output element and the epilogue are those of conv_igemm_kernel, so the
// two kernels produce bit-identical results.  It measured 3 % SLOWER than the 4-wave kernel: a recorded experiment, compiled
// only into tuning builds (-DSAE_TUNING, selected there with SAE_F8=1); the product library does not contain it.
// ------------------------------------------------------------------------------------------
template <int MI, int NI, int WM, int WN, bool MOD = false>
__global__ __launch_bounds__(kBlock8) void conv_igemm_f8_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ wp,
                                                                float* __restrict__ y, const IgemmParams p) {
    static_assert(WM * WN == 8, "8 waves per workgroup");
    constexpr int T = 9, CK = 8;
    constexpr int BM = 32 * MI * WM;
    constexpr int BN = 32 * NI * WN;
    constexpr int XCAP = 2 * BN;                            // patch positions (host-checked): one per thread
    static_assert(XCAP == kBlock8, "one patch position per thread");
    constexpr int A_FLOATS = T * CK * BM;                   // weight tile of one chunk
    constexpr int A_INSTR = A_FLOATS / 4 / kWave;           // DMA instructions per chunk (64 16-byte cells each)
    static_assert((A_FLOATS / 4) % kWave == 0 && BM % 4 == 0, "A tile is a whole number of wave DMAs");
    __shared__ float As[2][A_FLOATS];
    __shared__ float Xs[2][CK * XCAP];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = BN >> (p.tw_log2 + p.th_log2);
    int bt = blockIdx.x;
    if (p.xcd_order && (gridDim.x & 7) == 0) bt = (bt & 7) * (gridDim.x >> 3) + (bt >> 3);   // see conv_igemm_kernel
    const int tix = bt % p.tiles_x; bt /= p.tiles_x;
    const int tiy = bt % p.tiles_y;
    const int tin = bt / p.tiles_y;
    const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
    const int m0 = blockIdx.y * BM;

    const int PH = TH + 2, PW = TW + 2;
    const int IP = PH * PW;
    const int CP = TN * IP;           // staged positions (<= XCAP, checked on the host)
    const int HW = p.H * p.W;

    int poff = -1;
    [[maybe_unused]] int sidx = 0;    // in_scale row of this thread's patch position
    if (tid < CP) {
        const int pn = tid / IP;
        const int rem = tid - pn * IP;
        const int r = rem / PW;
        const int c = rem - r * PW;
        const int iy = oy0 - p.pad + r, ix = ox0 - p.pad + c;
        if (n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W) {
            poff = pn * p.C * HW + iy * p.W + ix;
            if constexpr (MOD) sidx = (n0 + pn) * p.C;
        }
    }

    int pixbase[NI];
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        pixbase[ni] = pn * IP + py * PW + px;
    }
    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) tapoff[t] = (t / 3) * PW + (t % 3);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int ni = 0; ni < NI; ++ni)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    float xv[CK];
    [[maybe_unused]] float sv[MOD ? CK : 1];

    // lane's 16-byte cell of DMA instruction j: cell e = 64 j + lane = row (tap * CK + ch) * (BM / 4) + col4
    auto dma_a = [&](int c0, int buf) {
        for (int j = wid; j < A_INSTR; j += WM * WN) {
            const int e = j * kWave + lane;
            const int row = e / (BM / 4), col4 = e - row * (BM / 4);
            const int tap = row / CK, ch = row - tap * CK;
            const float* src = wp + ((int64_t)tap * p.Cp + c0 + ch) * p.Mp + m0 + col4 * 4;
            __builtin_amdgcn_global_load_lds(src, (__attribute__((address_space(3))) void*)(&As[buf][j * kWave * 4]), 16, 0, 0);
        }
    };
    auto load_x = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            // branch-free: invalid positions read element 0 of the tile's first image and are zeroed in store_x
            const bool ok = (c0 + ch) < p.C && poff >= 0;
            xv[ch] = xb[ok ? (int64_t)(c0 + ch) * HW + poff : 0];
            if constexpr (MOD) sv[ch] = p.in_scale[sidx + ((c0 + ch) < p.C ? c0 + ch : 0)];
        }
    };
    auto store_x = [&](int c0, int buf) {
        if (tid < CP) {
#pragma unroll
            for (int ch = 0; ch < CK; ++ch) {
                float v = ((c0 + ch) < p.C && poff >= 0) ? xv[ch] : 0.0f;
                if constexpr (MOD) v *= sv[ch];
                Xs[buf][ch * XCAP + tid] = v;
            }
        }
    };

    const int c_begin = blockIdx.z * p.chunks_per_split * CK;
    int c_end = c_begin + p.chunks_per_split * CK;
    if (c_end > p.Cp) c_end = p.Cp;
    dma_a(c_begin, 0);
    load_x(c_begin);
    store_x(c_begin, 0);
    __builtin_amdgcn_s_waitcnt(0x0F70);      // vmcnt(0): this wave's DMA has landed
    __syncthreads();
    int buf = 0;
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        const bool more = c0 + CK < c_end;
        if (more) {
            dma_a(c0 + CK, buf ^ 1);         // lands in the other buffer under the MFMAs below
            load_x(c0 + CK);
        }
        const float* Ac = As[buf];
        const float* Xc = Xs[buf];
#pragma unroll
        for (int t = 0; t < T; ++t) {
            // the LDS write of the next patch sits in the middle of the MFMA stream (same basic block: its ds_write
            // instructions issue in the shadow of the matrix pipe)
            if (t == 6 && more) store_x(c0 + CK, buf ^ 1);
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                const int ch = 2 * kk + half;
                float a[MI], b[NI];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) a[mi] = Ac[(t * CK + ch) * BM + (wm * MI + mi) * 32 + l31];
#pragma unroll
                for (int ni = 0; ni < NI; ++ni) b[ni] = Xc[ch * XCAP + pixbase[ni] + tapoff[t]];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int ni = 0; ni < NI; ++ni)
                        acc[mi][ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[mi], b[ni], acc[mi][ni], 0, 0, 0);
            }
        }
        __builtin_amdgcn_s_waitcnt(0x0F70);  // vmcnt(0) before the barrier: see the header comment
        __syncthreads();
        buf ^= 1;
    }

    // epilogue: D row = (r&3) + 8*(r>>2) + 4*half, col = l31
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
        const int pp = (wn * NI + ni) * 32 + l31;
        const int px = pp & (TW - 1);
        const int py = (pp >> p.tw_log2) & (TH - 1);
        const int pn = pp >> (p.tw_log2 + p.th_log2);
        const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
        if (n < p.N && oy < p.OH && ox < p.OW) {
            float* yb = y + (int64_t)blockIdx.z * p.slab_stride +
                        ((int64_t)n * p.M * p.YH + (int64_t)oy * p.oys) * p.YW + (int64_t)ox * p.oxs;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) {
                        float v = acc[mi][ni][r];
                        if (p.act) {
                            if (p.bias) v += p.bias[m];
                            v = ((v > 0.0f) ? v : v * p.act_slope) * p.act_scale;
                        }
                        yb[(int64_t)m * p.YH * p.YW] = v;
                    }
                }
        }
    }
}

// ------------------------------------------------------------------------------------------
// stride-2 transposed gather ("tr"): out[m][o] = sum_{c,k : o + pad = 2 i + k} wp[k][c][m] * in[c][i]
// 3x3 taps only.  N-tiles of a wave = the 4 parity classes of the same 32 q positions.
// ------------------------------------------------------------------------------------------
struct TrRegion {
    int QH, QW;           // half-resolution rectangle [qy_base, QH) x [qx_base, QW)
    int qy_base, qx_base;
    int tw, th, tn;       // q tile = TN images x TH x TW positions (any sizes with TN*TH*TW <= BQ: the
                          // transposed problems have 2^k + 1 wide grids, power-of-two tiles waste 25-90 %)
    int tiles_x, tiles_y, tiles_n;
    int blocks;           // tiles_x * tiles_y * tiles_n
    int flat;             // tr2: tiles are runs of 128 consecutive positions of the WHOLE grid in row-major order (tiles_x per image)
};
struct TrParams {
    int N, C, IH, IW;     // input tensor (the small, "y side" image)
    int M, OH, OW;        // output tensor (the large, "x side" image)
    int debug_skip_store;   // profiling aid of tuning builds (SAE_TR_NOSTORE): results are NOT written; ignored by the product
    int Cp, Mp;
    int pad;
    const float* in_scale;  // [N][C] or null: style modulation of the input, applied while staging (see IgemmParams)
    int zmask;              // always 0 (see IgemmParams)
    int mtiles, main_items, strip_items;  // tr2: M tiles; (q tile, M tile) items of region 0 and of regions 1 + 2
    // conv_igemm_tr_kernel, split K (blockIdx.z): chunks per slice and the distance between the partial outputs
    int chunks_per_split;
    int64_t slab_stride;
    // main region + right / bottom strips, all in ONE launch (blockIdx.x runs through the regions):
    // launched one after the other the two thin strips cost a full K loop of latency each on a
    // nearly empty GPU
    TrRegion reg[3];
};

// WM x WN = 4 or 8 waves.  The 8-wave form (128 rows x 128 q positions) halves the weight traffic per MFMA -- a tap of the
// transposed problem feeds only one of the four output parity classes, so the 128 x 64q tile needs twice the weight bytes
// per MFMA of the forward gather, and the issue of those loads was 33-39 % of the kernel (profiles/r2_phase_clock_*.txt).
template <int MI, int WM, int WN, int CK, bool MOD = false>
__global__ __launch_bounds__(64 * WM * WN) void conv_igemm_tr_kernel(const float* __restrict__ x,
                                                               const float* __restrict__ wp,
                                                               float* __restrict__ y, const TrParams p) {
    static_assert(WM * WN == 4 || WM * WN == 8, "4 or 8 waves per workgroup");
    constexpr int NT = 64 * WM * WN;
    constexpr int T = 9;
    constexpr int BM = 32 * MI * WM;
    constexpr int BQ = 32 * WN;                  // q positions per workgroup
    constexpr int XCAP = (25 * BQ) / 16;         // (TW+1)(TH+1)/(TW*TH) <= 25/16 for TW,TH >= 4
    constexpr int PPT = (XCAP + NT - 1) / NT;
    constexpr int A_VEC = T * CK * BM / 4;
    constexpr int APT = (A_VEC + NT - 1) / NT;
    __shared__ float As[T * CK * BM];
    __shared__ float Xs[CK * XCAP];

    SAE_CLOCK_BEGIN
    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    int bt = blockIdx.x;
    int ridx = 0;
    if (bt >= p.reg[0].blocks) {
        bt -= p.reg[0].blocks; ridx = 1;
        if (bt >= p.reg[1].blocks) { bt -= p.reg[1].blocks; ridx = 2; }
    }
    const TrRegion g = p.reg[ridx];
    const int TW = g.tw, TH = g.th, TN = g.tn;
    const int tix = bt % g.tiles_x; bt /= g.tiles_x;
    const int tiy = bt % g.tiles_y;
    const int tin = bt / g.tiles_y;
    const int qx0 = g.qx_base + tix * TW, qy0 = g.qy_base + tiy * TH, n0 = tin * TN;
    const int m0 = blockIdx.y * BM;

    const int PH = TH + 1, PW = TW + 1;          // patch row r <-> input row qy0 - 1 + r
    const int RS = PW;
    const int IP = PH * RS;
    const int CP = TN * IP;
    const int HW = p.IH * p.IW;

    int poff[PPT];
    int sidx[MOD ? PPT : 1];   // in_scale row of the slot's image (tiles that span several images only)
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int e = tid + NT * s;
        int off = -1;
        if constexpr (MOD) sidx[s] = 0;
        if (e < CP) {
            const int pn = e / IP;
            const int rem = e - pn * IP;
            const int r = rem / RS;
            const int c = rem - r * RS;
            const int iy = qy0 - 1 + r, ix = qx0 - 1 + c;
            if (n0 + pn < p.N && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW) {
                off = pn * p.C * HW + iy * p.IW + ix;
                if constexpr (MOD) sidx[s] = (n0 + pn) * p.C;
            }
        }
        poff[s] = off;
    }
    const bool one_image = TN == 1;
    float sc[MOD ? CK : 1];

    const int pp = wn * 32 + l31;
    const int pn = pp / (TW * TH);
    const int prem = pp - pn * (TW * TH);
    const int py = prem / TW;
    const int px = prem - py * TW;
    const bool lane_ok = pn < TN;                 // lanes beyond TN*TH*TW positions idle
    const int pixbase = lane_ok ? pn * IP + py * RS + px : 0;
    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ky = t / 3, kx = t % 3;
        // tap k contributes from input index q - (k == 2 ? 1 : 0); patch row of input q is q - q0 + 1
        tapoff[t] = ((ky == 2) ? 0 : 1) * RS + ((kx == 2) ? 0 : 1);
    }

    f32x16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][cl][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    float xv[CK][PPT];
    f32x4 av[APT];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const bool ch_ok = (c0 + ch) < p.C;
            const float* xc = xb + (int64_t)(c0 + ch) * HW;
#pragma unroll
            for (int s = 0; s < PPT; ++s) xv[ch][s] = (ch_ok && poff[s] >= 0) ? xc[poff[s]] : 0.0f;
        }
        if constexpr (MOD) {
            if (one_image) {
                const float* srow = p.in_scale + n0 * p.C + (lane & p.zmask);
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) sc[ch] = srow[(c0 + ch) < p.C ? c0 + ch : 0];
            }
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + NT * i;
            if (e4 < A_VEC) {
                const int row = e4 / (BM / 4);
                const int col4 = e4 - row * (BM / 4);
                const int tap = row / CK, ch = row - tap * CK;
                av[i] = *reinterpret_cast<const f32x4*>(
                    wp + ((int64_t)tap * p.Cp + c0 + ch) * p.Mp + m0 + col4 * 4);
            }
        }
    };
    auto store_chunk = [&](int c0) {
        if constexpr (MOD) {
            if (one_image) {
#pragma unroll
                for (int ch = 0; ch < CK; ++ch)
#pragma unroll
                    for (int s = 0; s < PPT; ++s) xv[ch][s] *= sc[ch];
            } else {     // small images, several per tile: the factor depends on the slot's image
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) {
                    const int cc = (c0 + ch) < p.C ? c0 + ch : 0;
#pragma unroll
                    for (int s = 0; s < PPT; ++s) xv[ch][s] *= p.in_scale[sidx[s] + cc];
                }
            }
        }
#pragma unroll
        for (int ch = 0; ch < CK; ++ch)
#pragma unroll
            for (int s = 0; s < PPT; ++s) {
                const int e = tid + NT * s;
                if (e < CP) Xs[ch * XCAP + e] = xv[ch][s];
            }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + NT * i;
            if (e4 < A_VEC) *reinterpret_cast<f32x4*>(As + e4 * 4) = av[i];
        }
    };

    // split K over blockIdx.z (launches with few workgroups and long channel loops: the 4 x 4 ... 17 x 17 tails and the
    // encoder's 1024-channel 7 x 7 layer); the partial outputs go to slabs that conv_splitk_reduce_kernel sums in fixed order
    const int c_begin = blockIdx.z * p.chunks_per_split * CK;
    int c_end = c_begin + p.chunks_per_split * CK;
    if (c_end > p.Cp) c_end = p.Cp;
    y += (int64_t)blockIdx.z * p.slab_stride;
    load_chunk(c_begin);
    SAE_CLOCK_PHASE(0)
    for (int c0 = c_begin; c0 < c_end; c0 += CK) {
        __syncthreads();
        SAE_CLOCK_PHASE(2)
        store_chunk(c0);
        SAE_CLOCK_PHASE(3)
        __syncthreads();
        SAE_CLOCK_PHASE(4)
        if (c0 + CK < c_end) load_chunk(c0 + CK);
        SAE_CLOCK_PHASE(5)
#pragma unroll
        for (int t = 0; t < T; ++t) {
            const int cls = ((t / 3) & 1) * 2 + ((t % 3) & 1);   // parity class fed by this tap
#pragma unroll
            for (int kk = 0; kk < CK / 2; ++kk) {
                const int ch = 2 * kk + half;
                const float b = Xs[ch * XCAP + pixbase + tapoff[t]];
#pragma unroll
                for (int mi = 0; mi < MI; ++mi) {
                    const float a = As[(t * CK + ch) * BM + (wm * MI + mi) * 32 + l31];
                    // cls is a compile-time constant after unrolling t
                    if (cls == 0) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][0], 0, 0, 0);
                    else if (cls == 1) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][1], 0, 0, 0);
                    else if (cls == 2) acc[mi][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][2], 0, 0, 0);
                    else acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][3], 0, 0, 0);
                }
            }
        }
        SAE_CLOCK_PHASE(1)
    }

    const int n = n0 + pn, qy = qy0 + py, qx = qx0 + px;
#ifdef SAE_TUNING
    if (p.debug_skip_store) {   // keep the accumulators alive without the store traffic
        float keep = 0.0f;
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) keep += acc[mi][cl][0] + acc[mi][cl][15];
        if (keep == 12345.678f) y[0] = keep;
        return;
    }
#endif
    if (lane_ok && n < p.N && qy < g.QH && qx < g.QW) {   // q beyond this launch's region belongs to another launch
        // (pairing the two x-classes of a lane into one 4-byte-aligned 8-byte store was measured
        // slower, 87 vs 94 TFLOP/s: the rows are 2^k + 1 wide, so half of those stores are misaligned)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int oy = 2 * qy + (cl >> 1) - p.pad;
            const int ox = 2 * qx + (cl & 1) - p.pad;
            if (oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW) {
                float* yb = y + ((int64_t)n * p.M * p.OH + oy) * p.OW + ox;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (m < p.M) yb[(int64_t)m * p.OH * p.OW] = acc[mi][cl][r];
                    }
            }
        }
    }
    SAE_CLOCK_PHASE(6)
    SAE_CLOCK_END
}

// ------------------------------------------------------------------------------------------
// transposed gather, second generation ("tr2"): 32*MI rows x 128 q positions, CK = 16 or 8 channels per chunk.
//
// Same arithmetic in the same order as conv_igemm_tr_kernel (8-channel sub-chunks, tap-major, channel pairs), so its
// results are BIT-identical; what changes is how operands reach LDS and how results leave the registers:
//  * the input patch rows are widened to whole 16-byte quads [qx0 - 4, qx0 + TW) (TW a multiple of 4, input rows a multiple
//    of four floats wide) and a chunk's quads are dealt out over the workgroup: 3 dwordx4 per thread and 16-channel chunk
//    instead of 8 dword loads per 8 channels.  With the 9 weight quads that is 12 vector-memory instructions per 144 MFMAs
//    of a wave (13 per 72 before); their issue was 33-39 % of the kernel (profiles/r2_phase_clock_*.txt);
//  * 1-D grid in XCD-aware order with the M tile as the FASTEST index: the workgroups that share an input patch (all M
//    tiles of a q tile) and its spatial neighbours run at the same time behind the same L2 (the plain order read the
//    input 6.1x from the fabric, profiles/r2_pmc_f32_quad.txt);
//  * the epilogue goes through LDS wave by wave: a lane holds both x-classes of a q position, i.e. two adjacent output
//    columns, so the four classes of a wave's 32 positions are two complete output rows of 64 consecutive columns; they
//    leave as 4-byte aligned dwordx4 stores (32 per lane instead of 128 stride-2 dword stores that wrote every 64-byte
//    line twice);
//  * the style factors of a modulated launch are fetched once into LDS (tiles never span images then).
// Tiles are TN x TH x TW positions with TN (TH + 1) (TW + 4) <= kTr2Cap, chosen per region on the host.
// ------------------------------------------------------------------------------------------
constexpr int kTr2Cap = 192;        // patch floats per channel (rectangular tiles)
constexpr int kTr2FlatCap = 288;    // ... of a flat tile: the rows a run of 128 positions touches, full width (65-wide grids: 4 x 72)
constexpr int kTr2MaxC = 2048;      // style factors held in LDS (modulated launches)

// FLAT (small 2^k + 1 grids: 17, 33, 65 wide): a tile is a run of 128 consecutive positions of the whole q grid of one image in
// row-major order -- no main region + strips (a 33 x 33 grid is 1024 + 65 positions: the strips were separate tiles with one
// active wave and a scalar epilogue, 15 - 30 % of these launches) and no padded tile rows: 94 % of the lanes carry a position
// (9 tiles of 128 for 1089).  The patch is the 3 - 9 full-width input rows the run touches; a lane's two x-classes are two
// adjacent output columns and leave as one 8-byte store.  Same arithmetic per output in the same order: bit-identical.
template <int MI, int CK, bool MOD = false, bool FLAT = false>
__global__ __launch_bounds__(kBlock, 2) void conv_igemm_tr2_kernel(const float* __restrict__ x,
                                                                const float* __restrict__ wp,
                                                                float* __restrict__ y, const TrParams p) {
    static_assert(CK == 8 || CK == 16, "8- or 16-channel chunks");
    constexpr int T = 9;
    constexpr int BM = 32 * MI;
    constexpr int XCAP = FLAT ? kTr2FlatCap : kTr2Cap;
    constexpr int QCAP = XCAP / 4;
    constexpr int QPT = (CK * QCAP + kBlock - 1) / kBlock;       // quad slots per thread per chunk
    constexpr int A_VEC = T * CK * BM / 4;
    constexpr int APT = (A_VEC + kBlock - 1) / kBlock;
    constexpr int XS = CK * XCAP + 4 * kBlock;                   // + a dump area for the slots beyond the patch
    constexpr int AS = T * CK * BM;
    constexpr int CS = 4 * 8 * 128;                              // epilogue: 8 rows x 128 floats per wave and pass
    constexpr int LDSF = (AS + XS > CS) ? AS + XS : CS;
    __shared__ __attribute__((aligned(16))) float smem[LDSF];
    __shared__ float Ss[MOD ? kTr2MaxC : 1];
    float* As = smem;
    float* Xs = smem + AS;

    SAE_CLOCK_BEGIN
    const int tid = threadIdx.x;
    __builtin_assume(tid < kBlock);
    const int lane = tid & 63, wn = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;

    // Work items = (q tile, M tile), M tile fastest.  XCD k (= workgroup id % 8) takes the k-th contiguous eighth of the strip
    // items and then the k-th contiguous eighth of the main region's: the thin strip tiles (one active wave, latency-bound
    // K loop) run FIRST, next to main tiles on the same CUs, instead of alone on an emptying GPU at the end of the launch
    const int xcd = (int)blockIdx.x & 7, slot = (int)blockIdx.x >> 3;
    const int ps = (p.strip_items + 7) >> 3, pm = (p.main_items + 7) >> 3;
    int wk;
    if (slot < ps) {
        wk = xcd * ps + slot;
        if (wk >= p.strip_items) return;
        wk += p.main_items;
    } else {
        wk = xcd * pm + slot - ps;
        if (wk >= p.main_items) return;
    }
    const int mt = wk % p.mtiles;
    int bt = wk / p.mtiles;
    int ridx = 0;
    if (bt >= p.reg[0].blocks) {
        bt -= p.reg[0].blocks; ridx = 1;
        if (bt >= p.reg[1].blocks) { bt -= p.reg[1].blocks; ridx = 2; }
    }
    const TrRegion g = p.reg[ridx];
    const int TW = g.tw, TH = g.th, TN = g.tn;
    const int tix = bt % g.tiles_x; bt /= g.tiles_x;
    const int tiy = bt % g.tiles_y;
    const int tin = bt / g.tiles_y;
    // FLAT: tile tix of image tin starts at position 128 tix = (row qy0, column s0) of the grid; TW = the grid's width, TH the
    // most rows a run can touch (patch rows beyond the image are zero-filled like any padding)
    const int qx0 = FLAT ? 0 : g.qx_base + tix * TW;
    const int qy0 = FLAT ? (128 * tix) / TW : g.qy_base + tiy * TH;
    const int n0 = tin * TN;
    [[maybe_unused]] const int s0 = FLAT ? 128 * tix - qy0 * TW : 0;
    const int m0 = mt * BM;

    const int PH = TH + 1;                 // patch row r <-> input row qy0 - 1 + r
    const int RS = ((TW + 3) & ~3) + 4;    // patch column c <-> input column qx0 - 4 + c (qx0 is a multiple of 4)
    const int RQ = RS >> 2;
    const int IP = PH * RS;
    const int QI = PH * RQ, QN = TN * QI;  // quads per image, per channel
    const int HW = p.IH * p.IW;

    // quad slot k of a thread = quad j = tid + kBlock * k of the chunk: channel j / QN, quad j % QN of the patch.  Slots
    // outside the image (zero padding) or beyond the chunk fetch element 0 and are zeroed / dumped at the LDS write, so the
    // K loop has no divergent branch and every load is "uniform 64-bit base + loop-invariant 32-bit lane offset"
    unsigned qbyte[QPT];
    int qdst[QPT], qch[QPT];       // qch < 0: never valid
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int j = tid + kBlock * k;
        const int ch = j / QN;
        const int q = j - ch * QN;
        const int pn = q / QI;
        const int rem = q - pn * QI;
        const int r = rem / RQ;
        const int qc = rem - r * RQ;
        const int iy = qy0 - 1 + r, ix = qx0 - 4 + 4 * qc;
        const bool slot = ch < CK;
        const bool in = slot && n0 + pn < p.N && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW;
        qbyte[k] = in ? 4u * (unsigned)((pn * p.C + ch) * HW + iy * p.IW + ix) : 0u;
        qch[k] = in ? ch : -1;
        qdst[k] = slot ? ch * XCAP + 4 * q : CK * XCAP + 4 * tid;
    }
    unsigned abyte[APT];
#pragma unroll
    for (int i = 0; i < APT; ++i) {
        const int e4 = tid + kBlock * i;
        const int row = e4 / (BM / 4);
        const int col4 = e4 - row * (BM / 4);
        const int tap = (row / CK) < T ? row / CK : 0, ch = row - (row / CK) * CK;
        abyte[i] = 4u * (unsigned)((tap * p.Cp + ch) * p.Mp + col4 * 4);
    }
    if constexpr (MOD) {    // tiles of a modulated launch lie inside one image (host)
        for (int c = tid; c < p.C; c += kBlock) Ss[c] = p.in_scale[(int64_t)n0 * p.C + c];
    }

    const int pp = wn * 32 + l31;
    const int npos = FLAT ? (g.QH * TW - 128 * tix < 128 ? g.QH * TW - 128 * tix : 128) : TN * TH * TW;
    const int pn_l = FLAT ? 0 : pp / (TW * TH);
    const int prem = FLAT ? pp + s0 : pp - pn_l * (TW * TH);
    const int py = prem / TW;
    const int px = prem - py * TW;
    const int pixbase = pp < npos ? pn_l * IP + py * RS + px : 0;     // lanes beyond the tile compute on position 0
    const bool wave_active = wn * 32 < npos;                          // (whole waves beyond it skip their MFMAs: strip tiles)
    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ky = t / 3, kx = t % 3;
        // tap k contributes from input index q - (k == 2 ? 1 : 0)
        tapoff[t] = ((ky == 2) ? 0 : 1) * RS + ((kx == 2) ? 3 : 4);
    }

    f32x16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][cl][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    f32x4 xq[QPT];
    f32x4 av[APT];

    auto load_chunk = [&](int c0) {
        const char* xc = reinterpret_cast<const char*>(xb + (int64_t)c0 * HW);
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const bool ok = qch[k] >= 0 && c0 + qch[k] < p.C;      // channels beyond C (last chunk): element 0, zeroed later
            xq[k] = *reinterpret_cast<const f32x4*>(xc + (ok ? qbyte[k] : 0u));
        }
        const char* wb = reinterpret_cast<const char*>(wp + (int64_t)c0 * p.Mp + m0);
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + kBlock * i;
            if (e4 < A_VEC) av[i] = *reinterpret_cast<const f32x4*>(wb + abyte[i]);
        }
    };
    auto store_chunk = [&](int c0) {
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const bool ok = qch[k] >= 0 && c0 + qch[k] < p.C;
            f32x4 v = xq[k];
            if constexpr (MOD) v *= Ss[ok ? c0 + qch[k] : 0];
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
            *reinterpret_cast<f32x4*>(Xs + qdst[k]) = v;
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e4 = tid + kBlock * i;
            if (e4 < A_VEC) *reinterpret_cast<f32x4*>(As + e4 * 4) = av[i];
        }
    };

    load_chunk(0);
    SAE_CLOCK_PHASE(0)
    for (int c0 = 0; c0 < p.Cp; c0 += CK) {
        __syncthreads();
        SAE_CLOCK_PHASE(2)
        store_chunk(c0);
        SAE_CLOCK_PHASE(3)
        __syncthreads();
        SAE_CLOCK_PHASE(4)
        if (c0 + CK < p.Cp) load_chunk(c0 + CK);
        SAE_CLOCK_PHASE(5)
        if (wave_active)
#pragma unroll
        for (int sub = 0; sub < CK / 8; ++sub)
#pragma unroll
            for (int t = 0; t < T; ++t) {
                const int cls = ((t / 3) & 1) * 2 + ((t % 3) & 1);   // parity class fed by this tap
#pragma unroll
                for (int kk = 0; kk < 4; ++kk) {
                    const int ch = sub * 8 + 2 * kk + half;
                    const float b = Xs[ch * XCAP + pixbase + tapoff[t]];
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi) {
                        const float a = As[(t * CK + ch) * BM + mi * 32 + l31];
                        // cls is a compile-time constant after unrolling t
                        if (cls == 0) acc[mi][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][0], 0, 0, 0);
                        else if (cls == 1) acc[mi][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][1], 0, 0, 0);
                        else if (cls == 2) acc[mi][2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][2], 0, 0, 0);
                        else acc[mi][3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[mi][3], 0, 0, 0);
                    }
                }
            }
        SAE_CLOCK_PHASE(1)
    }

    // ---- epilogue.  Per pass a wave parks 8 rows (m) x [cy][32 positions][cx] in its own LDS block; lane (cy = bit 4,
    // k = lane & 15) then owns the quad of positions 2k, 2k + 1 = four consecutive columns of output row 2 qy + cy - pad
    // for rows m8 = 2 v + (lane >> 5).
    __syncthreads();                  // every wave is done with As / Xs
    if (!wave_active) return;
    if constexpr (FLAT) {
        // a lane's classes (cy, 0) and (cy, 1) are output columns 2 qx - pad and 2 qx - pad + 1 of row 2 qy + cy - pad: one 8-byte
        // store (rows are 2^k + 1 floats: 4-byte aligned); a wave instruction writes 2 x 256 contiguous bytes
        typedef float f32x2 __attribute__((ext_vector_type(2)));
        struct __attribute__((packed, aligned(4))) U2 { f32x2 v; };
        const int qy = qy0 + py, qx = px;
        const int ox = 2 * qx - p.pad;
        const bool live = pp < npos;
        const bool c0 = live && ox >= 0 && ox < p.OW, c1 = live && ox + 1 >= 0 && ox + 1 < p.OW;
        const int64_t plane = (int64_t)p.OH * p.OW;
#pragma unroll
        for (int cy = 0; cy < 2; ++cy) {
            const int oy = 2 * qy + cy - p.pad;
            if (oy < 0 || oy >= p.OH || !(c0 || c1)) continue;
            float* yb = y + ((int64_t)n0 * p.M * p.OH + oy) * p.OW + ox;
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                    if (m < p.M) {
                        float* yp = yb + (int64_t)m * plane;
                        if (c0 && c1) reinterpret_cast<U2*>(yp)->v = f32x2{acc[mi][2 * cy][r], acc[mi][2 * cy + 1][r]};
                        else if (c0) yp[0] = acc[mi][2 * cy][r];
                        else yp[1] = acc[mi][2 * cy + 1][r];
                    }
                }
        }
        SAE_CLOCK_PHASE(6)
        SAE_CLOCK_END
        return;
    }
    if (TW & 1) {                     // one-column strips: positions 2k, 2k + 1 are not neighbours; plain stores
        const int n = n0 + pn_l, qy = qy0 + py, qx = qx0 + px;
        if (pp < npos && n < p.N && qy < g.QH && qx < g.QW) {
#pragma unroll
            for (int cl = 0; cl < 4; ++cl) {
                const int oy = 2 * qy + (cl >> 1) - p.pad;
                const int ox = 2 * qx + (cl & 1) - p.pad;
                if (oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW) {
                    float* yb = y + ((int64_t)n * p.M * p.OH + oy) * p.OW + ox;
#pragma unroll
                    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            const int m = m0 + mi * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                            if (m < p.M) yb[(int64_t)m * p.OH * p.OW] = acc[mi][cl][r];
                        }
                }
            }
        }
        return;
    }
    float* Cw = smem + wn * (8 * 128);
    const int ecy = (lane >> 4) & 1, ek = lane & 15;
    const int epp = wn * 32 + 2 * ek;
    const int epn = epp / (TW * TH);
    const int eprem = epp - epn * (TW * TH);
    const int epy = eprem / TW;
    const int epx = eprem - epy * TW;                 // even; position 2k + 1 is its right neighbour (TW is even)
    const int en = n0 + epn, eqy = qy0 + epy, eqx = qx0 + epx;
    const int oy = 2 * eqy + ecy - p.pad;
    const int ox = 2 * eqx - p.pad;
    unsigned emask = 0;               // bit e: output column ox + e exists and belongs to this tile's region
    if (epp < npos && en < p.N && eqy < g.QH && oy >= 0 && oy < p.OH) {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (eqx + (e >> 1) < g.QW && ox + e >= 0 && ox + e < p.OW) emask |= 1u << e;
    }
    const int64_t plane = (int64_t)p.OH * p.OW;
    float* yb = y + (int64_t)en * p.M * plane + (int64_t)oy * p.OW + ox;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    struct __attribute__((packed, aligned(4))) U4 { f32x4 v; };     // 4-byte aligned quad (rows are 2^k + 1 wide)
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int cy = 0; cy < 2; ++cy)
                    *reinterpret_cast<f32x2*>(Cw + (4 * half + i) * 128 + cy * 64 + 2 * l31) =
                        f32x2{acc[mi][cy * 2][4 * j + i], acc[mi][cy * 2 + 1][4 * j + i]};
            wave_sync();
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int m8 = 2 * v + half;
                const f32x4 c = *reinterpret_cast<const f32x4*>(Cw + m8 * 128 + (lane & 31) * 4);
                const int m = m0 + mi * 32 + 8 * j + m8;
                if (m < p.M) {
                    float* yp = yb + (int64_t)m * plane;
                    if (emask == 15u) reinterpret_cast<U4*>(yp)->v = c;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (emask & (1u << e)) yp[e] = c[e];
                    }
                }
            }
            wave_sync();
        }
    SAE_CLOCK_PHASE(6)
    SAE_CLOCK_END
}

// ------------------------------------------------------------------------------------------
// "bx" arithmetic of the transposed gather (128 x 64q tile).  Same cell layouts as
// conv_igemm_bx_kernel.  The 16 k of an MFMA are 8 channels x the two taps of a pair that feed
// the SAME parity class: (0,2) (6,8) -> class 0, (1,7) -> 1, (3,5) -> 2, (4, zero cell) -> 3.
// ------------------------------------------------------------------------------------------
template <int MI, int WM, int WN>
__global__ __launch_bounds__(kBlock) void conv_igemm_tr_bx_kernel(const float* __restrict__ x,
                                                                  const u32x4* __restrict__ wpb,
                                                                  float* __restrict__ y, const TrParams p) {
    static_assert(WM * WN == 4, "4 waves per workgroup");
    constexpr int T = 9, CK = 8;
    constexpr int BM = 32 * MI * WM;
    constexpr int BQ = 32 * WN;
    constexpr int XCAP = (25 * BQ) / 16;
    constexpr int PPT = (XCAP + kBlock - 1) / kBlock;
    constexpr int A_CELLS = T * 3 * BM;
    constexpr int APT = (A_CELLS + kBlock - 1) / kBlock;
    __shared__ u32x4 As[A_CELLS + 1];
    __shared__ u32x4 Xs[3 * XCAP];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wid = tid >> 6;
    const int l31 = lane & 31, half = lane >> 5;
    const int wm = wid / WN, wn = wid % WN;

    int bt = blockIdx.x;
    int ridx = 0;
    if (bt >= p.reg[0].blocks) {
        bt -= p.reg[0].blocks; ridx = 1;
        if (bt >= p.reg[1].blocks) { bt -= p.reg[1].blocks; ridx = 2; }
    }
    const TrRegion g = p.reg[ridx];
    const int TW = g.tw, TH = g.th, TN = g.tn;
    const int tix = bt % g.tiles_x; bt /= g.tiles_x;
    const int tiy = bt % g.tiles_y;
    const int tin = bt / g.tiles_y;
    const int qx0 = g.qx_base + tix * TW, qy0 = g.qy_base + tiy * TH, n0 = tin * TN;
    const int m0 = blockIdx.y * BM;

    const int PH = TH + 1, PW = TW + 1;
    const int IP = PH * PW;
    const int CP = TN * IP;
    const int HW = p.IH * p.IW;

    int poff[PPT];
#pragma unroll
    for (int s = 0; s < PPT; ++s) {
        const int e = tid + kBlock * s;
        int off = -1;
        if (e < CP) {
            const int pn = e / IP;
            const int rem = e - pn * IP;
            const int r = rem / PW;
            const int c = rem - r * PW;
            const int iy = qy0 - 1 + r, ix = qx0 - 1 + c;
            if (n0 + pn < p.N && iy >= 0 && iy < p.IH && ix >= 0 && ix < p.IW)
                off = pn * p.C * HW + iy * p.IW + ix;
        }
        poff[s] = off;
    }

    const int pp = wn * 32 + l31;
    const int pn = pp / (TW * TH);
    const int prem = pp - pn * (TW * TH);
    const int py = prem / TW;
    const int px = prem - py * TW;
    const bool lane_ok = pn < TN;
    const int pixbase = lane_ok ? pn * IP + py * PW + px : 0;
    constexpr int TLO[5] = {0, 6, 1, 3, 4};
    constexpr int THI[5] = {2, 8, 7, 5, 4};
    constexpr int GCLS[5] = {0, 0, 1, 2, 3};
    int boff[5], aoff[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) {
        const int t = half ? THI[g] : TLO[g];
        const int ky = t / 3, kx = t % 3;
        boff[g] = ((ky == 2) ? 0 : 1) * PW + ((kx == 2) ? 0 : 1);
        aoff[g] = (g == 4 && half) ? A_CELLS : t * 3 * BM + wm * MI * 32 + l31;
    }
    const int astep4 = half ? 0 : BM, mstep4 = half ? 0 : 32;   // the zero cell has no planes

    f32x16 acc[MI][4];
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int cl = 0; cl < 4; ++cl)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][cl][r] = 0.0f;

    const float* xb = x + (int64_t)n0 * p.C * HW;
    const u32x4* wt = wpb + (int64_t)blockIdx.y * (p.Cp / CK) * A_CELLS;
    float xv[CK][PPT];
    u32x4 av[APT];

    auto load_chunk = [&](int c0) {
#pragma unroll
        for (int ch = 0; ch < CK; ++ch) {
            const bool ch_ok = (c0 + ch) < p.C;
            const float* xc = xb + (int64_t)(c0 + ch) * HW;
#pragma unroll
            for (int s = 0; s < PPT; ++s) xv[ch][s] = (ch_ok && poff[s] >= 0) ? xc[poff[s]] : 0.0f;
        }
        const u32x4* wc = wt + (int64_t)(c0 / CK) * A_CELLS;
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + kBlock * i;
            if (e < A_CELLS) av[i] = wc[e];
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int s = 0; s < PPT; ++s) {
            const int e = tid + kBlock * s;
            if (e < CP) {
                float v[CK];
#pragma unroll
                for (int ch = 0; ch < CK; ++ch) v[ch] = xv[ch][s];
                bf16x8 s0, s1, s2;
                split3_bf16(v, s0, s1, s2);
                Xs[e] = __builtin_bit_cast(u32x4, s0);
                Xs[XCAP + e] = __builtin_bit_cast(u32x4, s1);
                Xs[2 * XCAP + e] = __builtin_bit_cast(u32x4, s2);
            }
        }
#pragma unroll
        for (int i = 0; i < APT; ++i) {
            const int e = tid + kBlock * i;
            if (e < A_CELLS) As[e] = av[i];
        }
    };

    if (tid == 0) As[A_CELLS] = u32x4{0u, 0u, 0u, 0u};
    load_chunk(0);
    for (int c0 = 0; c0 < p.Cp; c0 += CK) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (c0 + CK < p.Cp) load_chunk(c0 + CK);
#pragma unroll
        for (int g = 0; g < 5; ++g) {
            bf16x8 a[MI][3], b[3];
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) b[sp] = __builtin_bit_cast(bf16x8, Xs[sp * XCAP + pixbase + boff[g]]);
#pragma unroll
            for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                for (int sp = 0; sp < 3; ++sp)
                    a[mi][sp] = __builtin_bit_cast(
                        bf16x8, As[aoff[g] + ((g == 4) ? sp * astep4 + mi * mstep4 : sp * BM + mi * 32)]);
            constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
            constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
#pragma unroll
            for (int q = 0; q < 6; ++q)
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
                    acc[mi][GCLS[g]] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[mi][TA[q]], b[TB[q]],
                                                                               acc[mi][GCLS[g]], 0, 0, 0);
        }
    }

    const int n = n0 + pn, qy = qy0 + py, qx = qx0 + px;
    if (lane_ok && n < p.N && qy < g.QH && qx < g.QW) {
#pragma unroll
        for (int cl = 0; cl < 4; ++cl) {
            const int oy = 2 * qy + (cl >> 1) - p.pad;
            const int ox = 2 * qx + (cl & 1) - p.pad;
            if (oy >= 0 && oy < p.OH && ox >= 0 && ox < p.OW) {
                float* yb = y + ((int64_t)n * p.M * p.OH + oy) * p.OW + ox;
#pragma unroll
                for (int mi = 0; mi < MI; ++mi)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = m0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        if (m < p.M) yb[(int64_t)m * p.OH * p.OW] = acc[mi][cl][r];
                    }
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// wgrad: slab[slice][tap][a][b] = sum over the slice's pixels of  S[a][pix] * L[b][pix*stride + tap - pad]
//   S = gy (a = y-side channel m), L = x (b = x-side channel c)
// ------------------------------------------------------------------------------------------
struct WgradParams {
    int N, C, H, W;       // L tensor (x side)
    int M, OH, OW;        // S tensor (y side)
    int pad;
    int tw_log2, th_log2; // pixel chunk = TN x TH x TW = 64 pixels of the y side
    int tiles_x, tiles_y, tiles_n;
    int chunks, chunks_per_slice;
    int Ap, Bp;           // slab dims (padded M, C)
    // style modulation of either operand (null = none): L[n][c][..] * l_scale[n * C + c] (the input of a modulated
    // forward conv) or S[n][m][..] * s_scale[n * M + m] (the input of a modulated TRANSPOSED conv, which is the
    // y side of the forward-orientation problem), applied while the operand is written to LDS
    const float* l_scale;
    const float* s_scale;
    int zmask;            // always 0 (see IgemmParams)
    int xcd_order;        // 1: XCD-aware workgroup order (see conv_wgrad_kernel)
};

constexpr int kWgPix = 64;

template <int KS, int S>
struct WgPatchCap {
    static constexpr int value = (KS == 1) ? 65 : (S == 1 ? 145 : 325);   // odd strides
};

// MODE 0: the 4 waves tile the (a, b) plane WA x WB, every wave walks all 64 pixels of a chunk.
// MODE 1: narrow layers (M, C <= 32): ONE 32 x 32 x taps tile per workgroup; the 4 waves split the
//         pixels of a chunk (k-pairs w, w+4, ...) and are summed through LDS before the slab store,
//         so no wave multiplies padding.
// MODE 2: MODE 1 with the B-tile lanes enumerating (channel, tap) pairs (C * taps <= 32: the RGB
//         stems): one MFMA per k-pair instead of one per tap.
// MOD: an operand is style-modulated while staged (WgradParams::l_scale / s_scale); a separate instantiation
// WQ (stride 1, one image per 64-pixel chunk, rows of both tensors a multiple of four floats): both operands are staged as
// aligned 16-byte quads -- a wave loads the 64 pixels of FOUR gy channels with one dwordx4 instruction, and the widened
// x patch [x0 - 4, x0 + TW + 4) of one channel with one (3x3; 1x1: four channels) -- 20 (3x3) or 16 (1x1) vector-memory
// instructions per wave and chunk instead of 64.  At one wave per SIMD the issue of those instructions is not hidden by
// anything (profiles/r2_phase_clock_*.txt: 21 % of the kernel for 3x3, 54-61 % for 1x1).  LDS rows are 66 (S, 1x1 L) or
// 162 (3x3 L) floats: 8-byte aligned for ds_write_b64 and conflict-free across the 32 channels of an MFMA operand read.
// Stride 2 (3x3, pad 0): the x rows are 2^k + 1 floats, so the patch [2 x0, 2 x0 + 2 TW] is fetched as quads that are only
// 4-byte aligned (global_load_dwordx4 takes them, tools/probe/unaligned_x4_probe.hip) -- 85 quads per channel, two loads
// per channel and wave instead of six; a quad that would run past the end of its row is fetched from W - 4 and shifted,
// so nothing is read outside the tensor.  24 instead of 80 vector-memory instructions per wave and chunk.
template <int KS, int S, int TA, int TB, int WA, int WB, int MODE, bool MOD = false, bool WQ = false>
__global__ __launch_bounds__(kBlock) void conv_wgrad_kernel(const float* __restrict__ xl,
                                                            const float* __restrict__ gs,
                                                            float* __restrict__ slab, const WgradParams p) {
    constexpr bool PIXSPLIT = MODE == 1 || MODE == 2;
    constexpr bool PACKCT = MODE == 2;     // MODE 4 = MODE 0 with the operand double buffer forced on
    static_assert(PIXSPLIT ? (WA == 1 && WB == 1 && TA == 1 && TB == 1) : (WA * WB == 4), "wave arrangement");
    static_assert(!WQ || (!PIXSPLIT && !MOD && (S == 1 || KS == 3)), "quad staging: MODE 0 / 4, operands not modulated in the kernel");
    constexpr int T = KS * KS;
    constexpr int TT = PACKCT ? 1 : T;      // accumulator tap-tiles per (ta, tb)
    constexpr int BA = 32 * TA * WA, BB = 32 * TB * WB;
    constexpr int PK = kWgPix;
    constexpr int SLD = WQ ? PK + 2 : PK + 1;
    constexpr int LP = WQ ? (KS == 1 ? PK + 2 : (S == 1 ? 162 : 342)) : WgPatchCap<KS, S>::value;
    __shared__ float Ss[BA * SLD];
    __shared__ float Ls[BB * LP];
    __shared__ float red[PIXSPLIT ? 4 * 32 * 33 : 1];

    SAE_CLOCK_BEGIN
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    // the wave index is wave-uniform: say so, or every per-wave base address lives in VGPRs
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wa = PIXSPLIT ? 0 : wid / WB, wb = PIXSPLIT ? 0 : wid % WB;
    // XCD-aware order: the (a, b) tiles of one pixel slice read the same pixels; workgroup ids go round the 8 XCDs
    // (id % 8), so in launch order they sit behind 8 different L2s and each fetches its own copy.  Re-labelled so that
    // all tiles of a slice share an XCD and are dispatched together (needs slices % 8 == 0).
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order && (gridDim.z & 7) == 0) {
        const int mn = gridDim.x * gridDim.y;
        const int lin = bx + gridDim.x * (by + gridDim.y * bz);
        const int j = lin >> 3;
        const int tile = j % mn;
        bz = (j / mn) * 8 + (lin & 7);
        bx = tile % gridDim.x;
        by = tile / gridDim.x;
    }
    const int b0 = bx * BB, a0 = by * BA, slice = bz;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int TN = PK >> (p.tw_log2 + p.th_log2);
    const int PH = (KS == 1) ? TH : (TH - 1) * S + KS;
    const int PW = (KS == 1) ? TW : (TW - 1) * S + KS;
    const int RS = (WQ && KS == 3) ? (S == 1 ? TW + 8 : ((PW + 3) >> 2) << 2) : PW;      // WQ: patch rows widened to whole quads
    const int IP = PH * RS;
    const int CPs = TN * IP;
    const int HWl = p.H * p.W, HWs = p.OH * p.OW;

    int tapoff[T];
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int ky = t / KS, kx = t % KS;
        tapoff[t] = (KS == 1) ? 0 : ky * RS + kx;
    }

    f32x16 acc[TA][TB][TT];
#pragma unroll
    for (int ta = 0; ta < TA; ++ta)
#pragma unroll
        for (int tb = 0; tb < TB; ++tb)
#pragma unroll
            for (int t = 0; t < TT; ++t)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ta][tb][t][r] = 0.0f;

    // MODE 2: lane j of the B tile stands for (channel j / T, tap j % T)
    int pk_row = 0, pk_tapoff = 0;
    if (PACKCT) {
        const int ch = l31 / T, t = l31 - ch * T;
        const int ky = t / KS, kx = t % KS;
        pk_row = (ch < BB ? ch : BB - 1) * LP;
        pk_tapoff = (KS == 1) ? 0 : ky * RS + kx;
    }

    const int ch_begin = slice * p.chunks_per_slice;
    int ch_end = ch_begin + p.chunks_per_slice;
    if (ch_end > p.chunks) ch_end = p.chunks;

    // Register prefetch (software pipeline): the global loads of chunk k+1 are issued before the
    // MFMA loop of chunk k and written to LDS after the barrier that ends it, so HBM/L2 latency
    // hides under the matrix pipe even at one workgroup per CU.
    //   S: thread (wave w, lane l) holds pixel l of channels a = w, w+4, ...
    //   L: wave w holds channels b = w, w+4, ...; lane l holds patch elements l, l+64, ...
    constexpr int NS = BA / 4;
    constexpr int NLB = BB / 4;
    constexpr int LS = (KS == 1) ? 1 : (LP + kWave - 1) / kWave;
    float sv[WQ ? 1 : NS];
    float lv[WQ ? 1 : NLB][LS];
    // WQ: S quads: lane = (channel sub-index cs = lane / 16, quad q = lane % 16), load i covers channels 16 i + 4 wid + cs;
    // L quads (3x3): lane = quad of the widened patch, load j is channel wid + 4 j; (1x1): the S mapping
    constexpr int NSQ = BA / 16;
    constexpr int NLQ = (KS == 1) ? BB / 16 : (S == 1 ? BB / 4 : BB / 2);   // stride 2: two loads per channel
    f32x4 sq[WQ ? NSQ : 1];
    f32x4 lq[WQ ? NLQ : 1];
    bool sq_ok = false, lq_ok = false;
    [[maybe_unused]] bool lq_ok2 = false;          // stride 2: validity of the second quad of a channel
    [[maybe_unused]] int lq_sh[2] = {0, 0};        // ... and how far each was moved left to stay inside its row
    constexpr bool BRANCHFREE = KS == 3 && S == 2 && !PIXSPLIT;
    bool s_ok = false;      // validity of the prefetched chunk's elements (applied in store_chunk)
    int l_okmask = 0;
    [[maybe_unused]] int pf_n0 = 0;   // first image of the prefetched chunk (operand modulation)
    // factors of the prefetched chunk when it lies in one image (TN == 1: every layer with >= 64 pixels per image),
    // fetched with the chunk so their latency hides under the MFMAs like the operands' own
    [[maybe_unused]] float ssc[MOD ? NS : 1];
    [[maybe_unused]] float lsc[MOD ? NLB : 1];

    auto load_chunk = [&](int chunk) {
        int bt = chunk;
        const int tix = bt % p.tiles_x; bt /= p.tiles_x;
        const int tiy = bt % p.tiles_y;
        const int tin = bt / p.tiles_y;
        const int ox0 = tix * TW, oy0 = tiy * TH, n0 = tin * TN;
        if constexpr (MOD) {
            pf_n0 = n0;
            if (TN == 1) {
                const int z = lane & p.zmask;
                if (p.s_scale) {
#pragma unroll
                    for (int i = 0; i < NS; ++i) {
                        const int a = a0 + wid + 4 * i;
                        ssc[i] = p.s_scale[n0 * p.M + (a < p.M ? a : 0) + z];
                    }
                }
                if (p.l_scale) {
#pragma unroll
                    for (int j = 0; j < NLB; ++j) {
                        const int b = b0 + wid + 4 * j;
                        lsc[j] = p.l_scale[n0 * p.C + (b < p.C ? b : 0) + z];
                    }
                }
            }
        }
        if constexpr (WQ) {
            const int q = lane & 15, cs = lane >> 4;
            const int py = (4 * q) >> p.tw_log2, px = (4 * q) & (TW - 1);
            {   // S: gy[n0][a0 + a][oy0 + py][ox0 + px .. + 3]
                const int oy = oy0 + py, ox = ox0 + px;
                sq_ok = oy < p.OH && ox < p.OW;
                const char* sbase = reinterpret_cast<const char*>(gs + ((int64_t)n0 * p.M + a0) * HWs);
                const unsigned pix = sq_ok ? (unsigned)(oy * p.OW + ox) : 0u;
#pragma unroll
                for (int i = 0; i < NSQ; ++i) {
                    const int a = 16 * i + 4 * wid + cs;
                    const unsigned off = (a0 + a < p.M) ? 4u * ((unsigned)(a * HWs) + pix) : 0u;
                    sq[i] = *reinterpret_cast<const f32x4*>(sbase + off);
                }
            }
            const char* lbase = reinterpret_cast<const char*>(xl + ((int64_t)n0 * p.C + b0) * HWl);
            if constexpr (KS == 1) {   // L: x at the same pixels (1x1, stride 1, pad 0)
                const int iy = oy0 + py - p.pad, ix = ox0 + px - p.pad;
                lq_ok = iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const unsigned pix = lq_ok ? (unsigned)(iy * p.W + ix) : 0u;
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int b = 16 * j + 4 * wid + cs;
                    const unsigned off = (b0 + b < p.C) ? 4u * ((unsigned)(b * HWl) + pix) : 0u;
                    lq[j] = *reinterpret_cast<const f32x4*>(lbase + off);
                }
            } else if constexpr (S == 2) {   // L: quads lane, lane + 64 of the patch of channel wid + 4 j (pad 0)
                const int RQ = RS >> 2;
                unsigned pix[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int qi = lane + 64 * h;
                    const int r = qi / RQ, qc = qi - r * RQ;
                    const int iy = oy0 * 2 + r, ix = ox0 * 2 + 4 * qc;
                    const bool ok = r < PH && iy < p.H && ix < p.W;
                    const int ixc = ix + 4 <= p.W ? ix : p.W - 4;       // keep the quad inside its row
                    lq_sh[h] = ok ? ix - ixc : 0;
                    pix[h] = ok ? (unsigned)(iy * p.W + ixc) : 0u;
                    if (h == 0) lq_ok = ok; else lq_ok2 = ok;
                }
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int b = wid + 4 * (j >> 1);
                    const unsigned off = (b0 + b < p.C) ? 4u * ((unsigned)(b * HWl) + pix[j & 1]) : 0u;
                    struct __attribute__((packed, aligned(4))) U4 { f32x4 v; };     // 4-byte aligned quad
                    lq[j] = reinterpret_cast<const U4*>(lbase + off)->v;
                }
            } else {                   // L: quad `lane` of the widened patch of channel wid + 4 j
                const int RQ = RS >> 2;
                const int r = lane / RQ, qc = lane - r * RQ;
                const int iy = oy0 - p.pad + r, ix = ox0 - 4 + 4 * qc;
                lq_ok = r < PH && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
                const unsigned pix = lq_ok ? (unsigned)(iy * p.W + ix) : 0u;
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int b = wid + 4 * j;
                    const unsigned off = (b0 + b < p.C) ? 4u * ((unsigned)(b * HWl) + pix) : 0u;
                    lq[j] = *reinterpret_cast<const f32x4*>(lbase + off);
                }
            }
        } else {
        // addressing: one wave-uniform 64-bit base per tensor + 32-bit (lane + channel) offsets,
        // so loads use the SGPR-base form and no per-channel pointer is kept in registers
        {
            const int px = lane & (TW - 1);
            const int py = (lane >> p.tw_log2) & (TH - 1);
            const int pn = lane >> (p.tw_log2 + p.th_log2);
            const int n = n0 + pn, oy = oy0 + py, ox = ox0 + px;
            const bool ok = n < p.N && oy < p.OH && ox < p.OW;
            const float* sbase = gs + ((int64_t)n0 * p.M + a0) * HWs;
            const int lane_off = pn * p.M * HWs + oy * p.OW + ox;
            // BRANCHFREE: no load sits in a divergent branch (hipcc drains vmcnt at the join, which serialised
            // the loads of the register-starved stride-2 instantiation: 78 -> 98 TFLOP/s); invalid elements
            // read element 0 of the block's tile and are zeroed in store_chunk.  The stride-1 kernel keeps
            // predicated loads (measured 111 vs 102 TFLOP/s with the select-and-mask form).
            s_ok = ok;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int a = wid + 4 * i;
                if constexpr (BRANCHFREE) sv[i] = sbase[(ok && a0 + a < p.M) ? lane_off + a * HWs : 0];
                else sv[i] = (ok && a0 + a < p.M) ? sbase[lane_off + a * HWs] : 0.0f;
            }
        }
        {
            int poff[LS];
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const int e = lane + kWave * s;
                int off = -1;
                if (e < CPs) {
                    const int pn = e / IP;
                    const int rem = e - pn * IP;
                    const int r = rem / RS;
                    const int cl = rem - r * RS;
                    int c = cl;
                    // natural column order: lanes of a wgrad B-read differ in CHANNEL, not pixel, so the
                    // stride-2 pixel walk causes no bank conflict and the global loads stay contiguous
                    int iy, ix;
                    if (KS == 1) { iy = (oy0 + r) * S - p.pad; ix = (ox0 + c) * S - p.pad; }
                    else { iy = oy0 * S - p.pad + r; ix = ox0 * S - p.pad + c; }
                    if (n0 + pn < p.N && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W)
                        off = pn * p.C * HWl + iy * p.W + ix;
                }
                poff[s] = off;
            }
            const float* lbase = xl + ((int64_t)n0 * p.C + b0) * HWl;
            l_okmask = 0;
#pragma unroll
            for (int s = 0; s < LS; ++s) l_okmask |= (poff[s] >= 0 ? 1 : 0) << s;
#pragma unroll
            for (int j = 0; j < NLB; ++j) {
                const int b = wid + 4 * j;
                const bool ch_ok = (b0 + b) < p.C;
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    if constexpr (BRANCHFREE) lv[j][s] = lbase[(ch_ok && poff[s] >= 0) ? poff[s] + b * HWl : 0];
                    else lv[j][s] = (ch_ok && poff[s] >= 0) ? lbase[poff[s] + b * HWl] : 0.0f;
                }
            }
        }
        }
    };
    auto store_chunk = [&]() {
        if constexpr (WQ) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            const int q = lane & 15, cs = lane >> 4;
#pragma unroll
            for (int i = 0; i < NSQ; ++i) {
                const int a = 16 * i + 4 * wid + cs;
                f32x4 v = sq[i];
                const bool ok = sq_ok && a0 + a < p.M;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
                float* dst = Ss + a * SLD + 4 * q;
                *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
                *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
            }
            if constexpr (KS == 3 && S == 2) {
#pragma unroll
                for (int j = 0; j < NLQ; ++j) {
                    const int b = wid + 4 * (j >> 1);
                    const int sh = lq_sh[j & 1];
                    const f32x4 u = lq[j];
                    // a quad fetched `sh` floats to the left of its place: drop the first sh, the tail lies beyond the row
                    f32x4 v = u;
                    if (sh == 1) v = f32x4{u[1], u[2], u[3], 0.0f};
                    if (sh == 2) v = f32x4{u[2], u[3], 0.0f, 0.0f};
                    if (sh == 3) v = f32x4{u[3], 0.0f, 0.0f, 0.0f};
                    const bool ok = ((j & 1) ? lq_ok2 : lq_ok) && b0 + b < p.C;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
                    const int slot = 4 * (lane + 64 * (j & 1));
                    if (slot + 3 < LP) {
                        float* dst = Ls + b * LP + slot;
                        *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
                        *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
                    }
                }
                return;
            }
#pragma unroll
            for (int j = 0; j < NLQ; ++j) {
                const int b = (KS == 1) ? 16 * j + 4 * wid + cs : wid + 4 * j;
                f32x4 v = lq[j];
                const bool ok = lq_ok && b0 + b < p.C;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
                // 3x3: lanes beyond the patch (r >= PH) have lq_ok false and still own a distinct slot < 64 quads = 256
                // floats: keep them inside the row by folding onto the last quad slots of the row's spare space
                const int slot = (KS == 1) ? 4 * q : 4 * lane;
                if (KS == 1 || 4 * lane + 3 < LP) {
                    float* dst = Ls + b * LP + slot;
                    *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
                    *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
                }
            }
            return;
        }
        if (MOD && p.s_scale) {        // modulated S operand: factor of (image of this lane's pixel, channel a)
            if (TN == 1) {
#pragma unroll
                for (int i = 0; i < NS; ++i) sv[i] *= ssc[i];
            } else {
                const int n = pf_n0 + (lane >> (p.tw_log2 + p.th_log2));
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int a = a0 + wid + 4 * i;
                    sv[i] *= p.s_scale[(n < p.N ? n : 0) * p.M + (a < p.M ? a : 0)];
                }
            }
        }
        if (MOD && p.l_scale) {        // modulated L operand: factor of (image of the patch element, channel b)
            if (TN == 1) {
#pragma unroll
                for (int j = 0; j < NLB; ++j)
#pragma unroll
                    for (int s = 0; s < LS; ++s) lv[j][s] *= lsc[j];
            } else {
#pragma unroll
                for (int s = 0; s < LS; ++s) {
                    const int e = lane + kWave * s;
                    const int n = pf_n0 + (e < CPs ? e / IP : 0);
                    const int row = (n < p.N ? n : 0) * p.C;
#pragma unroll
                    for (int j = 0; j < NLB; ++j) {
                        const int b = b0 + wid + 4 * j;
                        lv[j][s] *= p.l_scale[row + (b < p.C ? b : 0)];
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
            Ss[(wid + 4 * i) * SLD + lane] = (!BRANCHFREE || (s_ok && a0 + wid + 4 * i < p.M)) ? sv[i] : 0.0f;
#pragma unroll
        for (int j = 0; j < NLB; ++j) {
            const bool ch_ok = (b0 + wid + 4 * j) < p.C;
#pragma unroll
            for (int s = 0; s < LS; ++s) {
                const int e = lane + kWave * s;
                if (e < CPs)
                    Ls[(wid + 4 * j) * LP + e] = (!BRANCHFREE || (ch_ok && ((l_okmask >> s) & 1))) ? lv[j][s] : 0.0f;
            }
        }
    };

    if (ch_begin < ch_end) load_chunk(ch_begin);
    SAE_CLOCK_PHASE(0)
    for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
        __syncthreads();   // previous chunk fully consumed
        SAE_CLOCK_PHASE(2)
        store_chunk();
        SAE_CLOCK_PHASE(3)
        __syncthreads();
        SAE_CLOCK_PHASE(4)
        if (chunk + 1 < ch_end) load_chunk(chunk + 1);   // in flight under the MFMAs below
        SAE_CLOCK_PHASE(5)

        // ---- MFMA over the 64 pixels, two per instruction.  The LDS operands of k-pair i+1 are
        // fetched into a second register set before the MFMAs of k-pair i issue (explicit
        // double buffering): with one wave per SIMD nothing else hides the ds_read latency.
        constexpr int KSTEP = PIXSPLIT ? 4 : 1;
        constexpr int KITER = (PK / 2) / KSTEP;           // k-pairs per wave per chunk (even)
        static_assert(KITER % 2 == 0, "k-pair loop is unrolled by two");
        constexpr int NB = PACKCT ? 1 : TB * T;
        auto fetch = [&](int kp, float (&a)[TA], float (&b)[NB]) {
            const int pk = 2 * kp + half;
            const int px = pk & (TW - 1);
            const int py = (pk >> p.tw_log2) & (TH - 1);
            const int pn = pk >> (p.tw_log2 + p.th_log2);
            const int pbase = pn * IP + ((KS == 1) ? py * RS + px : py * S * RS + px * S) + ((WQ && KS == 3 && S == 1) ? 4 - p.pad : 0);
#pragma unroll
            for (int ta = 0; ta < TA; ++ta) a[ta] = Ss[((wa * TA + ta) * 32 + l31) * SLD + pk];
            if constexpr (PACKCT) {
                b[0] = Ls[pk_row + pbase + pk_tapoff];
            } else {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb) {
                    const float* lrow = Ls + ((wb * TB + tb) * 32 + l31) * LP + pbase;
#pragma unroll
                    for (int t = 0; t < T; ++t) b[tb * T + t] = lrow[tapoff[t]];
                }
            }
        };
        auto mma = [&](const float (&a)[TA], const float (&b)[NB]) {
            if constexpr (PACKCT) {
                acc[0][0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[0], b[0], acc[0][0][0], 0, 0, 0);
            } else {
#pragma unroll
                for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                    for (int t = 0; t < T; ++t)
#pragma unroll
                        for (int ta = 0; ta < TA; ++ta)
                            acc[ta][tb][t] =
                                __builtin_amdgcn_mfma_f32_32x32x2f32(a[ta], b[tb * T + t], acc[ta][tb][t], 0, 0, 0);
            }
        };
        const int kp0 = PIXSPLIT ? wid : 0;
        // the stride-2 instantiation already sits at the register ceiling (80 prefetch + 144
        // accumulator registers): it keeps a single operand set
        constexpr bool DOUBLE_BUFFER = (MODE == 4) || !(KS == 3 && S == 2 && MODE == 0);
        if constexpr (DOUBLE_BUFFER) {
            float a_even[TA], b_even[NB], a_odd[TA], b_odd[NB];
            fetch(kp0, a_even, b_even);
            for (int it = 0; it < KITER; it += 2) {
                const int kp = kp0 + it * KSTEP;
                fetch(kp + KSTEP, a_odd, b_odd);
                mma(a_even, b_even);
                if (it + 2 < KITER) fetch(kp + 2 * KSTEP, a_even, b_even);
                mma(a_odd, b_odd);
            }
        } else {
#pragma unroll 2
            for (int it = 0; it < KITER; ++it) {
                float a_cur[TA], b_cur[NB];
                fetch(kp0 + it * KSTEP, a_cur, b_cur);
                mma(a_cur, b_cur);
            }
        }
        SAE_CLOCK_PHASE(1)
    }

    // ---- slab store: rows = a (m), cols = b (c)
    if constexpr (PIXSPLIT) {
        // sum the four waves' partial tiles through LDS (fixed order), one tap-tile at a time
#pragma unroll
        for (int t = 0; t < TT; ++t) {
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r)
                red[(wid * 32 + (r & 3) + 8 * (r >> 2) + 4 * half) * 33 + l31] = acc[0][0][t][r];
            __syncthreads();
            for (int e = tid; e < 32 * 32; e += kBlock) {
                const int i = e >> 5, j = e & 31;
                const float v = (red[(0 * 32 + i) * 33 + j] + red[(1 * 32 + i) * 33 + j]) +
                                (red[(2 * 32 + i) * 33 + j] + red[(3 * 32 + i) * 33 + j]);
                int tap = t, bcol = b0 + j;
                if (PACKCT) { bcol = j / T; tap = j - bcol * T; }
                if (bcol < p.Bp && (!PACKCT || bcol < p.C))
                    slab[(((int64_t)slice * T + tap) * p.Ap + a0 + i) * p.Bp + bcol] = v;
            }
        }
    } else {
#pragma unroll
        for (int ta = 0; ta < TA; ++ta)
#pragma unroll
            for (int tb = 0; tb < TB; ++tb)
#pragma unroll
                for (int t = 0; t < T; ++t) {
                    const int bcol = b0 + (wb * TB + tb) * 32 + l31;
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int arow = a0 + (wa * TA + ta) * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
                        slab[(((int64_t)slice * T + t) * p.Ap + arow) * p.Bp + bcol] = acc[ta][tb][t][r];
                    }
                }
    }
    SAE_CLOCK_PHASE(6)
    SAE_CLOCK_END
}

// ------------------------------------------------------------------------------------------
// wgrad, second generation ("wg16"): 64 a x 32 b x 9 taps per workgroup on v_mfma_f32_16x16x4_f32, two or three
// workgroups per CU.
//
// conv_wgrad_kernel<3, S, ..., WQ> gives every wave a 32 x 32 x 9-tap tile: 144 accumulator registers + the chunk's
// prefetch registers = one wave per SIMD, and nothing overlaps its staging phases (profiles/r2_phase_clock_*.txt: 70 % of
// the wave's time in the MFMA loop; 0.73 of peak in the step).  Here a wave owns 32 a x 16 b x 9 taps as 2 x 9 tiles of the
// 16 x 16 x 4 instruction (same FLOP rate, 4 accumulator registers per tile): 72 accumulator registers, 36 (stride 2: 60)
// prefetch registers, so two independent workgroups share a CU and one's MFMAs cover the other's staging, as in the gather.
//  * operands: lane l supplies A[a = l & 15][k = l >> 4] and B[k][b = l & 15], k = four consecutive pixels of the chunk; the
//    32 lanes of an LDS access group read 16 channel rows x 2 adjacent pixels.  Row pitches 66 (S) and 162 (L, stride 1) are
//    = 2 (mod 32): banks 2 r + {0, 1}, conflict-free (the 32 x 32 form read 32 rows per group: r and r + 16 collided,
//    SQ_LDS_BANK_CONFLICT = 48 % of the LDS cycles, profiles/r3_pmc_f32.txt).  Stride 2 (pitch 342, pixel offsets even)
//    keeps two-way conflicts on its L reads.
//  * staging: S as in the WQ path (a wave loads the 64 pixels of four gy channels per dwordx4); the L patch quads of the 32
//    channels are dealt out over the workgroup (quad j = tid + 256 k: channel j / QN, quad j % QN), 5 (stride 2: 11) per
//    thread -- 9 (15) vector-memory instructions per thread and chunk.  Stride 2 fetches 4-byte aligned quads and moves a
//    quad that would cross its row's end to W - 4, as the WQ path does.
//  * the k of an instruction sums 4 pixels, so the summation ORDER differs from conv_wgrad_kernel (not bit-identical to it;
//    both are checked against the oracle).  Slab layout and the fixed-order reduction are unchanged: deterministic.
// Taken for 3 x 3 layers with one image per 64-pixel chunk under the WQ path's conditions (host: conv_wgrad_impl).
// ------------------------------------------------------------------------------------------
// (measured and dropped: __launch_bounds__(kBlock, 3) -- 168 registers with 27 spilled dwords, 83 - 102 instead of 130 TFLOP/s;
// stride 2 with the L operand read as conflict-free 8-byte pairs on a pitch of 388 floats: 109.8 vs 111.4, the two-way
// conflicts of its dword reads are not what bounds it)
template <int S>
__global__ __launch_bounds__(kBlock, 2) void conv_wgrad16_kernel(const float* __restrict__ xl,
                                                                 const float* __restrict__ gs,
                                                                 float* __restrict__ slab, const WgradParams p) {
    constexpr int T = 9, BA = 64, BB = 32, PK = kWgPix;
    constexpr int SLD = PK + 2;
    constexpr int LP = (S == 1) ? 162 : 342;
    constexpr int QCAP = (S == 1) ? 40 : 85;                       // quads per channel of the widened patch (host-checked)
    constexpr int QPT = (BB * QCAP + kBlock - 1) / kBlock;          // L quad slots per thread and chunk
    constexpr int NSQ = BA / 16;
    constexpr int NB = T;                                           // operand registers of the L side per step
    __shared__ __attribute__((aligned(16))) float Ss[BA * SLD];
    __shared__ __attribute__((aligned(16))) float Ls[BB * LP + 4 * kBlock];   // + a dump area for the slots beyond the patch
    typedef float f32x2 __attribute__((ext_vector_type(2)));

    SAE_CLOCK_BEGIN
    const int tid = threadIdx.x;
    __builtin_assume(tid < kBlock);
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, k4 = lane >> 4;
    const int wa = wid >> 1, wb = wid & 1;
    // XCD-aware order: all (a, b) tiles of a pixel slice behind one L2 (see conv_wgrad_kernel)
    int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
    if (p.xcd_order && (gridDim.z & 7) == 0) {
        const int mn = gridDim.x * gridDim.y;
        const int lin = bx + gridDim.x * (by + gridDim.y * bz);
        const int j = lin >> 3;
        const int tile = j % mn;
        bz = (j / mn) * 8 + (lin & 7);
        bx = tile % gridDim.x;
        by = tile / gridDim.x;
    }
    const int b0 = bx * BB, a0 = by * BA, slice = bz;

    const int TW = 1 << p.tw_log2, TH = 1 << p.th_log2;
    const int PH = (TH - 1) * S + 3;
    const int PW = (TW - 1) * S + 3;
    const int RS = (S == 1) ? TW + 8 : ((PW + 3) >> 2) << 2;      // patch rows widened to whole quads
    const int RQ = RS >> 2, QN = PH * RQ;                          // quads per row, per channel (<= QCAP)
    const int HWl = p.H * p.W, HWs = p.OH * p.OW;

    f32x4 acc[2][T];
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int t = 0; t < T; ++t) acc[ta][t] = f32x4{0.0f, 0.0f, 0.0f, 0.0f};

    // chunk-invariant part of the L quad slots: patch row, first column relative to the tile's origin, byte offset relative
    // to (image n0, channel b0, row oy0 * S - pad, column ox0 * S), LDS index.  lrc < 0: the slot never holds data.
    int lrc[QPT], lbyte[QPT], ldst[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int j = tid + kBlock * k;
        const int ch = j / QN;
        const int q = j - ch * QN;
        const int r = q / RQ;
        const int qc = q - r * RQ;
        const int col = (S == 1) ? 4 * qc - 4 : 4 * qc;
        const bool slot = ch < BB;
        lrc[k] = (slot && b0 + ch < p.C) ? (r << 16) | (col + 4) : -1;
        lbyte[k] = 4 * (ch * HWl + r * p.W + col);
        ldst[k] = slot ? ch * LP + 4 * q : BB * LP + 4 * tid;
    }
    // S quads: lane = (channel sub-index cs = lane / 16, quad q = lane % 16), load i covers channel 16 i + 4 wid + cs
    const int sq_q = lane & 15, sq_cs = lane >> 4;
    const int sq_py = (4 * sq_q) >> p.tw_log2, sq_px = (4 * sq_q) & (TW - 1);

    const int ch_begin = slice * p.chunks_per_slice;
    int ch_end = ch_begin + p.chunks_per_slice;
    if (ch_end > p.chunks) ch_end = p.chunks;

    f32x4 sq[NSQ], lq[QPT];
    bool sq_ok = false;
    unsigned lq_ok = 0;                     // bit k: slot k of the prefetched chunk holds image data
    [[maybe_unused]] unsigned lq_sh = 0;    // stride 2, two bits per slot: how far the quad was moved left to stay inside its row

    auto load_chunk = [&](int chunk) {
        int bt = chunk;
        const int tix = bt % p.tiles_x; bt /= p.tiles_x;
        const int tiy = bt % p.tiles_y;
        const int n0 = bt / p.tiles_y;
        const int ox0 = tix * TW, oy0 = tiy * TH;
        {
            const int oy = oy0 + sq_py, ox = ox0 + sq_px;
            sq_ok = oy < p.OH && ox < p.OW;
            const char* sbase = reinterpret_cast<const char*>(gs + ((int64_t)n0 * p.M + a0) * HWs);
            const unsigned pix = sq_ok ? (unsigned)(oy * p.OW + ox) : 0u;
#pragma unroll
            for (int i = 0; i < NSQ; ++i) {
                const int a = 16 * i + 4 * wid + sq_cs;
                const unsigned off = (a0 + a < p.M) ? 4u * ((unsigned)(a * HWs) + pix) : 0u;
                sq[i] = *reinterpret_cast<const f32x4*>(sbase + off);
            }
        }
        const char* lbase = reinterpret_cast<const char*>(xl + ((int64_t)n0 * p.C + b0) * HWl);
        const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S;
        const int rel = 4 * (iy0 * p.W + ix0);
        lq_ok = 0;
        if constexpr (S == 2) lq_sh = 0;
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            const int iy = iy0 + (lrc[k] >> 16), ix = ix0 + (lrc[k] & 0xffff) - 4;
            const bool ok = lrc[k] >= 0 && iy >= 0 && iy < p.H && ix >= 0 && ix < p.W;
            int off = lbyte[k] + rel;
            if constexpr (S == 2) {
                const int sh = (ok && ix + 4 > p.W) ? ix + 4 - p.W : 0;       // keep the quad inside its row
                off -= 4 * sh;
                lq_sh |= (unsigned)sh << (2 * k);
                struct __attribute__((packed, aligned(4))) U4 { f32x4 v; };     // 4-byte aligned quad
                lq[k] = reinterpret_cast<const U4*>(lbase + (ok ? off : 0))->v;
            } else {
                lq[k] = *reinterpret_cast<const f32x4*>(lbase + (ok ? off : 0));
            }
            lq_ok |= (ok ? 1u : 0u) << k;
        }
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < NSQ; ++i) {
            const int a = 16 * i + 4 * wid + sq_cs;
            f32x4 v = sq[i];
            const bool ok = sq_ok && a0 + a < p.M;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
            float* dst = Ss + a * SLD + 4 * sq_q;
            *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
        }
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            f32x4 v = lq[k];
            if constexpr (S == 2) {
                const int sh = (lq_sh >> (2 * k)) & 3;     // fetched sh floats to the left of its place: the tail lies beyond the row
                const f32x4 u = v;
                if (sh == 1) v = f32x4{u[1], u[2], u[3], 0.0f};
                if (sh == 2) v = f32x4{u[2], u[3], 0.0f, 0.0f};
                if (sh == 3) v = f32x4{u[3], 0.0f, 0.0f, 0.0f};
            }
            const bool ok = (lq_ok >> k) & 1u;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = ok ? v[e] : 0.0f;
            float* dst = Ls + ldst[k];
            *reinterpret_cast<f32x2*>(dst) = f32x2{v[0], v[1]};
            *reinterpret_cast<f32x2*>(dst + 2) = f32x2{v[2], v[3]};
        }
    };

    const float* const srow = Ss + (wa * 32 + l15) * SLD + k4;
    const float* const lrow = Ls + (wb * 16 + l15) * LP + ((S == 1) ? 4 - p.pad : 0);
    if (ch_begin < ch_end) load_chunk(ch_begin);
    SAE_CLOCK_PHASE(0)
    for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
        __syncthreads();   // previous chunk fully consumed
        SAE_CLOCK_PHASE(2)
        store_chunk();
        SAE_CLOCK_PHASE(3)
        __syncthreads();
        SAE_CLOCK_PHASE(4)
        if (chunk + 1 < ch_end) load_chunk(chunk + 1);   // in flight under the MFMAs below
        SAE_CLOCK_PHASE(5)
        // 16 steps of 4 pixels: 2 + 9 operand reads, 18 MFMAs; the operands of step i + 1 are read before the MFMAs of step i
        auto fetch = [&](int step, float (&a)[2], float (&b)[NB]) {
            const int pk = 4 * step + k4;
            const int px = pk & (TW - 1), py = pk >> p.tw_log2;
            const float* lp = lrow + py * S * RS + px * S;
            a[0] = srow[4 * step];
            a[1] = srow[16 * SLD + 4 * step];
#pragma unroll
            for (int t = 0; t < T; ++t) b[t] = lp[(t / 3) * RS + (t % 3)];
        };
        auto mma = [&](const float (&a)[2], const float (&b)[NB]) {
#pragma unroll
            for (int t = 0; t < T; ++t)
#pragma unroll
                for (int ta = 0; ta < 2; ++ta)
                    acc[ta][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[ta], b[t], acc[ta][t], 0, 0, 0);
        };
        float a_even[2], b_even[NB], a_odd[2], b_odd[NB];
        fetch(0, a_even, b_even);
        for (int it = 0; it < PK / 4; it += 2) {
            fetch(it + 1, a_odd, b_odd);
            mma(a_even, b_even);
            if (it + 2 < PK / 4) fetch(it + 2, a_even, b_even);
            mma(a_odd, b_odd);
        }
        SAE_CLOCK_PHASE(1)
    }

    // ---- slab store: D row = 4 * (lane >> 4) + r (a), column = lane & 15 (b)
    const int bcol = b0 + wb * 16 + l15;
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int t = 0; t < T; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int arow = a0 + wa * 32 + ta * 16 + 4 * k4 + r;
                slab[(((int64_t)slice * T + t) * p.Ap + arow) * p.Bp + bcol] = acc[ta][t][r];
            }
    SAE_CLOCK_PHASE(6)
    SAE_CLOCK_END
}

// ------------------------------------------------------------------------------------------
// "bx" arithmetic of the 3x3 wgrad (see conv_igemm_bx_kernel for the split).  K = pixels: the 16 k
// of an MFMA are two OCTETS, an octet = 8 consecutive pixels of one row of the y side, held as one
// 16-byte cell per (split, channel).  S (gy) cells are staged as they lie; the x-side operand of tap
// (ky, kx) is the octet shifted by kx columns: for stride 1 the lane reads the aligned cell plus the
// first dword of its right neighbour and forms kx = 1 by a 16-bit funnel shift (v_alignbit) and
// kx = 2 by dropping a dword; for stride 2 the patch columns are stored de-interleaved (even cells,
// odd cells) so kx = 0 / 1 / 2 = even / odd / even shifted by one element.
// One workgroup per CU (accumulators: 9 tap tiles = 144 AGPRs per wave), register prefetch of the
// next chunk; operand splitting happens when the prefetched fp32 values are written to LDS.
// ------------------------------------------------------------------------------------------
struct WgBxParams {
    int N, C, H, W;       // L tensor (x side)
    int M, OH, OW;        // S tensor (y side), OW % 8 == 0
    int pad;
    int tw8_log2, th_log2; // y-side tile: TH rows x (8 << tw8_log2) columns x TN images = 8 * NO pixels
    int tiles_x, tiles_y, tiles_n;
    int chunks, chunks_per_slice;
    int Ap, Bp;
};

typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // 4-byte aligned 16-byte global access

template <int S>
struct WgBx {
    static constexpr int NO = (S == 1) ? 16 : 8;      // octets per chunk
    static constexpr int SOS = NO + 1;                // odd cell strides: conflict-free 16-byte reads across channels
    static constexpr int LCAP = (S == 1) ? 30 : 45;   // x-side cells per channel
    static constexpr int LOS = (S == 1) ? 31 : 45;
    static constexpr int UCAP = (S == 1) ? 30 : 27;   // staging units per channel (s2: a unit = 16 columns = even + odd cell)
};

template <int S, int WA, int WB>
__global__ __launch_bounds__(kBlock) void conv_wgrad_bx_kernel(const float* __restrict__ xl,
                                                               const float* __restrict__ gs,
                                                               float* __restrict__ slab, const WgBxParams p) {
    static_assert(WA * WB == 4, "4 waves per workgroup");
    constexpr int T = 9;
    constexpr int BA = 32 * WA, BB = 32 * WB;
    constexpr int NO = WgBx<S>::NO, SOS = WgBx<S>::SOS, LOS = WgBx<S>::LOS;
    constexpr int SPT = BA * NO / kBlock;
    constexpr int LPT = (BB * WgBx<S>::UCAP + kBlock - 1) / kBlock;
    constexpr int UW = (S == 1) ? 8 : 16;              // columns per staging unit
    static_assert(BA * NO % kBlock == 0, "S cells per thread");
    __shared__ u32x4 Ss[3 * BA * SOS];
    __shared__ u32x4 Ls[3 * BB * LOS];

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, half = lane >> 5;
    const int wa = wid / WB, wb = wid % WB;
    const int b0 = blockIdx.x * BB, a0 = blockIdx.y * BA, slice = blockIdx.z;

    const int TWO = 1 << p.tw8_log2, TH = 1 << p.th_log2;
    const int TN = NO >> (p.tw8_log2 + p.th_log2);
    const int PH = (TH - 1) * S + 3;
    const int NOLE = TWO + 1;                           // stride 1: cells per patch row; stride 2: even cells per row
    const int RC = (S == 1) ? NOLE : 2 * TWO + 1;       // cells per patch row
    const int UPC = TN * PH * NOLE;                     // staging units per channel
    const int HWl = p.H * p.W, HWs = p.OH * p.OW;

    // chunk-independent part of the staging maps (kept packed: these live across the whole kernel)
    //   S: cell e = tid + 256 i -> channel a = e / NO, octet o = e % NO; 256 % NO == 0, so a thread's
    //      cells share the octet (same pixel position) and differ by 256 / NO channels
    constexpr int SA = kBlock / NO;
    const int s_o = tid % NO, s_a = tid / NO;
    const int s_ox = 8 * (s_o & (TWO - 1));
    const int s_oy = (s_o >> p.tw8_log2) & (TH - 1);
    const int s_n = s_o >> (p.tw8_log2 + p.th_log2);
    const int s_off = (s_n * p.M + s_a) * HWs + s_oy * p.OW + s_ox;
    const int s_cell = s_a * SOS + s_o;
    //   L: unit u = tid + 256 i -> (channel, image, patch row, unit column); meta = cell | row << 12 |
    //      column << 17 | image << 24 | single-column unit << 30
    int l_off[LPT], l_meta[LPT];
#pragma unroll
    for (int i = 0; i < LPT; ++i) {
        const int u = tid + kBlock * i;
        const int ub = u / UPC, ur = u - ub * UPC;
        const int j = ur % NOLE;
        const int rr = ur / NOLE;
        const int r = rr % PH, pn = rr / PH;
        const bool ok = ub < BB && b0 + ub < p.C;
        l_off[i] = ok ? (pn * p.C + ub) * HWl + r * p.W + UW * j : -1;
        const int cell = (ub < BB) ? ub * LOS + (pn * PH + r) * RC + j : 0xfff;   // 0xfff: writes nothing
        l_meta[i] = cell | (r << 12) | ((UW * j) << 17) | (pn << 24) | ((S == 2 && j == TWO) ? (1 << 30) : 0);
    }

    f32x16 acc[T];
#pragma unroll
    for (int t = 0; t < T; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int ch_begin = slice * p.chunks_per_slice;
    int ch_end = ch_begin + p.chunks_per_slice;
    if (ch_end > p.chunks) ch_end = p.chunks;

    float sv[SPT][8];
    float lv[LPT][UW];      // raw windows as loaded; the border fix-up waits until store_chunk so that
    int lcode[LPT];         // nothing in load_chunk depends on a load (its latency hides under the MFMAs)
    int s_okmask = 0;
    bool chunk_border = false;   // workgroup-uniform: the prefetched chunk needs the border fix-up
    float ltail[(S == 2) ? LPT : 1];

    auto load_chunk = [&](int chunk) {
        int bt = chunk;
        const int tix = bt % p.tiles_x; bt /= p.tiles_x;
        const int tiy = bt % p.tiles_y;
        const int tin = bt / p.tiles_y;
        const int ox0 = tix * 8 * TWO, oy0 = tiy * TH, n0 = tin * TN;
        chunk_border = ox0 == 0 || ox0 + 8 * TWO >= p.OW || oy0 == 0 || oy0 + TH >= p.OH || n0 + TN > p.N ||
                       a0 + BA > p.M || b0 + BB > p.C;
        {
            const float* sbase = gs + ((int64_t)n0 * p.M + a0) * HWs + oy0 * p.OW + ox0;
            const bool pos_ok = n0 + s_n < p.N && oy0 + s_oy < p.OH && ox0 + s_ox < p.OW;
            s_okmask = 0;
#pragma unroll
            for (int i = 0; i < SPT; ++i) {
                const bool ok = pos_ok && a0 + s_a + SA * i < p.M;
                s_okmask |= (ok ? 1 : 0) << i;
                const f32x4u* g = reinterpret_cast<const f32x4u*>(ok ? sbase + s_off + SA * i * HWs : gs);
                const f32x4 v0 = g[0], v1 = g[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { sv[i][e] = v0[e]; sv[i][4 + e] = v1[e]; }
            }
        }
        // x side.  No branch may contain a load here: hipcc drains vmcnt at every control-flow join, which
        // would serialise the chunk's loads.  Every unit reads a full window from an address that is always
        // inside the tensor (clamped into its row, or the tensor base for rows that do not exist) and the
        // border cases are resolved with selects afterwards.
        const int iy0 = oy0 * S - p.pad, ix0 = ox0 * S - p.pad;
        const float* lbase = xl + ((int64_t)n0 * p.C + b0) * HWl + iy0 * p.W + ix0;
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int m = l_meta[i];
            const int iy = iy0 + ((m >> 12) & 31), ix = ix0 + ((m >> 17) & 127);
            const bool row_ok = l_off[i] >= 0 && n0 + ((m >> 24) & 63) < p.N && iy >= 0 && iy < p.H;
            if constexpr (S == 1) {
                // window [ixc, ixc + 8) with ixc = clamp(ix, 0, W - 8); d = ix - ixc is -1 at the left image
                // border (pad 1), 0 inside, > 0 at the right border, where only the row's last cell can be
                // and only its elements 0 and 1 are ever read (host: OW % tile width == 0)
                int ixc = ix < 0 ? 0 : ix;
                if (ixc > p.W - 8) ixc = p.W - 8;
                const int d = row_ok ? ix - ixc : 64;
                const f32x4u* g = reinterpret_cast<const f32x4u*>(row_ok ? lbase + l_off[i] + (ixc - ix) : xl);
                const f32x4 v0 = g[0], v1 = g[1];
#pragma unroll
                for (int e = 0; e < 4; ++e) { lv[i][e] = v0[e]; lv[i][4 + e] = v1[e]; }
                lcode[i] = d;
            } else {
                // stride 2, pad 0 (host): 16-column units are inside their row; the single-column tail unit
                // (even cell of column 2 TW) reads one element
                const bool single = (m >> 30) & 1;
                const bool vec = row_ok && !single;
                const float* gu = lbase + l_off[i];
                const f32x4u* g = reinterpret_cast<const f32x4u*>(vec ? gu : xl);
                ltail[i] = *((row_ok && single) ? gu : xl);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 v = g[q];
#pragma unroll
                    for (int e = 0; e < 4; ++e) lv[i][4 * q + e] = v[e];
                }
                lcode[i] = vec ? 1 : ((row_ok && single) ? 2 : 0);
            }
        }
    };
    // Border fix-up happens on the packed bf16 cells (dword funnel shifts and selects, no arrays) and only
    // in chunks that touch an image border or a channel / batch tail (workgroup-uniform branch).
    //   stride 1: code = window shift d: -1 -> the window starts one column right of the cell (left border),
    //             6 / 7 -> right border cell, whose elements 0, 1 are window elements d, d + 1; 64 -> dead row
    //   stride 2: code 1 = 16 columns, 2 = single-column tail unit, 0 = dead row
    auto put3 = [&](u32x4* base, int plane, int c, const bf16x8& s0, const bf16x8& s1, const bf16x8& s2, int code) {
        u32x4 v[3] = {__builtin_bit_cast(u32x4, s0), __builtin_bit_cast(u32x4, s1), __builtin_bit_cast(u32x4, s2)};
        if (chunk_border && S == 1) {
#pragma unroll
            for (int sp = 0; sp < 3; ++sp) {
                const u32x4 q = v[sp];
                const u32x4 sh = {q[0] << 16, (q[0] >> 16) | (q[1] << 16), (q[1] >> 16) | (q[2] << 16),
                                  (q[2] >> 16) | (q[3] << 16)};
                u32x4 r = (code == -1) ? sh : q;
                if (code > 0) r[0] = (code == 6) ? q[3] : (q[3] >> 16);
                if (code > 7) r = u32x4{0u, 0u, 0u, 0u};
                v[sp] = r;
            }
        }
        base[c] = v[0];
        base[plane + c] = v[1];
        base[2 * plane + c] = v[2];
    };
    auto store_chunk = [&]() {
#pragma unroll
        for (int i = 0; i < SPT; ++i) {
            float fx[8];
#pragma unroll
            for (int e = 0; e < 8; ++e) fx[e] = (!chunk_border || ((s_okmask >> i) & 1)) ? sv[i][e] : 0.0f;
            bf16x8 s0, s1, s2;
            split3_bf16(fx, s0, s1, s2);
            put3(Ss, BA * SOS, s_cell + SA * SOS * i, s0, s1, s2, 0);
        }
#pragma unroll
        for (int i = 0; i < LPT; ++i) {
            const int c = l_meta[i] & 0xfff;
            if (c == 0xfff) continue;
            if constexpr (S == 1) {
                bf16x8 s0, s1, s2;
                split3_bf16(lv[i], s0, s1, s2);
                put3(Ls, BB * LOS, c, s0, s1, s2, lcode[i]);
            } else {
                const int code = lcode[i];
                float ev[8], od[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    ev[e] = (code == 1) ? lv[i][2 * e] : 0.0f;
                    od[e] = (code == 1) ? lv[i][2 * e + 1] : 0.0f;
                }
                if (code == 2) ev[0] = ltail[i];
                bf16x8 s0, s1, s2;
                split3_bf16(ev, s0, s1, s2);
                put3(Ls, BB * LOS, c, s0, s1, s2, 0);
                if (!((l_meta[i] >> 30) & 1)) {
                    split3_bf16(od, s0, s1, s2);
                    put3(Ls, BB * LOS, c + NOLE, s0, s1, s2, 0);
                }
            }
        }
    };

    // operands: A cells of one K block (two octets); per (K block, ky) the aligned x-side cell(s) + the
    // first dword of the right neighbour.  One wave per SIMD: the LDS reads of step (kb, ky) + 1 are
    // issued before the 18 MFMAs of step (kb, ky) (two alternating register sets).
    struct FragA { u32x4 a[3]; };
    struct FragB {
        u32x4 be[3];
        unsigned bn[3];
        u32x4 bo[(S == 2) ? 3 : 1];
    };
    const int a_row = (wa * 32 + l31) * SOS;
    const int b_row = (wb * 32 + l31) * LOS;
    auto fetchA = [&](int kb, FragA& f) {
        const int o = 2 * kb + half;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) f.a[sp] = Ss[sp * BA * SOS + a_row + o];
    };
    auto fetchB = [&](int kb, int ky, FragB& f) {
        const int o = 2 * kb + half;
        const int poct = o & (TWO - 1);
        const int py = (o >> p.tw8_log2) & (TH - 1);
        const int pn = o >> (p.tw8_log2 + p.th_log2);
        const int c0 = b_row + (pn * PH + py * S + ky) * RC + poct;
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            f.be[sp] = Ls[sp * BB * LOS + c0];
            f.bn[sp] = Ls[sp * BB * LOS + c0 + 1][0];
            if constexpr (S == 2) f.bo[sp] = Ls[sp * BB * LOS + c0 + NOLE];
        }
    };
    auto shift1 = [](const u32x4& c, unsigned n) {
        u32x4 r;
        r[0] = (c[0] >> 16) | (c[1] << 16);
        r[1] = (c[1] >> 16) | (c[2] << 16);
        r[2] = (c[2] >> 16) | (c[3] << 16);
        r[3] = (c[3] >> 16) | (n << 16);
        return r;
    };
    auto mma = [&](const FragA& fa, const FragB& f, auto KY) {
        constexpr int ky = decltype(KY)::value;
        constexpr int TA[6] = {0, 2, 1, 0, 1, 0};
        constexpr int TB[6] = {2, 0, 1, 1, 0, 0};
        bf16x8 a[3], b[3][3];     // b[kx][split]
#pragma unroll
        for (int sp = 0; sp < 3; ++sp) {
            a[sp] = __builtin_bit_cast(bf16x8, fa.a[sp]);
            const u32x4 e = f.be[sp];
            b[0][sp] = __builtin_bit_cast(bf16x8, e);
            if constexpr (S == 1) {
                b[1][sp] = __builtin_bit_cast(bf16x8, shift1(e, f.bn[sp]));
                const u32x4 d = {e[1], e[2], e[3], f.bn[sp]};
                b[2][sp] = __builtin_bit_cast(bf16x8, d);
            } else {
                b[1][sp] = __builtin_bit_cast(bf16x8, f.bo[sp]);
                b[2][sp] = __builtin_bit_cast(bf16x8, shift1(e, f.bn[sp]));
            }
        }
#pragma unroll
        for (int q = 0; q < 6; ++q)
#pragma unroll
            for (int kx = 0; kx < 3; ++kx)
                acc[ky * 3 + kx] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[TA[q]], b[kx][TB[q]],
                                                                           acc[ky * 3 + kx], 0, 0, 0);
    };
    using K0 = std::integral_constant<int, 0>;
    using K1 = std::integral_constant<int, 1>;
    using K2 = std::integral_constant<int, 2>;

    if (ch_begin < ch_end) load_chunk(ch_begin);
    for (int chunk = ch_begin; chunk < ch_end; ++chunk) {
        __syncthreads();
        store_chunk();
        __syncthreads();
        if (chunk + 1 < ch_end) load_chunk(chunk + 1);
        constexpr int KB = NO / 2;
        static_assert(KB % 2 == 0, "K-block loop is unrolled by two");
        FragA a0, a1;
        FragB f0, f1;
        fetchA(0, a0);
        fetchB(0, 0, f0);
        for (int kb = 0; kb < KB; kb += 2) {
            fetchB(kb, 1, f1); mma(a0, f0, K0{});
            fetchB(kb, 2, f0); mma(a0, f1, K1{});
            fetchA(kb + 1, a1); fetchB(kb + 1, 0, f1); mma(a0, f0, K2{});
            fetchB(kb + 1, 1, f0); mma(a1, f1, K0{});
            fetchB(kb + 1, 2, f1); mma(a1, f0, K1{});
            if (kb + 2 < KB) { fetchA(kb + 2, a0); fetchB(kb + 2, 0, f0); }
            mma(a1, f1, K2{});
        }
    }

    // slab store: rows = a (m), cols = b (c)
#pragma unroll
    for (int t = 0; t < T; ++t) {
        const int bcol = b0 + wb * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int arow = a0 + wa * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
            slab[(((int64_t)slice * T + t) * p.Ap + arow) * p.Bp + bcol] = acc[t][r];
        }
    }
}

// Weight gradient of the 1x1 layers with at most four channels on one side (FromRGB 3 -> C, ToRGB C -> 3, stride 1, pad 0):
//   gw[m][c] = alpha * sum_n f[n] sum_pix gy[n][m][pix] x[n][c][pix]
// a streaming reduction -- the "big" tensor (128 channels) is read ONCE as 16-byte quads, four of its channels per workgroup
// against all (<= 4) channels of the "small" one (re-read from L2), 16 fp32 sums per thread, reduced through the wave and LDS.
// On the MFMA path the 3-channel side pads to a 32- or 128-wide tile (> 90 % zeros) and the kernel ran at 1.6 - 3 TB/s
// (0.33 ms for the 537 MB of ToRGB's input at B = 16, 0.46 ms for FromRGB's 1.34 GB at B = 40; the bytes need 0.11 / 0.28).
// Partials per (image, pixel slice) in a fixed order, second stage below: deterministic.
struct ThinWgParams {
    int N, CB, CS, split;
    int64_t HW;
    int quads_per_slice;           // 16-byte quads of a plane handled by one workgroup
    const float* big_scale;        // [N][CB] activation factor of the big side (modulated conv) or null
    const float* small_scale;      // [N][CS] ... of the small side or null
};

__global__ __launch_bounds__(kBlock) void conv1x1_thin_wgrad_kernel(const float* __restrict__ big, const float* __restrict__ small,
                                                                    float* __restrict__ partial, const ThinWgParams p) {
    __shared__ float red[kBlock / kWave][16];
    const int cb0 = blockIdx.x * 4, n = blockIdx.y, sp = blockIdx.z;
    const float* bp = big + ((int64_t)n * p.CB + cb0) * p.HW;
    const float* sp_ = small + (int64_t)n * p.CS * p.HW;
    const int64_t q_begin = (int64_t)sp * p.quads_per_slice;
    int64_t q_end = q_begin + p.quads_per_slice;
    if (q_end > p.HW / 4) q_end = p.HW / 4;
    float acc[4][4] = {};
    for (int64_t q = q_begin + threadIdx.x; q < q_end; q += kBlock) {
        f32x4 bv[4], sv[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {       // rows beyond the channel count re-read row 0 and are dropped at the end
            bv[i] = *reinterpret_cast<const f32x4*>(bp + (int64_t)(cb0 + i < p.CB ? i : 0) * p.HW + 4 * q);
            sv[i] = *reinterpret_cast<const f32x4*>(sp_ + (int64_t)(i < p.CS ? i : 0) * p.HW + 4 * q);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int k = 0; k < 4; ++k)
                acc[i][k] += (bv[i][0] * sv[k][0] + bv[i][1] * sv[k][1]) + (bv[i][2] * sv[k][2] + bv[i][3] * sv[k][3]);
    }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            float v = acc[i][k];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
            if (lane == 0) red[wid][i * 4 + k] = v;
        }
    __syncthreads();
    if (threadIdx.x < 16) {
        const int i = threadIdx.x >> 2, k = threadIdx.x & 3;
        float v = 0.0f;
#pragma unroll
        for (int w = 0; w < kBlock / kWave; ++w) v += red[w][threadIdx.x];
        if (cb0 + i < p.CB && k < p.CS) {
            if (p.big_scale) v *= p.big_scale[(int64_t)n * p.CB + cb0 + i];
            if (p.small_scale) v *= p.small_scale[(int64_t)n * p.CS + k];
        }
        // partial[(n * split + sp)][cb][k]
        partial[(((int64_t)n * p.split + sp) * gridDim.x * 4 + cb0 + i) * 4 + k] = v;
    }
}

// gw[m * sm + c * sc] = alpha * sum over the (image, slice) partials, one wave per output element: lane l sums parts l, l + 64,
// ..., then the wave's shuffle tree (fixed order; one thread per element walked up to 1 024 partials as a dependent chain)
__global__ __launch_bounds__(kBlock) void conv1x1_thin_wgrad_reduce_kernel(const float* __restrict__ partial, float* __restrict__ gw,
                                                                           int CB, int CBp, int CS, int parts, int big_is_m,
                                                                           int64_t sm, int64_t sc, float alpha) {
    const int lane = threadIdx.x & (kWave - 1);
    const int e = blockIdx.x * (kBlock / kWave) + (threadIdx.x >> 6);
    const bool live = e < CB * CS;
    const int cb = live ? e / CS : 0, k = live ? e - cb * CS : 0;
    float v = 0.0f;
    for (int t = lane; t < parts; t += kWave) v += partial[((int64_t)t * CBp + cb) * 4 + k];
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off, 64);
    if (live && lane == 0) {
        const int m = big_is_m ? cb : k, c = big_is_m ? k : cb;
        gw[m * sm + c * sc] = alpha * v;
    }
}

// l_scale / s_scale / spi (slices per image, 0 = none): operand modulation applied per K-slice.  When every slice
// of the pixel range lies inside one image n, the factor of a modulated operand, l_scale[n][c] or s_scale[n][m], is
// constant over the slice and can multiply the slice's partial sum here instead of every staged operand element
// in the main kernel (host: conv_wgrad_impl), which then runs its un-modulated instantiation.
__global__ __launch_bounds__(kBlock) void conv_wgrad_reduce_kernel(const float* __restrict__ slab,
                                                                   float* __restrict__ gw, int M, int C,
                                                                   int Ap, int Bp, int taps, int slices,
                                                                   int64_t sm, int64_t sc, float alpha,
                                                                   const float* __restrict__ l_scale,
                                                                   const float* __restrict__ s_scale, int spi) {
    // Four lanes per output element, each summing every fourth slice with eight loads in flight, combined in a
    // fixed order ((q0 + q1) + (q2 + q3)): a chain of `slices` dependent loads per thread on ~2 workgroups per CU
    // ran at 1.2 TB/s.  Deterministic (the order depends only on `slices`).
    const int64_t total = (int64_t)taps * M * C;
    const int64_t slice_stride = (int64_t)taps * Ap * Bp;
    const int q = threadIdx.x & 3;
    for (int64_t i0 = ((int64_t)blockIdx.x * kBlock + threadIdx.x) >> 2; i0 < ((total + 63) & ~(int64_t)63);
         i0 += ((int64_t)gridDim.x * kBlock) >> 2) {
        const bool live = i0 < total;          // whole quads stay together for the shuffles
        const int64_t i = live ? i0 : 0;
        const int c = (int)(i % C);
        const int64_t t = i / C;
        const int m = (int)(t % M);
        const int tap = (int)(t / M);
        const float* s = slab + ((int64_t)tap * Ap + m) * Bp + c;
        float acc = 0.0f;
        int sl = q;
        if (spi > 0) {       // per-slice operand factors (same slice order as below: the sum stays deterministic)
            for (; sl < slices; sl += 4) {
                const int n = sl / spi;
                float f = 1.0f;
                if (l_scale) f *= l_scale[(int64_t)n * C + c];
                if (s_scale) f *= s_scale[(int64_t)n * M + m];
                acc += s[sl * slice_stride] * f;
            }
        }
        for (; sl + 28 < slices; sl += 32) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = s[(sl + 4 * u) * slice_stride];
#pragma unroll
            for (int u = 0; u < 8; ++u) acc += v[u];
        }
        for (; sl < slices; sl += 4) acc += s[sl * slice_stride];
        acc += __shfl_xor(acc, 1, 64);
        acc += __shfl_xor(acc, 2, 64);
        if (live && q == 0) gw[m * sm + c * sc + tap] = alpha * acc;
    }
}

// ------------------------------------------------------------------------------------------
// host side: planning and dispatch
// ------------------------------------------------------------------------------------------
bool desc_ok(const sae_conv2d_desc* d, const char* who) {
    if (!d) { fail(SAE_EINVAL, "%s: null descriptor", who); return false; }
    if (d->n < 0 || d->c < 1 || d->m < 1 || d->h < 1 || d->w < 1) {
        fail(SAE_EINVAL, "%s: bad tensor dims", who); return false;
    }
    if (d->kh != d->kw || (d->kh != 1 && d->kh != 3) || (d->stride != 1 && d->stride != 2) || d->pad < 0 ||
        d->pad >= d->kh) {
        fail(SAE_EINVAL, "%s: unsupported geometry k=%dx%d stride=%d pad=%d (k in {1,3}, stride in {1,2}, pad < k)",
             who, d->kh, d->kw, d->stride, d->pad);
        return false;
    }
    if (d->oh != (d->h + 2 * d->pad - d->kh) / d->stride + 1 || d->ow != (d->w + 2 * d->pad - d->kw) / d->stride + 1 ||
        d->oh < 1 || d->ow < 1) {
        fail(SAE_EINVAL, "%s: oh/ow do not match the conv formula", who); return false;
    }
    const int64_t lim = (int64_t)1 << 31;
    if (d->n * d->c * d->h * d->w >= lim * 4 || d->n * d->m * d->oh * d->ow >= lim * 4 || d->c * d->h * d->w >= lim ||
        d->m * d->oh * d->ow >= lim || d->n >= (1 << 24) || d->c >= (1 << 24) || d->m >= (1 << 24)) {
        fail(SAE_EINVAL, "%s: tensor too large for 32-bit tile arithmetic", who); return false;
    }
    return true;
}

inline int round_up(int v, int q) { return (v + q - 1) / q * q; }

// tile shape of a forward-type launch producing `mout` channels
struct FwdShape { int cfg; int bm, bn; int ck; };   // cfg 0: 128x128, 1: 64x256, 2: 32x512
FwdShape fwd_shape(int mout, int ks, int stride, int ow) {
    FwdShape s{};
    // tuning knob (benchmarks only): SAE_IGEMM_WIDE=1 gives 3x3 stride-1 layers a 128 x 256 tile
    static const int wide_knob = tuning_knob("SAE_IGEMM_WIDE", 0);
    // exact fp32, 3x3 stride 1, maps at least 128 wide: the 64 x 256 tile for wide layers too -- a 32 x 8 pixel tile has less
    // halo and 30 % less staging traffic per MFMA than 128 x 128 on 16 x 8 pixels (9 instead of 11 vector-memory instructions
    // per chunk); same-box A/B (tools/ab_conv.py): 128 -> 128 @256^2 128.1 -> 130.9 TFLOP/s, 256 -> 256 @128^2 131.5 -> 136.1;
    // at 64^2 and below (long K loops, few tiles) the 128-row tile stays ahead by up to 1 %.  Bit-identical results.
    static const int prefer64_knob = tuning_knob("SAE_IGEMM_PREFER64", 1);
    const bool wide_map_64 = prefer64_knob && conv_math() == 0 && ks == 3 && stride == 1 && ow >= 128;
    static const int bx_s2_knob = tuning_knob("SAE_BX_S2", 1);
    if (mout > 64 && wide_knob && ks == 3 && stride == 1) { s.cfg = 3; s.bm = 128; s.bn = 256; }
    else if (mout > 32 && ks == 3 && stride == 2 && conv_math() == 1 && bx_s2_knob) { s.cfg = 6; s.bm = 64; s.bn = 128; }   // bf16x6 stride 2
    else if (mout > 64 && round_up(mout, 64) * 100 >= round_up(mout, 128) * 92 && !wide_map_64) { s.cfg = 0; s.bm = 128; s.bn = 128; }
    else if (mout > 64) { s.cfg = 1; s.bm = 64; s.bn = 256; }    // e.g. 409 -> 448 instead of 512 padded rows
    else if (mout > 32 || stride == 2) { s.cfg = 1; s.bm = 64; s.bn = 256; }
    else {
        // narrow layers (M <= 32).  a 32 x 256 tile at three workgroups per CU
        // (SAE_IGEMM_NARROW=0, tuning knob: 32 x 512 at one)
        // measured on 32->32 3x3 @128x128 B=128: 104 TFLOP/s (32 x 256) vs 81 (32 x 512)
        static const int narrow_knob = tuning_knob("SAE_IGEMM_NARROW", 1);
        if (narrow_knob && ks == 3 && stride == 1) { s.cfg = 4; s.bm = 32; s.bn = 256; }
        else { s.cfg = 2; s.bm = 32; s.bn = 512; }
    }
    // channels per K-chunk: 3x3 -> 8 (72 k per chunk); 1x1 -> 32, 16 for the 512-pixel tile (LDS)
    s.ck = (ks == 1) ? (s.cfg == 2 ? 16 : 32) : 8;
    return s;
}

// choose TW/TH (powers of two >= 4) for a tile of `bn` pixels over an oh x ow grid
void pick_tile(int bn, int oh, int ow, int max_tw, int* tw_log2, int* th_log2) {
    int tw = 1 << ilog2_ceil(ow);
    if (tw < 4) tw = 4;
    if (tw > max_tw) tw = max_tw;
    int th = 1 << ilog2_ceil(oh);
    if (th < 4) th = 4;
    while (tw * th > bn) th >>= 1;
    if (th < 1) th = 1;
    *tw_log2 = ilog2_ceil(tw);
    *th_log2 = ilog2_ceil(th);
}

struct TrShape { int cfg; int bm, bq; int ck; };   // cfg 0: 128 x 64q, 1: 64 x 128q, 2: 32 x 128q, 3: 64 x 128q with 16-channel chunks,
                                                      // 4: 128 x 128q on 8 waves
TrShape tr_shape(int mout) {
    TrShape s{};
    s.ck = 8;
    static const int cfg_knob = tuning_knob("SAE_TR_CFG", -1);
    if (mout > 32 && cfg_knob == 3) { s.cfg = 3; s.bm = 64; s.bq = 128; s.ck = 16; }
    // (cfg 4, one 8-wave workgroup per CU, measured 73-93 TFLOP/s against 92-104 for two independent 4-wave workgroups:
    // the second workgroup's MFMAs are what covers the staging phases; kept behind the knob)
    else if (mout > 64 && cfg_knob == 4 && conv_math() == 0) { s.cfg = 4; s.bm = 128; s.bq = 128; }
    // exact fp32: the 64 x 128q tile also for wide layers -- a weight element then serves 128 q positions instead of 64 (a tap
    // of the transposed problem feeds one output parity class only, so the weights are the larger share of the staging):
    // 106.6 / 107.9 / 94.6 vs 103.5 / 103.3 / 92.6 TFLOP/s on the three D shapes.  bf16x6 keeps its 128 x 64q kernel.
    else if (mout > 64 && cfg_knob != 1 && (cfg_knob == 0 || conv_math() == 1)) { s.cfg = 0; s.bm = 128; s.bq = 64; }
    else if (mout > 32) { s.cfg = 1; s.bm = 64; s.bq = 128; }
    else { s.cfg = 2; s.bm = 32; s.bq = 128; }
    return s;
}

struct WgShape { int ba, bb; int mode; };
WgShape wg_shape(int m, int c, int ks, int stride) {
    if (c * ks * ks <= 32) return {32, 32, 2};                       // RGB stems: (channel, tap) packed
    if (m <= 32 && c <= 32 && ks == 3 && stride == 1) return {32, 32, 1};
    if (ks == 1) return {128, 128, 0};
    if (stride == 1) return {64, 64, 0};
    return {128, 32, 0};
}

struct WgPlan {
    WgShape sh; int tw_log2, th_log2; int tiles_x, tiles_y, tiles_n; int chunks, cps, slices; int Ap, Bp; int taps;
    bool bx; int tw8_log2;   // bf16-split arithmetic: tiles of TH rows x (8 << tw8_log2) columns
};
// Number of K slices of a weight-gradient launch.  "Fill the chip once" (256 / tiles, 512 / tiles for the kernel that runs two
// workgroups per CU) is right when the tile count divides the slots; Dpatch's 384- and 768-channel layers have 24 ... 144 tiles
// and it is not: 256 -> 384 @8^2, 384 images: 24 tiles x 11 slices = 264 workgroups = one full round of the 256 CUs and a
// second one for the last eight (0.39 of peak, profiles/r4_roofline_by_shape_church256.txt).  The candidates are priced in
// microseconds -- rounds of workgroups x chunks per slice x one chunk (4.7 MFLOP on a CU's matrix cores at the 0.8 the kernels
// reach), plus the slabs written and read back by the fixed-order reduction at 4 TB/s -- and the cheapest one runs, if it
// is at least 7 % cheaper than the plain rule's.  A last round with at most one workgroup per CU of the two-per-CU kernel is
// priced at 0.6 (a workgroup that has its CU to itself runs that much faster; 384 -> 384 @8^2: 576 workgroups, model 0.71 ms,
// measured 0.70).
int wg_pick_slices(int tiles, int chunks, bool two_per_cu, int64_t slab_floats, int plain) {
    static const int knob = tuning_knob("SAE_WGRAD_SLICE_MODEL", 1);
    if (!knob || tiles < 1 || chunks < 2) return plain;
    const int slots = two_per_cu ? 512 : 256;
    constexpr double kChunkUs = 9.4;
    auto cost = [&](int s) {
        const int cps = ceil_div(chunks, s);
        const int eff = ceil_div(chunks, cps);            // slices that exist with this length
        const int64_t wgs = (int64_t)tiles * eff;
        const int64_t full = wgs / slots, rem = wgs - full * slots;
        double rounds = (double)full;
        if (rem > 0) rounds += (two_per_cu && rem <= 256) ? 0.6 : 1.0;
        double us = rounds * cps * kChunkUs + (2.0 * eff + 1.0) * 4.0 * (double)slab_floats / 4e6;
        if (two_per_cu && (eff & 7) != 0) us *= 1.02;     // the XCD-aware order of that kernel wants a multiple of eight
        return us;
    };
    int best = plain;
    double best_us = cost(plain);
    const int hi = chunks < 96 ? chunks : 96;
    for (int s = 1; s <= hi; ++s) {
        const double us = cost(s);
        if (us < best_us) { best_us = us; best = s; }
    }
    return best_us < 0.93 * cost(plain) ? best : plain;
}

// wg16: the plan of conv_wgrad16_kernel (64 a x 32 b workgroup tiles, two workgroups per CU) for a 3 x 3 layer
WgPlan wg_plan(const sae_conv2d_desc* d, bool wg16 = false) {
    WgPlan w{};
    w.sh = wg_shape((int)d->m, (int)d->c, d->kh, d->stride);
    if (wg16) w.sh = {64, 32, 5};
    w.taps = d->kh * d->kw;
    pick_tile(kWgPix, (int)d->oh, (int)d->ow, 32, &w.tw_log2, &w.th_log2);
    int tw = 1 << w.tw_log2, th = 1 << w.th_log2, tn = kWgPix / (tw * th);
    w.bx = false;
    if (!wg16 && conv_math() == 1 && d->kh == 3 && w.sh.mode == 0 && d->ow % 8 == 0 && d->ow >= 16) {
        // octet tiles: 128 (stride 1) / 64 (stride 2) pixels per chunk, rows of 16 or 32 columns
        const int no = (d->stride == 1) ? 16 : 8;
        const int two = (d->ow > 16) ? 4 : 2;
        int bth = 1 << ilog2_ceil(d->oh);
        if (bth > no / two) bth = no / two;
        const int btn = no / (two * bth);
        const int ph = (bth - 1) * d->stride + 3;
        bool fits = (d->stride == 1) ? btn * ph * (two + 1) <= 30
                                     : (btn * ph * (2 * two + 1) <= 45 && btn * ph * (two + 1) <= 27);
        // border handling of the staging loads: whole tiles only; stride 2 with pad 0 only
        if (d->ow % (8 * two) != 0 || (d->stride == 2 && d->pad != 0)) fits = false;
        if (fits) {
            w.bx = true;
            w.tw8_log2 = (two == 4) ? 2 : 1;
            w.th_log2 = ilog2_ceil(bth);
            tw = 8 * two; th = bth; tn = btn;
        }
    }
    w.tiles_x = ceil_div((int)d->ow, tw);
    w.tiles_y = ceil_div((int)d->oh, th);
    w.tiles_n = ceil_div((int)d->n, tn);
    w.chunks = w.tiles_x * w.tiles_y * w.tiles_n;
    w.Ap = round_up((int)d->m, w.sh.ba);
    w.Bp = round_up((int)d->c, w.sh.bb);
    const int mn_tiles = (w.Ap / w.sh.ba) * (w.Bp / w.sh.bb);
    // MODE 0 runs one workgroup per CU (register prefetch): one full wave of 256 workgroups; the
    // narrow modes co-reside 2-3 per CU
    int slices = ceil_div(w.sh.mode == 0 ? 256 : 512, mn_tiles);
    if (wg16) slices = round_up(slices, 8);       // (the XCD-aware order needs slices % 8 == 0)
    if (slices > w.chunks) slices = w.chunks;
    if (slices < 1) slices = 1;
    if ((w.sh.mode == 0 || wg16) && !w.bx && w.taps == 9)     // (the 1x1 layers are HBM-bound: another model)
        slices = wg_pick_slices(mn_tiles, w.chunks, wg16, (int64_t)w.taps * w.Ap * w.Bp, slices);
    w.cps = ceil_div(w.chunks, slices);
    // keep a K slice inside one image where the images are large (>= 32 chunks of 64 pixels): the factors of a
    // style-modulated operand can then be applied per slice in the reduction and the main kernel stays the plain
    // (quad-staged) one.  Costs more, smaller slabs only for the generator's 512-channel 64 x 64 layers (4 -> 8 / 16 slices).
    if ((w.sh.mode == 0 || wg16) && !w.bx && tn == 1) {
        const int cpi = w.tiles_x * w.tiles_y;
        // (batches of at most 16 images only: the generator's; for D / Dpatch at 24 ... 384 images 40 slabs of 9 MB cost
        // more, and moving cps to a divisor of the image breaks the one-workgroup-per-CU balance: 26.9 -> 30.6 ms measured)
        if (cpi >= 32 && w.tiles_n <= 16) {
            if (w.cps > cpi) w.cps = cpi;
            else while (cpi % w.cps != 0) --w.cps;
        }
    }
    w.slices = ceil_div(w.chunks, w.cps);
    return w;
}

// sum of the K-slice partial outputs (fixed order -> deterministic), then the optional fused
// bias + leaky-ReLU epilogue (hw = OH*OW, channels = M locate the bias of a flat element)
__global__ __launch_bounds__(kBlock) void conv_splitk_reduce_kernel(const float* __restrict__ slab,
                                                                    float* __restrict__ y, int64_t numel4,
                                                                    int64_t slab_stride, int ksplit,
                                                                    const float* __restrict__ bias, int act,
                                                                    float act_slope, float act_scale, int hw,
                                                                    int channels, const float* __restrict__ residual,
                                                                    float res_scale, const float* __restrict__ noise,
                                                                    const float* __restrict__ noise_w) {
    const float nwv = noise ? noise_w[0] : 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < numel4; i += (int64_t)gridDim.x * kBlock) {
        f32x4 acc = *reinterpret_cast<const f32x4*>(slab + i * 4);
        for (int s = 1; s < ksplit; ++s) acc += *reinterpret_cast<const f32x4*>(slab + s * slab_stride + i * 4);
        if (act) {
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float v = acc[e];
                if (noise) {        // element (n, m, pix) takes noise[n][pix]
                    const int64_t f = i * 4 + e, plane = f / hw;
                    v = v + nwv * noise[(plane / channels) * hw + (f - plane * hw)];
                }
                if (bias) v += bias[((i * 4 + e) / hw) % channels];
                acc[e] = ((v > 0.0f) ? v : v * act_slope) * act_scale;
            }
        }
        if (residual) acc = (acc + *reinterpret_cast<const f32x4*>(residual + i * 4)) * res_scale;
        *reinterpret_cast<f32x4*>(y + i * 4) = acc;
    }
}

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
