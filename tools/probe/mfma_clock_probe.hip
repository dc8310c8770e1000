// Probe (gfx950): what does the fp32 matrix pipe sustain under the conv kernels' instruction mix, and at which clock?
//   mode 0: v_mfma_f32_32x32x2_f32 back to back on four independent accumulators, operands in registers
//   mode 1: + the LDS reads of the conv kernels (one ds_read2_b32 + two ds_read_b32 per four MFMAs)
//   mode 2: + the conv's staging per K chunk of 144 MFMAs the way conv_igemm_kernel does it: barrier, 16 dword + 9 x
//           16-byte registers written to LDS, barrier, the next chunk's global loads issued in one burst (weights from an
//           L2-resident region shared by all workgroups, activations streamed), then the MFMAs
//   mode 3: the same work as a software pipeline: two LDS buffers, ONE barrier per chunk, the LDS writes of chunk c+1 and
//           the global loads of chunk c+2 issued one or two at a time between the MFMA groups of chunk c
//   mode 4: mode 3 with 4-channel chunks (72 MFMAs per barrier; both buffers together are as large as mode 2's one, so
//           two workgroups still fit a CU)
// Every launch reports TFLOP/s from HIP events and the shader clock seen by the waves themselves:
// clock64() (s_memtime) against wall_clock64() (s_memrealtime, constant 100 MHz).
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/mfma_clock_probe.hip -o tools/probe/mfma_clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned kWeights = 1u << 21;     // floats of the L2/MALL-resident "weight" region (8 MB); offsets are masked:
                                            // a 64-bit modulo costs ~150 instructions and would dominate the probe

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, const float* __restrict__ wsrc,
                                             float* __restrict__ sink, long long span, int chunks,
                                             unsigned long long* __restrict__ clk) {
    __shared__ float As[9 * 8 * 128];
    __shared__ float Xs[8 * 288];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned smask = (unsigned)span - 1u;     // span is a power of two
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int i = tid; i < 9 * 8 * 128; i += 256) As[i] = 1e-3f * (float)(i & 15);
    for (int i = tid; i < 8 * 288; i += 256) Xs[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    float a0 = 1e-3f * lane, a1 = 2e-3f, b0 = 1e-3f, b1 = 3e-3f;
    const float* g = src + (long long)blockIdx.x * 256 + tid;      // coalesced dwords, like the conv's row segments
    if (MODE == 5) {
        // the two workgroups of a CU start together and stay in phase (both staging at the same time): delay the one
        // whose wave 0 sits in an odd wave slot by about half a chunk (HW_ID bits 3:0 = wave slot within the SIMD)
        __shared__ int odd;
        if (tid == 0) odd = __builtin_amdgcn_s_getreg((3 << 11) | 4) & 1;
        __syncthreads();
        if (odd) { __builtin_amdgcn_s_sleep(100); }
    }
    float xv[16];
    f32x4 av[9];
    for (int c = 0; c < chunks; ++c) {
        if (MODE == 2 || MODE == 5) {
            const unsigned off = (unsigned)c * gridDim.x * 256u;
#pragma unroll
            for (int i = 0; i < 16; ++i) xv[i] = g[(off + (unsigned)i * 65536u) & smask];
#pragma unroll
            for (int i = 0; i < 9; ++i)
                av[i] = *reinterpret_cast<const f32x4*>(wsrc + (((unsigned)c * 9216u + (i * 256 + tid) * 4) & (kWeights - 1)));
        }
#pragma unroll
        for (int s = 0; s < 36; ++s) {          // 36 steps x 4 MFMAs = one K chunk of the 128 x 128 conv tile
            if (MODE >= 1) {
                const int row = (s * 128 + (lane & 31)) % (9 * 8 * 128 - 32);
                a0 = As[row]; a1 = As[row + 32];
                b0 = Xs[(s * 8 + lane) % (8 * 288 - 40)]; b1 = Xs[(s * 8 + lane) % (8 * 288 - 40) + 33];
            }
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[3], 0, 0, 0);
        }
        if (MODE == 2 || MODE == 5) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < 16; ++i) Xs[(tid + 256 * i) % (8 * 288)] = xv[i];
#pragma unroll
            for (int i = 0; i < 9; ++i) *reinterpret_cast<f32x4*>(As + 4 * (tid + 256 * i)) = av[i];
            __syncthreads();
        }
    }
    float keep = 0.0f;
    for (int i = 0; i < 4; ++i) keep += acc[i][0] + acc[i][7] + acc[i][15];
    if (keep == 12345.678f) sink[0] = keep;
    if (tid == 0) {
        atomicAdd(&clk[0], clock64() - c0);
        atomicAdd(&clk[1], wall_clock64() - r0);
    }
}

// the software pipeline: CK channels per chunk, two LDS buffers.  FLAGS: 1 global loads, 2 LDS writes, 4 barrier,
// 8 = two register sets (the loads of chunk c+2 start at step 0 of chunk c, a full chunk ahead of their LDS write)
template <int CK, int FLAGS>
__global__ __launch_bounds__(256) void probe_pipe(const float* __restrict__ src, const float* __restrict__ wsrc,
                                                  float* __restrict__ sink, long long span, int chunks,
                                                  unsigned long long* __restrict__ clk) {
    constexpr int ASZ = 9 * CK * 128, XSZ = CK * 288;
    constexpr int NSTEP = 9 * CK / 2;                  // (tap, channel pair) steps of 4 MFMAs
    constexpr int NX = 2 * CK;                         // activation dwords per thread per chunk
    constexpr int NA = (ASZ / 4 + 255) / 256;          // weight float4 per thread per chunk (the last one partial)
    constexpr int NITEM = NX + NA;
    constexpr bool TWOSETS = (FLAGS & 8) != 0;
    static_assert(NITEM <= NSTEP, "at most two staging instructions per step");
    __shared__ float As[2][ASZ];
    __shared__ float Xs[2][XSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned smask = (unsigned)span - 1u;     // span is a power of two
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int i = tid; i < 2 * ASZ; i += 256) As[0][i] = 1e-3f * (float)(i & 15);
    for (int i = tid; i < 2 * XSZ; i += 256) Xs[0][i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    const float* g = src + (long long)blockIdx.x * 256 + tid;      // coalesced dwords, like the conv's row segments
    float xv[2][NX];
    f32x4 av[2][NA];
    for (int k = 0; k < 2; ++k) {
        for (int i = 0; i < NX; ++i) xv[k][i] = 1e-3f * i;
        for (int i = 0; i < NA; ++i) av[k][i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    }
    auto gload = [&](int c, int item, float (&x)[NX], f32x4 (&a)[NA]) {                // one global load of chunk c
        if (!(FLAGS & 1)) return;
        if (item < NX) {
            if (FLAGS & 16) return;                 // 16: no activation loads
            const unsigned off = (unsigned)c * gridDim.x * 256u;
            if (FLAGS & 64) {                       // 64: the same bytes as dwordx4 loads (a quarter of the instructions)
                if ((item & 3) == 0) {
                    const f32x4 v = *reinterpret_cast<const f32x4*>(src + (((off + (unsigned)item * 65536u) & smask) & ~1023u) +
                                                                    blockIdx.x * 1024 + tid * 4);
                    x[item] = v[0]; x[item + 1] = v[1]; x[item + 2] = v[2]; x[item + 3] = v[3];
                }
            } else {
                x[item] = g[(off + (unsigned)item * 65536u) & smask];
            }
        } else {
            if (FLAGS & 32) return;                 // 32: no weight loads
            const int i = item - NX;
            if (4 * (i * 256 + tid) < ASZ)
                a[i] = *reinterpret_cast<const f32x4*>(wsrc + (((unsigned)c * ASZ + (i * 256 + tid) * 4) & (kWeights - 1)));
        }
    };
    auto lwrite = [&](int buf, int item, const float (&x)[NX], const f32x4 (&a)[NA]) {             // its LDS write
        if (!(FLAGS & 2)) return;
        if (item < NX) Xs[buf][(tid + 256 * item) % XSZ] = x[item];
        else {
            const int i = item - NX;
            if (4 * (i * 256 + tid) < ASZ) *reinterpret_cast<f32x4*>(&As[buf][4 * (tid + 256 * i)]) = a[i];
        }
    };
    float ra[3][2], rb[3][2];
    auto chunk = [&](int c, float (&xw)[NX], f32x4 (&aw)[NA], float (&xl)[NX], f32x4 (&al)[NA]) {
        // xw/aw: registers holding chunk c+1 (written to LDS now); xl/al: registers receiving chunk c+2
        const int cur = c & 1;
        auto fetch = [&](int s, float (&a)[2], float (&b)[2]) {
            const int row = (s * 128 + (lane & 31)) % (ASZ - 32);
            a[0] = As[cur][row]; a[1] = As[cur][row + 32];
            b[0] = Xs[cur][(s * 8 + lane) % (XSZ - 40)]; b[1] = Xs[cur][(s * 8 + lane) % (XSZ - 40) + 33];
        };
        fetch(0, ra[0], rb[0]);
        fetch(1, ra[1], rb[1]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 2 < NSTEP) fetch(s + 2, ra[(s + 2) % 3], rb[(s + 2) % 3]);
            if (TWOSETS) {          // one LDS write and one global load per step from step 0
                if (s < NITEM) { gload(c + 2, s, xl, al); lwrite(cur ^ 1, s, xw, aw); }
            } else {                // one register set: all writes first (two per step), then the loads into the same registers
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = 2 * s + q;
                    if (j < NITEM) lwrite(cur ^ 1, j, xw, aw);
                    else if (j < 2 * NITEM) gload(c + 2, j - NITEM, xl, al);
                }
            }
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % 3][0], rb[s % 3][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % 3][0], rb[s % 3][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % 3][1], rb[s % 3][0], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s % 3][1], rb[s % 3][1], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        if (FLAGS & 4) __syncthreads();
    };
    for (int c = 0; c < chunks; c += 2) {
        if (TWOSETS) {
            chunk(c, xv[1], av[1], xv[0], av[0]);
            chunk(c + 1, xv[0], av[0], xv[1], av[1]);
        } else {
            chunk(c, xv[0], av[0], xv[0], av[0]);
            chunk(c + 1, xv[0], av[0], xv[0], av[0]);
        }
    }
    float keep = 0.0f;
    for (int i = 0; i < 4; ++i) keep += acc[i][0] + acc[i][7] + acc[i][15];
    for (int i = 0; i < NX; ++i) keep += xv[0][i] + xv[1][i];
    for (int i = 0; i < NA; ++i) keep += av[0][i][0] + av[1][i][3];
    if (keep == 12345.678f) sink[0] = keep;
    if (tid == 0) {
        atomicAdd(&clk[0], clock64() - c0);
        atomicAdd(&clk[1], wall_clock64() - r0);
    }
}

// the 128 x 256 tile as a software pipeline at one wave per SIMD: 8 MFMAs per (tap, channel pair) step on 128
// accumulator registers, per chunk and thread 9 dwordx4 weight loads + 4 dwordx4 activation-quad loads and as many
// ds_write_b128, two LDS buffers, one barrier per chunk.  SPREAD: staging instructions per step (1 or 2)
template <int SPREAD>
__global__ __launch_bounds__(256) void probe_wide(const float* __restrict__ src, const float* __restrict__ wsrc,
                                                  float* __restrict__ sink, long long span, int chunks,
                                                  unsigned long long* __restrict__ clk) {
    constexpr int CK = 8, ASZ = 9 * CK * 128, XSZ = CK * 416, NSTEP = 36, NA = 9, NX = 4, NITEM = NA + NX;
    __shared__ float As[2][ASZ];
    __shared__ float Xs[2][XSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const unsigned smask = (unsigned)span - 1u;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x16 acc[8];
    for (int i = 0; i < 8; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int i = tid; i < 2 * ASZ; i += 256) As[0][i] = 1e-3f * (float)(i & 15);
    for (int i = tid; i < 2 * XSZ; i += 256) Xs[0][i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    f32x4 st[NITEM];
    for (int i = 0; i < NITEM; ++i) st[i] = f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f};
    auto gload = [&](int c, int item) {
        if (item < NA) st[item] = *reinterpret_cast<const f32x4*>(wsrc + (((unsigned)c * ASZ + (item * 256 + tid) * 4) & (kWeights - 1)));
        else st[item] = *reinterpret_cast<const f32x4*>(src + ((((unsigned)c * gridDim.x + blockIdx.x) * 4096u + (item - NA) * 1024 + tid * 4) & smask));
    };
    auto lwrite = [&](int buf, int item) {
        if (item < NA) *reinterpret_cast<f32x4*>(&As[buf][4 * (tid + 256 * item)]) = st[item];
        else *reinterpret_cast<f32x4*>(&Xs[buf][(4 * (tid + 256 * (item - NA))) % (XSZ - 4)]) = st[item];
    };
    float ra[2][2], rb[2][4];
    for (int c = 0; c < chunks; ++c) {
        const int cur = c & 1;
        auto fetch = [&](int s, float (&a)[2], float (&b)[4]) {
            const int row = (s * 128 + (lane & 31)) % (ASZ - 32);
            a[0] = As[cur][row]; a[1] = As[cur][row + 32];
            const int xb = (s * 8 + lane) % (XSZ - 140);
            b[0] = Xs[cur][xb]; b[1] = Xs[cur][xb + 40]; b[2] = Xs[cur][xb + 80]; b[3] = Xs[cur][xb + 120];
        };
        fetch(0, ra[0], rb[0]);
#pragma unroll
        for (int s = 0; s < NSTEP; ++s) {
            if (s + 1 < NSTEP) fetch(s + 1, ra[(s + 1) & 1], rb[(s + 1) & 1]);
#pragma unroll
            for (int q = 0; q < SPREAD; ++q) {
                const int j = SPREAD * s + q;
                if (j < NITEM) lwrite(cur ^ 1, j);
                else if (j < 2 * NITEM) gload(c + 2, j - NITEM);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int ni = 0; ni < 4; ++ni)
                    acc[mi * 4 + ni] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s & 1][mi], rb[s & 1][ni], acc[mi * 4 + ni], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
    }
    float keep = 0.0f;
    for (int i = 0; i < 8; ++i) keep += acc[i][0] + acc[i][7] + acc[i][15];
    for (int i = 0; i < NITEM; ++i) keep += st[i][0] + st[i][3];
    if (keep == 12345.678f) sink[0] = keep;
    if (tid == 0) {
        atomicAdd(&clk[0], clock64() - c0);
        atomicAdd(&clk[1], wall_clock64() - r0);
    }
}

// mode 6: the chunk's weights go global -> LDS by DMA (global_load_lds_dwordx4: no registers, no ds_write), activations
// through registers as 2 dwordx4; ONE LDS buffer, so the DMA is issued after the barrier that ends the MFMA phase and waited
// for before the next one -- its latency is exposed to this workgroup and has to be covered by the others.  ~70 registers
// per wave: launched with three workgroups per CU.
__global__ __launch_bounds__(256) void probe_dma(const float* __restrict__ src, const float* __restrict__ wsrc,
                                                 float* __restrict__ sink, long long span, int chunks,
                                                 unsigned long long* __restrict__ clk) {
    constexpr int ASZ = 9 * 8 * 128, XSZ = 8 * 288;
    __shared__ float As[ASZ];
    __shared__ float Xs[XSZ];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    const unsigned smask = (unsigned)span - 1u;
    const unsigned long long c0 = clock64(), r0 = wall_clock64();
    f32x16 acc[4];
    for (int i = 0; i < 4; ++i)
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    for (int i = tid; i < ASZ; i += 256) As[i] = 1e-3f * (float)(i & 15);
    for (int i = tid; i < XSZ; i += 256) Xs[i] = 1e-3f * (float)(i & 7);
    __syncthreads();
    f32x4 xq[2] = {f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}, f32x4{1e-3f, 2e-3f, 3e-3f, 4e-3f}};
    float ra[2][2], rb[2][2];
    for (int c = 0; c < chunks; ++c) {
        // activations of the next chunk: in flight under the MFMAs
#pragma unroll
        for (int i = 0; i < 2; ++i)
            xq[i] = *reinterpret_cast<const f32x4*>(src + ((((unsigned)c * gridDim.x + blockIdx.x) * 2048u + i * 1024 + tid * 4) & smask));
        auto fetch = [&](int s, float (&a)[2], float (&b)[2]) {
            const int row = (s * 128 + (lane & 31)) % (ASZ - 32);
            a[0] = As[row]; a[1] = As[row + 32];
            b[0] = Xs[(s * 8 + lane) % (XSZ - 40)]; b[1] = Xs[(s * 8 + lane) % (XSZ - 40) + 33];
        };
        fetch(0, ra[0], rb[0]);
#pragma unroll
        for (int s = 0; s < 36; ++s) {
            if (s + 1 < 36) fetch(s + 1, ra[(s + 1) & 1], rb[(s + 1) & 1]);
            __builtin_amdgcn_sched_barrier(0);
            acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s & 1][0], rb[s & 1][0], acc[0], 0, 0, 0);
            acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s & 1][0], rb[s & 1][1], acc[1], 0, 0, 0);
            acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s & 1][1], rb[s & 1][0], acc[2], 0, 0, 0);
            acc[3] = __builtin_amdgcn_mfma_f32_32x32x2f32(ra[s & 1][1], rb[s & 1][1], acc[3], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        // weights of the next chunk straight into LDS: 9 x 256 lanes x 16 bytes
        for (int j = wid; j < 36; j += 4)
            __builtin_amdgcn_global_load_lds(wsrc + (((unsigned)c * ASZ + (j * 64 + lane) * 4) & (kWeights - 1)),
                                             (__attribute__((address_space(3))) void*)(&As[j * 256]), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Xs[(4 * (tid + 256 * i)) % (XSZ - 4)]) = xq[i];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    float keep = 0.0f;
    for (int i = 0; i < 4; ++i) keep += acc[i][0] + acc[i][7] + acc[i][15];
    if (keep == 12345.678f) sink[0] = keep;
    if (tid == 0) {
        atomicAdd(&clk[0], clock64() - c0);
        atomicAdd(&clk[1], wall_clock64() - r0);
    }
}

template <typename K>
static void run(const char* name, K kernel, int mfma_per_chunk, const float* src, const float* wsrc, float* sink,
                long long span, int blocks, int chunks, unsigned long long* clk, int launches) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    for (int l = 0; l < launches; ++l) {
        hipMemset(clk, 0, 16);
        hipEventRecord(e0);
        kernel<<<blocks, 256>>>(src, wsrc, sink, span, chunks, clk);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms = 0;
        hipEventElapsedTime(&ms, e0, e1);
        unsigned long long h[2];
        hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
        const double flops = (double)blocks * 4 * chunks * mfma_per_chunk * 4096.0;
        const double mhz = (double)h[0] / ((double)h[1] / (wall_khz * 1e3)) / 1e6;
        printf("%s launch %d: %8.3f ms  %6.1f TFLOP/s = %.3f of 157.3   shader clock %4.0f MHz\n", name, l, ms,
               flops / ms / 1e9, flops / ms / 1e9 / 157.3, mhz);
    }
    if (hipGetLastError() != hipSuccess) printf("%s: launch error\n", name);
}

int main(int argc, char** argv) {
    const int chunks = argc > 1 ? atoi(argv[1]) : 2000;     // ~8 ms per launch at full rate
    const int blocks = 512;                                  // two workgroups (8 waves) per CU where LDS allows
    const int n = argc > 2 ? atoi(argv[2]) : 3;
    const long long span = 256ll << 20;                      // 1 GiB of floats streamed as "activations"
    float *src, *wsrc, *sink;
    unsigned long long* clk;
    hipMalloc(&src, span * 4 + (64 << 20)); hipMalloc(&wsrc, kWeights * 4 + 65536); hipMalloc(&sink, 64); hipMalloc(&clk, 16);
    hipMemset(src, 0, span * 4 + (64 << 20));
    hipMemset(wsrc, 0, kWeights * 4 + 65536);
    run("mode 0 (MFMA only)          ", probe<0>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("mode 1 (+ LDS reads)        ", probe<1>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("mode 2 (+ staging, 2 barr.) ", probe<2>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("mode 5 (mode 2, odd slots late)", probe<5>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("mode 6 (weights by LDS-DMA, 3 workgroups/CU)", probe_dma, 144, src, wsrc, sink, span, 768, chunks, clk, n);
    run("mode 6 (weights by LDS-DMA, 2 workgroups/CU)", probe_dma, 144, src, wsrc, sink, span, 512, chunks, clk, n);
    run("wide 128x256, 1 staging instr/step", probe_wide<1>, 288, src, wsrc, sink, span, 256, chunks, clk, n);
    run("wide 128x256, 2 staging instr/step", probe_wide<2>, 288, src, wsrc, sink, span, 256, chunks, clk, n);
    run("pipe CK8 loads+writes+barrier", probe_pipe<8, 7>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 barrier only        ", probe_pipe<8, 4>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 writes+barrier      ", probe_pipe<8, 6>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 loads+barrier       ", probe_pipe<8, 5>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 loads+writes        ", probe_pipe<8, 3>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 all, no X loads      ", probe_pipe<8, 7 + 16>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 all, no A loads      ", probe_pipe<8, 7 + 32>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK8 all, X as dwordx4    ", probe_pipe<8, 7 + 64>, 144, src, wsrc, sink, span, blocks, chunks, clk, n);
    run("pipe CK4 all, X as dwordx4    ", probe_pipe<4, 7 + 64>, 72, src, wsrc, sink, span, blocks, 2 * chunks, clk, n);
    run("pipe CK4 all, no A loads      ", probe_pipe<4, 7 + 32>, 72, src, wsrc, sink, span, blocks, 2 * chunks, clk, n);
    run("pipe CK4 loads+writes+barrier", probe_pipe<4, 7>, 72, src, wsrc, sink, span, blocks, 2 * chunks, clk, n);
    run("pipe CK4 barrier only        ", probe_pipe<4, 4>, 72, src, wsrc, sink, span, blocks, 2 * chunks, clk, n);
    run("pipe CK4 writes+barrier      ", probe_pipe<4, 6>, 72, src, wsrc, sink, span, blocks, 2 * chunks, clk, n);
    return 0;
}
