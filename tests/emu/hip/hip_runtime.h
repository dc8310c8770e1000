// hipemu — a tiny CPU emulator of the HIP constructs used by csrc/*.hip.
//
// TEST INFRASTRUCTURE ONLY (tests/emu).  It exists because the build container has no GPU and
// GPU minutes are scarce: tests/emu/build_emu.py compiles the UNMODIFIED kernel sources of
// swapping_autoencoder_pytorch_amd/csrc with the host clang++ against this header (it shadows
// <hip/hip_runtime.h> on the include path) into tests/emu/libsae_emu.so, and the CPU test-suite
// runs the kernels' real index arithmetic, LDS staging, barriers and MFMA lane layouts against
// the oracle.  Nothing in the product package includes, links or loads any of this.
//
// Model: one OS thread.  Each HIP thread of a block is a ucontext fiber; blocks run one after
// another.  __syncthreads() and the wave-collective operations (MFMA, shuffles, readfirstlane)
// are rendezvous points: a fiber that arrives yields to the round-robin scheduler until every
// live thread of the block / wave has arrived.  `__shared__` becomes `static` (blocks are
// sequential, so one copy is correct).  The MFMA lane layouts implemented here are the ones
// documented for gfx950 (v_mfma_f32_32x32x2_f32 / v_mfma_f32_16x16x4_f32):
//   A operand lane l -> A[i = l & 31][k = l >> 5],  B operand lane l -> B[k = l >> 5][j = l & 31]
//   D: col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5)              (32x32x2)
//   A[l & 15][l >> 4], B[l >> 4][l & 15], D: col = l & 15, row = 4 * (l >> 4) + reg   (16x16x4)
// and products are accumulated as a k-ordered fmaf chain, as the hardware does.
#pragma once
#define HIPEMU 1      // sources that need a plain-C++ model of an inline-asm instruction key on this

#include <ucontext.h>
#include <sys/mman.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <vector>
#include <atomic>

// ---- keywords -------------------------------------------------------------------------------
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static

// ---- basic types ----------------------------------------------------------------------------
struct dim3 {
    unsigned x, y, z;
    constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }

typedef int hipError_t;
static const hipError_t hipSuccess = 0;
typedef struct hipemu_stream* hipStream_t;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipPeekAtLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "hipemu: no error"; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) {
    std::memset(p, v, n);
    return hipSuccess;
}

static const int warpSize = 64;

namespace hipemu {

static const int kWave = 64;
static const size_t kStackBytes = 256 * 1024;
static const int kMaxThreads = 1024;

struct WaveState {
    unsigned gen = 0;
    int arrived = 0;
    int alive = 0;
    float fa[2][kWave];
    float fb[2][kWave];
    uint64_t bits[2][kWave];
    uint16_t ha[2][kWave][8];
    uint16_t hb[2][kWave][8];
};

struct Fiber {
    ucontext_t ctx;
    bool done = false;
    int tid = 0;
    dim3 tidx;
};

struct BlockState {
    std::vector<Fiber> fibers;
    std::vector<WaveState> waves;
    ucontext_t sched;
    int cur = 0;
    unsigned gen = 0;
    int arrived = 0;
    int alive = 0;
    std::function<void()> body;
};

inline BlockState*& blk() {
    static BlockState* b = nullptr;
    return b;
}
inline char*& stack_pool() {
    static char* p = nullptr;
    return p;
}

struct Builtins {
    dim3 threadIdx, blockIdx, blockDim, gridDim;
};
inline Builtins& bi() {
    static Builtins b;
    return b;
}

inline void yield() {
    BlockState& B = *blk();
    Fiber& f = B.fibers[B.cur];
    swapcontext(&f.ctx, &B.sched);
}

inline void block_barrier() {
    BlockState& B = *blk();
    unsigned my = B.gen;
    B.arrived++;
    if (B.arrived >= B.alive) {
        B.gen++;
        B.arrived = 0;
        return;
    }
    while (B.gen == my) yield();
}

inline int lane_id() { return blk()->fibers[blk()->cur].tid % kWave; }
inline WaveState& my_wave() { return blk()->waves[blk()->fibers[blk()->cur].tid / kWave]; }

// Rendezvous of the live lanes of the calling lane's wave; returns the slot (0/1) whose exchange
// buffers were filled before the rendezvous.
inline void wave_rendezvous(WaveState& W) {
    unsigned my = W.gen;
    W.arrived++;
    if (W.arrived >= W.alive) {
        W.gen++;
        W.arrived = 0;
        return;
    }
    while (W.gen == my) yield();
}

inline void fiber_entry() {
    BlockState& B = *blk();
    B.body();
    Fiber& f = B.fibers[B.cur];
    f.done = true;
    B.alive--;
    WaveState& W = B.waves[f.tid / kWave];
    W.alive--;
    // a thread that exits no longer takes part in barriers: release waiters if it was the last
    if (B.arrived > 0 && B.arrived >= B.alive) {
        B.gen++;
        B.arrived = 0;
    }
    if (W.arrived > 0 && W.arrived >= W.alive) {
        W.gen++;
        W.arrived = 0;
    }
    swapcontext(&f.ctx, &B.sched);
}

inline void run_block(const std::function<void()>& body, dim3 block) {
    const int nthreads = (int)(block.x * block.y * block.z);
    if (nthreads > kMaxThreads) {
        std::fprintf(stderr, "hipemu: block too large (%d)\n", nthreads);
        std::abort();
    }
    if (!stack_pool()) {
        void* p = mmap(nullptr, kStackBytes * kMaxThreads, PROT_READ | PROT_WRITE,
                       MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
        if (p == MAP_FAILED) {
            std::perror("hipemu mmap");
            std::abort();
        }
        stack_pool() = (char*)p;
    }
    BlockState B;
    B.body = body;
    B.fibers.resize(nthreads);
    B.waves.resize((nthreads + kWave - 1) / kWave);
    B.alive = nthreads;
    for (int t = 0; t < nthreads; ++t) {
        Fiber& f = B.fibers[t];
        f.tid = t;
        f.tidx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
        B.waves[t / kWave].alive++;
        getcontext(&f.ctx);
        f.ctx.uc_stack.ss_sp = stack_pool() + (size_t)t * kStackBytes;
        f.ctx.uc_stack.ss_size = kStackBytes;
        f.ctx.uc_link = nullptr;
        makecontext(&f.ctx, (void (*)())fiber_entry, 0);
    }
    BlockState* prev = blk();
    blk() = &B;
    int remaining = nthreads;
    long spins = 0;
    while (remaining > 0) {
        for (int t = 0; t < nthreads; ++t) {
            Fiber& f = B.fibers[t];
            if (f.done) continue;
            B.cur = t;
            bi().threadIdx = f.tidx;
            swapcontext(&B.sched, &f.ctx);
            if (f.done) remaining--;
        }
        if (++spins > 200000000L) {
            std::fprintf(stderr, "hipemu: deadlock suspected (barrier never released)\n");
            std::abort();
        }
    }
    blk() = prev;
}

template <typename... KArgs, typename... Args>
inline void launch(void (*kernel)(KArgs...), dim3 grid, dim3 block, size_t /*shmem*/,
                   hipStream_t /*stream*/, Args... args) {
    bi().gridDim = grid;
    bi().blockDim = block;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                bi().blockIdx = dim3(bx, by, bz);
                std::function<void()> body = [=]() { kernel(static_cast<KArgs>(args)...); };
                run_block(body, block);
            }
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

inline f32x16 mfma_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    WaveState& W = my_wave();
    const int l = lane_id();
    const int slot = W.gen & 1;
    W.fa[slot][l] = a;
    W.fb[slot][l] = b;
    wave_rendezvous(W);
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        float acc = c[r];
        for (int k = 0; k < 2; ++k) acc = std::fmaf(W.fa[slot][k * 32 + row], W.fb[slot][k * 32 + col], acc);
        c[r] = acc;
    }
    return c;
}

inline f32x4 mfma_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    WaveState& W = my_wave();
    const int l = lane_id();
    const int slot = W.gen & 1;
    W.fa[slot][l] = a;
    W.fb[slot][l] = b;
    wave_rendezvous(W);
    const int col = l & 15;
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * (l >> 4) + r;
        float acc = c[r];
        for (int k = 0; k < 4; ++k) acc = std::fmaf(W.fa[slot][k * 16 + row], W.fb[slot][k * 16 + col], acc);
        c[r] = acc;
    }
    return c;
}

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

inline float bf16_bits_to_float(uint16_t h) {
    uint32_t u = (uint32_t)h << 16;
    float f;
    std::memcpy(&f, &u, 4);
    return f;
}

// v_mfma_f32_32x32x16_bf16: lane l supplies A[i = l&31][k = 8*(l>>5) + e] and B[k][j = l&31],
// e = 0..7; C/D layout as the f32 32x32 form.  Products of bf16 are exact in fp32; the 16-term
// sum is formed in double and rounded once (the hardware's internal order is not documented).
inline f32x16 mfma_32x32x16bf16(bf16x8 a, bf16x8 b, f32x16 c, int, int, int) {
    WaveState& W = my_wave();
    const int l = lane_id();
    const int slot = W.gen & 1;
    std::memcpy(W.ha[slot][l], &a, 16);
    std::memcpy(W.hb[slot][l], &b, 16);
    wave_rendezvous(W);
    const int col = l & 31;
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        double acc = c[r];
        for (int k = 0; k < 16; ++k)
            acc += (double)bf16_bits_to_float(W.ha[slot][(k >> 3) * 32 + row][k & 7]) *
                   (double)bf16_bits_to_float(W.hb[slot][(k >> 3) * 32 + col][k & 7]);
        c[r] = (float)acc;
    }
    return c;
}

template <typename T>
inline T exchange(T v, int src_lane) {
    static_assert(sizeof(T) <= 8, "hipemu exchange: type too wide");
    WaveState& W = my_wave();
    const int l = lane_id();
    const int slot = W.gen & 1;
    uint64_t raw = 0;
    std::memcpy(&raw, &v, sizeof(T));
    W.bits[slot][l] = raw;
    wave_rendezvous(W);
    T out;
    uint64_t r = W.bits[slot][src_lane & (kWave - 1)];
    std::memcpy(&out, &r, sizeof(T));
    return out;
}

}  // namespace hipemu

#define threadIdx (hipemu::bi().threadIdx)
#define blockIdx (hipemu::bi().blockIdx)
#define blockDim (hipemu::bi().blockDim)
#define gridDim (hipemu::bi().gridDim)

#define hipLaunchKernelGGL(kernel, ...) hipemu::launch(kernel, __VA_ARGS__)

static inline void __syncthreads() { hipemu::block_barrier(); }

#define __builtin_amdgcn_mfma_f32_32x32x2f32 hipemu::mfma_32x32x2f32
#define __builtin_amdgcn_mfma_f32_16x16x4f32 hipemu::mfma_16x16x4f32
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 hipemu::mfma_32x32x16bf16
// LDS-DMA (global_load_lds_*): lane l of the wave writes `size` bytes at (wave-uniform LDS base) + size * l.
// The copy is performed at once; the real instruction is asynchronous (ordered by vmcnt + a barrier), which
// a correct program cannot observe.
#define __builtin_amdgcn_global_load_lds(g, l, size, off, aux) \
    std::memcpy((char*)(l) + (size) * hipemu::lane_id() + (off), (const void*)(g), (size))
#define __builtin_amdgcn_s_waitcnt(imm) ((void)0)
// wave-level LDS exchange: a rendezvous of the wave's live lanes (on the GPU the lanes run in lockstep)
#define __builtin_amdgcn_wave_barrier() hipemu::wave_rendezvous(hipemu::my_wave())
#define __builtin_amdgcn_fence(order, scope) ((void)0)
#define __builtin_amdgcn_sched_barrier(mask) ((void)0)
#ifndef __clang__
#define __builtin_assume(x) ((void)0)
#endif
#define __builtin_amdgcn_sched_group_barrier(mask, n, id) ((void)0)
#define __builtin_amdgcn_s_setprio(p) ((void)0)


// ---- raw buffer loads (buffer_load_dword* through a V# descriptor): address = base + soffset + voffset; every dword with
// voffset + 4 > num_records - soffset reads as zero (the hardware's per-dword range check of raw buffers; the scalar offset DOES
// take part: measured on gfx950 in round 6 -- a descriptor whose num_records had been reduced by soffset on top zeroed valid
// data of the last image, profiles/r6_buffer_range_check.txt).  A "negative" voffset is a huge unsigned one: out of range.
struct hipemu_rsrc {
    const char* base;
    unsigned num_records;
};
// Guard range (tests only): bytes a buffer load must never touch, e.g. the 64 bytes that follow a tensor.  A load that passes
// its range check and still reads inside [lo, hi) is counted instead of performed -- what a GPU memory-access fault would be
// when the tensor ends its allocator segment (the memcpy alone cannot see it).  hipemu_set_guard(0, 0) switches it off.
inline const char* volatile hipemu_guard_lo = nullptr;
inline const char* volatile hipemu_guard_hi = nullptr;
inline std::atomic<long long> hipemu_guard_count{0};
extern "C" inline __attribute__((visibility("default"), used)) void hipemu_set_guard(const void* lo, const void* hi) {
    hipemu_guard_lo = (const char*)lo; hipemu_guard_hi = (const char*)hi; hipemu_guard_count = 0;
}
extern "C" inline __attribute__((visibility("default"), used)) long long hipemu_guard_hits() { return hipemu_guard_count.load(); }
typedef unsigned hipemu_u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned hipemu_u32x2 __attribute__((ext_vector_type(2)));
namespace hipemu {
template <typename V, int N>
inline V raw_buffer_load(hipemu_rsrc r, unsigned voffset, unsigned soffset) {
    V out;
    for (int i = 0; i < N; ++i) {
        unsigned v = 0;
        const unsigned long long off = (unsigned long long)voffset + 4ull * i;
        if (off + soffset + 4 <= r.num_records) {
            const char* a = r.base + soffset + off;
            if (a + 4 > hipemu_guard_lo && a < hipemu_guard_hi) ++hipemu_guard_count;
            else std::memcpy(&v, a, 4);
        }
        out[i] = v;
    }
    return out;
}
inline unsigned raw_buffer_load1(hipemu_rsrc r, unsigned voffset, unsigned soffset) {
    unsigned v = 0;
    if ((unsigned long long)voffset + soffset + 4 <= r.num_records) {
        const char* a = r.base + soffset + voffset;
        if (a + 4 > hipemu_guard_lo && a < hipemu_guard_hi) ++hipemu_guard_count;
        else std::memcpy(&v, a, 4);
    }
    return v;
}
}  // namespace hipemu
#define __amdgpu_buffer_rsrc_t hipemu_rsrc
#define __builtin_amdgcn_make_buffer_rsrc(p, stride, num, flags) (hipemu_rsrc{(const char*)(p), (unsigned)(num)})
#define __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, aux) hipemu::raw_buffer_load<hipemu_u32x4, 4>((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, aux) hipemu::raw_buffer_load<hipemu_u32x2, 2>((r), (unsigned)(voff), (unsigned)(soff))
#define __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, aux) hipemu::raw_buffer_load1((r), (unsigned)(voff), (unsigned)(soff))

template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) {
    const int l = hipemu::lane_id();
    const int base = l & ~(width - 1);
    return hipemu::exchange(v, base + ((l ^ mask) & (width - 1)));
}
template <typename T>
static inline T __shfl_down(T v, unsigned delta, int width = 64) {
    const int l = hipemu::lane_id();
    const int base = l & ~(width - 1);
    int src = (l & (width - 1)) + (int)delta;
    if (src >= width) src = l & (width - 1);
    return hipemu::exchange(v, base + src);
}
template <typename T>
static inline T __shfl(T v, int src, int width = 64) {
    const int l = hipemu::lane_id();
    const int base = l & ~(width - 1);
    return hipemu::exchange(v, base + (src & (width - 1)));
}
static inline int __builtin_amdgcn_readfirstlane(int v) { return hipemu::exchange(v, 0); }

static inline float atomicAdd(float* p, float v) {
    float old = *p;
    *p = old + v;
    return old;
}
static inline int atomicAdd(int* p, int v) {
    int old = *p;
    *p = old + v;
    return old;
}
static inline unsigned atomicAdd(unsigned* p, unsigned v) {
    unsigned old = *p;
    *p = old + v;
    return old;
}

static inline float __fmaf_rn(float a, float b, float c) { return std::fmaf(a, b, c); }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fsub_rn(float a, float b) { volatile float r = a - b; return r; }
// floorf / ceilf: the C library's global-namespace functions (<cmath>)
static inline float __frsqrt_rn(float x) { return 1.0f / std::sqrt(x); }
using std::fmaf;
using std::fmaxf;
using std::fminf;
