// Shared host/device helpers for the gfx950 hot-path library (see include/sae_hip.h).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#include "sae_hip.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));   // A/B operand of v_mfma_f32_32x32x16_bf16

namespace sae {

constexpr int kWave = 64;       // CDNA wavefront
constexpr int kBlock = 256;     // 4 waves, one per SIMD of a CU

char* err_buf();                // thread-local message buffer (sae_api.hip)
int fail(int code, const char* fmt, ...);

// hipGetLastError() is per thread and sticky: a probe that failed inside another library (e.g. during the framework's
// device discovery) would otherwise be reported by our first check_launch.  Every entry point clears it on entry.
inline void clear_stale_error() { (void)hipGetLastError(); }

inline int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(SAE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
    return SAE_OK;
}

__host__ __device__ inline int64_t ceil_div64(int64_t a, int64_t b) { return (a + b - 1) / b; }
__host__ __device__ inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

// Dispatch knobs for A/B experiments exist only in builds with -DSAE_TUNING (tools/build_variant*.sh, the emulator of
// tests/emu): the product library reads no environment variable besides SAE_CONV_MATH (sae_api.hip).
#ifdef SAE_TUNING
inline int tuning_knob(const char* name, int dflt) {
    const char* e = getenv(name);
    return e ? atoi(e) : dflt;
}
#else
inline int tuning_knob(const char*, int dflt) { return dflt; }
#endif

inline int ilog2_ceil(int64_t v) {  // smallest e with (1 << e) >= v, v >= 1
    int e = 0;
    while (((int64_t)1 << e) < v) ++e;
    return e;
}

// conv arithmetic: 0 = v_mfma_f32_32x32x2_f32 (default), 1 = three-way bf16 split, six bf16 MFMAs per
// product block, fp32 accumulate (see conv2d.hip "bx"); set by sae_set_conv_math / SAE_CONV_MATH
int conv_math();

// Lanes of ONE wave exchanging data through LDS: the wave's LDS instructions execute in order, so nothing has to be waited
// for; this only keeps the compiler from moving LDS accesses across the exchange point.
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

// Packed fp32 adds with the operand-select / negate modifiers of VOP3P, which hipcc does not emit for pairs it has to build
// (it negates and moves instead: two instructions more per pair).  On pairs p = (p.x, p.y):
//   pk_sub(a, b)  = (a.x - b.x, a.y - b.y)
//   pk_c01(e01, e23) = (e01.x - e23.x, e01.y + e23.x)      columns 0, 1 of  e B  for a row e = (e0, e1, e2, e3), B of Winograd F(2x2,3x3)
//   pk_c23(e01, e23) = (e23.x - e01.y, e01.y - e23.y)      columns 2, 3
// (the emulator of tests/emu models them in plain C++: HIPEMU is defined by its hip_runtime.h)
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef HIPEMU
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) { return f32x2{a[0] - b[0], a[1] - b[1]}; }
__device__ __forceinline__ f32x2 pk_c01(f32x2 e01, f32x2 e23) { return f32x2{e01[0] - e23[0], e01[1] + e23[0]}; }
__device__ __forceinline__ f32x2 pk_c23(f32x2 e01, f32x2 e23) { return f32x2{e23[0] - e01[1], e01[1] - e23[1]}; }
#else
__device__ __forceinline__ f32x2 pk_sub(f32x2 a, f32x2 b) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
__device__ __forceinline__ f32x2 pk_c01(f32x2 e01, f32x2 e23) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,0]" : "=v"(d) : "v"(e01), "v"(e23));
    return d;
}
__device__ __forceinline__ f32x2 pk_c23(f32x2 e01, f32x2 e23) {
    f32x2 d;
    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[1,0]" : "=v"(d) : "v"(e23), "v"(e01));
    return d;
}
#endif

}  // namespace sae
