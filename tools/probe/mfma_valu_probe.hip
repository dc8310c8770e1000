// Probe (gfx950): what does a vector-ALU instruction cost the fp32 matrix pipe, at one and at two waves per SIMD?
// Each wave runs groups of four DEPENDENT v_mfma_f32_32x32x2_f32 (one accumulator tile per group, NACC tiles in rotation: the
// instruction mix of csrc/winograd_fused.hip's K loop) followed by NV v_fma_f32 instructions on private registers (inline asm, so
// that the count is exact), dependent (one chain) or independent (eight chains).  256 workgroups (one per CU) of 4 * WPS waves.
//   WPS = 1: 16 accumulator tiles per wave (256 AGPRs: one wave per SIMD)      WPS = 2: 8 tiles per wave, two waves per SIMD
// Prints executed-MFMA TFLOP/s and the fraction of the 157.3 TFLOP/s peak per (WPS, NV, chain form).
//   hipcc --offload-arch=gfx950 -O3 tools/probe/mfma_valu_probe.hip -o tools/probe/mfma_valu_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int WPS, int NV, bool DEP>
__global__ __launch_bounds__(256 * WPS, 1) void probe(float* __restrict__ sink, int iters) {
    constexpr int NACC = 16 / WPS;
    f32x16 acc[NACC];
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][r] = 0.0f;
    float a = 1e-3f * (threadIdx.x & 63), b = 2e-3f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = 1e-3f * i + a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int g = 0; g < NACC; ++g) {
#pragma unroll
            for (int s = 0; s < 4; ++s) acc[g] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[g], 0, 0, 0);
#pragma unroll
            for (int k = 0; k < NV; ++k) {
                if (DEP)
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[0]) : "v"(a), "v"(b));
                else
                    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[k & 7]) : "v"(a), "v"(b));
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    float keep = 0.0f;
#pragma unroll
    for (int i = 0; i < NACC; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) keep += acc[i][r];
#pragma unroll
    for (int i = 0; i < 8; ++i) keep += v[i];
    if (keep == 123.456f) sink[threadIdx.x] = keep;
}

template <int WPS, int NV, bool DEP>
void run(float* sink) {
    const int iters = 2000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL((probe<WPS, NV, DEP>), dim3(256), dim3(256 * WPS), 0, 0, sink, 50);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL((probe<WPS, NV, DEP>), dim3(256), dim3(256 * WPS), 0, 0, sink, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.0f;
    hipEventElapsedTime(&ms, e0, e1);
    // per wave and iteration: 16 / WPS groups x 4 MFMAs x 4096 FLOPs; 256 workgroups x 4 WPS waves
    const double flops = 256.0 * 4 * WPS * iters * (16.0 / WPS) * 4 * 4096.0 * 2 / 2;   // 32x32x2: 2048 MACs = 4096 FLOPs
    const double tf = flops / (ms * 1e-3) / 1e12;
    // cycles per group of four MFMAs at 2.4 GHz, per SIMD: ideal 256
    printf("waves/SIMD %d  VALU per 4 MFMAs %2d (%s)  %.3f ms  %6.1f TFLOP/s  %.3f of peak\n", WPS, NV, DEP ? "one chain " : "8 chains  ", ms, tf,
           tf / 157.3);
}

int main() {
    float* sink;
    hipMalloc(&sink, 4096);
    run<1, 0, false>(sink);
    run<1, 2, false>(sink);
    run<1, 4, false>(sink);
    run<1, 8, false>(sink);
    run<1, 16, false>(sink);
    run<1, 8, true>(sink);
    run<1, 16, true>(sink);
    run<2, 0, false>(sink);
    run<2, 4, false>(sink);
    run<2, 8, false>(sink);
    run<2, 16, false>(sink);
    run<2, 32, false>(sink);
    run<2, 16, true>(sink);
    run<2, 32, true>(sink);
    return 0;
}
