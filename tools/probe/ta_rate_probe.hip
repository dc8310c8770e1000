// Probe (gfx950): what a buffer load of a given shape costs the texture addresser / L1 when the data HITS the cache -- cycles per
// wave instruction with 4 waves per CU (one per SIMD, as in the Winograd kernels) hammering loads of one shape from a small
// resident buffer.  Shapes: dwords per lane (1, 2, 3, 4) x lane stride in bytes (4, 8, 16) x byte misalignment of the window.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/ta_rate_probe.hip -o tools/probe/ta_rate_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x3 __attribute__((ext_vector_type(3)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int DW>
__global__ __launch_bounds__(256, 1) void hammer(const float* __restrict__ src, float* __restrict__ dst, long long* cyc, int iters,
                                                 int lane_stride, int misalign, int rows) {
    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(src), 0, 1u << 20, 0x00020000);
    // 16 lanes per row run, 4 runs 1 KB apart (the tile block's 4 tile rows), waves 8 KB apart
    unsigned off = (unsigned)(wid * 8192 + (lane >> 4) * 1024 + (lane & 15) * lane_stride + misalign);
    float acc = 0.0f;
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 6; ++r) {
            const unsigned so = (unsigned)(((it & 7) * 6 + r) * 32768) % (1u << 19);
            if (DW == 4) { const f32x4 v = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs, off, so, 0)); acc += v[0] + v[3]; }
            if (DW == 3) { const u32x3 v = __builtin_amdgcn_raw_buffer_load_b96(rs, off, so, 0); acc += __builtin_bit_cast(float, v[0]) + __builtin_bit_cast(float, v[2]); }
            if (DW == 2) { const f32x2 v = __builtin_bit_cast(f32x2, __builtin_amdgcn_raw_buffer_load_b64(rs, off, so, 0)); acc += v[0] + v[1]; }
            if (DW == 1) { acc += __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, so, 0)); }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    dst[blockIdx.x * 256 + tid] = acc;
    if (tid == 0) cyc[blockIdx.x] = t1 - t0;
}

int main() {
    float* d; float* o; long long* c;
    hipMalloc(&d, 1 << 20); hipMalloc(&o, 256 * 256 * 4); hipMalloc(&c, 256 * 8);
    hipMemset(d, 0, 1 << 20);
    const int iters = 2000;
    printf("# cycles (s_memtime ticks of 100 MHz -> converted with the event time) per wave load instruction, 4 waves per CU, 256 CUs, hits\n");
    for (int dw = 1; dw <= 4; ++dw)
        for (int stride : {4, 8, 16})
            for (int mis : {0, 4}) {
                if (stride < 4 * dw && stride != 8) continue;
                hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    hipEventRecord(e0);
                    if (dw == 1) hammer<1><<<256, 256>>>(d, o, c, iters, stride, mis, 4);
                    if (dw == 2) hammer<2><<<256, 256>>>(d, o, c, iters, stride, mis, 4);
                    if (dw == 3) hammer<3><<<256, 256>>>(d, o, c, iters, stride, mis, 4);
                    if (dw == 4) hammer<4><<<256, 256>>>(d, o, c, iters, stride, mis, 4);
                    hipEventRecord(e1); hipEventSynchronize(e1);
                    hipEventElapsedTime(&ms, e0, e1);
                }
                // per CU: 4 waves x iters x 6 instructions in ms
                const double ns_per_instr_cu = ms * 1e6 / (4.0 * iters * 6);
                printf("dwords %d  lane stride %2d B  misalign %d B : %7.2f ns per wave instruction per CU  (= %6.1f cycles at 2.1 GHz)   %6.1f GB/s per CU requested\n",
                       dw, stride, mis, ns_per_instr_cu, ns_per_instr_cu * 2.1, 64.0 * 4 * dw / ns_per_instr_cu);
            }
    return 0;
}
