// Probe (gfx950): __builtin_amdgcn_global_load_lds, 16 bytes per lane.  Checks that lane l of a wave lands at
// LDS base + 16 l, that a wave-uniform base works per instruction, and the vmcnt(0) + barrier visibility rule.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void glds_copy(const u32x4* __restrict__ src, u32x4* __restrict__ dst, int cells) {
    __shared__ u32x4 buf[2048];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wid = __builtin_amdgcn_readfirstlane(tid >> 6);
    // wave w copies cells [64 (w + 4 i), +64) for i = 0.. : destination base is wave-uniform
    for (int i = 0; 64 * (wid + 4 * i) < cells; ++i) {
        const int c0 = 64 * (wid + 4 * i);
        const u32x4* g = src + blockIdx.x * cells + c0 + lane;
        __builtin_amdgcn_global_load_lds(g, (__attribute__((address_space(3))) void*)(buf + c0), 16, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int c = tid; c < cells; c += 256) {
        // read a cell written by ANOTHER wave
        dst[blockIdx.x * cells + c] = buf[(c + 64) % cells];
    }
}

int main() {
    const int cells = 2048, blocks = 64;
    std::vector<u32x4> h(cells * blocks), o(cells * blocks);
    for (size_t i = 0; i < h.size(); ++i) h[i] = u32x4{(unsigned)i, (unsigned)(i * 3), (unsigned)(i ^ 0x5a5a), 7u};
    u32x4 *d, *e;
    hipMalloc(&d, h.size() * 16); hipMalloc(&e, h.size() * 16);
    hipMemcpy(d, h.data(), h.size() * 16, hipMemcpyHostToDevice);
    glds_copy<<<blocks, 256>>>(d, e, cells);
    hipMemcpy(o.data(), e, h.size() * 16, hipMemcpyDeviceToHost);
    size_t bad = 0;
    for (int b = 0; b < blocks; ++b)
        for (int c = 0; c < cells; ++c) {
            const u32x4 want = h[b * cells + (c + 64) % cells], got = o[b * cells + c];
            if (want[0] != got[0] || want[1] != got[1] || want[2] != got[2] || want[3] != got[3]) ++bad;
        }
    printf("glds probe: %zu mismatching cells of %zu\n", bad, h.size());
    return bad != 0;
}
