// Does the SCALAR offset of a raw buffer load take part in the descriptor's range check on gfx950?
//
//   hipcc --offload-arch=gfx950 -O3 -o buffer_range_probe buffer_range_probe.hip && ./buffer_range_probe
//
// A 4 KiB allocation holds float i at index i.  The descriptor covers its first 1024 bytes (num_records = 1024); the bytes
// behind are valid memory with distinguishable contents, so a load that escapes the check returns data, one that is stopped
// returns 0.  Each row: voffset, soffset, what came back.  If soffset were ignored, (voffset 512, soffset 512) would return
// float 256; if the check is voffset >= num_records - soffset it returns 0.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void probe(const float* base, float* out, const unsigned* voff, const unsigned* soff, int n) {
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 1024, 0x00020000);
    for (int i = 0; i < n; ++i) {
        const unsigned s = __builtin_amdgcn_readfirstlane(soff[i]);
        out[i] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, voff[i], s, 0));
    }
}

int main() {
    const int N = 1024;
    float h[N];
    for (int i = 0; i < N; ++i) h[i] = (float)i;
    const unsigned hv[] = {0, 1020, 1024, 508, 512, 0, 4, 1016, 1020, 2048};
    const unsigned hs[] = {0, 0, 0, 512, 512, 1020, 1020, 4, 4, 0};
    const int n = sizeof(hv) / sizeof(hv[0]);
    float *d, *o;
    unsigned *dv, *ds;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4); hipMalloc(&dv, n * 4); hipMalloc(&ds, n * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipMemcpy(dv, hv, n * 4, hipMemcpyHostToDevice);
    hipMemcpy(ds, hs, n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(1), 0, 0, d, o, dv, ds, n);
    float ho[16];
    hipMemcpy(ho, o, n * 4, hipMemcpyDeviceToHost);
    printf("# raw buffer, num_records = 1024 bytes over an allocation of 4096 (float i at index i)\n");
    printf("# voffset soffset  address  returned   (in range by voffset alone / by voffset + soffset)\n");
    for (int i = 0; i < n; ++i)
        printf("%8u %7u %8u %9.0f   (%s / %s)\n", hv[i], hs[i], hv[i] + hs[i], ho[i], hv[i] + 4 <= 1024 ? "in" : "out",
               hv[i] + hs[i] + 4 <= 1024 ? "in" : "out");
    return 0;
}
