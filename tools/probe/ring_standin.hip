// Stand-in for ONE all-reduce kernel of a ring over N GPUs, as the local GPU sees it (bench.py --ring-rehearsal, kernel form
// "persistent"): a few persistent workgroups ("channels") that walk the bucket in 2 (N - 1) steps of a 1/N slice -- reduce-scatter
// steps read the local slice and a staging buffer and write the sum, all-gather steps copy -- and that take at least `hold_us` per
// step (s_sleep on the wall clock: the link time a step would wait for).  Nothing leaves the GPU.  Test tooling, not product code.
//   hipcc --offload-arch=gfx950 -O3 -fPIC -shared tools/probe/ring_standin.hip -o tools/probe/libring_standin.so
#include <hip/hip_runtime.h>

__global__ __launch_bounds__(256) void ring_standin_kernel(const float* __restrict__ bucket, float* __restrict__ staging,
                                                           long slice, int n_gpus, int hold_us) {
    const long per = (slice + gridDim.x - 1) / gridDim.x;
    const long lo = (long)blockIdx.x * per;
    long hi = lo + per;
    if (hi > slice) hi = slice;
    for (int s = 0; s < 2 * (n_gpus - 1); ++s) {
        const unsigned long long t0 = __builtin_readcyclecounter();       // s_memtime: shader clock
        const float* src = bucket + (long)(s % n_gpus) * slice;
        if (s < n_gpus - 1) {
            for (long i = lo + threadIdx.x; i < hi; i += 256) staging[i] += src[i];
        } else {
            for (long i = lo + threadIdx.x; i < hi; i += 256) staging[i] = src[i];
        }
        __syncthreads();
        // ~100 MHz constant clock: wall_clock64(); hold the step for hold_us
        const unsigned long long w0 = wall_clock64();
        while ((long long)(wall_clock64() - w0) < (long long)hold_us * 100) __builtin_amdgcn_s_sleep(32);
        (void)t0;
    }
}

extern "C" int ring_standin_launch(const void* bucket, void* staging, long numel, int n_gpus, int channels, int hold_us, void* stream) {
    const long slice = numel / n_gpus;
    if (slice < 1 || channels < 1) return 0;
    hipLaunchKernelGGL(ring_standin_kernel, dim3((unsigned)channels), dim3(256), 0, (hipStream_t)stream, (const float*)bucket,
                       (float*)staging, slice, n_gpus, hold_us);
    return (int)hipGetLastError();
}
