// Probe (gfx950): global_load_dwordx4 from addresses that are only 4-byte aligned (rows of 2^k + 1 floats, the blur
// outputs feeding the stride-2 convs).  Checks the data and times a row-wise gather of 65-float rows as dword loads
// against the same bytes as 17 quads per row.
//   hipcc --offload-arch=gfx950 -O3 -Wno-unused-value tools/probe/unaligned_x4_probe.hip -o tools/probe/unaligned_x4_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x4 __attribute__((ext_vector_type(4)));

// plane of H x W floats (W odd); each workgroup gathers 5 rows x 68 floats starting at column x0 (any alignment)
template <bool QUADS>
__global__ __launch_bounds__(256) void gather(const float* __restrict__ src, float* __restrict__ dst, int W, int planes_per_block) {
    const int tid = threadIdx.x;
    float acc = 0.0f;
    for (int pl = 0; pl < planes_per_block; ++pl) {
        const float* plane = src + ((long long)blockIdx.x * planes_per_block + pl) * (long long)W * W;
        const int x0 = (blockIdx.x * 7 + pl * 3) % (W - 68), y0 = (blockIdx.x * 5 + pl) % (W - 5);
        if (QUADS) {
            if (tid < 85) {
                const int r = tid / 17, q = tid % 17;
                const float* a = plane + (y0 + r) * W + x0 + 4 * q;
                f32x4 v;
                // 4-byte aligned address: the cast promises nothing more (packed struct keeps align 4)
                struct __attribute__((packed, aligned(4))) U { f32x4 v; };
                v = reinterpret_cast<const U*>(a)->v;
                acc += v[0] + 2.0f * v[1] + 3.0f * v[2] + 4.0f * v[3];
            }
        } else {
            for (int e = tid; e < 340; e += 256) {
                const int r = e / 68, c = e % 68;
                acc += (float)((c & 3) + 1) * plane[(y0 + r) * W + x0 + c];
            }
        }
    }
    dst[blockIdx.x * 256 + tid] = acc;
}

int main() {
    const int W = 257, planes = 4096, ppb = 8, blocks = planes / ppb;
    std::vector<float> h((size_t)planes * W * W);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (float)((i * 2654435761u) >> 20 & 1023) * (1.0f / 64.0f);
    float *d, *o0, *o1;
    hipMalloc(&d, h.size() * 4); hipMalloc(&o0, blocks * 256 * 4); hipMalloc(&o1, blocks * 256 * 4);
    hipMemcpy(d, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms[2];
    for (int v = 0; v < 2; ++v) {
        for (int it = 0; it < 3; ++it) {
            hipEventRecord(e0);
            if (v) gather<true><<<blocks, 256>>>(d, o1, W, ppb); else gather<false><<<blocks, 256>>>(d, o0, W, ppb);
            hipEventRecord(e1); hipEventSynchronize(e1);
            hipEventElapsedTime(&ms[v], e0, e1);
        }
    }
    std::vector<float> a(blocks * 256), b(blocks * 256);
    hipMemcpy(a.data(), o0, a.size() * 4, hipMemcpyDeviceToHost);
    hipMemcpy(b.data(), o1, b.size() * 4, hipMemcpyDeviceToHost);
    // per block the sums over all threads must agree (different thread -> element maps)
    int bad = 0;
    for (int blk = 0; blk < blocks; ++blk) {
        double sa = 0, sb = 0;
        for (int t = 0; t < 256; ++t) { sa += a[blk * 256 + t]; sb += b[blk * 256 + t]; }
        if (fabs(sa - sb) > 1e-3 * fabs(sa)) ++bad;
    }
    printf("unaligned dwordx4: %s (%d of %d block sums differ); dword gather %.3f ms, quad gather %.3f ms (hipGetLastError %d)\n",
           bad ? "WRONG" : "correct", bad, blocks, ms[0], ms[1], (int)hipGetLastError());
    return bad != 0;
}
