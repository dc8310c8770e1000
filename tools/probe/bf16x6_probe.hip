// Probe (gfx950): operand layout of v_mfma_f32_32x32x16_bf16 and accuracy of the 6-term bf16 split
// product against the exact-f32 MFMA and a double reference.  Not part of the library.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

__device__ inline void split3(float a, __bf16& h, __bf16& m, __bf16& l) {
    h = (__bf16)a; float r = a - (float)h;
    m = (__bf16)r; r = r - (float)m;
    l = (__bf16)r;
}

// A [32][K], B [K][32] row-major f32, K multiple of 16.  One wave.
__global__ void probe(const float* A, const float* B, int K, float* D6, float* D32, float* D1) {
    int l = threadIdx.x, i = l & 31, h = l >> 5;
    f32x16 acc6 = {0}, acc1 = {0}, acc32 = {0};
    for (int k0 = 0; k0 < K; k0 += 16) {
        bf16x8 a0, a1, a2, b0, b1, b2;
        for (int e = 0; e < 8; ++e) {
            __bf16 x, y, z;
            split3(A[i * K + k0 + 8 * h + e], x, y, z); a0[e] = x; a1[e] = y; a2[e] = z;
            split3(B[(k0 + 8 * h + e) * 32 + i], x, y, z); b0[e] = x; b1[e] = y; b2[e] = z;
        }
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b2, acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a2, b0, acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc6, 0, 0, 0);
        acc6 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc6, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc1, 0, 0, 0);
        for (int kk = 0; kk < 16; kk += 2)
            acc32 = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kk + h], B[(k0 + kk + h) * 32 + i], acc32, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * h;
        D6[row * 32 + i] = acc6[r]; D1[row * 32 + i] = acc1[r]; D32[row * 32 + i] = acc32[r];
    }
}

// throughput: 4 independent accumulators, N back-to-back MFMAs per wave
__global__ void rate_bf16(float* out, int iters) {
    bf16x8 a, b; for (int e = 0; e < 8; ++e) { a[e] = (__bf16)(float)(threadIdx.x + e); b[e] = (__bf16)1.0f; }
    f32x16 c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    for (int it = 0; it < iters; ++it) {
        c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c3, 0, 0, 0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = c0[0] + c1[1] + c2[2] + c3[3];
}

int main() {
    const int K = 1152;
    std::vector<float> A(32 * K), B(K * 32);
    srand(7);
    for (auto& v : A) v = (rand() / (float)RAND_MAX * 2 - 1) * expf((rand() % 9) - 4.0f);
    for (auto& v : B) v = (rand() / (float)RAND_MAX * 2 - 1);
    float *dA, *dB, *d6, *d32, *d1;
    hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4);
    hipMalloc(&d6, 4096); hipMalloc(&d32, 4096); hipMalloc(&d1, 4096);
    hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice);
    hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
    probe<<<1, 64>>>(dA, dB, K, d6, d32, d1);
    std::vector<float> D6(1024), D32(1024), D1(1024);
    hipMemcpy(D6.data(), d6, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(D32.data(), d32, 4096, hipMemcpyDeviceToHost);
    hipMemcpy(D1.data(), d1, 4096, hipMemcpyDeviceToHost);
    double e6 = 0, e32 = 0, e1 = 0, nrm = 0, eabs6 = 0, eabs32 = 0, sabs = 0;
    for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
        double ref = 0, sa = 0;
        for (int k = 0; k < K; ++k) { ref += (double)A[i * K + k] * B[k * 32 + j]; sa += fabs((double)A[i * K + k] * B[k * 32 + j]); }
        e6 += pow(D6[i * 32 + j] - ref, 2); e32 += pow(D32[i * 32 + j] - ref, 2); e1 += pow(D1[i * 32 + j] - ref, 2);
        nrm += ref * ref;
        eabs6 = fmax(eabs6, fabs(D6[i * 32 + j] - ref) / sa); eabs32 = fmax(eabs32, fabs(D32[i * 32 + j] - ref) / sa);
    }
    printf("K=%d rel-L2 err: bf16x6 %.3e  f32-mfma %.3e  bf16x1 %.3e\n", K, sqrt(e6 / nrm), sqrt(e32 / nrm), sqrt(e1 / nrm));
    printf("max |err|/sum|ab|: bf16x6 %.3e  f32-mfma %.3e\n", eabs6, eabs32);
    float* dout; hipMalloc(&dout, 256 * 8 * 256 * 4);
    hipEvent_t e0, e1e; hipEventCreate(&e0); hipEventCreate(&e1e);
    int iters = 20000;
    rate_bf16<<<256 * 8, 256>>>(dout, 100);
    hipEventRecord(e0); rate_bf16<<<256 * 8, 256>>>(dout, iters); hipEventRecord(e1e); hipEventSynchronize(e1e);
    float ms; hipEventElapsedTime(&ms, e0, e1e);
    double fl = 256.0 * 8 * 4 * iters * 4 * 2.0 * 32 * 32 * 16;
    printf("bf16 32x32x16 rate: %.1f TFLOP/s (%.3f ms)\n", fl / ms * 1e-9, ms);
    return 0;
}
