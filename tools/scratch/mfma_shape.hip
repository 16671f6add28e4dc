// Bare MFMA loops on random register operands: v_mfma_f32_32x32x16_bf16 vs v_mfma_f32_16x16x32_bf16, one wave per SIMD, every CU busy.
// Prints TFLOP/s of each (the guide's DVFS item 7: at equal cycles per FLOP the 16x16x32 loop holds a higher clock).
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

__global__ __launch_bounds__(256) void k32(const bf16x8 *in, float *out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(i + j) & 3], c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
__global__ __launch_bounds__(256) void k16(const bf16x8 *in, float *out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
    f32x4 c[16] = {};
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int i = 0; i < 16; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) c[i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a[(i + j) & 3], b[(i >> 2) & 3], c[i], 0, 0, 0);
    float s = 0;
    for (int i = 0; i < 16; ++i) for (int e = 0; e < 4; ++e) s += c[i][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main(int argc, char **argv) {
    const bool zeros = argc > 1;
    const int n = 4096 * 8;
    unsigned short *h = (unsigned short *)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) { float f = zeros ? 0.f : (rand() / (float)RAND_MAX) * 2 - 1; unsigned u; memcpy(&u, &f, 4); h[i] = (unsigned short)(u >> 16); }
    bf16x8 *d; float *o;
    hipMalloc(&d, n * 2); hipMalloc(&o, 256 * 256 * 4);
    hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 20000;
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 2; ++which) {
            for (int w = 0; w < 30; ++w) { if (which) k16<<<256, 256>>>(d, o, iters); else k32<<<256, 256>>>(d, o, iters); }  // ~2 s warm
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) { if (which) k16<<<256, 256>>>(d, o, iters); else k32<<<256, 256>>>(d, o, iters); }
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            // per launch: 256 blocks x 4 waves x iters x (16 mfma 32x32x16 = 16*32768 flop | 32 mfma 16x16x32 = 32*16384 flop)
            const double flop = 10.0 * 256 * 4 * (double)iters * 16 * 32768.0;
            printf("%s %s: %.3f ms, %.0f TFLOP/s\n", zeros ? "zeros" : "random", which ? "16x16x32" : "32x32x16", ms / 10, flop / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
