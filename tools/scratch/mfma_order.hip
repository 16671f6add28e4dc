// Bare 32x32x16 bf16 MFMA loops on random register operands under the power cap: does the ORDER of the same sixteen products matter?
//   chain : c[i] += a[j] b[(i+j)&3], j inner - consecutive MFMAs share the accumulator, both operands change (what a k-step chain looks like)
//   shareA: the same products, i inner      - consecutive MFMAs share the A operand, accumulator and B change
//   shareB: c[i] += a[(i+j)&3] b[j], i inner - consecutive MFMAs share the B operand
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
template <int MODE>
__global__ __launch_bounds__(256) void k32(const bf16x8 *in, float *out, int iters) {
    bf16x8 a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = in[(threadIdx.x * 8 + i) % 4096]; b[i] = in[(threadIdx.x * 8 + 4 + i) % 4096]; }
    f32x16 c[4] = {};
    for (int it = 0; it < iters; ++it) {
        if (MODE == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int j = 0; j < 4; ++j) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(i + j) & 3], c[i], 0, 0, 0);
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[j], b[(i + j) & 3], c[i], 0, 0, 0);
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int i = 0; i < 4; ++i) c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(i + j) & 3], b[j], c[i], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
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
    const char *names[3] = {"chain ", "shareA", "shareB"};
    for (int rep = 0; rep < 3; ++rep) {
        for (int which = 0; which < 3; ++which) {
            auto launch = [&]() { if (which == 0) k32<0><<<256, 256>>>(d, o, iters); else if (which == 1) k32<1><<<256, 256>>>(d, o, iters); else k32<2><<<256, 256>>>(d, o, iters); };
            for (int w = 0; w < 30; ++w) launch();
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int w = 0; w < 10; ++w) launch();
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            const double flop = 10.0 * 256 * 4 * (double)iters * 16 * 32768.0;
            printf("%s %s: %.3f ms, %.0f TFLOP/s\n", zeros ? "zeros " : "random", names[which], ms / 10, flop / (ms * 1e-3) / 1e12);
        }
    }
    return 0;
}
