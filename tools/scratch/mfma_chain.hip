// Dependent MFMA chains: NC independent accumulators used round-robin by one wave (32x32x16 bf16 and 16x16x32 bf16), 1 or 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
template <int NC, int SHAPE>
__global__ __launch_bounds__(256) void k(float *out, int iters) {
    bf16x8 a = {}, b = {};
    f32x16 c[4] = {};
    f32x4 d[4] = {};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if constexpr (SHAPE == 0) c[u % NC] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c[u % NC], 0, 0, 0);
            else d[u % NC] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d[u % NC], 0, 0, 0);
        }
    }
    float s = 0;
    for (int i = 0; i < 4; ++i) { for (int e = 0; e < 16; ++e) s += c[i][e]; for (int e = 0; e < 4; ++e) s += d[i][e]; }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NC, int SHAPE>
void run(float *o, int wps) {
    const int iters = 5000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NC, SHAPE><<<256 * wps, 256>>>(o, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NC, SHAPE><<<256 * wps, 256>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double cyc = ms * 1e-3 * 2.4e9 / ((double)wps * iters * 8);
    printf("%s chains=%d waves/SIMD=%d: %.1f cycles per MFMA (pipe time %d)\n", SHAPE ? "16x16x32" : "32x32x16", NC, wps, cyc, SHAPE ? 16 : 32);
}
int main() {
    float *o; hipMalloc(&o, 1024 * 256 * 4);
    run<1, 0>(o, 1); run<2, 0>(o, 1); run<4, 0>(o, 1); run<1, 0>(o, 2); run<2, 0>(o, 2); run<4, 0>(o, 2);
    run<1, 1>(o, 1); run<2, 1>(o, 1); run<4, 1>(o, 1); run<1, 1>(o, 2); run<2, 1>(o, 2); run<4, 1>(o, 2);
    return 0;
}
