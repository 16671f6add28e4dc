// Diagnostic library (kfunca_amd/_build/libkfunca_diag.so; NOT part of libkfunca_hip.so, not declared in include/kfunca_hip.h).
// kf_diag_mfma_ceiling: what the matrix pipes of THIS box sustain under its power cap on random bf16 register operands when
// nothing else is asked of them - one wave per SIMD on every CU issuing back-to-back MFMAs, no loads, no LDS, no VALU. bench.py
// prints it beside the dominant kernel's rate (roofline.frac_of_capped_mfma): the 2.5 PFLOP/s peak assumes 2.4 GHz, the cap
// holds such loops at 1.7-2.0 GHz.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;

// 16 x v_mfma_f32_32x32x16_bf16 per iteration (the attention streams' shape): 16 * 32768 FLOP per wave
__global__ __launch_bounds__(256) void ceiling_32x32x16(const bf16x8 *in, float *out, int iters) {
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

// 32 x v_mfma_f32_16x16x32_bf16 per iteration (the GEMM kernels' shape): 32 * 16384 FLOP per wave
__global__ __launch_bounds__(256) void ceiling_16x16x32(const bf16x8 *in, float *out, int iters) {
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

static uint16_t bf16_of(float f) { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); }

// shape 0: 32x32x16, 1: 16x16x32. Runs the loop for `warm_s` seconds untimed (the power controller settles), then times
// launches for about `timed_s` seconds with HIP events on `stream`; *tflops = executed FLOP / time. zeros != 0: all-zero
// operands (what the same loop holds when no data toggles). Returns 0, or the hipError_t that stopped it.
extern "C" __attribute__((visibility("default"))) int kf_diag_mfma_ceiling(int shape, int zeros, double warm_s, double timed_s, void *stream,
                                                                             double *tflops) {
    hipStream_t s = (hipStream_t)stream;
    int dev = 0, cus = 256;
    hipGetDevice(&dev);
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev);
    const int n = 4096 * 8, iters = 4000;  // one launch: cus x 4 waves x iters x 16 x 32768 FLOP (~0.3 ms at 1.8 PFLOP/s)
    uint16_t *h = (uint16_t *)malloc(n * 2);
    srand(1);
    for (int i = 0; i < n; ++i) h[i] = zeros ? 0 : bf16_of((rand() / (float)RAND_MAX) * 2 - 1);
    bf16x8 *d = nullptr;
    float *o = nullptr;
    hipError_t e = hipMalloc(&d, n * 2);
    if (e == hipSuccess) e = hipMalloc(&o, (size_t)cus * 256 * 4);
    if (e == hipSuccess) e = hipMemcpy(d, h, n * 2, hipMemcpyHostToDevice);
    free(h);
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (e == hipSuccess) e = hipEventCreate(&e0);
    if (e == hipSuccess) e = hipEventCreate(&e1);
    auto launch = [&](int count) {
        for (int w = 0; w < count; ++w) {
            if (shape) ceiling_16x16x32<<<cus, 256, 0, s>>>(d, o, iters);
            else ceiling_32x32x16<<<cus, 256, 0, s>>>(d, o, iters);
        }
    };
    const double flop_launch = (double)cus * 4 * iters * 16 * 32768.0;
    if (e == hipSuccess) {
        // calibrate on 8 launches, then warm / time by launch counts (no host clock inside the timed region)
        hipEventRecord(e0, s); launch(8); hipEventRecord(e1, s);
        e = hipEventSynchronize(e1);
        float ms = 1.f;
        hipEventElapsedTime(&ms, e0, e1);
        const double per = ms > 0 ? ms / 8 * 1e-3 : 3e-4;
        int nwarm = (int)(warm_s / per), ntimed = (int)(timed_s / per);
        if (ntimed < 4) ntimed = 4;
        if (nwarm > 0) launch(nwarm);
        hipEventRecord(e0, s); launch(ntimed); hipEventRecord(e1, s);
        if (e == hipSuccess) e = hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
        if (tflops) *tflops = ms > 0 ? flop_launch * ntimed / (ms * 1e-3) / 1e12 : 0.0;
        if (e == hipSuccess) e = hipGetLastError();
    }
    if (e0) hipEventDestroy(e0);
    if (e1) hipEventDestroy(e1);
    if (d) hipFree(d);
    if (o) hipFree(o);
    return (int)e;
}
