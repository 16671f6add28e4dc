// How do MFMA and VALU instructions overlap on one SIMD? Each wave loops over {1 MFMA 32x32x16 bf16 (or two 16x16x32), NV independent VALU ops};
// waves per SIMD = 1, 2, 4. Prints cycles per slot at a nominal 2.4 GHz. Zero operands: no power cap.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(2))) float f2;

template <int NV, int KIND, int SHAPE> // KIND 0: v_fma_f32, 1: v_exp_f32, 2: v_pk_fma_f32; SHAPE 0: 32x32x16, 1: 16x16x32 (two per slot)
__global__ __launch_bounds__(256) void k(float *out, int iters, float seed) {
    bf16x8 a = {}, b = {};
    f32x16 c0 = {}, c1 = {};
    f32x4 d0 = {}, d1 = {}, d2 = {}, d3 = {};
    float v[8];
    for (int i = 0; i < 8; ++i) v[i] = seed + i;
    f2 w[8];
    for (int i = 0; i < 8; ++i) w[i] = f2{seed, seed + i};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if constexpr (SHAPE == 0) {
                if (u) c1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c1, 0, 0, 0);
                else c0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c0, 0, 0, 0);
            } else {
                if (u) { d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d2, 0, 0, 0); d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d3, 0, 0, 0); }
                else { d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d0, 0, 0, 0); d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, d1, 0, 0, 0); }
            }
#pragma unroll
            for (int n = 0; n < NV; ++n) {
                if constexpr (KIND == 0) asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[n & 7]));
                else if constexpr (KIND == 1) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7]));
                else if constexpr (KIND == 3) { if (n == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(v[n & 7])); else asm volatile("v_fma_f32 %0, %0, %0, %0" : "+v"(v[n & 7])); }
                else asm volatile("v_pk_fma_f32 %0, %0, %0, %0" : "+v"(w[n & 7]));
            }
        }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += c0[e] + c1[e];
    for (int e = 0; e < 4; ++e) s += d0[e] + d1[e] + d2[e] + d3[e];
    for (int i = 0; i < 8; ++i) s += v[i] + w[i][0] + w[i][1];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int NV, int KIND, int SHAPE>
void run(float *o, int wps) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wps; // 256 CUs x wps blocks of 4 waves = wps waves per SIMD
    k<NV, KIND, SHAPE><<<blocks, 256>>>(o, iters, 0.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NV, KIND, SHAPE><<<blocks, 256>>>(o, iters, 0.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma_cycles = (double)wps * iters * 2 * 32;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%s %-6s NV=%2d waves/SIMD=%d: %7.3f ms, %5.1f cycles per slot, MFMA util %.2f\n", SHAPE ? "16x16x32" : "32x32x16",
           KIND == 0 ? "fma" : KIND == 1 ? "exp" : KIND == 3 ? "1exp+fma" : "pk_fma", NV, wps, ms, cyc / ((double)wps * iters * 2), mfma_cycles / cyc);
}
#define ALLW(NV, KIND, SHAPE) run<NV, KIND, SHAPE>(o, 1); run<NV, KIND, SHAPE>(o, 2); run<NV, KIND, SHAPE>(o, 4);
int main() {
    float *o; hipMalloc(&o, 1024 * 256 * 4 * 4);
    ALLW(0, 0, 0) ALLW(2, 0, 0) ALLW(4, 0, 0) ALLW(6, 0, 0) ALLW(8, 0, 0) ALLW(12, 0, 0) ALLW(16, 0, 0)
    ALLW(1, 1, 0) ALLW(2, 1, 0) ALLW(4, 1, 0)
    ALLW(4, 2, 0) ALLW(8, 2, 0)
    ALLW(0, 0, 1) ALLW(4, 0, 1) ALLW(8, 0, 1) ALLW(2, 1, 1)
    // the attention streams' mixes (round 6): per MFMA one v_exp_f32 + (NV - 1) other VALU - D = 128 forward 4.7 VALU / MFMA,
    // D = 64 forward 9.3, D = 64 dK/dV 4.4: what would a SECOND wave per SIMD buy? (zero operands: no power cap; nominal 2.4 GHz)
    printf("---- one v_exp_f32 + (NV - 1) v_fma_f32 per MFMA\n");
    ALLW(4, 3, 0) ALLW(5, 3, 0) ALLW(9, 3, 0) ALLW(10, 3, 0)
    return 0;
}
