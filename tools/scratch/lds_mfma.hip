// MFMA fed from LDS: per wave, groups of 4 ds_read_b128 (a 32 x 16 A fragment each) feeding R MFMAs per fragment (R = 1: 1 KiB of
// LDS per MFMA, the attention kernels' ratio; R = 2: the GEMM's), software-pipelined one group ahead, 1 / 2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) short s16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

__device__ __forceinline__ void rd4(unsigned a, int off, s16x8 (&f)[4]) {
    asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                 : "=&v"(f[0]), "=&v"(f[1]), "=&v"(f[2]), "=&v"(f[3]) : "v"(a + off) : "memory");
}
template <int N> __device__ __forceinline__ void w4(s16x8 (&f)[4]) {
    asm volatile("s_waitcnt lgkmcnt(%c4)" : "+v"(f[0]), "+v"(f[1]), "+v"(f[2]), "+v"(f[3]) : "n"(N) : "memory");
}
template <int R, int TR>
__global__ __launch_bounds__(512) void k(float *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) ((int *)smem)[i] = 0;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem + (wid & 7) * 8192 + lane * 16;
    bf16x8 b0 = {}, b1 = {};
    f32x16 c[4] = {};
    s16x8 g0[4], g1[4];
    rd4(base, 0, g0);
    for (int it = 0; it < iters; ++it) {
        rd4(base, 4096, g1);
        w4<4>(g0);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g0[i]), b0, c[i], 0, 0, 0);
            if (R == 2) c[(i + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g0[i]), b1, c[(i + 2) & 3], 0, 0, 0);
        }
        rd4(base, 0, g0);
        w4<4>(g1);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            c[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g1[i]), b0, c[i], 0, 0, 0);
            if (R == 2) c[(i + 2) & 3] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, g1[i]), b1, c[(i + 2) & 3], 0, 0, 0);
        }
    }
    w4<0>(g0);
    float s = 0;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 16; ++e) s += c[i][e];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + (float)g0[0][0];
}
template <int R>
void run(float *o, int wps) {
    const int iters = 4000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int threads = 256 * wps;
    k<R, 0><<<256, threads, 65536>>>(o, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<R, 0><<<256, threads, 65536>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double mfma = (double)wps * iters * 8 * R * 32; // pipe cycles per SIMD
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("LDS bytes per MFMA %4d, waves/SIMD=%d: %.3f ms, MFMA util %.2f, LDS %.0f B/clk/CU\n", 1024 / R, wps, ms, mfma / cyc,
           (double)wps * 4 * iters * 8 * 1024.0 / cyc);
}
int main() {
    float *o; hipMalloc(&o, 256 * 512 * 4);
    hipFuncSetAttribute((const void *)k<1, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    hipFuncSetAttribute((const void *)k<2, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
    run<1>(o, 1); run<1>(o, 2); run<2>(o, 1); run<2>(o, 2);
    return 0;
}
