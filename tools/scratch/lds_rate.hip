// LDS read throughput per CU: ds_read_b128 vs ds_read_b64_tr_b16 vs ds_read_b64, conflict-free addresses, 1 / 2 / 4 waves per SIMD.
// Prints bytes per clock per CU at a nominal 2.4 GHz.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __attribute__((ext_vector_type(4))) int i4;
typedef __attribute__((ext_vector_type(2))) int i2;

template <int KIND> // 0: b128, 1: b64_tr_b16, 2: b64
__global__ __launch_bounds__(256) void k(int *out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    for (int i = threadIdx.x; i < 8192; i += 256) ((int *)smem)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const unsigned base = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem + wid * 4096;
    // b128: lane i reads 16 B at i * 16 (a contiguous 1 KiB: conflict-free). b64 / tr: lane i reads 8 B at i * 8.
    const unsigned a = base + (KIND == 0 ? lane * 16 : lane * 8);
    i4 acc4 = {0, 0, 0, 0};
    i2 acc2 = {0, 0};
    for (int it = 0; it < iters; ++it) {
        if constexpr (KIND == 0) {
            i4 r0, r1, r2, r3;
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3) : "v"(a) : "memory");
            acc4 += r0 + r1 + r2 + r3;
        } else if constexpr (KIND == 1) {
            i2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64_tr_b16 %0, %8\n\tds_read_b64_tr_b16 %1, %8 offset:512\n\tds_read_b64_tr_b16 %2, %8 offset:1024\n\tds_read_b64_tr_b16 %3, %8 offset:1536\n\t"
                         "ds_read_b64_tr_b16 %4, %8 offset:2048\n\tds_read_b64_tr_b16 %5, %8 offset:2560\n\tds_read_b64_tr_b16 %6, %8 offset:3072\n\tds_read_b64_tr_b16 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a) : "memory");
            acc2 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        } else {
            i2 r0, r1, r2, r3, r4, r5, r6, r7;
            asm volatile("ds_read_b64 %0, %8\n\tds_read_b64 %1, %8 offset:512\n\tds_read_b64 %2, %8 offset:1024\n\tds_read_b64 %3, %8 offset:1536\n\t"
                         "ds_read_b64 %4, %8 offset:2048\n\tds_read_b64 %5, %8 offset:2560\n\tds_read_b64 %6, %8 offset:3072\n\tds_read_b64 %7, %8 offset:3584\n\ts_waitcnt lgkmcnt(0)"
                         : "=&v"(r0), "=&v"(r1), "=&v"(r2), "=&v"(r3), "=&v"(r4), "=&v"(r5), "=&v"(r6), "=&v"(r7) : "v"(a) : "memory");
            acc2 += r0 + r1 + r2 + r3 + r4 + r5 + r6 + r7;
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc4[0] + acc4[1] + acc4[2] + acc4[3] + acc2[0] + acc2[1];
}
template <int KIND>
void run(int *o, int wps) {
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * wps;
    k<KIND><<<blocks, 256, 32768>>>(o, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<KIND><<<blocks, 256, 32768>>>(o, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double bytes_per_cu = (double)wps * 4 * iters * 4096.0; // each wave reads 4 KiB per iteration
    printf("%-18s waves/SIMD=%d: %7.3f ms, %.1f B/clk/CU (nominal 2.4 GHz)\n", KIND == 0 ? "ds_read_b128" : KIND == 1 ? "ds_read_b64_tr_b16" : "ds_read_b64", wps, ms,
           bytes_per_cu / (ms * 1e-3 * 2.4e9));
}
int main() {
    int *o; hipMalloc(&o, 1024 * 256 * 4);
    for (int w = 1; w <= 4; w *= 2) { run<0>(o, w); run<1>(o, w); run<2>(o, w); }
    return 0;
}
