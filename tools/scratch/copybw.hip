// micro-benchmark: float4 copy variants (diagnostic, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ __launch_bounds__(256) void copy_one(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) b[i] = a[i];
}
template <int U> __global__ __launch_bounds__(256) void copy_blk(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) b[base + u * 256] = v[u];
}
template <int U> __global__ __launch_bounds__(256) void copy_stride(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    const size_t stride = (size_t)gridDim.x * 256 * U;
    for (size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x; base < n; base += stride) {
        float4 v[U];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) v[u] = a[base + u * 256];
#pragma unroll
        for (int u = 0; u < U; ++u) if (base + u * 256 < n) b[base + u * 256] = v[u];
    }
}
template <int U> __global__ __launch_bounds__(256) void copy_blk_nt(const float4 *__restrict__ a, float4 *__restrict__ b, size_t n) {
    size_t base = (size_t)blockIdx.x * 256 * U + threadIdx.x;
    float4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { typedef float v4f __attribute__((ext_vector_type(4))); v4f t = __builtin_nontemporal_load((const v4f *)(a + base + u * 256)); v[u] = make_float4(t.x, t.y, t.z, t.w); }
#pragma unroll
    for (int u = 0; u < U; ++u) if (base + u * 256 < n) { typedef float v4f __attribute__((ext_vector_type(4))); v4f t = {v[u].x, v[u].y, v[u].z, v[u].w}; __builtin_nontemporal_store(t, (v4f *)(b + base + u * 256)); }
}
template <typename F> float timeit(F f) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; ++i) f();
    hipEventRecord(e0);
    for (int i = 0; i < 20; ++i) f();
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); return ms / 20;
}
int main() {
    const size_t bytes = (size_t)1 << 30, n = bytes / 16;
    float4 *a, *b; CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 1, bytes)); CK(hipMemset(b, 0, bytes));
    auto rep = [&](const char *name, float ms) { printf("%-28s %7.3f ms  %7.1f GB/s (read + write)\n", name, ms, 2.0 * bytes / ms / 1e6); };
    rep("one float4 / thread", timeit([&] { copy_one<<<(unsigned)((n + 255) / 256), 256>>>(a, b, n); }));
    rep("block 4 / thread", timeit([&] { copy_blk<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); }));
    rep("block 8 / thread", timeit([&] { copy_blk<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); }));
    rep("block 4 nt", timeit([&] { copy_blk_nt<4><<<(unsigned)((n + 1023) / 1024), 256>>>(a, b, n); }));
    rep("block 8 nt", timeit([&] { copy_blk_nt<8><<<(unsigned)((n + 2047) / 2048), 256>>>(a, b, n); }));
    for (int g : {1024, 2048, 4096, 8192, 16384}) {
        char nm[64];
        snprintf(nm, 64, "grid-stride 4, %d blocks", g); rep(nm, timeit([&] { copy_stride<4><<<g, 256>>>(a, b, n); }));
        snprintf(nm, 64, "grid-stride 8, %d blocks", g); rep(nm, timeit([&] { copy_stride<8><<<g, 256>>>(a, b, n); }));
    }
    rep("hipMemcpyDtoD", timeit([&] { hipMemcpyAsync(b, a, bytes, hipMemcpyDeviceToDevice, 0); }));
    return 0;
}
