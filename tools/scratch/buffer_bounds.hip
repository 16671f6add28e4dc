// What gfx950's raw-buffer range check covers (for ragged attention tiles): is soffset part of it? does an out-of-range LDS-DMA load
// write zeros or nothing? are out-of-range stores dropped? Prints one line per question.
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdint.h>
typedef __attribute__((ext_vector_type(4))) int i32x4;

__global__ void probe(const uint32_t *src, uint32_t nrec, uint32_t *out, uint32_t *dst, uint32_t dst_nrec) {
    __shared__ __attribute__((aligned(16))) uint32_t lds[64 * 4 * 4];
    const int lane = threadIdx.x;
    for (int i = lane; i < 64 * 4 * 4; i += 64) lds[i] = 0xAAAAAAAAu;
    __syncthreads();
    __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void *)src, 0, nrec, 0x00020000);
    i32x4 rsa = {__builtin_amdgcn_readfirstlane((int)(uintptr_t)src), __builtin_amdgcn_readfirstlane((int)((uintptr_t)src >> 32)), __builtin_amdgcn_readfirstlane((int)nrec), 0x00020000};
    // (a) plain loads: voffset = 16 * lane (in range for lane < nrec / 16), soffset 0
    uint32_t a = __builtin_amdgcn_raw_buffer_load_b32(rs, 16 * lane, 0, 0);
    // (b) voffset small, soffset pushes past the end: lanes read at 4 * lane + soffset
    uint32_t b = __builtin_amdgcn_raw_buffer_load_b32(rs, 4 * lane, (int)nrec, 0);
    // (c) voffset past the end by itself
    uint32_t c = __builtin_amdgcn_raw_buffer_load_b32(rs, (int)nrec + 4 * lane, 0, 0);
    out[lane] = a; out[64 + lane] = b; out[128 + lane] = c;
    // (d) LDS-DMA, 16 bytes per lane: lanes whose voffset + 16 > nrec are out of range
    const unsigned ldsb = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)lds;
    unsigned voff = 16 * lane;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, 0 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(ldsb), "v"(voff), "s"(rsa) : "memory", "m0");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[192 + 4 * lane + j] = lds[4 * lane + j];
    // (e) LDS-DMA with soffset pushing past the end (second LDS quarter)
    unsigned voff2 = 16 * (lane & 3);
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tbuffer_load_dwordx4 %1, %2, %3 offen lds\n\ts_waitcnt vmcnt(0)" : : "s"(ldsb + 1024), "v"(voff2), "s"(rsa), "s"(nrec) : "memory", "m0");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[448 + 4 * lane + j] = lds[256 + 4 * lane + j];
    // (f) stores: 16 bytes per lane into dst through a descriptor of dst_nrec bytes
    __amdgpu_buffer_rsrc_t rd = __builtin_amdgcn_make_buffer_rsrc((void *)dst, 0, dst_nrec, 0x00020000);
    i32x4 val = {(int)(0x1000 + lane), (int)(0x2000 + lane), (int)(0x3000 + lane), (int)(0x4000 + lane)};
    __builtin_amdgcn_raw_buffer_store_b128(val, rd, 16 * lane, 0, 0);
}

int main() {
    const int N = 4096;
    uint32_t *h = (uint32_t *)malloc(N * 4), *src, *out, *dst;
    for (int i = 0; i < N; ++i) h[i] = 0x55000000u + i;
    hipMalloc(&src, N * 4); hipMalloc(&out, 1024 * 4); hipMalloc(&dst, 64 * 16);
    hipMemcpy(src, h, N * 4, hipMemcpyHostToDevice);
    hipMemset(out, 0xEE, 1024 * 4); hipMemset(dst, 0xDD, 64 * 16);
    const uint32_t nrec = 40 * 16 + 8;   // lanes 0..39 wholly inside, lane 40 straddles (8 of its 16 bytes inside), 41.. outside
    probe<<<1, 64>>>(src, nrec, out, dst, 40 * 16 + 8);
    hipDeviceSynchronize();
    uint32_t o[1024], d[256];
    hipMemcpy(o, out, sizeof(o), hipMemcpyDeviceToHost); hipMemcpy(d, dst, sizeof(d), hipMemcpyDeviceToHost);
    printf("(a) b32 load, voffset: lane 39 %08x (in) lane 40 %08x (in: 4 of 8 bytes) lane 41 %08x (out)\n", o[39], o[40], o[41]);
    printf("(b) b32 load, voffset 4*lane + soffset = num_records: lane 0 %08x lane 5 %08x  [data = 55000%03x.. would mean soffset is NOT range-checked]\n", o[64], o[69], (nrec / 4));
    printf("(c) b32 load, voffset = num_records + 4*lane: lane 0 %08x lane 5 %08x\n", o[128], o[133]);
    printf("(d) LDS-DMA x4: lane 39 %08x %08x %08x %08x | lane 40 (straddles) %08x %08x %08x %08x | lane 41 (out) %08x %08x %08x %08x   [aaaaaaaa = not written]\n",
           o[192 + 156], o[192 + 157], o[192 + 158], o[192 + 159], o[192 + 160], o[192 + 161], o[192 + 162], o[192 + 163], o[192 + 164], o[192 + 165], o[192 + 166], o[192 + 167]);
    printf("(e) LDS-DMA x4 with soffset = num_records: lane 0 %08x %08x lane 1 %08x\n", o[448], o[449], o[452]);
    printf("(f) store x4 through %d-byte descriptor: lane 39 %08x %08x %08x %08x | lane 40 %08x %08x %08x %08x | lane 41 %08x %08x   [dddddddd = dropped]\n", 40 * 16 + 8,
           d[156], d[157], d[158], d[159], d[160], d[161], d[162], d[163], d[164], d[165]);
    return 0;
}
