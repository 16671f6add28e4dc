// GEMM for gfx950: C[M,N] = alpha * op(A) * op(B) + beta * C (row-major), + fused row-bias epilogue.
//
// Replaces src/device/gemm_kernel.cu:8-38, whose arithmetic is an un-vendored CUTLASS SIMT GEMM
// (src/device/launcher_cuda.h:537-614). Hand-written for CDNA4:
//   f32      v_mfma_f32_32x32x2_f32 (exact f32 fma chain), 128x128x16 block tile (64x64x16 for small grids), LDS tiles
//            kept k-major so every fragment read is a conflict-free ds_read_b32.
//   bf16/f16 large grids: 256x256x64 block tile, v_mfma_f32_16x16x32, LDS-DMA staging - an 8-wave form (two waves per SIMD
//            half a phase apart) and a 4-wave form (one wave per SIMD, 256 in-place AGPR accumulators);
//            smaller grids: 128x128x64 block tile, v_mfma_f32_32x32x16, double-buffered LDS, one barrier per K tile.
//   Every kernel consumes all four op(A) / op(B) layouts in place: an operand whose contraction dim is the strided one is
//   staged as it lies in memory and read with ds_read_b64_tr_b16 - no re-layout pass, no workspace.
//   f64      v_mfma_f64_16x16x4_f64, 64x64x16 block tile (the reference's only GEMM test is f64: 123x457x234,
//            test/test_gemm.py:9-17 - ragged, so at the C ABI it takes the fallback below; the operator pads it).
//   ragged shapes: a plain LDS-tiled FMA kernel.
#include <stdio.h>
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace kf {

typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(8))) _Float16 f16x8;
typedef __attribute__((ext_vector_type(16))) float f32x16;

struct GemmArgs {
    const void *A, *B;
    void *C;
    const void *bias;
    int64_t M, N, K, lda, ldb, ldc;
    float alpha, beta;
    int epilogue;
    int group_m; // tile rows per group of the grouped tile order (0: row-major)
    // element-wise epilogue operands (kf_gemm_ex): C = (alpha AB + beta C + bias) o mul + add; aux receives the value in brackets
    const void *mul, *add;
    void *aux;
    int64_t ldmul, ldadd, ldaux;
    // split-K (128-tile 16-bit kernel on grids far below one round of CUs): slice s of `split` accumulates K tiles
    // [s nt / split, (s + 1) nt / split) and stores its raw f32 tile to part[s][M][N]; gemm_splitk_fold_kernel finishes
    int split;
    float *part;
    // 16-bit operands, FLOAT output (and float beta C): the accumulators are f32 already - a weight gradient that is going to be summed
    // over ranks, or accumulated over micro-batches, leaves without the 16-bit rounding (kf_gemm_ex: epilogue.c_f32, kf_gemm_problem.c_f32)
    int c_f32;
};

// XCD-aware remap: consecutive logical tile ids land on the same XCD (its own 4 MiB L2) so
// neighbouring tiles share operand panels in L2. Bijective for any grid size.
__device__ __forceinline__ uint32_t xcd_remap(uint32_t bid, uint32_t nwg) {
    const uint32_t nx = 8, xcd = bid % nx, q = nwg / nx, r = nwg % nx;
    const uint32_t base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + bid / nx;
}

// Grouped tile order: logical ids walk groups of `gm` tile rows column by column, so the q consecutive ids one XCD
// owns (xcd_remap) cover a gm x (q / gm) rectangle of C: gm A panels + q / gm B panels through that XCD's L2 instead of
// one A panel + a whole row of B panels.
__device__ __forceinline__ void grouped_tile(uint32_t id, uint32_t tiles_m, uint32_t tiles_n, uint32_t gm, uint32_t &tm, uint32_t &tn) {
    if (gm == 0) {
        tm = id / tiles_n;
        tn = id % tiles_n;
        return;
    }
    const uint32_t per = gm * tiles_n, grp = id / per, first = grp * gm;
    const uint32_t rows = min(gm, tiles_m - first), in = id - grp * per;
    tm = first + in % rows;
    tn = in / rows;
}

// ------------------------------------------------------------------------------------------
// generic fallback: any shape, any layout, f32/f64/bf16/f16; 32x32 tile, 256 threads (2x2 per lane)
// ------------------------------------------------------------------------------------------
// two f32 -> one word of two 16-bit values, round-to-nearest-even: bf16 with the hardware's packed converter (one instruction; the
// software form is ~10 per value, and the 4-wave kernel's epilogue converts 256 values per lane - 5 us of a 105 us launch at 4096^3)
template <bool BF>
__device__ __forceinline__ uint32_t g_pack2(float lo, float hi) {
    if constexpr (BF) return f32x2_to_bf16x2_hw(lo, hi);
    else return (uint32_t)f32_to_f16(lo).x | ((uint32_t)f32_to_f16(hi).x << 16);
}

template <typename T> struct GAcc { using type = float; };
template <> struct GAcc<double> { using type = double; };
template <typename T> __device__ __forceinline__ typename GAcc<T>::type g_load(const T *p) { return (typename GAcc<T>::type)(*p); }
template <> __device__ __forceinline__ float g_load<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float g_load<f16_t>(const f16_t *p) { return f16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void g_store(T *p, typename GAcc<T>::type v) { *p = (T)v; }
template <> __device__ __forceinline__ void g_store<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void g_store<f16_t>(f16_t *p, float v) { *p = f32_to_f16(v); }

// the element-wise tail of the epilogue for one output element (scalar sites: generic, f32, f64 and the 128-tile 16-bit kernel)
template <typename T>
__device__ __forceinline__ typename GAcc<T>::type g_epi(const GemmArgs &g, int64_t m, int64_t n, typename GAcc<T>::type v) {
    if (g.aux) g_store<T>((T *)g.aux + m * g.ldaux + n, v);
    if (g.mul) v *= g_load<T>((const T *)g.mul + m * g.ldmul + n);
    if (g.add) v += g_load<T>((const T *)g.add + m * g.ldadd + n);
    return v;
}
// ... and for eight consecutive columns of a 16-bit row (the 256-tile kernels): one 16-byte access per operand when its rows
// are 16-byte aligned, element accesses otherwise
template <bool BF>
__device__ __forceinline__ void h_ld8(const void *base, int64_t ld, int64_t row, int64_t col, float (&f)[8]) {
    const uint16_t *p = (const uint16_t *)base + row * ld + col;
    uint32_t w[4];
    if (ld % 8 == 0 && (uintptr_t)base % 16 == 0) {
        const uint4 q = *(const uint4 *)p;
        w[0] = q.x, w[1] = q.y, w[2] = q.z, w[3] = q.w;
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e) w[e] = (uint32_t)p[2 * e] | ((uint32_t)p[2 * e + 1] << 16);
    }
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const uint16_t o = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
        f[e] = BF ? bf16_to_f32(bf16_t{o}) : f16_to_f32(f16_t{o});
    }
}
template <bool BF>
__device__ __forceinline__ void h_st8(void *base, int64_t ld, int64_t row, int64_t col, const float (&f)[8]) {
    uint16_t *p = (uint16_t *)base + row * ld + col;
    uint32_t w[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        w[e] = g_pack2<BF>(f[2 * e], f[2 * e + 1]);
    }
    if (ld % 8 == 0 && (uintptr_t)base % 16 == 0) {
        *(uint4 *)p = uint4{w[0], w[1], w[2], w[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) p[e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
    }
}
template <bool BF>
__device__ __forceinline__ void h_epi8(const GemmArgs &g, int64_t row, int64_t col, float (&v)[8]) {
    if (g.aux) h_st8<BF>(g.aux, g.ldaux, row, col, v);
    if (g.mul) {
        float x[8];
        h_ld8<BF>(g.mul, g.ldmul, row, col, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] *= x[e];
    }
    if (g.add) {
        float x[8];
        h_ld8<BF>(g.add, g.ldadd, row, col, x);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] += x[e];
    }
}

template <typename T>
__global__ __launch_bounds__(256) void gemm_generic_kernel(const GemmArgs g, int ta, int tb) {
    using A_t = typename GAcc<T>::type;
    constexpr int TS = 32, TK = 32;
    __shared__ A_t As[TK][TS + 1], Bs[TK][TS + 1];
    const T *A = (const T *)g.A, *B = (const T *)g.B;
    T *C = (T *)g.C;
    const int tx = threadIdx.x % 16, ty = threadIdx.x / 16;
    // 1-D grid, N tiles fastest (a 2-D grid's y extent stops at 65535: 2 M flattened rows)
    const int64_t ntn = (g.N + TS - 1) / TS;
    const int64_t m0 = (int64_t)(blockIdx.x / ntn) * TS, n0 = (int64_t)(blockIdx.x % ntn) * TS;
    A_t acc[2][2] = {{0, 0}, {0, 0}};
    for (int64_t k0 = 0; k0 < g.K; k0 += TK) {
        for (int i = threadIdx.x; i < TS * TK; i += 256) {
            const int mm = ta ? i % TS : i / TK, kk = ta ? i / TS : i % TK; // walk the contiguous axis fastest
            const int64_t m = m0 + mm, k = k0 + kk;
            A_t v = 0;
            if (m < g.M && k < g.K) v = g_load<T>(A + (ta ? k * g.lda + m : m * g.lda + k));
            As[kk][mm] = v;
        }
        for (int i = threadIdx.x; i < TS * TK; i += 256) {
            const int nn = tb ? i / TK : i % TS, kk = tb ? i % TK : i / TS;
            const int64_t n = n0 + nn, k = k0 + kk;
            A_t v = 0;
            if (n < g.N && k < g.K) v = g_load<T>(B + (tb ? n * g.ldb + k : k * g.ldb + n));
            Bs[kk][nn] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int kk = 0; kk < TK; ++kk) {
            const A_t a0 = As[kk][ty], a1 = As[kk][ty + 16], b0 = Bs[kk][tx], b1 = Bs[kk][tx + 16];
            acc[0][0] = fma(a0, b0, acc[0][0]);
            acc[0][1] = fma(a0, b1, acc[0][1]);
            acc[1][0] = fma(a1, b0, acc[1][0]);
            acc[1][1] = fma(a1, b1, acc[1][1]);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t m = m0 + ty + 16 * i, n = n0 + tx + 16 * j;
            if (m < g.M && n < g.N) {
                A_t v = (A_t)g.alpha * acc[i][j];
                if constexpr (sizeof(T) == 2) {
                    if (g.c_f32) { // float C behind 16-bit operands
                        float *Cf = (float *)g.C + m * g.ldc + n;
                        if (g.beta != 0.f) v += g.beta * *Cf;
                        if (g.epilogue == KF_EPI_BIAS_ROW) v += g_load<T>((const T *)g.bias + n);
                        *Cf = g_epi<T>(g, m, n, v);
                        continue;
                    }
                }
                if (g.beta != 0.f) v += (A_t)g.beta * g_load<T>(C + m * g.ldc + n);
                if (g.epilogue == KF_EPI_BIAS_ROW) v += g_load<T>((const T *)g.bias + n);
                g_store<T>(C + m * g.ldc + n, g_epi<T>(g, m, n, v));
            }
        }
}

// ------------------------------------------------------------------------------------------
// f32: MFMA 32x32x2, 128x128x16 tiles, all layouts in place
// ------------------------------------------------------------------------------------------
constexpr int F_BM = 128, F_BN = 128, F_BK = 16, F_LD = 132;

// stage one operand tile (T x 16, T = 128 or 64) into registers; KCONTIG: global rows are along the T-axis with k contiguous;
// else global rows are k with the T-axis contiguous. (Two named float4s, not an array: hipcc keeps a by-reference float4[2]
// in scratch; the 64-wide tile uses the first one only.)
template <bool KCONTIG, int T>
__device__ __forceinline__ void f32_load_tile(const float *base, int64_t ld, int64_t x0, int64_t k0, float4 &r0, float4 &r1) {
    const int t = threadIdx.x;
    if constexpr (KCONTIG) {
        const int row = t / 4, kq = t % 4;
        r0 = *(const float4 *)(base + (x0 + row) * ld + k0 + kq * 4);
        if constexpr (T == 128) r1 = *(const float4 *)(base + (x0 + row + 64) * ld + k0 + kq * 4);
    } else if constexpr (T == 128) {
        const int k = t / 32, xq = t % 32;
        r0 = *(const float4 *)(base + (k0 + k) * ld + x0 + xq * 4);
        r1 = *(const float4 *)(base + (k0 + k + 8) * ld + x0 + xq * 4);
    } else {
        const int k = t / 16, xq = t % 16;
        r0 = *(const float4 *)(base + (k0 + k) * ld + x0 + xq * 4);
    }
}
template <bool KCONTIG, int T>
__device__ __forceinline__ void f32_write_tile(float (*s)[F_LD], const float4 &r0, const float4 &r1) {
    const int t = threadIdx.x;
    if constexpr (KCONTIG) {
        const int row = t / 4, kq = t % 4;
        s[kq * 4 + 0][row] = r0.x;
        s[kq * 4 + 1][row] = r0.y;
        s[kq * 4 + 2][row] = r0.z;
        s[kq * 4 + 3][row] = r0.w;
        if constexpr (T == 128) {
            s[kq * 4 + 0][row + 64] = r1.x;
            s[kq * 4 + 1][row + 64] = r1.y;
            s[kq * 4 + 2][row + 64] = r1.z;
            s[kq * 4 + 3][row + 64] = r1.w;
        }
    } else if constexpr (T == 128) {
        const int k = t / 32, xq = t % 32;
        *(float4 *)&s[k][xq * 4] = r0;
        *(float4 *)&s[k + 8][xq * 4] = r1;
    } else {
        const int k = t / 16, xq = t % 16;
        *(float4 *)&s[k][xq * 4] = r0;
    }
}

// T = 128: 2 x 2 MFMA tiles per wave; T = 64 (small problems: four times as many blocks): one
template <bool TA, bool TB, int T>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const GemmArgs g) {
    constexpr int NI = T / 64;
    __shared__ __attribute__((aligned(16))) float As[2][F_BK][F_LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][F_BK][F_LD];
    const float *A = (const float *)g.A, *B = (const float *)g.B;
    float *C = (float *)g.C;
    const uint32_t tiles_n = (uint32_t)(g.N / T);
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / tiles_n) * T, n0 = (int64_t)(tile % tiles_n) * T;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;

    f32x16 acc[NI][NI];
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra0, ra1, rb0, rb1;
    ra1 = rb1 = make_float4(0.f, 0.f, 0.f, 0.f);
    f32_load_tile<!TA, T>(A, g.lda, m0, 0, ra0, ra1);
    f32_load_tile<TB, T>(B, g.ldb, n0, 0, rb0, rb1);
    f32_write_tile<!TA, T>(As[0], ra0, ra1);
    f32_write_tile<TB, T>(Bs[0], rb0, rb1);
    __syncthreads();

    const int nt = (int)(g.K / F_BK);
    const int kl = lane >> 5, xl = lane & 31;
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            f32_load_tile<!TA, T>(A, g.lda, m0, (int64_t)(t + 1) * F_BK, ra0, ra1);
            f32_load_tile<TB, T>(B, g.ldb, n0, (int64_t)(t + 1) * F_BK, rb0, rb1);
        }
#pragma unroll
        for (int ks = 0; ks < F_BK; ks += 2) {
            float a[NI], b[NI];
#pragma unroll
            for (int i = 0; i < NI; ++i) {
                a[i] = As[cur][ks + kl][wr * (T / 2) + i * 32 + xl];
                b[i] = Bs[cur][ks + kl][wc * (T / 2) + i * 32 + xl];
            }
#pragma unroll
            for (int i = 0; i < NI; ++i)
#pragma unroll
                for (int j = 0; j < NI; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) {
            f32_write_tile<!TA, T>(As[cur ^ 1], ra0, ra1);
            f32_write_tile<TB, T>(Bs[cur ^ 1], rb0, rb1);
        }
        __syncthreads();
    }

    // C/D map of the 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < NI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j) {
            const int64_t n = n0 + wc * (T / 2) + j * 32 + xl;
            const float bias = g.epilogue == KF_EPI_BIAS_ROW ? ((const float *)g.bias)[n] : 0.f;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int64_t m = m0 + wr * (T / 2) + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * kl;
                float v = g.alpha * acc[i][j][e];
                if (g.beta != 0.f) v += g.beta * C[m * g.ldc + n];
                C[m * g.ldc + n] = g_epi<float>(g, m, n, v + bias);
            }
        }
}

// ------------------------------------------------------------------------------------------
// f64: v_mfma_f64_16x16x4_f64 (the reference's GEMM is f32 / f64, and its one GEMM test is f64), 64x64x16 block tile,
// 4 waves each owning 32x32 as 2 x 2 MFMA tiles; k-major LDS tiles as in the f32 kernel; all layouts in place.
// A / B fragment: one double per lane, row (column) = lane & 15, k = lane >> 4. C / D: col = lane & 15, row = (lane >> 4) + 4 reg.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) double f64x4;
constexpr int D_T = 64, D_BK = 16, D_LD = 66;

template <bool KCONTIG>
__device__ __forceinline__ void f64_load_tile(const double *base, int64_t ld, int64_t x0, int64_t k0, double2 &r0, double2 &r1) {
    const int t = threadIdx.x;
    if constexpr (KCONTIG) { // rows along the tile axis, k contiguous: thread = (row, 4 k)
        const int row = t / 4, kq = t % 4;
        const double *p = base + (x0 + row) * ld + k0 + kq * 4;
        r0 = *(const double2 *)p;
        r1 = *(const double2 *)(p + 2);
    } else { // k rows, tile axis contiguous: thread = (k, 4 x)
        const int k = t / 16, xq = t % 16;
        const double *p = base + (k0 + k) * ld + x0 + xq * 4;
        r0 = *(const double2 *)p;
        r1 = *(const double2 *)(p + 2);
    }
}
template <bool KCONTIG>
__device__ __forceinline__ void f64_write_tile(double (*s)[D_LD], const double2 &r0, const double2 &r1) {
    const int t = threadIdx.x;
    if constexpr (KCONTIG) {
        const int row = t / 4, kq = t % 4;
        s[kq * 4 + 0][row] = r0.x;
        s[kq * 4 + 1][row] = r0.y;
        s[kq * 4 + 2][row] = r1.x;
        s[kq * 4 + 3][row] = r1.y;
    } else {
        const int k = t / 16, xq = t % 16;
        *(double2 *)&s[k][xq * 4] = r0;
        *(double2 *)&s[k][xq * 4 + 2] = r1;
    }
}

template <bool TA, bool TB>
__global__ __launch_bounds__(256) void gemm_f64_kernel(const GemmArgs g) {
    __shared__ __attribute__((aligned(16))) double As[2][D_BK][D_LD];
    __shared__ __attribute__((aligned(16))) double Bs[2][D_BK][D_LD];
    const double *A = (const double *)g.A, *B = (const double *)g.B;
    double *C = (double *)g.C;
    const uint32_t tiles_n = (uint32_t)(g.N / D_T);
    const uint32_t tile = xcd_remap(blockIdx.x, gridDim.x);
    const int64_t m0 = (int64_t)(tile / tiles_n) * D_T, n0 = (int64_t)(tile % tiles_n) * D_T;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int fi = lane & 15, fk = lane >> 4;

    f64x4 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.0;

    double2 ra0, ra1, rb0, rb1;
    f64_load_tile<!TA>(A, g.lda, m0, 0, ra0, ra1);
    f64_load_tile<TB>(B, g.ldb, n0, 0, rb0, rb1);
    f64_write_tile<!TA>(As[0], ra0, ra1);
    f64_write_tile<TB>(Bs[0], rb0, rb1);
    __syncthreads();

    const int nt = (int)(g.K / D_BK);
    for (int t = 0; t < nt; ++t) {
        const int cur = t & 1;
        if (t + 1 < nt) {
            f64_load_tile<!TA>(A, g.lda, m0, (int64_t)(t + 1) * D_BK, ra0, ra1);
            f64_load_tile<TB>(B, g.ldb, n0, (int64_t)(t + 1) * D_BK, rb0, rb1);
        }
#pragma unroll
        for (int ks = 0; ks < D_BK; ks += 4) {
            double a[2], b[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                a[i] = As[cur][ks + fk][wr * 32 + i * 16 + fi];
                b[i] = Bs[cur][ks + fk][wc * 32 + i * 16 + fi];
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (t + 1 < nt) {
            f64_write_tile<!TA>(As[cur ^ 1], ra0, ra1);
            f64_write_tile<TB>(Bs[cur ^ 1], rb0, rb1);
        }
        __syncthreads();
    }
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int64_t n = n0 + wc * 32 + j * 16 + fi;
            const double bias = g.epilogue == KF_EPI_BIAS_ROW ? ((const double *)g.bias)[n] : 0.0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int64_t m = m0 + wr * 32 + i * 16 + fk + 4 * e;
                double v = (double)g.alpha * acc[i][j][e];
                if (g.beta != 0.f) v += (double)g.beta * C[m * g.ldc + n];
                C[m * g.ldc + n] = g_epi<double>(g, m, n, v + bias);
            }
        }
}

// ------------------------------------------------------------------------------------------
// bf16 / f16: MFMA 32x32x16, 128x128x64 tiles, global_load_lds staging, K-contiguous operands
// ------------------------------------------------------------------------------------------
constexpr int H_BM = 128, H_BN = 128, H_BK = 64;
constexpr int H_TILE_BYTES = H_BM * H_BK * 2; // 16 KiB per operand tile
constexpr int H_STAGES = 4;                   // ring of [A tile | B tile] buffers: 128 KiB

template <bool BF> struct HFrag { using type = f16x8; };
template <> struct HFrag<true> { using type = bf16x8; };

template <bool BF>
__device__ __forceinline__ f32x16 h_mfma(typename HFrag<BF>::type a, typename HFrag<BF>::type b, f32x16 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
}

// LDS image of a [128 rows][64 k] 16-bit tile: 128-B rows, 16-B chunk c of row r stored at chunk
// position c ^ ((r >> 1) & 7) — makes the fragment ds_read_b128 (16 different rows per lane group,
// same logical chunk) hit 16 different 16-B slots of the 256-B bank row.
__device__ __forceinline__ int h_lds_off(int row, int chunk) { return row * 128 + ((chunk ^ ((row >> 1) & 7)) << 4); }

// one wave-instruction moves 8 rows x 128 B; 4 waves x 4 instructions cover the 128-row tile
__device__ __forceinline__ void h_stage(const char *gbase, int64_t ld_bytes, int64_t x0, int64_t k0_bytes, char *lds_tile) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row0 = (wid * 4 + i) * 8;
        const int row = row0 + (lane >> 3), pos = lane & 7;
        const int chunk = pos ^ ((row >> 1) & 7); // inverse of the read-side XOR (an involution)
        const char *src = gbase + (x0 + row) * ld_bytes + k0_bytes + chunk * 16;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + row0 * 128), 16, 0, 0);
    }
}

typedef __attribute__((ext_vector_type(4))) short g_s16x4;
typedef __attribute__((ext_vector_type(8))) short g_s16x8;

// An operand whose contraction dim is the strided one (A stored [K,M], B stored [K,N]) is staged as it lies in memory: a
// [64 k][128 rows / columns] image with 256-B rows, 16-B chunk ch of k-row r at chunk position
// ch ^ (((r & 3) << 2) | ((r >> 2) & 3)) (conflict-free for the 4-row x 16-column blocks ds_read_b64_tr_b16 moves; the same
// image as the attention kernels' K / V tiles). One wave-instruction moves 4 k-rows x 256 B.
__device__ __forceinline__ int h_tr_off(int row, int ch) { return row * 256 + ((ch ^ (((row & 3) << 2) | ((row >> 2) & 3))) << 4); }
__device__ __forceinline__ void h_stage_tr(const char *gbase, int64_t ld_bytes, int64_t x0, int64_t k0, char *lds_tile) {
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row0 = (wid * 4 + i) * 4;
        const int row = row0 + (lane >> 4), pos = lane & 15;
        const int chunk = pos ^ (((row & 3) << 2) | ((row >> 2) & 3));
        const char *src = gbase + (k0 + row) * ld_bytes + (x0 + chunk * 8) * 2;
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                         (__attribute__((address_space(3))) void *)(lds_tile + row0 * 256), 16, 0, 0);
    }
}
// 32 (rows / columns cb..cb+31) x 16 (k) fragment in the STANDARD k order (lane half h holds k = 8 h .. 8 h + 7): two
// transposed reads of k-rows 8 h + {0..3} and 8 h + {4..7}. Per-lane offsets of the two reads for k-step 0; k-step ks adds
// ks * 16 * 256 (the swizzle only sees the row modulo 16).
__device__ __forceinline__ void h_tr_lane_off(int cb, int (&off)[2]) {
    const int lane = threadIdx.x & 63;
    const int gq = lane >> 4, ii = lane & 15, qq = ii >> 2, p = ii & 3, h = gq >> 1;
    const int ch = ((cb + 16 * (gq & 1)) >> 3) + (p >> 1);
#pragma unroll
    for (int second = 0; second < 2; ++second) off[second] = h_tr_off(8 * h + qq + 4 * second, ch) + 8 * (p & 1);
}
template <bool BF, int OFF>
__device__ __forceinline__ typename HFrag<BF>::type h_tr_frag(unsigned a0, unsigned a1) {
    g_s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c4\n\tds_read_b64_tr_b16 %1, %3 offset:%c4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(a0), "v"(a1), "n"(OFF)
                 : "memory");
    g_s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename HFrag<BF>::type, r);
}

template <bool BF, bool TRA, bool TRB>
__global__ __launch_bounds__(256) void gemm_h_kernel(const GemmArgs g) {
    using frag_t = typename HFrag<BF>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[]; // [H_STAGES buffers][A tile | B tile]
    const char *A = (const char *)g.A, *B = (const char *)g.B;
    const uint32_t tiles_n = (uint32_t)(g.N / H_BN);
    const uint32_t id = xcd_remap(blockIdx.x, gridDim.x);
    const uint32_t ntiles = g.split > 1 ? gridDim.x / (uint32_t)g.split : gridDim.x;
    const uint32_t tile = g.split > 1 ? id % ntiles : id, slice = g.split > 1 ? id / ntiles : 0;
    const int64_t m0 = (int64_t)(tile / tiles_n) * H_BM, n0 = (int64_t)(tile % tiles_n) * H_BN;
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wr = wid >> 1, wc = wid & 1;
    const int xl = lane & 31, hl = lane >> 5;
    const int64_t lda_b = g.lda * 2, ldb_b = g.ldb * 2;

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    auto stage_tile = [&](int t, char *buf) __attribute__((always_inline)) {
        if constexpr (TRA) h_stage_tr(A, lda_b, m0, (int64_t)t * H_BK, buf);
        else h_stage(A, lda_b, m0, (int64_t)t * H_BK * 2, buf);
        if constexpr (TRB) h_stage_tr(B, ldb_b, n0, (int64_t)t * H_BK, buf + H_TILE_BYTES);
        else h_stage(B, ldb_b, n0, (int64_t)t * H_BK * 2, buf + H_TILE_BYTES);
    };
    int toffA[2][2], toffB[2][2]; // transposed-read operands: [tile i][first / second read]
    if constexpr (TRA) { h_tr_lane_off(wr * 64, toffA[0]); h_tr_lane_off(wr * 64 + 32, toffA[1]); }
    if constexpr (TRB) { h_tr_lane_off(wc * 64, toffB[0]); h_tr_lane_off(wc * 64 + 32, toffB[1]); }
    const unsigned smem_u = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;

    const int ntk = (int)(g.K / H_BK);
    const int tk0 = g.split > 1 ? (int)((int64_t)slice * ntk / g.split) : 0;          // this block's K tiles: [tk0, tk0 + nt)
    const int nt = (g.split > 1 ? (int)((int64_t)(slice + 1) * ntk / g.split) : ntk) - tk0;
    // A ring of H_STAGES tile buffers, three to four tiles in flight: one tile is 16 MFMAs of 32 cycles per wave - a quarter of a microsecond,
    // far less than an L2 / HBM round trip, and with the tile t + 1 requested only at the top of iteration t every iteration sat out most
    // of that latency (0.72 us per K tile at 2048^3, twice the kernel's LDS bound). A wave issues 8 LDS-DMA operations per tile, so
    // s_waitcnt vmcnt(16) = "everything but the two youngest tiles has landed"; past the end the last tile is fetched again (never
    // read), which keeps that count constant. Every LDS read is inline asm: the compiler would put a vmcnt(0) in front of a read it
    // can see (it cannot tell the ring slots apart). Measured at 2048^3: 34.2 -> 31.7 us (500 -> 540 TFLOP/s). What is left is the
    // operand stream itself: 32 KiB per K tile per CU at 0.65 us = 49 GB/s per CU, 12.6 TB/s over the chip, against the 66-73 GB/s per CU
    // the guide measures as the most LDS-DMA delivers from L2 (MI355X_MICROARCH.md, indexed rows) - a 128^2 tile per CU is bound by its
    // 64 FLOP per staged byte at about 750 TFLOP/s, which is why the large shapes take 256^2 tiles.
    auto clampt = [&](int t) { return tk0 + (t < nt ? t : nt - 1); };
#pragma unroll
    for (int t = 0; t < H_STAGES; ++t) stage_tile(clampt(t), smem + t * 2 * H_TILE_BYTES);
    const int offA[2] = {h_lds_off(wr * 64 + xl, hl), h_lds_off(wr * 64 + 32 + xl, hl)};      // K-contiguous operands: k-step ks flips chunk bits
    const int offB[2] = {h_lds_off(wc * 64 + xl, hl), h_lds_off(wc * 64 + 32 + xl, hl)};      // (ks * 2 + hl) ^ sw = (hl ^ sw) ^ (ks * 2): XOR 32 * ks
    auto rd = [&](frag_t &dst, unsigned addr) __attribute__((always_inline)) { asm volatile("ds_read_b128 %0, %1" : "=v"(dst) : "v"(addr) : "memory"); };
    auto read_step = [&](auto ks_c, unsigned cur_u, frag_t (&a)[2], frag_t (&b)[2]) __attribute__((always_inline)) {
        constexpr int KS = decltype(ks_c)::value;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            if constexpr (TRA) a[i] = h_tr_frag<BF, KS * 16 * 256>(cur_u + toffA[i][0], cur_u + toffA[i][1]);
            else rd(a[i], cur_u + (unsigned)(offA[i] ^ (KS * 32)));
            if constexpr (TRB) b[i] = h_tr_frag<BF, KS * 16 * 256>(cur_u + H_TILE_BYTES + toffB[i][0], cur_u + H_TILE_BYTES + toffB[i][1]);
            else rd(b[i], cur_u + H_TILE_BYTES + (unsigned)(offB[i] ^ (KS * 32)));
        }
    };
    auto landed = [&](frag_t (&a)[2], frag_t (&b)[2]) __attribute__((always_inline)) {
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a[0]), "+v"(a[1]), "+v"(b[0]), "+v"(b[1]) : : "memory");
    };
    auto mma = [&](frag_t (&a)[2], frag_t (&b)[2]) __attribute__((always_inline)) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) acc[i][j] = h_mfma<BF>(a[i], b[j], acc[i][j]);
    };
    frag_t fa[2][2], fb[2][2]; // two fragment sets: k-step ks + 1 is read under the MFMAs of k-step ks
    // The barrier sits in front of a tile's LAST k-step, not its first: by then every wave has received (not just issued) its reads of the
    // tile, so the slot can be handed to tile t + 4 at once, and the next tile's first fragments are read under the four MFMAs that are
    // left - no wave stands at a barrier with an empty MFMA queue, no first read of a tile is waited for in the open.
    asm volatile("s_waitcnt vmcnt(24)\n\ts_barrier" ::: "memory"); // tile 0 has landed for everyone
    read_step(std::integral_constant<int, 0>{}, smem_u, fa[0], fb[0]);
    landed(fa[0], fb[0]);
    for (int t = 0; t < nt; ++t) {
        const unsigned cur_u = smem_u + (unsigned)((t % H_STAGES) * 2 * H_TILE_BYTES), nxt_u = smem_u + (unsigned)(((t + 1) % H_STAGES) * 2 * H_TILE_BYTES);
        read_step(std::integral_constant<int, 1>{}, cur_u, fa[1], fb[1]);
        mma(fa[0], fb[0]);
        landed(fa[1], fb[1]);
        read_step(std::integral_constant<int, 2>{}, cur_u, fa[0], fb[0]);
        mma(fa[1], fb[1]);
        landed(fa[0], fb[0]);
        read_step(std::integral_constant<int, 3>{}, cur_u, fa[1], fb[1]);
        mma(fa[0], fb[0]);
        landed(fa[1], fb[1]);
        asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); // tile t + 1 has landed for everyone; nobody reads tile t any more
        stage_tile(clampt(t + H_STAGES), smem + (t % H_STAGES) * 2 * H_TILE_BYTES);
        read_step(std::integral_constant<int, 0>{}, nxt_u, fa[0], fb[0]);
        mma(fa[1], fb[1]);
        landed(fa[0], fb[0]);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // the re-fetched tail tiles: nothing may still be writing LDS when the workgroup retires

    if (g.split > 1) { // the raw partial tile; alpha, beta, bias and the tail belong to the fold
        float *P = g.part + (int64_t)slice * g.M * g.N;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const int64_t n = n0 + wc * 64 + j * 32 + xl;
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const int64_t m = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl;
                    P[m * g.N + n] = acc[i][j][e];
                }
            }
        return;
    }
    // Round 5 (the C4 tail cost 30 % at 2048^3): the accumulators go through LDS - the ring is free now - and every
    // lane then owns EIGHT consecutive columns of a row: one 16-byte access per operand (C read for beta, the tail's mul / add / aux, C
    // written) instead of sixteen 2-byte ones per accumulator tuple, 16 lanes per 256-byte row. The arithmetic per element is unchanged
    // (alpha acc, + beta C, + bias, tail).
    constexpr int EROW = H_BN + 4; // floats per staged row (16-byte aligned rows, the two lane halves on different banks)
    static_assert(H_BM * EROW * 4 <= H_STAGES * 2 * H_TILE_BYTES, "the staged tile must fit into the ring");
    float *stg = (float *)smem;
    __syncthreads(); // every wave's re-fetched tail tiles have landed (its own vmcnt(0) above) and nobody reads the ring any more
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e)
                stg[(wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * hl) * EROW + wc * 64 + j * 32 + xl] = acc[i][j][e];
    __syncthreads();
    const bool wide4 = g.ldc % 4 == 0 && (uintptr_t)g.C % 16 == 0;
#pragma unroll
    for (int r = 0; r < H_BM * H_BN / 8 / 256; ++r) {
        const int gi = (int)threadIdx.x + 256 * r, row = gi >> 4, c8 = (gi & 15) * 8;
        const int64_t m = m0 + row, n = n0 + c8;
        const float4 lo = *(const float4 *)(stg + row * EROW + c8), hi = *(const float4 *)(stg + row * EROW + c8 + 4);
        float v[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = g.alpha * v[e];
        float *Cf = (float *)g.C + m * g.ldc + n;
        if (g.beta != 0.f) {
            float old[8];
            if (g.c_f32) {
#pragma unroll
                for (int e = 0; e < 8; ++e) old[e] = Cf[e];
            } else {
                h_ld8<BF>(g.C, g.ldc, m, n, old);
            }
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += g.beta * old[e];
        }
        if (g.epilogue == KF_EPI_BIAS_ROW) {
            float bias[8];
            h_ld8<BF>(g.bias, 0, 0, n, bias);
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bias[e];
        }
        h_epi8<BF>(g, m, n, v);
        if (g.c_f32) {
            if (wide4) {
                *(float4 *)Cf = float4{v[0], v[1], v[2], v[3]};
                *(float4 *)(Cf + 4) = float4{v[4], v[5], v[6], v[7]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) Cf[e] = v[e];
            }
        } else {
            h_st8<BF>(g.C, g.ldc, m, n, v);
        }
    }
}

// split-K fold: C = alpha sum_s part[s] + beta C + bias, then the element-wise tail; slices added in slice order (fixed: the result
// does not depend on scheduling). Four consecutive columns per lane, float4 partial loads.
template <bool BF>
__global__ __launch_bounds__(256) void gemm_splitk_fold_kernel(const GemmArgs g) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x, per_row = g.N / 4;
    if (q >= g.M * per_row) return;
    const int64_t m = q / per_row, n = (q - m * per_row) * 4;
    float v[4] = {0.f, 0.f, 0.f, 0.f};
    for (int s = 0; s < g.split; ++s) {
        const float4 p = *(const float4 *)(g.part + ((int64_t)s * g.M + m) * g.N + n);
        v[0] += p.x; v[1] += p.y; v[2] += p.z; v[3] += p.w;
    }
    uint16_t *C = (uint16_t *)g.C + m * g.ldc + n;
    float *Cf = (float *)g.C + m * g.ldc + n;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float t = g.alpha * v[e];
        if (g.beta != 0.f) t += g.beta * (g.c_f32 ? Cf[e] : (BF ? bf16_to_f32(bf16_t{C[e]}) : f16_to_f32(f16_t{C[e]})));
        if (g.epilogue == KF_EPI_BIAS_ROW) {
            const uint16_t bb = ((const uint16_t *)g.bias)[n + e];
            t += BF ? bf16_to_f32(bf16_t{bb}) : f16_to_f32(f16_t{bb});
        }
        if (BF) t = g_epi<bf16_t>(g, m, n + e, t); else t = g_epi<f16_t>(g, m, n + e, t);
        if (g.c_f32) Cf[e] = t;
        else C[e] = (uint16_t)g_pack2<BF>(t, 0.f);
    }
}

// ------------------------------------------------------------------------------------------
// bf16 / f16, large shapes: 256 x 256 x 64 block tile, 8 waves (2 x 4, 128 x 64 per wave), MFMA 16x16x32,
// two waves per SIMD running HALF A PHASE APART: while one group of four waves is in a matrix segment
// (16 MFMAs) the other is in its load segment (fragment ds_reads + one half-tile of LDS-DMA), swapping at
// every raw s_barrier. A K tile is cut into four half-tiles of 128 rows x 64 k, chosen so that each is read
// in exactly ONE phase by every wave:
//     HA0 = rows {wr*128 + 0..63}   (A0 fragments, phase 0)      HB0 = cols {wc*64 + 0..31}  (B0, phase 0)
//     HA1 = rows {wr*128 + 64..127} (A1 fragments, phase 1)      HB1 = cols {wc*64 + 32..63} (B1, phase 2)
// Quadrant order (A0,B0) (A1,B0) (A1,B1) (A0,B1): all four fragment sets stay in registers, so a half-tile's
// LDS region is free one phase after its single read and is re-staged 5-6 phases before its next read:
//     phase 0: stage HA1(T+1)   phase 1: HB1(T+1)   phase 2: HA0(T+2)   phase 3: HB0(T+2)
// Every wave issues 2 DMA operations per phase, so ONE counted wait per phase, s_waitcnt vmcnt(8) (the four
// youngest half-tiles stay in flight), retires exactly what the next phase reads; the barrier publishes it.
// LDS: 2 K tiles x 4 half-tiles x (16 KiB + 256 B) = 130 KiB.
// Operand layouts (TRA / TRB): a K-contiguous operand (A [M,K], B stored [N,K]) is staged as a [128 rows][128 B]
// image and its 16 x 32 fragments are one ds_read_b128 each. An operand whose contraction dim is the STRIDED one
// (A stored [K,M] for A^T B, B stored [K,N] for plain A B) is staged as it lies in memory, a [64 k][256 B] image of the
// same 128 rows / columns, and its fragments are read with two ds_read_b64_tr_b16 (the hardware transposes 4 k x 16
// lanes), so no operand is ever re-laid out in HBM. That image keeps the reads conflict-free with an XOR of the 16-B
// chunk index by (k & 3) << 2 applied on the DMA source, plus 32 B of padding after every PAIR of 4-row groups
// (a DMA instruction's 1 KiB lands contiguously, so padding can only sit between instructions); the k-step and
// the lo / hi row quad of a fragment are then immediates (8192 + 128, 1024) off one per-tile address.
// ------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(4))) float f32x4;
constexpr int G_BM = 256, G_BN = 256, G_BK = 64, G_NT = 512;
constexpr int G_HALF = 128 * G_BK * 2 + 256; // 16 KiB + the transposed image's padding (32 B x 7, rounded up)
constexpr int G_TILE = 4 * G_HALF;          // HA0 | HB0 | HA1 | HB1
constexpr int G_LDS = 2 * G_TILE;           // 130 KiB


// one 16 (m or n) x 32 (k) fragment out of a [k][256 B] image: rows 8*(lane>>4) + {0..3} and + {4..7} of the k-step
template <bool BF, int OFF>
__device__ __forceinline__ typename HFrag<BF>::type g_tr_frag(unsigned addr) {
    g_s16x4 lo, hi;
    asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%c3\n\tds_read_b64_tr_b16 %1, %2 offset:%c4"
                 : "=&v"(lo), "=&v"(hi)
                 : "v"(addr), "n"(OFF), "n"(OFF + 1024)
                 : "memory");
    g_s16x8 r;
    r[0] = lo[0]; r[1] = lo[1]; r[2] = lo[2]; r[3] = lo[3];
    r[4] = hi[0]; r[5] = hi[1]; r[6] = hi[2]; r[7] = hi[3];
    return __builtin_bit_cast(typename HFrag<BF>::type, r);
}

template <bool BF>
__device__ __forceinline__ f32x4 g_mfma16(typename HFrag<BF>::type a, typename HFrag<BF>::type b, f32x4 c) {
    if constexpr (BF)
        return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
    else
        return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
}

template <bool BF, bool TRA, bool TRB>
__global__ __launch_bounds__(G_NT, 2) void gemm_h256_kernel(const GemmArgs g) {
    using frag_t = typename HFrag<BF>::type;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const int wr = __builtin_amdgcn_readfirstlane(wid) >> 2, wc = __builtin_amdgcn_readfirstlane(wid) & 3;
    const uint32_t tiles_n = (uint32_t)(g.N / G_BN);
    uint32_t tm, tn;
    grouped_tile(xcd_remap(blockIdx.x, gridDim.x), (uint32_t)(g.M / G_BM), tiles_n, (uint32_t)g.group_m, tm, tn);
    const int64_t m0 = (int64_t)tm * G_BM, n0 = (int64_t)tn * G_BN;
    const int nt = (int)(g.K / G_BK);

    // ---- LDS-DMA source pointers. K-contiguous operand: wave w moves rows (2w + i) * 8 .. + 7 (i = 0, 1) of every
    // half-tile. Transposed-read operand: wave w moves k rows (2w + i) * 4 .. + 3, 256 B (128 rows / columns) each.
    const char *srcA[2][2], *srcB[2][2]; // [half 0/1][i]
    int ldsoffA[2], ldsoffB[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int hr = (wid * 2 + i) * 8 + (lane >> 3), pos = lane & 7;
        const int chunk = pos ^ ((hr >> 1) & 7);
        const int kr = (wid * 2 + i) * 4 + (lane >> 4);                // transposed image: k row of this lane
        const int j0 = ((lane & 15) ^ ((kr & 3) << 2)) * 8;            // ... and the first of its 8 rows / columns
        if constexpr (TRA) {
            ldsoffA[i] = (wid * 2 + i) * 1024 + 32 * wid;
            srcA[0][i] = (const char *)g.A + ((int64_t)kr * g.lda + m0 + j0) * 2; // HA0 = rows 0..127: whole 256-B lines per k
            srcA[1][i] = srcA[0][i] + 128 * 2;
        } else {
            ldsoffA[i] = (wid * 2 + i) * 8 * 128;
            const int64_t arow = m0 + (hr >> 6) * 128 + (hr & 63);
            srcA[0][i] = (const char *)g.A + arow * g.lda * 2 + chunk * 16;
            srcA[1][i] = srcA[0][i] + 64 * g.lda * 2;
        }
        if constexpr (TRB) {
            ldsoffB[i] = (wid * 2 + i) * 1024 + 32 * wid;
            srcB[0][i] = (const char *)g.B + ((int64_t)kr * g.ldb + n0 + j0) * 2; // HB0 = columns 0..127
            srcB[1][i] = srcB[0][i] + 128 * 2;
        } else {
            ldsoffB[i] = (wid * 2 + i) * 8 * 128;
            const int64_t brow = n0 + (hr >> 5) * 64 + (hr & 31);
            srcB[0][i] = (const char *)g.B + brow * g.ldb * 2 + chunk * 16;
            srcB[1][i] = srcB[0][i] + 32 * g.ldb * 2;
        }
    }
    const int64_t kstepA = TRA ? (int64_t)G_BK * g.lda * 2 : (int64_t)G_BK * 2;
    const int64_t kstepB = TRB ? (int64_t)G_BK * g.ldb * 2 : (int64_t)G_BK * 2;
    // which: 0 HA0, 1 HB0, 2 HA1, 3 HB1 (also the slot inside a tile buffer)
    auto stage = [&](int which, int kt) {
        const int64_t ktc = kt < nt ? kt : nt - 1; // past the end: re-fetch the last tile (never read)
        char *dst = smem + (kt & 1) * G_TILE + which * G_HALF;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const char *src = (which & 1) ? srcB[which >> 1][i] + ktc * kstepB : srcA[which >> 1][i] + ktc * kstepA;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src,
                                             (__attribute__((address_space(3))) void *)(dst + ((which & 1) ? ldsoffB[i] : ldsoffA[i])), 16, 0, 0);
        }
    };

    // ---- fragment read offsets: row r of a 16-row tile, k-chunk (ks*4 + lane>>4), XOR swizzle on (row>>1)&7
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & 7;
    const int offk0 = fr * 128 + (((0 + fg) ^ sw) << 4), offk1 = fr * 128 + (((4 + fg) ^ sw) << 4);
    const int abase = TRA ? 0 : wr * 64 * 128, bbase = TRB ? 0 : wc * 32 * 128; // this wave's rows inside HA* / HB*
    // transposed image: per-lane address of tile 0's low row quad; tile t is this XOR (16-B chunk index ^ 2 t), see above
    const int tq = fr >> 2;
    const int lbT = fg * 2048 + tq * 256 + (((((fr & 3) >> 1)) ^ (tq << 2)) << 4) + 8 * (fr & 1);
    const int pbT = 32 * fg;
    const unsigned smem_u = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;
    const int ecA = wr * 128, ecB = wc * 64; // (first chunk of this wave's columns) << 4

    f32x4 acc[4][8]; // [n-tile 0..3][m-tile 0..7]: D = B_frag x A_frag, i.e. C^T tiles (rows = n on registers)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[i][j][e] = 0.f;
    frag_t a0[4][2], a1[4][2], b0[2][2], b1[2][2];

    // ---- prologue: the issues of "phases -6 .. -1"
    stage(0, 0); stage(1, 0); stage(2, 0); stage(3, 0); stage(0, 1); stage(1, 1);
    asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    if (wr == 1) __builtin_amdgcn_s_barrier(); // waves 4-7 run half a phase behind waves 0-3
    asm volatile("" ::: "memory");

#define G_SEG_END()                            \
    asm volatile("" ::: "memory");             \
    __builtin_amdgcn_sched_barrier(0);         \
    __builtin_amdgcn_s_barrier();              \
    __builtin_amdgcn_sched_barrier(0);         \
    asm volatile("" ::: "memory");

    // fragment loads of one half-tile, NTL tiles of 16 rows / columns. PART 0 runs in the load segment: both k-steps
    // of a ds_read_b128 operand, k-step 0 of a transposed-read operand; PART 1 (transposed-read operands only) issues
    // k-step 1; PART 2 both. The transposed A operand (16 + 16 read instructions per half-tile) is split: k-step 1 is
    // issued at the top of the wave's own matrix segment, under the k-step-0 MFMAs, which keeps the load segment as
    // short as the other group's matrix segment; the B operand (8 + 8) loads whole in the load segment.
#define G_LOAD(FR, NTL, HB, TR, EC, PART)                                                             \
    _Pragma("unroll") for (int t = 0; t < NTL; ++t) {                                                 \
        if constexpr (TR) {                                                                           \
            const unsigned x = (unsigned)((lbT ^ ((EC) + t * 32)) + pbT) + smem_u + (unsigned)((HB) - smem); \
            if constexpr ((PART) == 0 || (PART) == 2) FR[t][0] = g_tr_frag<BF, 0>(x);                     \
            if constexpr ((PART) == 1 || (PART) == 2) FR[t][1] = g_tr_frag<BF, 8192 + 128>(x);            \
        } else if constexpr ((PART) == 0 || (PART) == 2) {                                                \
            FR[t][0] = *(const frag_t *)((HB) + t * 2048 + offk0);                                    \
            FR[t][1] = *(const frag_t *)((HB) + t * 2048 + offk1);                                    \
        }                                                                                             \
    }
// the asm-issued transposed reads are invisible to the compiler's wait insertion: counted lgkmcnt waits by hand,
// placed AFTER the barrier (the reads' latency overlaps the barrier wait, as the compiler arranges for ds_read_b128)
#define G_LGKM(N)                                                        \
    __builtin_amdgcn_sched_barrier(0);                                   \
    asm volatile("s_waitcnt lgkmcnt(%c0)" ::"n"(N) : "memory");          \
    __builtin_amdgcn_sched_barrier(0);
#define G_MFMA(KS, ACC, NOFF, MOFF, BF_, AF_)                                                           \
    _Pragma("unroll") for (int n = 0; n < 2; ++n)                                                     \
        _Pragma("unroll") for (int m = 0; m < 4; ++m)                                                 \
            ACC[NOFF + n][MOFF + m] = g_mfma16<BF>(BF_[n][KS], AF_[m][KS], ACC[NOFF + n][MOFF + m]);
    constexpr int NLA = 8; // k-step-1 read instructions of a transposed A half-tile (4 tiles x lo, hi)
    constexpr bool SPLITB = TRA && TRB; // both operands transposed: B's k-step 1 moves into the matrix segment too

    for (int kt = 0; kt < nt; ++kt) {
        const char *buf = smem + (kt & 1) * G_TILE;
        const char *ha0 = buf + abase, *hb0 = buf + G_HALF + bbase, *ha1 = buf + 2 * G_HALF + abase, *hb1 = buf + 3 * G_HALF + bbase;
        // ---------------- phase 0: (A0, B0) ----------------
        G_LOAD(b0, 2, hb0, TRB, ecB, (SPLITB ? 0 : 2))
        G_LOAD(a0, 4, ha0, TRA, ecA, 0)
        stage(2, kt + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G_SEG_END()
        if constexpr (TRA) {
            if constexpr (SPLITB) { G_LOAD(b0, 2, hb0, TRB, ecB, 1) }
            G_LOAD(a0, 4, ha0, TRA, ecA, 1)
            G_LGKM(NLA + (SPLITB ? 4 : 0))
        } else if constexpr (TRB) {
            G_LGKM(0)
        }
        __builtin_amdgcn_s_setprio(1);
        G_MFMA(0, acc, 0, 0, b0, a0)
        if constexpr (TRA) { G_LGKM(0) }
        G_MFMA(1, acc, 0, 0, b0, a0)
        __builtin_amdgcn_s_setprio(0);
        G_SEG_END()
        // ---------------- phase 1: (A1, B0) ----------------
        G_LOAD(a1, 4, ha1, TRA, ecA, 0)
        stage(3, kt + 1);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G_SEG_END()
        if constexpr (TRA) {
            G_LOAD(a1, 4, ha1, TRA, ecA, 1)
            G_LGKM(NLA)
        }
        __builtin_amdgcn_s_setprio(1);
        G_MFMA(0, acc, 0, 4, b0, a1)
        if constexpr (TRA) { G_LGKM(0) }
        G_MFMA(1, acc, 0, 4, b0, a1)
        __builtin_amdgcn_s_setprio(0);
        G_SEG_END()
        // ---------------- phase 2: (A1, B1) ----------------
        G_LOAD(b1, 2, hb1, TRB, ecB, (SPLITB ? 0 : 2))
        stage(0, kt + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G_SEG_END()
        if constexpr (SPLITB) {
            G_LOAD(b1, 2, hb1, TRB, ecB, 1)
            G_LGKM(4)
        } else if constexpr (TRB) {
            G_LGKM(0)
        }
        __builtin_amdgcn_s_setprio(1);
        G_MFMA(0, acc, 2, 4, b1, a1)
        if constexpr (SPLITB) { G_LGKM(0) }
        G_MFMA(1, acc, 2, 4, b1, a1)
        __builtin_amdgcn_s_setprio(0);
        G_SEG_END()
        // ---------------- phase 3: (A0, B1): no fragment reads ----------------
        stage(1, kt + 2);
        asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        G_SEG_END()
        __builtin_amdgcn_s_setprio(1);
        G_MFMA(0, acc, 2, 0, b1, a0)
        G_MFMA(1, acc, 2, 0, b1, a0)
        __builtin_amdgcn_s_setprio(0);
        G_SEG_END()
    }
#undef G_MFMA
#undef G_SEG_END
#undef G_LOAD
#undef G_LGKM
    if (wr == 0) __builtin_amdgcn_s_barrier(); // pairs with the extra barrier of waves 4-7
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");

    // ---- epilogue. acc[n][m][e] = C[m0 + wr*128 + m*16 + (lane & 15)][n0 + wc*64 + n*16 + (lane >> 4)*4 + e]:
    // four consecutive columns per lane. Column tiles n and n + 1 are neighbours, so one v_permlane16_swap per register
    // pair (16-lane row 1 of tile n <-> row 0 of tile n + 1, row 3 <-> row 2) leaves every lane with EIGHT consecutive
    // columns: one 16-byte store per lane, 64 contiguous bytes per C row per instruction instead of 32 (8-byte stores
    // wrote 2.4x the algorithmic bytes beyond L2). A transposed-read operand splits its 256 rows / columns into halves
    // 0..127 | 128..255 instead (whole 256-B lines per DMA row), so a wave's tiles are then
    // {half*128 + wr*64 + ..} / {half*128 + wc*32 + ..}.
    uint16_t *C = (uint16_t *)g.C;
    const bool wide = g.ldc % 8 == 0 && (uintptr_t)g.C % 16 == 0;
    auto colbase = [&](int n) { return TRB ? (n >> 1) * 128 + wc * 32 + (n & 1) * 16 : wc * 64 + n * 16; };
#pragma unroll
    for (int np = 0; np < 4; np += 2) {
        float bias[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (g.epilogue == KF_EPI_BIAS_ROW) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t bb = ((const uint16_t *)g.bias)[n0 + colbase(np + t) + fg * 4 + e];
                    bias[t][e] = BF ? bf16_to_f32(bf16_t{bb}) : f16_to_f32(f16_t{bb});
                }
        }
        // after the exchange: lane rows 0 / 2 hold columns 0..7 / 8..15 of tile np, rows 1 / 3 those of tile np + 1
        const int64_t col = n0 + ((fg & 1) ? colbase(np + 1) : colbase(np)) + (fg >> 1) * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int64_t row = m0 + (TRA ? (m >> 2) * 128 + wr * 64 + (m & 3) * 16 : wr * 128 + m * 16) + fr;
            uint16_t *dst = C + row * g.ldc + col;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = g.alpha * acc[np][m][e] + bias[0][e], hi = g.alpha * acc[np + 1][m][e] + bias[1][e];
                const auto sw = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                v[e] = __uint_as_float(sw[0]);
                v[4 + e] = __uint_as_float(sw[1]);
            }
            if (g.beta != 0.f) {
                uint32_t ow[4];
                if (wide) {
                    const uint4 old = *(const uint4 *)dst;
                    ow[0] = old.x, ow[1] = old.y, ow[2] = old.z, ow[3] = old.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ow[e] = (uint32_t)dst[2 * e] | ((uint32_t)dst[2 * e + 1] << 16);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t o = (uint16_t)(ow[e >> 1] >> ((e & 1) * 16));
                    v[e] += g.beta * (BF ? bf16_to_f32(bf16_t{o}) : f16_to_f32(f16_t{o}));
                }
            }
            h_epi8<BF>(g, row, col, v);
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[e] = g_pack2<BF>(v[2 * e], v[2 * e + 1]);
            }
            if (wide) {
                *(uint4 *)dst = uint4{w[0], w[1], w[2], w[3]};
            } else { // C rows not 16-byte aligned
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// bf16 / f16, large shapes, second form: the same 256 x 256 x 64 block tile and the same LDS images, but FOUR waves
// (2 x 2, 128 x 128 of C per wave), ONE wave per SIMD with the whole register file: 256 accumulator registers (AGPRs)
// + two sets of fragments (2 x 64 VGPRs). A wave reads each LDS byte for 128 rows / columns of MFMA work instead of 64
// (LDS traffic per K tile 128 KiB instead of 192), there is ONE barrier per K tile instead of eight, and nothing
// depends on a partner wave: the fragments of the next k-step are read under the 64 MFMAs of the current one.
//     S0: 64 MFMAs of (tile t, k-step 0) | under them: read fragments (t, k-step 1)
//     P : s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier   - tile t + 1 has landed for everyone, tile t's buffer is free
//     S1: 64 MFMAs of (t, k-step 1)      | under them: 16 DMA operations of tile t + 2, read fragments (t + 1, k-step 0)
// Half-tiles: HA0 / HA1 = rows 0..127 / 128..255 of the A tile, HB0 / HB1 likewise for B; wave (wr, wc) reads HA[wr], HB[wc].
// Where a 4096^3 launch's 110 us go (round 3, random operands, back to back; each row = the kernel with parts compiled out): MFMAs alone
// 71 us (2048 cycles per K tile at 2.2 GHz + 12 us of launch, prologue and epilogue), + the LDS-DMA stream 83, + the fragment reads 94,
// everything 110; the data movement without the MFMAs 67. No single unit is the bound - every activity added lowers the clock the power
// limit leaves (1.87 GHz with all of them, profiles/r01_gemm_clock.json) and the loop's MFMA issue sits at 84 % of that clock.
// (Round 3, tried and dropped - a patch of it sits in the git history beside the commit that introduced the constant-0 first k-step: one workgroup per CU walking the tiles of a
// multi-round grid, the LDS-DMA a K loop issues past its end fetching the NEXT tile's first K tiles, the epilogue's stores draining under
// the next tile's first MFMAs behind a counted wait. Bit-identical, and within +-3 % of one tile per workgroup on every shape measured in
// one process (4096 x 12288 x 4096 ... 8192^3): the dispatcher already starts the next workgroup as fast as a tile loop does.)
// ------------------------------------------------------------------------------------------
constexpr int W4_NT = 256;
#ifndef W4_DMA_GROUPS
#define W4_DMA_GROUPS 16 // the 16 LDS-DMA operations of a K tile go behind the first W4_DMA_GROUPS x 4 MFMAs after the barrier (4, 8 or 16)
#endif

// DIAG (diagnostic build only, tools/gemm_clock.py): stamps s_memtime / s_memrealtime around the main loop and writes the
// two differences to g.bias (a buffer nothing else reads) - the in-kernel clock is their ratio x 100 MHz.
// The body is a device function of (problem, workgroup id, workgroups of the problem): gemm_w4_kernel runs one problem per launch,
// gemm_w4_pair_kernel two (the backward pair dA = dC B^T, dB = A^T dC) in ONE grid, so the second problem's first tiles start
// under the first problem's last ones instead of behind a kernel boundary.
template <bool BF, bool TRA, bool TRB, bool DIAG, bool TAIL = false>
__device__ __forceinline__ void gemm_w4_body(const GemmArgs &g, const uint32_t bid, const uint32_t nwg, char *smem) {
    using frag_t = typename HFrag<BF>::type;
    const int lane = threadIdx.x & 63, wid = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int wr = wid >> 1, wc = wid & 1;
    const uint32_t tiles_n = (uint32_t)(g.N / G_BN);
    uint32_t tm, tn;
    grouped_tile(xcd_remap(bid, nwg), (uint32_t)(g.M / G_BM), tiles_n, (uint32_t)g.group_m, tm, tn);
    const int64_t m0 = (int64_t)tm * G_BM, n0 = (int64_t)tn * G_BN;
    const int nt = (int)(g.K / G_BK);

    // ---- LDS-DMA: a half-tile is 16 operations of 1 KiB; wave w issues operations w*4 + i, i = 0..3.
    // K-contiguous operand: operation j moves rows 8j..8j+7 (128 B each; source chunk = position ^ ((row >> 1) & 7), which
    // depends on i only through i & 1). Transposed-read operand: operation j moves k rows 4j..4j+3 (256 B each; source
    // chunk ^ (k & 3) << 2; 32 B of padding after every pair of operations).
    const char *srcA[2], *srcB[2]; // [i & 1]
    int64_t opstepA, opstepB, halfA, halfB, kstepA, kstepB;
    {
        const int pos = lane & 7, r8 = lane >> 3;
        const int j0 = ((lane & 15) ^ (((lane >> 4) & 3) << 2)) * 8;
        if constexpr (TRA) {
            srcA[0] = srcA[1] = (const char *)g.A + ((int64_t)(wid * 16 + (lane >> 4)) * g.lda + m0 + j0) * 2;
            opstepA = 4 * g.lda * 2, halfA = 128 * 2, kstepA = (int64_t)G_BK * g.lda * 2;
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                srcA[p] = (const char *)g.A + (m0 + wid * 32 + r8) * g.lda * 2 + ((pos ^ ((p * 4 + (lane >> 4)) & 7)) << 4);
            opstepA = 8 * g.lda * 2, halfA = 128 * g.lda * 2, kstepA = G_BK * 2;
        }
        if constexpr (TRB) {
            srcB[0] = srcB[1] = (const char *)g.B + ((int64_t)(wid * 16 + (lane >> 4)) * g.ldb + n0 + j0) * 2;
            opstepB = 4 * g.ldb * 2, halfB = 128 * 2, kstepB = (int64_t)G_BK * g.ldb * 2;
        } else {
#pragma unroll
            for (int p = 0; p < 2; ++p)
                srcB[p] = (const char *)g.B + (n0 + wid * 32 + r8) * g.ldb * 2 + ((pos ^ ((p * 4 + (lane >> 4)) & 7)) << 4);
            opstepB = 8 * g.ldb * 2, halfB = 128 * g.ldb * 2, kstepB = G_BK * 2;
        }
    }
    // one DMA operation: which = 0 HA0, 1 HB0, 2 HA1, 3 HB1 (also the slot inside a tile buffer), i = 0..3
    auto stage_op = [&](int which, int i, int kt) __attribute__((always_inline)) {
        const int64_t ktc = kt < nt ? kt : nt - 1; // past the end: re-fetch the last tile (never read)
        const bool isB = which & 1;
        const int h = which >> 1, j = wid * 4 + i;
        const bool tr = isB ? TRB : TRA;
        const char *src = (isB ? srcB[i & 1] + i * opstepB + h * halfB + ktc * kstepB : srcA[i & 1] + i * opstepA + h * halfA + ktc * kstepA);
        char *dst = smem + (kt & 1) * G_TILE + which * G_HALF + j * 1024 + (tr ? 32 * (j >> 1) : 0);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)src, (__attribute__((address_space(3))) void *)dst, 16, 0, 0);
    };

    // ---- fragment reads (same images and address forms as gemm_h256_kernel)
    const int fr = lane & 15, fg = lane >> 4;
    const int sw = (fr >> 1) & 7;
    const int offk[2] = {fr * 128 + (((0 + fg) ^ sw) << 4), fr * 128 + (((4 + fg) ^ sw) << 4)};
    const int tq = fr >> 2;
    const int lbT = fg * 2048 + tq * 256 + (((((fr & 3) >> 1)) ^ (tq << 2)) << 4) + 8 * (fr & 1);
    const int pbT = 32 * fg;
    const unsigned smem_u = (unsigned)(uintptr_t)(const __attribute__((address_space(3))) char *)smem;

    f32x4 acc[8][8]; // [n-tile][m-tile]: D = B_frag x A_frag, i.e. C^T tiles (first written by K tile 0's first k-step)
    frag_t fa[2][8], fb[2][8]; // [set][tile]

    // Everything in the loop is volatile inline asm, i.e. issued exactly in source order: MFMAs with "+a" accumulators (all
    // 256 AGPRs, updated in place - left to itself the register allocator rotates accumulator tiles through copies at
    // this pressure), fragment reads, and the counted waits that publish them. An accumulator is touched once per
    // k-step (64 MFMAs apart) and fragments are consumed only after an lgkmcnt(0), so no MFMA hazard needs padding.
    auto rd = [&](frag_t &dst, unsigned base, auto off) __attribute__((always_inline)) {
        asm volatile("ds_read_b128 %0, %1 offset:%c2" : "=v"(dst) : "v"(base), "n"(decltype(off)::value));
    };
    auto mm = [&](f32x4 &c, const frag_t &b, const frag_t &a) __attribute__((always_inline)) {
        if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(b), "v"(a));
    };
    // the first k-step of the output tile starts its 64 accumulators from the inline constant 0: nothing to zero (the ~500
    // v_accvgpr_write the zero-initialisation used to cost stood between the first data landing and the first MFMA)
    auto mmz = [&](f32x4 &c, const frag_t &b, const frag_t &a) __attribute__((always_inline)) {
        if constexpr (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(b), "v"(a));
        else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(b), "v"(a));
    };
    // read number r (0..15) of a k-step: A tiles 0..7, then B tiles 0..7, in two halves that go behind DIFFERENT MFMAs of a group.
    // A K-contiguous operand's fragment is one ds_read_b128 (first half; the second is empty). A transposed-read operand's is two
    // ds_read_b64_tr_b16, one per half, off ONE address per PAIR of tiles: tile t sits at byte (t >> 1) << 6 XORed into the lane's
    // chunk swizzle (four per-lane bases xbA / xbB, set up once) plus the immediate (t & 1) * 32, so the loop's only address
    // arithmetic is base + buffer. (Both operands transposed - the TN product dW = A^T dC - used to spend 56 v_add_u32 and both
    // reads of a fragment in the single gap behind a group's first MFMA; a lone wave fits about two instructions into a 16-cycle
    // MFMA: TN ran 8 % behind NT.)
    unsigned xbA[4], xbB[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        xbA[j] = (unsigned)((lbT ^ (j << 6)) + pbT) + (2 * wr) * G_HALF;
        xbB[j] = (unsigned)((lbT ^ (j << 6)) + pbT) + (2 * wc + 1) * G_HALF;
    }
    g_s16x4 trlo, trhi;
    auto trd = [&](g_s16x4 &dst, unsigned addr, auto off) __attribute__((always_inline)) {
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%c2" : "=v"(dst) : "v"(addr), "n"(decltype(off)::value) : "memory");
    };
    auto trjoin = [&]() __attribute__((always_inline)) {
        g_s16x8 r;
        r[0] = trlo[0]; r[1] = trlo[1]; r[2] = trlo[2]; r[3] = trlo[3];
        r[4] = trhi[0]; r[5] = trhi[1]; r[6] = trhi[2]; r[7] = trhi[3];
        return __builtin_bit_cast(frag_t, r);
    };
#define W4_READ_A(SET, KS, BUFU, R)                                                                            \
    {                                                                                                          \
        constexpr int t = (R) & 7;                                                                             \
        constexpr int toff = ((KS) == 0 ? 0 : 8192 + 128) + (t & 1) * 32;                                      \
        if constexpr ((R) < 8) {                                                                               \
            if constexpr (TRA) trd(trlo, xbA[t >> 1] + (BUFU), std::integral_constant<int, toff>{});           \
            else rd(fa[SET][t], (BUFU) + (2 * wr) * G_HALF + offk[KS], std::integral_constant<int, t * 2048>{}); \
        } else {                                                                                               \
            if constexpr (TRB) trd(trlo, xbB[t >> 1] + (BUFU), std::integral_constant<int, toff>{});           \
            else rd(fb[SET][t], (BUFU) + (2 * wc + 1) * G_HALF + offk[KS], std::integral_constant<int, t * 2048>{}); \
        }                                                                                                      \
    }
#define W4_READ_B(SET, KS, BUFU, R)                                                                            \
    {                                                                                                          \
        constexpr int t = (R) & 7;                                                                             \
        constexpr int toff = ((KS) == 0 ? 0 : 8192 + 128) + (t & 1) * 32 + 1024;                               \
        if constexpr ((R) < 8) {                                                                               \
            if constexpr (TRA) { trd(trhi, xbA[t >> 1] + (BUFU), std::integral_constant<int, toff>{}); fa[SET][t] = trjoin(); } \
        } else {                                                                                               \
            if constexpr (TRB) { trd(trhi, xbB[t >> 1] + (BUFU), std::integral_constant<int, toff>{}); fb[SET][t] = trjoin(); } \
        }                                                                                                      \
    }
#define W4_READ1(SET, KS, BUFU, R) W4_READ_A(SET, KS, BUFU, R) W4_READ_B(SET, KS, BUFU, R)
    // one k-step: 64 MFMAs on fragment set SET; after every fourth one, read R of the next fragment set and (DMA) one
    // LDS-DMA operation of tile kt + 2
#define W4_STEP(SET, NKS, NBUFU, DMA) W4_STEPM(mm, SET, NKS, NBUFU, DMA)
#define W4_STEPM(MM, SET, NKS, NBUFU, DMA)                                                                          \
    W4_GROUP(MM, SET, NKS, NBUFU, DMA, 0) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 1) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 2) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 3) \
    W4_GROUP(MM, SET, NKS, NBUFU, DMA, 4) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 5) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 6) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 7) \
    W4_GROUP(MM, SET, NKS, NBUFU, DMA, 8) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 9) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 10) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 11) \
    W4_GROUP(MM, SET, NKS, NBUFU, DMA, 12) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 13) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 14) W4_GROUP(MM, SET, NKS, NBUFU, DMA, 15)
#define W4_GROUP(MM, SET, NKS, NBUFU, DMA, G)                                                                      \
    {                                                                                                          \
        constexpr int n = (G) >> 1, mb = ((G) & 1) * 4;                                                        \
        MM(acc[n][mb + 0], fb[SET][n], fa[SET][mb + 0]);                                                       \
        if constexpr (DMA && (G) < W4_DMA_GROUPS) stage_op((G) * (16 / W4_DMA_GROUPS) / 4, (G) * (16 / W4_DMA_GROUPS) % 4, kt + 2); \
        MM(acc[n][mb + 1], fb[SET][n], fa[SET][mb + 1]);                                                       \
        if constexpr (DMA && (G) < W4_DMA_GROUPS && W4_DMA_GROUPS <= 8) stage_op(((G) * (16 / W4_DMA_GROUPS) + 1) / 4, ((G) * (16 / W4_DMA_GROUPS) + 1) % 4, kt + 2); \
        W4_READ_A(1 - (SET), NKS, NBUFU, G)                                                                    \
        MM(acc[n][mb + 2], fb[SET][n], fa[SET][mb + 2]);                                                       \
        W4_READ_B(1 - (SET), NKS, NBUFU, G)                                                                    \
        if constexpr (DMA && (G) < W4_DMA_GROUPS && W4_DMA_GROUPS <= 4) stage_op(((G) * 4 + 2) / 4, ((G) * 4 + 2) % 4, kt + 2); \
        MM(acc[n][mb + 3], fb[SET][n], fa[SET][mb + 3]);                                                       \
        if constexpr (DMA && (G) < W4_DMA_GROUPS && W4_DMA_GROUPS <= 4) stage_op(((G) * 4 + 3) / 4, ((G) * 4 + 3) % 4, kt + 2); \
    }

    // ---- prologue: tiles 0 and 1 in flight, tile 0 landed, fragments (0, k-step 0) in set 0
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
        for (int which = 0; which < 4; ++which)
#pragma unroll
            for (int i = 0; i < 4; ++i) stage_op(which, i, kt);
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#define W4_R(R) W4_READ1(0, 0, smem_u, R)
    W4_R(0) W4_R(1) W4_R(2) W4_R(3) W4_R(4) W4_R(5) W4_R(6) W4_R(7) W4_R(8) W4_R(9) W4_R(10) W4_R(11) W4_R(12) W4_R(13) W4_R(14) W4_R(15)
#undef W4_R
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");

    uint64_t t0 = 0, r0 = 0;
    if constexpr (DIAG) {
        t0 = __builtin_amdgcn_s_memtime();
        r0 = __builtin_amdgcn_s_memrealtime();
    }
    { // K tile 0: the accumulators start from 0 in its first k-step
        const int kt = 0;
        const unsigned bufu = smem_u, nbufu = smem_u + G_TILE;
        W4_STEPM(mmz, 0, 1, bufu, false)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        W4_STEP(1, 0, nbufu, true)
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    for (int kt = 1; kt < nt; ++kt) {
        const unsigned bufu = smem_u + (kt & 1) * G_TILE, nbufu = smem_u + ((kt + 1) & 1) * G_TILE;
        W4_STEP(0, 1, bufu, false)                                                        // S0
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");          // P
        W4_STEP(1, 0, nbufu, true)                                                        // S1
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
#undef W4_STEP
#undef W4_STEPM
#undef W4_GROUP
#undef W4_READ1
#undef W4_READ_A
#undef W4_READ_B
    if constexpr (DIAG) {
        const uint64_t t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
        if (threadIdx.x == 0) {
            uint64_t *d = (uint64_t *)g.bias + 2 * (size_t)bid;
            d[0] = t1 - t0;
            d[1] = r1 - r0;
        }
    }
    // the last MFMAs are still in the pipe: nothing may read their accumulators yet. The "+a" operands order every
    // compiler-generated read (epilogue, spill code) of the last row of tiles behind the wait.
    asm volatile("s_waitcnt vmcnt(0)\n\ts_nop 15\n\ts_nop 15\n\ts_nop 15"
                 : "+a"(acc[7][0]), "+a"(acc[7][1]), "+a"(acc[7][2]), "+a"(acc[7][3]), "+a"(acc[7][4]), "+a"(acc[7][5]), "+a"(acc[7][6]), "+a"(acc[7][7])
                 :
                 : "memory");

    // ---- epilogue: as gemm_h256_kernel's (lane exchange between neighbouring column tiles, 16-byte stores)
    uint16_t *C = (uint16_t *)g.C;
    const bool wide = g.ldc % 8 == 0 && (uintptr_t)g.C % 16 == 0;
#pragma unroll
    for (int np = 0; np < 8; np += 2) {
        float bias[2][4] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
        if (g.epilogue == KF_EPI_BIAS_ROW) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const uint16_t bb = ((const uint16_t *)g.bias)[n0 + wc * 128 + (np + t) * 16 + fg * 4 + e];
                    bias[t][e] = BF ? bf16_to_f32(bf16_t{bb}) : f16_to_f32(f16_t{bb});
                }
        }
        const int64_t col = n0 + wc * 128 + (np + (fg & 1)) * 16 + (fg >> 1) * 8;
#pragma unroll
        for (int m = 0; m < 8; ++m) {
            const int64_t row = m0 + wr * 128 + m * 16 + fr;
            uint16_t *dst = C + row * g.ldc + col;
            float v[8];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float lo = g.alpha * acc[np][m][e] + bias[0][e], hi = g.alpha * acc[np + 1][m][e] + bias[1][e];
                const auto swp = __builtin_amdgcn_permlane16_swap(__float_as_uint(lo), __float_as_uint(hi), false, false);
                v[e] = __uint_as_float(swp[0]);
                v[4 + e] = __uint_as_float(swp[1]);
            }
            if (g.c_f32) { // float C (8 consecutive columns of a row: two 16-byte stores when the rows allow)
                float *df = (float *)g.C + row * g.ldc + col;
                const bool wide4 = g.ldc % 4 == 0 && (uintptr_t)g.C % 16 == 0;
                if (g.beta != 0.f) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) v[e] += g.beta * df[e];
                }
                if constexpr (TAIL) h_epi8<BF>(g, row, col, v);
                if (wide4) {
                    *(float4 *)df = float4{v[0], v[1], v[2], v[3]};
                    *(float4 *)(df + 4) = float4{v[4], v[5], v[6], v[7]};
                } else {
#pragma unroll
                    for (int e = 0; e < 8; ++e) df[e] = v[e];
                }
                continue;
            }
            if (g.beta != 0.f) {
                uint32_t ow[4];
                if (wide) {
                    const uint4 old = *(const uint4 *)dst;
                    ow[0] = old.x, ow[1] = old.y, ow[2] = old.z, ow[3] = old.w;
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) ow[e] = (uint32_t)dst[2 * e] | ((uint32_t)dst[2 * e + 1] << 16);
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const uint16_t o = (uint16_t)(ow[e >> 1] >> ((e & 1) * 16));
                    v[e] += g.beta * (BF ? bf16_to_f32(bf16_t{o}) : f16_to_f32(f16_t{o}));
                }
            }
            if constexpr (TAIL) h_epi8<BF>(g, row, col, v);
            uint32_t w[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                w[e] = g_pack2<BF>(v[2 * e], v[2 * e + 1]);
            }
            if (wide) {
                *(uint4 *)dst = uint4{w[0], w[1], w[2], w[3]};
            } else {
#pragma unroll
                for (int e = 0; e < 8; ++e) dst[e] = (uint16_t)(w[e >> 1] >> ((e & 1) * 16));
            }
        }
    }
}

template <bool BF, bool TRA, bool TRB, bool DIAG = false>
__global__ __launch_bounds__(W4_NT) void gemm_w4_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_w4_body<BF, TRA, TRB, DIAG>(g, blockIdx.x, gridDim.x, smem);
}
// the same kernel with the element-wise tail (mul / add / aux operands) in its epilogue: its own instantiation, so the plain product
// carries none of the tail's pointers through its loop
template <bool BF, bool TRA, bool TRB>
__global__ __launch_bounds__(W4_NT) void gemm_w4_tail_kernel(const GemmArgs g) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    gemm_w4_body<BF, TRA, TRB, false, true>(g, blockIdx.x, gridDim.x, smem);
}

// Two problems, one grid: g0 is an NT product (A [M,K], B stored [N,K]: dA = dC W^T), g1 a TN product (A stored [K,M], B [K,N]:
// dW = A^T dC); n0, n1 = their tile counts, both multiples of 8. Workgroup ids are dealt to XCDs round-robin, so problem 0 takes
// the first n0 / 8 ids of every XCD and problem 1 the rest: each problem keeps the XCD-major tile order xcd_remap gives it alone.
template <bool BF>
__global__ __launch_bounds__(W4_NT) void gemm_w4_pair_kernel(const GemmArgs g0, const GemmArgs g1, const uint32_t n0, const uint32_t n1) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const uint32_t xcd = blockIdx.x & 7u, slot = blockIdx.x >> 3, s0 = n0 >> 3;
    if (slot < s0) gemm_w4_body<BF, false, false, false>(g0, slot * 8 + xcd, n0, smem); // TRB template flag = "B stored [K,N]": NT is <false, false>
    else gemm_w4_body<BF, true, true, false>(g1, (slot - s0) * 8 + xcd, n1, smem);       // TN: A stored [K,M] (TRA), B stored [K,N] (TRB)
}

static inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// ---- ragged shapes through the matrix kernels (round 5) ---------------------------------------------------------------------------
// A 16-bit or f32 product whose extents are not whole tiles used to fall to the scalar kernel (4000^3 bf16: 5 ms = 26 TFLOP/s; M = 16 ..
// 64 rows against 8192 x 8192: 0.7 ms). With caller scratch (kf_gemm_workspace_bytes) the operands are copied into zero-padded images of
// whole tiles, the tile kernels run on those (zeros add nothing: the valid region carries exactly the tile kernel's sums), and the valid
// part of C is copied back. Copies move 16 bytes per lane wherever the ragged side happens to be 16-byte aligned.
template <int ES>
__global__ __launch_bounds__(256) void gemm_pad_copy_kernel(const char *src, int64_t ld_src, int64_t rows, int64_t cols, char *dst, int64_t cols_p, int64_t total) {
    constexpr int PER = 16 / ES;
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; // one 16-byte piece of the padded image
    if (i >= total) return;
    const int64_t per_row = cols_p / PER, r = i / per_row, c0 = (i - r * per_row) * PER;
    uint4 v = make_uint4(0u, 0u, 0u, 0u);
    if (r < rows && c0 < cols) {
        const char *q = src + (r * ld_src + c0) * ES;
        if (c0 + PER <= cols && ((uintptr_t)q & 15) == 0) {
            v = *(const uint4 *)q;
        } else {
            uint32_t w[4] = {0u, 0u, 0u, 0u};
#pragma unroll
            for (int e = 0; e < PER; ++e)
                if (c0 + e < cols) {
                    if constexpr (ES == 2) w[e >> 1] |= (uint32_t)(*(const uint16_t *)(q + 2 * e)) << (16 * (e & 1));
                    else w[e] = *(const uint32_t *)(q + 4 * e);
                }
            v = make_uint4(w[0], w[1], w[2], w[3]);
        }
    }
    *(uint4 *)(dst + i * 16) = v;
}
template <int ES>
__global__ __launch_bounds__(256) void gemm_unpad_copy_kernel(const char *src, int64_t cols_p, char *dst, int64_t ld_dst, int64_t rows, int64_t cols) {
    constexpr int PER = 16 / ES;
    const int64_t per_row = (cols + PER - 1) / PER, i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= rows * per_row) return;
    const int64_t r = i / per_row, c0 = (i - r * per_row) * PER;
    const uint4 v = *(const uint4 *)(src + (r * cols_p + c0) * ES);
    char *q = dst + (r * ld_dst + c0) * ES;
    if (c0 + PER <= cols && ((uintptr_t)q & 15) == 0) {
        *(uint4 *)q = v;
    } else {
        const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int e = 0; e < PER; ++e)
            if (c0 + e < cols) {
                if constexpr (ES == 2) *(uint16_t *)(q + 2 * e) = (uint16_t)(w[e >> 1] >> (16 * (e & 1)));
                else *(uint32_t *)(q + 4 * e) = w[e];
            }
    }
}

static bool h_fast_ok(int64_t M, int64_t N, int64_t K) { return M % H_BM == 0 && N % H_BN == 0 && K % H_BK == 0 && M > 0 && N > 0 && K > 0; }

} // namespace kf

using namespace kf;

// the 256-tile kernel when its grid covers a good part of the chip (256 CUs, one block each); smaller problems get four
// times as many 128-tile blocks instead (2048^3: 64 tiles of 256^2 would leave three quarters of the CUs idle).
// Where the line is (round 5, random bf16 operands, back to back, KF_GEMM_H256_MIN): 64 tiles 394 vs 721 TFLOP/s for the 128-tile kernel,
// 81 tiles 573 vs 595, 100 tiles 592-610 vs 532-592, 121 tiles 720-752 vs 627-656, 128 tiles 816-850 vs 704-724, 144 tiles (3072^3) 886-923
// vs 599 - the line had stood at 160 tiles.
static bool h256_ok(int64_t M, int64_t N, int64_t K) {
    return M % G_BM == 0 && N % G_BN == 0 && K % G_BK == 0 && M > 0 && N > 0 && K > 0 && (M / G_BM) * (N / G_BN) >= knob_int(KNOB_GEMM_H256_MIN, 100) &&
           !knob(KNOB_GEMM_128);
}

// Split-K plan: only where the 128-tile 16-bit kernel would leave most of the chip idle (at most 128 tiles = half a round of 256 CUs)
// AND the contraction is long enough to pay for the f32 partial round trip (at least 8 K tiles of 64 per slice). Slices = the largest
// power of two that keeps tiles x slices within two rounds. A skinny product (M = 256, N = 4096, K = 16384: 64 tiles x 256 K tiles)
// goes from 64 busy CUs to 512 workgroups.
// (Round 3, tried and dropped; the measurements are what is kept: 2048^3 as 64 tiles of 256^2 x 4 K slices in the
// 4-wave kernel, the slices of a tile finishing it by a reduce-scatter through the workspace inside the launch (flags, no fold kernel).
// Correct, and 67 us against this path's 33: the bare 8-K-tile loop + prologue + a quarter epilogue is already 20 us, the 48 MB of f32
// pieces cost 19 us to write and 14 us to read back (2.5-3.4 TB/s: they do not stay in the 8 x 4 MiB of L2), and the agent-scope
// release / acquire another 12 us (L2 write-back + invalidate on every workgroup). A 128^2 tile per CU is LDS-bound instead: 512 B of
// LDS-DMA writes + 1024 B of fragment reads per unit of k against 8 MFMA cycles = 12 cycles per k, a ceiling of 0.67 of peak.)
static int splitk_slices(int dtype, int64_t M, int64_t N, int64_t K) {
    if (!(dtype == KF_BF16 || dtype == KF_F16) || !h_fast_ok(M, N, K) || h256_ok(M, N, K) || knob(KNOB_GEMM_NO_SPLITK)) return 1;
    const int64_t tiles = (M / H_BM) * (N / H_BN), nt = K / H_BK;
    if (tiles > 128) return 1;
    int s = 1;
    while (s < 16 && tiles * (s * 2) <= 512 && nt / (s * 2) >= 8) s *= 2;
    return s;
}

struct PadPlan {
    bool use;
    int64_t Mp, Np, Kp;
    size_t a_bytes, b_bytes, c_bytes, bias_bytes, inner_bytes, total;
};
static inline size_t pad256(size_t x) { return (x + 255) & ~(size_t)255; }
static PadPlan pad_plan(int dtype, int64_t M, int64_t N, int64_t K) {
    PadPlan p{};
    const bool half = dtype == KF_BF16 || dtype == KF_F16;
    if (!(half || dtype == KF_F32) || M <= 0 || N <= 0 || K <= 0 || knob(KNOB_GEMM_NO_PAD)) return p;
    if ((double)M * (double)N * (double)K < (double)(1 << 24)) return p; // (below ~256^3 the scalar kernel's one launch is as good as five)
    const int64_t tm = half ? 128 : 64, tk = half ? 64 : 16;
    p.Mp = (M + tm - 1) / tm * tm;
    p.Np = (N + tm - 1) / tm * tm;
    p.Kp = (K + tk - 1) / tk * tk;
    if (p.Mp == M && p.Np == N && p.Kp == K) return p; // whole tiles: nothing to pad (operands that are merely misaligned keep the scalar kernel)
    const size_t es = half ? 2 : 4;
    p.a_bytes = pad256((size_t)p.Mp * p.Kp * es);
    p.b_bytes = pad256((size_t)p.Kp * p.Np * es);
    p.c_bytes = pad256((size_t)p.Mp * p.Np * es);
    p.bias_bytes = pad256((size_t)p.Np * es);
    const int sl = splitk_slices(dtype, p.Mp, p.Np, p.Kp);
    p.inner_bytes = sl > 1 ? pad256((size_t)sl * p.Mp * p.Np * sizeof(float)) : 0;
    p.total = p.a_bytes + p.b_bytes + p.c_bytes + p.bias_bytes + p.inner_bytes;
    p.use = true;
    return p;
}


extern "C" int kf_gemm_workspace_bytes(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, size_t *bytes) {
    KF_REQUIRE(bytes, KF_ERR_INVALID, "kf_gemm_workspace_bytes: null out pointer");
    *bytes = 0;
    // every kernel reads every operand layout in place; the only scratch any GEMM takes is split-K's f32 partial tiles, and a call
    // without it simply runs unsplit
    (void)trans_a; (void)trans_b;
    const int s = splitk_slices(dtype, M, N, K);
    if (s > 1) *bytes = (size_t)s * (size_t)M * (size_t)N * sizeof(float);
    else if (const PadPlan pp = pad_plan(dtype, M, N, K); pp.use) *bytes = pp.total; // ragged extents: zero-padded images + what the padded product takes
    return KF_OK;
}

// Round 3: the 4-wave form on every grid. It used to win only up to two rounds of tiles; with its transposed reads spread over the MFMA
// gaps and the shorter epilogue it is ahead of the 8-wave form everywhere measured (tools/scratch/w4_vs_w8.py, same process, interleaved):
// 4096 x 12288 x 4096 NN 1258 vs 1120 TFLOP/s, TN 1234 vs 1091; 4096 x 4096 x 16384 TN 1315 vs 1130; 8192^3 NN 1396 vs 1264, NT 1454
// vs 1390; 5120^3 TN 1058 vs 934. KF_GEMM_W8 still selects the 8-wave kernel (tests keep it covered).
static bool h256_use_w4(int64_t M, int64_t N) {
    (void)M; (void)N;
    return !knob(KNOB_GEMM_W8);
}

template <bool BF>
static int launch_h256(const GemmArgs &g, bool tra, bool trb, bool w4, hipStream_t st) {
    const unsigned grid = (unsigned)((g.M / G_BM) * (g.N / G_BN));
    const bool tail = g.mul || g.add || g.aux;
#define KF_H256(TA, TB)                                                                                                   \
    {                                                                                                                     \
        KF_ENSURE_LDS((gemm_h256_kernel<BF, TA, TB>), G_LDS); \
        gemm_h256_kernel<BF, TA, TB><<<grid, G_NT, G_LDS, st>>>(g);                                                       \
    }
#define KF_W4(TA, TB)                                                                                                     \
    {                                                                                                                     \
        if (tail) {                                                                                                       \
            KF_ENSURE_LDS((gemm_w4_tail_kernel<BF, TA, TB>), G_LDS);                                                      \
            gemm_w4_tail_kernel<BF, TA, TB><<<grid, W4_NT, G_LDS, st>>>(g);                                               \
        } else {                                                                                                          \
            KF_ENSURE_LDS((gemm_w4_kernel<BF, TA, TB>), G_LDS);                                                           \
            gemm_w4_kernel<BF, TA, TB><<<grid, W4_NT, G_LDS, st>>>(g);                                                    \
        }                                                                                                                 \
    }
    if (w4) {
        if (!tra && !trb) KF_W4(false, false)
        else if (!tra && trb) KF_W4(false, true)
        else if (tra && !trb) KF_W4(true, false)
        else KF_W4(true, true)
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
#undef KF_W4
    if (!tra && !trb) KF_H256(false, false)
    else if (!tra && trb) KF_H256(false, true)
    else if (tra && !trb) KF_H256(true, false)
    else KF_H256(true, true)
#undef KF_H256
    KF_LAUNCH_CHECK();
    return KF_OK;
}

// diagnostic entry, compiled only into the separate diagnostic library (-DKF_DIAG_BUILD, kfunca_amd/_build.py build_diag;
// tools/gemm_clock.py) - never into libkfunca_hip.so: bf16 A [M,K] x B stored [N,K] through the 4-wave kernel with clock
// stamps; diag receives {core-clock cycles, 100 MHz ticks} of the main loop per workgroup
#ifdef KF_DIAG_BUILD
extern "C" int kfdbg_gemm_clock(int64_t M, int64_t N, int64_t K, const void *A, const void *B, void *C, void *diag, void *stream) {
    KF_REQUIRE(h256_ok(M, N, K) && A && B && C && diag, KF_ERR_INVALID, "kfdbg_gemm_clock: 256-tile shapes only");
    GemmArgs g{A, B, C, diag, M, N, K, K, K, N, 1.f, 0.f, KF_EPI_NONE, 4};
    KF_ENSURE_LDS((gemm_w4_kernel<true, false, false, true>), G_LDS);
    gemm_w4_kernel<true, false, false, true><<<(unsigned)((M / G_BM) * (N / G_BN)), W4_NT, G_LDS, as_stream(stream)>>>(g);
    KF_LAUNCH_CHECK();
    return KF_OK;
}
#endif

static int gemm_impl(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda, const void *B,
                     int64_t ldb, float beta, void *C, int64_t ldc, int epilogue, const void *bias, const kf_gemm_epilogue *ex, void *stream,
                     void *workspace = nullptr, size_t workspace_bytes = 0);

extern "C" int kf_gemm(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
                       int64_t lda, const void *B, int64_t ldb, float beta, void *C, int64_t ldc, int epilogue,
                       const void *bias, void *workspace, size_t workspace_bytes, void *stream) {
    return gemm_impl(dtype, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, epilogue, bias, nullptr, stream, workspace, workspace_bytes);
}

// the backward pair of one linear layer (dA = dC W^T: NT, dW = A^T dC: TN) on the 4-wave 256-tile kernel as ONE grid
static bool grouped_single_grid(int dtype, int count, const kf_gemm_problem *p) {
    const bool half = dtype == KF_BF16 || dtype == KF_F16;
    auto pair_ok = [&](const kf_gemm_problem &q) {
        return q.A && q.B && q.C && h_fast_ok(q.M, q.N, q.K) && h256_ok(q.M, q.N, q.K) && h256_use_w4(q.M, q.N) && ((q.M / G_BM) * (q.N / G_BN)) % 8 == 0 &&
               (uintptr_t)q.A % 16 == 0 && (uintptr_t)q.B % 16 == 0 && q.lda % 8 == 0 && q.ldb % 8 == 0 && q.ldc >= q.N &&
               q.lda >= (q.trans_a ? q.M : q.K) && q.ldb >= (q.trans_b ? q.K : q.N);
    };
    return half && count == 2 && p && !p[0].trans_a && p[0].trans_b && p[1].trans_a && !p[1].trans_b && pair_ok(p[0]) && pair_ok(p[1]) && !knob(KNOB_GEMM_NO_GROUP);
}

extern "C" int kf_gemm_grouped_single_grid(int dtype, int count, const kf_gemm_problem *p) { return grouped_single_grid(dtype, count, p) ? 1 : 0; }

extern "C" int kf_gemm_grouped(int dtype, int count, const kf_gemm_problem *p, void *stream) {
    KF_REQUIRE(count >= 0 && (p || count == 0), KF_ERR_INVALID, "kf_gemm_grouped: null problem list");
    // one grid, no kernel boundary between the two products
    if (grouped_single_grid(dtype, count, p)) {
        hipStream_t st = as_stream(stream);
        GemmArgs g0{p[0].A, p[0].B, p[0].C, nullptr, p[0].M, p[0].N, p[0].K, p[0].lda, p[0].ldb, p[0].ldc, p[0].alpha, p[0].beta, KF_EPI_NONE, 0};
        GemmArgs g1{p[1].A, p[1].B, p[1].C, nullptr, p[1].M, p[1].N, p[1].K, p[1].lda, p[1].ldb, p[1].ldc, p[1].alpha, p[1].beta, KF_EPI_NONE, 0};
        g0.c_f32 = p[0].c_f32 ? 1 : 0;
        g1.c_f32 = p[1].c_f32 ? 1 : 0;
        g0.group_m = g1.group_m = (int)knob_int(KNOB_GEMM_GROUP_M, 4);
        const unsigned n0 = (unsigned)((p[0].M / G_BM) * (p[0].N / G_BN)), n1 = (unsigned)((p[1].M / G_BM) * (p[1].N / G_BN));
        KF_PROF(dtype == KF_BF16 ? "gemm_bf16_mfma_pair" : "gemm_f16_mfma_pair", st);
        if (dtype == KF_BF16) {
            KF_ENSURE_LDS((gemm_w4_pair_kernel<true>), G_LDS);
            gemm_w4_pair_kernel<true><<<n0 + n1, W4_NT, G_LDS, st>>>(g0, g1, n0, n1);
        } else {
            KF_ENSURE_LDS((gemm_w4_pair_kernel<false>), G_LDS);
            gemm_w4_pair_kernel<false><<<n0 + n1, W4_NT, G_LDS, st>>>(g0, g1, n0, n1);
        }
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    for (int i = 0; i < count; ++i) {
        kf_gemm_epilogue e{};
        e.c_f32 = p[i].c_f32;
        const int rc = p[i].c_f32 ? kf_gemm_ex(dtype, p[i].trans_a, p[i].trans_b, p[i].M, p[i].N, p[i].K, p[i].alpha, p[i].A, p[i].lda, p[i].B, p[i].ldb,
                                               p[i].beta, p[i].C, p[i].ldc, &e, stream)
                                  : kf_gemm(dtype, p[i].trans_a, p[i].trans_b, p[i].M, p[i].N, p[i].K, p[i].alpha, p[i].A, p[i].lda, p[i].B, p[i].ldb,
                                            p[i].beta, p[i].C, p[i].ldc, KF_EPI_NONE, nullptr, nullptr, 0, stream);
        if (rc != KF_OK) return rc;
    }
    return KF_OK;
}

extern "C" int kf_gemm_ex(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda,
                          const void *B, int64_t ldb, float beta, void *C, int64_t ldc, const kf_gemm_epilogue *epi, void *stream) {
    KF_REQUIRE(epi, KF_ERR_INVALID, "kf_gemm_ex: null epilogue");
    KF_REQUIRE(!epi->mul || epi->ldmul >= N, KF_ERR_INVALID, "kf_gemm_ex: mul operand's leading dimension too small");
    KF_REQUIRE(!epi->add || epi->ldadd >= N, KF_ERR_INVALID, "kf_gemm_ex: add operand's leading dimension too small");
    KF_REQUIRE(!epi->aux || epi->ldaux >= N, KF_ERR_INVALID, "kf_gemm_ex: aux output's leading dimension too small");
    KF_REQUIRE(!epi->aux || epi->aux != C, KF_ERR_INVALID, "kf_gemm_ex: aux must not alias C");
    return gemm_impl(dtype, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, epi->bias ? KF_EPI_BIAS_ROW : KF_EPI_NONE, epi->bias, epi,
                     stream);
}

static int gemm_impl(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda, const void *B,
                     int64_t ldb, float beta, void *C, int64_t ldc, int epilogue, const void *bias, const kf_gemm_epilogue *ex, void *stream,
                     void *workspace, size_t workspace_bytes) {
    KF_REQUIRE(dtype == KF_F32 || dtype == KF_F64 || dtype == KF_F16 || dtype == KF_BF16, KF_ERR_UNSUPPORTED,
               "kf_gemm: dtype %d not supported (float, double, half, bfloat16)", dtype);
    KF_REQUIRE(M >= 0 && N >= 0 && K >= 0, KF_ERR_INVALID, "kf_gemm: negative extent");
    if (M == 0 || N == 0) return KF_OK;
    KF_REQUIRE(A && B && C, KF_ERR_INVALID, "kf_gemm: null operand");
    KF_REQUIRE(lda >= (trans_a ? M : K) && ldb >= (trans_b ? K : N) && ldc >= N, KF_ERR_INVALID, "kf_gemm: leading dimension too small");
    KF_REQUIRE(epilogue == KF_EPI_NONE || (epilogue == KF_EPI_BIAS_ROW && bias), KF_ERR_INVALID, "kf_gemm: bad epilogue");
    KF_REQUIRE((M + 63) / 64 * ((N + 63) / 64) <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "kf_gemm: too many output tiles for one launch");
    hipStream_t st = as_stream(stream);
    GemmArgs g{A, B, C, bias, M, N, K, lda, ldb, ldc, alpha, beta, epilogue, 0};
    if (ex) { g.mul = ex->mul; g.add = ex->add; g.aux = ex->aux; g.ldmul = ex->ldmul; g.ldadd = ex->ldadd; g.ldaux = ex->ldaux; g.c_f32 = ex->c_f32 ? 1 : 0; }
    KF_REQUIRE(!g.c_f32 || dtype == KF_BF16 || dtype == KF_F16, KF_ERR_INVALID, "kf_gemm_ex: c_f32 asks for a float output behind 16-bit operands (dtype %d)", dtype);
    // tile rows per group of the XCD-aware tile walk: 4, and 8 once the grid is >= 1024 tiles of 256^2 (8192^3 NN: 1187 -> 1234 TFLOP/s per
    // launch, same box, tools/scratch/gemm_stride.py; 2..8 alike at 768 tiles, 1 and 16 behind everywhere)
    g.group_m = (int)knob_int(KNOB_GEMM_GROUP_M, (M / 256) * (N / 256) >= 1024 ? 8 : 4);

    const bool al16 = ((uintptr_t)A % 16 == 0) && ((uintptr_t)B % 16 == 0);
    if (dtype == KF_F32 && M % 64 == 0 && N % 64 == 0 && K % F_BK == 0 && K > 0 && al16 && lda % 4 == 0 && ldb % 4 == 0) {
        // 128-tiles unless they would leave a quarter of the CUs idle and 64-tiles are possible... or are the only option
        const bool t128 = M % F_BM == 0 && N % F_BN == 0 && (M / F_BM) * (N / F_BN) >= 192;
        KF_PROF(t128 ? "gemm_f32_mfma" : "gemm_f32_mfma_t64", st);
#define KF_F32G(T_)                                                                                          \
    {                                                                                                        \
        const unsigned grid = (unsigned)((M / T_) * (N / T_));                                               \
        if (!trans_a && !trans_b) gemm_f32_kernel<false, false, T_><<<grid, 256, 0, st>>>(g);                \
        else if (!trans_a && trans_b) gemm_f32_kernel<false, true, T_><<<grid, 256, 0, st>>>(g);             \
        else if (trans_a && !trans_b) gemm_f32_kernel<true, false, T_><<<grid, 256, 0, st>>>(g);             \
        else gemm_f32_kernel<true, true, T_><<<grid, 256, 0, st>>>(g);                                       \
    }
        if (t128) KF_F32G(128) else KF_F32G(64)
#undef KF_F32G
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    if (dtype == KF_F64 && M % D_T == 0 && N % D_T == 0 && K % D_BK == 0 && K > 0 && al16 && lda % 2 == 0 && ldb % 2 == 0 && !knob(KNOB_GEMM_F64_GENERIC)) {
        const unsigned grid = (unsigned)((M / D_T) * (N / D_T));
        KF_PROF("gemm_f64_mfma", st);
        if (!trans_a && !trans_b) gemm_f64_kernel<false, false><<<grid, 256, 0, st>>>(g);
        else if (!trans_a && trans_b) gemm_f64_kernel<false, true><<<grid, 256, 0, st>>>(g);
        else if (trans_a && !trans_b) gemm_f64_kernel<true, false><<<grid, 256, 0, st>>>(g);
        else gemm_f64_kernel<true, true><<<grid, 256, 0, st>>>(g);
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    if ((dtype == KF_BF16 || dtype == KF_F16) && h_fast_ok(M, N, K) && al16 && lda % 8 == 0 && ldb % 8 == 0) {
        if (h256_ok(M, N, K)) { // every operand layout is consumed in place
            // products with mul / add / aux operands take the 4-wave kernel's tail instantiation (round 3; same loop, same accumulation
            // order, so aux is bit-identical to the plain product)
            const bool w4 = h256_use_w4(M, N) || g.c_f32; // profile labels name the kernel that ran (tests assert them); the float output lives in the 4-wave kernel
            KF_PROF(dtype == KF_BF16 ? (w4 ? "gemm_bf16_mfma" : "gemm_bf16_mfma_w8") : (w4 ? "gemm_f16_mfma" : "gemm_f16_mfma_w8"), st);
            return dtype == KF_BF16 ? launch_h256<true>(g, trans_a != 0, !trans_b, w4, st) : launch_h256<false>(g, trans_a != 0, !trans_b, w4, st);
        }
        unsigned grid = (unsigned)((M / H_BM) * (N / H_BN));
        const size_t lds = (size_t)H_STAGES * 2 * H_TILE_BYTES;
        const int slices = splitk_slices(dtype, M, N, K);
        const bool split = slices > 1 && workspace && workspace_bytes >= (size_t)slices * M * N * sizeof(float) && (uintptr_t)workspace % 16 == 0 &&
                           N % 4 == 0;
        if (split) { g.split = slices; g.part = (float *)workspace; grid *= (unsigned)slices; }
        KF_PROF(dtype == KF_BF16 ? (split ? "gemm_bf16_mfma_128_splitk" : "gemm_bf16_mfma_128") : (split ? "gemm_f16_mfma_128_splitk" : "gemm_f16_mfma_128"), st);
        const bool tra = trans_a != 0, trb = !trans_b; // transposed-read operands: consumed as they lie in memory, no re-layout pass
#define KF_H128(BF_, TA, TB)                                                                                                   \
    {                                                                                                                          \
        KF_ENSURE_LDS((gemm_h_kernel<BF_, TA, TB>), (int)lds); \
        gemm_h_kernel<BF_, TA, TB><<<grid, 256, lds, st>>>(g);                                                                 \
    }
#define KF_H128_L(BF_)                                  \
    if (!tra && !trb) KF_H128(BF_, false, false)         \
    else if (!tra && trb) KF_H128(BF_, false, true)      \
    else if (tra && !trb) KF_H128(BF_, true, false)      \
    else KF_H128(BF_, true, true)
        if (dtype == KF_BF16) { KF_H128_L(true) } else { KF_H128_L(false) }
#undef KF_H128_L
#undef KF_H128
        KF_LAUNCH_CHECK();
        if (split) {
            const unsigned gf = (unsigned)((M * (N / 4) + 255) / 256);
            if (dtype == KF_BF16) gemm_splitk_fold_kernel<true><<<gf, 256, 0, st>>>(g);
            else gemm_splitk_fold_kernel<false><<<gf, 256, 0, st>>>(g);
            KF_LAUNCH_CHECK();
        }
        return KF_OK;
    }
    if (const PadPlan pp = pad_plan(dtype, M, N, K); pp.use && workspace && workspace_bytes >= pp.total && (uintptr_t)workspace % 16 == 0 &&
                                                     !(g.mul || g.add || g.aux || g.c_f32) && K > 0) {
        const int es = dtype == KF_F32 ? 4 : 2;
        char *pa = (char *)workspace, *pb = pa + pp.a_bytes, *pc = pb + pp.b_bytes, *pbias = pc + pp.c_bytes, *inner = pbias + pp.bias_bytes;
        const int64_t ar = trans_a ? K : M, ac = trans_a ? M : K, arp = trans_a ? pp.Kp : pp.Mp, acp = trans_a ? pp.Mp : pp.Kp;
        const int64_t br = trans_b ? N : K, bc = trans_b ? K : N, brp = trans_b ? pp.Np : pp.Kp, bcp = trans_b ? pp.Kp : pp.Np;
        bool use_a = false, use_b = false, use_c = false;
        // every pad / unpad launch is one thread per 16-byte piece of an image: the largest image bounds them all
        {
            const int64_t big = std::max(std::max(pp.Mp * pp.Kp, pp.Kp * pp.Np), pp.Mp * pp.Np) * es / 16;
            KF_REQUIRE((big + 255) / 256 <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "kf_gemm: a padded operand image of %lld pieces exceeds the grid limit", (long long)big);
        }
        {
            KF_PROF("gemm_pad", st);
            auto pad = [&](const void *src, int64_t ld, int64_t rows, int64_t cols, char *dst, int64_t rows_p, int64_t cols_p) {
                const int64_t total = rows_p * (cols_p * es / 16);
                const unsigned gr = (unsigned)((total + 255) / 256);
                if (es == 2) gemm_pad_copy_kernel<2><<<gr, 256, 0, st>>>((const char *)src, ld, rows, cols, dst, cols_p, total);
                else gemm_pad_copy_kernel<4><<<gr, 256, 0, st>>>((const char *)src, ld, rows, cols, dst, cols_p, total);
            };
            // an operand that already is whole tiles on 16-byte rows is read where it lies (M = 16 rows against a tile-aligned 8192 x 8192 B:
            // copying B would cost twice the product)
            const int64_t ldq = 16 / es;
            use_a = ar == arp && ac == acp && (uintptr_t)A % 16 == 0 && lda % ldq == 0;
            use_b = br == brp && bc == bcp && (uintptr_t)B % 16 == 0 && ldb % ldq == 0;
            use_c = M == pp.Mp && N == pp.Np;
            if (!use_a) pad(A, lda, ar, ac, pa, arp, acp);
            if (!use_b) pad(B, ldb, br, bc, pb, brp, bcp);
            if (!use_c && beta != 0.f) pad(C, ldc, M, N, pc, pp.Mp, pp.Np);
            if (!use_c && epilogue == KF_EPI_BIAS_ROW) pad(bias, N, 1, N, pbias, 1, pp.Np);
            KF_LAUNCH_CHECK();
        }
        const int rc = gemm_impl(dtype, trans_a, trans_b, pp.Mp, pp.Np, pp.Kp, alpha, use_a ? A : (const void *)pa, use_a ? lda : acp, use_b ? B : (const void *)pb,
                                 use_b ? ldb : bcp, beta, use_c ? C : (void *)pc, use_c ? ldc : pp.Np, epilogue,
                                 epilogue == KF_EPI_BIAS_ROW ? (use_c ? bias : (const void *)pbias) : nullptr, nullptr, stream, pp.inner_bytes ? inner : nullptr,
                                 pp.inner_bytes);
        if (rc != KF_OK) return rc;
        if (!use_c) {
            KF_PROF("gemm_pad", st);
            const int64_t total = M * ((N + 16 / es - 1) / (16 / es));
            const unsigned gr = (unsigned)((total + 255) / 256);
            if (es == 2) gemm_unpad_copy_kernel<2><<<gr, 256, 0, st>>>(pc, pp.Np, (char *)C, ldc, M, N);
            else gemm_unpad_copy_kernel<4><<<gr, 256, 0, st>>>(pc, pp.Np, (char *)C, ldc, M, N);
            KF_LAUNCH_CHECK();
        }
        return KF_OK;
    }
    if (const PadPlan pp = pad_plan(dtype, M, N, K); pp.use && workspace && !(g.mul || g.add || g.aux || g.c_f32)) {
        // a caller that DID bring scratch lands on the scalar kernel (20-40x slower): say why, once
        static bool told = false;
        if (!told) {
            told = true;
            fprintf(stderr, "[kfunca_hip] kf_gemm %lld x %lld x %lld: ragged extents need %zu bytes of 16-byte-aligned scratch (kf_gemm_workspace_bytes), got %zu at %p - "
                            "running the scalar kernel\n", (long long)M, (long long)N, (long long)K, pp.total, workspace_bytes, workspace);
        }
    }
    const int64_t gtiles = ((N + 31) / 32) * ((M + 31) / 32);
    KF_REQUIRE(gtiles <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "kf_gemm: %lld output tiles exceed the grid limit", (long long)gtiles);
    const unsigned grid = (unsigned)gtiles;
    KF_PROF("gemm_generic", st);
    switch (dtype) {
    case KF_F32: gemm_generic_kernel<float><<<grid, 256, 0, st>>>(g, trans_a, trans_b); break;
    case KF_F64: gemm_generic_kernel<double><<<grid, 256, 0, st>>>(g, trans_a, trans_b); break;
    case KF_BF16: gemm_generic_kernel<bf16_t><<<grid, 256, 0, st>>>(g, trans_a, trans_b); break;
    default: gemm_generic_kernel<f16_t><<<grid, 256, 0, st>>>(g, trans_a, trans_b); break;
    }
    KF_LAUNCH_CHECK();
    return KF_OK;
}
