// Row normalisations for gfx950: rms_norm and layer_norm, forward and backward.
//
// The reference lists rms_norm as a roadmap item (README.md:28) and ships only its building block: norm_stat_kernel
// (src/device/norm_ops_kernel.cu:6-61) computes mean and invstd = 1 / sqrt(M2 / n + eps) (src/device/utils/welford_norm.h:170-187,
// biased variance). These kernels finish it for the row-wise forms a transformer block uses, with the same invstd definition:
//   layer_norm: y = (x - mean) * rstd * w + b        rstd = 1 / sqrt(mean((x - mean)^2) + eps)
//   rms_norm:   y = x * rstd * w                     rstd = 1 / sqrt(mean(x^2) + eps)
// Both are HBM-bound byte movers: the forward reads x once and writes y once - a row lives in the registers of the lanes
// that loaded it (16-byte packs, one wave per row up to 4 KiB rows, one 256-thread block per row up to 64 KiB), so the
// statistics are the exact two-pass ones (mean first, then centred squares) at no extra traffic. The backward reads x and dy
// once and writes dx once; dw = sum_rows dy * xhat and db = sum_rows dy accumulate in registers over the rows a block walks
// and leave as one f32 partial row per block, folded in block order by a second small kernel: no atomics, bitwise reproducible.
// Rows that do not fit the register tile (or whose length is not a multiple of the pack) take a generic strided-loop kernel.
#include <math.h>

#include <algorithm>
#include <type_traits>

#include "common.h"

namespace kf {

struct NormArgs {
    const void *x, *w, *b, *dy;
    void *y, *dx;
    float *mean, *rstd;       // per-row statistics (forward: written, may be null; backward: read; mean unused for rms)
    float *part;              // backward: [gridDim.x][2][cols] f32 partial column sums (dw | db)
    int64_t rows, cols, ldx;  // ldx: row stride of x / y / dy / dx in elements
    float eps;
    int rms;
};

template <typename T> struct NPack;
template <> struct NPack<float> { static constexpr int V = 4; };
template <> struct NPack<bf16_t> { static constexpr int V = 8; };
template <> struct NPack<f16_t> { static constexpr int V = 8; };

template <typename T, int V>
__device__ __forceinline__ void n_unpack(const uint4 &p, float (&f)[V]) {
    if constexpr (sizeof(T) == 4) {
        f[0] = __uint_as_float(p.x); f[1] = __uint_as_float(p.y); f[2] = __uint_as_float(p.z); f[3] = __uint_as_float(p.w);
    } else {
        const uint32_t w[4] = {p.x, p.y, p.z, p.w};
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (std::is_same<T, bf16_t>::value) {
                f[2 * i] = __uint_as_float(w[i] << 16);
                f[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
            } else {
                f[2 * i] = f16_to_f32(f16_t{(uint16_t)(w[i] & 0xffff)});
                f[2 * i + 1] = f16_to_f32(f16_t{(uint16_t)(w[i] >> 16)});
            }
        }
    }
}
template <typename T, int V>
__device__ __forceinline__ uint4 n_pack(const float (&f)[V]) {
    uint4 p;
    if constexpr (sizeof(T) == 4) {
        p.x = __float_as_uint(f[0]); p.y = __float_as_uint(f[1]); p.z = __float_as_uint(f[2]); p.w = __float_as_uint(f[3]);
    } else {
        uint32_t w[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if constexpr (std::is_same<T, bf16_t>::value) {
                w[i] = f32x2_to_bf16x2_hw(f[2 * i], f[2 * i + 1]);
            } else {
                const uint32_t lo = f32_to_f16(f[2 * i]).x, hi = f32_to_f16(f[2 * i + 1]).x;
                w[i] = lo | (hi << 16);
            }
        }
        p.x = w[0]; p.y = w[1]; p.z = w[2]; p.w = w[3];
    }
    return p;
}

// sum over the TPR lanes that share a row (TPR = 64: one wave; 256 / 512 / 1024: the whole block, through LDS). Every lane gets the total.
template <int TPR>
__device__ __forceinline__ float n_row_sum(float v, float *red) {
#pragma unroll
    for (int m = (TPR < 64 ? TPR : 64) / 2; m > 0; m >>= 1) v += __shfl_xor(v, m, 64); // (TPR < 64: several rows per wave, each in its own lane group)
    if constexpr (TPR > 64) {
        constexpr int NW = TPR / 64;
        const int wid = threadIdx.x >> 6;
        __syncthreads(); // the previous use of red[] is over
        if ((threadIdx.x & 63) == 0) red[wid] = v;
        __syncthreads();
        if constexpr (NW == 4) {
            v = (red[0] + red[1]) + (red[2] + red[3]); // fixed order
        } else {
            float t[NW / 4];
#pragma unroll
            for (int i = 0; i < NW / 4; ++i) t[i] = (red[4 * i] + red[4 * i + 1]) + (red[4 * i + 2] + red[4 * i + 3]);
            v = t[0];
#pragma unroll
            for (int i = 1; i < NW / 4; ++i) v += t[i];
        }
    }
    return v;
}

// ---- forward: a row in registers ---------------------------------------------------------------------------------
template <typename T, int TPR, int PACKS>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const NormArgs a) {
    constexpr int V = NPack<T>::V, RPB = 256 / TPR;
    __shared__ float red[4];
    const int tr = threadIdx.x % TPR;
    const int64_t row = (int64_t)blockIdx.x * RPB + threadIdx.x / TPR;
    const bool live = row < a.rows; // whole waves (TPR = 64) or whole blocks share it: the barriers below are uniform
    const T *x = (const T *)a.x + (live ? row : 0) * a.ldx;
    float xv[PACKS][V];
    float s = 0.f;
    {
        // all of a lane's row loads are issued before the first value is touched (with the unpack and the sum inside the guarded branch
        // every pack was its own round trip); a pack behind the row's end is zeros, which add nothing to either sum
        uint4 raw[PACKS];
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            raw[p] = (live && c < a.cols) ? *(const uint4 *)(x + c) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            n_unpack<T, V>(raw[p], xv[p]);
#pragma unroll
            for (int i = 0; i < V; ++i) s += a.rms ? xv[p][i] * xv[p][i] : xv[p][i];
        }
    }
    const float inv_n = 1.0f / (float)a.cols;
    s = n_row_sum<TPR>(s, red);
    float mean = 0.f, var;
    if (a.rms) {
        var = s * inv_n;
    } else {
        mean = s * inv_n;
        float q = 0.f;
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            if (c < a.cols) {
#pragma unroll
                for (int i = 0; i < V; ++i) { const float d = xv[p][i] - mean; q += d * d; }
            }
        }
        var = n_row_sum<TPR>(q, red) * inv_n;
    }
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (live && tr == 0) {
        if (a.mean) a.mean[row] = mean;
        if (a.rstd) a.rstd[row] = rstd;
    }
    if (!live) return;
    T *y = (T *)a.y + row * a.ldx;
    // weight and bias packs four at a time: their (L2-served) loads in flight together, then the arithmetic, then the stores
    constexpr int G = PACKS < 4 ? PACKS : 4;
#pragma unroll
    for (int p0 = 0; p0 < PACKS; p0 += G) {
        uint4 wr[G], br[G];
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int64_t c = ((int64_t)(p0 + g) * TPR + tr) * V;
            const bool in = c < a.cols;
            wr[g] = (a.w && in) ? *(const uint4 *)((const T *)a.w + c) : make_uint4(0u, 0u, 0u, 0u);
            br[g] = (a.b && in) ? *(const uint4 *)((const T *)a.b + c) : make_uint4(0u, 0u, 0u, 0u);
        }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const int p = p0 + g;
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            float wv[V], bv[V], o[V];
            n_unpack<T, V>(wr[g], wv);
            n_unpack<T, V>(br[g], bv);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float t = (xv[p][i] - mean) * rstd;
                if (a.w) t *= wv[i];
                if (a.b) t += bv[i];
                o[i] = t;
            }
            if (c < a.cols) *(uint4 *)(y + c) = n_pack<T, V>(o);
        }
    }
}

// ---- backward: a block walks rows blockIdx.x, + gridDim.x, ...; x and dy of a row in registers ------------------------
//   g = dy * w;  layer: dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  rms: dx = rstd * (g - xhat * mean(g * xhat))
// RMS is a compile-time flag here: without the mean, the mean(g) term and db the bf16 row costs ~9 instead of ~13 vector instructions
// per element. A lane owns at most TWO packs of a row (16 elements of a 16-bit row): long rows spread over 512 or 1024 threads instead
// of deepening the per-lane tile, which keeps the register count near 100 and four or more waves per SIMD in flight.
// the two row sums of the backward with ONE barrier per row: the partial sums of the block's waves go through an LDS buffer that
// alternates with the row's parity (the buffer written for row r + 1 was last read for row r - 1, before row r's barrier)
template <int TPR>
__device__ __forceinline__ void n_row_sum2(float &s1, float &s2, float (*red)[2][16], int parity) {
#pragma unroll
    for (int m = (TPR < 64 ? TPR : 64) / 2; m > 0; m >>= 1) { s1 += __shfl_xor(s1, m, 64); s2 += __shfl_xor(s2, m, 64); }
    if constexpr (TPR > 64) {
        constexpr int NW = TPR / 64;
        const int wid = threadIdx.x >> 6;
        if ((threadIdx.x & 63) == 0) { red[parity][0][wid] = s1; red[parity][1][wid] = s2; }
        __syncthreads();
        float t1 = 0.f, t2 = 0.f;
#pragma unroll
        for (int i = 0; i < NW; i += 4) { // fixed order
            t1 += (red[parity][0][i] + red[parity][0][i + 1]) + (red[parity][0][i + 2] + red[parity][0][i + 3]);
            t2 += (red[parity][1][i] + red[parity][1][i + 1]) + (red[parity][1][i + 2] + red[parity][1][i + 3]);
        }
        s1 = t1; s2 = t2;
    }
}

// PF = rows requested ahead of the one being worked on (0: none - wave-per-row blocks, whose four rows and many resident blocks
// already keep the memory system busy; 2: a row spread over a whole block, where the row sums' barrier would otherwise expose a
// full memory latency per row: with one pack per lane a row ahead costs 8 registers)
template <typename T, int TPR, int PACKS, bool RMS, int PF>
__global__ __launch_bounds__(TPR < 256 ? 256 : TPR) void norm_bwd_kernel(const NormArgs a) {
    constexpr int NT = TPR < 256 ? 256 : TPR; // threads per block: four wave-rows, or ONE row across 4 / 8 / 16 waves
    constexpr int V = NPack<T>::V, RPB = NT / TPR;
    __shared__ float red[2][2][16];
    const int tr = threadIdx.x % TPR, sub = threadIdx.x / TPR;
    // Registers hold the row as it was LOADED (16-byte packs of x and dy: half the registers of their f32 images for 16-bit rows);
    // x-hat and g = dy w are recomputed for the store.
    uint4 wr[PACKS];
    float dw[PACKS][V], db[RMS ? 1 : PACKS][V];
    bool okc[PACKS];
#pragma unroll
    for (int p = 0; p < PACKS; ++p) {
        const int64_t c = ((int64_t)p * TPR + tr) * V;
        okc[p] = c < a.cols;
        // no weight = a weight of ones: g = dy w is then ONE multiply per element in both passes (the per-element select on "has a weight"
        // was 64 of the ~770 vector instructions of a 4-pack layer-norm row, in a kernel bound by its vector instructions)
        constexpr uint32_t kOne = sizeof(T) == 4 ? 0x3F800000u : (std::is_same<T, bf16_t>::value ? 0x3F803F80u : 0x3C003C00u);
        wr[p] = okc[p] ? (a.w ? *(const uint4 *)((const T *)a.w + c) : uint4{kOne, kOne, kOne, kOne}) : uint4{0, 0, 0, 0};
#pragma unroll
        for (int i = 0; i < V; ++i) {
            dw[p][i] = 0.f;
            if constexpr (!RMS) db[p][i] = 0.f;
        }
    }
    const float inv_n = 1.0f / (float)a.cols;
    const int64_t nrb = (a.rows + RPB - 1) / RPB; // row groups
    constexpr int NS = PF + 1;
    uint4 xq[NS][PACKS], dq[NS][PACKS]; // slot 0: the row being worked on; slots 1..PF: the rows after it
    auto fetch = [&](int64_t rb, uint4 (&xo)[PACKS], uint4 (&dO)[PACKS]) __attribute__((always_inline)) {
        const int64_t row = rb * RPB + sub;
        const bool live = rb < nrb && row < a.rows;
        const T *x = (const T *)a.x + (live ? row : 0) * a.ldx, *dy = (const T *)a.dy + (live ? row : 0) * a.ldx;
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            if (live && okc[p]) { xo[p] = *(const uint4 *)(x + c); dO[p] = *(const uint4 *)(dy + c); }
            else { xo[p] = uint4{0, 0, 0, 0}; dO[p] = uint4{0, 0, 0, 0}; }
        }
    };
#pragma unroll
    for (int k = 0; k < PF; ++k) fetch((int64_t)blockIdx.x + (int64_t)k * gridDim.x, xq[k], dq[k]);
    int parity = 0;
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x, parity ^= 1) {
        fetch(rb + (int64_t)PF * gridDim.x, xq[PF], dq[PF]); // PF row groups ahead (PF = 0: this one)
        const int64_t row = rb * RPB + sub;
        const bool live = row < a.rows;
        const int64_t rr = live ? row : 0;
        const float rstd = live ? a.rstd[rr] : 0.f;
        float mean = 0.f;
        if constexpr (!RMS) mean = live ? a.mean[rr] : 0.f;
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            float xv[V], dv[V], wv[V];
            n_unpack<T, V>(xq[0][p], xv);
            n_unpack<T, V>(dq[0][p], dv);
            n_unpack<T, V>(wr[p], wv);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                const float xh = RMS ? xv[i] * rstd : (xv[i] - mean) * rstd;
                const float g = dv[i] * wv[i];
                if constexpr (!RMS) s1 += g;
                s2 += g * xh;
                dw[p][i] += dv[i] * xh; // (zero packs of dead rows / columns add zero)
                if constexpr (!RMS) db[p][i] += dv[i];
            }
        }
        n_row_sum2<TPR>(s1, s2, red, parity);
        s1 *= inv_n;
        s2 *= inv_n;
        // the packs are unpacked AGAIN for the store (opaque to the optimiser, which would otherwise keep the first pass's f32 images
        // alive across the row sums)
#pragma unroll
        for (int p = 0; p < PACKS; ++p)
            asm volatile("" : "+v"(xq[0][p].x), "+v"(xq[0][p].y), "+v"(xq[0][p].z), "+v"(xq[0][p].w), "+v"(dq[0][p].x), "+v"(dq[0][p].y), "+v"(dq[0][p].z),
                         "+v"(dq[0][p].w));
        if (live) {
            T *dx = (T *)a.dx + row * a.ldx;
#pragma unroll
            for (int p = 0; p < PACKS; ++p) {
                const int64_t c = ((int64_t)p * TPR + tr) * V;
                if (okc[p]) {
                    float xv[V], dv[V], wv[V], o[V];
                    n_unpack<T, V>(xq[0][p], xv);
                    n_unpack<T, V>(dq[0][p], dv);
                    n_unpack<T, V>(wr[p], wv);
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        const float xh = RMS ? xv[i] * rstd : (xv[i] - mean) * rstd;
                        const float g = dv[i] * wv[i];
                        o[i] = RMS ? rstd * (g - xh * s2) : rstd * (g - s1 - xh * s2);
                    }
                    *(uint4 *)(dx + c) = n_pack<T, V>(o);
                }
            }
        }
        if constexpr (PF > 0) {
#pragma unroll
            for (int k = 0; k < PF; ++k)
#pragma unroll
                for (int p = 0; p < PACKS; ++p) { xq[k][p] = xq[k + 1][p]; dq[k][p] = dq[k + 1][p]; }
        }
    }
    if (!a.part) return;
    // the block's partial column sums: with RPB row slots per block (TPR = 64) the slots are added in slot order through LDS
    float *part = a.part + (int64_t)blockIdx.x * 2 * a.cols;
    if constexpr (RPB == 1) {
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            if (c < a.cols) {
#pragma unroll
                for (int i = 0; i < V; ++i) { part[c + i] = dw[p][i]; part[a.cols + c + i] = RMS ? 0.f : db[RMS ? 0 : p][i]; }
            }
        }
    } else {
        __shared__ float slab[RPB][TPR * V + 1];
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < V; ++i) slab[sub][tr * V + i] = which ? (RMS ? 0.f : db[RMS ? 0 : p][i]) : dw[p][i];
                __syncthreads();
                if (sub == 0 && c < a.cols) {
#pragma unroll
                    for (int i = 0; i < V; ++i) {
                        float t = slab[0][tr * V + i];
#pragma unroll
                        for (int r = 1; r < RPB; ++r) t += slab[r][tr * V + i];
                        part[which * a.cols + c + i] = t;
                    }
                }
            }
        }
    }
}

// dw[c] = sum over blocks of part[blk][0][c], db likewise from part[blk][1][c], in a FIXED order: a block owns 64 columns, its
// four waves take the partial rows k = wave, wave + 4, ... (eight independent running sums per lane keep the loads in flight:
// one thread walking all partial rows of a column serially took 0.26 ms for 1024 x 4096), and the four waves' sums are added in
// wave order through LDS.
template <typename T>
__global__ __launch_bounds__(256) void norm_fold_kernel(const float *part, int nblk, int64_t cols, void *dw, void *db) {
    __shared__ float red[2][4][64];
    const int cl = threadIdx.x & 63, kl = threadIdx.x >> 6;
    const int64_t c = (int64_t)blockIdx.x * 64 + cl;
    float sw[8], sb[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { sw[u] = 0.f; sb[u] = 0.f; }
    if (c < cols) {
        int k = kl;
        for (; k + 28 < nblk; k += 32) {
#pragma unroll
            for (int u = 0; u < 8; ++u) {
                sw[u] += part[((int64_t)(k + 4 * u) * 2) * cols + c];
                sb[u] += part[((int64_t)(k + 4 * u) * 2 + 1) * cols + c];
            }
        }
        for (; k < nblk; k += 4) {
            sw[0] += part[((int64_t)k * 2) * cols + c];
            sb[0] += part[((int64_t)k * 2 + 1) * cols + c];
        }
    }
    red[0][kl][cl] = ((sw[0] + sw[1]) + (sw[2] + sw[3])) + ((sw[4] + sw[5]) + (sw[6] + sw[7]));
    red[1][kl][cl] = ((sb[0] + sb[1]) + (sb[2] + sb[3])) + ((sb[4] + sb[5]) + (sb[6] + sb[7]));
    __syncthreads();
    if (kl != 0 || c >= cols) return;
    const float tw = (red[0][0][cl] + red[0][1][cl]) + (red[0][2][cl] + red[0][3][cl]);
    const float tb = (red[1][0][cl] + red[1][1][cl]) + (red[1][2][cl] + red[1][3][cl]);
    auto st = [](void *p, int64_t i, float v) {
        if constexpr (sizeof(T) == 4) ((float *)p)[i] = v;
        else if constexpr (std::is_same<T, bf16_t>::value) ((bf16_t *)p)[i] = f32_to_bf16(v);
        else ((f16_t *)p)[i] = f32_to_f16(v);
    };
    if (dw) st(dw, c, tw);
    if (db) st(db, c, tb);
}

// ---- generic rows (any length, any alignment): one block per row, strided loops, x re-read from L2 -------------------
template <typename T> __device__ __forceinline__ float n_ld(const T *p) { return (float)*p; }
template <> __device__ __forceinline__ float n_ld<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float n_ld<f16_t>(const f16_t *p) { return f16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void n_st(T *p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void n_st<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void n_st<f16_t>(f16_t *p, float v) { *p = f32_to_f16(v); }

template <typename T>
__global__ __launch_bounds__(256) void norm_fwd_generic_kernel(const NormArgs a) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T *x = (const T *)a.x + row * a.ldx;
    const float inv_n = 1.0f / (float)a.cols;
    float s = 0.f;
    for (int64_t c = threadIdx.x; c < a.cols; c += 256) { const float v = n_ld(x + c); s += a.rms ? v * v : v; }
    s = n_row_sum<256>(s, red);
    float mean = 0.f, var;
    if (a.rms) var = s * inv_n;
    else {
        mean = s * inv_n;
        float q = 0.f;
        for (int64_t c = threadIdx.x; c < a.cols; c += 256) { const float d = n_ld(x + c) - mean; q += d * d; }
        var = n_row_sum<256>(q, red) * inv_n;
    }
    const float rstd = 1.0f / sqrtf(var + a.eps);
    if (threadIdx.x == 0) {
        if (a.mean) a.mean[row] = mean;
        if (a.rstd) a.rstd[row] = rstd;
    }
    T *y = (T *)a.y + row * a.ldx;
    for (int64_t c = threadIdx.x; c < a.cols; c += 256) {
        float t = (n_ld(x + c) - mean) * rstd;
        if (a.w) t *= n_ld((const T *)a.w + c);
        if (a.b) t += n_ld((const T *)a.b + c);
        n_st(y + c, t);
    }
}

// backward, generic: dx per row here; the column sums by norm_colsum_generic_kernel (one thread per column, rows in order)
template <typename T>
__global__ __launch_bounds__(256) void norm_bwd_generic_kernel(const NormArgs a) {
    __shared__ float red[4];
    const int64_t row = blockIdx.x;
    const T *x = (const T *)a.x + row * a.ldx, *dy = (const T *)a.dy + row * a.ldx;
    const float rstd = a.rstd[row], mean = a.rms ? 0.f : a.mean[row], inv_n = 1.0f / (float)a.cols;
    float s1 = 0.f, s2 = 0.f;
    for (int64_t c = threadIdx.x; c < a.cols; c += 256) {
        const float g = n_ld(dy + c) * (a.w ? n_ld((const T *)a.w + c) : 1.f), xh = (n_ld(x + c) - mean) * rstd;
        s1 += g;
        s2 += g * xh;
    }
    s2 = n_row_sum<256>(s2, red) * inv_n;
    s1 = a.rms ? 0.f : n_row_sum<256>(s1, red) * inv_n;
    T *dx = (T *)a.dx + row * a.ldx;
    for (int64_t c = threadIdx.x; c < a.cols; c += 256) {
        const float g = n_ld(dy + c) * (a.w ? n_ld((const T *)a.w + c) : 1.f), xh = (n_ld(x + c) - mean) * rstd;
        n_st(dx + c, rstd * (g - s1 - xh * s2));
    }
}
template <typename T>
__global__ __launch_bounds__(256) void norm_colsum_generic_kernel(const NormArgs a, void *dw, void *db) {
    const int64_t c = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (c >= a.cols) return;
    float sw = 0.f, sb = 0.f;
    for (int64_t r = 0; r < a.rows; ++r) {
        const float d = n_ld((const T *)a.dy + r * a.ldx + c);
        const float xh = (n_ld((const T *)a.x + r * a.ldx + c) - (a.rms ? 0.f : a.mean[r])) * a.rstd[r];
        sw += d * xh;
        sb += d;
    }
    if (dw) n_st((T *)dw + c, sw);
    if (db) n_st((T *)db + c, sb);
}

// the register-tile plan of a row: threads per row and packs per thread (0: generic kernel)
// The forward keeps one register array of packs * V floats (up to 128: 32768-element 16-bit rows, 256 threads per row). The backward
// keeps the packs of two rows (current + prefetched, x and dy) and the dw / db accumulators: at most two packs per lane, rows up to
// 1024 threads x 2 packs (16384 16-bit elements, 8192 f32 ones).
struct NormPlan { int tpr, packs; };
static NormPlan norm_plan(int dtype, int64_t cols, int64_t ldx, const void *const *ptrs, int nptr, bool bwd) {
    const int es = dtype_size(dtype), V = 16 / es;
    if (cols % V != 0 || ldx % V != 0) return {0, 0};
    for (int i = 0; i < nptr; ++i)
        if (ptrs[i] && (uintptr_t)ptrs[i] % 16 != 0) return {0, 0};
    const int64_t npk = cols / V;
    // rows of up to 32 packs (512 B): 8, 16 or 32 lanes per row, 32 / 16 / 8 rows per block - a whole wave on a 128-byte row left 56 of its 64 lanes idle
    // (round 5, bf16 [4 Mi, 64]: forward 0.9 TB/s, backward 0.9)
    for (int t = 8; t <= 32; t *= 2)
        if (npk <= t && (!bwd || knob_int(KNOB_NORM_BWD_TPR, 0) == 0)) return {t, 1};
    if (bwd) {
        const long forced = knob_int(KNOB_NORM_BWD_TPR, 0); // A/B switch: threads per row (64 | 256 | 512 | 1024)
        for (int p = 1; p <= (forced == 64 ? 4 : 2); p *= 2) // short rows: one wave per row (no barrier), up to 2 packs per lane (4 packs without a row ahead:
            if (npk <= 64 * p && (forced == 0 || forced == 64)) return {64, p}; // bf16 [131072, 2048] 4.2 - 4.4 TB/s against 5.05 with the row across one 256-thread block)
        for (int p = 1; p <= 2; ++p) // longer ones: one pack per lane wherever 1024 threads reach (the fewest registers), two beyond
            for (int t = 256; t <= 1024; t *= 2)
                if (npk <= (int64_t)t * p && (forced == 0 || forced == t || (p == 2 && t == 1024))) return {t, p};
        return {0, 0};
    }
    if (npk <= 64 * 4) { // one wave per row, up to 4 packs per lane (4 KiB rows)
        for (int p = 1; p <= 4; p *= 2)
            if (npk <= 64 * p) return {64, p};
    }
    for (int p = 2; p <= 16; p *= 2)
        if (npk <= 256 * (int64_t)p) return {256, p};
    return {0, 0};
}
constexpr int kNormMaxBlocks = 1024; // partial rows the backward's scratch is sized for
static int norm_bwd_blocks(const NormPlan &pl, int64_t rows) { // upper bound (workspace sizing)
    const int rpb = pl.tpr < 256 ? 256 / pl.tpr : 1;
    const int64_t nrb = (rows + rpb - 1) / rpb;
    return (int)std::min<int64_t>(nrb, kNormMaxBlocks);
}
// The backward is one persistent round: as many blocks as the chip holds at once for THIS instantiation (occupancy x CUs, at most
// kNormMaxBlocks), each walking its share of the rows. A fixed 1024 blocks ran 1.33 rounds when the RMS form's register count
// let three blocks share a CU (768 resident): its last third ran at a third of the bandwidth (3.83 -> 3.48 TB/s).
template <typename T, int TPR, int PACKS, bool RMS>
static int norm_bwd_launch(NormArgs &a, hipStream_t st, int &nblk) {
    // (1024 threads leave 128 registers: the 16-bit layer form fits one row ahead, two spill; with TWO packs per lane the 16-bit forms fit one (rms) or none (layer) -
    //  round 5: both had been built with two and ran out of scratch, bf16 [21845, 12288] backward 2.2 - 2.6 TB/s)
    constexpr int PF = TPR > 64 ? ((sizeof(T) == 2 && TPR == 1024) ? (PACKS == 2 ? (RMS ? 1 : 0) : (RMS ? 2 : 1)) : 2) : 0;
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, norm_bwd_kernel<T, TPR, PACKS, RMS, PF>, TPR < 256 ? 256 : TPR, 0) != hipSuccess || per_cu < 1) per_cu = 1;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        resident = std::min(kNormMaxBlocks, per_cu * prop.multiProcessorCount);
    }
    constexpr int NT = TPR < 256 ? 256 : TPR, RPB = NT / TPR;
    const int64_t nrb = (a.rows + RPB - 1) / RPB;
    nblk = (int)std::min<int64_t>(nrb, resident);
    norm_bwd_kernel<T, TPR, PACKS, RMS, PF><<<(unsigned)nblk, NT, 0, st>>>(a);
    return KF_OK;
}

} // namespace kf

using namespace kf;

static int norm_check(const char *who, int kind, int dtype, int64_t rows, int64_t cols, int64_t ld) {
    KF_REQUIRE(kind == KF_NORM_RMS || kind == KF_NORM_LAYER, KF_ERR_INVALID, "%s: unknown norm kind %d", who, kind);
    KF_REQUIRE(dtype == KF_F32 || dtype == KF_BF16 || dtype == KF_F16, KF_ERR_UNSUPPORTED, "%s: dtype %d not supported (float, half, bfloat16)", who, dtype);
    KF_REQUIRE(rows >= 0 && cols > 0 && ld >= cols, KF_ERR_INVALID, "%s: bad extents rows %lld cols %lld ld %lld", who, (long long)rows, (long long)cols, (long long)ld);
    KF_REQUIRE(rows <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "%s: %lld rows exceed one launch", who, (long long)rows);
    return KF_OK;
}

#define KF_NORM_DISPATCH(KERNEL, T, PL, ...)                                                     \
    switch ((PL).tpr * 100 + (PL).packs) {                                                       \
    case 801: KERNEL<T, 8, 1> __VA_ARGS__; break;                                                \
    case 1601: KERNEL<T, 16, 1> __VA_ARGS__; break;                                              \
    case 3201: KERNEL<T, 32, 1> __VA_ARGS__; break;                                              \
    case 6401: KERNEL<T, 64, 1> __VA_ARGS__; break;                                              \
    case 6402: KERNEL<T, 64, 2> __VA_ARGS__; break;                                              \
    case 6404: KERNEL<T, 64, 4> __VA_ARGS__; break;                                              \
    case 25602: KERNEL<T, 256, 2> __VA_ARGS__; break;                                            \
    case 25604: KERNEL<T, 256, 4> __VA_ARGS__; break;                                            \
    case 25608: KERNEL<T, 256, 8> __VA_ARGS__; break;                                            \
    default: KERNEL<T, 256, 16> __VA_ARGS__; break;                                              \
    }
#define KF_NORM_DISPATCH_BWD(KERNEL, T, PL, RMS_, ...)                                           \
    switch ((PL).tpr * 100 + (PL).packs) {                                                       \
    case 801: KERNEL<T, 8, 1, RMS_> __VA_ARGS__; break;                                          \
    case 1601: KERNEL<T, 16, 1, RMS_> __VA_ARGS__; break;                                        \
    case 3201: KERNEL<T, 32, 1, RMS_> __VA_ARGS__; break;                                        \
    case 6401: KERNEL<T, 64, 1, RMS_> __VA_ARGS__; break;                                        \
    case 6402: KERNEL<T, 64, 2, RMS_> __VA_ARGS__; break;                                        \
    case 6404: KERNEL<T, 64, 4, RMS_> __VA_ARGS__; break;                                        \
    case 25601: KERNEL<T, 256, 1, RMS_> __VA_ARGS__; break;                                      \
    case 25602: KERNEL<T, 256, 2, RMS_> __VA_ARGS__; break;                                      \
    case 51201: KERNEL<T, 512, 1, RMS_> __VA_ARGS__; break;                                      \
    case 51202: KERNEL<T, 512, 2, RMS_> __VA_ARGS__; break;                                      \
    case 102401: KERNEL<T, 1024, 1, RMS_> __VA_ARGS__; break;                                    \
    default: KERNEL<T, 1024, 2, RMS_> __VA_ARGS__; break;                                        \
    }

extern "C" int kf_norm_fwd(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, const void *x, const void *weight, const void *bias,
                           double eps, void *y, float *mean, float *rstd, void *stream) {
    int rc = norm_check("kf_norm_fwd", kind, dtype, rows, cols, ld);
    if (rc != KF_OK) return rc;
    if (rows == 0) return KF_OK;
    KF_REQUIRE(x && y, KF_ERR_INVALID, "kf_norm_fwd: null operand");
    KF_REQUIRE(kind == KF_NORM_LAYER || !bias, KF_ERR_INVALID, "kf_norm_fwd: rms_norm takes no bias");
    hipStream_t st = as_stream(stream);
    NormArgs a{x, weight, bias, nullptr, y, nullptr, mean, rstd, nullptr, rows, cols, ld, (float)eps, kind == KF_NORM_RMS};
    const void *ptrs[4] = {x, y, weight, bias};
    const NormPlan pl = norm_plan(dtype, cols, ld, ptrs, 4, false);
    if (pl.tpr == 0) {
        KF_PROF("norm_fwd_generic", st);
        if (dtype == KF_F32) norm_fwd_generic_kernel<float><<<(unsigned)rows, 256, 0, st>>>(a);
        else if (dtype == KF_BF16) norm_fwd_generic_kernel<bf16_t><<<(unsigned)rows, 256, 0, st>>>(a);
        else norm_fwd_generic_kernel<f16_t><<<(unsigned)rows, 256, 0, st>>>(a);
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    const unsigned grid = (unsigned)((rows + (256 / pl.tpr) - 1) / (256 / pl.tpr));
    KF_PROF("norm_fwd", st);
    if (dtype == KF_F32) { KF_NORM_DISPATCH(norm_fwd_kernel, float, pl, <<<grid, 256, 0, st>>>(a)) }
    else if (dtype == KF_BF16) { KF_NORM_DISPATCH(norm_fwd_kernel, bf16_t, pl, <<<grid, 256, 0, st>>>(a)) }
    else { KF_NORM_DISPATCH(norm_fwd_kernel, f16_t, pl, <<<grid, 256, 0, st>>>(a)) }
    KF_LAUNCH_CHECK();
    return KF_OK;
}

extern "C" int kf_norm_bwd_workspace_bytes(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, size_t *bytes) {
    KF_REQUIRE(bytes, KF_ERR_INVALID, "kf_norm_bwd_workspace_bytes: null out pointer");
    *bytes = 0;
    int rc = norm_check("kf_norm_bwd_workspace_bytes", kind, dtype, rows, cols, ld);
    if (rc != KF_OK) return rc;
    // sized for the register-tile plan whether or not the pointers later turn out aligned (the generic kernels need none)
    const NormPlan pl = norm_plan(dtype, cols, ld, nullptr, 0, true);
    if (pl.tpr != 0 && rows > 0) *bytes = (size_t)norm_bwd_blocks(pl, rows) * 2 * (size_t)cols * sizeof(float);
    return KF_OK;
}

extern "C" int kf_norm_bwd(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, const void *x, const void *weight, const float *mean,
                           const float *rstd, const void *dy, void *dx, void *dweight, void *dbias, void *workspace, size_t workspace_bytes,
                           void *stream) {
    int rc = norm_check("kf_norm_bwd", kind, dtype, rows, cols, ld);
    if (rc != KF_OK) return rc;
    if (rows == 0) return KF_OK;
    KF_REQUIRE(x && dy && dx && rstd && (kind == KF_NORM_RMS || mean), KF_ERR_INVALID, "kf_norm_bwd: null operand");
    KF_REQUIRE(kind == KF_NORM_LAYER || !dbias, KF_ERR_INVALID, "kf_norm_bwd: rms_norm has no bias gradient");
    hipStream_t st = as_stream(stream);
    NormArgs a{x, weight, nullptr, dy, nullptr, dx, const_cast<float *>(mean), const_cast<float *>(rstd), nullptr, rows, cols, ld, 0.f, kind == KF_NORM_RMS};
    const void *ptrs[4] = {x, dy, dx, weight};
    const NormPlan pl = norm_plan(dtype, cols, ld, ptrs, 4, true);
    const bool sums = dweight || dbias;
    const unsigned gc = (unsigned)((cols + 255) / 256);
    if (pl.tpr == 0) {
        KF_PROF("norm_bwd_generic", st);
#define KF_NORM_GEN(T)                                                                    \
    norm_bwd_generic_kernel<T><<<(unsigned)rows, 256, 0, st>>>(a);                        \
    KF_LAUNCH_CHECK();                                                                    \
    if (sums) norm_colsum_generic_kernel<T><<<gc, 256, 0, st>>>(a, dweight, dbias);
        if (dtype == KF_F32) { KF_NORM_GEN(float) } else if (dtype == KF_BF16) { KF_NORM_GEN(bf16_t) } else { KF_NORM_GEN(f16_t) }
#undef KF_NORM_GEN
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    int nblk = norm_bwd_blocks(pl, rows);
    if (sums) {
        const size_t need = (size_t)nblk * 2 * (size_t)cols * sizeof(float);
        KF_REQUIRE(workspace && workspace_bytes >= need, KF_ERR_WORKSPACE, "kf_norm_bwd: workspace of %zu bytes required, got %zu", need, workspace_bytes);
        a.part = (float *)workspace;
    }
    {
        KF_PROF("norm_bwd", st);
#define KF_NB(T_)                                                                          \
    if (a.rms) { KF_NORM_DISPATCH_BWD(norm_bwd_launch, T_, pl, true, (a, st, nblk)) }      \
    else { KF_NORM_DISPATCH_BWD(norm_bwd_launch, T_, pl, false, (a, st, nblk)) }
        if (dtype == KF_F32) { KF_NB(float) } else if (dtype == KF_BF16) { KF_NB(bf16_t) } else { KF_NB(f16_t) }
#undef KF_NB
        KF_LAUNCH_CHECK();
    }
    if (sums) {
        KF_PROF("norm_bwd_fold", st);
        const unsigned gf = (unsigned)((cols + 63) / 64);
        if (dtype == KF_F32) norm_fold_kernel<float><<<gf, 256, 0, st>>>(a.part, nblk, cols, dweight, dbias);
        else if (dtype == KF_BF16) norm_fold_kernel<bf16_t><<<gf, 256, 0, st>>>(a.part, nblk, cols, dweight, dbias);
        else norm_fold_kernel<f16_t><<<gf, 256, 0, st>>>(a.part, nblk, cols, dweight, dbias);
        KF_LAUNCH_CHECK();
    }
    return KF_OK;
}
