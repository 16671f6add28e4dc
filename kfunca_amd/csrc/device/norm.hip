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
            uint32_t lo, hi;
            if constexpr (std::is_same<T, bf16_t>::value) { lo = f32_to_bf16(f[2 * i]).x; hi = f32_to_bf16(f[2 * i + 1]).x; }
            else { lo = f32_to_f16(f[2 * i]).x; hi = f32_to_f16(f[2 * i + 1]).x; }
            w[i] = lo | (hi << 16);
        }
        p.x = w[0]; p.y = w[1]; p.z = w[2]; p.w = w[3];
    }
    return p;
}

// sum over the TPR lanes that share a row (TPR = 64: one wave; 256: the block, through LDS). Every lane gets the total.
template <int TPR>
__device__ __forceinline__ float n_row_sum(float v, float *red) {
#pragma unroll
    for (int m = 32; m > 0; m >>= 1) v += __shfl_xor(v, m, 64);
    if constexpr (TPR == 256) {
        const int wid = threadIdx.x >> 6;
        __syncthreads(); // the previous use of red[] is over
        if ((threadIdx.x & 63) == 0) red[wid] = v;
        __syncthreads();
        v = (red[0] + red[1]) + (red[2] + red[3]); // fixed order
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
#pragma unroll
    for (int p = 0; p < PACKS; ++p) {
        const int64_t c = ((int64_t)p * TPR + tr) * V;
        if (live && c < a.cols) {
            n_unpack<T, V>(*(const uint4 *)(x + c), xv[p]);
#pragma unroll
            for (int i = 0; i < V; ++i) s += a.rms ? xv[p][i] * xv[p][i] : xv[p][i];
        } else {
#pragma unroll
            for (int i = 0; i < V; ++i) xv[p][i] = 0.f;
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
#pragma unroll
    for (int p = 0; p < PACKS; ++p) {
        const int64_t c = ((int64_t)p * TPR + tr) * V;
        if (c < a.cols) {
            float wv[V], bv[V], o[V];
            if (a.w) n_unpack<T, V>(*(const uint4 *)((const T *)a.w + c), wv);
            if (a.b) n_unpack<T, V>(*(const uint4 *)((const T *)a.b + c), bv);
#pragma unroll
            for (int i = 0; i < V; ++i) {
                float t = (xv[p][i] - mean) * rstd;
                if (a.w) t *= wv[i];
                if (a.b) t += bv[i];
                o[i] = t;
            }
            *(uint4 *)(y + c) = n_pack<T, V>(o);
        }
    }
}

// ---- backward: a block walks rows blockIdx.x, + gridDim.x, ...; x and dy of a row in registers ------------------------
//   g = dy * w;  layer: dx = rstd * (g - mean(g) - xhat * mean(g * xhat));  rms: dx = rstd * (g - xhat * mean(g * xhat))
// RMS is a compile-time flag here: without the mean, the mean(g) term and db the bf16 row costs ~9 instead of ~13 vector instructions
// per element, and this kernel is VALU-limited on 16-bit rows (6 bytes per element)
template <typename T, int TPR, int PACKS, bool RMS>
__global__ __launch_bounds__(256) void norm_bwd_kernel(const NormArgs a) {
    constexpr int V = NPack<T>::V, RPB = 256 / TPR;
    __shared__ float red[4];
    const int tr = threadIdx.x % TPR, sub = threadIdx.x / TPR;
    float dw[PACKS][V], db[PACKS][V], wv[PACKS][V];
#pragma unroll
    for (int p = 0; p < PACKS; ++p) {
        const int64_t c = ((int64_t)p * TPR + tr) * V;
        if (a.w && c < a.cols) n_unpack<T, V>(*(const uint4 *)((const T *)a.w + c), wv[p]);
#pragma unroll
        for (int i = 0; i < V; ++i) {
            dw[p][i] = 0.f; db[p][i] = 0.f;
            if (!(a.w && c < a.cols)) wv[p][i] = 1.f;
        }
    }
    const float inv_n = 1.0f / (float)a.cols;
    const int64_t nrb = (a.rows + RPB - 1) / RPB; // row groups
    for (int64_t rb = blockIdx.x; rb < nrb; rb += gridDim.x) {
        const int64_t row = rb * RPB + sub;
        const bool live = row < a.rows;
        const int64_t rr = live ? row : 0;
        const T *x = (const T *)a.x + rr * a.ldx, *dy = (const T *)a.dy + rr * a.ldx;
        const float rstd = live ? a.rstd[rr] : 0.f;
        float mean = 0.f;
        if constexpr (!RMS) mean = live ? a.mean[rr] : 0.f;
        float xh[PACKS][V], g[PACKS][V];
        float s1 = 0.f, s2 = 0.f;
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
            if (live && c < a.cols) {
                float xv[V], dv[V];
                n_unpack<T, V>(*(const uint4 *)(x + c), xv);
                n_unpack<T, V>(*(const uint4 *)(dy + c), dv);
#pragma unroll
                for (int i = 0; i < V; ++i) {
                    xh[p][i] = RMS ? xv[i] * rstd : (xv[i] - mean) * rstd;
                    g[p][i] = dv[i] * wv[p][i];
                    if constexpr (!RMS) s1 += g[p][i];
                    s2 += g[p][i] * xh[p][i];
                    dw[p][i] += dv[i] * xh[p][i];
                    if constexpr (!RMS) db[p][i] += dv[i];
                }
            } else {
#pragma unroll
                for (int i = 0; i < V; ++i) { xh[p][i] = 0.f; g[p][i] = 0.f; }
            }
        }
        s2 = n_row_sum<TPR>(s2, red) * inv_n;
        if constexpr (!RMS) s1 = n_row_sum<TPR>(s1, red) * inv_n;
        if (live) {
            T *dx = (T *)a.dx + row * a.ldx;
#pragma unroll
            for (int p = 0; p < PACKS; ++p) {
                const int64_t c = ((int64_t)p * TPR + tr) * V;
                if (c < a.cols) {
                    float o[V];
#pragma unroll
                    for (int i = 0; i < V; ++i) o[i] = RMS ? rstd * (g[p][i] - xh[p][i] * s2) : rstd * (g[p][i] - s1 - xh[p][i] * s2);
                    *(uint4 *)(dx + c) = n_pack<T, V>(o);
                }
            }
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
                for (int i = 0; i < V; ++i) { part[c + i] = dw[p][i]; part[a.cols + c + i] = db[p][i]; }
            }
        }
    } else {
        __shared__ float slab[RPB][64 * V + 1];
#pragma unroll
        for (int p = 0; p < PACKS; ++p) {
            const int64_t c = ((int64_t)p * TPR + tr) * V;
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < V; ++i) slab[sub][tr * V + i] = which ? db[p][i] : dw[p][i];
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
// The backward keeps five register arrays of packs * V floats (dw, db, w, xhat, g): its tile stops at packs * V = 32
// (8192-element rows with 256 threads); the forward keeps one (up to 128 floats: 32768-element 16-bit rows).
struct NormPlan { int tpr, packs; };
static NormPlan norm_plan(int dtype, int64_t cols, int64_t ldx, const void *const *ptrs, int nptr, bool bwd) {
    const int es = dtype_size(dtype), V = 16 / es;
    const int maxp = bwd ? 32 / V : 16;
    if (cols % V != 0 || ldx % V != 0) return {0, 0};
    for (int i = 0; i < nptr; ++i)
        if (ptrs[i] && (uintptr_t)ptrs[i] % 16 != 0) return {0, 0};
    const int64_t npk = cols / V;
    if (npk <= 64 * 4) { // one wave per row, up to 4 packs per lane (4 KiB rows)
        for (int p = 1; p <= 4; p *= 2)
            if (npk <= 64 * p) return {64, p};
    }
    for (int p = 2; p <= maxp; p *= 2)
        if (npk <= 256 * (int64_t)p) return {256, p};
    return {0, 0};
}
constexpr int kNormMaxBlocks = 1024; // partial rows the backward's scratch is sized for
static int norm_bwd_blocks(const NormPlan &pl, int64_t rows) { // upper bound (workspace sizing)
    const int64_t nrb = (rows + (256 / pl.tpr) - 1) / (256 / pl.tpr);
    return (int)std::min<int64_t>(nrb, kNormMaxBlocks);
}
// The backward is one persistent round: as many blocks as the chip holds at once for THIS instantiation (occupancy x CUs, at most
// kNormMaxBlocks), each walking its share of the rows. A fixed 1024 blocks ran 1.33 rounds when the RMS form's register count
// let three blocks share a CU (768 resident): its last third ran at a third of the bandwidth (3.83 -> 3.48 TB/s).
template <typename T, int TPR, int PACKS, bool RMS>
static int norm_bwd_launch(NormArgs &a, hipStream_t st, int &nblk) {
    static int resident = 0;
    if (resident == 0) {
        int per_cu = 0, dev = 0;
        hipDeviceProp_t prop;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, norm_bwd_kernel<T, TPR, PACKS, RMS>, 256, 0) != hipSuccess || per_cu < 1) per_cu = 2;
        if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) prop.multiProcessorCount = 256;
        resident = std::min(kNormMaxBlocks, per_cu * prop.multiProcessorCount);
    }
    constexpr int RPB = 256 / TPR;
    const int64_t nrb = (a.rows + RPB - 1) / RPB;
    nblk = (int)std::min<int64_t>(nrb, resident);
    norm_bwd_kernel<T, TPR, PACKS, RMS><<<(unsigned)nblk, 256, 0, st>>>(a);
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
    case 6401: KERNEL<T, 64, 1, RMS_> __VA_ARGS__; break;                                        \
    case 6402: KERNEL<T, 64, 2, RMS_> __VA_ARGS__; break;                                        \
    case 6404: KERNEL<T, 64, 4, RMS_> __VA_ARGS__; break;                                        \
    case 25602: KERNEL<T, 256, 2, RMS_> __VA_ARGS__; break;                                      \
    case 25604: KERNEL<T, 256, 4, RMS_> __VA_ARGS__; break;                                      \
    default:                                                                                     \
        if constexpr (sizeof(T) == 4) { KERNEL<T, 256, 8, RMS_> __VA_ARGS__; }                   \
        break;                                                                                   \
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
