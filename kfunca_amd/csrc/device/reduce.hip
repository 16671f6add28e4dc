// sum / mean / moment (mean + variance) reductions for gfx950 (keepdim over one logical dim, any layout).
//
// Replaces the reference's reduce engine (src/device/utils/tensor_reduce.h:35-1083 driven by
// src/device/reduce_ops_kernel.cu:6-59). Native wave64 design rather than its 32-lane one:
//   inner  — the reduced dim is contiguous: rows are streamed with coalesced (16-B when aligned)
//            loads, lanes of one row combine with 64-wide wave shuffles, rows wider than a wave
//            finish through LDS;
//   outer  — the reduced dim is strided and an output dim is contiguous (column sums): lanes walk
//            columns (coalesced), 4 row-groups per block combine through LDS;
//   generic— anything else: one lane per output element.
// Few-outputs/many-inputs shapes split the reduced extent over gridDim.y into an f32/f64/i64
// partial buffer supplied by the caller, and a second tiny kernel folds the partials in a fixed
// order — no atomics, no semaphores (the reference's last-block-done semaphores are never zeroed,
// tensor_reduce.h:748,1057-1058), bitwise reproducible run to run.
// Accumulation: float for half/bf16/float (the reference accumulates sum in the input dtype,
// reduce_ops_kernel.cu:13-18 — ours is at least as accurate and inside its 1e-2 test tolerance),
// double for double, int64 for integers (identical to in-dtype wraparound after truncation).
// mean multiplies by the reference's factor = nout/numel evaluated in the input dtype
// (reduce_ops_kernel.cu:49-53) — for integer dtypes that is an integer quotient (0 unless R == 1).
// Moments (mean_var_kernel, reduce_ops_kernel.cu:61-153, and norm_stat_kernel, norm_ops_kernel.cu:6-61) run on
// the same three paths with a (mean, M2, n) accumulator: a lane folds each 16-byte pack into its running
// triple with Chan's pairwise update (one reciprocal per pack instead of the reference's one division per
// element, WelfordOps::reduce), lanes / row groups / splits combine with the same formula
// (WelfordOps::combine), and the projection writes two outputs: var | sqrt(var) | 1/sqrt(var_biased + eps),
// and the mean.
#include <type_traits>

#include "common.h"
#include "offset_calc.h"

namespace kf {

constexpr int kRB = 256; // threads per block

template <typename T> struct RAcc { using type = int64_t; };
template <> struct RAcc<float> { using type = float; };
template <> struct RAcc<bf16_t> { using type = float; };
template <> struct RAcc<f16_t> { using type = float; };
template <> struct RAcc<double> { using type = double; };

template <typename T> __device__ __forceinline__ typename RAcc<T>::type r_load(const char *p) { return (typename RAcc<T>::type)(*(const T *)p); }
template <> __device__ __forceinline__ float r_load<bf16_t>(const char *p) { return bf16_to_f32(*(const bf16_t *)p); }
template <> __device__ __forceinline__ float r_load<f16_t>(const char *p) { return f16_to_f32(*(const f16_t *)p); }
template <> __device__ __forceinline__ int64_t r_load<bool>(const char *p) { return *(const uint8_t *)p != 0; }

template <typename T> __device__ __forceinline__ void r_store(char *p, typename RAcc<T>::type v) { *(T *)p = (T)v; }
template <> __device__ __forceinline__ void r_store<bf16_t>(char *p, float v) { *(bf16_t *)p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void r_store<f16_t>(char *p, float v) { *(f16_t *)p = f32_to_f16(v); }
template <> __device__ __forceinline__ void r_store<bool>(char *p, int64_t v) { *(uint8_t *)p = (uint8_t)(v != 0); }

template <typename A> __device__ __forceinline__ A shfl_xor_acc(A v, int m) { return __shfl_xor(v, m, 64); }
template <> __device__ __forceinline__ int64_t shfl_xor_acc<int64_t>(int64_t v, int m) {
    int lo = __shfl_xor((int)(uint32_t)v, m, 64), hi = __shfl_xor((int)(v >> 32), m, 64);
    return ((int64_t)hi << 32) | (uint32_t)lo;
}
template <> __device__ __forceinline__ double shfl_xor_acc<double>(double v, int m) {
    int64_t b = __double_as_longlong(v);
    return __longlong_as_double(shfl_xor_acc<int64_t>(b, m));
}

struct RedArgs {
    const char *in;
    char *out;
    char *out1;        // moments: the mean output (out = variance-like output)
    void *ws;          // partials [nsplit][nout] of the accumulate type (nsplit > 1)
    int64_t R;         // reduced extent
    int64_t r_stride;  // bytes between consecutive reduced elements
    int64_t C;         // outer path: contiguous output extent (dim 1)
    uint32_t nout;     // total outputs
    uint32_t nouter;   // outer path: outputs beyond dim 1
    int nsplit;
    int tx;            // inner path: lanes per row (power of two <= 256)
    OffsetCalc<3> oc;  // output-dims calculator: [0] = out bytes, [1] = in bytes, [2] = out1 bytes
    OffsetCalc<1> rc;  // generic path: reduced-dims calculator (in bytes)
    uint32_t rtot;     // generic path: total reduced elements
};

// ---- reduction policies ------------------------------------------------------------------------
// Ops::A accumulator, Ops::add_pack folds VEC loaded elements, Ops::comb merges two accumulators (the
// engine fixes the order, so results are reproducible), Ops::Fin is the projection + store.
template <typename T>
struct SumOps {
    static constexpr bool kPackRows = false; // (sums keep one running sum per row group: the summation order tests and fixtures know)
    using X = typename RAcc<T>::type;
    using A = X;
    struct Fin {
        A factor;   // mean: nout/numel in the input dtype; sum: 1
        int scale;  // 0: sum (no multiply)
        __device__ __forceinline__ void store(const RedArgs &a, uint32_t o0, uint32_t, uint32_t e, A v) const {
            r_store<T>(a.out + o0 + e * sizeof(T), scale ? v * factor : v);
        }
    };
    static __device__ __forceinline__ A zero() { return A(0); }
    template <int VEC> static __device__ __forceinline__ void add_pack(A &a, const X (&x)[VEC]) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) a += x[e];
    }
    static __device__ __forceinline__ A comb(A a, A b) { return a + b; }
    static __device__ __forceinline__ A shfl(A a, int m) { return shfl_xor_acc<A>(a, m); }
};

template <typename S> __device__ __forceinline__ S fast_rcp(S v) { return S(1) / v; }
template <> __device__ __forceinline__ float fast_rcp<float>(float v) { return __frcp_rn(v); }

template <typename T>
struct MomentOps {
    static constexpr bool kPackRows = true; // reduce_outer_kernel folds four rows of a column as one pack
    using X = typename RAcc<T>::type; // float | double
    struct A { X mean, m2, n; };
    struct Fin {
        X correction, eps;
        int mode;      // KF_MOM_VAR | KF_MOM_STD | KF_MOM_INVSTD
        int out_f32;   // 16-bit inputs may write f32 statistics (the reference's acc_type outputs, norm_ops_kernel.cu:13-15)
        __device__ __forceinline__ void put(char *p, uint32_t e, X v) const {
            if (out_f32) *(float *)(p + e * 4) = (float)v;
            else r_store<T>(p + e * sizeof(T), v);
        }
        __device__ __forceinline__ void store(const RedArgs &a, uint32_t o0, uint32_t o1, uint32_t e, A v) const {
            X r;
            if (mode == KF_MOM_INVSTD) {
                r = X(1) / sqrt(v.m2 / v.n + eps); // welford_norm.h:183
            } else {
                const X div = v.n > correction ? v.n - correction : X(0); // WelfordOps::project, reduce_ops_kernel.cu:122-128
                r = v.m2 / div;
                if (mode == KF_MOM_STD) r = sqrt(r);
            }
            put(a.out + o0, e, r);
            put(a.out1 + o1, e, v.mean);
        }
    };
    static __device__ __forceinline__ A zero() { return A{X(0), X(0), X(0)}; }
    template <int VEC> static __device__ __forceinline__ void add_pack(A &a, const X (&x)[VEC]) {
        X pm = x[0];
#pragma unroll
        for (int e = 1; e < VEC; ++e) pm += x[e];
        pm *= X(1) / X(VEC);
        X pm2 = X(0);
#pragma unroll
        for (int e = 0; e < VEC; ++e) pm2 += (x[e] - pm) * (x[e] - pm);
        const X n1 = a.n + X(VEC), w = X(VEC) * fast_rcp<X>(n1), d = pm - a.mean;
        a.mean += d * w;
        a.m2 += pm2 + d * d * a.n * w;
        a.n = n1;
    }
    static __device__ __forceinline__ A comb(A a, A b) { // WelfordOps::combine, reduce_ops_kernel.cu:104-121
        const X n = a.n + b.n;
        if (n == X(0)) return a;
        const X w = b.n / n, d = b.mean - a.mean;
        return A{a.mean + d * w, a.m2 + b.m2 + d * d * a.n * w, n};
    }
    static __device__ __forceinline__ A shfl(A a, int m) {
        return A{shfl_xor_acc<X>(a.mean, m), shfl_xor_acc<X>(a.m2, m), shfl_xor_acc<X>(a.n, m)};
    }
};

template <typename T, int VEC>
struct __attribute__((packed, aligned(sizeof(T)))) RPack { T v[VEC]; }; // (16-byte accesses at element alignment: see elementwise.hip's Pack)

template <typename Ops, typename T, int VEC>
__device__ __forceinline__ void fold_pack(typename Ops::A &acc, const RPack<T, VEC> &p) {
    typename Ops::X x[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) x[e] = r_load<T>((const char *)&p.v[e]);
    Ops::template add_pack<VEC>(acc, x);
}

// ---- inner: reduced dim contiguous ---------------------------------------------------------
template <typename T, typename Ops, int VEC>
__global__ __launch_bounds__(kRB) void reduce_inner_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    __shared__ A smem[kRB];
    const int tx = a.tx, ty = kRB / tx;
    const int lx = threadIdx.x % tx, ly = threadIdx.x / tx;
    const uint32_t o = blockIdx.x * ty + ly;
    const bool live = o < a.nout;
    A acc0 = Ops::zero(), acc1 = Ops::zero(), acc2 = Ops::zero(), acc3 = Ops::zero();
    uint32_t off[3] = {0, 0, 0};
    if (live) {
        a.oc.get(o, off);
        const int64_t chunk = ((a.R / VEC + a.nsplit - 1) / a.nsplit) * VEC;
        const int64_t r0 = (int64_t)blockIdx.y * chunk;
        const int64_t r1 = r0 + chunk < a.R ? r0 + chunk : a.R;
        const char *row = a.in + off[1];
        const int64_t step = (int64_t)tx * VEC;
        int64_t r = r0 + (int64_t)lx * VEC;
        for (; r + 3 * step < r1; r += 4 * step) {
            RPack<T, VEC> p0 = *(const RPack<T, VEC> *)(row + r * sizeof(T));
            RPack<T, VEC> p1 = *(const RPack<T, VEC> *)(row + (r + step) * sizeof(T));
            RPack<T, VEC> p2 = *(const RPack<T, VEC> *)(row + (r + 2 * step) * sizeof(T));
            RPack<T, VEC> p3 = *(const RPack<T, VEC> *)(row + (r + 3 * step) * sizeof(T));
            fold_pack<Ops, T, VEC>(acc0, p0);
            fold_pack<Ops, T, VEC>(acc1, p1);
            fold_pack<Ops, T, VEC>(acc2, p2);
            fold_pack<Ops, T, VEC>(acc3, p3);
        }
        for (; r < r1; r += step) {
            RPack<T, VEC> p0 = *(const RPack<T, VEC> *)(row + r * sizeof(T));
            fold_pack<Ops, T, VEC>(acc0, p0);
        }
    }
    A acc = Ops::comb(Ops::comb(acc0, acc1), Ops::comb(acc2, acc3));
    if (tx <= kWave) {
        for (int m = tx >> 1; m > 0; m >>= 1) acc = Ops::comb(acc, Ops::shfl(acc, m));
    } else {
        smem[threadIdx.x] = acc;
        __syncthreads();
        for (int s = tx >> 1; s >= kWave; s >>= 1) {
            if (lx < s) smem[threadIdx.x] = Ops::comb(smem[threadIdx.x], smem[threadIdx.x + s]);
            __syncthreads();
        }
        acc = smem[threadIdx.x];
        if (lx < kWave)
            for (int m = kWave >> 1; m > 0; m >>= 1) acc = Ops::comb(acc, Ops::shfl(acc, m));
    }
    if (live && lx == 0) {
        if (a.nsplit == 1)
            fin.store(a, off[0], off[2], 0, acc);
        else
            ((A *)a.ws)[(size_t)blockIdx.y * a.nout + o] = acc;
    }
}

// ---- row sums over a FEW columns (one or two packs per row; round 5): one lane per row, its packs in flight together, no shuffles; a wave
// stores 64 consecutive results (sum(1) of f32 [32 Mi, 8]: two lanes per row and a shuffle step ran at 3.9 TB/s)
template <typename T, typename Ops, int VEC>
__global__ __launch_bounds__(256) void reduce_inner_few_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    const uint32_t o = blockIdx.x * 256 + threadIdx.x;
    if (o >= a.nout) return;
    uint32_t off[3];
    a.oc.get(o, off);
    const char *row = a.in + off[1];
    const int64_t npk = a.R / VEC; // (<= 4, whole packs)
    RPack<T, VEC> p[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) p[q] = *(const RPack<T, VEC> *)(row + (int64_t)(q < npk ? q : npk - 1) * (VEC * sizeof(T))); // (clamped, no branch around the load)
    A acc = Ops::zero();
#pragma unroll
    for (int q = 0; q < 4; ++q)
        if (q < npk) fold_pack<Ops, T, VEC>(acc, p[q]);
    fin.store(a, off[0], off[2], 0, acc);
}

// ---- column sums over a FEW rows (R <= 8; round 5): one lane per pack of columns, the rows' packs in flight together, rows added in order -
// no LDS, no barrier, no idle row groups (mean(0) of bf16 [2, 128 Mi] ran at 2.6 TB/s through the four-row-group kernel below)
template <typename T, typename Ops, int VEC>
__global__ __launch_bounds__(256) void reduce_outer_few_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    using X = typename Ops::X;
    const int64_t c = ((int64_t)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (c >= a.C) return;
    for (uint32_t z = blockIdx.z; z < a.nouter; z += gridDim.z) {
        uint32_t off[3];
        a.oc.get(z, off);
        const char *col = a.in + off[1] + c * sizeof(T);
        RPack<T, VEC> p[8];
#pragma unroll
        for (int r = 0; r < 8; ++r) p[r] = *(const RPack<T, VEC> *)(col + (int64_t)(r < a.R ? r : a.R - 1) * a.r_stride); // (clamped, no branch around the load)
        A acc[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) acc[e] = Ops::zero();
#pragma unroll
        for (int r = 0; r < 8; ++r)
            if (r < a.R) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const X x[1] = {r_load<T>((const char *)&p[r].v[e])};
                    Ops::template add_pack<1>(acc[e], x);
                }
            }
#pragma unroll
        for (int e = 0; e < VEC; ++e) fin.store(a, off[0], off[2], (uint32_t)(c + e), acc[e]);
    }
}

// ---- outer: reduced dim strided, dim 1 contiguous (column sums) ------------------------------
template <typename T, typename Ops, int VEC>
__global__ __launch_bounds__(kRB) void reduce_outer_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    using X = typename Ops::X;
    __shared__ A smem[3][kWave][VEC];
    const int lx = threadIdx.x & 63, ly = threadIdx.x >> 6; // 64 column-lanes x 4 row groups
    const int64_t c = ((int64_t)blockIdx.x * kWave + lx) * VEC;
    const bool live = c < a.C;
    const int64_t chunk = (a.R + a.nsplit - 1) / a.nsplit;
    const int64_t r0 = (int64_t)blockIdx.y * chunk;
    const int64_t r1 = r0 + chunk < a.R ? r0 + chunk : a.R;
    for (uint32_t z = blockIdx.z; z < a.nouter; z += gridDim.z) {
        uint32_t off[3];
        a.oc.get(z, off); // oc walks dims >= 2 here
        A acc[4][VEC];
#pragma unroll
        for (int u = 0; u < 4; ++u)
#pragma unroll
            for (int e = 0; e < VEC; ++e) acc[u][e] = Ops::zero();
        if (live) {
            const char *col = a.in + off[1] + c * sizeof(T);
            int64_t r = r0 + ly;
            for (; r + 12 < r1; r += 16) {
                RPack<T, VEC> p[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) p[u] = *(const RPack<T, VEC> *)(col + (r + 4 * u) * a.r_stride);
                if constexpr (Ops::kPackRows) {
                    // moments: the four rows of a column fold as ONE pack (Chan's pairwise update: one reciprocal and one dependent
                    // mean / M2 step per four elements instead of per element - the column statistics ran at 4.8 TB/s against the
                    // column sums' 5.5 on the per-element form)
#pragma unroll
                    for (int e = 0; e < VEC; ++e) {
                        const X x[4] = {r_load<T>((const char *)&p[0].v[e]), r_load<T>((const char *)&p[1].v[e]), r_load<T>((const char *)&p[2].v[e]),
                                        r_load<T>((const char *)&p[3].v[e])};
                        Ops::template add_pack<4>(acc[0][e], x);
                    }
                } else {
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int e = 0; e < VEC; ++e) {
                            const X x[1] = {r_load<T>((const char *)&p[u].v[e])};
                            Ops::template add_pack<1>(acc[u][e], x);
                        }
                }
            }
            for (; r < r1; r += 4) {
                RPack<T, VEC> p0 = *(const RPack<T, VEC> *)(col + r * a.r_stride);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const X x[1] = {r_load<T>((const char *)&p0.v[e])};
                    Ops::template add_pack<1>(acc[0][e], x);
                }
            }
        }
        A tot[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) tot[e] = Ops::comb(Ops::comb(acc[0][e], acc[1][e]), Ops::comb(acc[2][e], acc[3][e]));
        __syncthreads();
        if (ly > 0)
#pragma unroll
            for (int e = 0; e < VEC; ++e) smem[ly - 1][lx][e] = tot[e];
        __syncthreads();
        if (ly == 0 && live) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) tot[e] = Ops::comb(tot[e], Ops::comb(Ops::comb(smem[0][lx][e], smem[1][lx][e]), smem[2][lx][e]));
            if (a.nsplit == 1) {
#pragma unroll
                for (int e = 0; e < VEC; ++e) fin.store(a, off[0], off[2], (uint32_t)(c + e), tot[e]);
            } else {
                A *w = (A *)a.ws + (size_t)blockIdx.y * a.nout + (size_t)z * a.C + c;
#pragma unroll
                for (int e = 0; e < VEC; ++e) w[e] = tot[e];
            }
        }
    }
}

// ---- outer, small problems: ONE launch. A block is 16 columns x 16 row groups (a split over blocks would need the fold
// launch: two launches of ~10 us each are what a 1024 x 1024 column sum cost). Each lane walks R / 16 rows with 8 loads in
// flight, the 16 row groups meet in LDS in fixed order - same bits run to run.
template <typename T, typename Ops>
__global__ __launch_bounds__(kRB) void reduce_outer_tall_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    using X = typename Ops::X;
    __shared__ A smem[16][17];
    const int lx = threadIdx.x & 15, ly = threadIdx.x >> 4;
    // 16 columns of a 4-byte type are HALF a 128-byte line: the block with the other half must sit on the same XCD (block ids are dealt
    // round-robin over the 8 XCDs, each with its own L2), or every line is fetched twice (rocprofv3, C1 sum(0): 2.01x the algorithmic bytes).
    // Ids b and b + 8 share an XCD: they take neighbouring column groups.
    const unsigned nb = gridDim.x, cb = (nb % 8 == 0) ? (blockIdx.x & 7u) * (nb >> 3) + (blockIdx.x >> 3) : blockIdx.x;
    const int64_t c = (int64_t)cb * 16 + lx;
    const bool live = c < a.C;
    for (uint32_t z = blockIdx.z; z < a.nouter; z += gridDim.z) {
        uint32_t off[3];
        a.oc.get(z, off);
        A acc[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u] = Ops::zero();
        if (live) {
            const char *col = a.in + off[1] + c * sizeof(T);
            int64_t r = ly;
            for (; r + 112 < a.R; r += 128) {
                X x[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = r_load<T>(col + (r + 16 * u) * a.r_stride);
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const X one[1] = {x[u]};
                    Ops::template add_pack<1>(acc[u], one);
                }
            }
            for (; r < a.R; r += 16) {
                const X one[1] = {r_load<T>(col + r * a.r_stride)};
                Ops::template add_pack<1>(acc[0], one);
            }
        }
        A tot = Ops::comb(Ops::comb(Ops::comb(acc[0], acc[1]), Ops::comb(acc[2], acc[3])), Ops::comb(Ops::comb(acc[4], acc[5]), Ops::comb(acc[6], acc[7])));
        __syncthreads();
        smem[ly][lx] = tot;
        __syncthreads();
        if (ly == 0 && live) {
            A t = smem[0][lx];
            for (int g = 1; g < 16; ++g) t = Ops::comb(t, smem[g][lx]);
            fin.store(a, off[0], off[2], (uint32_t)c, t);
        }
    }
}

// second stage: fold partials[nsplit][nout] in split order
template <typename T, typename Ops>
__global__ __launch_bounds__(kRB) void reduce_fold_kernel(const RedArgs a, const typename Ops::Fin fin, int outer_layout) {
    using A = typename Ops::A;
    const uint32_t o = blockIdx.x * kRB + threadIdx.x;
    if (o >= a.nout) return;
    A acc = ((const A *)a.ws)[o];
    for (int s = 1; s < a.nsplit; ++s) acc = Ops::comb(acc, ((const A *)a.ws)[(size_t)s * a.nout + o]);
    uint32_t off[3];
    uint32_t c = 0;
    if (outer_layout) { // o = z * C + c
        const uint32_t z = o / (uint32_t)a.C;
        c = o - z * (uint32_t)a.C;
        a.oc.get(z, off);
    } else {
        a.oc.get(o, off);
    }
    fin.store(a, off[0], off[2], c, acc);
}

// ---- generic: one lane per output -----------------------------------------------------------
template <typename T, typename Ops>
__global__ __launch_bounds__(kRB) void reduce_generic_kernel(const RedArgs a, const typename Ops::Fin fin) {
    using A = typename Ops::A;
    using X = typename Ops::X;
    const uint32_t o = blockIdx.x * kRB + threadIdx.x;
    if (o >= a.nout) return;
    uint32_t off[3];
    a.oc.get(o, off);
    const char *base = a.in + off[1];
    A acc = Ops::zero();
    for (uint32_t r = 0; r < a.rtot; ++r) {
        uint32_t ro[1];
        a.rc.get(r, ro);
        const X x[1] = {r_load<T>(base + ro[0])};
        Ops::template add_pack<1>(acc, x);
    }
    fin.store(a, off[0], off[2], 0, acc);
}

// ------------------------------------------------------------------------------------------
enum { PATH_INNER = 0, PATH_OUTER = 1, PATH_GENERIC = 2 };

struct Plan {
    int path = PATH_GENERIC;
    int vec = 1;
    int tx = 1;
    int nsplit = 1;
    bool tall = false; // outer path: single-launch 16-column x 16-row-group blocks (small problems)
    int64_t R = 1, r_stride = 0, C = 1;
    int64_t nout = 1, nouter = 1, rtot = 1;
    int red_dims[KF_MAX_DIMS], nred = 0;
    int out_dims[KF_MAX_DIMS], nod = 0;
    size_t ws_bytes = 0;
};

static int pow2_floor(int64_t v) {
    int p = 1;
    while ((int64_t)p * 2 <= v) p *= 2;
    return p;
}

// `moments`: 2 outputs (variance-like, mean) + 1 input, in the iterator's order (reduce_ops.cpp:24-25)
static int make_plan(const kf_iter_desc *d, Plan &p, bool moments = false) {
    const int in = moments ? 2 : 1;
    if (moments) {
        KF_REQUIRE(d->noutputs == 2 && d->ntensors == 3, KF_ERR_INVALID, "kf_reduce_moments: wants 2 outputs + 1 input");
        const int dt = d->dtype[2];
        KF_REQUIRE(dt == KF_F32 || dt == KF_F64 || dt == KF_F16 || dt == KF_BF16, KF_ERR_UNSUPPORTED,
                   "kf_reduce_moments: floating dtypes only (mean_var_kernel dispatches on floating types, reduce_ops_kernel.cu:149-153)");
        KF_REQUIRE(d->dtype[0] == d->dtype[1] && (d->dtype[0] == dt || (d->dtype[0] == KF_F32 && (dt == KF_F16 || dt == KF_BF16))),
                   KF_ERR_UNSUPPORTED, "kf_reduce_moments: outputs must have the input dtype (or f32 for 16-bit inputs)");
        for (int i = 0; i < d->ndim; ++i)
            KF_REQUIRE((d->stride_bytes[0][i] == 0) == (d->stride_bytes[1][i] == 0) || d->shape[i] == 1, KF_ERR_INVALID,
                       "kf_reduce_moments: the two outputs must reduce the same dims");
    } else {
        KF_REQUIRE(d->noutputs == 1 && d->ntensors == 2, KF_ERR_INVALID, "kf_reduce: wants 1 output + 1 input");
        KF_REQUIRE(d->dtype[0] == d->dtype[1], KF_ERR_UNSUPPORTED, "kf_reduce: output dtype must equal input dtype");
    }
    KF_REQUIRE(d->ndim >= 1 && d->ndim <= KF_MAX_DIMS, KF_ERR_INVALID, "kf_reduce: ndim out of range");
    const int es = dtype_size(d->dtype[in]);
    const int eso = dtype_size(d->dtype[0]);
    KF_REQUIRE(es > 0, KF_ERR_INVALID, "kf_reduce: bad dtype");
    for (int i = 0; i < d->ndim; ++i) {
        if (d->stride_bytes[0][i] == 0 && d->shape[i] > 1) {
            p.red_dims[p.nred++] = i;
            p.rtot *= d->shape[i];
        } else {
            p.out_dims[p.nod++] = i;
            p.nout *= d->shape[i];
        }
    }
    KF_REQUIRE(desc_is_32bit(d), KF_ERR_INDEX_RANGE, "kf_reduce: descriptor is not 32-bit indexable");
    const int acc_bytes = ((d->dtype[in] == KF_F32 || d->dtype[in] == KF_F16 || d->dtype[in] == KF_BF16) ? 4 : 8) * (moments ? 3 : 1);
    const int64_t target_blocks = 1024;
    const int64_t in_s0 = d->stride_bytes[in][0];
    if (p.nred == 1 && p.red_dims[0] == 0 && in_s0 == es) {
        p.path = PATH_INNER;
        p.R = d->shape[0];
        p.r_stride = es;
        int vec = 16 / es;
        for (; vec > 1; vec >>= 1) { // rows must start on a pack boundary
            bool ok = p.R % vec == 0 && (uintptr_t)d->data[in] % es == 0; // (element alignment is enough for a pack)
            if (ok) break;
        }
        p.vec = vec == 16 / es ? vec : 1;
        int64_t per = (p.R + p.vec - 1) / p.vec;
        p.tx = per >= kRB ? kRB : pow2_floor(per < 1 ? 1 : per);
        if (p.tx < per && p.tx < kRB) p.tx *= 2;
        // many short rows: four packs per lane instead of one (a quarter of the lanes per row, four times the rows per block): the per-row work -
        // offset calculation, shuffle tree, the one-lane store - is shared by four times the bytes ([1 Mi rows x 256] f32: 3.6 -> see DESIGN §4)
        if (p.tx >= 4 && p.tx < kRB && (p.nout * (p.tx / 4) + kRB - 1) / kRB >= target_blocks) p.tx /= 4;
        const int ty = kRB / p.tx;
        const int64_t gx = (p.nout + ty - 1) / ty;
        int64_t ns = 1;
        if (gx < target_blocks / 2) {
            ns = target_blocks / gx;
            const int64_t max_split = p.R / ((int64_t)p.tx * p.vec * 8); // >= 8 packs per lane per split
            if (ns > max_split) ns = max_split;
            if (ns > 256) ns = 256;
            if (ns < 1) ns = 1;
        }
        p.nsplit = (int)ns;
    } else if (p.nred == 1 && p.red_dims[0] == 0 && d->ndim >= 2 && d->stride_bytes[in][1] == es &&
               d->stride_bytes[0][1] == eso && (!moments || d->stride_bytes[1][1] == eso)) {
        p.path = PATH_OUTER;
        p.R = d->shape[0];
        p.r_stride = in_s0;
        p.C = d->shape[1];
        p.nouter = p.nout / p.C;
        int vec = 16 / es;
        const int64_t vb = 16;
        // (16-byte alignment stays a condition HERE: column packs at an odd element offset measured slower than one element per lane -
        //  x[:, 1:4097].sum(0) 2.9 against 3.5 TB/s - unlike the row walks of the inner path and of the elementwise kernels)
        bool ok = p.C % vec == 0 && (uintptr_t)d->data[in] % vb == 0 && in_s0 % vb == 0;
        for (int i = 2; ok && i < d->ndim; ++i)
            if (d->stride_bytes[in][i] % vb) ok = false;
        p.vec = ok ? vec : 1;
        const int64_t gx = (p.C + (int64_t)kWave * p.vec - 1) / ((int64_t)kWave * p.vec);
        const int64_t gz = p.nouter < 1024 ? p.nouter : 1024;
        int64_t ns = 1;
        if (gx * gz < target_blocks / 2) {
            ns = target_blocks / (gx * gz);
            const int64_t max_split = p.R / 16;
            if (ns > max_split) ns = max_split;
            if (ns > 256) ns = 256;
            if (ns < 1) ns = 1;
        }
        p.nsplit = (int)ns;
        // small column reductions that would be split: one launch of 16-column x 16-row-group blocks instead of split + fold
        const int64_t in_bytes = p.R * p.C * p.nouter * es;
        if (ns > 1 && in_bytes <= ((int64_t)16 << 20) && (p.C + 15) / 16 * p.nouter >= 32 && !knob(KNOB_REDUCE_NO_TALL)) {
            p.tall = true;
            p.nsplit = 1;
        }
    } else {
        p.path = PATH_GENERIC;
    }
    p.ws_bytes = p.nsplit > 1 ? (size_t)p.nsplit * p.nout * acc_bytes : 0;
    return KF_OK;
}

template <typename T, typename Ops>
static int run_reduce(const char *what, const kf_iter_desc *d, const Plan &p, const typename Ops::Fin &fin, void *ws, hipStream_t st) {
    const bool moments = d->noutputs == 2;
    const int in = moments ? 2 : 1;
    KF_PROF(moments ? (p.path == PATH_INNER ? "moments_inner" : p.path == PATH_OUTER ? "moments_outer" : "moments_generic")
                    : (p.path == PATH_INNER ? "reduce_inner" : p.path == PATH_OUTER ? "reduce_outer" : "reduce_generic"), st);
    (void)what;
    RedArgs a;
    memset(&a, 0, sizeof(a));
    a.in = (const char *)d->data[in];
    a.out = (char *)d->data[0];
    a.out1 = moments ? (char *)d->data[1] : nullptr;
    a.ws = ws;
    a.R = p.R;
    a.r_stride = p.r_stride;
    a.C = p.C;
    a.nout = (uint32_t)p.nout;
    a.nouter = (uint32_t)p.nouter;
    a.nsplit = p.nsplit;
    a.tx = p.tx;
    a.rtot = (uint32_t)p.rtot;

    // sub-descriptor holding only the output dims (optionally skipping the first `skip` of them)
    auto build_out_calc = [&](int skip) {
        kf_iter_desc s;
        memset(&s, 0, sizeof(s));
        s.ntensors = 3;
        s.noutputs = 1;
        int n = 0;
        for (int k = skip; k < p.nod; ++k) {
            const int i = p.out_dims[k];
            s.shape[n] = d->shape[i];
            s.stride_bytes[0][n] = d->stride_bytes[0][i];
            s.stride_bytes[1][n] = d->stride_bytes[in][i];
            s.stride_bytes[2][n] = moments ? d->stride_bytes[1][i] : 0;
            ++n;
        }
        if (n == 0) {
            s.shape[0] = 1;
            n = 1;
        }
        s.ndim = n;
        int idx[3] = {0, 1, 2};
        return OffsetCalc<3>::build(a.oc, &s, idx, 1);
    };

    if (p.path == PATH_INNER) {
        KF_REQUIRE(build_out_calc(0), KF_ERR_INVALID, "kf_reduce: bad shape/stride");
        if constexpr (!Ops::kPackRows) {
            if (p.vec > 1 && p.R <= 2 * p.vec && p.nsplit == 1 && p.nout >= 65536) { // many rows of one or two packs: a lane per row (four packs: 5.5 against 6.1 TB/s for the shuffle form)
                reduce_inner_few_kernel<T, Ops, 16 / sizeof(T)><<<(unsigned)((p.nout + 255) / 256), 256, 0, st>>>(a, fin);
                KF_LAUNCH_CHECK();
                return KF_OK;
            }
        }
        const int ty = kRB / p.tx;
        dim3 grid((unsigned)((p.nout + ty - 1) / ty), (unsigned)p.nsplit);
        if (p.vec > 1)
            reduce_inner_kernel<T, Ops, 16 / sizeof(T)><<<grid, kRB, 0, st>>>(a, fin);
        else
            reduce_inner_kernel<T, Ops, 1><<<grid, kRB, 0, st>>>(a, fin);
        KF_LAUNCH_CHECK();
        if (p.nsplit > 1) {
            reduce_fold_kernel<T, Ops><<<(unsigned)((p.nout + kRB - 1) / kRB), kRB, 0, st>>>(a, fin, 0);
            KF_LAUNCH_CHECK();
        }
    } else if (p.path == PATH_OUTER) {
        // out_dims[0] is dim 1 (the contiguous one) because dims are visited in order
        KF_REQUIRE(build_out_calc(1), KF_ERR_INVALID, "kf_reduce: bad shape/stride");
        const int64_t gx = (p.C + (int64_t)kWave * p.vec - 1) / ((int64_t)kWave * p.vec);
        if (p.tall) {
            dim3 gt((unsigned)((p.C + 15) / 16), 1, (unsigned)(p.nouter < 1024 ? p.nouter : 1024));
            reduce_outer_tall_kernel<T, Ops><<<gt, kRB, 0, st>>>(a, fin);
            KF_LAUNCH_CHECK();
            return KF_OK;
        }
        if constexpr (!Ops::kPackRows) { // (sums and means; the moments keep their pairwise update)
            if (p.R <= 8 && p.nsplit == 1) {
                const int64_t packs = (p.C + p.vec - 1) / p.vec;
                dim3 gf((unsigned)((packs + 255) / 256), 1, (unsigned)(p.nouter < 1024 ? p.nouter : 1024));
                if (p.vec > 1) reduce_outer_few_kernel<T, Ops, 16 / sizeof(T)><<<gf, 256, 0, st>>>(a, fin);
                else reduce_outer_few_kernel<T, Ops, 1><<<gf, 256, 0, st>>>(a, fin);
                KF_LAUNCH_CHECK();
                return KF_OK;
            }
        }
        dim3 grid((unsigned)gx, (unsigned)p.nsplit, (unsigned)(p.nouter < 1024 ? p.nouter : 1024));
        if (p.vec > 1)
            reduce_outer_kernel<T, Ops, 16 / sizeof(T)><<<grid, kRB, 0, st>>>(a, fin);
        else
            reduce_outer_kernel<T, Ops, 1><<<grid, kRB, 0, st>>>(a, fin);
        KF_LAUNCH_CHECK();
        if (p.nsplit > 1) {
            reduce_fold_kernel<T, Ops><<<(unsigned)((p.nout + kRB - 1) / kRB), kRB, 0, st>>>(a, fin, 1);
            KF_LAUNCH_CHECK();
        }
    } else {
        KF_REQUIRE(build_out_calc(0), KF_ERR_INVALID, "kf_reduce: bad shape/stride");
        kf_iter_desc s;
        memset(&s, 0, sizeof(s));
        s.ntensors = 1;
        int n = 0;
        for (int k = 0; k < p.nred; ++k) {
            s.shape[n] = d->shape[p.red_dims[k]];
            s.stride_bytes[0][n] = d->stride_bytes[in][p.red_dims[k]];
            ++n;
        }
        if (n == 0) {
            s.shape[0] = 1;
            n = 1;
        }
        s.ndim = n;
        int idx[1] = {0};
        KF_REQUIRE(OffsetCalc<1>::build(a.rc, &s, idx, 1), KF_ERR_INVALID, "kf_reduce: bad shape/stride");
        reduce_generic_kernel<T, Ops><<<(unsigned)((p.nout + kRB - 1) / kRB), kRB, 0, st>>>(a, fin);
        KF_LAUNCH_CHECK();
    }
    return KF_OK;
}

template <typename T>
static int run_sum(int op, const kf_iter_desc *d, const Plan &p, void *ws, hipStream_t st) {
    using A = typename RAcc<T>::type;
    typename SumOps<T>::Fin fin;
    fin.scale = op == KF_RED_MEAN;
    const int64_t numel = p.nout * p.rtot;
    // reference: factor = static_cast<acc_t>(nout) / numel with acc_t = scalar_t (reduce_ops_kernel.cu:49-53)
    if constexpr (std::is_same<A, int64_t>::value)
        fin.factor = numel ? p.nout / numel : 0;
    else
        fin.factor = (A)p.nout / (A)numel;
    return run_reduce<T, SumOps<T>>("kf_reduce", d, p, fin, ws, st);
}

template <typename T>
static int run_moments(int mode, const kf_iter_desc *d, const Plan &p, double correction, double eps, void *ws, hipStream_t st) {
    using X = typename RAcc<T>::type;
    typename MomentOps<T>::Fin fin;
    fin.correction = (X)correction;
    fin.eps = (X)eps;
    fin.mode = mode;
    fin.out_f32 = d->dtype[0] == KF_F32 && d->dtype[2] != KF_F32;
    return run_reduce<T, MomentOps<T>>("kf_reduce_moments", d, p, fin, ws, st);
}

} // namespace kf

using namespace kf;

extern "C" int kf_reduce_workspace_bytes(const kf_iter_desc *d, size_t *bytes) {
    KF_REQUIRE(d && bytes, KF_ERR_INVALID, "kf_reduce_workspace_bytes: null argument");
    Plan p;
    int rc = make_plan(d, p);
    if (rc != KF_OK) return rc;
    *bytes = p.ws_bytes;
    return KF_OK;
}

extern "C" int kf_reduce(int op, const kf_iter_desc *d, void *workspace, size_t workspace_bytes, void *stream) {
    KF_REQUIRE(d, KF_ERR_INVALID, "kf_reduce: null descriptor");
    KF_REQUIRE(op == KF_RED_SUM || op == KF_RED_MEAN, KF_ERR_INVALID, "kf_reduce: unknown op %d", op);
    Plan p;
    int rc = make_plan(d, p);
    if (rc != KF_OK) return rc;
    if (p.nout == 0) return KF_OK;
    KF_REQUIRE(d->data[0] && d->data[1], KF_ERR_INVALID, "kf_reduce: null data pointer");
    KF_REQUIRE(p.ws_bytes == 0 || (workspace && workspace_bytes >= p.ws_bytes), KF_ERR_WORKSPACE,
               "kf_reduce: workspace of %zu bytes required, got %zu", p.ws_bytes, workspace_bytes);
    hipStream_t st = as_stream(stream);
    switch (d->dtype[1]) {
    case KF_BOOL: return run_sum<bool>(op, d, p, workspace, st);
    case KF_U8: return run_sum<uint8_t>(op, d, p, workspace, st);
    case KF_I8: return run_sum<int8_t>(op, d, p, workspace, st);
    case KF_I16: return run_sum<int16_t>(op, d, p, workspace, st);
    case KF_I32: return run_sum<int32_t>(op, d, p, workspace, st);
    case KF_I64: return run_sum<int64_t>(op, d, p, workspace, st);
    case KF_F16: return run_sum<f16_t>(op, d, p, workspace, st);
    case KF_BF16: return run_sum<bf16_t>(op, d, p, workspace, st);
    case KF_F32: return run_sum<float>(op, d, p, workspace, st);
    case KF_F64: return run_sum<double>(op, d, p, workspace, st);
    default: KF_REQUIRE(false, KF_ERR_INVALID, "kf_reduce: bad dtype");
    }
    return KF_OK;
}

extern "C" int kf_reduce_moments_workspace_bytes(const kf_iter_desc *d, size_t *bytes) {
    KF_REQUIRE(d && bytes, KF_ERR_INVALID, "kf_reduce_moments_workspace_bytes: null argument");
    Plan p;
    int rc = make_plan(d, p, true);
    if (rc != KF_OK) return rc;
    *bytes = p.ws_bytes;
    return KF_OK;
}

extern "C" int kf_reduce_moments(int mode, const kf_iter_desc *d, double correction, double eps, void *workspace,
                                 size_t workspace_bytes, void *stream) {
    KF_REQUIRE(d, KF_ERR_INVALID, "kf_reduce_moments: null descriptor");
    KF_REQUIRE(mode == KF_MOM_VAR || mode == KF_MOM_STD || mode == KF_MOM_INVSTD, KF_ERR_INVALID, "kf_reduce_moments: unknown mode %d", mode);
    Plan p;
    int rc = make_plan(d, p, true);
    if (rc != KF_OK) return rc;
    if (p.nout == 0) return KF_OK;
    KF_REQUIRE(d->data[0] && d->data[1] && d->data[2], KF_ERR_INVALID, "kf_reduce_moments: null data pointer");
    KF_REQUIRE(p.ws_bytes == 0 || (workspace && workspace_bytes >= p.ws_bytes), KF_ERR_WORKSPACE,
               "kf_reduce_moments: workspace of %zu bytes required, got %zu", p.ws_bytes, workspace_bytes);
    hipStream_t st = as_stream(stream);
    switch (d->dtype[2]) {
    case KF_F16: return run_moments<f16_t>(mode, d, p, correction, eps, workspace, st);
    case KF_BF16: return run_moments<bf16_t>(mode, d, p, correction, eps, workspace, st);
    case KF_F32: return run_moments<float>(mode, d, p, correction, eps, workspace, st);
    case KF_F64: return run_moments<double>(mode, d, p, correction, eps, workspace, st);
    default: KF_REQUIRE(false, KF_ERR_UNSUPPORTED, "kf_reduce_moments: bad dtype");
    }
    return KF_OK;
}
