// Elementwise loops for gfx950: add/sub/mul/div, copy/convert, fill.
//
// Replaces the reference's loop engine (src/device/utils/tensor_loops.h:16-369 + the functors in
// binary/unary/nullary_ops_kernel.cu). Design differences, MI355X-first:
//   * every same-dtype path moves 16 B per lane (bf16/half included — the reference runs every
//     half/bf16/int binary op through a scalar cast loop, binary_ops_kernel.cu:34-39);
//   * ONE 16-byte pack per lane and a grid as large as the problem (many short waves stream faster on this
//     memory system than few long ones: 6.2 TB/s against 4.8-5.4 for a capped grid with 4 packs in flight);
//   * broadcast operands with a contiguous (or stride-0) inner dimension stay on the 16-B path;
//   * mixed dtypes use one runtime-cast kernel per accumulate class instead of a per-functor
//     template zoo.
// Arithmetic follows the reference exactly: operands are cast to the accumulate type of the common
// dtype (float for half/bf16/float, double, int64 for integers, bool), combined, and cast to the
// output dtype on store (accumulate_type.h:17-27, tensor_memory_access.h:13-37).
#include <stdlib.h>

#include "common.h"
#include "offset_calc.h"

namespace kf {

// ------------------------------------------------------------------------------------------
// scalar conversions (static_cast semantics of the reference's fetch_and_cast/cast_and_store)
// ------------------------------------------------------------------------------------------
template <typename A>
__device__ __forceinline__ A load_as(int dt, const char *p) {
    switch (dt) {
    case KF_BOOL: return (A)(*(const uint8_t *)p != 0);
    case KF_U8: return (A)(*(const uint8_t *)p);
    case KF_I8: return (A)(*(const int8_t *)p);
    case KF_I16: return (A)(*(const int16_t *)p);
    case KF_I32: return (A)(*(const int32_t *)p);
    case KF_I64: return (A)(*(const int64_t *)p);
    case KF_F16: return (A)f16_to_f32(*(const f16_t *)p);
    case KF_BF16: return (A)bf16_to_f32(*(const bf16_t *)p);
    case KF_F32: return (A)(*(const float *)p);
    case KF_F64: return (A)(*(const double *)p);
    default: return A(0);
    }
}

template <typename A>
__device__ __forceinline__ void store_from(int dt, char *p, A v) {
    switch (dt) {
    case KF_BOOL: *(uint8_t *)p = (uint8_t)(v != A(0)); break;
    case KF_U8: *(uint8_t *)p = (uint8_t)v; break;
    case KF_I8: *(int8_t *)p = (int8_t)v; break;
    case KF_I16: *(int16_t *)p = (int16_t)v; break;
    case KF_I32: *(int32_t *)p = (int32_t)v; break;
    case KF_I64: *(int64_t *)p = (int64_t)v; break;
    case KF_F16: *(f16_t *)p = f32_to_f16((float)v); break;
    case KF_BF16: *(bf16_t *)p = f32_to_bf16((float)v); break;
    case KF_F32: *(float *)p = (float)v; break;
    case KF_F64: *(double *)p = (double)v; break;
    default: break;
    }
}

template <typename A>
__device__ __forceinline__ A apply_op(int op, A a, A b) {
    switch (op) {
    case KF_EW_ADD: return a + b;
    case KF_EW_SUB: return a - b;
    case KF_EW_MUL: return a * b;
    default: return a / b;
    }
}
template <>
__device__ __forceinline__ int64_t apply_op<int64_t>(int op, int64_t a, int64_t b) {
    switch (op) {
    case KF_EW_ADD: return (int64_t)((uint64_t)a + (uint64_t)b);
    case KF_EW_SUB: return (int64_t)((uint64_t)a - (uint64_t)b);
    case KF_EW_MUL: return (int64_t)((uint64_t)a * (uint64_t)b);
    default: return b == 0 ? 0 : (b == -1 ? (int64_t)(0 - (uint64_t)a) : a / b); // x/0 is UB in the reference; 0 here
    }
}
template <>
__device__ __forceinline__ bool apply_op<bool>(int op, bool a, bool b) {
    // bool arithmetic promotes to int and converts back (C++), as the reference's Functor<bool> does
    switch (op) {
    case KF_EW_ADD: return a || b;
    case KF_EW_SUB: return a != b;
    case KF_EW_MUL: return a && b;
    default: return a; // a / true; a / false is UB in the reference
    }
}

// storage type <-> accumulate type for the same-dtype vector paths
template <typename T> struct Acc { using type = T; };
template <> struct Acc<bf16_t> { using type = float; };
template <> struct Acc<f16_t> { using type = float; };
template <> struct Acc<int32_t> { using type = int64_t; };

template <typename T> __device__ __forceinline__ typename Acc<T>::type to_acc(T v) { return (typename Acc<T>::type)v; }
template <> __device__ __forceinline__ float to_acc<bf16_t>(bf16_t v) { return bf16_to_f32(v); }
template <> __device__ __forceinline__ float to_acc<f16_t>(f16_t v) { return f16_to_f32(v); }
template <typename T> __device__ __forceinline__ T from_acc(typename Acc<T>::type v) { return (T)v; }
template <> __device__ __forceinline__ bf16_t from_acc<bf16_t>(float v) { return f32_to_bf16(v); }
template <> __device__ __forceinline__ f16_t from_acc<f16_t>(float v) { return f32_to_f16(v); }

// VEC elements moved as one access. Only the ELEMENT's alignment is promised: the global path takes 16-byte accesses at any dword- or
// halfword-aligned address (unaligned access mode; the compiler still emits dwordx4), so a row slice that starts at an odd element keeps the wide form
// (round 5: bf16 x[:, 1:4097] + y[:, 3:4099] ran element by element at 1.9 TB/s).
template <typename T, int VEC>
struct __attribute__((packed, aligned(sizeof(T)))) Pack {
    T v[VEC];
};

constexpr int kBlock = 256;
constexpr int kUnroll = 1; // ONE 16-byte pack per lane and a grid as large as the problem: measured 6.2 TB/s on a float4 copy against
                           // 4.8-5.4 for 4 packs in flight per lane on a capped grid and 3.6-3.8 for 8 (tools/scratch/copybw.hip)

// ------------------------------------------------------------------------------------------
// same-dtype kernel: NIN inputs of type T, one output of type T. MODE: 0 = arithmetic (op at
// run time), 1 = copy (raw bits), 2 = fill (raw bits), 3 = arithmetic with a scalar right operand
// (the value the reference would have filled a whole tensor with, register.cpp:172-206).
// ------------------------------------------------------------------------------------------
template <int NT>
struct SameArgs {
    char *ptr[NT];
    int64_t nvec;        // number of VEC-wide items
    int op;
    uint32_t bcast0;     // bit t set: operand t has stride 0 along dim 0 (vector path: splat)
    uint64_t fill_bits;  // MODE 2 and 3: the scalar, already rounded to T
    OffsetCalc<NT> oc;   // !CONTIG only
};

template <typename T, int VEC, int NIN, int MODE, bool CONTIG>
__global__ __launch_bounds__(kBlock) void ew_same_kernel(const SameArgs<NIN + 1> args) {
    constexpr int NT = NIN + 1;
    using P = Pack<T, VEC>;
    using A = typename Acc<T>::type;
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;

    P fillv;
    if constexpr (MODE == 2 || MODE == 3) {
        T one;
        __builtin_memcpy(&one, &args.fill_bits, sizeof(T));
#pragma unroll
        for (int e = 0; e < VEC; ++e) fillv.v[e] = one;
    }

    for (; i < args.nvec; i += stride * kUnroll) {
        P in[kUnroll][NIN > 0 ? NIN : 1];
        uint32_t ooff[kUnroll];
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t idx = i + u * stride;
            if (idx < args.nvec) {
                if constexpr (CONTIG) {
#pragma unroll
                    for (int t = 0; t < NIN; ++t) in[u][t] = *(const P *)(args.ptr[t + 1] + idx * (int64_t)sizeof(P));
                } else {
                    uint32_t off[NT];
                    args.oc.get((uint32_t)idx, off);
                    ooff[u] = off[0];
#pragma unroll
                    for (int t = 0; t < NIN; ++t) {
                        if (VEC > 1 && ((args.bcast0 >> (t + 1)) & 1u)) {
                            T s = *(const T *)(args.ptr[t + 1] + off[t + 1]);
#pragma unroll
                            for (int e = 0; e < VEC; ++e) in[u][t].v[e] = s;
                        } else {
                            in[u][t] = *(const P *)(args.ptr[t + 1] + off[t + 1]);
                        }
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) {
            const int64_t idx = i + u * stride;
            if (idx < args.nvec) {
                P out;
                if constexpr (MODE == 2) {
                    out = fillv;
                } else if constexpr (MODE == 1) {
                    out = in[u][0];
                } else if constexpr (MODE == 3) {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        out.v[e] = from_acc<T>(apply_op<A>(args.op, to_acc<T>(in[u][0].v[e]), to_acc<T>(fillv.v[e])));
                } else {
#pragma unroll
                    for (int e = 0; e < VEC; ++e)
                        out.v[e] = from_acc<T>(apply_op<A>(args.op, to_acc<T>(in[u][0].v[e]), to_acc<T>(in[u][NIN > 1 ? 1 : 0].v[e])));
                }
                if constexpr (CONTIG)
                    *(P *)(args.ptr[0] + idx * (int64_t)sizeof(P)) = out;
                else
                    *(P *)(args.ptr[0] + ooff[u]) = out;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------
// runtime-cast kernel: any dtype per operand, accumulate class A, strided (32-bit indexable).
// MODE 0 arithmetic, 1 copy/convert.
// ------------------------------------------------------------------------------------------
template <int NT>
struct CastArgs {
    char *ptr[NT];
    int dtype[NT];
    uint32_t n;
    int op;
    OffsetCalc<NT> oc;
};

template <typename A, int NIN, int MODE>
__global__ __launch_bounds__(kBlock) void ew_cast_kernel(const CastArgs<NIN + 1> args) {
    constexpr int NT = NIN + 1;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < args.n; i += stride) {
        uint32_t off[NT];
        args.oc.get(i, off);
        A a = load_as<A>(args.dtype[1], args.ptr[1] + off[1]);
        A r;
        if constexpr (MODE == 1) {
            r = a;
        } else {
            A b = load_as<A>(args.dtype[NT - 1], args.ptr[NT - 1] + off[NT - 1]);
            r = apply_op<A>(args.op, a, b);
        }
        store_from<A>(args.dtype[0], args.ptr[0] + off[0], r);
    }
}

// The same arithmetic on CONTIGUOUS operands (round 5): eight consecutive elements per lane, moved with 8- / 16-byte accesses whatever the
// element width (1-byte types 8 B, 2-byte 16 B, 4-byte 2 x 16 B, 8-byte 4 x 16 B), converted one by one with the conversions of load_as /
// store_from - bit-identical to the strided kernel. f32 + bf16 -> f32 at 128 Mi elements: 0.44 -> see DESIGN §4 (the one-element-per-lane
// form with its offset calculator ran at 2.6 - 3.6 TB/s).
template <typename T> __device__ __forceinline__ void load_raw8(const char *p, T (&t)[8]) {
    constexpr int B = 8 * (int)sizeof(T);
    if constexpr (B == 8) {
        const uint2 r = *(const uint2 *)p;
        __builtin_memcpy(t, &r, 8);
    } else {
        uint4 r[B / 16];
#pragma unroll
        for (int i = 0; i < B / 16; ++i) r[i] = ((const uint4 *)p)[i];
        __builtin_memcpy(t, r, B);
    }
}
template <typename T> __device__ __forceinline__ void store_raw8(char *p, const T (&t)[8]) {
    constexpr int B = 8 * (int)sizeof(T);
    if constexpr (B == 8) {
        uint2 r;
        __builtin_memcpy(&r, t, 8);
        *(uint2 *)p = r;
    } else {
        uint4 r[B / 16];
        __builtin_memcpy(r, t, B);
#pragma unroll
        for (int i = 0; i < B / 16; ++i) ((uint4 *)p)[i] = r[i];
    }
}
template <typename A>
__device__ __forceinline__ void load8_as(int dt, const char *base, size_t group, A (&v)[8]) {
#define KF_L8(T_, EXPR_)                                    \
    {                                                       \
        T_ t[8];                                            \
        load_raw8<T_>(base + group * (8 * sizeof(T_)), t);  \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) { const T_ x = t[e]; v[e] = EXPR_; } \
        return;                                             \
    }
    switch (dt) {
    case KF_BOOL: KF_L8(uint8_t, (A)(x != 0))
    case KF_U8: KF_L8(uint8_t, (A)x)
    case KF_I8: KF_L8(int8_t, (A)x)
    case KF_I16: KF_L8(int16_t, (A)x)
    case KF_I32: KF_L8(int32_t, (A)x)
    case KF_I64: KF_L8(int64_t, (A)x)
    case KF_F16: KF_L8(f16_t, (A)f16_to_f32(x))
    case KF_BF16: KF_L8(bf16_t, (A)bf16_to_f32(x))
    case KF_F32: KF_L8(float, (A)x)
    case KF_F64: KF_L8(double, (A)x)
    default: break;
    }
#undef KF_L8
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = A(0);
}
template <typename A>
__device__ __forceinline__ void store8_from(int dt, char *base, size_t group, const A (&v)[8]) {
#define KF_S8(T_, EXPR_)                                    \
    {                                                       \
        T_ t[8];                                            \
        _Pragma("unroll") for (int e = 0; e < 8; ++e) { const A x = v[e]; t[e] = EXPR_; } \
        store_raw8<T_>(base + group * (8 * sizeof(T_)), t); \
        return;                                             \
    }
    switch (dt) {
    case KF_BOOL: KF_S8(uint8_t, (uint8_t)(x != A(0)))
    case KF_U8: KF_S8(uint8_t, (uint8_t)x)
    case KF_I8: KF_S8(int8_t, (int8_t)x)
    case KF_I16: KF_S8(int16_t, (int16_t)x)
    case KF_I32: KF_S8(int32_t, (int32_t)x)
    case KF_I64: KF_S8(int64_t, (int64_t)x)
    case KF_F16: KF_S8(f16_t, f32_to_f16((float)x))
    case KF_BF16: KF_S8(bf16_t, f32_to_bf16((float)x))
    case KF_F32: KF_S8(float, (float)x)
    case KF_F64: KF_S8(double, (double)x)
    default: break;
    }
#undef KF_S8
}

template <typename A, int NIN, int MODE>
__global__ __launch_bounds__(kBlock) void ew_cast8_kernel(const CastArgs<NIN + 1> args) { // args.n = groups of eight elements
    constexpr int NT = NIN + 1;
    const uint32_t stride = gridDim.x * kBlock;
    for (uint32_t i = blockIdx.x * kBlock + threadIdx.x; i < args.n; i += stride) {
        A a[8], r[8];
        load8_as<A>(args.dtype[1], args.ptr[1], i, a);
        if constexpr (MODE == 1) {
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = a[e];
        } else {
            A b[8];
            load8_as<A>(args.dtype[NT - 1], args.ptr[NT - 1], i, b);
#pragma unroll
            for (int e = 0; e < 8; ++e) r[e] = apply_op<A>(args.op, a[e], b[e]);
        }
        store8_from<A>(args.dtype[0], args.ptr[0], i, r);
    }
}

// ------------------------------------------------------------------------------------------
// tiled transpose copy: the output walks memory along dim 0, the input along some other dim `j`
// (permute(...).contiguous() of a matrix-like view). A 64 x 64 element tile goes through LDS so that
// BOTH sides move whole coalesced rows: read with lanes along the input-contiguous dim, write with
// lanes along the output-contiguous dim. Remaining dims are a batch walked by the offset calculator.
// ------------------------------------------------------------------------------------------
struct TransArgs {
    const char *in;
    char *out;
    uint32_t n0, n1;            // extent of dim 0 (output-contiguous) and of dim j (input-contiguous)
    uint32_t in_s0, out_s1;     // byte stride of the input along dim 0 / of the output along dim j
    uint32_t tiles0, tiles1, nbatch;
    OffsetCalc<2> bc;           // batch dims: [0] = out bytes, [1] = in bytes
};

template <typename U>
__global__ __launch_bounds__(kBlock) void ew_transpose_kernel(const TransArgs a) {
    constexpr int TD = 64;
    __shared__ U tile[TD][TD + (sizeof(U) >= 4 ? 1 : 2)];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    for (uint32_t blk = blockIdx.x; blk < a.tiles0 * a.tiles1 * a.nbatch; blk += gridDim.x) {
        const uint32_t bt = blk / (a.tiles0 * a.tiles1), rem = blk - bt * (a.tiles0 * a.tiles1);
        const uint32_t t1 = rem / a.tiles0, t0 = rem - t1 * a.tiles0;
        uint32_t boff[2];
        a.bc.get(bt, boff);
        const uint32_t i0 = t0 * TD, j0 = t1 * TD;
        __syncthreads();
        for (int r = ty; r < TD; r += 4) { // row r of the tile = index along dim 0; lanes along dim j
            const uint32_t i = i0 + r, j = j0 + tx;
            if (i < a.n0 && j < a.n1) tile[r][tx] = *(const U *)(a.in + boff[1] + (size_t)i * a.in_s0 + (size_t)j * sizeof(U));
        }
        __syncthreads();
        for (int r = ty; r < TD; r += 4) { // row r = index along dim j; lanes along dim 0
            const uint32_t j = j0 + r, i = i0 + tx;
            if (i < a.n0 && j < a.n1) *(U *)(a.out + boff[0] + (size_t)j * a.out_s1 + (size_t)i * sizeof(U)) = tile[tx][r];
        }
    }
}

// Block -> tile of the tiled transposes. Blocks go to the eight XCDs round-robin. SD x SD consecutive blocks of ONE XCD take the tiles of a super-tile whose
// rows are 1 KiB on both sides (4 x 4 tiles of 4-byte elements, 8 x 8 of 2-byte ones): what an XCD's L2 sees of both matrices within a short time is whole
// 1-KiB row pieces instead of lone 256- / 128-byte ones at the matrices' row pitch. Round 5, 16384^2: f32 0.42-0.46 -> 0.38-0.40 ms, bf16 0.31 -> 0.21 (4 x 4: 0.24).
template <int ES>
__device__ __forceinline__ void trans_tile_of(uint32_t tiles0, uint32_t tiles1, uint32_t rem, uint32_t &t0, uint32_t &t1) {
    t1 = rem / tiles0;
    t0 = rem - t1 * tiles0;
    constexpr uint32_t SL = ES == 4 ? 2 : 3, SD = 1u << SL, SN = SD * SD;
    if ((tiles0 & (SD - 1)) == 0 && (tiles1 & (SD - 1)) == 0 && ((tiles0 * tiles1) & (8 * SN - 1)) == 0) {
        const uint32_t x = rem & 7, q = rem >> 3, super = (q / SN) * 8 + x, in = q & (SN - 1), s0 = tiles0 >> SL;
        t0 = (super % s0) * SD + (in & (SD - 1));
        t1 = (super / s0) * SD + (in >> SL);
    }
}

// The same tile with 16-byte global accesses on BOTH sides (4- and 2-byte elements, whole tiles, 16-byte aligned rows): a lane
// loads one pack of V = 16 / sizeof(U) consecutive elements of an input row, scatters it into the tile with V LDS stores,
// then gathers V elements of one tile COLUMN (V LDS loads) into the pack it stores to the output row. One tile per block.
// Row stride TD + 1 (4-byte elements: a lane group's V-row steps land on distinct banks) or TD + 2 (2-byte).
template <typename U>
__global__ __launch_bounds__(kBlock) void ew_transpose_vec_kernel(const TransArgs a) {
    constexpr int TD = 64, V = 16 / (int)sizeof(U), G = TD / V; // packs per tile row
    __shared__ U tile[TD][TD + (sizeof(U) >= 4 ? 1 : 2)];
    const uint32_t blk = blockIdx.x;
    const uint32_t bt = blk / (a.tiles0 * a.tiles1), rem = blk - bt * (a.tiles0 * a.tiles1);
    uint32_t t0, t1;
    trans_tile_of<sizeof(U)>(a.tiles0, a.tiles1, rem, t0, t1);
    uint32_t boff[2];
    a.bc.get(bt, boff);
    const uint32_t i0 = t0 * TD, j0 = t1 * TD;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < TD * G / kBlock; ++k) { // input rows (dim 0), packs along dim j
        const int id = t + k * kBlock, r = id / G, g = id % G;
        const Pack<U, V> p = *(const Pack<U, V> *)(a.in + boff[1] + (size_t)(i0 + r) * a.in_s0 + (size_t)(j0 + g * V) * sizeof(U));
#pragma unroll
        for (int e = 0; e < V; ++e) tile[r][g * V + e] = p.v[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TD * G / kBlock; ++k) { // output rows (dim j), packs along dim 0
        const int id = t + k * kBlock, r = id / G, g = id % G;
        Pack<U, V> p;
#pragma unroll
        for (int e = 0; e < V; ++e) p.v[e] = tile[g * V + e][r];
        *(Pack<U, V> *)(a.out + boff[0] + (size_t)(j0 + r) * a.out_s1 + (size_t)(i0 + g * V) * sizeof(U)) = p;
    }
}

// out = op(x, y) where ONE operand is read along a dim other than the output's contiguous one (x + y.permute(1, 0)): the transposed operand
// goes through the LDS tile exactly like the copy above, the straight one is read in the store phase, 16 bytes per lane on every side.
// (Round 5: the strided kernel read that operand 4 bytes per lane at the row pitch: f32 [4096, 8192] 1.7 TB/s.)
struct TransBinArgs {
    const char *tin, *sin; // the transposed operand, the straight one
    char *out;
    uint32_t n0, n1;
    uint32_t tin_s0, out_s1, sin_s1; // byte strides: transposed operand along dim 0; output and straight operand along dim j
    uint32_t tiles0, tiles1, nbatch;
    int op, t_first;                 // t_first: the transposed operand is the FIRST input of the operator
    OffsetCalc<3> bc;                // batch dims: [0] = out bytes, [1] = transposed operand, [2] = straight operand
};
template <typename T>
__global__ __launch_bounds__(kBlock) void ew_transpose_binary_kernel(const TransBinArgs a) {
    using A = typename Acc<T>::type;
    constexpr int TD = 64, V = 16 / (int)sizeof(T), G = TD / V;
    __shared__ T tile[TD][TD + (sizeof(T) >= 4 ? 1 : 2)];
    const uint32_t blk = blockIdx.x;
    const uint32_t bt = blk / (a.tiles0 * a.tiles1), rem = blk - bt * (a.tiles0 * a.tiles1);
    uint32_t t0, t1;
    trans_tile_of<sizeof(T)>(a.tiles0, a.tiles1, rem, t0, t1);
    uint32_t boff[3];
    a.bc.get(bt, boff);
    const uint32_t i0 = t0 * TD, j0 = t1 * TD;
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < TD * G / kBlock; ++k) {
        const int id = t + k * kBlock, r = id / G, g = id % G;
        const Pack<T, V> p = *(const Pack<T, V> *)(a.tin + boff[1] + (size_t)(i0 + r) * a.tin_s0 + (size_t)(j0 + g * V) * sizeof(T));
#pragma unroll
        for (int e = 0; e < V; ++e) tile[r][g * V + e] = p.v[e];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TD * G / kBlock; ++k) {
        const int id = t + k * kBlock, r = id / G, g = id % G;
        const Pack<T, V> sp = *(const Pack<T, V> *)(a.sin + boff[2] + (size_t)(j0 + r) * a.sin_s1 + (size_t)(i0 + g * V) * sizeof(T));
        Pack<T, V> o;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const A x = to_acc<T>(tile[g * V + e][r]), y = to_acc<T>(sp.v[e]);
            o.v[e] = from_acc<T>(a.t_first ? apply_op<A>(a.op, x, y) : apply_op<A>(a.op, y, x));
        }
        *(Pack<T, V> *)(a.out + boff[0] + (size_t)(j0 + r) * a.out_s1 + (size_t)(i0 + g * V) * sizeof(T)) = o;
    }
}

// ------------------------------------------------------------------------------------------
// contiguous float-family conversion (f32 <-> bf16 <-> f16: Tensor::half()/bfloat16()/float()),
// 8 elements per lane: 16-B accesses on the 16-bit side, 2 x 16 B on the f32 side
// ------------------------------------------------------------------------------------------
template <typename S, typename D>
__global__ __launch_bounds__(kBlock) void ew_convert8_kernel(const S *in, D *out, int64_t n8) {
    const int64_t stride = (int64_t)gridDim.x * kBlock;
    for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n8; i += stride) {
        const Pack<S, 8> v = *(const Pack<S, 8> *)(in + i * 8);
        Pack<D, 8> o;
#pragma unroll
        for (int e = 0; e < 8; ++e) o.v[e] = from_acc<D>(to_acc<S>(v.v[e]));
        *(Pack<D, 8> *)(out + i * 8) = o;
    }
}

// ------------------------------------------------------------------------------------------
// host-side dispatch
// ------------------------------------------------------------------------------------------
static inline int grid_for(int64_t nitems) {
    int64_t blocks = (nitems + kBlock - 1) / kBlock;
    const int64_t cap = 0x7fffffff; // no cap: many short waves beat few long ones on this memory system (see kUnroll)
    if (blocks > cap) blocks = cap;
    if (blocks < 1) blocks = 1;
    return (int)blocks;
}

static bool desc_contiguous(const kf_iter_desc *d) {
    // reference TensorIterator::is_contiguous (tensor_iterator.cpp:407-415)
    if (desc_numel(d) == 1) return true;
    if (d->ndim != 1) return false;
    for (int t = 0; t < d->ntensors; ++t)
        if (d->stride_bytes[t][0] != dtype_size(d->dtype[t])) return false;
    return true;
}

// Largest VEC (elements) so that every operand can be walked in 16-byte (or smaller pow2) packs
// along dim 0: dim 0 contiguous or stride 0 (inputs only), sizes/strides/pointers aligned.
static int pick_vec(const kf_iter_desc *d, int esize) {
    int vec = 16 / esize;
    for (; vec > 1; vec >>= 1) {
        const int64_t vb = (int64_t)vec * esize;
        bool ok = d->shape[0] % vec == 0;
        for (int t = 0; ok && t < d->ntensors; ++t) {
            const int64_t s0 = d->stride_bytes[t][0];
            if (!(s0 == esize || (s0 == 0 && t >= d->noutputs))) ok = false;
            if ((uintptr_t)d->data[t] % esize) ok = false; // (element alignment only: see Pack)
            // KF_EW_ALIGNED_ONLY: the pre-round-5 dispatch - a pack is taken only at pack alignment (A/B switch and escape hatch: the
            // relaxed form leans on gfx950's unaligned global access mode; tests/test_kernel_hazards.py pins that the relaxed
            // instantiations still compile to 16-byte accesses)
            if (knob(KNOB_EW_ALIGNED_ONLY) && ((uintptr_t)d->data[t] % vb)) ok = false;
            for (int k = 1; ok && knob(KNOB_EW_ALIGNED_ONLY) && k < d->ndim; ++k)
                if (d->stride_bytes[t][k] % vb) ok = false;
        }
        if (ok) return vec;
    }
    return 1;
}

template <typename T, int NIN, int MODE>
static int launch_same(const kf_iter_desc *d, int op, uint64_t fill_bits, hipStream_t st) {
    constexpr int NT = NIN + 1;
    constexpr int VMAX = 16 / sizeof(T);
    KF_PROF(MODE == 0 ? "ew_arith" : MODE == 1 ? "ew_copy" : MODE == 2 ? "ew_fill" : "ew_arith_scalar", st);
    SameArgs<NT> a;
    memset(&a, 0, sizeof(a));
    for (int t = 0; t < NT; ++t) a.ptr[t] = (char *)d->data[t];
    a.op = op;
    a.fill_bits = fill_bits;
    const int64_t numel = desc_numel(d);
    int opidx[NT];
    for (int t = 0; t < NT; ++t) opidx[t] = t;

    if (desc_contiguous(d)) {
        bool aligned = numel % VMAX == 0;
        for (int t = 0; t < NT; ++t)
            if ((uintptr_t)d->data[t] % sizeof(T)) aligned = false;
        if (aligned && VMAX > 1) {
            a.nvec = numel / VMAX;
            ew_same_kernel<T, VMAX, NIN, MODE, true><<<grid_for(a.nvec), kBlock, 0, st>>>(a);
        } else {
            a.nvec = numel;
            ew_same_kernel<T, 1, NIN, MODE, true><<<grid_for(a.nvec), kBlock, 0, st>>>(a);
        }
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    KF_REQUIRE(desc_is_32bit(d), KF_ERR_INDEX_RANGE, "kf_elementwise: strided descriptor is not 32-bit indexable");
    if constexpr (MODE == 1) { // permute(...).contiguous(): a transposed walk goes through the LDS-tiled kernel
        const int64_t es = sizeof(T);
        int j = -1;
        if (d->ndim >= 2 && d->stride_bytes[0][0] == es && d->stride_bytes[1][0] != es && d->shape[0] >= 16)
            for (int i = 1; i < d->ndim; ++i)
                if (d->stride_bytes[1][i] == es && d->shape[i] >= 16) { j = i; break; }
        if (j > 0) {
            TransArgs t;
            memset(&t, 0, sizeof(t));
            t.in = (const char *)d->data[1];
            t.out = (char *)d->data[0];
            t.n0 = (uint32_t)d->shape[0];
            t.n1 = (uint32_t)d->shape[j];
            t.in_s0 = (uint32_t)d->stride_bytes[1][0];
            t.out_s1 = (uint32_t)d->stride_bytes[0][j];
            t.tiles0 = (t.n0 + 63) / 64;
            t.tiles1 = (t.n1 + 63) / 64;
            kf_iter_desc bd; // the remaining dims
            memset(&bd, 0, sizeof(bd));
            bd.ntensors = 2;
            bd.noutputs = 1;
            int nb = 0;
            int64_t nbatch = 1;
            for (int i = 1; i < d->ndim; ++i) {
                if (i == j) continue;
                bd.shape[nb] = d->shape[i];
                bd.stride_bytes[0][nb] = d->stride_bytes[0][i];
                bd.stride_bytes[1][nb] = d->stride_bytes[1][i];
                nbatch *= d->shape[i];
                ++nb;
            }
            if (nb == 0) { bd.shape[0] = 1; nb = 1; }
            bd.ndim = nb;
            int two[2] = {0, 1};
            const int64_t total = (int64_t)t.tiles0 * t.tiles1 * nbatch;
            if (OffsetCalc<2>::build(t.bc, &bd, two, 1) && total < 0x7fffffffLL) {
                t.nbatch = (uint32_t)nbatch;
                // whole tiles, 16-byte aligned rows on both sides and every batch offset: the 16-byte-per-lane form
                bool vec_ok = (sizeof(T) == 4 || sizeof(T) == 2) && t.n0 % 64 == 0 && t.n1 % 64 == 0 && t.in_s0 % 16 == 0 && t.out_s1 % 16 == 0 &&
                              (uintptr_t)t.in % 16 == 0 && (uintptr_t)t.out % 16 == 0;
                for (int i = 0; vec_ok && i < nb; ++i)
                    if (bd.stride_bytes[0][i] % 16 || bd.stride_bytes[1][i] % 16) vec_ok = false;
                if (vec_ok) {
                    if constexpr (sizeof(T) == 4 || sizeof(T) == 2) ew_transpose_vec_kernel<T><<<(unsigned)total, kBlock, 0, st>>>(t);
                } else {
                    const int grid = (int)(total < 256 * 16 ? total : 256 * 16);
                    ew_transpose_kernel<T><<<grid, kBlock, 0, st>>>(t);
                }
                KF_LAUNCH_CHECK();
                return KF_OK;
            }
        }
    }
    if constexpr (MODE == 0 && NIN == 2 && (sizeof(T) == 4 || sizeof(T) == 2)) { // one operand transposed against the output: the LDS-tiled binary kernel
        const int64_t es = sizeof(T);
        int tt = 0, j = -1;
        for (int t = 1; t <= 2 && d->ndim >= 2 && d->stride_bytes[0][0] == es; ++t)
            if (d->stride_bytes[t][0] != es && d->stride_bytes[3 - t][0] == es && d->shape[0] % 64 == 0)
                for (int i = 1; i < d->ndim && j < 0; ++i)
                    if (d->stride_bytes[t][i] == es && d->shape[i] % 64 == 0) { tt = t; j = i; }
        if (j > 0 && d->data[tt] != d->data[0]) { // (an output that IS the transposed operand keeps the element-by-element kernel)
            const int ss = 3 - tt;
            TransBinArgs t;
            memset(&t, 0, sizeof(t));
            t.tin = (const char *)d->data[tt];
            t.sin = (const char *)d->data[ss];
            t.out = (char *)d->data[0];
            t.n0 = (uint32_t)d->shape[0];
            t.n1 = (uint32_t)d->shape[j];
            t.tin_s0 = (uint32_t)d->stride_bytes[tt][0];
            t.out_s1 = (uint32_t)d->stride_bytes[0][j];
            t.sin_s1 = (uint32_t)d->stride_bytes[ss][j];
            t.tiles0 = t.n0 / 64;
            t.tiles1 = t.n1 / 64;
            t.op = op;
            t.t_first = tt == 1;
            kf_iter_desc bd;
            memset(&bd, 0, sizeof(bd));
            bd.ntensors = 3;
            bd.noutputs = 1;
            int nb = 0;
            int64_t nbatch = 1;
            bool ok = d->stride_bytes[tt][0] % 16 == 0 && d->stride_bytes[tt][0] > 0 && d->stride_bytes[0][j] % 16 == 0 && d->stride_bytes[ss][j] % 16 == 0 &&
                      d->stride_bytes[0][j] > 0 && d->stride_bytes[ss][j] > 0 && (uintptr_t)t.tin % 16 == 0 && (uintptr_t)t.sin % 16 == 0 && (uintptr_t)t.out % 16 == 0;
            for (int i = 1; i < d->ndim; ++i) {
                if (i == j) continue;
                bd.shape[nb] = d->shape[i];
                bd.stride_bytes[0][nb] = d->stride_bytes[0][i];
                bd.stride_bytes[1][nb] = d->stride_bytes[tt][i];
                bd.stride_bytes[2][nb] = d->stride_bytes[ss][i];
                if (d->stride_bytes[0][i] % 16 || d->stride_bytes[tt][i] % 16 || d->stride_bytes[ss][i] % 16) ok = false;
                nbatch *= d->shape[i];
                ++nb;
            }
            if (nb == 0) { bd.shape[0] = 1; nb = 1; }
            bd.ndim = nb;
            int three[3] = {0, 1, 2};
            const int64_t total = (int64_t)t.tiles0 * t.tiles1 * nbatch;
            if (ok && total < 0x7fffffffLL && total >= 64 && OffsetCalc<3>::build(t.bc, &bd, three, 1)) {
                t.nbatch = (uint32_t)nbatch;
                ew_transpose_binary_kernel<T><<<(unsigned)total, kBlock, 0, st>>>(t);
                KF_LAUNCH_CHECK();
                return KF_OK;
            }
        }
    }
    int vec = pick_vec(d, sizeof(T));
    if (vec != VMAX) vec = 1; // two instantiations only: full 16-B packs or scalar
    for (int t = 0; t < NT; ++t)
        if (d->stride_bytes[t][0] == 0 && t >= d->noutputs) a.bcast0 |= 1u << t;
    KF_REQUIRE(OffsetCalc<NT>::build(a.oc, d, opidx, vec), KF_ERR_INVALID, "kf_elementwise: bad shape/stride");
    a.nvec = numel / vec;
    if (vec == VMAX && VMAX > 1)
        ew_same_kernel<T, VMAX, NIN, MODE, false><<<grid_for(a.nvec), kBlock, 0, st>>>(a);
    else
        ew_same_kernel<T, 1, NIN, MODE, false><<<grid_for(a.nvec), kBlock, 0, st>>>(a);
    KF_LAUNCH_CHECK();
    return KF_OK;
}

template <typename A, int NIN, int MODE>
static int launch_cast(const kf_iter_desc *d, int op, hipStream_t st) {
    constexpr int NT = NIN + 1;
    KF_REQUIRE(desc_is_32bit(d), KF_ERR_INDEX_RANGE, "kf_elementwise: mixed-dtype descriptor is not 32-bit indexable");
    KF_PROF(MODE == 0 ? "ew_arith_cast" : "ew_convert", st);
    CastArgs<NT> a;
    memset(&a, 0, sizeof(a));
    int opidx[NT];
    for (int t = 0; t < NT; ++t) {
        a.ptr[t] = (char *)d->data[t];
        a.dtype[t] = d->dtype[t];
        opidx[t] = t;
    }
    a.op = op;
    a.n = (uint32_t)desc_numel(d);
    bool groups = desc_contiguous(d) && a.n % 8 == 0 && a.n >= 8; // whole groups of eight on 16-byte boundaries: the wide form
    for (int t = 0; groups && t < NT; ++t)
        if ((uintptr_t)d->data[t] % 16) groups = false;
    if (groups) {
        a.n /= 8;
        ew_cast8_kernel<A, NIN, MODE><<<grid_for(a.n), kBlock, 0, st>>>(a);
        KF_LAUNCH_CHECK();
        return KF_OK;
    }
    KF_REQUIRE(OffsetCalc<NT>::build(a.oc, d, opidx, 1), KF_ERR_INVALID, "kf_elementwise: bad shape/stride");
    ew_cast_kernel<A, NIN, MODE><<<grid_for(a.n), kBlock, 0, st>>>(a);
    KF_LAUNCH_CHECK();
    return KF_OK;
}

// accumulate class of a dtype: 0 float, 1 double, 2 int64, 3 bool
static int acc_class(int dt) {
    switch (dt) {
    case KF_F16: case KF_BF16: case KF_F32: return 0;
    case KF_F64: return 1;
    case KF_BOOL: return 3;
    default: return 2;
    }
}

static uint16_t host_f32_to_bf16(float f) {
    uint32_t u;
    memcpy(&u, &f, 4);
    if ((u & 0x7fffffffu) > 0x7f800000u) return 0x7FC0;
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
static uint16_t host_f32_to_f16(float f) {
    _Float16 h = (_Float16)f;
    uint16_t r;
    memcpy(&r, &h, 2);
    return r;
}

// value -> acc type of the output dtype -> output dtype (nullary_ops_kernel.cu:20-25)
static uint64_t fill_pattern(int dt, double v) {
    uint64_t bits = 0;
    switch (dt) {
    case KF_BOOL: { uint8_t x = (v != 0.0); memcpy(&bits, &x, 1); } break;
    case KF_U8: { uint8_t x = (uint8_t)(int64_t)v; memcpy(&bits, &x, 1); } break;
    case KF_I8: { int8_t x = (int8_t)(int64_t)v; memcpy(&bits, &x, 1); } break;
    case KF_I16: { int16_t x = (int16_t)(int64_t)v; memcpy(&bits, &x, 2); } break;
    case KF_I32: { int32_t x = (int32_t)(int64_t)v; memcpy(&bits, &x, 4); } break;
    case KF_I64: { int64_t x = (int64_t)v; memcpy(&bits, &x, 8); } break;
    case KF_F16: { uint16_t x = host_f32_to_f16((float)v); memcpy(&bits, &x, 2); } break;
    case KF_BF16: { uint16_t x = host_f32_to_bf16((float)v); memcpy(&bits, &x, 2); } break;
    case KF_F32: { float x = (float)v; memcpy(&bits, &x, 4); } break;
    case KF_F64: { memcpy(&bits, &v, 8); } break;
    }
    return bits;
}

template <int NIN, int MODE>
static int launch_raw_by_size(int esize, const kf_iter_desc *d, uint64_t bits, hipStream_t st) {
    switch (esize) {
    case 1: return launch_same<uint8_t, NIN, MODE>(d, 0, bits, st);
    case 2: return launch_same<uint16_t, NIN, MODE>(d, 0, bits, st);
    case 4: return launch_same<uint32_t, NIN, MODE>(d, 0, bits, st);
    default: return launch_same<uint64_t, NIN, MODE>(d, 0, bits, st);
    }
}

} // namespace kf

using namespace kf;

extern "C" int kf_elementwise(int op, const kf_iter_desc *d, int compute_dtype, double scalar, void *stream) {
    KF_REQUIRE(d, KF_ERR_INVALID, "kf_elementwise: null descriptor");
    KF_REQUIRE(op >= KF_EW_ADD && op <= KF_EW_DIV_SCALAR, KF_ERR_INVALID, "kf_elementwise: unknown op %d", op);
    KF_REQUIRE(d->ndim >= 1 && d->ndim <= KF_MAX_DIMS, KF_ERR_INVALID, "kf_elementwise: ndim %d out of range", d->ndim);
    const int nin = op <= KF_EW_DIV ? 2 : (op == KF_EW_FILL ? 0 : 1);
    KF_REQUIRE(d->noutputs == 1 && d->ntensors == nin + 1, KF_ERR_INVALID,
               "kf_elementwise: op %d wants 1 output + %d inputs, got %d/%d", op, nin, d->noutputs, d->ntensors);
    for (int t = 0; t < d->ntensors; ++t) {
        KF_REQUIRE(d->dtype[t] >= 0 && d->dtype[t] < KF_DTYPE_COUNT, KF_ERR_INVALID, "kf_elementwise: bad dtype");
        KF_REQUIRE(d->data[t], KF_ERR_INVALID, "kf_elementwise: null data pointer for operand %d", t);
    }
    const int64_t numel = desc_numel(d);
    if (numel == 0) return KF_OK;
    KF_REQUIRE(numel > 0, KF_ERR_INVALID, "kf_elementwise: negative extent");
    hipStream_t st = as_stream(stream);
    const int odt = d->dtype[0];

    if (op == KF_EW_FILL) return launch_raw_by_size<0, 2>(dtype_size(odt), d, fill_pattern(odt, scalar), st);

    if (op >= KF_EW_ADD_SCALAR) { // out = in (op) scalar, all in one dtype; the scalar is rounded to it first, as fill_ would
        KF_REQUIRE(d->dtype[1] == odt, KF_ERR_UNSUPPORTED, "kf_elementwise: scalar ops want input dtype == output dtype");
        const uint64_t bits = fill_pattern(odt, scalar);
        const int aop = op - KF_EW_ADD_SCALAR;
        switch (odt) {
        case KF_F32: return launch_same<float, 1, 3>(d, aop, bits, st);
        case KF_F64: return launch_same<double, 1, 3>(d, aop, bits, st);
        case KF_BF16: return launch_same<bf16_t, 1, 3>(d, aop, bits, st);
        case KF_F16: return launch_same<f16_t, 1, 3>(d, aop, bits, st);
        case KF_I32: return launch_same<int32_t, 1, 3>(d, aop, bits, st);
        case KF_I64: return launch_same<int64_t, 1, 3>(d, aop, bits, st);
        default: KF_REQUIRE(false, KF_ERR_UNSUPPORTED, "kf_elementwise: scalar ops cover f32/f64/bf16/f16/i32/i64 (broadcast a 1-element tensor for the rest)");
        }
    }

    if (op == KF_EW_COPY) {
        if (d->dtype[1] == odt) return launch_raw_by_size<1, 1>(dtype_size(odt), d, 0, st);
        {
            const int idt = d->dtype[1];
            const bool fam = (idt == KF_F32 || idt == KF_F16 || idt == KF_BF16) && (odt == KF_F32 || odt == KF_F16 || odt == KF_BF16);
            if (fam && desc_contiguous(d) && numel % 8 == 0 && (uintptr_t)d->data[0] % 32 == 0 && (uintptr_t)d->data[1] % 32 == 0) {
                KF_PROF("ew_convert", st);
                const int64_t n8 = numel / 8;
                const int g = grid_for(n8);
#define KF_CVT(SC, ST, DC, DT) \
    if (idt == SC && odt == DC) ew_convert8_kernel<ST, DT><<<g, kBlock, 0, st>>>((const ST *)d->data[1], (DT *)d->data[0], n8);
                KF_CVT(KF_F32, float, KF_BF16, bf16_t)
                KF_CVT(KF_F32, float, KF_F16, f16_t)
                KF_CVT(KF_BF16, bf16_t, KF_F32, float)
                KF_CVT(KF_F16, f16_t, KF_F32, float)
                KF_CVT(KF_BF16, bf16_t, KF_F16, f16_t)
                KF_CVT(KF_F16, f16_t, KF_BF16, bf16_t)
#undef KF_CVT
                KF_LAUNCH_CHECK();
                return KF_OK;
            }
        }
        switch (acc_class(odt)) { // value is cast straight to the output dtype (unary_ops_kernel.cu:13-17)
        case 0: return launch_cast<float, 1, 1>(d, op, st);
        case 1: return launch_cast<double, 1, 1>(d, op, st);
        case 2: return launch_cast<int64_t, 1, 1>(d, op, st);
        default: return launch_cast<bool, 1, 1>(d, op, st);
        }
    }

    KF_REQUIRE(compute_dtype >= 0 && compute_dtype < KF_DTYPE_COUNT, KF_ERR_INVALID, "kf_elementwise: bad compute dtype");
    const bool same = d->dtype[0] == compute_dtype && d->dtype[1] == compute_dtype && d->dtype[2] == compute_dtype;
    if (same) {
        switch (compute_dtype) {
        case KF_F32: return launch_same<float, 2, 0>(d, op, 0, st);
        case KF_F64: return launch_same<double, 2, 0>(d, op, 0, st);
        case KF_BF16: return launch_same<bf16_t, 2, 0>(d, op, 0, st);
        case KF_F16: return launch_same<f16_t, 2, 0>(d, op, 0, st);
        case KF_I32: return launch_same<int32_t, 2, 0>(d, op, 0, st);
        case KF_I64: return launch_same<int64_t, 2, 0>(d, op, 0, st);
        default: break;
        }
    }
    switch (acc_class(compute_dtype)) {
    case 0: return launch_cast<float, 2, 0>(d, op, st);
    case 1: return launch_cast<double, 2, 0>(d, op, st);
    case 2: return launch_cast<int64_t, 2, 0>(d, op, st);
    default: return launch_cast<bool, 2, 0>(d, op, st);
    }
}
