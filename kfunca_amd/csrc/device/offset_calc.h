// Linear index -> per-operand byte offsets for strided iteration spaces (<= 12 dims).
// Division by the (runtime) dim sizes uses a multiply-high + shift; valid for n < 2^31, which the
// 32-bit-indexable contract of the strided kernels guarantees.
#pragma once

#include "common.h"

namespace kf {

struct FastDivU32 {
    uint32_t d = 1, magic = 1, shift = 0;
    FastDivU32() = default;
    // n / d = (mulhi(n, magic) + n) >> shift for every n < 2^31, with shift = ceil(log2 d) and magic = ceil(2^(32 + shift) / d) - 2^32
    // (the round-up reciprocal with its implicit leading one split off, so it fits 32 bits; a power of two gets magic 0). Divisors here
    // are dim sizes <= 2^31 - 1, so shift <= 31 and 2^(32 + shift) fits 64 bits.
    explicit FastDivU32(uint32_t divisor) : d(divisor) {
        shift = divisor <= 1 ? 0u : 32u - (uint32_t)__builtin_clz(divisor - 1);
        const uint64_t pow = 1ull << (32 + shift);
        magic = (uint32_t)((pow + divisor - 1) / divisor - (1ull << 32));
    }
    __device__ __forceinline__ uint32_t div(uint32_t n) const { return (__umulhi(n, magic) + n) >> shift; }
};

template <int NT>
struct OffsetCalc {
    int ndim;
    FastDivU32 size[KF_MAX_DIMS];
    uint32_t stride[KF_MAX_DIMS][NT];

    // dim0_div: dimension 0 is walked in units of `dim0_div` elements (vectorised kernels).
    static bool build(OffsetCalc &oc, const kf_iter_desc *d, const int *operand_idx, int64_t dim0_div = 1) {
        oc.ndim = d->ndim;
        for (int i = 0; i < KF_MAX_DIMS; ++i) {
            int64_t s = i < d->ndim ? d->shape[i] : 1;
            if (i == 0) s /= dim0_div;
            if (s <= 0 || s > 0x7fffffff) return false;
            oc.size[i] = FastDivU32((uint32_t)s);
            for (int t = 0; t < NT; ++t) {
                int64_t st = 0;
                if (i < d->ndim && operand_idx[t] >= 0) st = d->stride_bytes[operand_idx[t]][i];
                if (i == 0) st *= dim0_div;
                if (st < 0 || st > 0x7fffffff) return false;
                oc.stride[i][t] = (uint32_t)st;
            }
        }
        return true;
    }

    __device__ __forceinline__ void get(uint32_t linear, uint32_t (&off)[NT]) const {
#pragma unroll
        for (int t = 0; t < NT; ++t) off[t] = 0;
#pragma unroll
        for (int i = 0; i < KF_MAX_DIMS; ++i) {
            if (i == ndim) break;
            uint32_t q = size[i].div(linear);
            uint32_t r = linear - q * size[i].d;
            linear = q;
#pragma unroll
            for (int t = 0; t < NT; ++t) off[t] += r * stride[i][t];
        }
    }
};

// numel and every operand's max byte offset fit in int32 (reference: tensor_iterator.cpp:381-396)
static inline bool desc_is_32bit(const kf_iter_desc *d) {
    int64_t n = 1;
    for (int i = 0; i < d->ndim; ++i) n *= d->shape[i];
    if (n > 0x7fffffffLL) return false;
    for (int t = 0; t < d->ntensors; ++t) {
        int64_t mx = 1;
        for (int i = 0; i < d->ndim; ++i) {
            if (d->stride_bytes[t][i] < 0) return false;
            mx += (d->shape[i] - 1) * d->stride_bytes[t][i];
        }
        if (mx > 0x7fffffffLL) return false;
    }
    return true;
}

static inline int64_t desc_numel(const kf_iter_desc *d) {
    int64_t n = 1;
    for (int i = 0; i < d->ndim; ++i) n *= d->shape[i];
    return n;
}

} // namespace kf
