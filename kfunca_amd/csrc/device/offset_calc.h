// Linear index -> per-operand byte offsets for strided iteration spaces (<= 12 dims).
// Division by the (runtime) dim sizes uses a multiply-high + shift; valid for n < 2^31, which the
// 32-bit-indexable contract of the strided kernels guarantees.
#pragma once

#include "common.h"

namespace kf {

struct FastDivU32 {
    uint32_t d = 1, magic = 1, shift = 0;
    FastDivU32() = default;
    explicit FastDivU32(uint32_t divisor) : d(divisor) {
        // smallest s with 2^s >= d, magic = floor(2^32 * (2^s - d) / d) + 1
        for (shift = 0; shift < 32; ++shift)
            if ((1ull << shift) >= d) break;
        uint64_t one = 1;
        magic = (uint32_t)(((one << 32) * ((one << shift) - d)) / d + 1);
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
