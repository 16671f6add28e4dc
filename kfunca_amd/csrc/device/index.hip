// index_put_ scatter for gfx950.
// Replaces src/device/index_ops_kernel.cu:3-15 + src/device/utils/tensor_index.h:19-143.
// One lane per (values, indices...) element: read the int64 index of every indexed dim (coalesced —
// the index tensors are walked with the iteration space), wrap negatives once, form the byte
// offset into `self`, and move the value's bits. HBM-bound byte work: no arithmetic on values, so
// the result is bit-exact for every dtype. Duplicated targets: last writer wins in unspecified
// order, as in the reference (no atomics, tensor_index.h:56-75).
#include "common.h"
#include "offset_calc.h"

namespace kf {

constexpr int kIB = 256;

struct IndexArgs {
    char *self;
    const char *values;
    const char *index[KF_MAX_TENSORS - 2];
    int64_t size[KF_MAX_TENSORS - 2];
    int64_t stride[KF_MAX_TENSORS - 2];
    int nidx;
    uint32_t n;
    OffsetCalc<KF_MAX_TENSORS> oc;
};

template <typename U>
__global__ __launch_bounds__(kIB) void index_put_kernel(const IndexArgs a) {
    const uint32_t stride = gridDim.x * kIB;
    for (uint32_t i = blockIdx.x * kIB + threadIdx.x; i < a.n; i += stride) {
        uint32_t off[KF_MAX_TENSORS];
        a.oc.get(i, off);
        int64_t target = 0;
#pragma unroll
        for (int k = 0; k < KF_MAX_TENSORS - 2; ++k) {
            if (k < a.nidx) {
                int64_t idx = *(const int64_t *)(a.index[k] + off[k + 2]);
                if (idx < 0) idx += a.size[k];
                target += idx * a.stride[k];
            }
        }
        *(U *)(a.self + off[0] + target) = *(const U *)(a.values + off[1]);
    }
}

} // namespace kf

using namespace kf;

extern "C" int kf_index_put(const kf_iter_desc *d, int nidx, const int64_t *sizes, const int64_t *strides_bytes,
                            void *stream) {
    KF_REQUIRE(d && sizes && strides_bytes, KF_ERR_INVALID, "kf_index_put: null argument");
    KF_REQUIRE(nidx >= 1 && nidx <= KF_MAX_TENSORS - 2, KF_ERR_INVALID, "kf_index_put: nidx %d out of range", nidx);
    KF_REQUIRE(d->ntensors == nidx + 2 && d->noutputs == 1, KF_ERR_INVALID, "kf_index_put: wants self, values and %d indices", nidx);
    KF_REQUIRE(d->ndim >= 1 && d->ndim <= KF_MAX_DIMS, KF_ERR_INVALID, "kf_index_put: ndim out of range");
    KF_REQUIRE(d->dtype[0] == d->dtype[1], KF_ERR_INVALID, "kf_index_put: self/values dtype mismatch");
    for (int k = 0; k < nidx; ++k) KF_REQUIRE(d->dtype[k + 2] == KF_I64, KF_ERR_INVALID, "kf_index_put: indices must be int64");
    const int64_t n = desc_numel(d);
    if (n == 0) return KF_OK;
    KF_REQUIRE(desc_is_32bit(d), KF_ERR_INDEX_RANGE, "kf_index_put: descriptor is not 32-bit indexable");
    IndexArgs a;
    memset(&a, 0, sizeof(a));
    a.self = (char *)d->data[0];
    a.values = (const char *)d->data[1];
    a.nidx = nidx;
    a.n = (uint32_t)n;
    int opidx[KF_MAX_TENSORS];
    for (int t = 0; t < KF_MAX_TENSORS; ++t) opidx[t] = t < d->ntensors ? t : -1;
    for (int k = 0; k < nidx; ++k) {
        a.index[k] = (const char *)d->data[k + 2];
        a.size[k] = sizes[k];
        a.stride[k] = strides_bytes[k];
        KF_REQUIRE(a.index[k], KF_ERR_INVALID, "kf_index_put: null index pointer");
    }
    KF_REQUIRE(a.self && a.values, KF_ERR_INVALID, "kf_index_put: null data pointer");
    KF_REQUIRE(OffsetCalc<KF_MAX_TENSORS>::build(a.oc, d, opidx, 1), KF_ERR_INVALID, "kf_index_put: bad shape/stride");
    int64_t blocks = (n + kIB - 1) / kIB;
    if (blocks > 2048) blocks = 2048;
    hipStream_t st = as_stream(stream);
    KF_PROF("index_put", st);
    switch (dtype_size(d->dtype[0])) {
    case 1: index_put_kernel<uint8_t><<<(unsigned)blocks, kIB, 0, st>>>(a); break;
    case 2: index_put_kernel<uint16_t><<<(unsigned)blocks, kIB, 0, st>>>(a); break;
    case 4: index_put_kernel<uint32_t><<<(unsigned)blocks, kIB, 0, st>>>(a); break;
    case 8: index_put_kernel<uint64_t><<<(unsigned)blocks, kIB, 0, st>>>(a); break;
    default: KF_REQUIRE(false, KF_ERR_INVALID, "kf_index_put: bad dtype");
    }
    KF_LAUNCH_CHECK();
    return KF_OK;
}
