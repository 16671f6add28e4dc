// RCCL over xGMI: the single exchange step of the batch-sharded path — a sum all-reduce of the
// weight gradients. The reference has no distributed code at all; this is the
// one collective call site of the rebuild. One process per GPU; the 128-byte unique id is
// produced on rank 0 and handed to the other ranks by the host (any out-of-band channel).
#include <rccl/rccl.h>

#include "common.h"

using namespace kf;

#define KF_NCCL_TRY(expr)                                                                         \
    do {                                                                                          \
        ncclResult_t r_ = (expr);                                                                 \
        if (r_ != ncclSuccess) {                                                                  \
            ::kf::set_error("%s failed: %s (%s:%d)", #expr, ncclGetErrorString(r_), __FILE__, __LINE__); \
            return KF_ERR_COMM;                                                                   \
        }                                                                                         \
    } while (0)

static_assert(sizeof(ncclUniqueId) <= KF_COMM_ID_BYTES, "ncclUniqueId does not fit KF_COMM_ID_BYTES");

extern "C" int kf_comm_unique_id(char id[KF_COMM_ID_BYTES]) {
    KF_REQUIRE(id, KF_ERR_INVALID, "kf_comm_unique_id: null out pointer");
    ncclUniqueId u;
    KF_NCCL_TRY(ncclGetUniqueId(&u));
    memset(id, 0, KF_COMM_ID_BYTES);
    memcpy(id, &u, sizeof(u));
    return KF_OK;
}

extern "C" int kf_comm_init(void **comm, const char id[KF_COMM_ID_BYTES], int rank, int world_size) {
    KF_REQUIRE(comm && id, KF_ERR_INVALID, "kf_comm_init: null argument");
    KF_REQUIRE(world_size >= 1 && rank >= 0 && rank < world_size, KF_ERR_INVALID, "kf_comm_init: bad rank %d / world %d", rank, world_size);
    ncclUniqueId u;
    memcpy(&u, id, sizeof(u));
    ncclComm_t c;
    KF_NCCL_TRY(ncclCommInitRank(&c, world_size, u, rank));
    *comm = c;
    return KF_OK;
}

extern "C" int kf_comm_destroy(void *comm) {
    if (!comm) return KF_OK;
    KF_NCCL_TRY(ncclCommDestroy((ncclComm_t)comm));
    return KF_OK;
}

extern "C" int kf_allreduce_sum(void *comm, void *buf, size_t count, int dtype, void *stream) {
    KF_REQUIRE(comm && (buf || count == 0), KF_ERR_INVALID, "kf_allreduce_sum: null argument");
    if (count == 0) return KF_OK;
    ncclDataType_t dt;
    switch (dtype) {
    case KF_F32: dt = ncclFloat32; break;
    case KF_F64: dt = ncclFloat64; break;
    case KF_F16: dt = ncclFloat16; break;
    case KF_BF16: dt = ncclBfloat16; break;
    case KF_I32: dt = ncclInt32; break;
    case KF_I64: dt = ncclInt64; break;
    case KF_U8: dt = ncclUint8; break;
    case KF_I8: dt = ncclInt8; break;
    default: KF_REQUIRE(false, KF_ERR_UNSUPPORTED, "kf_allreduce_sum: dtype %d not supported", dtype);
    }
    KF_NCCL_TRY(ncclAllReduce(buf, buf, count, dt, ncclSum, (ncclComm_t)comm, as_stream(stream)));
    return KF_OK;
}

// several buffers, ONE collective launch (ncclGroupStart / End): the pieces of a gradient bucket that are not contiguous
extern "C" int kf_allreduce_sum_multi(void *comm, int n, void *const *bufs, const size_t *counts, int dtype, void *stream) {
    KF_REQUIRE(comm && n >= 0 && (n == 0 || (bufs && counts)), KF_ERR_INVALID, "kf_allreduce_sum_multi: null argument");
    if (n == 0) return KF_OK;
    if (n == 1) return kf_allreduce_sum(comm, bufs[0], counts[0], dtype, stream);
    KF_NCCL_TRY(ncclGroupStart());
    int rc = KF_OK;
    for (int i = 0; i < n && rc == KF_OK; ++i) rc = kf_allreduce_sum(comm, bufs[i], counts[i], dtype, stream);
    KF_NCCL_TRY(ncclGroupEnd());
    return rc;
}
