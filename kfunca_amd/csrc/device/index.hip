// index_put_ scatter, row gather (embedding) and its sorted scatter-add backward for gfx950.
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

// ---- row gather (embedding, README.md:30; the read side of the index arithmetic of tensor_index.h:56-104) -------------------
// out[n, :] = table[wrap(idx[n]), :]: the output is walked as a flat array of U-sized units, one per lane, grid = the problem
// (this memory system wants many short waves): unit u -> row u / upr, unit u % upr of it. Lanes of one
// row read the same index word (a broadcast out of L1). Byte mover: bit-exact for every dtype.
template <typename U>
__global__ __launch_bounds__(256) void index_get_kernel(const char *table, int64_t nrows, int64_t row_bytes, const int64_t *idx, int64_t units,
                                                        uint32_t upr, char *out) {
    const int64_t u = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (u >= units) return;
    const int64_t n = u / upr;
    const uint32_t c = (uint32_t)(u - n * upr);
    int64_t r = idx[n];
    if (r < 0) r += nrows;
    *(U *)(out + n * row_bytes + (int64_t)c * sizeof(U)) = *(const U *)(table + r * row_bytes + (int64_t)c * sizeof(U));
}

// ---- row scatter-add (embedding backward; no reference counterpart) --------------------------------------------------------
// The indices are wrapped, then stably sorted (kf_sort, int64 keys + positions); then
// dst[r, :] = sum over the run of equal sorted indices r of src[pos[j], :], added in run order (the stable sort's order =
// input order): one wave per run START (the other waves leave at once), f32 accumulation, no atomics - bitwise reproducible.
template <typename T> __device__ __forceinline__ float ia_ld(const T *p) { return (float)*p; }
template <> __device__ __forceinline__ float ia_ld<bf16_t>(const bf16_t *p) { return bf16_to_f32(*p); }
template <> __device__ __forceinline__ float ia_ld<f16_t>(const f16_t *p) { return f16_to_f32(*p); }
template <typename T> __device__ __forceinline__ void ia_st(T *p, float v) { *p = (T)v; }
template <> __device__ __forceinline__ void ia_st<bf16_t>(bf16_t *p, float v) { *p = f32_to_bf16(v); }
template <> __device__ __forceinline__ void ia_st<f16_t>(f16_t *p, float v) { *p = f32_to_f16(v); }

template <typename K>
__global__ __launch_bounds__(256) void index_wrap_kernel(const int64_t *idx, int64_t n, int64_t nrows, K *out) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) {
        int64_t r = idx[i];
        r = r < 0 ? r + nrows : r;
        // an index outside [-nrows, nrows) names no row: it gets the key nrows (one past the last row), sorts behind every real row and its
        // run is skipped by the add kernels - truncated to K it would look like a valid row AND break the promise made to the radix sort
        // that no key differs above bit log2(nrows)
        out[i] = (K)((r < 0 || r >= nrows) ? nrows : r);
    }
}

// One wave per run of equal destination rows, rows moved as 16-byte packs (round 5; rows of whole packs on 16-byte boundaries): the run's
// positions are fetched 64 at a time into the lanes and handed round with readlane, a lane keeps the f32 sums of up to PACKS packs of the
// row (a wider row is walked in column blocks), the run's rows are added in input order - the element-per-lane kernel below adds in the
// same order, so the two agree bit for bit. bf16 [32768 x 4096] into 128256 rows: 0.44 -> see DESIGN §4 (2-byte loads, the run re-read per
// 256-column chunk).
template <typename T, typename K, int PACKS>
__global__ __launch_bounds__(256) void index_add_sorted_vec_kernel(const K *key, const int64_t *pos, int64_t n, const T *src, int64_t cols, int64_t nrows, T *dst) {
    constexpr int V = 16 / (int)sizeof(T);
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    const K k = key[j];
    if (j > 0 && key[j - 1] == k) return; // not a run start
    if ((int64_t)k >= nrows) return;      // the run of out-of-range indices (index_wrap_kernel): no row
    const int64_t npk = cols / V;
    for (int64_t b0 = 0; b0 < npk; b0 += 64 * PACKS) { // column block: PACKS packs per lane
        float acc[PACKS][V];
#pragma unroll
        for (int q = 0; q < PACKS; ++q)
#pragma unroll
            for (int e = 0; e < V; ++e) acc[q][e] = 0.f;
        for (int64_t base = j;; base += 64) { // the run, 64 rows at a time
            const int64_t jj = base + lane;
            const bool same = jj < n && key[jj] == k;
            const uint64_t m = __ballot(same);
            const int cnt = m == ~0ull ? 64 : __builtin_ctzll(~m); // leading rows of this chunk that belong to the run
            const int64_t p = same ? pos[jj] : 0;
            for (int r = 0; r < cnt; ++r) {
                const int64_t pr = ((int64_t)__builtin_amdgcn_readlane((int)(p >> 32), r) << 32) | (uint32_t)__builtin_amdgcn_readlane((int)p, r);
                const uint4 *row = (const uint4 *)(src + pr * cols);
                uint4 raw[PACKS];
#pragma unroll
                for (int q = 0; q < PACKS; ++q) {
                    const int64_t c = b0 + (int64_t)q * 64 + lane;
                    raw[q] = row[c < npk ? c : npk - 1]; // (clamped, no branch around the load)
                }
#pragma unroll
                for (int q = 0; q < PACKS; ++q) {
                    T t[V];
                    __builtin_memcpy(t, &raw[q], 16);
#pragma unroll
                    for (int e = 0; e < V; ++e) acc[q][e] += ia_ld(&t[e]);
                }
            }
            if (cnt < 64) break;
        }
#pragma unroll
        for (int q = 0; q < PACKS; ++q) {
            const int64_t c = b0 + (int64_t)q * 64 + lane;
            if (c < npk) {
                T t[V];
#pragma unroll
                for (int e = 0; e < V; ++e) ia_st(&t[e], acc[q][e]);
                uint4 o;
                __builtin_memcpy(&o, t, 16);
                ((uint4 *)(dst + (int64_t)k * cols))[c] = o;
            }
        }
    }
}

template <typename T, typename K>
__global__ __launch_bounds__(256) void index_add_sorted_kernel(const K *key, const int64_t *pos, int64_t n, const T *src, int64_t cols,
                                                               int64_t nrows, T *dst) {
    const int64_t j = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (j >= n) return;
    const K k = key[j];
    if (j > 0 && key[j - 1] == k) return; // not a run start
    const int64_t r = (int64_t)k; // already wrapped
    if (r >= nrows) return;       // the run of out-of-range indices: no row
    for (int64_t c0 = 0; c0 < cols; c0 += 64 * 4) { // 4 columns per lane per sweep, the run re-walked per column chunk
        float acc[4] = {0.f, 0.f, 0.f, 0.f};
        for (int64_t jj = j; jj < n && key[jj] == k; ++jj) {
            const T *row = src + pos[jj] * cols;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int64_t c = c0 + (int64_t)i * 64 + lane;
                if (c < cols) acc[i] += ia_ld(row + c);
            }
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int64_t c = c0 + (int64_t)i * 64 + lane;
            if (c < cols) ia_st(dst + r * cols + c, acc[i]);
        }
    }
}

} // namespace kf

using namespace kf;

extern "C" int kf_index_get(const void *table, int64_t nrows, int64_t row_bytes, const int64_t *idx, int64_t n, void *out, void *stream) {
    KF_REQUIRE(nrows >= 0 && row_bytes >= 0 && n >= 0, KF_ERR_INVALID, "kf_index_get: negative extent");
    if (n == 0 || row_bytes == 0) return KF_OK;
    KF_REQUIRE(table && idx && out && nrows > 0, KF_ERR_INVALID, "kf_index_get: null operand or empty table");
    // the widest unit that divides the row and both base addresses
    const uintptr_t bits = (uintptr_t)table | (uintptr_t)out | (uintptr_t)row_bytes;
    const int us = (bits % 16 == 0) ? 16 : (bits % 8 == 0) ? 8 : (bits % 4 == 0) ? 4 : (bits % 2 == 0) ? 2 : 1;
    const int64_t upr = row_bytes / us, units = upr * n;
    KF_REQUIRE(upr <= 0xffffffffLL && (units + 255) / 256 <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "kf_index_get: too many units for one launch");
    const unsigned grid = (unsigned)((units + 255) / 256);
    hipStream_t st = as_stream(stream);
    KF_PROF("index_get", st);
    const char *t = (const char *)table;
    char *o = (char *)out;
    switch (us) {
    case 16: index_get_kernel<uint4><<<grid, 256, 0, st>>>(t, nrows, row_bytes, idx, units, (uint32_t)upr, o); break;
    case 8: index_get_kernel<uint64_t><<<grid, 256, 0, st>>>(t, nrows, row_bytes, idx, units, (uint32_t)upr, o); break;
    case 4: index_get_kernel<uint32_t><<<grid, 256, 0, st>>>(t, nrows, row_bytes, idx, units, (uint32_t)upr, o); break;
    case 2: index_get_kernel<uint16_t><<<grid, 256, 0, st>>>(t, nrows, row_bytes, idx, units, (uint32_t)upr, o); break;
    default: index_get_kernel<uint8_t><<<grid, 256, 0, st>>>(t, nrows, row_bytes, idx, units, (uint32_t)upr, o); break;
    }
    KF_LAUNCH_CHECK();
    return KF_OK;
}

static inline size_t ia_align(size_t v) { return (v + 255) / 256 * 256; }

extern "C" size_t kf_index_add_workspace_bytes(int64_t n) {
    if (n <= 0) return 0;
    return 3 * ia_align((size_t)n * 8) + ia_align(kf_sort_workspace_bytes(KF_I64, 1, n)); // wrapped | sorted | positions | sort scratch
}

extern "C" int kf_index_add(int dtype, const int64_t *idx, int64_t n, const void *src, int64_t cols, int64_t nrows, void *dst, void *workspace,
                            size_t workspace_bytes, void *stream) {
    KF_REQUIRE(dtype == KF_F32 || dtype == KF_BF16 || dtype == KF_F16, KF_ERR_UNSUPPORTED, "kf_index_add: dtype %d not supported", dtype);
    KF_REQUIRE(n >= 0 && cols >= 0 && nrows >= 0, KF_ERR_INVALID, "kf_index_add: negative extent");
    if (n == 0 || cols == 0) return KF_OK;
    KF_REQUIRE(idx && src && dst && nrows > 0, KF_ERR_INVALID, "kf_index_add: null operand or empty destination");
    const size_t need = kf_index_add_workspace_bytes(n);
    KF_REQUIRE(workspace && workspace_bytes >= need, KF_ERR_WORKSPACE, "kf_index_add: workspace of %zu bytes required, got %zu", need, workspace_bytes);
    KF_REQUIRE((n + 3) / 4 <= 0x7fffffffLL, KF_ERR_INDEX_RANGE, "kf_index_add: too many indices for one launch");
    hipStream_t st = as_stream(stream);
    char *ws = (char *)workspace;
    void *wrapped = ws, *sorted = ws + ia_align((size_t)n * 8);
    int64_t *pos = (int64_t *)(ws + 2 * ia_align((size_t)n * 8));
    void *sort_ws = ws + 3 * ia_align((size_t)n * 8);
    // rows are numbered below 2^31 wherever a 32-bit key can hold them: four radix passes instead of eight (bf16 embedding backward, 32768 tokens:
    // the sort 0.133 -> see DESIGN §4)
    const bool k32 = nrows < 0x7fffffffLL;   // (the key nrows itself must fit: it marks out-of-range indices)
    {
        KF_PROF("index_wrap", st);
        if (k32) index_wrap_kernel<int32_t><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(idx, n, nrows, (int32_t *)wrapped);
        else index_wrap_kernel<int64_t><<<(unsigned)((n + 255) / 256), 256, 0, st>>>(idx, n, nrows, (int64_t *)wrapped);
        KF_LAUNCH_CHECK();
    }
    const int kcode = k32 ? KF_I32 : KF_I64;
    const size_t sws = kf_sort_workspace_bytes(kcode, 1, n);
    int bits = 1;
    while (bits < 63 && ((int64_t)1 << bits) <= nrows) ++bits; // keys are 0..nrows inclusive: the key bytes above that are the same in every key
    int rc = sort_with_key_bits(kcode, wrapped, sorted, pos, 1, n, 0, sws ? sort_ws : nullptr, sws, stream, bits); // stable: equal rows keep input order
    if (rc != KF_OK) return rc;
    const unsigned grid = (unsigned)((n + 3) / 4);
    KF_PROF("index_add_sorted", st);
    const int es = dtype == KF_F32 ? 4 : 2;
    const bool vec = (cols * es) % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
    const int64_t npk = cols * es / 16;
#define KF_IA(T_, K_)                                                                                                                              \
    {                                                                                                                                              \
        if (!vec) index_add_sorted_kernel<T_, K_><<<grid, 256, 0, st>>>((const K_ *)sorted, pos, n, (const T_ *)src, cols, nrows, (T_ *)dst);        \
        else if (npk <= 64) index_add_sorted_vec_kernel<T_, K_, 1><<<grid, 256, 0, st>>>((const K_ *)sorted, pos, n, (const T_ *)src, cols, nrows, (T_ *)dst); \
        else if (npk <= 128) index_add_sorted_vec_kernel<T_, K_, 2><<<grid, 256, 0, st>>>((const K_ *)sorted, pos, n, (const T_ *)src, cols, nrows, (T_ *)dst); \
        else if (npk <= 256) index_add_sorted_vec_kernel<T_, K_, 4><<<grid, 256, 0, st>>>((const K_ *)sorted, pos, n, (const T_ *)src, cols, nrows, (T_ *)dst); \
        else index_add_sorted_vec_kernel<T_, K_, 8><<<grid, 256, 0, st>>>((const K_ *)sorted, pos, n, (const T_ *)src, cols, nrows, (T_ *)dst);             \
    }
#define KF_IA_K(T_) \
    if (k32) KF_IA(T_, int32_t) else KF_IA(T_, int64_t)
    if (dtype == KF_F32) { KF_IA_K(float) } else if (dtype == KF_BF16) { KF_IA_K(bf16_t) } else { KF_IA_K(f16_t) }
#undef KF_IA_K
#undef KF_IA
    KF_LAUNCH_CHECK();
    return KF_OK;
}

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
