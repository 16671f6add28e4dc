/*
 * kfunca_hip.h — the C ABI of the MI355X (gfx950) device layer.
 *
 * This header is the drop-in boundary: everything the reference's host core
 * (src/core) reaches in its device layer (src/device/include/*.h) is reachable here as an
 * `extern "C"` function taking plain pointers, sizes and POD descriptors — no C++ types, no
 * torch types, nothing thrown. Each entry cites the reference interface it replaces.
 *
 * Conventions
 *   - every function returns `int` status: 0 = KF_OK, anything else is an error whose text is
 *     available from kf_last_error() (thread-local). The host core turns a non-zero status into
 *     the reference's `utils::Error` (reference: src/core/utils/exception.h:123-131).
 *   - the device layer owns NO memory: outputs, workspaces and semaphores are allocated by the
 *     caller (the host core's caching allocator) and handed in. This breaks the reference's
 *     L1→L2 link cycle (last paragraph).
 *   - `stream` is a hipStream_t passed as void*; NULL means the device's null stream.
 *   - dtype codes follow the reference's ScalarType order (src/core/include/scalar_type.h:9-27).
 */
#ifndef KFUNCA_HIP_H_
#define KFUNCA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define KF_ABI_VERSION 7 /* 7 (no signature changed): kf_attn_* run the matrix-core kernels on ANY sequence lengths with Skv >= Sq (no multiple-of-128 rule), the backward workspace's row-constant arrays pad Sq to 32 and its dS part has a second, half-size layout (the causal half: taken when the workspace does not hold full rows for every pair, or under KF_ATTN_DS_TRI) - a caller must size the workspace with THIS library's query, kf_index_add drops indices outside [-nrows, nrows); 6: + KF_ERR_OOM from kf_malloc, kf_gemm_epilogue.c_f32 / kf_gemm_problem.c_f32 (float output behind 16-bit operands); 2: + kf_reduce_moments*, KF_EW_*_SCALAR, kf_graph_*, kf_attn_*_scaled; 3: + kf_sort*; 4: + kf_knobs_reload, kf_norm_*, kf_index_get, kf_gemm_ex, KF_EPI_*; 5: + kf_gemm_grouped_single_grid, kf_allreduce_sum_multi, kf_profile_samples, kf_attn_bwd accepts any workspace >= the statistics (all additive) */

/* ---- status ------------------------------------------------------------------------------ */
enum {
    KF_OK = 0,
    KF_ERR_HIP = 1,         /* a HIP runtime call or a kernel launch failed                  */
    KF_ERR_INVALID = 2,     /* bad argument (shape/dtype/alignment/null pointer)             */
    KF_ERR_UNSUPPORTED = 3, /* valid request the device layer has no kernel for              */
    KF_ERR_INDEX_RANGE = 4, /* descriptor not 32-bit indexable; caller must split (a6)       */
    KF_ERR_WORKSPACE = 5,   /* workspace missing or too small                                */
    KF_ERR_COMM = 6,        /* RCCL failure                                                  */
    KF_ERR_OOM = 7          /* kf_malloc: the device has no room for this allocation; nothing else is affected (HIP's sticky last-error is cleared), the caller may free and retry */
};

/* ---- dtypes: reference ScalarType order (scalar_type.h:9-27) ------------------------------ */
enum {
    KF_BOOL = 0,
    KF_U8 = 1,
    KF_I8 = 2,
    KF_I16 = 3,
    KF_I32 = 4,
    KF_I64 = 5,
    KF_F16 = 6,
    KF_BF16 = 7,
    KF_F32 = 8,
    KF_F64 = 9,
    KF_DTYPE_COUNT = 10
};

#define KF_MAX_DIMS 12   /* reference MAX_TENSOR_DIMS, tensor.h:7 */
#define KF_MAX_TENSORS 8 /* reference TensorIterator::MAX_TENSORS, tensor_iterator.h:22-24 */

/*
 * The post-build() state of the reference's TensorIterator (tensor_iterator.h:27-47), as a POD.
 * Operands are ordered outputs first, then inputs. Dimension 0 is the fastest-moving one
 * (the iterator reorders dims fastest-first, tensor_iterator.cpp:181-244). Strides are BYTES;
 * a stride of 0 marks a broadcast (input) or reduced (output) dimension.
 */
typedef struct kf_iter_desc {
    int32_t ndim;      /* 1..KF_MAX_DIMS after coalescing                     */
    int32_t ntensors;  /* noutputs + ninputs, <= KF_MAX_TENSORS              */
    int32_t noutputs;
    int32_t reserved;
    int32_t dtype[KF_MAX_TENSORS];
    int64_t shape[KF_MAX_DIMS];
    int64_t stride_bytes[KF_MAX_TENSORS][KF_MAX_DIMS];
    void *data[KF_MAX_TENSORS];
} kf_iter_desc;

/* ---- runtime: replaces memory_engine.h:5-10 and Launcher (launcher_cuda.h:105-354) -------- */
const char *kf_last_error(void);
int kf_abi_version(void);
/* content hash (16 hex digits) of the device sources this library was linked from: kfunca_amd/_build.py stamps it in at link time
 * (no reference counterpart: launcher_cuda.h has no build identity). bench.py prints it as device_lib_sha and refuses to quote
 * profile-derived figures when it differs from the source tree it runs in. */
const char *kf_build_source_sha(void);
int kf_device_count(int *count);
int kf_set_device(int device);              /* dset_device, memory_engine.h:5           */
int kf_get_device(int *device);
int kf_malloc(void **ptr, size_t bytes);    /* dmalloc, memory_engine.h:6               */
int kf_free(void *ptr);                     /* dfree, memory_engine.h:7                 */
int kf_memcpy_h2d(void *dst, const void *src, size_t bytes, void *stream); /* :8  (synchronous on return) */
int kf_memcpy_d2h(void *dst, const void *src, size_t bytes, void *stream); /* :9  (synchronous on return) */
int kf_memcpy_d2d(void *dst, const void *src, size_t bytes, void *stream); /* async on stream           */
int kf_memset_zero(void *ptr, size_t bytes, void *stream);                 /* dmemset_zeros, :10 (async) */
int kf_stream_create(void **stream);        /* Launcher::stream_begin, launcher_cuda.h:113-118 */
int kf_stream_destroy(void *stream);
int kf_stream_sync(void *stream);           /* Launcher::stream_sync, launcher_cuda.h:125-127  */
int kf_stream_wait_event(void *stream, void *event); /* cross-stream ordering without a host sync */
int kf_device_sync(void);
int kf_event_create(void **event);          /* per-launch timing mode, launcher_cuda.h:336-345 */
int kf_event_destroy(void *event);
int kf_event_record(void *event, void *stream);
int kf_event_sync(void *event);
int kf_event_elapsed_ms(void *start, void *stop, float *ms);

/* HIP graphs for launch-bound sequences (no reference counterpart: its Launcher submits kernel by kernel,
 * launcher_cuda.h:330-354). Every compute entry of this library only enqueues work on the stream it is given and owns
 * no memory, so a sequence of calls between begin and end is recorded instead of executed; kf_graph_launch replays it
 * with one submission. `stream` must be a stream from kf_stream_create (not NULL). */
int kf_graph_begin_capture(void *stream);
int kf_graph_end_capture(void *stream, void **graph_exec);
int kf_graph_launch(void *graph_exec, void *stream);
int kf_graph_destroy(void *graph_exec);

/* per-launch timing mode: replaces Launcher::set_profiling_mode + the cudaEvent pair around each
 * submit (launcher_cuda.h:253-255,336-345). While enabled, every kernel launch made through this
 * library is bracketed by a HIP event pair on its own stream (no host sync); kf_profile_get()
 * synchronises and reports, per kernel name, the number of launches and their summed duration. */
int kf_profile_enable(int on);
int kf_profile_reset(void);
int kf_profile_count(int *n);
int kf_profile_get(int i, char name[64], double *total_ms, int64_t *launches);
/* the durations of entry i's individual launches, in launch order (at most the first 65536 are kept): percentiles, run-to-run spread */
int kf_profile_samples(int i, float *ms, int capacity, int *written);

/* A/B switches (KF_* environment variables: kernel-selection knobs for benchmarks and parity tests, the counterpart of
 * the reference's compile-time constants, SURVEY.md section 5 "Config / flags") are read once per process; this re-reads them. */
int kf_knobs_reload(void);

typedef struct kf_device_props {
    char name[256];
    char arch[64];
    int32_t compute_units;
    int32_t wavefront_size;
    int32_t max_threads_per_block;
    int32_t clock_khz;
    int32_t memory_clock_khz;
    int32_t memory_bus_bits;
    int64_t lds_per_block;
    int64_t l2_bytes;
    uint64_t total_mem;
    uint64_t free_mem;
} kf_device_props;
int kf_device_props_get(int device, kf_device_props *out); /* device_info.h:5 (query half) */

/* ---- elementwise loops: replaces binary/unary/nullary_ops_kernel.h ------------------------ */
enum {
    KF_EW_ADD = 0,  /* add_kernel,  binary_ops_kernel.h:5 */
    KF_EW_SUB = 1,  /* sub_kernel,  :6 */
    KF_EW_MUL = 2,  /* mul_kernel,  :7 */
    KF_EW_DIV = 3,  /* div_kernel,  :8 */
    KF_EW_COPY = 4, /* copy_kernel, unary_ops_kernel.h:5 (also dtype convert) */
    KF_EW_FILL = 5, /* fill_kernel, nullary_ops_kernel.h:5 */
    /* out = in (op) scalar: what `tensor + 2.0` computes in the reference through a filled temporary
     * (register.cpp:172-206: empty_like(self).fill_(s), then the binary kernel) without the temporary */
    KF_EW_ADD_SCALAR = 6,
    KF_EW_SUB_SCALAR = 7,
    KF_EW_MUL_SCALAR = 8,
    KF_EW_DIV_SCALAR = 9
};
/*
 * out = f(in...) over the iteration space of `desc` (1 output; 2 inputs for ADD..DIV, 1 for
 * COPY and the *_SCALAR ops, 0 for FILL). *_SCALAR: input and output share one dtype (f32, f64, bf16,
 * f16, i32, i64); `scalar` is first rounded to that dtype exactly as FILL would, then the op runs in
 * its accumulate type — bit-identical to the reference's fill-then-op sequence. `compute_dtype` is the reference's iter.common_dtype() for ADD..DIV
 * (binary_ops_kernel.cu:34-60; arithmetic runs in its accumulate type: float for half/bf16,
 * int64 for ints, accumulate_type.h:17-27), ignored for COPY (value is cast to the output dtype,
 * unary_ops_kernel.cu:13-17) and FILL (`scalar` cast to the output's accumulate type then to the
 * output dtype, nullary_ops_kernel.cu:20-25).
 * Contiguous descriptors of any size are accepted; strided ones must be 32-bit indexable
 * (KF_ERR_INDEX_RANGE otherwise — the host splits, tensor_iterator.cpp:415-480).
 */
int kf_elementwise(int op, const kf_iter_desc *desc, int compute_dtype, double scalar, void *stream);

/* ---- reductions: replaces reduce_ops_kernel.h:5-6 ------------------------------------------ */
enum {
    KF_RED_SUM = 0, /* sum_kernel  */
    KF_RED_MEAN = 1 /* mean_kernel */
};
/*
 * desc: 1 output, 1 input, built by the reference's build_for_reduce (tensor_iterator.cpp:523-528):
 * reduced dims are the ones where the OUTPUT stride is 0 and shape > 1.
 * The caller supplies scratch of at least kf_reduce_workspace_bytes(desc) bytes (may be 0);
 * it needs no initialisation.
 */
int kf_reduce_workspace_bytes(const kf_iter_desc *desc, size_t *bytes);
int kf_reduce(int op, const kf_iter_desc *desc, void *workspace, size_t workspace_bytes, void *stream);

/* ---- moments: replaces mean_var_kernel (reduce_ops_kernel.h:7, reduce_ops_kernel.cu:61-153) and
 *      norm_stat_kernel (norm_ops_kernel.h, norm_ops_kernel.cu:6-61) ------------------------------------ */
enum {
    KF_MOM_VAR = 0,   /* out0 = M2 / max(n - correction, 0)            (mean_var, take_sqrt = false) */
    KF_MOM_STD = 1,   /* out0 = sqrt(M2 / max(n - correction, 0))      (mean_var, take_sqrt = true)  */
    KF_MOM_INVSTD = 2 /* out0 = 1 / sqrt(M2 / n + eps)                 (norm_stat, welford_norm.h:183) */
};
/*
 * desc: 2 outputs + 1 input in the reference iterator's order (reduce_ops.cpp:24-25): [0] = the variance-like
 * output, [1] = the mean, [2] = the input; both outputs reduce the same dims (stride 0 there).
 * Floating dtypes only; the outputs have the input's dtype, or f32 for f16 / bf16 inputs.
 * Scratch as for kf_reduce (query kf_reduce_moments_workspace_bytes).
 */
int kf_reduce_moments_workspace_bytes(const kf_iter_desc *desc, size_t *bytes);
int kf_reduce_moments(int mode, const kf_iter_desc *desc, double correction, double eps, void *workspace,
                      size_t workspace_bytes, void *stream);

/* ---- row normalisations: finishes the reference's roadmap item rms_norm (README.md:28) on its building block norm_stat
 *      (norm_ops_kernel.h:5, norm_ops_kernel.cu:6-61; invstd = 1 / sqrt(M2 / n + eps), welford_norm.h:170-187) -------------- */
enum {
    KF_NORM_RMS = 0,  /* y = x * rstd * w,                 rstd = 1 / sqrt(mean(x^2) + eps)            */
    KF_NORM_LAYER = 1 /* y = (x - mean) * rstd * w + b,    rstd = 1 / sqrt(mean((x - mean)^2) + eps)   */
};
/*
 * x, y, dy, dx: [rows, cols] with row stride `ld` elements (ld >= cols); weight, bias, dweight, dbias: [cols] (any may be
 * NULL: w = 1, b = 0; RMS takes no bias); dtype in {KF_F32, KF_BF16, KF_F16} for all of them; statistics in f32:
 * mean[rows] (LAYER only), rstd[rows] - the forward writes them (NULL: not kept), the backward reads them.
 * Backward: dx = rstd * (g - mean(g) - xhat * mean(g * xhat)) with g = dy * w, xhat = (x - mean) * rstd (RMS: no mean(g)
 * term, mean = 0); dweight = sum_rows dy * xhat, dbias = sum_rows dy - summed in a fixed order through caller scratch of
 * kf_norm_bwd_workspace_bytes() bytes (no atomics: bitwise reproducible; the scratch needs no initialisation).
 */
int kf_norm_fwd(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, const void *x, const void *weight, const void *bias,
                double eps, void *y, float *mean, float *rstd, void *stream);
int kf_norm_bwd_workspace_bytes(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, size_t *bytes);
int kf_norm_bwd(int kind, int dtype, int64_t rows, int64_t cols, int64_t ld, const void *x, const void *weight, const float *mean,
                const float *rstd, const void *dy, void *dx, void *dweight, void *dbias, void *workspace, size_t workspace_bytes,
                void *stream);

/* ---- index_put_: replaces index_ops_kernel.h:5 --------------------------------------------- */
/*
 * desc operands: [0] = self viewed with stride 0 (index_ops.cpp:23-25), [1] = values,
 * [2..2+nidx) = int64 index tensors. For each element n of the iteration space:
 *   self_bytes[ sum_i wrap(idx_i[n], sizes[i]) * strides_bytes[i] ] = values[n]
 * Negative indices wrap once; no bounds check; duplicates: last writer wins, unspecified order
 * (tensor_index.h:56-75).
 */
int kf_index_put(const kf_iter_desc *desc, int nidx, const int64_t *sizes, const int64_t *strides_bytes,
                 void *stream);

/* ---- row gather: the reference's roadmap item "embedding" (README.md:30), on the index arithmetic of tensor_index.h:56-104 ---- */
/*
 * out[n, :] = table[wrap(idx[n]), :] for n < n: rows of row_bytes bytes, moved as bytes (bit-exact for every dtype); negative
 * indices wrap once, no bounds check (as the reference's index kernels, tensor_index.h:56-75). idx is int64, on the device.
 */
int kf_index_get(const void *table, int64_t nrows, int64_t row_bytes, const int64_t *idx, int64_t n, void *out, void *stream);
/*
 * The gather's backward without atomics: dst[r, :] = sum over {n : wrap(idx[n]) == r} of src[n, :], added in input order in f32
 * (the indices are wrapped, stably sorted with kf_sort, and each run of equal rows is summed by one wave: bitwise
 * reproducible). Rows no index names are left untouched (zero dst first). An index outside [-nrows, nrows) names no row and its
 * src row is DROPPED (no out-of-bounds write, no error: the call never synchronises). dtype in {KF_F32, KF_BF16, KF_F16}; src [n, cols],
 * dst [nrows, cols] contiguous; caller scratch of kf_index_add_workspace_bytes(n) bytes, no initialisation needed.
 */
size_t kf_index_add_workspace_bytes(int64_t n);
int kf_index_add(int dtype, const int64_t *idx, int64_t n, const void *src, int64_t cols, int64_t nrows, void *dst, void *workspace,
                 size_t workspace_bytes, void *stream);

/* ---- sort: replaces sort_ops_kernel.h (segmented_sort_pairs, sort_ops_kernel.cu:402-505) ----- */
/*
 * Stable sort of nseg contiguous segments of n keys each (what sort_stable_kernel, sort_ops_kernel.cu:556-618, hands to
 * segmented_sort_pairs<scalar_t, int64_t>(keys_in, keys_out, nullptr, values_out, nsegments, nsort, descending)):
 * keys_out[s][j] = the j-th key of segment s in ascending (descending != 0: descending) order, pos_out[s][j] = its
 * int64 position inside the input segment; equal keys keep their input order in both directions. Order is that of the
 * reference's key transforms (sorting_common.h:23-260): integers by value; floats by value with -0.0 before +0.0 and
 * NaNs by bit pattern (positive NaNs last, negative NaNs first, ascending). Every dtype except KF_BOOL; n <= INT_MAX.
 * keys_in and keys_out must be different buffers. Segments of at most 8192 keys are sorted in LDS and need no
 * workspace; longer ones take kf_sort_workspace_bytes() of caller scratch (16-byte aligned, need not be zeroed).
 */
size_t kf_sort_workspace_bytes(int dtype, int64_t nseg, int64_t n);
int kf_sort(int dtype, const void *keys_in, void *keys_out, int64_t *pos_out, int64_t nseg, int64_t n, int descending,
            void *workspace, size_t workspace_bytes, void *stream);

/* ---- GEMM: replaces gemm_kernel.h:5 (+ NT/TN forms the backward needs) --------------------- */
enum {
    KF_EPI_NONE = 0,
    KF_EPI_BIAS_ROW = 1 /* C[m,n] += bias[n] after alpha/beta (fused broadcast add, config C4) */
};
/*
 * C[M,N] = alpha * op(A)[M,K] * op(B)[K,N] + beta * C, row-major, leading dims in ELEMENTS.
 *   trans_a == 0: A is stored [M,K] (lda >= K);  trans_a == 1: A is stored [K,M] (lda >= M)
 *   trans_b == 0: B is stored [K,N] (ldb >= N);  trans_b == 1: B is stored [N,K] (ldb >= K)
 * dtype in {KF_F32, KF_F64, KF_F16, KF_BF16}; A, B, C share it; accumulation is f32 (f64 for f64).
 * beta == 0 never reads C (the reference reads uninitialised memory there, gemm_ops.cpp:10-16).
 * Every kernel reads every operand layout in place, so no GEMM NEEDS scratch. kf_gemm_workspace_bytes() is non-zero only for
 * skinny 16-bit products (at most 128 output tiles of 128 x 128, at least 16 K tiles of 64): given that much 16-byte aligned
 * scratch, kf_gemm splits the contraction into 2..16 slices whose f32 partial tiles a second kernel adds in slice order
 * (deterministic; M 256, N 4096, K 16384: 169 -> 49 us); with NULL / 0 the same call runs unsplit. KF_GEMM_NO_SPLITK disables it.
 * Ragged extents (round 5): a 16-bit or f32 product of at least 2^24 multiply-adds whose M / N / K are not whole tiles (128 x 128 x 64,
 * f32 64 x 64 x 16) also reports scratch: given it, kf_gemm copies the ragged operands into zero-padded images of whole tiles, runs the
 * tile kernels on those and copies the valid part of C back (bf16 4000^3: 5.0 -> 0.11 ms; 16 x 8192 x 8192: 0.70 -> 0.05 ms); without it
 * (or with KF_GEMM_NO_PAD, which also makes the query return 0) the scalar kernel runs as before. Bytes of C outside [M, N] stay untouched.
 * The scratch is three padded images (+ split-K partials): ~2 (Mp Kp + Kp Np + Mp Np) bytes in 16 bits - gigabytes for vocabulary-sized
 * products; a workspace that is too small or not 16-byte aligned is not an error: the scalar kernel runs and the library says so once on stderr.
 */
int kf_gemm_workspace_bytes(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, size_t *bytes);
int kf_gemm(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
            int64_t lda, const void *B, int64_t ldb, float beta, void *C, int64_t ldc, int epilogue,
            const void *bias, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same product with an element-wise tail fused into the store (the reference's roadmap: fused projections, README.md:32; its
 * own API can only express these as separate add / mul kernels over the GEMM's output, binary_ops.cpp:6-91):
 *     t = alpha * op(A) op(B) + beta * C + bias[n]        (bias NULL: none)
 *     aux[m,n] = t                                        (aux NULL: not kept; the backward of a gating product needs it)
 *     C[m,n]  = t * mul[m,n] + add[m,n]                   (mul NULL: 1; add NULL: 0)
 * mul, add, aux: [M,N] row-major of the GEMM's dtype with their own leading dimensions. A residual connection is add = the
 * residual stream; a gated MLP's h = (x Wg) o (x Wu) is mul = x Wu with aux = x Wg. Every kernel applies the tail.
 */
typedef struct kf_gemm_epilogue {
    const void *bias; /* [N] or NULL */
    const void *mul;  /* [M,N] or NULL */
    int64_t ldmul;
    const void *add;  /* [M,N] or NULL */
    int64_t ldadd;
    void *aux;        /* [M,N] or NULL */
    int64_t ldaux;
    int32_t c_f32;    /* ABI 6, 16-bit dtypes only: C (and beta C) is FLOAT [M, ldc floats per row]; the f32 accumulators leave unrounded -
                         a weight gradient summed over ranks or micro-batches (beta = 1) loses nothing to the 16-bit format. bias / mul / add /
                         aux stay 16-bit. 0: C has the operands' dtype. */
} kf_gemm_epilogue;
int kf_gemm_ex(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda,
               const void *B, int64_t ldb, float beta, void *C, int64_t ldc, const kf_gemm_epilogue *epi, void *stream);

/*
 * Several independent products in one call (no element-wise tail). The backward pair of a linear layer - p[0] = dA = dC B^T
 * (trans_a 0, trans_b 1), p[1] = dB = A^T dC (trans_a 1, trans_b 0) - in 16 bits on shapes of the 4-wave 256-tile kernel
 * (M, N multiples of 256, K of 64, at most 512 tiles each, tile counts multiples of 8) runs as ONE grid: the second product's
 * first tiles start under the first one's last tiles instead of behind a kernel boundary. Anything else is the same as calling
 * kf_gemm once per problem, in order. KF_GEMM_NO_GROUP disables the single-grid form.
 */
typedef struct kf_gemm_problem {
    int32_t trans_a, trans_b;
    int64_t M, N, K;
    float alpha, beta;
    const void *A;
    int64_t lda;
    const void *B;
    int64_t ldb;
    void *C;
    int64_t ldc;
    int32_t c_f32;    /* ABI 6: this product's C is float (see kf_gemm_epilogue.c_f32) */
} kf_gemm_problem;
int kf_gemm_grouped(int dtype, int count, const kf_gemm_problem *problems, void *stream);
/* 1 when kf_gemm_grouped would run these problems as ONE grid (the backward pair of a 16-bit linear layer on 256-tile shapes with
 * few enough tiles), 0 when it would fall back to one kf_gemm per problem WITHOUT a workspace - a caller that owns split-K scratch
 * (kf_gemm_workspace_bytes) should then issue the kf_gemm calls itself so that skinny products are still split. */
int kf_gemm_grouped_single_grid(int dtype, int count, const kf_gemm_problem *p);

/* ---- causal attention: replaces causal_attention_kernel.h:5 (+ backward) ------------------- */
/*
 * q:[B,H,Sq,D], k,v:[B,H,Skv,D], o:[B,H,Sq,D] contiguous; lse:[B,H,Sq] float32 (may be NULL for
 * inference): lse = m + log(l), the natural-log-sum-exp of the scaled, masked scores.
 * O = softmax(mask(Q K^T / sqrt(D))) V, mask keeps key n for query m iff m >= n (absolute
 * indices, top-left aligned: causal_attention_ref.h:36-41).
 * dtype in {KF_F32, KF_BF16, KF_F16}, D <= 256, any Sq / Skv. Which kernels run, and what that costs (bf16, B 8 H 32 D 128 on MI355X; profiles/r06_attn_ragged.txt):
 *   tier 1  16-bit tensors, D = 64 or 128, Skv >= Sq, ANY lengths (round 6): the generated one-wave-per-SIMD streams (forward 256 queries per
 *           block, dK/dV 256 keys per block) + the stored-dS dQ kernel. Rows beyond a tensor's end are zero-filled / dropped by the kernels'
 *           buffer descriptors: no padded copies, no alignment of S to anything. S = 4096: fwd 1.06, dK/dV 1.83, dQ 0.88 ms;
 *           S = 4000: the same per token (1.004 x); S = 4095 / 3969: 1.007 x. The backward needs workspace for at least one pair's dS here.
 *   tier 2  16-bit, D = 64 or 128, Sq and Skv multiples of 128, Skv < Sq (or KF_ATTN_FWD_V3 / KF_ATTN_DKV_V4): the 8-wave forward
 *           and the 32-key-per-wave dK/dV hand kernels: 1.25 x / 1.22 x tier 1's time at the same shape.
 *   tier 3  f32 tensors (the reference's dtype), D = 64 or 128, Sq and Skv multiples of 32: exact-f32 MFMA (C3 in f32: forward 8.8 ms).
 *   tier 4  everything else (D <= 256 off 64 / 128, 16-bit Skv < Sq off the 128-row tiles, f32 off 32 rows): generic vector-ALU kernels,
 *           correct, 20-100x slower. kfunca_amd's causal_attention zero-pads the head size (and f32 row counts) on its side to stay above this tier.
 * The *_scaled forms take the softmax scale explicitly instead of 1 / sqrt(D): a host that zero-pads a smaller head
 * size up to 64 or 128 columns (zero columns change neither Q K^T nor P V) passes 1 / sqrt(its own D) and lands on the MFMA
 * kernels - kfunca_amd's causal_attention does exactly that.
 */
int kf_attn_fwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q,
                const void *k, const void *v, void *o, float *lse, void *stream);
int kf_attn_fwd_scaled(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                       const void *k, const void *v, void *o, float *lse, void *stream);
/*
 * dq,dk,dv from d_o. Needs o and lse from the forward. The workspace holds three f32 rows of statistics (delta[B,H,Sq] and the two
 * row-constant arrays the dK/dV kernel reads, the latter two with Sq rounded up to 32 rows per pair: B*H*(Sq + 2 ceil32(Sq))*4 bytes, each
 * array rounded up to 256 - the MINIMUM, O(B H S)) and, on the
 * matrix-core path for 16-bit tensors, whatever lies beyond them holds dS = P o (dP - delta) in 16 bits, one 128-KiB square per (256-query
 * block, 256-key block): the dK/dV kernel writes it, the dQ kernel computes dQ = scale dS K from it, so the backward executes the 5
 * matrix products of the algorithm instead of 7. Two layouts, picked by what the workspace holds (round 6): FULL ROWS - nq x nk squares per
 * (batch, head) pair, nq = ceil(Sq / 256), nk = ceil(Skv / 256): 32 MiB at S = 4096 - when there is room for every pair (the dQ kernel
 * streams 3-12 % faster from rows on 2 MiB boundaries), else THE CAUSAL HALF - only the squares at or below a row's diagonal, nq (nq + 1) / 2
 * when Sq = Skv: 17 MiB at S = 4096 - for as many pairs at a time as fit. The pairs are processed in GROUPS of as many as the workspace holds
 * dS for, so ANY workspace_bytes >= the minimum is accepted, for both head sizes and every S: with room for less than one pair's dS
 * (or with KF_ATTN_SPLIT_BWD set) the dQ kernel recomputes S and dP instead (minimum workspace, 7 products).
 * kf_attn_bwd_workspace_bytes() RECOMMENDS minimum + full rows for all pairs while that is within KF_ATTN_DS_CAP_MB MiB (default 16384), else
 * minimum + the causal half of as many pairs as the cap holds; with KF_ATTN_DS_TRI=1 always the causal half (config C3: 4.25 GiB instead of 8):
 * bounded whatever the problem size. No initialisation needed. Results depend neither on the group size nor on the layout (bit-identical).
 * No atomics in either form: dq, dk, dv are bitwise reproducible run to run.
 */
int kf_attn_bwd_workspace_bytes(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D,
                                size_t *bytes);
int kf_attn_bwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q,
                const void *k, const void *v, const void *o, const float *lse, const void *d_o, void *dq,
                void *dk, void *dv, void *workspace, size_t workspace_bytes, void *stream);
int kf_attn_bwd_scaled(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                       const void *k, const void *v, const void *o, const float *lse, const void *d_o, void *dq,
                       void *dk, void *dv, void *workspace, size_t workspace_bytes, void *stream);

/*
 * The same two entries for tensors that are NOT contiguous [B,H,S,D]: each operand comes with the ELEMENT strides of its batch,
 * head and row dims (the head dim D itself is contiguous). This is what removes the layout copies around attention in a
 * transformer block (the reference's roadmap item qkv_linear, README.md:32): q, k, v are read in place from the packed
 * [B*S, 3*H*D] output of the QKV projection (batch = S*3*H*D, head = D, row = 3*H*D, bases offset by H*D), o is written
 * as [B*S, H*D] - the layout the output projection consumes - and the backward writes dq, dk, dv straight into a packed
 * [B*S, 3*H*D] gradient. Contiguous [B,H,S,D] is {H*S*D, S*D, D}. lse stays [B,H,Sq] contiguous f32.
 * 16-bit matrix-core path only (dtype KF_BF16 / KF_F16, D = 64 or 128; Skv >= Sq with any lengths, or Sq and Skv multiples of 128;
 * KF_ERR_UNSUPPORTED otherwise: make contiguous copies and call the plain entries). Strides are multiples of 8 elements, operands 16-byte aligned.
 * Workspace as kf_attn_bwd_workspace_bytes().
 */
typedef struct kf_attn_layout {
    int64_t batch, head, row; /* element strides of dims B, H, S */
} kf_attn_layout;
int kf_attn_fwd_strided(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                        const kf_attn_layout *lq, const void *k, const kf_attn_layout *lk, const void *v, const kf_attn_layout *lv,
                        void *o, const kf_attn_layout *lo, float *lse, void *stream);
int kf_attn_bwd_strided(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, float scale, const void *q,
                        const kf_attn_layout *lq, const void *k, const kf_attn_layout *lk, const void *v, const kf_attn_layout *lv,
                        const void *o, const kf_attn_layout *lo, const float *lse, const void *d_o, const kf_attn_layout *ldo, void *dq,
                        const kf_attn_layout *ldq, void *dk, const kf_attn_layout *ldk, void *dv, const kf_attn_layout *ldv,
                        void *workspace, size_t workspace_bytes, void *stream);

/* ---- collectives (RCCL over xGMI): the one exchange step of the batch-sharded path (§8e) ---- */
#define KF_COMM_ID_BYTES 128
int kf_comm_unique_id(char id[KF_COMM_ID_BYTES]);
int kf_comm_init(void **comm, const char id[KF_COMM_ID_BYTES], int rank, int world_size);
int kf_comm_destroy(void *comm);
/* in-place sum all-reduce of `count` elements of dtype (KF_F32, KF_BF16, KF_F16, KF_F64, ints) */
int kf_allreduce_sum(void *comm, void *buf, size_t count, int dtype, void *stream);
/* the same over n buffers as ONE collective launch (an RCCL group): the tensors of all_reduce_(list) that are not one flat range */
int kf_allreduce_sum_multi(void *comm, int n, void *const *bufs, const size_t *counts, int dtype, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* KFUNCA_HIP_H_ */
