/*
 * oracle.h — CPU restatement of the reference's tensor-kernel hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library; it is
 * the checker, never the product: nothing under kfunca_amd/ links, imports or calls it.
 *
 * Every function restates reference semantics at TENSOR level (shapes, element strides, dtypes),
 * not at iterator level, so it checks the host TensorIterator as well as the kernels. The reference
 * has no CPU path and cannot be built here (nvcc + un-vendored CUTLASS): PARITY
 * UNPINNED - nothing here has been compared with an output of kfunca itself. The oracle is checked
 * against the reference's OWN test oracle — the numpy / torch-CPU expressions in test/test_tensor.py,
 * test/test_gemm.py and test/test_nn.py — through the committed fixtures in tests/golden
 * (tests/test_oracle.py). Backward of GEMM / attention, the norms and the gather have no reference
 * counterpart (README.md:28-32): those are checked against torch-CPU autograd fixtures.
 */
#ifndef KFUNCA_ORACLE_H_
#define KFUNCA_ORACLE_H_
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* dtype codes: reference ScalarType order, src/core/include/scalar_type.h:9-27 */
enum { ORC_BOOL = 0, ORC_U8, ORC_I8, ORC_I16, ORC_I32, ORC_I64, ORC_F16, ORC_BF16, ORC_F32, ORC_F64 };
enum { ORC_ADD = 0, ORC_SUB, ORC_MUL, ORC_DIV };
enum { ORC_SUM = 0, ORC_MEAN };
#define ORC_MAX_DIMS 12

typedef struct orc_tensor {
    void *data;
    int32_t dtype;
    int32_t ndim;
    int64_t shape[ORC_MAX_DIMS];
    int64_t stride[ORC_MAX_DIMS]; /* in ELEMENTS, as Tensor::stride() (tensor_impl.h:99-110) */
} orc_tensor;

/* tensor_iterator.cpp:32-44 update_common_dtype */
int orc_promote(int a, int b);
/* binary_ops.cpp:6-91 + binary_ops_kernel.cu:6-60; same-ndim broadcasting tensor_iterator.cpp:110-128 */
int orc_binary(int op, const orc_tensor *a, const orc_tensor *b, orc_tensor *out);
/* unary_ops.cpp:7-24 copy_/convert/clone: dst = static_cast<dst dtype>(src), src broadcast to dst */
int orc_copy(const orc_tensor *src, orc_tensor *dst);
/* nullary_ops.cpp:6-14 + nullary_ops_kernel.cu:20-25 */
int orc_fill(orc_tensor *dst, double value);
/* reduce_ops.cpp:8-20 + reduce_ops_kernel.cu:6-59; out has shape[dim] == 1 (keepdim) */
int orc_reduce(int op, const orc_tensor *in, int dim, orc_tensor *out);
/* index_ops.cpp:6-38 + tensor_index.h:56-75; idx[i] are int64 tensors of identical shape; values same shape */
int orc_moments(int mode, const orc_tensor *in, int dim, double correction, double eps, orc_tensor *out0, orc_tensor *out1);
int orc_index_put(orc_tensor *self, int nidx, const orc_tensor *idx, const orc_tensor *values);
/* gemm_kernel.cu:8-38 with the in-tree statement of the arithmetic, block_utils.h:46-77 (fma_dot_ref):
 * C = alpha * op(A) op(B) + beta * C, row-major, k-ordered fma chain per output element */
int orc_gemm(int dtype, int trans_a, int trans_b, int64_t M, int64_t N, int64_t K, float alpha, const void *A,
             int64_t lda, const void *B, int64_t ldb, float beta, void *C, int64_t ldc, const void *bias_row);
/* causal_attention_ref.h:25-64; lse[b,h,m] = out_m + log(out_l) (may be NULL) */
int orc_attn_fwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k,
                 const void *v, void *o, float *lse);
/* closed-form softmax-Jacobian backward of the above (no reference counterpart) */
int orc_attn_bwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k,
                 const void *v, const void *d_o, void *dq, void *dk, void *dv);
/* the same in double with nothing rounded, plus each output element's error scales - m*: sum of |terms| (worst case), q*: the terms in
 * quadrature (statistical), bdq: the worst case of a rounded delta through dq (oracle.c) - the reference the 16-bit kernels' scale-aware
 * bounds are taken against (oracle/checks.py). Output pointers may be NULL; d_o == NULL: forward half only */
int orc_attn_ref64(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k, const void *v,
                   const void *d_o, double *o, double *lse, double *mo, double *dq, double *dk, double *dv, double *mdq, double *mdk,
                   double *mdv, double *bdq, double *qo, double *qdq, double *qdk, double *qdv);
/* rms_norm / layer_norm rows (README.md:28 roadmap item; invstd = 1 / sqrt(M2 / n + eps) as welford_norm.h:170-187); double
 * arithmetic on the dtype-rounded inputs, x / y / dy / dx contiguous [rows, cols]; kind 0 = rms, 1 = layer */
int orc_norm_fwd(int kind, int dtype, int64_t rows, int64_t cols, const void *x, const void *w, const void *b, double eps, void *y,
                 float *mean_out, float *rstd_out);
int orc_norm_bwd(int kind, int dtype, int64_t rows, int64_t cols, const void *x, const void *w, double eps, const void *dy, void *dx,
                 void *dw, void *db);
/* embedding gather (README.md:30): out[n,:] = table[wrap(idx[n]),:], byte copy of rows */
int orc_index_get(const void *table, int64_t nrows, int64_t row_bytes, const int64_t *idx, int64_t n, void *out);
/* half.h:150-208 conversions, exposed for the fixtures */
float orc_bf16_to_f32(uint16_t v);
uint16_t orc_f32_to_bf16(float f);
float orc_f16_to_f32(uint16_t v);
uint16_t orc_f32_to_f16(float f);
int orc_num_threads(void);

#ifdef __cplusplus
}
#endif
#endif
