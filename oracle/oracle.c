/*
 * oracle.c — CPU restatement of the reference's tensor-kernel hot path.  TEST INFRASTRUCTURE ONLY
 * (see oracle.h: only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it).
 *
 * Parity status: PARITY UNPINNED. No output of kfunca itself has been compared with this file: the reference is
 * unbuildable here (every op bottoms out in nvcc-only .cu files and an un-vendored CUTLASS) and its tests hold no golden
 * vectors. What this file IS checked against (tests/test_oracle.py, tests/golden/*.npz) is the reference's own TEST oracle:
 * the numpy / torch-CPU expressions of test/test_tensor.py, test_gemm.py, test_nn.py evaluated on seeded inputs, and
 * torch-CPU autograd / f64 numpy for everything the reference has no counterpart of (backward passes, bf16, norms, gather).
 * Citations are relative to /root/reference.
 */
#include "oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* ---- scalar types: src/core/include/half.h:150-208 ------------------------------------------ */
float orc_bf16_to_f32(uint16_t v) {
    uint32_t u = (uint32_t)v << 16;
    float f;
    memcpy(&f, &u, 4);
    return f;
}
uint16_t orc_f32_to_bf16(float f) { /* round-to-nearest-even; NaN -> 0x7FC0 (half.h:195-208) */
    if (isnan(f)) return 0x7FC0;
    uint32_t u;
    memcpy(&u, &f, 4);
    return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16);
}
float orc_f16_to_f32(uint16_t h) { /* IEEE binary16 -> binary32, exact (half.h:24-147) */
    const uint32_t sign = (uint32_t)(h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1F, m = h & 0x3FFu, u;
    if (e == 0) {
        if (m == 0) {
            u = sign;
        } else { /* subnormal: renormalise */
            int sh = 0;
            while (!(m & 0x400u)) { m <<= 1; ++sh; }
            m &= 0x3FFu;
            u = sign | ((uint32_t)(113 - sh) << 23) | (m << 13);
        }
    } else if (e == 31) {
        u = sign | 0x7F800000u | (m << 13);
    } else {
        u = sign | ((e + 112) << 23) | (m << 13);
    }
    float f;
    memcpy(&f, &u, 4);
    return f;
}
uint16_t orc_f32_to_f16(float f) { /* binary32 -> binary16, round-to-nearest-even (half.h:150-182) */
    uint32_t u;
    memcpy(&u, &f, 4);
    const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
    const uint32_t a = u & 0x7FFFFFFFu;
    if (a > 0x7F800000u) return (uint16_t)(sign | 0x7E00u);
    if (a >= 0x47800000u) return (uint16_t)(sign | 0x7C00u); /* >= 65536 (incl. inf) -> inf */
    if (a < 0x33000001u) return sign;                         /* <= 2^-25 -> 0 (ties to even) */
    int e = (int)(a >> 23) - 127;
    uint32_t m = (a & 0x7FFFFFu) | 0x800000u;
    uint32_t shift, half_e;
    if (e < -14) { shift = (uint32_t)(13 + (-14 - e)); half_e = 0; } else { shift = 13; half_e = (uint32_t)(e + 15); }
    uint32_t q = m >> shift, rem = m & ((1u << shift) - 1u), halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (q & 1u))) ++q;
    uint32_t r = half_e == 0 ? q : ((half_e - 1) << 10) + q; /* q carries the implicit bit: adds 1 to the exponent */
    if (r >= 0x7C00u) r = 0x7C00u;
    return (uint16_t)(sign | r);
}

static int dt_size(int dt) {
    switch (dt) {
    case ORC_BOOL: case ORC_U8: case ORC_I8: return 1;
    case ORC_I16: case ORC_F16: case ORC_BF16: return 2;
    case ORC_I32: case ORC_F32: return 4;
    default: return 8;
    }
}
/* accumulate class (accumulate_type.h:17-27): 0 float, 1 double, 2 int64, 3 bool */
static int acc_class(int dt) {
    switch (dt) {
    case ORC_F16: case ORC_BF16: case ORC_F32: return 0;
    case ORC_F64: return 1;
    case ORC_BOOL: return 3;
    default: return 2;
    }
}
static int is_float(int t) { return t == ORC_F64 || t == ORC_F32 || t == ORC_F16 || t == ORC_BF16; }
static int is_uint(int t) { return t == ORC_U8 || t == ORC_BOOL; }

int orc_promote(int a, int b) { /* tensor_iterator.cpp:32-44 */
    if (is_float(a) && is_float(b)) return a >= b ? a : b;
    if (is_float(a) || is_float(b)) return is_float(a) ? a : b;
    if (is_uint(a) && is_uint(b)) return a >= b ? a : b;
    if (is_uint(a) || is_uint(b)) return is_uint(a) ? b : a;
    return a >= b ? a : b;
}

/* fetch_and_cast<dest_t> (tensor_memory_access.h:13-24): static_cast from the stored type */
#define ORC_LOAD(NAME, TYPE)                                              \
    static TYPE NAME(int dt, const void *p) {                             \
        switch (dt) {                                                     \
        case ORC_BOOL: return (TYPE)(*(const uint8_t *)p != 0);           \
        case ORC_U8: return (TYPE)(*(const uint8_t *)p);                  \
        case ORC_I8: return (TYPE)(*(const int8_t *)p);                   \
        case ORC_I16: return (TYPE)(*(const int16_t *)p);                 \
        case ORC_I32: return (TYPE)(*(const int32_t *)p);                 \
        case ORC_I64: return (TYPE)(*(const int64_t *)p);                 \
        case ORC_F16: return (TYPE)orc_f16_to_f32(*(const uint16_t *)p);  \
        case ORC_BF16: return (TYPE)orc_bf16_to_f32(*(const uint16_t *)p);\
        case ORC_F32: return (TYPE)(*(const float *)p);                   \
        default: return (TYPE)(*(const double *)p);                       \
        }                                                                 \
    }
ORC_LOAD(ld_f, float)
ORC_LOAD(ld_d, double)
ORC_LOAD(ld_i, int64_t)
static int ld_b(int dt, const void *p) { return ld_d(dt, p) != 0.0; }

/* cast_and_store<src_t> (tensor_memory_access.h:26-37) */
#define ORC_STORE(NAME, TYPE)                                              \
    static void NAME(int dt, void *p, TYPE v) {                            \
        switch (dt) {                                                      \
        case ORC_BOOL: *(uint8_t *)p = (uint8_t)(v != 0); break;           \
        case ORC_U8: *(uint8_t *)p = (uint8_t)v; break;                    \
        case ORC_I8: *(int8_t *)p = (int8_t)v; break;                      \
        case ORC_I16: *(int16_t *)p = (int16_t)v; break;                   \
        case ORC_I32: *(int32_t *)p = (int32_t)v; break;                   \
        case ORC_I64: *(int64_t *)p = (int64_t)v; break;                   \
        case ORC_F16: *(uint16_t *)p = orc_f32_to_f16((float)v); break;    \
        case ORC_BF16: *(uint16_t *)p = orc_f32_to_bf16((float)v); break;  \
        case ORC_F32: *(float *)p = (float)v; break;                       \
        default: *(double *)p = (double)v; break;                          \
        }                                                                  \
    }
ORC_STORE(st_f, float)
ORC_STORE(st_d, double)
ORC_STORE(st_i, int64_t)

/* ---- index walking --------------------------------------------------------------------------- */
static int64_t numel_of(const orc_tensor *t) {
    int64_t n = 1;
    for (int i = 0; i < t->ndim; ++i) n *= t->shape[i];
    return n;
}
/* element offset of operand `t` for the multi-index `idx` of an iteration space `shape`;
 * size-1 dims of the operand broadcast (tensor_iterator.cpp:148-162) */
static int64_t off_of(const orc_tensor *t, const int64_t *idx, const int64_t *shape) {
    int64_t o = 0;
    for (int i = 0; i < t->ndim; ++i)
        if (!(t->shape[i] == 1 && shape[i] != 1)) o += idx[i] * t->stride[i];
    return o;
}
static void unravel(int64_t lin, int ndim, const int64_t *shape, int64_t *idx) {
    for (int i = ndim - 1; i >= 0; --i) {
        idx[i] = shape[i] ? lin % shape[i] : 0;
        lin = shape[i] ? lin / shape[i] : 0;
    }
}

/* ---- elementwise ----------------------------------------------------------------------------- */
int orc_binary(int op, const orc_tensor *a, const orc_tensor *b, orc_tensor *out) {
    if (a->ndim != b->ndim || a->ndim != out->ndim) return 1; /* same-ndim rule, tensor_iterator.cpp:16-30 */
    int64_t shape[ORC_MAX_DIMS];
    for (int i = 0; i < a->ndim; ++i) {
        const int64_t x = a->shape[i], y = b->shape[i];
        if (!(x == y || x == 1 || y == 1)) return 2;
        shape[i] = x == 1 ? y : x;
        if (out->shape[i] != shape[i]) return 3;
    }
    const int common = orc_promote(a->dtype, b->dtype), cls = acc_class(common);
    const int64_t n = numel_of(out);
    const int sa = dt_size(a->dtype), sb = dt_size(b->dtype), so = dt_size(out->dtype);
#pragma omp parallel for schedule(static) if (n > (1 << 16))
    for (int64_t lin = 0; lin < n; ++lin) {
        int64_t idx[ORC_MAX_DIMS];
        unravel(lin, out->ndim, shape, idx);
        const char *pa = (const char *)a->data + off_of(a, idx, shape) * sa;
        const char *pb = (const char *)b->data + off_of(b, idx, shape) * sb;
        char *po = (char *)out->data + off_of(out, idx, shape) * so;
        if (cls == 0) {
            const float x = ld_f(a->dtype, pa), y = ld_f(b->dtype, pb);
            st_f(out->dtype, po, op == ORC_ADD ? x + y : op == ORC_SUB ? x - y : op == ORC_MUL ? x * y : x / y);
        } else if (cls == 1) {
            const double x = ld_d(a->dtype, pa), y = ld_d(b->dtype, pb);
            st_d(out->dtype, po, op == ORC_ADD ? x + y : op == ORC_SUB ? x - y : op == ORC_MUL ? x * y : x / y);
        } else if (cls == 2) {
            const int64_t x = ld_i(a->dtype, pa), y = ld_i(b->dtype, pb);
            int64_t r;
            if (op == ORC_ADD) r = (int64_t)((uint64_t)x + (uint64_t)y);
            else if (op == ORC_SUB) r = (int64_t)((uint64_t)x - (uint64_t)y);
            else if (op == ORC_MUL) r = (int64_t)((uint64_t)x * (uint64_t)y);
            else r = y == 0 ? 0 : (y == -1 ? (int64_t)(0 - (uint64_t)x) : x / y); /* x/0: UB in the reference */
            st_i(out->dtype, po, r);
        } else {
            const int x = ld_b(a->dtype, pa), y = ld_b(b->dtype, pb);
            st_i(out->dtype, po, op == ORC_ADD ? (x || y) : op == ORC_SUB ? (x != y) : op == ORC_MUL ? (x && y) : x);
        }
    }
    return 0;
}

int orc_copy(const orc_tensor *src, orc_tensor *dst) {
    if (src->ndim != dst->ndim) return 1;
    for (int i = 0; i < src->ndim; ++i)
        if (!(src->shape[i] == dst->shape[i] || src->shape[i] == 1)) return 2;
    const int64_t n = numel_of(dst);
    const int ss = dt_size(src->dtype), sd = dt_size(dst->dtype), cls = acc_class(dst->dtype);
#pragma omp parallel for schedule(static) if (n > (1 << 16))
    for (int64_t lin = 0; lin < n; ++lin) {
        int64_t idx[ORC_MAX_DIMS];
        unravel(lin, dst->ndim, dst->shape, idx);
        const char *ps = (const char *)src->data + off_of(src, idx, dst->shape) * ss;
        char *pd = (char *)dst->data + off_of(dst, idx, dst->shape) * sd;
        if (src->dtype == dst->dtype) memcpy(pd, ps, (size_t)sd); /* CopyFunctor<scalar_t>: bit copy */
        else if (cls == 0) st_f(dst->dtype, pd, ld_f(src->dtype, ps));
        else if (cls == 1) st_d(dst->dtype, pd, ld_d(src->dtype, ps));
        else if (cls == 2) st_i(dst->dtype, pd, ld_i(src->dtype, ps));
        else st_i(dst->dtype, pd, ld_b(src->dtype, ps));
    }
    return 0;
}

int orc_fill(orc_tensor *dst, double value) { /* FillFunctor<acc_t>(double) then cast to out dtype */
    const int64_t n = numel_of(dst);
    const int sd = dt_size(dst->dtype), cls = acc_class(dst->dtype);
    for (int64_t lin = 0; lin < n; ++lin) {
        int64_t idx[ORC_MAX_DIMS];
        unravel(lin, dst->ndim, dst->shape, idx);
        char *pd = (char *)dst->data + off_of(dst, idx, dst->shape) * sd;
        if (cls == 0) st_f(dst->dtype, pd, (float)value);
        else if (cls == 1) st_d(dst->dtype, pd, value);
        else if (cls == 2) st_i(dst->dtype, pd, (int64_t)value);
        else st_i(dst->dtype, pd, value != 0.0);
    }
    return 0;
}

/* ---- reductions ------------------------------------------------------------------------------- */
/* The reference's order of summation is an implementation detail of its GPU reduce tree
 * (tensor_reduce.h:454-923) and its tests allow 1e-2 (test_tensor.py:118); the oracle therefore
 * states the correctly rounded result: floating sums accumulate in double and are cast once.
 * Integer sums wrap exactly as in-dtype accumulation does. mean = sum * factor with
 * factor = static_cast<scalar_t>(nout) / numel evaluated in the input dtype
 * (reduce_ops_kernel.cu:49-53) — an integer quotient for integer dtypes. */
int orc_reduce(int op, const orc_tensor *in, int dim, orc_tensor *out) {
    if (in->ndim != out->ndim || dim < 0 || dim >= in->ndim) return 1;
    for (int i = 0; i < in->ndim; ++i)
        if (out->shape[i] != (i == dim ? 1 : in->shape[i])) return 2;
    if (in->dtype != out->dtype) return 3;
    const int64_t R = in->shape[dim], nout = numel_of(out), numel = nout * R;
    const int si = dt_size(in->dtype), so = dt_size(out->dtype), cls = acc_class(in->dtype);
#pragma omp parallel for schedule(static) if (numel > (1 << 16))
    for (int64_t lin = 0; lin < nout; ++lin) {
        int64_t idx[ORC_MAX_DIMS];
        unravel(lin, out->ndim, out->shape, idx);
        int64_t base = 0;
        for (int i = 0; i < in->ndim; ++i) base += idx[i] * in->stride[i];
        char *po = (char *)out->data + off_of(out, idx, out->shape) * so;
        if (cls == 0 || cls == 1) {
            double acc = 0.0;
            for (int64_t r = 0; r < R; ++r) acc += ld_d(in->dtype, (const char *)in->data + (base + r * in->stride[dim]) * si);
            if (op == ORC_MEAN) {
                if (cls == 1) acc *= (double)nout / (double)numel;
                else acc *= (double)((float)nout / (float)numel);
            }
            st_d(out->dtype, po, acc);
        } else {
            int64_t acc = 0;
            for (int64_t r = 0; r < R; ++r) {
                const char *p = (const char *)in->data + (base + r * in->stride[dim]) * si;
                acc = (int64_t)((uint64_t)acc + (uint64_t)(cls == 3 ? ld_b(in->dtype, p) : ld_i(in->dtype, p)));
            }
            if (op == ORC_MEAN) acc = (int64_t)((uint64_t)acc * (uint64_t)(numel ? nout / numel : 0));
            st_i(out->dtype, po, acc);
        }
    }
    return 0;
}

/* ---- moments: mean_var (reduce_ops.cpp:22-28, reduce_ops_kernel.cu:61-153) and norm_stat
 *      (norm_ops_kernel.cu:6-61, welford_norm.h:170-187) ---------------------------------------------
 * The reference's Welford recurrences compute, up to rounding, mean = sum(x)/n and M2 = sum((x-mean)^2);
 * the oracle states exactly that with a two-pass double accumulation and one final cast.
 *   mode 0: var = M2 / max(n - correction, 0)   (WelfordOps::project, correction = 1 from reduce_ops.cpp:26)
 *   mode 1: sqrt of mode 0                      (take_sqrt)
 *   mode 2: 1 / sqrt(M2 / n + eps)              (welford_norm.h:183, eps = 1e-12 from norm_ops_kernel.cu:40)
 * out0 = the variance-like value, out1 = the mean (the iterator's output order, reduce_ops.cpp:24). */
int orc_moments(int mode, const orc_tensor *in, int dim, double correction, double eps, orc_tensor *out0, orc_tensor *out1) {
    if (in->ndim != out0->ndim || in->ndim != out1->ndim || dim < 0 || dim >= in->ndim) return 1;
    for (int i = 0; i < in->ndim; ++i)
        if (out0->shape[i] != (i == dim ? 1 : in->shape[i]) || out1->shape[i] != out0->shape[i]) return 2;
    if (out0->dtype != out1->dtype) return 3;
    if (acc_class(in->dtype) > 1) return 4; /* DISPATCH_FLOATING_TYPES */
    const int64_t R = in->shape[dim], nout = numel_of(out0);
    const int si = dt_size(in->dtype), so = dt_size(out0->dtype);
#pragma omp parallel for schedule(static) if (nout * R > (1 << 16))
    for (int64_t lin = 0; lin < nout; ++lin) {
        int64_t idx[ORC_MAX_DIMS];
        unravel(lin, out0->ndim, out0->shape, idx);
        int64_t base = 0;
        for (int i = 0; i < in->ndim; ++i) base += idx[i] * in->stride[i];
        double sum = 0.0, m2 = 0.0;
        for (int64_t r = 0; r < R; ++r) sum += ld_d(in->dtype, (const char *)in->data + (base + r * in->stride[dim]) * si);
        const double mean = R ? sum / (double)R : 0.0;
        for (int64_t r = 0; r < R; ++r) {
            const double x = ld_d(in->dtype, (const char *)in->data + (base + r * in->stride[dim]) * si) - mean;
            m2 += x * x;
        }
        double v;
        if (mode == 2) {
            v = 1.0 / sqrt(m2 / (double)R + eps);
        } else {
            const double div = (double)R > correction ? (double)R - correction : 0.0;
            v = m2 / div;
            if (mode == 1) v = sqrt(v);
        }
        st_d(out0->dtype, (char *)out0->data + off_of(out0, idx, out0->shape) * so, v);
        st_d(out1->dtype, (char *)out1->data + off_of(out1, idx, out1->shape) * so, mean);
    }
    return 0;
}

/* ---- index_put_ ------------------------------------------------------------------------------- */
int orc_index_put(orc_tensor *self, int nidx, const orc_tensor *idx, const orc_tensor *values) {
    if (nidx != self->ndim) return 1; /* index_ops.cpp:7-8 */
    if (self->dtype != values->dtype) return 2;
    const int64_t n = numel_of(&idx[0]);
    const int es = dt_size(self->dtype);
    for (int64_t lin = 0; lin < n; ++lin) { /* ascending n: a later duplicate overwrites an earlier one */
        int64_t mi[ORC_MAX_DIMS];
        unravel(lin, idx[0].ndim, idx[0].shape, mi);
        int64_t target = 0;
        for (int k = 0; k < nidx; ++k) {
            if (idx[k].dtype != ORC_I64) return 3;
            int64_t v = *((const int64_t *)idx[k].data + off_of(&idx[k], mi, idx[0].shape));
            if (v < 0) v += self->shape[k]; /* tensor_index.h:66-68 */
            target += v * self->stride[k];
        }
        memcpy((char *)self->data + target * es, (const char *)values->data + off_of(values, mi, idx[0].shape) * es, (size_t)es);
    }
    return 0;
}

/* ---- GEMM -------------------------------------------------------------------------------------- */
int orc_gemm(int dtype, int ta, int tb, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda,
             const void *B, int64_t ldb, float beta, void *C, int64_t ldc, const void *bias_row) {
    if (!(dtype == ORC_F32 || dtype == ORC_F64 || dtype == ORC_F16 || dtype == ORC_BF16)) return 1;
    const int es = dt_size(dtype);
    if (dtype == ORC_F64) {
        double *Bd = (double *)malloc(sizeof(double) * (size_t)(K * N ? K * N : 1));
        for (int64_t k = 0; k < K; ++k)
            for (int64_t j = 0; j < N; ++j) Bd[k * N + j] = ((const double *)B)[tb ? j * ldb + k : k * ldb + j];
#pragma omp parallel for schedule(static)
        for (int64_t i = 0; i < M; ++i) {
            double *acc = (double *)calloc((size_t)(N ? N : 1), sizeof(double));
            for (int64_t k = 0; k < K; ++k) {
                const double a = ((const double *)A)[ta ? k * lda + i : i * lda + k];
                const double *brow = Bd + k * N;
                for (int64_t j = 0; j < N; ++j) acc[j] = fma(a, brow[j], acc[j]);
            }
            double *c = (double *)C + i * ldc;
            for (int64_t j = 0; j < N; ++j) {
                double v = (double)alpha * acc[j];
                if (beta != 0.f) v += (double)beta * c[j];
                if (bias_row) v += ((const double *)bias_row)[j];
                c[j] = v;
            }
            free(acc);
        }
        free(Bd);
        return 0;
    }
    /* float accumulate, k-ordered fma chain: sum += (float)a * (float)b (block_utils.h:60-71) */
    float *Bf = (float *)malloc(sizeof(float) * (size_t)(K * N ? K * N : 1));
#pragma omp parallel for schedule(static)
    for (int64_t k = 0; k < K; ++k)
        for (int64_t j = 0; j < N; ++j) Bf[k * N + j] = ld_f(dtype, (const char *)B + (tb ? j * ldb + k : k * ldb + j) * es);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < M; ++i) {
        float *acc = (float *)calloc((size_t)(N ? N : 1), sizeof(float));
        for (int64_t k = 0; k < K; ++k) {
            const float a = ld_f(dtype, (const char *)A + (ta ? k * lda + i : i * lda + k) * es);
            const float *brow = Bf + k * N;
            for (int64_t j = 0; j < N; ++j) acc[j] = fmaf(a, brow[j], acc[j]);
        }
        char *c = (char *)C + i * ldc * es;
        for (int64_t j = 0; j < N; ++j) {
            float v = alpha * acc[j];
            if (beta != 0.f) v += beta * ld_f(dtype, c + j * es);
            if (bias_row) v += ld_f(dtype, (const char *)bias_row + j * es);
            st_f(dtype, c + j * es, v);
        }
        free(acc);
    }
    free(Bf);
    return 0;
}

/* ---- causal attention --------------------------------------------------------------------------- */
static void to_float(int dtype, const void *src, int64_t n, float *dst) {
    const int es = dt_size(dtype);
    for (int64_t i = 0; i < n; ++i) dst[i] = ld_f(dtype, (const char *)src + i * es);
}

int orc_attn_fwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k,
                 const void *v, void *o, float *lse) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_BF16)) return 1;
    const int es = dt_size(dtype);
    const float scale = 1.0f / sqrtf((float)D); /* causal_attention_ref.h:33 */
#pragma omp parallel for schedule(dynamic)
    for (int64_t bh = 0; bh < B * H; ++bh) {
        float *qf = (float *)malloc(sizeof(float) * (size_t)(Sq * D)), *kf = (float *)malloc(sizeof(float) * (size_t)(Skv * D));
        float *vf = (float *)malloc(sizeof(float) * (size_t)(Skv * D)), *s = (float *)malloc(sizeof(float) * (size_t)Skv);
        float *orow = (float *)malloc(sizeof(float) * (size_t)D);
        to_float(dtype, (const char *)q + bh * Sq * D * es, Sq * D, qf);
        to_float(dtype, (const char *)k + bh * Skv * D * es, Skv * D, kf);
        to_float(dtype, (const char *)v + bh * Skv * D * es, Skv * D, vf);
        for (int64_t m = 0; m < Sq; ++m) {
            const int64_t nvis = m + 1 < Skv ? m + 1 : Skv; /* m >= n kept (causal_attention_ref.h:36-41) */
            float mx = -INFINITY;
            for (int64_t n = 0; n < nvis; ++n) {
                float sum = 0.f;
                for (int64_t d = 0; d < D; ++d) sum += qf[m * D + d] * kf[n * D + d];
                s[n] = sum * scale;
                if (s[n] > mx) mx = s[n];
            }
            float l = 0.f;
            for (int64_t n = 0; n < nvis; ++n) l += expf(s[n] - mx);
            for (int64_t d = 0; d < D; ++d) orow[d] = 0.f;
            for (int64_t n = 0; n < nvis; ++n) {
                const float p = expf(s[n] - mx) / l;
                for (int64_t d = 0; d < D; ++d) orow[d] += p * vf[n * D + d];
            }
            for (int64_t d = 0; d < D; ++d) st_f(dtype, (char *)o + ((bh * Sq + m) * D + d) * es, orow[d]);
            if (lse) lse[bh * Sq + m] = mx + logf(l);
        }
        free(qf); free(kf); free(vf); free(s); free(orow);
    }
    return 0;
}

int orc_attn_bwd(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k,
                 const void *v, const void *d_o, void *dq, void *dk, void *dv) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_BF16)) return 1;
    const int es = dt_size(dtype);
    const float scale = 1.0f / sqrtf((float)D);
#pragma omp parallel for schedule(dynamic)
    for (int64_t bh = 0; bh < B * H; ++bh) {
        float *qf = (float *)malloc(sizeof(float) * (size_t)(Sq * D)), *kf = (float *)malloc(sizeof(float) * (size_t)(Skv * D));
        float *vf = (float *)malloc(sizeof(float) * (size_t)(Skv * D)), *gof = (float *)malloc(sizeof(float) * (size_t)(Sq * D));
        float *dkf = (float *)calloc((size_t)(Skv * D), sizeof(float)), *dvf = (float *)calloc((size_t)(Skv * D), sizeof(float));
        float *p = (float *)malloc(sizeof(float) * (size_t)Skv), *dp = (float *)malloc(sizeof(float) * (size_t)Skv);
        float *dqrow = (float *)malloc(sizeof(float) * (size_t)D);
        to_float(dtype, (const char *)q + bh * Sq * D * es, Sq * D, qf);
        to_float(dtype, (const char *)k + bh * Skv * D * es, Skv * D, kf);
        to_float(dtype, (const char *)v + bh * Skv * D * es, Skv * D, vf);
        to_float(dtype, (const char *)d_o + bh * Sq * D * es, Sq * D, gof);
        for (int64_t m = 0; m < Sq; ++m) {
            const int64_t nvis = m + 1 < Skv ? m + 1 : Skv;
            float mx = -INFINITY;
            for (int64_t n = 0; n < nvis; ++n) {
                float sum = 0.f;
                for (int64_t d = 0; d < D; ++d) sum += qf[m * D + d] * kf[n * D + d];
                p[n] = sum * scale;
                if (p[n] > mx) mx = p[n];
            }
            float l = 0.f;
            for (int64_t n = 0; n < nvis; ++n) l += expf(p[n] - mx);
            double delta = 0.0; /* sum_n P dP == sum_d dO O */
            for (int64_t n = 0; n < nvis; ++n) {
                p[n] = expf(p[n] - mx) / l;
                float sum = 0.f;
                for (int64_t d = 0; d < D; ++d) sum += gof[m * D + d] * vf[n * D + d];
                dp[n] = sum;
                delta += (double)p[n] * (double)sum;
            }
            for (int64_t d = 0; d < D; ++d) dqrow[d] = 0.f;
            for (int64_t n = 0; n < nvis; ++n) {
                const float ds = p[n] * (dp[n] - (float)delta) * scale; /* dS / sqrt(D) */
                for (int64_t d = 0; d < D; ++d) {
                    dqrow[d] += ds * kf[n * D + d];
                    dkf[n * D + d] += ds * qf[m * D + d];
                    dvf[n * D + d] += p[n] * gof[m * D + d];
                }
            }
            for (int64_t d = 0; d < D; ++d) st_f(dtype, (char *)dq + ((bh * Sq + m) * D + d) * es, dqrow[d]);
        }
        for (int64_t i = 0; i < Skv * D; ++i) {
            st_f(dtype, (char *)dk + (bh * Skv * D + i) * es, dkf[i]);
            st_f(dtype, (char *)dv + (bh * Skv * D + i) * es, dvf[i]);
        }
        free(qf); free(kf); free(vf); free(gof); free(dkf); free(dvf); free(p); free(dp); free(dqrow);
    }
    return 0;
}

/* The same two functions in DOUBLE on the dtype-rounded inputs, nothing rounded on the way or at the end, plus, for every output element,
 * the two SCALES of the error a 16-bit kernel may make in it - such a kernel rounds P and dS to 8 (bf16) or 11 (f16) significant bits
 * before the second contraction, forms delta = rowsum(dO o O) from the ROUNDED O, and rounds each output once:
 *   m*  worst case: the sum of the absolute values of the terms the element is a sum of (every rounding pushing the same way);
 *   q*  statistical: the same terms in QUADRATURE (independent roundings add like a random walk; one term: q = m) - the scale of a row's
 *       or a head's error NORM. A row with few terms (the last keys of dK / dV, the first queries of O / dQ) has q close to m, a long
 *       row has q close to |ref| itself: the bound follows the data instead of assuming either.
 * Semantics: causal_attention_ref.h:25-64 (scale 1/sqrt(D) :33, mask keeps m >= n :36-41, max-subtracted softmax :43-58).
 *   o[m,d]  = sum_n p[m,n] v[n,d]            terms p v                  dv[n,d] = sum_m p[m,n] dO[m,d]          terms p dO
 *   dq[m,d] = scale sum_n ds[m,n] k[n,d]     terms ds k, and the delta error e[m] through scale (sum_n p k) e[m]: ONE term (coherent over n);
 *             its worst case is returned apart as bdq[m,d] = da[m] scale |sum_n p[m,n] k[n,d]| (not inside mdq)
 *   dk[n,d] = scale sum_m ds[m,n] q[m,d]     terms (ds + p e[m]) q: the delta errors of different query rows are independent
 *   ds = p (dp - delta), delta[m] = sum_n p dp = sum_d dO O;  |e[m]| <= eps da[m], da[m] = sum_d |dO[m,d] O[m,d]| (worst case),
 *   da2[m] = sqrt(sum_d (dO O)^2) (statistical)
 * Every output pointer is double[B,H,S,D] (lse: double[B,H,Sq]) and may be NULL; d_o == NULL computes the forward half only. */
int orc_attn_ref64(int dtype, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, const void *q, const void *k, const void *v,
                   const void *d_o, double *o, double *lse, double *mo, double *dq, double *dk, double *dv, double *mdq, double *mdk,
                   double *mdv, double *bdq, double *qo, double *qdq, double *qdk, double *qdv) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_BF16)) return 1;
    const int es = dt_size(dtype);
    const double scale = 1.0 / sqrt((double)D);
    for (int64_t bh = 0; bh < B * H; ++bh) {
        double *qf = (double *)malloc(sizeof(double) * (size_t)(Sq * D)), *kf = (double *)malloc(sizeof(double) * (size_t)(Skv * D));
        double *vf = (double *)malloc(sizeof(double) * (size_t)(Skv * D)), *gf = d_o ? (double *)malloc(sizeof(double) * (size_t)(Sq * D)) : NULL;
        double *rmx = (double *)malloc(sizeof(double) * (size_t)Sq), *rl = (double *)malloc(sizeof(double) * (size_t)Sq);
        double *rdelta = (double *)malloc(sizeof(double) * (size_t)Sq), *rdabs = (double *)malloc(sizeof(double) * (size_t)Sq);
        double *rda2 = (double *)malloc(sizeof(double) * (size_t)Sq);
        for (int64_t i = 0; i < Sq * D; ++i) qf[i] = ld_d(dtype, (const char *)q + (bh * Sq * D + i) * es);
        for (int64_t i = 0; i < Skv * D; ++i) kf[i] = ld_d(dtype, (const char *)k + (bh * Skv * D + i) * es);
        for (int64_t i = 0; i < Skv * D; ++i) vf[i] = ld_d(dtype, (const char *)v + (bh * Skv * D + i) * es);
        if (d_o) for (int64_t i = 0; i < Sq * D; ++i) gf[i] = ld_d(dtype, (const char *)d_o + (bh * Sq * D + i) * es);
#pragma omp parallel
        {
            double *p = (double *)malloc(sizeof(double) * (size_t)Skv), *dp = (double *)malloc(sizeof(double) * (size_t)Skv);
            double *t0 = (double *)malloc(sizeof(double) * (size_t)D * 4), *t1 = t0 + D, *t2 = t1 + D, *t3 = t2 + D;
#pragma omp for schedule(dynamic, 8)
            for (int64_t m = 0; m < Sq; ++m) { /* pass 1, a query row at a time: statistics, o, dq */
                const int64_t nvis = m + 1 < Skv ? m + 1 : Skv;
                const double *qr = qf + m * D;
                double mx = -INFINITY;
                for (int64_t n = 0; n < nvis; ++n) {
                    double sum = 0.0;
                    for (int64_t d = 0; d < D; ++d) sum += qr[d] * kf[n * D + d];
                    p[n] = sum * scale;
                    if (p[n] > mx) mx = p[n];
                }
                double l = 0.0;
                for (int64_t n = 0; n < nvis; ++n) { p[n] = exp(p[n] - mx); l += p[n]; }
                for (int64_t n = 0; n < nvis; ++n) p[n] /= l;
                rmx[m] = mx; rl[m] = l;
                if (lse) lse[bh * Sq + m] = mx + log(l);
                double *orow = t0, *morow = t1, *qorow = t2;
                for (int64_t d = 0; d < D; ++d) orow[d] = morow[d] = qorow[d] = 0.0;
                for (int64_t n = 0; n < nvis; ++n)
                    for (int64_t d = 0; d < D; ++d) {
                        const double t = p[n] * vf[n * D + d];
                        orow[d] += t; morow[d] += fabs(t); qorow[d] += t * t;
                    }
                for (int64_t d = 0; d < D; ++d) {
                    const int64_t at = (bh * Sq + m) * D + d;
                    if (o) o[at] = orow[d];
                    if (mo) mo[at] = morow[d];
                    if (qo) qo[at] = sqrt(qorow[d]);
                }
                if (!d_o) continue;
                double delta = 0.0;
                for (int64_t n = 0; n < nvis; ++n) {
                    double sum = 0.0;
                    for (int64_t d = 0; d < D; ++d) sum += gf[m * D + d] * vf[n * D + d];
                    dp[n] = sum;
                    delta += p[n] * sum;
                }
                rdelta[m] = delta;
                double dabs = 0.0, da2 = 0.0;
                for (int64_t d = 0; d < D; ++d) { const double t = gf[m * D + d] * orow[d]; dabs += fabs(t); da2 += t * t; }
                da2 = sqrt(da2);
                rdabs[m] = dabs; rda2[m] = da2;
                double *dqr = t0, *mdqr = t1, *pk = t2, *qq = t3; /* (orow is no longer needed) */
                for (int64_t d = 0; d < D; ++d) dqr[d] = mdqr[d] = pk[d] = qq[d] = 0.0;
                for (int64_t n = 0; n < nvis; ++n) {
                    const double ds = p[n] * (dp[n] - delta) * scale;
                    for (int64_t d = 0; d < D; ++d) {
                        const double t = ds * kf[n * D + d];
                        dqr[d] += t; mdqr[d] += fabs(t); qq[d] += t * t;
                        pk[d] += p[n] * kf[n * D + d]; /* signed: one delta error moves every dS of the row the same way */
                    }
                }
                for (int64_t d = 0; d < D; ++d) {
                    const int64_t at = (bh * Sq + m) * D + d;
                    const double c = fabs(pk[d]) * scale;
                    if (dq) dq[at] = dqr[d];
                    if (mdq) mdq[at] = mdqr[d];
                    if (bdq) bdq[at] = c * dabs;
                    if (qdq) qdq[at] = sqrt(qq[d] + c * da2 * c * da2);
                }
            }
            if (d_o) {
#pragma omp for schedule(dynamic, 8)
                for (int64_t n = 0; n < Skv; ++n) { /* pass 2, a key row at a time (p, dp recomputed from the saved statistics): dk, dv */
                    double *dkr = t0, *dvr = t1, *ak = t2, *av = t3;
                    for (int64_t d = 0; d < D; ++d) dkr[d] = dvr[d] = ak[d] = av[d] = 0.0;
                    double *mkr = (double *)calloc((size_t)D * 2, sizeof(double)), *mvr = mkr + D;
                    for (int64_t m = n; m < Sq; ++m) {
                        double s = 0.0, g = 0.0;
                        for (int64_t d = 0; d < D; ++d) { s += qf[m * D + d] * kf[n * D + d]; g += gf[m * D + d] * vf[n * D + d]; }
                        const double pp = exp(s * scale - rmx[m]) / rl[m], ds = pp * (g - rdelta[m]) * scale;
                        const double wk = fabs(ds) + pp * rdabs[m] * scale, w2 = ds * ds + pp * rda2[m] * scale * pp * rda2[m] * scale;
                        for (int64_t d = 0; d < D; ++d) {
                            const double qd = qf[m * D + d], gd = gf[m * D + d];
                            dkr[d] += ds * qd;
                            dvr[d] += pp * gd;
                            mkr[d] += wk * fabs(qd);
                            mvr[d] += pp * fabs(gd);
                            ak[d] += w2 * qd * qd;
                            av[d] += pp * gd * pp * gd;
                        }
                    }
                    for (int64_t d = 0; d < D; ++d) {
                        const int64_t at = (bh * Skv + n) * D + d;
                        if (dk) dk[at] = dkr[d];
                        if (dv) dv[at] = dvr[d];
                        if (mdk) mdk[at] = mkr[d];
                        if (mdv) mdv[at] = mvr[d];
                        if (qdk) qdk[at] = sqrt(ak[d]);
                        if (qdv) qdv[at] = sqrt(av[d]);
                    }
                    free(mkr);
                }
            }
            free(p); free(dp); free(t0);
        }
        free(qf); free(kf); free(vf); free(gf); free(rmx); free(rl); free(rdelta); free(rdabs); free(rda2);
    }
    return 0;
}

/* ---- row normalisations (README.md:28 roadmap rms_norm; invstd as welford_norm.h:170-187: 1 / sqrt(M2 / n + eps)) --------
 * kind 0 = rms (y = x rstd w, rstd = 1 / sqrt(mean(x^2) + eps)), 1 = layer (y = (x - mean) rstd w + b). Statistics and the
 * whole evaluation in double on the dtype-rounded inputs; outputs rounded once. */
int orc_norm_fwd(int kind, int dtype, int64_t rows, int64_t cols, const void *x, const void *w, const void *b, double eps, void *y,
                 float *mean_out, float *rstd_out) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_BF16) || (kind != 0 && kind != 1)) return 1;
    const int es = dt_size(dtype);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < rows; ++r) {
        const char *xr = (const char *)x + r * cols * es;
        double s = 0.0, mean = 0.0, q = 0.0;
        if (kind == 1) {
            for (int64_t c = 0; c < cols; ++c) s += ld_d(dtype, xr + c * es);
            mean = s / (double)cols;
        }
        for (int64_t c = 0; c < cols; ++c) { const double d = ld_d(dtype, xr + c * es) - mean; q += d * d; }
        const double rstd = 1.0 / sqrt(q / (double)cols + eps);
        if (mean_out) mean_out[r] = (float)mean;
        if (rstd_out) rstd_out[r] = (float)rstd;
        for (int64_t c = 0; c < cols; ++c) {
            double t = (ld_d(dtype, xr + c * es) - mean) * rstd;
            if (w) t *= ld_d(dtype, (const char *)w + c * es);
            if (b) t += ld_d(dtype, (const char *)b + c * es);
            st_d(dtype, (char *)y + (r * cols + c) * es, t);
        }
    }
    return 0;
}

/* closed-form backward of the above (no reference counterpart): g = dy w, xhat = (x - mean) rstd,
 * dx = rstd (g - mean(g) - xhat mean(g xhat)) (rms: without the mean(g) term), dw = sum_rows dy xhat, db = sum_rows dy */
int orc_norm_bwd(int kind, int dtype, int64_t rows, int64_t cols, const void *x, const void *w, double eps, const void *dy, void *dx,
                 void *dw, void *db) {
    if (!(dtype == ORC_F32 || dtype == ORC_F16 || dtype == ORC_BF16) || (kind != 0 && kind != 1)) return 1;
    const int es = dt_size(dtype);
    double *sw = (double *)calloc((size_t)cols, sizeof(double)), *sb = (double *)calloc((size_t)cols, sizeof(double));
    if (!sw || !sb) { free(sw); free(sb); return 2; }
    for (int64_t r = 0; r < rows; ++r) {
        const char *xr = (const char *)x + r * cols * es, *gr = (const char *)dy + r * cols * es;
        double s = 0.0, mean = 0.0, q = 0.0;
        if (kind == 1) {
            for (int64_t c = 0; c < cols; ++c) s += ld_d(dtype, xr + c * es);
            mean = s / (double)cols;
        }
        for (int64_t c = 0; c < cols; ++c) { const double d = ld_d(dtype, xr + c * es) - mean; q += d * d; }
        const double rstd = 1.0 / sqrt(q / (double)cols + eps);
        double s1 = 0.0, s2 = 0.0;
        for (int64_t c = 0; c < cols; ++c) {
            const double d = ld_d(dtype, gr + c * es), xh = (ld_d(dtype, xr + c * es) - mean) * rstd;
            const double g = d * (w ? ld_d(dtype, (const char *)w + c * es) : 1.0);
            s1 += g;
            s2 += g * xh;
            sw[c] += d * xh;
            sb[c] += d;
        }
        s1 = kind == 1 ? s1 / (double)cols : 0.0;
        s2 /= (double)cols;
        for (int64_t c = 0; c < cols; ++c) {
            const double d = ld_d(dtype, gr + c * es), xh = (ld_d(dtype, xr + c * es) - mean) * rstd;
            const double g = d * (w ? ld_d(dtype, (const char *)w + c * es) : 1.0);
            st_d(dtype, (char *)dx + (r * cols + c) * es, rstd * (g - s1 - xh * s2));
        }
    }
    for (int64_t c = 0; c < cols; ++c) {
        if (dw) st_d(dtype, (char *)dw + c * es, sw[c]);
        if (db) st_d(dtype, (char *)db + c * es, sb[c]);
    }
    free(sw);
    free(sb);
    return 0;
}

/* embedding gather (README.md:30 roadmap "embedding"; the read side of tensor_index.h:56-104's index arithmetic):
 * out[n, :] = table[wrap(idx[n]), :], negative indices wrap once, rows of row_bytes bytes (bit-exact byte copy) */
int orc_index_get(const void *table, int64_t nrows, int64_t row_bytes, const int64_t *idx, int64_t n, void *out) {
    for (int64_t i = 0; i < n; ++i) {
        int64_t r = idx[i];
        if (r < 0) r += nrows;
        if (r < 0 || r >= nrows) return 1;
        memcpy((char *)out + i * row_bytes, (const char *)table + r * row_bytes, (size_t)row_bytes);
    }
    return 0;
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
