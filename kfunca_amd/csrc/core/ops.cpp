// namespace gpu: operator implementations over the C ABI.
// Behavioural references: src/core/{binary,unary,nullary,reduce,gemm,nn,index}_ops.cpp,
// src/core/tensor_shape.cpp and the argument checks of src/device/{gemm,causal_attention}_kernel.cu.
#include "ops.h"

#include <algorithm>
#include <cmath>
#include <limits>

#include "device_api.h"
#include "tensor_iterator.h"

using utils::memory::DataPtr;
using utils::memory::DeviceAllocator;

namespace gpu {

namespace {

int code(ScalarType t) { return static_cast<int>(t); }

// elementwise launch: contiguous iteration spaces go down whole (the kernel indexes in 64 bits);
// strided ones are cut into 32-bit-indexable pieces first (reference tensor_loops.h:357-369)
void run_elementwise(TensorIterator &iter, int op, ScalarType compute, double scalar = 0.0) {
    if (iter.numel() == 0) return;
    void *stream = dev::stream(iter.device(0));
    auto launch = [&](const IterGeometry &g) {
        kf_iter_desc d;
        g.to_desc(d);
        DEV_CALL(kf_elementwise(op, &d, compute == ScalarType::Undefined ? 0 : code(compute), scalar, stream));
    };
    if (iter.geometry().is_contiguous()) launch(iter.geometry());
    else iter.geometry().for_each_32bit(launch);
}

Tensor &binary_out(int op, Tensor &out, const Tensor &left, const Tensor &right) {
    auto iter = TensorIterator().add_output(out).add_input(left).add_input(right).build_for_loops();
    run_elementwise(iter, op, iter.common_dtype());
    return out;
}

void run_reduce(TensorIterator &iter, int op) {
    if (iter.num_output_elements() == 0) return;
    CHECK_FAIL(iter.can_use_32bit_indexing(), "reduction over more than 2^31 bytes per operand is not supported yet");
    kf_iter_desc d;
    iter.geometry().to_desc(d);
    size_t need = 0;
    DEV_CALL(kf_reduce_workspace_bytes(&d, &need));
    const int device = iter.device(0);
    DataPtr scratch;
    if (need) scratch = DeviceAllocator::GetInstance()->allocate(need, device);
    DEV_CALL(kf_reduce(op, &d, scratch.get(), need, dev::stream(device)));
    // scratch returns to the cache here; reuse is stream-ordered behind the kernel that reads it
}

// two-output statistics reduction: iterator outputs are (variance-like, mean) as in reduce_ops.cpp:24
void run_moments(TensorIterator &iter, int mode, double correction, double eps) {
    if (iter.num_output_elements() == 0) return;
    CHECK_FAIL(iter.can_use_32bit_indexing(), "reduction over more than 2^31 bytes per operand is not supported yet");
    kf_iter_desc d;
    iter.geometry().to_desc(d);
    size_t need = 0;
    DEV_CALL(kf_reduce_moments_workspace_bytes(&d, &need));
    const int device = iter.device(0);
    DataPtr scratch;
    if (need) scratch = DeviceAllocator::GetInstance()->allocate(need, device);
    DEV_CALL(kf_reduce_moments(mode, &d, correction, eps, scratch.get(), need, dev::stream(device)));
}

} // namespace

// ---- binary (binary_ops.cpp:6-91) ------------------------------------------------------------------
Tensor &add_out(Tensor &out, const Tensor &l, const Tensor &r) { return binary_out(KF_EW_ADD, out, l, r); }
Tensor &sub_out(Tensor &out, const Tensor &l, const Tensor &r) { return binary_out(KF_EW_SUB, out, l, r); }
Tensor &mul_out(Tensor &out, const Tensor &l, const Tensor &r) { return binary_out(KF_EW_MUL, out, l, r); }
Tensor &div_out(Tensor &out, const Tensor &l, const Tensor &r) { return binary_out(KF_EW_DIV, out, l, r); }
Tensor &add_(Tensor &self, const Tensor &other) { return add_out(self, self, other); }
Tensor &sub_(Tensor &self, const Tensor &other) { return sub_out(self, self, other); }
Tensor &mul_(Tensor &self, const Tensor &other) { return mul_out(self, self, other); }
Tensor &div_(Tensor &self, const Tensor &other) { return div_out(self, self, other); }

namespace {
// d(a+b) = (g, g): the reference's only GradFunction (binary_ops.cpp:16-33)
class AddGradFunction : public GradFunction {
public:
    AddGradFunction(const Tensor &l, const Tensor &r) { inputs = {l, r}; }
    std::vector<Tensor> backward(Tensor g) override {
        std::vector<Tensor> out(2);
        if (inputs[0].requires_grad()) out[0] = g;
        if (inputs[1].requires_grad()) out[1] = g;
        return out;
    }
};
} // namespace

Tensor add(const Tensor &left, const Tensor &right) {
    Tensor out;
    add_out(out, left, right);
    out.set_requires_grad(left.requires_grad() || right.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new AddGradFunction(left, right));
    return out;
}

// ---- tensor (op) scalar (register.cpp:172-206) ---------------------------------------------------------
// The reference materialises `empty_like(self).fill_(s)` and runs the binary kernel on it: one full-size write plus
// one full-size read more than needed. Here the scalar rides in the launch (KF_EW_*_SCALAR, rounded to self's dtype
// first exactly as fill_ would); dtypes the scalar kernel does not cover broadcast a 1-element tensor instead.
namespace {
class AddScalarGradFunction : public GradFunction { // d(a + c) = g
public:
    explicit AddScalarGradFunction(const Tensor &a) { inputs = {a}; }
    std::vector<Tensor> backward(Tensor g) override { return {g}; }
};
bool scalar_kernel_covers(ScalarType t) {
    switch (t) {
    case ScalarType::Float: case ScalarType::Double: case ScalarType::BFloat16: case ScalarType::Half:
    case ScalarType::Int: case ScalarType::Long: return true;
    default: return false;
    }
}
Tensor &binary_scalar_out(int op, Tensor &out, const Tensor &self, double value) {
    CHECK_FAIL(self.defined());
    if (!scalar_kernel_covers(self.dtype()) || (out.defined() && out.dtype() != self.dtype())) {
        Tensor one = empty(std::vector<int64_t>(self.dim() > 0 ? self.dim() : 1, 1), self.dtype(), self.device());
        one.fill_(any_t{value});
        return binary_out(op, out, self, one);
    }
    auto iter = TensorIterator().add_output(out).add_input(self).build_for_loops();
    run_elementwise(iter, op + (KF_EW_ADD_SCALAR - KF_EW_ADD), ScalarType::Undefined, value);
    return out;
}
} // namespace
Tensor add(const Tensor &self, double s) {
    Tensor out;
    binary_scalar_out(KF_EW_ADD, out, self, s);
    out.set_requires_grad(self.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new AddScalarGradFunction(self));
    return out;
}
namespace {
class ScaleGradFunction : public GradFunction { // d(a * c) = g * c (c = 1 / s for a / s)
public:
    ScaleGradFunction(const Tensor &a, double c) : c_(c) { inputs = {a}; }
    std::vector<Tensor> backward(Tensor g) override {
        Tensor out;
        binary_scalar_out(KF_EW_MUL, out, g, c_);
        return {out};
    }

private:
    double c_;
};
} // namespace
Tensor sub(const Tensor &self, double s) {
    Tensor out;
    binary_scalar_out(KF_EW_SUB, out, self, s);
    out.set_requires_grad(self.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new AddScalarGradFunction(self));
    return out;
}
Tensor mul(const Tensor &self, double s) {
    Tensor out;
    binary_scalar_out(KF_EW_MUL, out, self, s);
    out.set_requires_grad(self.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new ScaleGradFunction(self, s));
    return out;
}
Tensor div(const Tensor &self, double s) {
    Tensor out;
    binary_scalar_out(KF_EW_DIV, out, self, s);
    out.set_requires_grad(self.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new ScaleGradFunction(self, 1.0 / s));
    return out;
}
Tensor &add_(Tensor &self, double s) { return binary_scalar_out(KF_EW_ADD, self, self, s); }
Tensor &sub_(Tensor &self, double s) { return binary_scalar_out(KF_EW_SUB, self, self, s); }
Tensor &mul_(Tensor &self, double s) { return binary_scalar_out(KF_EW_MUL, self, self, s); }
Tensor &div_(Tensor &self, double s) { return binary_scalar_out(KF_EW_DIV, self, self, s); }

namespace {
// d(a - b) = (g, -g); d(a * b) = (g * b, g * a). Same-shape operands only: a broadcast operand's gradient would need the
// matching reduction, which nothing on the path (residual adds, gating) asks for.
class SubGradFunction : public GradFunction {
public:
    SubGradFunction(const Tensor &l, const Tensor &r) { inputs = {l, r}; }
    std::vector<Tensor> backward(Tensor g) override {
        std::vector<Tensor> out(2);
        if (inputs[0].requires_grad()) out[0] = g;
        if (inputs[1].requires_grad()) binary_scalar_out(KF_EW_MUL, out[1], g, -1.0);
        return out;
    }
};
class MulGradFunction : public GradFunction {
public:
    MulGradFunction(const Tensor &l, const Tensor &r) { inputs = {l, r}; }
    std::vector<Tensor> backward(Tensor g) override {
        std::vector<Tensor> out(2);
        if (inputs[0].requires_grad()) mul_out(out[0], g, inputs[1]);
        if (inputs[1].requires_grad()) mul_out(out[1], g, inputs[0]);
        return out;
    }
};
bool same_shape(const Tensor &a, const Tensor &b) { return a.sizes() == b.sizes(); }
} // namespace
Tensor sub(const Tensor &l, const Tensor &r) {
    Tensor out;
    sub_out(out, l, r);
    if ((l.requires_grad() || r.requires_grad()) && same_shape(l, r)) {
        out.set_requires_grad(true);
        out.set_grad_fn(new SubGradFunction(l, r));
    }
    return out;
}
Tensor mul(const Tensor &l, const Tensor &r) {
    Tensor out;
    mul_out(out, l, r);
    if ((l.requires_grad() || r.requires_grad()) && same_shape(l, r)) {
        out.set_requires_grad(true);
        out.set_grad_fn(new MulGradFunction(l, r));
    }
    return out;
}
Tensor div(const Tensor &l, const Tensor &r) { Tensor out; div_out(out, l, r); return out; }

// ---- unary / nullary (unary_ops.cpp:7-24, nullary_ops.cpp:6-14) --------------------------------------
Tensor &copy_(Tensor &self, const Tensor &other) {
    auto iter = TensorIterator().add_output(self).add_input(other).resize_outputs(false).check_mem_overlap(false).build_for_loops();
    run_elementwise(iter, KF_EW_COPY, ScalarType::Undefined);
    return self;
}

namespace {
class CloneGradFunction : public GradFunction { // contiguous() / clone: the gradient passes through
public:
    explicit CloneGradFunction(const Tensor &a) { inputs = {a}; }
    std::vector<Tensor> backward(Tensor g) override { return {g}; }
};
class ConvertGradFunction : public GradFunction { // dtype conversion: the gradient is converted back
public:
    explicit ConvertGradFunction(const Tensor &a) { inputs = {a}; }
    std::vector<Tensor> backward(Tensor g) override { return {g.dtype() == inputs[0].dtype() ? g : convert(g, inputs[0].dtype())}; }
};
} // namespace

Tensor clone(const Tensor &self) {
    Tensor out = empty_like(self);
    copy_(out, self);
    out.set_requires_grad(self.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new CloneGradFunction(self));
    return out;
}

Tensor convert(const Tensor &self, ScalarType dtype) {
    Tensor out = empty(self.sizes(), dtype, self.device());
    auto iter = TensorIterator().add_output(out).add_input(self).build_for_loops();
    run_elementwise(iter, KF_EW_COPY, ScalarType::Undefined);
    out.set_requires_grad(self.requires_grad() && is_floating_type(dtype));
    if (out.requires_grad()) out.set_grad_fn(new ConvertGradFunction(self));
    return out;
}

Tensor &fill_out(Tensor &out, const any_t &value) {
    auto iter = TensorIterator().add_output(out).resize_outputs(false).build();
    run_elementwise(iter, KF_EW_FILL, ScalarType::Undefined, static_cast<double>(value));
    return out;
}
Tensor &fill_(Tensor &self, const any_t &value) { return fill_out(self, value); }

// ---- reductions (reduce_ops.cpp:8-28) ------------------------------------------------------------------
Tensor sum(const Tensor &self, int64_t reduce_dim) {
    Tensor out;
    auto iter = TensorIterator().add_output(out).add_input(self).build_for_reduce(reduce_dim);
    run_reduce(iter, KF_RED_SUM);
    return out;
}

Tensor mean(const Tensor &self, int64_t reduce_dim) {
    Tensor out;
    auto iter = TensorIterator().add_output(out).add_input(self).build_for_reduce(reduce_dim);
    run_reduce(iter, KF_RED_MEAN);
    return out;
}

// reduce_ops.cpp:22-28: returns (mean, var | std) with the unbiased divisor (correction = 1), keepdim, input dtype
std::tuple<Tensor, Tensor> mean_var(const Tensor &self, int64_t reduce_dim, bool take_sqrt) {
    CHECK_FAIL(self.defined());
    CHECK_FAIL(is_floating_type(self.dtype()), "Unsupported ScalarType ", self.dtype()); // DISPATCH_FLOATING_TYPES, reduce_ops_kernel.cu:149-153
    Tensor mean, var;
    auto iter = TensorIterator().add_output(var).add_output(mean).add_input(self).build_for_reduce(reduce_dim);
    run_moments(iter, take_sqrt ? KF_MOM_STD : KF_MOM_VAR, /*correction=*/1.0, 0.0);
    return std::make_tuple(mean, var);
}

// norm_ops.cpp:8-10 / norm_ops_kernel.cu:6-61: (mean, 1/sqrt(biased var + 1e-12)) in the accumulate dtype.
// The reference accepts only dim == 0 of a 2-D tensor (norm_ops_kernel.cu:8); any dim of any rank works here.
std::tuple<Tensor, Tensor> norm_stat(const Tensor &self, int64_t dim) {
    CHECK_FAIL(self.defined());
    CHECK_FAIL(is_floating_type(self.dtype()), "Unsupported ScalarType ", self.dtype());
    CHECK_FAIL(dim >= -self.dim() && dim < self.dim(), "dim ", dim, " out of range");
    if (dim < 0) dim += self.dim();
    const ScalarType acc = accumulate_type(self.dtype());
    Tensor save_mean = empty_like_reduced(self, (int)dim, acc);
    Tensor save_invstd = empty_like_reduced(self, (int)dim, acc);
    auto iter = TensorIterator().add_output(save_invstd).add_output(save_mean).add_input(self).resize_outputs(false).build_for_reduce(dim);
    run_moments(iter, KF_MOM_INVSTD, 0.0, /*eps=*/1e-12);
    return std::make_tuple(save_mean, save_invstd);
}
// ---- sort / topk (sort_ops.cpp:6-19 + the host half of sort_ops_kernel.cu:507-632) ---------------------------
namespace {

// Dense strides over self's sizes that keep the memory order of the other dims and make `dim` the fastest
// (infer_dense_strides_dim_last, sort_ops_kernel.cu:523-554).
std::vector<int64_t> dense_strides_dim_last(const Tensor &self, int64_t dim) {
    const int nd = self.dim();
    std::vector<int> order;
    for (int i = 0; i < nd; ++i)
        if (i != dim) order.push_back(i);
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return self.stride(x) > self.stride(y); });
    order.push_back((int)dim);
    std::vector<int64_t> strides(nd);
    int64_t run = 1;
    for (int i = nd - 1; i >= 0; --i) {
        strides[order[i]] = run;
        run *= self.shape(order[i]);
    }
    return strides;
}

} // namespace

std::tuple<Tensor, Tensor> sort(const Tensor &self, int64_t dim, bool descending) {
    CHECK_FAIL(self.defined());
    CHECK_FAIL(self.dim() > 0 && dim >= -self.dim() && dim < self.dim(), "dim ", dim, " out of range");
    dim = maybe_wrap_dim((int)dim, self.dim());
    const int64_t numel = self.numel(), nsort = self.shape((int)dim);
    CHECK_FAIL(nsort <= std::numeric_limits<int>::max(), "The dimension being sorted can not have more than INT_MAX elements.");
    CHECK_FAIL(self.dtype() != ScalarType::Bool, "Sort currently does not support bool dtypes.");

    // keys with the sorted dim contiguous: self itself, or a dense copy with that dim last (sort_ops_kernel.cu:572-581)
    const bool direct = self.is_dense() && self.stride((int)dim) == 1;
    Tensor keys = self;
    if (!direct) {
        keys = empty_strided(self.sizes(), dense_strides_dim_last(self, dim), self.dtype(), self.device());
        keys.copy_(self);
    }
    // (direct: fresh row-major outputs as the reference's empty_like / empty, sort_ops_kernel.cu:583-585 - not self's own strides, whose extent-1 dims may carry
    //  anything: a later in-place operation on the result is refused or not by those strides; tests/test_gpu_host_diff_fuzz.py found the difference)
    Tensor values = direct ? empty_like(keys) : empty_strided(keys.sizes(), keys.strides(), keys.dtype(), keys.device());
    Tensor indices = direct ? empty(keys.sizes(), ScalarType::Long, keys.device()) : empty_strided(keys.sizes(), keys.strides(), ScalarType::Long, keys.device());
    if (numel > 0) {
        const int64_t nseg = numel / nsort;
        const int dt = code(self.dtype());
        const size_t need = kf_sort_workspace_bytes(dt, nseg, nsort);
        DataPtr scratch;
        if (need) scratch = DeviceAllocator::GetInstance()->allocate(need, self.device());
        DEV_CALL(kf_sort(dt, keys.data_ptr(), values.data_ptr(), static_cast<int64_t *>(indices.data_ptr()), nseg, nsort, descending,
                         scratch.get(), need, dev::stream(self.device())));
    }
    if (direct) return std::make_tuple(values, indices);
    Tensor values_out = empty_like(self); // back to self's layout (sort_ops_kernel.cu:609-614)
    Tensor indices_out = empty(self.sizes(), ScalarType::Long, self.device());
    values_out.copy_(values);
    indices_out.copy_(indices);
    return std::make_tuple(values_out, indices_out);
}

// topk_with_sort (sort_ops_kernel.cu:620-632): the first k entries of the stable sort along dim
std::tuple<Tensor, Tensor> topk(const Tensor &self, int64_t k, int64_t dim, bool largest) {
    CHECK_FAIL(self.defined());
    CHECK_FAIL(self.dim() > 0 && dim >= -self.dim() && dim < self.dim(), "dim ", dim, " out of range");
    dim = maybe_wrap_dim((int)dim, self.dim());
    CHECK_FAIL(k >= 0 && k <= self.shape((int)dim), "k ", k, " out of range for a dimension of ", self.shape((int)dim));
    Tensor sorted_values, sorted_indices;
    std::tie(sorted_values, sorted_indices) = sort(self, dim, largest);
    auto sizes = self.sizes();
    sizes[dim] = k;
    Tensor values = empty(sizes, self.dtype(), self.device());
    Tensor indices = empty(sizes, ScalarType::Long, self.device());
    values.copy_(sorted_values.narrow(dim, 0, k));
    indices.copy_(sorted_indices.narrow(dim, 0, k));
    return std::make_tuple(values, indices);
}

// ---- GEMM (gemm_ops.cpp:6-16 + the checks of gemm_kernel.cu:8-25) -----------------------------------------
namespace {

bool gemm_dtype_ok(ScalarType t) {
    return t == ScalarType::Float || t == ScalarType::Double || t == ScalarType::Half || t == ScalarType::BFloat16;
}

// C[M,N] = alpha op(A) op(B) + beta C on raw 2-D geometry
void launch_gemm(ScalarType dt, bool ta, bool tb, int64_t M, int64_t N, int64_t K, float alpha, const void *A, int64_t lda,
                 const void *B, int64_t ldb, float beta, void *C, int64_t ldc, int device, bool c_f32 = false) {
    if (c_f32) { // 16-bit operands, float C (a gradient slot of an f32 bucket): the epilogue flag of the C ABI, no split-K scratch
        kf_gemm_epilogue e{};
        e.c_f32 = 1;
        DEV_CALL(kf_gemm_ex(code(dt), ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, &e, dev::stream(device)));
        return;
    }
    size_t need = 0;
    DEV_CALL(kf_gemm_workspace_bytes(code(dt), ta, tb, M, N, K, &need));
    DataPtr scratch;
    if (need) scratch = DeviceAllocator::GetInstance()->allocate(need, device);
    DEV_CALL(kf_gemm(code(dt), ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, KF_EPI_NONE, nullptr, scratch.get(), need,
                     dev::stream(device)));
}

// Shapes the matrix-core kernels do not take (M, N, K off their tile multiples) are zero-padded on the host side of the
// boundary when the problem is big enough to care: zero rows / columns add nothing to any product, the copies cost
// O(MK + KN + MN) against O(MNK), and the ragged case (a 50257-column projection, say) runs on the MFMA kernels instead
// of the scalar fallback. A2 / B2 are the STORED contiguous 2-D operands, C2 is [M, N].
Tensor pad2d(const Tensor &t, int64_t rows, int64_t cols) {
    if (t.shape(0) == rows && t.shape(1) == cols) return t;
    Tensor p = zeros({rows, cols}, t.dtype(), t.device());
    Tensor head = p.narrow(0, 0, t.shape(0)).narrow(1, 0, t.shape(1));
    copy_(head, t);
    return p;
}
void gemm_any(ScalarType dt, bool ta, bool tb, int64_t M, int64_t N, int64_t K, float alpha, const Tensor &A2, const Tensor &B2, float beta,
              Tensor &C2, int device) {
    int64_t am = 0, ak = 0;
    if (dt == ScalarType::Half || dt == ScalarType::BFloat16) { am = 128; ak = 64; }
    else if (dt == ScalarType::Float || dt == ScalarType::Double) { am = 64; ak = 16; }
    const bool aligned = am && M % am == 0 && N % am == 0 && K % ak == 0;
    // a float C behind 16-bit operands (the slot of an f32 gradient bucket): written by the kernel itself, unrounded
    const bool c_f32 = C2.dtype() == ScalarType::Float && (dt == ScalarType::Half || dt == ScalarType::BFloat16);
    CHECK_FAIL(c_f32 || C2.dtype() == dt, "gemm: the output's dtype must be the operands' (or float behind 16-bit operands)");
    if (!am || aligned || M * N * K < ((int64_t)1 << 22)) {
        launch_gemm(dt, ta, tb, M, N, K, alpha, A2.data_ptr(), A2.shape(1), B2.data_ptr(), B2.shape(1), beta, C2.data_ptr(), N, device, c_f32);
        return;
    }
    auto up = [](int64_t v, int64_t a) { return (v + a - 1) / a * a; };
    const int64_t Mp = up(M, am), Np = up(N, am), Kp = up(K, ak);
    Tensor Ap = pad2d(A2, ta ? Kp : Mp, ta ? Mp : Kp), Bp = pad2d(B2, tb ? Np : Kp, tb ? Kp : Np);
    const bool same_c = Mp == M && Np == N;
    Tensor Cp = same_c ? C2 : (beta != 0.f ? pad2d(C2, Mp, Np) : empty({Mp, Np}, C2.dtype(), device));
    launch_gemm(dt, ta, tb, Mp, Np, Kp, alpha, Ap.data_ptr(), Ap.shape(1), Bp.data_ptr(), Bp.shape(1), beta, Cp.data_ptr(), Np, device, c_f32);
    if (!same_c) {
        Tensor head = Cp.narrow(0, 0, M).narrow(1, 0, N);
        copy_(C2, head);
    }
}

// Where a weight gradient is written: a bucketed leaf whose gradient starts empty takes it straight into its bucket slot (update_grad then
// has nothing to copy) - but only the FIRST product of a backward pass that asks (GradSink::take_slot): a weight used by two products gets
// a fresh tensor for the second, and the engine sums the two.
Tensor grad_target(const Tensor &b) {
    TensorImpl *bi = b.impl();
    if (!b.has_grad_fn())
        if (std::shared_ptr<GradSink> sink = bi->sink_.lock()) {
            if (!bi->grad_)
                if (Tensor s = sink->take_slot(bi); s.defined()) return s;
            // a later producer of the same pass (or a pass that accumulates): its own tensor, in the BUCKET's dtype - a float bucket's
            // gradients stay float all the way into the sum
            return empty(b.sizes(), sink->slot(bi).dtype(), b.device());
        }
    return empty(b.sizes(), b.dtype(), b.device());
}

// dA = alpha * dC B^T, dB = alpha * A^T dC with A flattened to [M,K] (no reference counterpart)
// Both gradients of a linear layer y = alpha a b from its output gradient g2 [M, N]: da = alpha g2 b^T (NT), db = alpha a2^T g2 (TN). A 16-bit
// layer on 256-tile shapes gets them from ONE grid where the device library can fuse the pair (kf_gemm_grouped_single_grid: the second
// product's first tiles start under the first one's last tiles); where it cannot - a skinny product such as dA of x[256, 4096] W[4096, 16384] -
// two ordinary calls run, which carry the split-K scratch kf_gemm_grouped's fall-back would not have. A bucketed weight whose gradient starts
// empty takes dW straight into its bucket slot (grad_target: update_grad then has nothing to copy); the slot is handed out ONCE per pass
// (GradSink::take_slot), so it is asked for once here whichever way the products run.
void linear_backward(const Tensor &a, const Tensor &b, const Tensor &g2, float alpha, int64_t M, int64_t N, int64_t K, Tensor &out_da, Tensor &out_db) {
    Tensor a2 = a.view({M, K});
    Tensor da, db;
    if (a.requires_grad() && b.requires_grad() && M % 256 == 0 && N % 256 == 0 && K % 256 == 0 &&
        (a.dtype() == ScalarType::Half || a.dtype() == ScalarType::BFloat16)) {
        da = empty(a.sizes(), a.dtype(), a.device());
        db = grad_target(b);
        kf_gemm_problem p[2] = {};
        p[0].trans_a = 0; p[0].trans_b = 1; p[0].M = M; p[0].N = K; p[0].K = N; p[0].alpha = alpha; p[0].beta = 0.f;
        p[0].A = g2.data_ptr(); p[0].lda = N; p[0].B = b.data_ptr(); p[0].ldb = N; p[0].C = da.data_ptr(); p[0].ldc = K;
        p[1].trans_a = 1; p[1].trans_b = 0; p[1].M = K; p[1].N = N; p[1].K = M; p[1].alpha = alpha; p[1].beta = 0.f;
        p[1].A = a2.data_ptr(); p[1].lda = K; p[1].B = g2.data_ptr(); p[1].ldb = N; p[1].C = db.data_ptr(); p[1].ldc = N;
        p[1].c_f32 = db.dtype() == ScalarType::Float ? 1 : 0; // the slot of an f32 gradient bucket: dW leaves the accumulators unrounded
        if (kf_gemm_grouped_single_grid(code(a.dtype()), 2, p)) {
            DEV_CALL(kf_gemm_grouped(code(a.dtype()), 2, p, dev::stream(a.device())));
            out_da = da;
            out_db = db;
            return;
        }
    }
    if (a.requires_grad()) {
        out_da = da.defined() ? da : empty(a.sizes(), a.dtype(), a.device());
        Tensor c2 = out_da.view({M, K});
        gemm_any(a.dtype(), false, true, M, K, N, alpha, g2, b, 0.f, c2, a.device());
    }
    if (b.requires_grad()) {
        out_db = db.defined() ? db : grad_target(b);
        gemm_any(a.dtype(), true, false, K, N, M, alpha, a2, g2, 0.f, out_db, b.device());
    }
}

class GemmGradFunction : public GradFunction {
public:
    GemmGradFunction(const Tensor &a, const Tensor &b, float alpha) : alpha_(alpha) { inputs = {a, b}; }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &a = inputs[0], &b = inputs[1];
        const int64_t K = b.shape(0), N = b.shape(1), M = a.numel() / K;
        Tensor gc = g.dense();
        std::vector<Tensor> out(2);
        linear_backward(a, b, gc.view({M, N}), alpha_, M, N, K, out[0], out[1]);
        return out;
    }

private:
    float alpha_;
};

} // namespace

void gemm_out(Tensor &out, const Tensor &a, const Tensor &b, float alpha, float beta) {
    // (the reference asks its FLAG here, gemm_kernel.cu:9 - it refuses every view, dense or not, e.g. x.view(T, d); this host asks the strides: a superset)
    CHECK_FAIL(out.is_dense() && a.is_dense() && b.is_dense());
    CHECK_FAIL(a.dim() >= 1);
    const int64_t k = a.shape(-1);
    CHECK_FAIL(k > 0);
    const int64_t m = a.numel() / k;
    CHECK_FAIL(b.dim() == 2 && b.shape(0) == k);
    CHECK_FAIL(a.dtype() == b.dtype());
    CHECK_FAIL(out.dtype() == a.dtype());
    const int64_t n = b.shape(-1);
    CHECK_FAIL(out.shape(-1) == n);
    CHECK_FAIL(n > 0 && out.numel() / n == m);
    CHECK_FAIL(gemm_dtype_ok(a.dtype()), "Unsupported ScalarType ", a.dtype());
    CHECK_FAIL(a.device() == b.device() && a.device() == out.device());
    Tensor a2 = a.view({m, k}), c2 = out.view({m, n});
    gemm_any(a.dtype(), false, false, m, n, k, alpha, a2, b, beta, c2, a.device());
}

Tensor gemm(const Tensor &a, const Tensor &b, float alpha, float beta) {
    auto out_size = a.sizes();
    CHECK_FAIL(!out_size.empty() && b.dim() >= 1);
    out_size.back() = b.shape(-1);
    Tensor out = empty(out_size, a.dtype(), a.device());
    gemm_out(out, a, b, alpha, beta);
    out.set_requires_grad(a.requires_grad() || b.requires_grad());
    if (out.requires_grad()) {
        CHECK_FAIL(beta == 0.f, "gemm(): autograd needs beta == 0 (the output operand is not an input)");
        out.set_grad_fn(new GemmGradFunction(a, b, alpha));
    }
    return out;
}

// ---- gemm with the fused element-wise tail ------------------------------------------------------------------------------------
namespace {
// y = (alpha a b + bias) o mul + add. With t = the bracket: d_add = g; d_mul = g o t (t kept only when mul wants a gradient);
// d_t = g o mul; d_bias = column sums of d_t; da = alpha d_t b^T, db = alpha a^T d_t.
class GemmFusedGradFunction : public GradFunction {
public:
    GemmFusedGradFunction(const Tensor &a, const Tensor &b, const Tensor &bias, const Tensor &mul, const Tensor &add, const Tensor &raw, float alpha)
        : alpha_(alpha), raw_(raw) {
        inputs = {a, b};
        ibias_ = imul_ = iadd_ = -1;
        if (bias.defined()) { ibias_ = (int)inputs.size(); inputs.push_back(bias); }
        if (mul.defined()) { imul_ = (int)inputs.size(); inputs.push_back(mul); }
        if (add.defined()) { iadd_ = (int)inputs.size(); inputs.push_back(add); }
    }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &a = inputs[0], &b = inputs[1];
        const int64_t K = b.shape(0), N = b.shape(1), M = a.numel() / K;
        Tensor g2 = g.dense().view({M, N});
        std::vector<Tensor> out(inputs.size());
        if (iadd_ >= 0 && inputs[iadd_].requires_grad()) out[iadd_] = g2.view(inputs[iadd_].sizes());
        Tensor dt = g2;
        if (imul_ >= 0) {
            const Tensor m2 = inputs[imul_].view({M, N});
            if (inputs[imul_].requires_grad()) out[imul_] = mul(g2, raw_.view({M, N})).view(inputs[imul_].sizes());
            if (a.requires_grad() || b.requires_grad() || (ibias_ >= 0 && inputs[ibias_].requires_grad())) dt = mul(g2, m2);
        }
        if (ibias_ >= 0 && inputs[ibias_].requires_grad()) out[ibias_] = sum(dt, 0).view({N});
        // da = alpha d_t b^T and db = alpha a^T d_t: plain products (the tail's derivative is the element-wise d_t above) - as ONE grid where the
        // pair fits (round 6: config C5's fused block ran them as two launches each: 11 single GEMM launches + 2 pairs per step; now 5 + 5)
        linear_backward(a, b, dt, alpha_, M, N, K, out[0], out[1]);
        return out;
    }

private:
    float alpha_;
    Tensor raw_;
    int ibias_, imul_, iadd_;
};
} // namespace

Tensor gemm_fused(const Tensor &a, const Tensor &b, float alpha, const Tensor &bias, const Tensor &mul_t, const Tensor &add_t) {
    CHECK_FAIL(a.defined() && b.defined() && a.is_dense() && b.is_dense() && a.dim() >= 1 && b.dim() == 2);
    const int64_t k = a.shape(-1), n = b.shape(1);
    CHECK_FAIL(k > 0 && b.shape(0) == k && n > 0 && a.dtype() == b.dtype() && a.device() == b.device());
    CHECK_FAIL(gemm_dtype_ok(a.dtype()), "Unsupported ScalarType ", a.dtype());
    const int64_t m = a.numel() / k;
    auto out_size = a.sizes();
    out_size.back() = n;
    for (const Tensor *t : {&mul_t, &add_t})
        if (t->defined()) CHECK_FAIL(t->is_dense() && t->dtype() == a.dtype() && t->device() == a.device() && t->numel() == m * n && t->shape(-1) == n,
                                     "gemm_fused: mul / add must be contiguous tensors of the output's shape and dtype");
    if (bias.defined()) CHECK_FAIL(bias.is_dense() && bias.dim() == 1 && bias.shape(0) == n && bias.dtype() == a.dtype() && bias.device() == a.device(),
                                   "gemm_fused: bias must be a contiguous [N] tensor of the operands' dtype");
    Tensor out = empty(out_size, a.dtype(), a.device());
    const bool keep_raw = mul_t.defined() && mul_t.requires_grad();
    Tensor raw = keep_raw ? empty({m, n}, a.dtype(), a.device()) : Tensor();
    if (m > 0) {
        // extents off the matrix kernels' tiles: zero-padded operands, as in gemm_any (the tail's operands too: a padded row / column of the
        // output is computed from zeros and dropped). bf16 4000^3 + bias: 5.2 ms on the scalar kernel -> the tile kernels.
        int64_t am = 0, ak = 0;
        if (a.dtype() == ScalarType::Half || a.dtype() == ScalarType::BFloat16) { am = 128; ak = 64; }
        else if (a.dtype() == ScalarType::Float) { am = 64; ak = 16; }
        auto up = [](int64_t v, int64_t q) { return (v + q - 1) / q * q; };
        const int64_t mp = am ? up(m, am) : m, np = am ? up(n, am) : n, kp = am ? up(k, ak) : k;
        const bool padded = am && (mp != m || np != n || kp != k) && m * n * k >= ((int64_t)1 << 22);
        kf_gemm_epilogue e{};
        if (!padded) {
            e.bias = bias.defined() ? bias.data_ptr() : nullptr;
            e.mul = mul_t.defined() ? mul_t.data_ptr() : nullptr;
            e.add = add_t.defined() ? add_t.data_ptr() : nullptr;
            e.aux = keep_raw ? raw.data_ptr() : nullptr;
            e.ldmul = e.ldadd = e.ldaux = n;
            DEV_CALL(kf_gemm_ex(code(a.dtype()), 0, 0, m, n, k, alpha, a.data_ptr(), k, b.data_ptr(), n, 0.f, out.data_ptr(), n, &e, dev::stream(a.device())));
        } else {
            Tensor ap = pad2d(a.view({m, k}), mp, kp), bp = pad2d(b, kp, np);
            Tensor biasp = bias.defined() ? pad2d(bias.view({1, n}), 1, np) : Tensor();
            Tensor mulp = mul_t.defined() ? pad2d(mul_t.view({m, n}), mp, np) : Tensor(), addp = add_t.defined() ? pad2d(add_t.view({m, n}), mp, np) : Tensor();
            Tensor outp = empty({mp, np}, a.dtype(), a.device()), rawp = keep_raw ? empty({mp, np}, a.dtype(), a.device()) : Tensor();
            e.bias = biasp.defined() ? biasp.data_ptr() : nullptr;
            e.mul = mulp.defined() ? mulp.data_ptr() : nullptr;
            e.add = addp.defined() ? addp.data_ptr() : nullptr;
            e.aux = keep_raw ? rawp.data_ptr() : nullptr;
            e.ldmul = e.ldadd = e.ldaux = np;
            DEV_CALL(kf_gemm_ex(code(a.dtype()), 0, 0, mp, np, kp, alpha, ap.data_ptr(), kp, bp.data_ptr(), np, 0.f, outp.data_ptr(), np, &e, dev::stream(a.device())));
            Tensor o2 = out.view({m, n});
            copy_(o2, outp.narrow(0, 0, m).narrow(1, 0, n));
            if (keep_raw) copy_(raw, rawp.narrow(0, 0, m).narrow(1, 0, n));
        }
    }
    bool grad = a.requires_grad() || b.requires_grad();
    for (const Tensor *t : {&bias, &mul_t, &add_t}) grad = grad || (t->defined() && t->requires_grad());
    if (grad) {
        out.set_requires_grad(true);
        out.set_grad_fn(new GemmFusedGradFunction(a, b, bias, mul_t, add_t, raw, alpha));
    }
    return out;
}

Tensor gemm_ex(const Tensor &a, bool trans_a, const Tensor &b, bool trans_b, float alpha) {
    CHECK_FAIL(a.dim() == 2 && b.dim() == 2 && a.is_dense() && b.is_dense());
    CHECK_FAIL(a.dtype() == b.dtype() && gemm_dtype_ok(a.dtype()));
    const int64_t M = trans_a ? a.shape(1) : a.shape(0), K = trans_a ? a.shape(0) : a.shape(1);
    const int64_t Kb = trans_b ? b.shape(1) : b.shape(0), N = trans_b ? b.shape(0) : b.shape(1);
    CHECK_FAIL(K == Kb);
    Tensor out = empty({M, N}, a.dtype(), a.device());
    gemm_any(a.dtype(), trans_a, trans_b, M, N, K, alpha, a, b, 0.f, out, a.device());
    return out;
}

// ---- causal attention (nn_ops.cpp:6-8 + the checks of causal_attention_kernel.cu:9-20) ---------------------
namespace {

void check_attention(const Tensor &q, const Tensor &k, const Tensor &v) {
    CHECK_FAIL(q.dim() == 4 && k.dim() == 4 && v.dim() == 4);
    CHECK_FAIL(k.shape(0) == q.shape(0) && k.shape(1) == q.shape(1) && k.shape(3) == q.shape(3));
    CHECK_FAIL(k.sizes() == v.sizes());
    CHECK_FAIL(q.dtype() == k.dtype() && q.dtype() == v.dtype());
    CHECK_FAIL(q.dtype() == ScalarType::Float || q.dtype() == ScalarType::Half || q.dtype() == ScalarType::BFloat16,
               "Unsupported ScalarType ", q.dtype());
    CHECK_FAIL(q.is_dense() && k.is_dense() && v.is_dense());
    CHECK_FAIL(q.device() == k.device() && q.device() == v.device());
}

// The MFMA kernels want D = 64 or 128 (and, f32 or Skv < Sq, whole tiles of rows); everything else takes the generic
// vector-ALU kernel (two orders of magnitude slower). For 16-bit tensors with D <= 128 and Skv >= Sq both can be padded
// with zeros at no cost in results: zero columns change neither Q K^T nor P V (the softmax scale stays 1 / sqrt(D) of the
// real head size: kf_attn_*_scaled); a padded key n >= Skv >= Sq > m is above the diagonal of every real query; a padded
// query has q = 0 and dO = 0, so it contributes exactly zero to dK and dV.
// The same holds for f32 tensors and the exact-f32 MFMA kernels (head size 64 or 128, rows in multiples of 32).
struct PadPlan {
    bool pad = false;
    int64_t Sqp = 0, Skp = 0, Dp = 0;
};
PadPlan pad_for_mfma(const Tensor &q, const Tensor &k) {
    const int64_t Sq = q.shape(2), Skv = k.shape(2), D = q.shape(3);
    PadPlan p;
    if (!(D > 0 && D <= 128 && Skv >= Sq && Sq > 0)) return p;
    const bool h16 = q.dtype() == ScalarType::Half || q.dtype() == ScalarType::BFloat16;
    if (!h16 && q.dtype() != ScalarType::Float) return p;
    // round 6: the 16-bit matrix-core kernels take ANY sequence lengths with Skv >= Sq at the C ABI itself (rows beyond a tensor's end are
    // zero-filled / dropped by the kernels' buffer descriptors - no padded copies); only a head size off 64 / 128 is still padded here
    const int64_t rows = h16 ? 1 : 32;
    p.Dp = D <= 64 ? 64 : 128;
    p.Sqp = (Sq + rows - 1) / rows * rows;
    p.Skp = (Skv + rows - 1) / rows * rows;
    p.pad = p.Dp != D || p.Sqp != Sq || p.Skp != Skv;
    return p;
}
Tensor pad_to(const Tensor &t, int64_t rows, int64_t cols) { // [B,H,S,D] -> [B,H,rows,cols] (or [B,H,S] -> [B,H,rows]), zero-filled
    auto shape = t.sizes();
    const int64_t S = shape[2];
    shape[2] = rows;
    if (shape.size() == 4) shape[3] = cols;
    Tensor p = zeros(shape, t.dtype(), t.device());
    Tensor head = p.narrow(2, 0, S);
    if (shape.size() == 4) head = head.narrow(3, 0, t.shape(3));
    copy_(head, t);
    return p;
}
Tensor unpad(const Tensor &t, int64_t rows, int64_t cols) {
    Tensor v = t.narrow(2, 0, rows);
    if (t.dim() == 4) v = v.narrow(3, 0, cols);
    return v.dense();
}

class AttentionGradFunction : public GradFunction {
public:
    AttentionGradFunction(const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out, const Tensor &lse) : out_(out), lse_(lse) {
        inputs = {q, k, v};
    }
    std::vector<Tensor> backward(Tensor g) override {
        auto [dq, dk, dv] = causal_attention_bwd(inputs[0], inputs[1], inputs[2], out_, lse_, g);
        return {dq, dk, dv};
    }

private:
    Tensor out_, lse_;
};

} // namespace

std::tuple<Tensor, Tensor> causal_attention_fwd(const Tensor &q, const Tensor &k, const Tensor &v) {
    check_attention(q, k, v);
    const int64_t B = q.shape(0), H = q.shape(1), Sq = q.shape(2), D = q.shape(3), Skv = k.shape(2);
    if (const PadPlan pp = pad_for_mfma(q, k); pp.pad) {
        const int64_t Sqp = pp.Sqp, Skp = pp.Skp, Dp = pp.Dp;
        Tensor qp = pad_to(q, Sqp, Dp), kp = pad_to(k, Skp, Dp), vp = pad_to(v, Skp, Dp);
        Tensor outp = empty_like(qp);
        Tensor lsep = empty({B, H, Sqp}, ScalarType::Float, q.device());
        DEV_CALL(kf_attn_fwd_scaled(code(q.dtype()), B, H, Sqp, Skp, Dp, 1.0f / std::sqrt((float)D), qp.data_ptr(), kp.data_ptr(), vp.data_ptr(),
                                    outp.data_ptr(), static_cast<float *>(lsep.data_ptr()), dev::stream(q.device())));
        return {unpad(outp, Sq, D), unpad(lsep, Sq, 0)};
    }
    Tensor out = empty_like(q);
    Tensor lse = empty({B, H, Sq}, ScalarType::Float, q.device());
    DEV_CALL(kf_attn_fwd(code(q.dtype()), B, H, Sq, Skv, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                         static_cast<float *>(lse.data_ptr()), dev::stream(q.device())));
    return {out, lse};
}

// Backward scratch: the size the device library recommends (statistics + dS of as many (batch, head) pairs as its cap allows); when
// the allocator cannot supply that, halve the dS part until it can - the library accepts anything down to the statistics alone
// (then the recomputing dQ kernel runs: kf_attn_bwd, include/kfunca_hip.h).
static DataPtr attn_bwd_scratch(int dt, int64_t B, int64_t H, int64_t Sq, int64_t Skv, int64_t D, int device, size_t &bytes) {
    size_t need = 0, floor_ = 0;
    DEV_CALL(kf_attn_bwd_workspace_bytes(dt, B, H, Sq, Skv, D, &need));
    floor_ = 3 * (((size_t)B * H * Sq * sizeof(float) + 255) / 256 * 256);
    for (;;) {
        try {
            bytes = need;
            return DeviceAllocator::GetInstance()->allocate(need, device);
        } catch (const utils::OutOfMemory &) { // only that: any other failure is not cured by asking for less
            if (need <= floor_) throw;
            need = floor_ + (need - floor_) / 2;
            if (need - floor_ < ((size_t)1 << 20)) need = floor_;
        }
    }
}

std::tuple<Tensor, Tensor, Tensor> causal_attention_bwd(const Tensor &q, const Tensor &k, const Tensor &v, const Tensor &out,
                                                        const Tensor &lse, const Tensor &grad_out) {
    check_attention(q, k, v);
    CHECK_FAIL(grad_out.sizes() == q.sizes() && grad_out.dtype() == q.dtype());
    const int64_t B = q.shape(0), H = q.shape(1), Sq = q.shape(2), D = q.shape(3), Skv = k.shape(2);
    if (const PadPlan pp = pad_for_mfma(q, k); pp.pad) {
        const int64_t Sqp = pp.Sqp, Skp = pp.Skp, Dp = pp.Dp;
        Tensor qp = pad_to(q, Sqp, Dp), kp = pad_to(k, Skp, Dp), vp = pad_to(v, Skp, Dp), op = pad_to(out, Sqp, Dp);
        Tensor lp = pad_to(lse, Sqp, 0), gp = pad_to(grad_out.dense(), Sqp, Dp);
        Tensor dqp = empty_like(qp), dkp = empty_like(kp), dvp = empty_like(vp);
        size_t need = 0;
        DataPtr scratch = attn_bwd_scratch(code(q.dtype()), B, H, Sqp, Skp, Dp, q.device(), need);
        DEV_CALL(kf_attn_bwd_scaled(code(q.dtype()), B, H, Sqp, Skp, Dp, 1.0f / std::sqrt((float)D), qp.data_ptr(), kp.data_ptr(), vp.data_ptr(),
                                    op.data_ptr(), static_cast<const float *>(lp.data_ptr()), gp.data_ptr(), dqp.data_ptr(), dkp.data_ptr(),
                                    dvp.data_ptr(), scratch.get(), need, dev::stream(q.device())));
        return {unpad(dqp, Sq, D), unpad(dkp, Skv, D), unpad(dvp, Skv, D)};
    }
    Tensor go = grad_out.dense();
    Tensor dq = empty_like(q), dk = empty_like(k), dv = empty_like(v);
    size_t need = 0;
    DataPtr scratch = attn_bwd_scratch(code(q.dtype()), B, H, Sq, Skv, D, q.device(), need);
    DEV_CALL(kf_attn_bwd(code(q.dtype()), B, H, Sq, Skv, D, q.data_ptr(), k.data_ptr(), v.data_ptr(), out.data_ptr(),
                         static_cast<const float *>(lse.data_ptr()), go.data_ptr(), dq.data_ptr(), dk.data_ptr(), dv.data_ptr(),
                         scratch.get(), need, dev::stream(q.device())));
    return {dq, dk, dv};
}

Tensor causal_attention(const Tensor &q, const Tensor &k, const Tensor &v) {
    auto [out, lse] = causal_attention_fwd(q, k, v);
    out.set_requires_grad(q.requires_grad() || k.requires_grad() || v.requires_grad());
    if (out.requires_grad()) out.set_grad_fn(new AttentionGradFunction(q, k, v, out, lse));
    return out;
}

// ---- index_put_ (index_ops.cpp:6-38) ----------------------------------------------------------------------------
Tensor &index_put_(Tensor &self, const std::vector<Tensor> &indices, const Tensor &values) {
    CHECK_FAIL((int)indices.size() == self.dim(), "Number of indices must match the number of dimensions in the tensor.");
    CHECK_FAIL(self.defined() && values.defined(), "Both self and values tensors must be defined.");
    CHECK_FAIL(self.dtype() == values.dtype(), "Data types of self and values tensors must match.");
    CHECK_FAIL(!indices.empty() && (int)indices.size() <= KF_MAX_TENSORS - 2, "index_put_ supports 1..", KF_MAX_TENSORS - 2, " indexed dims");
    const int64_t es = self.element_size_in_bytes();
    std::vector<int64_t> sizes, strides;
    for (int d = 0; d < self.dim(); ++d) {
        sizes.push_back(self.shape(d));
        strides.push_back(self.stride(d) * es);
        CHECK_FAIL(self.shape(d) != 0, "index is out of bounds for dimension with size 0");
    }
    // self, seen through the index shape with stride 0: the kernel adds the gathered byte offset
    auto ishape = indices[0].sizes();
    Tensor target = self.as_strided(ishape, std::vector<int64_t>(ishape.size(), 0), self.storage_offset());
    TensorIterator iter;
    iter.add_output(target).check_mem_overlap(false).resize_outputs(false);
    iter.add_input(values);
    for (auto &ix : indices) {
        CHECK_FAIL(ix.defined() && ix.dtype() == ScalarType::Long, "Indices must be of type Long.");
        iter.add_input(ix);
    }
    iter.build();
    if (iter.numel() == 0) return self;
    void *stream = dev::stream(self.device());
    iter.geometry().for_each_32bit([&](const IterGeometry &g) {
        kf_iter_desc d;
        g.to_desc(d);
        DEV_CALL(kf_index_put(&d, (int)sizes.size(), sizes.data(), strides.data(), stream));
    });
    return self;
}

// ---- attention on the packed QKV projection (README.md:32) ---------------------------------------------------------------------
namespace {
struct PackedLay { kf_attn_layout qkv, flat; };
PackedLay packed_layouts(int64_t S, int64_t H, int64_t D) {
    const int64_t d = H * D;
    return {{S * 3 * d, D, 3 * d}, {S * d, D, d}};
}
bool packed_fast(const Tensor &qkv, int64_t S, int64_t D) {
    return (qkv.dtype() == ScalarType::Half || qkv.dtype() == ScalarType::BFloat16) && (D == 64 || D == 128) && S > 0;   // (any S since round 6: Sq == Skv)
}

class PackedAttentionGradFunction : public GradFunction {
public:
    PackedAttentionGradFunction(const Tensor &qkv, const Tensor &out, const Tensor &lse, int64_t B, int64_t S, int64_t H)
        : out_(out), lse_(lse), B_(B), S_(S), H_(H) {
        inputs = {qkv};
    }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &qkv = inputs[0];
        const int64_t d = qkv.shape(1) / 3, D = d / H_;
        const int es = (int)qkv.element_size_in_bytes();
        Tensor gc = g.dense();
        Tensor dqkv = empty(qkv.sizes(), qkv.dtype(), qkv.device());
        const PackedLay L = packed_layouts(S_, H_, D);
        size_t need = 0;
        DataPtr scratch = attn_bwd_scratch(code(qkv.dtype()), B_, H_, S_, S_, D, qkv.device(), need);
        const char *p = static_cast<const char *>(qkv.data_ptr());
        char *gp = static_cast<char *>(dqkv.data_ptr());
        DEV_CALL(kf_attn_bwd_strided(code(qkv.dtype()), B_, H_, S_, S_, D, 1.0f / std::sqrt((float)D), p, &L.qkv, p + d * es, &L.qkv, p + 2 * d * es,
                                     &L.qkv, out_.data_ptr(), &L.flat, static_cast<const float *>(lse_.data_ptr()), gc.data_ptr(), &L.flat, gp, &L.qkv,
                                     gp + d * es, &L.qkv, gp + 2 * d * es, &L.qkv, scratch.get(), need, dev::stream(qkv.device())));
        return {dqkv};
    }

private:
    Tensor out_, lse_;
    int64_t B_, S_, H_;
};
} // namespace

Tensor causal_attention_qkv(const Tensor &qkv, int64_t B, int64_t S, int64_t H) {
    CHECK_FAIL(qkv.defined() && qkv.dim() == 2 && qkv.is_dense(), "causal_attention_qkv expects a contiguous [B*S, 3*H*D] tensor");
    CHECK_FAIL(B > 0 && S > 0 && H > 0 && qkv.shape(0) == B * S && qkv.shape(1) % (3 * H) == 0, "causal_attention_qkv: shape does not match B, S, H");
    const int64_t d = qkv.shape(1) / 3, D = d / H;
    if (!packed_fast(qkv, S, D)) {
        // off the strided kernels' shapes: the same result from the reference's own operators (which carry their own autograd)
        auto parts = tensor_split(qkv, {d, d, d}, 1);
        std::vector<Tensor> heads;
        for (auto &t : parts) heads.push_back(t.dense().view({B, S, H, D}).permute({0, 2, 1, 3}).dense());
        Tensor a = causal_attention(heads[0], heads[1], heads[2]);
        return a.permute({0, 2, 1, 3}).dense().view({B * S, d});
    }
    const int es = (int)qkv.element_size_in_bytes();
    Tensor out = empty({B * S, d}, qkv.dtype(), qkv.device());
    Tensor lse = empty({B, H, S}, ScalarType::Float, qkv.device());
    const PackedLay L = packed_layouts(S, H, D);
    const char *p = static_cast<const char *>(qkv.data_ptr());
    DEV_CALL(kf_attn_fwd_strided(code(qkv.dtype()), B, H, S, S, D, 1.0f / std::sqrt((float)D), p, &L.qkv, p + d * es, &L.qkv, p + 2 * d * es, &L.qkv,
                                 out.data_ptr(), &L.flat, static_cast<float *>(lse.data_ptr()), dev::stream(qkv.device())));
    if (qkv.requires_grad()) {
        out.set_requires_grad(true);
        out.set_grad_fn(new PackedAttentionGradFunction(qkv, out, lse, B, S, H));
    }
    return out;
}

// ---- rms_norm / layer_norm (README.md:28; statistics as norm_ops_kernel.cu:6-61 / welford_norm.h:170-187) ---------------------
namespace {
bool norm_dtype_ok(ScalarType t) { return t == ScalarType::Float || t == ScalarType::Half || t == ScalarType::BFloat16; }

// saved for the backward: x, weight, and the forward's f32 statistics
class NormGradFunction : public GradFunction {
public:
    NormGradFunction(int kind, const Tensor &x, const Tensor &w, const Tensor &b, const Tensor &mean, const Tensor &rstd)
        : kind_(kind), has_w_(w.defined()), has_b_(b.defined()), mean_(mean), rstd_(rstd) {
        inputs = {x};
        if (has_w_) inputs.push_back(w);
        if (has_b_) inputs.push_back(b);
    }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &x = inputs[0];
        const int64_t cols = x.shape(-1), rows = x.numel() / cols;
        Tensor gc = g.dense();
        Tensor dx = empty(x.sizes(), x.dtype(), x.device());
        Tensor dw, db;
        const bool want_w = has_w_ && inputs[1].requires_grad(), want_b = has_b_ && inputs[has_w_ ? 2 : 1].requires_grad();
        if (want_w || want_b) dw = empty({cols}, x.dtype(), x.device()); // the kernel produces both sums in one sweep
        if (want_b) db = empty({cols}, x.dtype(), x.device());
        size_t need = 0;
        DEV_CALL(kf_norm_bwd_workspace_bytes(kind_, code(x.dtype()), rows, cols, cols, &need));
        DataPtr scratch;
        if (need && dw.defined()) scratch = DeviceAllocator::GetInstance()->allocate(need, x.device());
        DEV_CALL(kf_norm_bwd(kind_, code(x.dtype()), rows, cols, cols, x.data_ptr(), has_w_ ? inputs[1].data_ptr() : nullptr,
                             kind_ == KF_NORM_LAYER ? static_cast<const float *>(mean_.data_ptr()) : nullptr,
                             static_cast<const float *>(rstd_.data_ptr()), gc.data_ptr(), dx.data_ptr(), dw.defined() ? dw.data_ptr() : nullptr,
                             db.defined() ? db.data_ptr() : nullptr, scratch.get(), dw.defined() ? need : 0, dev::stream(x.device())));
        std::vector<Tensor> out(inputs.size());
        if (x.requires_grad()) out[0] = dx;
        if (want_w) out[1] = dw;
        if (want_b) out[has_w_ ? 2 : 1] = db;
        return out;
    }

private:
    int kind_;
    bool has_w_, has_b_;
    Tensor mean_, rstd_;
};

Tensor norm_impl(int kind, const Tensor &x, const Tensor &w, const Tensor &b, double eps) {
    CHECK_FAIL(x.defined() && x.dim() >= 1 && x.is_dense(), "norm expects a contiguous tensor");
    CHECK_FAIL(norm_dtype_ok(x.dtype()), "norm supports float, half and bfloat16");
    const int64_t cols = x.shape(-1);
    CHECK_FAIL(cols > 0);
    const int64_t rows = x.numel() / cols;
    for (const Tensor *p : {&w, &b})
        if (p->defined()) CHECK_FAIL(p->dim() == 1 && p->shape(0) == cols && p->dtype() == x.dtype() && p->is_dense() && p->device() == x.device(),
                                     "norm weight / bias must be contiguous 1-D tensors of the normalised length and of x's dtype");
    Tensor y = empty(x.sizes(), x.dtype(), x.device());
    const bool grad = x.requires_grad() || (w.defined() && w.requires_grad()) || (b.defined() && b.requires_grad());
    Tensor mean, rstd;
    if (grad) {
        rstd = empty({rows}, ScalarType::Float, x.device());
        if (kind == KF_NORM_LAYER) mean = empty({rows}, ScalarType::Float, x.device());
    }
    DEV_CALL(kf_norm_fwd(kind, code(x.dtype()), rows, cols, cols, x.data_ptr(), w.defined() ? w.data_ptr() : nullptr,
                         b.defined() ? b.data_ptr() : nullptr, eps, y.data_ptr(), mean.defined() ? static_cast<float *>(mean.data_ptr()) : nullptr,
                         rstd.defined() ? static_cast<float *>(rstd.data_ptr()) : nullptr, dev::stream(x.device())));
    if (grad) {
        y.set_requires_grad(true);
        y.set_grad_fn(new NormGradFunction(kind, x, w, b, mean, rstd));
    }
    return y;
}
} // namespace

Tensor rms_norm(const Tensor &x, const Tensor &weight, double eps) { return norm_impl(KF_NORM_RMS, x, weight, Tensor(), eps); }
Tensor layer_norm(const Tensor &x, const Tensor &weight, const Tensor &bias, double eps) { return norm_impl(KF_NORM_LAYER, x, weight, bias, eps); }

// ---- embedding (README.md:30; index arithmetic of tensor_index.h:56-104) -------------------------------------------------------
namespace {
class EmbeddingGradFunction : public GradFunction { // dTable[r] = sum of the gradient rows gathered from r, in input order
public:
    EmbeddingGradFunction(const Tensor &table, const Tensor &indices) : indices_(indices) { inputs = {table}; }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &table = inputs[0];
        const int64_t nrows = table.shape(0), cols = table.shape(1), n = indices_.numel();
        Tensor gc = g.dense();
        Tensor dt = zeros(table.sizes(), table.dtype(), table.device());
        const size_t need = kf_index_add_workspace_bytes(n);
        DataPtr scratch;
        if (need) scratch = DeviceAllocator::GetInstance()->allocate(need, table.device());
        DEV_CALL(kf_index_add(code(table.dtype()), static_cast<const int64_t *>(indices_.data_ptr()), n, gc.data_ptr(), cols, nrows,
                              dt.data_ptr(), scratch.get(), need, dev::stream(table.device())));
        return {dt};
    }

private:
    Tensor indices_;
};
} // namespace

Tensor embedding(const Tensor &table, const Tensor &indices) {
    CHECK_FAIL(table.defined() && table.dim() == 2 && table.is_dense(), "embedding expects a contiguous 2-D table");
    CHECK_FAIL(indices.defined() && indices.dtype() == ScalarType::Long, "Indices must be of type Long.");
    CHECK_FAIL(indices.device() == table.device());
    Tensor ix = indices.dense();
    auto shape = ix.sizes();
    shape.push_back(table.shape(1));
    Tensor out = empty(shape, table.dtype(), table.device());
    DEV_CALL(kf_index_get(table.data_ptr(), table.shape(0), table.shape(1) * table.element_size_in_bytes(),
                          static_cast<const int64_t *>(ix.data_ptr()), ix.numel(), out.data_ptr(), dev::stream(table.device())));
    if (table.requires_grad()) {
        CHECK_FAIL(norm_dtype_ok(table.dtype()), "embedding backward supports float, half and bfloat16 tables");
        out.set_requires_grad(true);
        out.set_grad_fn(new EmbeddingGradFunction(table, ix));
    }
    return out;
}

// ---- shape ops (tensor_shape.cpp:41-89) ----------------------------------------------------------------------------
namespace {
class CatGradFunction : public GradFunction { // each input's gradient is its window of g
public:
    CatGradFunction(const std::vector<Tensor> &ts, int dim) : dim_(dim) { inputs = ts; }
    std::vector<Tensor> backward(Tensor g) override {
        std::vector<Tensor> out(inputs.size());
        int64_t at = 0;
        for (size_t i = 0; i < inputs.size(); ++i) {
            const int64_t len = inputs[i].shape(dim_);
            if (inputs[i].requires_grad()) out[i] = g.narrow(dim_, at, len);
            at += len;
        }
        return out;
    }

private:
    int dim_;
};
} // namespace

Tensor concat(const std::vector<Tensor> tensors, int64_t dim) {
    CHECK_FAIL(!tensors.empty(), "concat expects a non-empty list");
    const Tensor &first = tensors[0];
    const int d = maybe_wrap_dim((int)dim, first.dim());
    int64_t total = 0;
    for (size_t i = 0; i < tensors.size(); ++i) {
        const Tensor &t = tensors[i];
        CHECK_FAIL(t.device() == first.device());
        CHECK_FAIL(t.dim() == first.dim(), "Tensors must have same number of dimensions: got ", first.dim(), " and ", t.dim());
        for (int k = 0; k < first.dim(); ++k)
            CHECK_FAIL(k == d || t.shape(k) == first.shape(k), "Sizes of tensors must match except in dimension ", d, ". Expected size ",
                       first.shape(k), " but got size ", t.shape(k), " for tensor number ", i, " in the list.");
        total += t.shape(d);
    }
    auto out_size = first.sizes();
    out_size[d] = total;
    Tensor result = empty(out_size, first.dtype(), first.device());
    int64_t at = 0;
    bool any_grad = false;
    for (const Tensor &t : tensors) {
        Tensor window = result.narrow(d, at, t.shape(d));
        window.copy_(t);
        at += t.shape(d);
        any_grad = any_grad || t.requires_grad();
    }
    if (any_grad) {
        result.set_requires_grad(true);
        result.set_grad_fn(new CatGradFunction(tensors, d));
    }
    return result;
}

std::vector<Tensor> tensor_split(const Tensor &self, std::vector<int64_t> indices, int64_t dim) {
    CHECK_FAIL(self.dim() > 0, "tensor_split expected at least a 1-dimensional tensor, but got a tensor with ", self.dim(), " dims");
    const int d = maybe_wrap_dim((int)dim, self.dim());
    std::vector<Tensor> parts;
    int64_t at = 0;
    for (int64_t len : indices) {
        parts.push_back(self.slice(d, at, at + len));
        at += len;
    }
    CHECK_FAIL(at == self.shape(d));
    return parts;
}

} // namespace gpu
