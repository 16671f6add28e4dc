// Tensor / TensorImpl / TensorStorage / GradFunction: the C++ value-handle API the reference exposes
// (src/core/include/tensor.h:10-165, tensor_impl.h:62-214), re-implemented over the HIP C ABI.
// Names, arities and semantics are kept so that a register.cpp-equivalent compiles against this
// header unchanged; internals are new (exact contiguity tracking, std::array metadata, per-device
// allocator, device work on an explicit stream).
#pragma once

#include <array>
#include <cstring>
#include <memory>
#include <optional>
#include <ostream>
#include <sstream>
#include <string>
#include <tuple>
#include <vector>

#include "allocator.h"
#include "check.h"
#include "refptr.h"
#include "scalar_type.h"

#define MAX_TENSOR_DIMS 12

using utils::memory::intrusive_ptr;
using utils::memory::intrusive_ptr_target;

inline int maybe_wrap_dim(int d, int ndim) { return d < 0 ? (ndim + d) % ndim : d; }

// fixed-capacity dim vector (reference: d_array<int64_t, 12>, tensor_impl.h:20-58)
struct dim_t {
    std::array<int64_t, MAX_TENSOR_DIMS> val{};
    int64_t &operator[](int i) { return val[i]; }
    const int64_t &operator[](int i) const { return val[i]; }
    bool equals(const dim_t &o) const { return val == o.val; }
    bool equals(const int64_t (&o)[MAX_TENSOR_DIMS]) const {
        for (int i = 0; i < MAX_TENSOR_DIMS; ++i)
            if (val[i] != o[i]) return false;
        return true;
    }
};
std::ostream &operator<<(std::ostream &os, const dim_t &d);

// opaque scalar carrier: fill_ passes a double in it, item() returns one raw element in it
struct any_t {
    alignas(8) char val[16] = {0};
    any_t() = default;
    any_t(double d) { std::memcpy(val, &d, sizeof(d)); }
    operator double() const {
        double d;
        std::memcpy(&d, val, sizeof(d));
        return d;
    }
};

class TensorStorage : public intrusive_ptr_target {
public:
    TensorStorage(size_t bytes, int device);
    size_t size() const { return size_; }
    int device() const { return device_; }
    void *data_ptr() const { return ptr_.get(); }
    bool defined() const { return static_cast<bool>(ptr_); }

private:
    size_t size_;
    int device_;
    utils::memory::DataPtr ptr_;
};

class Tensor;
class TensorImpl;

// Where a leaf's gradient lands when the leaf belongs to a gradient bucket (gpu::GradBucket, comm.h): the bucket hands out the leaf's
// slot - a view into ONE flat buffer - and is told when the leaf's gradient of this backward pass is complete, which is what lets it
// start the all-reduce of a finished chunk while the rest of the backward is still running. No reference counterpart (the reference
// has no distributed code: SURVEY.md fact 5).
struct GradSink {
    virtual ~GradSink() = default;
    virtual Tensor slot(TensorImpl *leaf) = 0;     // the leaf's gradient storage inside the bucket
    // The slot for a backward function to WRITE its gradient into directly (beta = 0), handed out at most ONCE per backward pass: a
    // weight that feeds two products of one graph (weight tying) gets the slot for the first of them and an undefined tensor - "allocate
    // your own" - for the second, so the engine's fan-in sum sees two different tensors (two beta = 0 writes into the
    // same slot left 2 dW_last instead of dW_1 + dW_2).
    virtual Tensor take_slot(TensorImpl *leaf) = 0;
    virtual void arrived(TensorImpl *leaf) = 0;    // the leaf's gradient of this pass is in its slot
};

class TensorImpl : public intrusive_ptr_target {
public:
    TensorImpl(const std::vector<int64_t> &shape, ScalarType dtype);
    TensorImpl(const std::vector<int64_t> &shape, const std::vector<int64_t> &strides, ScalarType dtype);
    TensorImpl(const TensorImpl &other); // shares storage, drops grad (a view starts without one)

    int dim() const { return dim_; }
    int64_t shape(int d) const { return shape_[maybe_wrap_dim(d, dim_)]; }
    dim_t &shape() { return shape_; }
    const dim_t &shape() const { return shape_; }
    int64_t stride(int d) const { return stride_[d]; }
    dim_t &stride() { return stride_; }
    const dim_t &stride() const { return stride_; }
    std::vector<int64_t> sizes() const { return {shape_.val.begin(), shape_.val.begin() + dim_}; }
    std::vector<int64_t> strides() const { return {stride_.val.begin(), stride_.val.begin() + dim_}; }
    ScalarType dtype() const { return dtype_; }
    int64_t numel() const { return numel_; }
    void *data_ptr() const { return static_cast<char *>(storage_->data_ptr()) + storage_offset_ * (int64_t)element_size(dtype_); }
    size_t storage_bytes() const { return storage_->size(); }
    size_t storage_ref_count() const { return storage_.ref_count(); }
    int64_t storage_offset() const { return storage_offset_; }
    intrusive_ptr<TensorStorage> storage() const { return storage_; }
    bool defined() const { return storage_.get() != nullptr; }
    int device() const { return storage_->device(); }
    int64_t element_size_in_bytes() const { return (int64_t)element_size(dtype_); }
    // The reference's flag, with the reference's meaning (tensor_impl.h:108, tensor_impl.cpp:95): TRUE for a freshly allocated tensor, FALSE for whatever
    // as_strided_ made - every view, slice, select, permute, split part and the result of view() itself, dense or not. It decides what contiguous() returns:
    // a CLONE exactly when it is false, so that the result never aliases a view (x.view(6).contiguous() += 1 leaves x alone there, and here). Where the reference
    // merely REFUSES a tensor because of the flag (view(), numpy(), gemm) this host asks is_dense() instead: a superset with the same results.
    bool is_contiguous() const { return is_contiguous_; }
    // What the strides say: the elements lie in row-major order without gaps (size-1 dims never break it). The kernels' own precondition: the operators this
    // repository adds and every internal fast path ask THIS, not the flag.
    bool is_dense() const { return is_dense_; }
    bool requires_grad() const { return requires_grad_; }
    void set_requires_grad(bool f) { requires_grad_ = f; }

    void new_storage_(int device);
    void as_strided_(const std::vector<int64_t> &sizes, const std::vector<int64_t> &strides, int64_t storage_offset);

    std::unique_ptr<Tensor> grad_; // accumulated gradient of a leaf
    std::weak_ptr<GradSink> sink_;   // set by GradBucket::attach: gradients are written into the bucket (weak: the bucket owns its parameters, not the reverse)

private:
    void refresh_();
    int dim_ = 0;
    dim_t shape_, stride_;
    ScalarType dtype_ = ScalarType::Undefined;
    int64_t numel_ = 0;
    intrusive_ptr<TensorStorage> storage_;
    int64_t storage_offset_ = 0;
    bool is_contiguous_ = true;
    bool is_dense_ = true;
    bool requires_grad_ = false;
};

Tensor empty(std::vector<int64_t> shape, ScalarType dtype, int device = 0);
Tensor empty(const int64_t *shape, int ndim, ScalarType dtype, int device, bool inverse = false);
Tensor empty_like(const Tensor &self);
Tensor empty_strided(std::vector<int64_t> shape, std::vector<int64_t> strides, ScalarType dtype, int device);
Tensor empty_like_reduced(const Tensor &self, int dim, ScalarType dtype);
Tensor zeros(std::vector<int64_t> shape, ScalarType dtype, int device = 0);
std::ostream &operator<<(std::ostream &os, const Tensor &t);

class GradFunction : public intrusive_ptr_target {
public:
    virtual std::vector<Tensor> backward(Tensor grad_output) = 0;
    std::vector<Tensor> inputs;
};

class Tensor {
public:
    Tensor() = default;

    TensorImpl *impl() const { return impl_.get(); }
    int dim() const { return impl_->dim(); }
    int64_t shape(int d) const { return impl_->shape(d); }
    dim_t &shape() { return impl_->shape(); }
    const dim_t &shape() const { return impl_->shape(); }
    std::vector<int64_t> sizes() const { return impl_->sizes(); }
    std::vector<int64_t> strides() const { return impl_->strides(); }
    int64_t stride(int d) const { return impl_->stride(d); }
    dim_t &stride() { return impl_->stride(); }
    const dim_t &stride() const { return impl_->stride(); }
    ScalarType dtype() const { return impl_->dtype(); }
    int64_t numel() const { return impl_->numel(); }
    void *data_ptr() const { return impl_->data_ptr(); }
    template <typename T> T *data_ptr() const { return static_cast<T *>(impl_->data_ptr()); }
    size_t storage_bytes() const { return impl_->storage_bytes(); }
    size_t storage_ref_count() const { return impl_->storage_ref_count(); }
    size_t impl_ref_count() const { return impl_.ref_count(); }
    int64_t storage_offset() const { return impl_->storage_offset(); }
    intrusive_ptr<TensorStorage> storage() const { return impl_->storage(); }
    intrusive_ptr<GradFunction> grad_fn() const { return grad_fn_; }
    bool defined() const { return impl_.get() && impl_->defined(); }
    bool has_grad_fn() const { return grad_fn_.get() != nullptr; }
    int device() const { return impl_->device(); }
    int64_t element_size_in_bytes() const { return impl_->element_size_in_bytes(); }
    bool is_contiguous() const { return impl_->is_contiguous(); }   // the reference's flag (see TensorImpl)
    bool is_dense() const { return impl_->is_dense(); }             // row-major without gaps, by the strides
    bool requires_grad() const { return impl_->requires_grad(); }
    void set_requires_grad(bool flag) { impl_->set_requires_grad(flag); }
    Tensor *grad() { return impl_->grad_.get(); }
    std::string to_string() const;

    void set_grad_fn(GradFunction *fn) { grad_fn_.unsafe_set_ptr(fn); }
    void update_grad(Tensor grad);
    void backward(Tensor grad_output);
    void copy_from_cpu_ptr(void *ptr);
    void copy_to_cpu_ptr(void *ptr) const;
    any_t item(const std::vector<int64_t> &indices) const;
    Tensor &fill_(const any_t &value);
    int64_t offset(const std::vector<int64_t> &indices) const;
    Tensor contiguous() const;                              // *this when the flag is set, else a clone (tensor.cpp:161-165 of the reference)
    Tensor dense() const;                                   // *this when the strides are dense, else a clone: what the kernels need (no copy of a dense view)
    Tensor as_strided(std::vector<int64_t> sizes, std::vector<int64_t> strides, int64_t storage_offset = 0) const;
    Tensor permute(const std::vector<int64_t> dims) const;
    Tensor slice(int64_t dim, std::optional<int64_t> start, std::optional<int64_t> end, int64_t step = 1) const;
    Tensor select(int64_t dim, int64_t index) const;
    Tensor narrow(int64_t dim, int64_t start, int64_t length) const;
    Tensor view(std::vector<int64_t> sizes) const;
    bool can_use_32bit_indexing() const;
    std::vector<Tensor> split(std::vector<int64_t> indices, int64_t dim) const;

    Tensor _half() const;
    Tensor _bfloat16() const;
    Tensor _float() const;

    Tensor operator+(const Tensor &other) const;
    Tensor &operator+=(const Tensor &other);
    Tensor operator-(const Tensor &other) const;
    Tensor &operator-=(const Tensor &other);
    Tensor operator*(const Tensor &other) const;
    Tensor &operator*=(const Tensor &other);
    Tensor operator/(const Tensor &other) const;
    Tensor &operator/=(const Tensor &other);
    Tensor &copy_(const Tensor &other);
    Tensor sum(int64_t reduce_dim) const;
    Tensor mean(int64_t reduce_dim) const;
    std::tuple<Tensor, Tensor> sort(int64_t dim, bool descending) const;
    std::tuple<Tensor, Tensor> topk(int64_t k, int64_t dim, bool largest) const;
    std::tuple<Tensor, Tensor> mean_var(int64_t reduce_dim, bool take_sqrt) const;
    std::tuple<Tensor, Tensor> norm_stat(int64_t dim) const;
    Tensor &index_put_(const std::vector<Tensor> &indices, const Tensor &values);

private:
    friend Tensor make_tensor_(TensorImpl *impl);
    intrusive_ptr<TensorImpl> impl_;
    intrusive_ptr<GradFunction> grad_fn_;
};

Tensor make_tensor_(TensorImpl *impl);
