// TensorIterator: broadcast + dtype promotion + dim reorder/coalesce + output allocation + 32-bit
// splitting for elementwise and reduction ops, producing the POD descriptor the C ABI consumes.
// Behavioural reference: src/core/include/tensor_iterator.h:21-239, src/core/tensor_iterator.cpp.
//
// Structure differs from the reference: the geometry engine (IterGeometry) works on plain operand
// records and can be driven without any device memory — that is how the CPU-only tests check it —
// while TensorIterator is the Tensor-facing builder with the reference's fluent API.
#pragma once

#include <functional>
#include <memory>
#include <ostream>
#include <vector>

#include "kfunca_hip.h"
#include "tensor.h"

// one operand as the geometry engine sees it
struct IterOperand {
    bool defined = false;
    bool is_output = false;
    bool is_read_write = false;  // an output that is also an input (in-place op)
    bool will_resize = false;    // output to be (re)allocated with the broadcast shape
    int ndim = 0;
    int64_t shape[MAX_TENSOR_DIMS] = {0};
    int64_t stride[MAX_TENSOR_DIMS] = {0}; // elements
    ScalarType dtype = ScalarType::Undefined;
    int device = -1;
    char *data = nullptr;
    const void *identity = nullptr; // in-place detection is by object identity (tensor_iterator.cpp:78-91)
};

class IterGeometry {
public:
    enum { MAX_TENSORS = KF_MAX_TENSORS };
    // called when output `arg` must be allocated: fills op.data / op.stride for a fresh contiguous
    // tensor of `shape` (in the ORIGINAL dim order)
    using Allocator = std::function<void(int arg, const int64_t *shape, int ndim, ScalarType dtype, int device, IterOperand &op)>;

    int add(const IterOperand &op, bool is_output);
    void build(bool is_reduction, int64_t reduce_dim, bool resize_outputs, bool check_mem_overlap, const Allocator &alloc);

    // post-build state
    int ndim() const { return ndim_; }
    int ntensors() const { return (int)ops_.size(); }
    int noutputs() const { return noutputs_; }
    int ninputs() const { return ntensors() - noutputs_; }
    int64_t shape(int d) const { return shape_[d]; }
    int64_t stride_bytes(int arg, int d) const { return stride_bytes_[arg][d]; }
    int64_t perm(int d) const { return perm_[d]; }
    char *data(int arg) const { return data_[arg]; }
    const IterOperand &operand(int arg) const { return ops_[arg]; }
    ScalarType common_dtype() const { return common_dtype_; }
    int common_device() const { return common_device_; }
    int64_t numel() const;
    int64_t num_output_elements() const;
    bool can_use_32bit_indexing() const;
    bool is_contiguous() const;
    bool is_dim_reduced(int d) const;
    int dim_to_split() const;

    // narrow dim `d` to [start, start+size) (used by the 32-bit split)
    void narrow(int d, int64_t start, int64_t size);
    // visit 32-bit-indexable pieces (reference SplitUntil32Bit, tensor_iterator.h:194-237)
    void for_each_32bit(const std::function<void(const IterGeometry &)> &fn) const;
    void to_desc(kf_iter_desc &d) const;

private:
    void coalesce();
    std::vector<IterOperand> ops_;
    int noutputs_ = 0;
    int ndim_ = 0;
    bool is_reduction_ = false;
    int64_t shape_[MAX_TENSOR_DIMS] = {0};
    int64_t perm_[MAX_TENSOR_DIMS] = {0};
    int64_t stride_bytes_[MAX_TENSORS][MAX_TENSOR_DIMS] = {{0}};
    char *data_[MAX_TENSORS] = {nullptr};
    ScalarType common_dtype_ = ScalarType::Undefined;
    int common_device_ = -1;
};

struct SplitUntil32Bit;

class TensorIterator final {
public:
    TensorIterator() = default;
    TensorIterator &add_output(Tensor &output);
    TensorIterator &add_input(const Tensor &input);
    TensorIterator &add_output(Tensor &&output) = delete;
    TensorIterator &add_input(Tensor &&input) = delete;
    TensorIterator &resize_outputs(bool flag) { resize_outputs_ = flag; return *this; }
    TensorIterator &check_mem_overlap(bool flag) { check_mem_overlap_ = flag; return *this; }
    TensorIterator &build();
    TensorIterator &build_for_loops();
    TensorIterator &build_for_reduce(int64_t reduce_dim);

    int64_t numel() const { return geo_.numel(); }
    int64_t num_output_elements() const { return geo_.num_output_elements(); }
    bool can_use_32bit_indexing() const { return geo_.can_use_32bit_indexing(); }
    bool is_contiguous() const { return geo_.is_contiguous(); }
    int ntensors() const { return geo_.ntensors(); }
    int noutputs() const { return geo_.noutputs(); }
    int ninputs() const { return geo_.ninputs(); }
    int ndim() const { return geo_.ndim(); }
    int dim() const { return geo_.ndim(); }
    int64_t shape(int d) const { return geo_.shape(maybe_wrap_dim(d, geo_.ndim())); }
    int64_t stride_bytes(int arg, int d) const { return geo_.stride_bytes(arg, d); }
    int64_t perm(int d) const { return geo_.perm(d); }
    void *data_ptr(int arg) const { return geo_.data(arg); }
    int device(int arg = 0) const { return tensors_[arg]->device(); }
    const Tensor &tensor(int arg) const { return *tensors_[arg]; }
    Tensor &outputs(int arg) { return *tensors_[arg]; }
    ScalarType dtype(int arg = 0) const { return tensors_[arg]->dtype(); }
    ScalarType input_dtype(int arg = 0) const { return tensors_[geo_.noutputs() + arg]->dtype(); }
    ScalarType common_dtype() const {
        CHECK_FAIL(geo_.common_dtype() != ScalarType::Undefined, "Queried for invalid common dtype!");
        return geo_.common_dtype();
    }
    const IterGeometry &geometry() const { return geo_; }

private:
    std::vector<Tensor *> tensors_;
    int num_outputs_ = 0;
    bool resize_outputs_ = true;
    bool check_mem_overlap_ = true;
    bool is_reduction_ = false;
    int64_t reduce_dim_ = 0;
    IterGeometry geo_;
};

std::ostream &operator<<(std::ostream &os, const TensorIterator &iter);
