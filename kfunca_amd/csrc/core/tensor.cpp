// Tensor metadata, views, host<->device copies, printing and the reverse-mode engine.
// Behavioural reference: src/core/tensor.cpp, tensor_impl.cpp (cited per function).
#include "tensor.h"

#include <algorithm>

#include <iomanip>
#include <limits>
#include <queue>
#include <unordered_map>
#include <unordered_set>

#include "device_api.h"
#include "ops.h"

std::ostream &operator<<(std::ostream &os, const dim_t &d) {
    os << "dim_t:";
    for (int i = 0; i < MAX_TENSOR_DIMS; ++i) os << d[i] << ", ";
    return os << "\n";
}

// ---- storage / impl -------------------------------------------------------------------------
TensorStorage::TensorStorage(size_t bytes, int device) : size_(bytes), device_(device) {
    ptr_ = utils::memory::DeviceAllocator::GetInstance()->allocate(bytes, device);
}

static std::pair<int64_t, int64_t> offset_range(const int64_t *shape, const int64_t *stride, int ndim) {
    int64_t lo = 0, hi = 0; // element offsets touched by the view (reference memory_overlap.h:30-44)
    for (int i = 0; i < ndim; ++i) {
        if (shape[i] == 0) return {0, -1};
        const int64_t span = (shape[i] - 1) * stride[i];
        (span >= 0 ? hi : lo) += span;
    }
    return {lo, hi};
}

TensorImpl::TensorImpl(const std::vector<int64_t> &shape, ScalarType dtype) : dtype_(dtype) {
    CHECK_FAIL(shape.size() <= MAX_TENSOR_DIMS);
    dim_ = (int)shape.size();
    int64_t run = 1;
    for (int i = dim_ - 1; i >= 0; --i) {
        shape_[i] = shape[i];
        stride_[i] = run;
        run *= shape[i];
    }
    refresh_();
}

TensorImpl::TensorImpl(const std::vector<int64_t> &shape, const std::vector<int64_t> &strides, ScalarType dtype) : dtype_(dtype) {
    CHECK_FAIL(shape.size() <= MAX_TENSOR_DIMS);
    CHECK_FAIL(shape.size() == strides.size());
    dim_ = (int)shape.size();
    for (int i = 0; i < dim_; ++i) {
        shape_[i] = shape[i];
        stride_[i] = strides[i];
    }
    refresh_();
}

TensorImpl::TensorImpl(const TensorImpl &o)
    : intrusive_ptr_target(), dim_(o.dim_), shape_(o.shape_), stride_(o.stride_), dtype_(o.dtype_), numel_(o.numel_),
      storage_(o.storage_), storage_offset_(o.storage_offset_), is_contiguous_(o.is_contiguous_), is_dense_(o.is_dense_), requires_grad_(o.requires_grad_) {}

void TensorImpl::refresh_() { // numel + exact density (size-1 dims never break it); the is_contiguous_ FLAG is not derived from the strides (tensor.h)
    numel_ = 1;
    for (int i = 0; i < dim_; ++i) numel_ *= shape_[i];
    int64_t expect = 1;
    is_dense_ = true;
    for (int i = dim_ - 1; i >= 0; --i) {
        if (shape_[i] != 1 && stride_[i] != expect) is_dense_ = false;
        expect *= shape_[i];
    }
    for (int i = dim_; i < MAX_TENSOR_DIMS; ++i) shape_[i] = stride_[i] = 0;
}

void TensorImpl::new_storage_(int device) {
    CHECK_FAIL(!(storage_.get() && storage_->defined()));
    auto [lo, hi] = offset_range(shape_.val.data(), stride_.val.data(), dim_);
    const size_t elems = hi >= lo ? (size_t)(hi - lo + 1) : 0;
    storage_.unsafe_set_ptr(new TensorStorage(elems * element_size(dtype_), device));
}

void TensorImpl::as_strided_(const std::vector<int64_t> &sizes, const std::vector<int64_t> &strides_in, int64_t storage_offset) {
    const int ndim = (int)sizes.size();
    CHECK_FAIL(ndim <= MAX_TENSOR_DIMS);
    std::vector<int64_t> strides = strides_in;
    if (strides.empty()) { // contiguous strides for the requested sizes (reference tensor_impl.cpp:73-78)
        strides.assign(ndim, 1);
        int64_t run = 1;
        for (int i = ndim - 1; i >= 0; --i) {
            strides[i] = run;
            run *= sizes[i];
        }
    }
    CHECK_FAIL(ndim == (int)strides.size());
    auto [lo, hi] = offset_range(sizes.data(), strides.data(), ndim);
    if (hi >= lo) { // in-bounds check of the whole view (tensor_impl.cpp:80-85)
        CHECK_FAIL(lo + storage_offset >= 0);
        CHECK_FAIL((hi + storage_offset) * (int64_t)element_size(dtype_) < (int64_t)storage_bytes());
    }
    dim_ = ndim;
    for (int i = 0; i < ndim; ++i) {
        shape_[i] = sizes[i];
        stride_[i] = strides[i];
    }
    storage_offset_ = storage_offset;
    refresh_();
    is_contiguous_ = false; // whatever the strides are (reference tensor_impl.cpp:95): found by tests/test_gpu_host_diff_fuzz.py - contiguous() of a dense view CLONES there
}

// ---- factories --------------------------------------------------------------------------------
Tensor make_tensor_(TensorImpl *impl) {
    Tensor t;
    t.impl_.unsafe_set_ptr(impl);
    return t;
}

Tensor empty(std::vector<int64_t> shape, ScalarType dtype, int device) {
    auto impl = new TensorImpl(shape, dtype);
    Tensor t = make_tensor_(impl);
    impl->new_storage_(device);
    return t;
}

Tensor empty(const int64_t *shape, int ndim, ScalarType dtype, int device, bool inverse) {
    std::vector<int64_t> s(ndim);
    for (int i = 0; i < ndim; ++i) s[i] = shape[inverse ? ndim - 1 - i : i];
    return empty(s, dtype, device);
}

Tensor empty_like(const Tensor &self) { return empty(self.sizes(), self.dtype(), self.device()); }

Tensor empty_strided(std::vector<int64_t> shape, std::vector<int64_t> strides, ScalarType dtype, int device) {
    auto impl = new TensorImpl(shape, strides, dtype);
    Tensor t = make_tensor_(impl);
    impl->new_storage_(device);
    return t;
}

Tensor empty_like_reduced(const Tensor &self, int dim, ScalarType dtype) {
    auto sizes = self.sizes();
    if (dim >= 0) sizes[dim] = 1;
    return empty(sizes, dtype, self.device());
}

Tensor zeros(std::vector<int64_t> shape, ScalarType dtype, int device) {
    Tensor t = empty(shape, dtype, device);
    DEV_CALL(kf_memset_zero(t.data_ptr(), t.storage_bytes(), dev::stream(device)));
    return t;
}

// ---- host <-> device ----------------------------------------------------------------------------
void Tensor::copy_from_cpu_ptr(void *ptr) { DEV_CALL(kf_memcpy_h2d(data_ptr(), ptr, storage_bytes(), dev::stream(device()))); }
void Tensor::copy_to_cpu_ptr(void *ptr) const { DEV_CALL(kf_memcpy_d2h(ptr, data_ptr(), storage_bytes(), dev::stream(device()))); }

int64_t Tensor::offset(const std::vector<int64_t> &indices) const {
    CHECK_FAIL((int)indices.size() == dim());
    int64_t o = 0;
    for (size_t i = 0; i < indices.size(); ++i) o += indices[i] * stride((int)i);
    return o;
}

any_t Tensor::item(const std::vector<int64_t> &indices) const {
    any_t buf;
    const int64_t es = element_size_in_bytes();
    DEV_CALL(kf_memcpy_d2h(buf.val, static_cast<char *>(data_ptr()) + offset(indices) * es, (size_t)es, dev::stream(device())));
    return buf;
}

Tensor &Tensor::fill_(const any_t &value) { return gpu::fill_(*this, value); }
Tensor Tensor::contiguous() const { return is_contiguous() ? *this : gpu::clone(*this); }
Tensor Tensor::dense() const { return is_dense() ? *this : gpu::clone(*this); }

// ---- views: pure metadata, bit-exact by construction (reference tensor.cpp:167-290) ------------
namespace {
// Backward of every view (permute / slice / select / narrow / view / split all funnel through as_strided): the gradient
// of the base is a buffer laid out like the base's storage, zero outside the view, holding g inside it. The reference's
// autograd stops at add (binary_ops.cpp:16-33); this is what lets a whole block (config C5) run through Tensor::backward.
class ViewGradFunction : public GradFunction {
public:
    ViewGradFunction(const Tensor &base, std::vector<int64_t> sizes, std::vector<int64_t> strides, int64_t offset)
        : sizes_(std::move(sizes)), strides_(std::move(strides)), offset_(offset) {
        inputs = {base};
    }
    std::vector<Tensor> backward(Tensor g) override {
        const Tensor &base = inputs[0];
        const int64_t n = (int64_t)(base.storage_bytes() / (size_t)base.element_size_in_bytes());
        // zero-fill unless the view provably covers every element of the storage exactly once: numel == n alone does not
        // (an overlapping or stride-0 as_strided view can have numel == n and still leave holes)
        int64_t numel = 1;
        for (int64_t v : sizes_) numel *= v;
        bool dense = numel == n && offset_ == 0;
        if (dense) { // a permutation of a contiguous layout: sorted by stride, each stride = the product of the smaller extents
            std::vector<std::pair<int64_t, int64_t>> ds; // (stride, size), size-1 dims dropped
            for (size_t i = 0; i < sizes_.size(); ++i)
                if (sizes_[i] != 1) ds.emplace_back(strides_[i], sizes_[i]);
            std::sort(ds.begin(), ds.end());
            int64_t expect = 1;
            for (auto &d : ds) {
                if (d.first != expect) { dense = false; break; }
                expect *= d.second;
            }
        }
        Tensor buf = dense ? empty({n}, base.dtype(), base.device()) : zeros({n}, base.dtype(), base.device());
        Tensor gv = buf.as_strided(sizes_, strides_, offset_);
        gv.copy_(g);
        return {buf.as_strided(base.sizes(), base.strides(), base.storage_offset())};
    }

private:
    std::vector<int64_t> sizes_, strides_;
    int64_t offset_;
};
} // namespace

Tensor Tensor::as_strided(std::vector<int64_t> sizes, std::vector<int64_t> strides, int64_t storage_offset) const {
    auto impl = new TensorImpl(*impl_.get());
    Tensor out = make_tensor_(impl);
    impl->as_strided_(sizes, strides, storage_offset);
    if (requires_grad()) out.set_grad_fn(new ViewGradFunction(*this, out.sizes(), out.strides(), out.storage_offset()));
    return out;
}

Tensor Tensor::permute(const std::vector<int64_t> dims) const {
    const int nd = dim();
    CHECK_FAIL(nd == (int)dims.size());
    std::vector<int64_t> sizes(nd), strides(nd);
    std::vector<bool> seen(nd, false);
    for (int i = 0; i < nd; ++i) {
        const int d = maybe_wrap_dim((int)dims[i], nd);
        CHECK_FAIL(!seen[d], "permute(): duplicate dims are not allowed.");
        seen[d] = true;
        sizes[i] = shape(d);
        strides[i] = stride(d);
    }
    return as_strided(sizes, strides, storage_offset());
}

Tensor Tensor::slice(int64_t dim, std::optional<int64_t> start, std::optional<int64_t> end, int64_t step) const {
    const int d = maybe_wrap_dim((int)dim, this->dim());
    CHECK_FAIL(step > 0, "slice step must be positive");
    auto sizes = this->sizes();
    auto strides = this->strides();
    const int64_t n = sizes[d];
    int64_t lo = start.value_or(0), hi = end.value_or(std::numeric_limits<int64_t>::max());
    if (lo < 0) lo += n;
    if (hi < 0) hi += n;
    lo = lo < 0 ? 0 : (lo > n ? n : lo);
    hi = hi < lo ? lo : (hi > n ? n : hi);
    const int64_t off = storage_offset() + lo * strides[d];
    sizes[d] = (hi - lo + step - 1) / step;
    strides[d] *= step;
    return as_strided(sizes, strides, off);
}

Tensor Tensor::select(int64_t dim, int64_t index) const {
    CHECK_FAIL(this->dim() > 0, "select() cannot be applied to a 0-dim tensor.");
    const int d = maybe_wrap_dim((int)dim, this->dim());
    const int64_t n = shape(d);
    CHECK_FAIL(index >= -n && index < n, "select(): index ", index, " out of range for dimension of size ", n);
    if (index < 0) index += n;
    auto sizes = this->sizes();
    auto strides = this->strides();
    const int64_t off = storage_offset() + index * strides[d];
    sizes.erase(sizes.begin() + d);
    strides.erase(strides.begin() + d);
    return as_strided(sizes, strides, off);
}

Tensor Tensor::narrow(int64_t dim, int64_t start, int64_t length) const {
    CHECK_FAIL(this->dim() > 0, "narrow() cannot be applied to a 0-dim tensor.");
    CHECK_FAIL(length >= 0, "narrow(): length must be non-negative.");
    const int64_t n = shape((int)dim);
    if (start < 0) start += n;
    CHECK_FAIL(start <= n - length, "start (", start, ") + length (", length, ") exceeds dimension size (", n, ").");
    return slice(dim, start, start + length, 1);
}

static std::vector<int64_t> view_sizes(const Tensor &t, std::vector<int64_t> sizes) { // one -1 is inferred (tensor.cpp:269-289 of the reference)
    int64_t known = 1;
    int infer = -1;
    for (size_t i = 0; i < sizes.size(); ++i) {
        if (sizes[i] < 0) {
            CHECK_FAIL(infer < 0);
            infer = (int)i;
        } else {
            known *= sizes[i];
        }
    }
    if (infer >= 0) {
        CHECK_FAIL(known != 0);
        sizes[infer] = t.numel() / known;
        known *= sizes[infer];
    }
    CHECK_FAIL(known == t.numel());
    return sizes;
}

Tensor Tensor::view(std::vector<int64_t> sizes) const {
    // (the reference asks its FLAG, tensor.cpp:270: it refuses every tensor as_strided_ made, the result of view() itself included; this host asks the strides - a superset:
    //  whatever view() accepts there it accepts here, with the same result)
    CHECK_FAIL(is_dense());
    return as_strided(view_sizes(*this, std::move(sizes)), {}, storage_offset());
}

bool Tensor::can_use_32bit_indexing() const {
    const int64_t lim = std::numeric_limits<int32_t>::max();
    if (numel() > lim) return false;
    int64_t reach = 1;
    for (int d = 0; d < dim(); ++d) reach += (shape(d) - 1) * stride(d) * element_size_in_bytes();
    return reach <= lim;
}

std::vector<Tensor> Tensor::split(std::vector<int64_t> indices, int64_t dim) const { return gpu::tensor_split(*this, indices, dim); }
Tensor Tensor::_half() const { return gpu::convert(*this, ScalarType::Half); }
Tensor Tensor::_bfloat16() const { return gpu::convert(*this, ScalarType::BFloat16); }
Tensor Tensor::_float() const { return gpu::convert(*this, ScalarType::Float); }

Tensor Tensor::operator+(const Tensor &o) const { return gpu::add(*this, o); }
Tensor &Tensor::operator+=(const Tensor &o) { return gpu::add_(*this, o); }
Tensor Tensor::operator-(const Tensor &o) const { return gpu::sub(*this, o); }
Tensor &Tensor::operator-=(const Tensor &o) { return gpu::sub_(*this, o); }
Tensor Tensor::operator*(const Tensor &o) const { return gpu::mul(*this, o); }
Tensor &Tensor::operator*=(const Tensor &o) { return gpu::mul_(*this, o); }
Tensor Tensor::operator/(const Tensor &o) const { return gpu::div(*this, o); }
Tensor &Tensor::operator/=(const Tensor &o) { return gpu::div_(*this, o); }
Tensor &Tensor::copy_(const Tensor &o) { return gpu::copy_(*this, o); }
Tensor Tensor::sum(int64_t d) const { return gpu::sum(*this, d); }
Tensor Tensor::mean(int64_t d) const { return gpu::mean(*this, d); }
std::tuple<Tensor, Tensor> Tensor::sort(int64_t d, bool desc) const { return gpu::sort(*this, d, desc); }
std::tuple<Tensor, Tensor> Tensor::topk(int64_t k, int64_t d, bool largest) const { return gpu::topk(*this, k, d, largest); }
std::tuple<Tensor, Tensor> Tensor::mean_var(int64_t d, bool take_sqrt) const { return gpu::mean_var(*this, d, take_sqrt); }
std::tuple<Tensor, Tensor> Tensor::norm_stat(int64_t d) const { return gpu::norm_stat(*this, d); }
Tensor &Tensor::index_put_(const std::vector<Tensor> &idx, const Tensor &v) { return gpu::index_put_(*this, idx, v); }

// ---- autograd (reference tensor.cpp:75-126) ------------------------------------------------------
void Tensor::update_grad(Tensor grad) {
    auto *impl = impl_.get();
    if (std::shared_ptr<GradSink> sink = impl->sink_.lock()) {
        // a bucketed leaf: its gradient lives in the bucket's flat buffer. A backward function that knew the slot (GemmGradFunction
        // writes dW straight into it) hands the slot back - nothing to copy; anything else is copied in (first gradient) or added.
        Tensor slot = sink->slot(impl);
        if (!impl->grad_) {
            if (grad.data_ptr() != slot.data_ptr()) slot.copy_(grad);
            impl->grad_ = std::make_unique<Tensor>(slot);
        } else {
            *impl->grad_ += grad;
        }
        sink->arrived(impl);
        return;
    }
    if (impl->grad_) {
        *impl->grad_ += grad;
    } else {
        // First gradient of a leaf. The reference copies it (tensor.cpp:75-84); a gradient that nobody else can see - a fresh,
        // dense tensor owning its whole storage, whose ONLY handle is this parameter (the engine moves its accumulator slot in;
        // round 3: the count is exactly one, not "at most the three handles the engine happens to hold") - is adopted instead:
        // for a weight gradient that is one read + one write of the parameter's size saved per step. Anything shared (the SAME
        // tensor handed to two inputs, as add's backward does; a view; the caller's own grad_output; a tensor a GradFunction
        // keeps) is still copied.
        const bool exclusive = grad.impl_ref_count() == 1 && grad.storage_ref_count() == 1 && grad.is_dense() && grad.storage_offset() == 0 &&
                               (size_t)grad.numel() * (size_t)grad.element_size_in_bytes() <= grad.storage_bytes() && !grad.has_grad_fn();
        if (exclusive) {
            impl->grad_ = std::make_unique<Tensor>(std::move(grad));
        } else {
            Tensor g = empty_like(grad);
            g.copy_(grad);
            impl->grad_ = std::make_unique<Tensor>(g);
        }
    }
}

void Tensor::backward(Tensor grad_output) {
    // pass 1: how many consumers will feed each differentiable tensor
    std::unordered_map<TensorImpl *, int> pending;
    std::unordered_set<TensorImpl *> expanded;
    std::queue<Tensor *> work;
    work.push(this);
    while (!work.empty()) {
        Tensor *t = work.front();
        work.pop();
        if (!t->has_grad_fn()) continue;
        if (!expanded.insert(t->impl()).second) continue; // reached again through another consumer: its inputs are counted once
        for (auto &in : t->grad_fn_->inputs) {
            if (!in.requires_grad()) continue;
            pending[in.impl()] += 1;
            work.push(&in);
        }
    }
    // pass 2: propagate in dependency order, summing fan-in
    std::unordered_map<TensorImpl *, Tensor> acc;
    acc[impl()] = grad_output;
    work.push(this);
    while (!work.empty()) {
        Tensor *t = work.front();
        work.pop();
        // the slot is MOVED out: every consumer of t has reported (pending == 0), nobody adds to it again, and a leaf can see that the
        // handle it receives is the only one
        auto slot_it = acc.find(t->impl());
        Tensor g = std::move(slot_it->second);
        acc.erase(slot_it);
        if (t->has_grad_fn()) {
            GradFunction *fn = t->grad_fn_.get();
            std::vector<Tensor> gin = fn->backward(g);
            for (size_t i = 0; i < fn->inputs.size(); ++i) {
                Tensor &in = fn->inputs[i];
                if (!in.requires_grad()) continue;
                Tensor &slot = acc[in.impl()];
                slot = slot.defined() ? (slot + gin[i]) : gin[i];
                if (--pending[in.impl()] == 0) work.push(&in);
            }
        } else if (t->requires_grad()) {
            t->update_grad(std::move(g));
        }
    }
}

// ---- printing (reference tensor.cpp:323-377) --------------------------------------------------------
static double item_as_double(const Tensor &t, const std::vector<int64_t> &idx) {
    any_t raw = t.item(idx);
    switch (t.dtype()) {
    case ScalarType::Bool: return *reinterpret_cast<uint8_t *>(raw.val) != 0;
    case ScalarType::Byte: return *reinterpret_cast<uint8_t *>(raw.val);
    case ScalarType::Char: return *reinterpret_cast<int8_t *>(raw.val);
    case ScalarType::Short: return *reinterpret_cast<int16_t *>(raw.val);
    case ScalarType::Int: return *reinterpret_cast<int32_t *>(raw.val);
    case ScalarType::Long: return (double)*reinterpret_cast<int64_t *>(raw.val);
    case ScalarType::Half: return dtype::f16_bits_to_float(*reinterpret_cast<uint16_t *>(raw.val));
    case ScalarType::BFloat16: return dtype::bf16_bits_to_float(*reinterpret_cast<uint16_t *>(raw.val));
    case ScalarType::Float: return *reinterpret_cast<float *>(raw.val);
    case ScalarType::Double: return *reinterpret_cast<double *>(raw.val);
    default: return 0;
    }
}

static int64_t item_as_int64(const Tensor &t, const std::vector<int64_t> &idx) { // the integer types, exactly (a double holds 53 bits)
    any_t raw = t.item(idx);
    switch (t.dtype()) {
    case ScalarType::Bool: return *reinterpret_cast<uint8_t *>(raw.val) != 0;
    case ScalarType::Byte: return *reinterpret_cast<uint8_t *>(raw.val);
    case ScalarType::Char: return *reinterpret_cast<int8_t *>(raw.val);
    case ScalarType::Short: return *reinterpret_cast<int16_t *>(raw.val);
    case ScalarType::Int: return *reinterpret_cast<int32_t *>(raw.val);
    case ScalarType::Long: return *reinterpret_cast<int64_t *>(raw.val);
    default: return 0;
    }
}

static void print_rec(std::ostream &os, const Tensor &t, std::vector<int64_t> &idx, int d) {
    if (d == t.dim()) { // an element, as the reference streams its accumulate type: integers as integers, floats with five decimals, always signed
        if (is_floating_type(t.dtype())) os << std::fixed << std::showpos << std::setprecision(5) << item_as_double(t, idx) << std::noshowpos;
        else os << std::showpos << item_as_int64(t, idx) << std::noshowpos;
        return;
    }
    if (d > 0) os << "\n";
    os << std::string(2 * (d + 1), ' ') << "[";
    const int64_t shown = std::min<int64_t>(t.shape(d), 12);
    for (int64_t i = 0; i < shown; ++i) {
        if (i) os << ", ";
        idx.push_back(i);
        print_rec(os, t, idx, d + 1);
        idx.pop_back();
    }
    if (t.shape(d) > 12) { // the thirteenth and later entries of a dim: an ellipsis, on a line of its own for an outer dim
        os << ", ";
        if (d < t.dim() - 1) os << "\n" << std::string(2 * (d + 2), ' ');
        os << "...";
    }
    if (d < t.dim() - 1) os << "\n" << std::string(2 * (d + 1), ' ');
    os << "]";
}

// The reference writes "shape=[2,3,\b]" - a trailing comma and a backspace, which a terminal shows as "[2,3]" and a log file does not; this host writes what the
// terminal shows. Everything else of the text is the reference's, byte for byte (tests/test_gpu_host_diff_fuzz.py compares the two with that one substitution).
std::ostream &operator<<(std::ostream &os, const Tensor &t) {
    if (!t.defined()) return os << "Tensor(Undefined)";
    os << "tensor(shape=[";
    for (int i = 0; i < t.dim(); ++i) os << (i ? "," : "") << t.shape(i);
    os << "], stride=[";
    for (int i = 0; i < t.dim(); ++i) os << (i ? "," : "") << t.stride(i);
    os << "], storage_offset=" << t.storage_offset() << ", dtype=" << t.dtype() << ", numel=" << t.numel() << ", dim=" << t.dim()
       << ", device=" << t.device() << ") {\n";
    std::vector<int64_t> idx;
    print_rec(os, t, idx, 0);
    return os << "\n}";
}

std::string Tensor::to_string() const {
    std::ostringstream oss;
    oss << *this;
    return oss.str();
}
