#include "tensor_iterator.h"

#include <algorithm>
#include <cstdlib>
#include <limits>
#include <utility>

// =================================================================================================
// IterGeometry
// =================================================================================================
int IterGeometry::add(const IterOperand &op, bool is_output) {
    CHECK_FAIL((int)ops_.size() < MAX_TENSORS, "too many operands");
    if (is_output) CHECK_FAIL(noutputs_ == (int)ops_.size(), "outputs must be added before inputs");
    ops_.push_back(op);
    ops_.back().is_output = is_output;
    if (is_output) ++noutputs_;
    return (int)ops_.size() - 1;
}

namespace {

// every element offset distinct and the view fills a dense block (reference memory_overlap.h:10-28)
bool non_overlapping_and_dense(const IterOperand &t) {
    std::vector<std::pair<int64_t, int64_t>> v; // (stride, size)
    for (int i = t.ndim - 1; i >= 0; --i) {
        if (t.shape[i] == 0) return true; // no elements: nothing can overlap
        // (an extent-1 dim takes part with whatever stride it carries, as in the reference: x[::2][0:1] += 1 is refused there although the stride of that dim addresses
        //  nothing - found by tests/test_gpu_host_diff_fuzz.py; this host used to skip such dims)
        v.emplace_back(t.stride[i], t.shape[i]);
    }
    std::stable_sort(v.begin(), v.end(), [](const auto &a, const auto &b) { return a.first < b.first; });
    int64_t expect = 1;
    for (auto &p : v) {
        if (p.first != expect) return false;
        expect *= p.second;
    }
    return true;
}

// closed byte range an operand touches
std::pair<uintptr_t, uintptr_t> byte_range(const IterOperand &t) {
    int64_t lo = 0, hi = 0;
    for (int i = 0; i < t.ndim; ++i) {
        if (t.shape[i] == 0) return {1, 0}; // empty operand: an empty range, overlaps nothing
        const int64_t span = (t.shape[i] - 1) * t.stride[i];
        (span >= 0 ? hi : lo) += span;
    }
    const int64_t es = (int64_t)element_size(t.dtype);
    const auto base = reinterpret_cast<uintptr_t>(t.data);
    return {base + lo * es, base + hi * es};
}

} // namespace

void IterGeometry::build(bool is_reduction, int64_t reduce_dim, bool resize_outputs, bool check_mem_overlap, const Allocator &alloc) {
    is_reduction_ = is_reduction;
    const int nt = ntensors();

    // 1. one device for every defined operand
    common_device_ = -1;
    for (auto &t : ops_) {
        if (!t.defined) continue;
        if (common_device_ == -1 && t.device >= 0) common_device_ = t.device;
        else CHECK_FAIL(t.device == common_device_, "All defined tensors should in the same device");
    }
    // 2. one rank: operands broadcast along size-1 dims only, never along missing leading dims
    ndim_ = -1;
    for (auto &t : ops_) {
        if (!t.defined) continue;
        if (ndim_ < 0) ndim_ = t.ndim;
        else CHECK_FAIL(ndim_ == t.ndim, "All defined tensors should in the same dim");
    }
    CHECK_FAIL(ndim_ >= 0, "TensorIterator needs at least one defined operand");
    // 3. common dtype = fold of the promotion rule over the INPUTS
    common_dtype_ = ScalarType::Undefined;
    for (int i = noutputs_; i < nt; ++i)
        common_dtype_ = common_dtype_ == ScalarType::Undefined ? ops_[i].dtype : promote_types(common_dtype_, ops_[i].dtype);
    // 4. reductions allocate their keepdim output up front
    if (is_reduction_) {
        CHECK_FAIL(nt > noutputs_, "a reduction needs an input");
        const IterOperand &in = ops_[noutputs_];
        const int rd = maybe_wrap_dim((int)reduce_dim, ndim_);
        CHECK_FAIL(rd >= 0 && rd < ndim_, "reduce dim ", reduce_dim, " out of range");
        for (int i = 0; i < noutputs_; ++i) {
            if (ops_[i].defined) continue;
            int64_t shape[MAX_TENSOR_DIMS];
            for (int k = 0; k < ndim_; ++k) shape[k] = in.shape[k];
            shape[rd] = 1;
            alloc(i, shape, ndim_, common_dtype_, common_device_, ops_[i]);
            ops_[i].is_output = true;
        }
    }
    // 5. in-place outputs
    for (int i = 0; i < noutputs_; ++i) {
        if (!ops_[i].defined) continue;
        for (int j = noutputs_; j < nt; ++j)
            if (ops_[i].identity && ops_[i].identity == ops_[j].identity) ops_[i].is_read_write = true;
    }
    // 6. outputs must be dense and may alias an input only completely
    if (check_mem_overlap) {
        for (int i = 0; i < noutputs_; ++i) {
            if (!ops_[i].defined) continue;
            CHECK_FAIL(non_overlapping_and_dense(ops_[i]), "output has internal overlap or is not dense");
            const auto ro = byte_range(ops_[i]);
            for (int j = noutputs_; j < nt; ++j) {
                if (ops_[i].identity && ops_[i].identity == ops_[j].identity) continue;
                const auto ri = byte_range(ops_[j]);
                CHECK_FAIL(ro.second < ri.first || ri.second < ro.first, "output partially overlaps an input");
            }
        }
    }
    // 7. broadcast shape
    for (int d = ndim_ - 1; d >= 0; --d) {
        int64_t sz = -1;
        for (auto &t : ops_) {
            if (!t.defined) continue;
            if (sz < 0) sz = t.shape[d];
            else {
                CHECK_FAIL(sz == t.shape[d] || sz == 1 || t.shape[d] == 1, "shapes cannot be broadcast at dim ", d);
                if (sz == 1) sz = t.shape[d];
            }
        }
        shape_[d] = sz;
    }
    // 8. outputs cannot be broadcast: wrong-shaped ones are resized (or rejected)
    for (int i = 0; i < noutputs_; ++i) {
        IterOperand &o = ops_[i];
        if (!o.defined) {
            o.will_resize = true;
            continue;
        }
        bool same = true;
        for (int d = 0; d < ndim_; ++d) same = same && o.shape[d] == shape_[d];
        if (same) continue;
        if (resize_outputs && !o.is_read_write) {
            o.will_resize = true;
            continue;
        }
        CHECK_FAIL(is_reduction_, "output shape doesn't match the broadcast shape");
    }
    // 9. byte strides, 0 along broadcast (input) / reduced (output) dims
    for (int t = 0; t < nt; ++t) {
        const IterOperand &o = ops_[t];
        if (!o.defined) continue;
        const int64_t es = (int64_t)element_size(o.dtype);
        for (int d = 0; d < ndim_; ++d) stride_bytes_[t][d] = (o.shape[d] == 1 && shape_[d] != 1) ? 0 : o.stride[d] * es;
    }
    // 10. order dims fastest-first (insertion sort tolerant of "don't care" comparisons)
    for (int i = 0; i < ndim_; ++i) perm_[i] = ndim_ - 1 - i;
    if (ndim_ > 1) {
        auto order = [&](int64_t d0, int64_t d1) -> int { // >0: d0 must come after d1
            for (int t = 0; t < nt; ++t) {
                if (!ops_[t].defined || ops_[t].will_resize) continue;
                const int64_t s0 = stride_bytes_[t][d0], s1 = stride_bytes_[t][d1];
                if (is_reduction_ && t < noutputs_ && (s0 == 0) != (s1 == 0)) return s1 == 0 ? 1 : -1; // reduced dims first
                if (s0 == 0 || s1 == 0) continue;
                if (s0 != s1) return s0 < s1 ? -1 : 1;
                if (shape_[d0] > shape_[d1]) return 1;
            }
            return 0;
        };
        for (int i = 1; i < ndim_; ++i) {
            int hi = i;
            for (int lo = i - 1; lo >= 0; --lo) {
                const int c = order(perm_[lo], perm_[hi]);
                if (c > 0) {
                    std::swap(perm_[lo], perm_[hi]);
                    hi = lo;
                } else if (c < 0) {
                    break;
                }
            }
        }
        int64_t shp[MAX_TENSOR_DIMS], str[MAX_TENSORS][MAX_TENSOR_DIMS];
        for (int d = 0; d < ndim_; ++d) shp[d] = shape_[d];
        for (int t = 0; t < nt; ++t)
            for (int d = 0; d < ndim_; ++d) str[t][d] = stride_bytes_[t][d];
        for (int d = 0; d < ndim_; ++d) {
            shape_[d] = shp[perm_[d]];
            for (int t = 0; t < nt; ++t) stride_bytes_[t][d] = str[t][perm_[d]];
        }
    }
    // 11. allocate missing / resized outputs, contiguous in the ORIGINAL dim order
    for (int i = 0; i < noutputs_; ++i) {
        IterOperand &o = ops_[i];
        if (o.defined && !o.will_resize) continue;
        int64_t shape[MAX_TENSOR_DIMS];
        for (int d = 0; d < ndim_; ++d) shape[perm_[d]] = shape_[d];
        const ScalarType dt = common_dtype_ != ScalarType::Undefined ? common_dtype_ : o.dtype;
        alloc(i, shape, ndim_, dt, common_device_, o);
        o.is_output = true;
        const int64_t es = (int64_t)element_size(o.dtype);
        for (int d = 0; d < ndim_; ++d) stride_bytes_[i][d] = o.stride[perm_[d]] * es;
    }
    // 12. merge adjacent dims that walk memory as one
    coalesce();
    // 13. base pointers
    for (int t = 0; t < MAX_TENSORS; ++t) data_[t] = t < nt ? ops_[t].data : nullptr;
}

void IterGeometry::coalesce() {
    if (ndim_ <= 1) return;
    const int nt = ntensors();
    auto mergeable = [&](int a, int b) {
        if (shape_[a] == 1 || shape_[b] == 1) return true;
        for (int t = 0; t < nt; ++t)
            if (shape_[a] * stride_bytes_[t][a] != stride_bytes_[t][b]) return false;
        return true;
    };
    auto take_strides = [&](int dst, int src) {
        for (int t = 0; t < nt; ++t) stride_bytes_[t][dst] = stride_bytes_[t][src];
    };
    int last = 0;
    for (int d = 1; d < ndim_; ++d) {
        if (mergeable(last, d)) {
            if (shape_[last] == 1) take_strides(last, d);
            shape_[last] *= shape_[d];
        } else {
            ++last;
            if (last != d) {
                take_strides(last, d);
                shape_[last] = shape_[d];
            }
        }
    }
    ndim_ = last + 1;
}

int64_t IterGeometry::numel() const {
    int64_t n = 1;
    for (int d = 0; d < ndim_; ++d) n *= shape_[d];
    return n;
}

int64_t IterGeometry::num_output_elements() const {
    int64_t n = 1;
    for (int d = 0; d < ndim_; ++d)
        if (stride_bytes_[0][d] != 0 || shape_[d] == 0) n *= shape_[d];
    return n;
}

bool IterGeometry::can_use_32bit_indexing() const {
    const int64_t lim = std::numeric_limits<int32_t>::max();
    if (numel() > lim) return false;
    for (int t = 0; t < ntensors(); ++t) {
        int64_t reach = 1;
        for (int d = 0; d < ndim_; ++d) reach += (shape_[d] - 1) * stride_bytes_[t][d];
        if (reach > lim) return false;
    }
    return true;
}

bool IterGeometry::is_contiguous() const {
    if (numel() == 1) return true;
    if (ndim_ != 1) return false;
    for (int t = 0; t < ntensors(); ++t)
        if (stride_bytes_[t][0] != (int64_t)element_size(ops_[t].dtype)) return false;
    return true;
}

bool IterGeometry::is_dim_reduced(int d) const {
    for (int t = 0; t < noutputs_; ++t)
        if (stride_bytes_[t][d] == 0 && shape_[d] > 1) return true;
    return false;
}

int IterGeometry::dim_to_split() const {
    CHECK_FAIL(ndim_ >= 1);
    int64_t best = -1;
    int which = -1;
    for (int d = ndim_ - 1; d >= 0; --d) {
        if (shape_[d] == 0) continue;
        for (int t = 0; t < ntensors(); ++t) {
            const int64_t extent = (shape_[d] - 1) * std::llabs(stride_bytes_[t][d]);
            if (extent > best) {
                best = extent;
                which = d;
            }
        }
    }
    CHECK_FAIL(best >= 0);
    return which;
}

void IterGeometry::narrow(int d, int64_t start, int64_t size) {
    CHECK_FAIL(d < ndim_ && size >= 1);
    shape_[d] = size;
    for (int t = 0; t < ntensors(); ++t) data_[t] += stride_bytes_[t][d] * start;
    if (size == 1 && !is_reduction_) coalesce();
}

void IterGeometry::for_each_32bit(const std::function<void(const IterGeometry &)> &fn) const {
    if (can_use_32bit_indexing()) {
        fn(*this);
        return;
    }
    const int d = dim_to_split();
    CHECK_FAIL(shape_[d] >= 2);
    CHECK_FAIL(!is_dim_reduced(d), "splitting a reduced dimension is not supported");
    const int64_t left = shape_[d] / 2;
    IterGeometry a = *this, b = *this;
    a.narrow(d, 0, left);
    b.narrow(d, left, shape_[d] - left);
    a.for_each_32bit(fn);
    b.for_each_32bit(fn);
}

void IterGeometry::to_desc(kf_iter_desc &d) const {
    d = kf_iter_desc{};
    d.ndim = ndim_ > 0 ? ndim_ : 1;
    d.ntensors = ntensors();
    d.noutputs = noutputs_;
    for (int i = 0; i < KF_MAX_DIMS; ++i) d.shape[i] = i < ndim_ ? shape_[i] : (i == 0 ? 1 : 0);
    for (int t = 0; t < ntensors(); ++t) {
        d.dtype[t] = static_cast<int>(ops_[t].dtype);
        d.data[t] = data_[t];
        for (int i = 0; i < ndim_; ++i) d.stride_bytes[t][i] = stride_bytes_[t][i];
        if (ndim_ == 0) d.stride_bytes[t][0] = (int64_t)element_size(ops_[t].dtype);
    }
}

// =================================================================================================
// TensorIterator (Tensor-facing builder)
// =================================================================================================
TensorIterator &TensorIterator::add_output(Tensor &output) {
    CHECK_FAIL((int)tensors_.size() == num_outputs_, "outputs must be added before inputs");
    tensors_.push_back(&output);
    ++num_outputs_;
    return *this;
}

TensorIterator &TensorIterator::add_input(const Tensor &input) {
    tensors_.push_back(const_cast<Tensor *>(&input));
    return *this;
}

static void describe(const Tensor &t, IterOperand &op) {
    op.defined = t.defined();
    op.identity = &t;
    if (!op.defined) return;
    op.ndim = t.dim();
    for (int d = 0; d < op.ndim; ++d) {
        op.shape[d] = t.shape(d);
        op.stride[d] = t.stride(d);
    }
    op.dtype = t.dtype();
    op.device = t.device();
    op.data = static_cast<char *>(t.data_ptr());
}

TensorIterator &TensorIterator::build() {
    geo_ = IterGeometry();
    for (int i = 0; i < (int)tensors_.size(); ++i) {
        IterOperand op;
        describe(*tensors_[i], op);
        geo_.add(op, i < num_outputs_);
    }
    auto alloc = [this](int arg, const int64_t *shape, int ndim, ScalarType dtype, int device, IterOperand &op) {
        *tensors_[arg] = empty(shape, ndim, dtype, device, false);
        const void *ident = op.identity;
        describe(*tensors_[arg], op);
        op.identity = ident;
    };
    geo_.build(is_reduction_, reduce_dim_, resize_outputs_, check_mem_overlap_, alloc);
    return *this;
}

TensorIterator &TensorIterator::build_for_loops() {
    is_reduction_ = false;
    resize_outputs_ = true;
    return build();
}

TensorIterator &TensorIterator::build_for_reduce(int64_t reduce_dim) {
    is_reduction_ = true;
    resize_outputs_ = false;
    reduce_dim_ = reduce_dim;
    return build();
}

std::ostream &operator<<(std::ostream &os, const TensorIterator &it) {
    os << "TensorIterator(shape=[";
    for (int d = 0; d < it.ndim(); ++d) os << (d ? "," : "") << it.shape(d);
    os << "]";
    for (int t = 0; t < it.ntensors(); ++t) {
        os << ", stride_bytes_" << t << "=[";
        for (int d = 0; d < it.ndim(); ++d) os << (d ? "," : "") << it.stride_bytes(t, d);
        os << "]";
    }
    os << ", perm=[";
    for (int d = 0; d < it.ndim(); ++d) os << (d ? "," : "") << it.perm(d);
    return os << "], ninputs=" << it.ninputs() << ", noutputs=" << it.noutputs() << ")";
}
