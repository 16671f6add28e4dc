// docs/seam.cpp - the reference's device seam (src/device/include/*.h), implemented over include/kfunca_hip.h.
//
// This is the ONE translation unit a kfunca maintainer adds (as src/device_hip/seam.cpp) to put the MI355X device library
// behind the unchanged host core: build src/core as before, compile this file with the host compiler, link -lkfunca_hip
// instead of building src/device/*.cu. It includes ONLY reference headers + the C ABI, and every function below is the
// definition of a declaration in src/device/include/ (cited per function).
//
// It is compiled - syntax and types, against the real reference headers - by tests/test_seam_compiles.py whenever the reference
// tree is mounted (build container): g++ -std=c++20 -fsyntax-only -I src/core/include -I src/core/utils{,/memory}
// -I src/device/include. Nothing from the reference is copied into this repository; the file is documentation that type-checks.
#include <cmath>
#include <iostream>
#include <tuple>
#include <vector>

#include "kfunca_hip.h" // the C ABI (this repository's include/)

#include "tensor_iterator.h" // reference: src/core/include (Tensor, TensorIterator, DeviceAllocator, CHECK_FAIL, any_t)

#include "binary_ops_kernel.h" // reference: src/device/include - the declarations defined below
#include "causal_attention_kernel.h"
#include "device_info.h"
#include "gemm_kernel.h"
#include "index_ops_kernel.h"
#include "memory_engine.h"
#include "norm_ops_kernel.h"
#include "nullary_ops_kernel.h"
#include "reduce_ops_kernel.h"
#include "sort_ops_kernel.h"
#include "unary_ops_kernel.h"

namespace {

void check(int rc) { CHECK_FAIL(rc == KF_OK, kf_last_error()); } // status -> the reference's utils::Error (exception.h:123-131)

// the reference launches on the legacy default stream (launcher_cuda.h:315-353); NULL = the device's null stream
void *const kStream = nullptr;

// post-build() TensorIterator state (tensor_iterator.h:27-47) -> the POD the C ABI takes
kf_iter_desc to_desc(const TensorIterator &it) {
    kf_iter_desc d{};
    d.ndim = it.ndim();
    d.ntensors = it.ntensors();
    d.noutputs = it.noutputs();
    for (int i = 0; i < it.ndim(); ++i) d.shape[i] = it.shape(i);
    for (int t = 0; t < it.ntensors(); ++t) {
        d.dtype[t] = static_cast<int>(it.dtype(t)); // ScalarType order == KF_* codes (scalar_type.h:9-27)
        d.data[t] = it.data_ptr(t);
        for (int i = 0; i < it.ndim(); ++i) d.stride_bytes[t][i] = it.stride_bytes(t, i);
    }
    return d;
}

// gpu_kernel's outer loop (tensor_loops.h:357-369): 32-bit-indexable pieces, one launch each
template <typename F>
void for_each_piece(TensorIterator &it, F &&launch) {
    if (it.numel() == 0) return;
    if (it.can_use_32bit_indexing()) {
        launch(it);
        return;
    }
    for (auto &sub : it.with_32bit_indexing()) launch(sub);
}

void elementwise(TensorIterator &it, int op, int compute_dtype, double scalar) {
    for_each_piece(it, [&](TensorIterator &piece) {
        kf_iter_desc d = to_desc(piece);
        check(kf_elementwise(op, &d, compute_dtype, scalar, kStream));
    });
}

// caller-owned scratch from the host allocator (the reference's device layer calls back into it, tensor_reduce.h:1057-1058)
struct Scratch {
    DataPtr block;
    void *ptr = nullptr;
    Scratch(size_t bytes, int device) {
        if (bytes) {
            block = DeviceAllocator::GetInstance()->allocate(bytes, device);
            ptr = block.get();
        }
    }
};

void reduce(TensorIterator &it, int op) {
    if (it.num_output_elements() == 0) return;
    kf_iter_desc d = to_desc(it);
    size_t need = 0;
    check(kf_reduce_workspace_bytes(&d, &need));
    Scratch ws(need, it.device(0));
    check(kf_reduce(op, &d, ws.ptr, need, kStream));
}

} // namespace

// ---- memory_engine.h:5-10 ------------------------------------------------------------------------------------------------
void dset_device(const int device) { check(kf_set_device(device)); }
void *dmalloc(const size_t size) {
    void *p = nullptr;
    check(kf_malloc(&p, size));
    return p;
}
void dfree(void *ptr) { check(kf_free(ptr)); }
void dmemcpy_h2d(void *dst, const void *src, const size_t size) { check(kf_memcpy_h2d(dst, src, size, kStream)); }
void dmemcpy_d2h(void *dst, const void *src, const size_t size) { check(kf_memcpy_d2h(dst, src, size, kStream)); }
void dmemset_zeros(void *ptr, const size_t size) {
    check(kf_memset_zero(ptr, size, kStream));
    check(kf_stream_sync(kStream)); // the reference's memset returns synchronised (launcher_cuda.h:196-202)
}

// ---- binary_ops_kernel.h:5-8: arithmetic runs in the common dtype's accumulate type (binary_ops_kernel.cu:34-60) ----------
void add_kernel(TensorIterator &iter) { elementwise(iter, KF_EW_ADD, static_cast<int>(iter.common_dtype()), 0.0); }
void sub_kernel(TensorIterator &iter) { elementwise(iter, KF_EW_SUB, static_cast<int>(iter.common_dtype()), 0.0); }
void mul_kernel(TensorIterator &iter) { elementwise(iter, KF_EW_MUL, static_cast<int>(iter.common_dtype()), 0.0); }
void div_kernel(TensorIterator &iter) { elementwise(iter, KF_EW_DIV, static_cast<int>(iter.common_dtype()), 0.0); }

// ---- unary_ops_kernel.h:5, nullary_ops_kernel.h:5 -------------------------------------------------------------------------
void copy_kernel(TensorIterator &iter) { elementwise(iter, KF_EW_COPY, 0, 0.0); }
void fill_kernel(TensorIterator &iter, const any_t &value) { elementwise(iter, KF_EW_FILL, 0, static_cast<double>(value)); }

// ---- reduce_ops_kernel.h:5-7 ----------------------------------------------------------------------------------------------
void sum_kernel(TensorIterator &iter) { reduce(iter, KF_RED_SUM); }
void mean_kernel(TensorIterator &iter) { reduce(iter, KF_RED_MEAN); }
void mean_var_kernel(TensorIterator &iter, double correction, bool take_sqrt) {
    if (iter.num_output_elements() == 0) return;
    kf_iter_desc d = to_desc(iter); // outputs (var | std, mean), then the input: reduce_ops.cpp:24-25
    size_t need = 0;
    check(kf_reduce_moments_workspace_bytes(&d, &need));
    Scratch ws(need, iter.device(0));
    check(kf_reduce_moments(take_sqrt ? KF_MOM_STD : KF_MOM_VAR, &d, correction, 0.0, ws.ptr, need, kStream));
}

// ---- norm_ops_kernel.h:5: mean and invstd over dim 0 of a 2-D tensor, eps = 1e-12 (norm_ops_kernel.cu:6-61) ---------------
std::tuple<Tensor, Tensor> norm_stat_kernel(const Tensor &self, const int dim) {
    CHECK_FAIL(self.defined());
    CHECK_FAIL(dim == 0 && self.dim() == 2);
    Tensor invstd, mean;
    auto iter = TensorIterator().add_output(invstd).add_output(mean).add_input(self).build_for_reduce(dim);
    if (iter.num_output_elements() != 0) {
        kf_iter_desc d = to_desc(iter);
        size_t need = 0;
        check(kf_reduce_moments_workspace_bytes(&d, &need));
        Scratch ws(need, self.device());
        check(kf_reduce_moments(KF_MOM_INVSTD, &d, 0.0, 1e-12, ws.ptr, need, kStream));
    }
    return std::make_tuple(mean, invstd);
}

// ---- index_ops_kernel.h:5 -------------------------------------------------------------------------------------------------
void index_put_kernel(TensorIterator &iter, const std::vector<int64_t> index_size, const std::vector<int64_t> index_stride) {
    for_each_piece(iter, [&](TensorIterator &piece) {
        kf_iter_desc d = to_desc(piece);
        check(kf_index_put(&d, static_cast<int>(index_size.size()), index_size.data(), index_stride.data(), kStream));
    });
}

// ---- gemm_kernel.h:5 (argument checks as gemm_kernel.cu:8-25) -------------------------------------------------------------
void gemm_kernel(Tensor &out, const Tensor &a, const Tensor &b, float alpha, float beta) {
    CHECK_FAIL(out.is_contiguous() && a.is_contiguous() && b.is_contiguous());
    CHECK_FAIL(b.dim() == 2 && a.dim() >= 1 && a.dtype() == b.dtype() && out.dtype() == a.dtype());
    const int64_t k = a.shape(a.dim() - 1), m = a.numel() / k, n = b.shape(1);
    CHECK_FAIL(b.shape(0) == k && out.numel() == m * n);
    size_t need = 0;
    check(kf_gemm_workspace_bytes(static_cast<int>(a.dtype()), 0, 0, m, n, k, &need));
    Scratch ws(need, a.device());
    check(kf_gemm(static_cast<int>(a.dtype()), 0, 0, m, n, k, alpha, a.data_ptr(), k, b.data_ptr(), n, beta, out.data_ptr(), n, KF_EPI_NONE,
                  nullptr, ws.ptr, need, kStream));
}

// ---- causal_attention_kernel.h:5 (checks as causal_attention_kernel.cu:9-20; no S x S scratch, no discarded m / l) ----------
Tensor causal_attention_kernel(const Tensor &q, const Tensor &k, const Tensor &v) {
    CHECK_FAIL(q.dim() == 4 && k.dim() == 4 && v.dim() == 4);
    CHECK_FAIL(q.is_contiguous() && k.is_contiguous() && v.is_contiguous());
    CHECK_FAIL(q.dtype() == k.dtype() && q.dtype() == v.dtype());
    Tensor out = empty_like(q);
    check(kf_attn_fwd(static_cast<int>(q.dtype()), q.shape(0), q.shape(1), q.shape(2), k.shape(2), q.shape(3), q.data_ptr(), k.data_ptr(),
                      v.data_ptr(), out.data_ptr(), /*lse*/ nullptr, kStream));
    return out;
}

// ---- sort_ops_kernel.h:5-14: the device half is one stable segmented sort of dim-last rows; the dense dim-last copy and the
//      copy back stay host logic, written here with the reference's own view ops -----------------------------------------------
std::tuple<Tensor, Tensor> sort_stable_kernel(const Tensor &self, int64_t dim, bool descending) {
    const int d = maybe_wrap_dim(static_cast<int>(dim), self.dim());
    CHECK_FAIL(self.dtype() != ScalarType::Bool, "Sort currently does not support bool dtypes.");
    const int64_t n = self.shape(d);
    CHECK_FAIL(n <= std::numeric_limits<int>::max(), "The dimension being sorted can not have more than INT_MAX elements.");
    // A dense image of self with the sorted dim fastest and the other dims in their memory order, made by copy_ into a strided allocation - what the
    // reference's device half does (sort_ops_kernel.cu:569-577, 605-612). NOT permute().contiguous(): the reference's permute() drops a view's storage
    // offset (tensor.cpp:189), and through it this seam sorted x[0:3] when asked for x[2:5] (found by tests/test_gpu_host_diff_fuzz.py, round 6).
    const bool direct = self.is_contiguous() && self.stride(d) == 1;
    std::vector<int64_t> st(self.dim());
    Tensor keys = self;
    if (!direct) {
        std::vector<int> order;
        for (int i = 0; i < self.dim(); ++i)
            if (i != d) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return self.stride(x) > self.stride(y); });
        order.push_back(d);
        int64_t run = 1;
        for (int i = self.dim() - 1; i >= 0; --i) {
            st[order[i]] = run;
            run *= self.shape(order[i]);
        }
        keys = empty_strided(self.sizes(), st, self.dtype(), self.device());
        keys.copy_(self);
    }
    Tensor values = direct ? empty_like(keys) : empty_strided(self.sizes(), st, self.dtype(), self.device());
    Tensor positions = direct ? empty(self.sizes(), ScalarType::Long, self.device()) : empty_strided(self.sizes(), st, ScalarType::Long, self.device());
    if (self.numel() > 0) {
        const int64_t nseg = self.numel() / n;
        const size_t need = kf_sort_workspace_bytes(static_cast<int>(self.dtype()), nseg, n);
        Scratch ws(need, self.device());
        check(kf_sort(static_cast<int>(self.dtype()), keys.data_ptr(), values.data_ptr(), static_cast<int64_t *>(positions.data_ptr()), nseg, n,
                      descending ? 1 : 0, ws.ptr, need, kStream));
    }
    if (direct) return std::make_tuple(values, positions);
    Tensor values_out = empty_like(self), positions_out = empty(self.sizes(), ScalarType::Long, self.device());   // back to self's own order
    values_out.copy_(values);
    positions_out.copy_(positions);
    return std::make_tuple(values_out, positions_out);
}

std::tuple<Tensor, Tensor> topk_with_sort(const Tensor &self, int64_t k, int64_t dim, bool largest) {
    const int d = maybe_wrap_dim(static_cast<int>(dim), self.dim());
    auto sorted = sort_stable_kernel(self, d, largest);
    return std::make_tuple(std::get<0>(sorted).narrow(d, 0, k).contiguous(), std::get<1>(sorted).narrow(d, 0, k).contiguous());
}

// device_info.h:5 - void device_info(): the device's properties, a 1 GB copy-bandwidth test and a matrix throughput test
// (device_info.cu:191-216 prints cudaDeviceProp, times a vectorised float copy and an FP32 MAD loop). Over the C ABI: the properties
// from kf_device_props_get, the copy as the seam's own copy kernel on a 1 GB float buffer, the arithmetic test as the f32 GEMM the
// path replaces CUTLASS with - both timed with kf_event_*.
void device_info() {
    int ndev = 0;
    check(kf_device_count(&ndev));
    for (int i = 0; i < ndev; ++i) {
        kf_device_props p;
        check(kf_device_props_get(i, &p));
        std::cout << "[" << i << "] " << p.name << " (" << p.arch << "), " << p.compute_units << " CUs, wavefront " << p.wavefront_size << ", "
                  << p.clock_khz / 1000 << " MHz, LDS per block " << p.lds_per_block << " B, L2 " << p.l2_bytes << " B, memory "
                  << p.total_mem / (1ull << 30) << " GiB (" << p.memory_bus_bits << "-bit bus at " << p.memory_clock_khz / 1000 << " MHz)\n";
    }
    if (ndev == 0) return;
    void *e0 = nullptr, *e1 = nullptr;
    check(kf_event_create(&e0));
    check(kf_event_create(&e1));
    float ms = 0.f;
    {
        std::cout << "\n1GB copy test ... ";
        const int64_t n = 1024 * 1024 * 256;
        Tensor a = zeros({n}, ScalarType::Float, 0), b = empty({n}, ScalarType::Float, 0);
        for (int rep = 0; rep < 4; ++rep) {
            if (rep == 1) check(kf_event_record(e0, kStream));
            b.copy_(a);
        }
        check(kf_event_record(e1, kStream));
        check(kf_event_sync(e1));
        check(kf_event_elapsed_ms(e0, e1, &ms));
        std::cout << 3.0 * 2.0 * n * sizeof(float) / 1e6 / ms << " GBPS\n";
    }
    {
        std::cout << "FP32 GEMM 4096^3 test ... ";
        const int64_t n = 4096;
        Tensor a = zeros({n, n}, ScalarType::Float, 0), b = zeros({n, n}, ScalarType::Float, 0), c = empty({n, n}, ScalarType::Float, 0);
        for (int rep = 0; rep < 4; ++rep) {
            if (rep == 1) check(kf_event_record(e0, kStream));
            gemm_kernel(c, a, b, 1.0f, 0.0f);
        }
        check(kf_event_record(e1, kStream));
        check(kf_event_sync(e1));
        check(kf_event_elapsed_ms(e0, e1, &ms));
        std::cout << 3.0 * 2.0 * n * n * n / (ms / 1000) * 1e-12 << " TFLOPS" << std::endl;
    }
    check(kf_event_destroy(e0));
    check(kf_event_destroy(e1));
}
