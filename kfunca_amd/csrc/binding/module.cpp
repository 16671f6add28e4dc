// pybind11 module `kfunca_amd._C`: the Python surface of the reference's `kfunca` module
// (src/register.cpp:59-225) — same function names, arities and semantics — over the rebuilt host core.
// kfunca_amd/__init__.py re-exports it, so `import kfunca_amd as kfunca` runs the reference's tests.
#include <pybind11/numpy.h>
#include <cstdlib>
#include <pybind11/pybind11.h>
#include <pybind11/stl.h>

#include <iostream>
#include <vector>

#include "allocator.h"
#include "comm.h"
#include "device_api.h"
#include "ops.h"
#include "tensor.h"
#include "tensor_iterator.h"

namespace py = pybind11;

namespace {

struct NpType { char kind; int size; ScalarType st; };
const NpType kNpTypes[] = {{'b', 1, ScalarType::Bool}, {'u', 1, ScalarType::Byte}, {'i', 1, ScalarType::Char},
                           {'i', 2, ScalarType::Short}, {'i', 4, ScalarType::Int}, {'i', 8, ScalarType::Long},
                           {'f', 2, ScalarType::Half}, {'f', 4, ScalarType::Float}, {'f', 8, ScalarType::Double}};

Tensor from_numpy(py::array array, int device) {
    const char kind = array.dtype().kind();
    const int size = (int)array.dtype().itemsize();
    for (const auto &t : kNpTypes) {
        if (t.kind != kind || t.size != size) continue;
        py::array c = py::array::ensure(array, py::array::c_style); // the reference ignores strides (register.cpp:27-39); we honour them
        std::vector<int64_t> shape(c.shape(), c.shape() + c.ndim());
        Tensor out = empty(shape, t.st, device);
        if (out.numel() > 0) out.copy_from_cpu_ptr(const_cast<void *>(c.data()));
        return out;
    }
    throw std::runtime_error("Unsupported dtype in from_numpy()");
}

py::array to_numpy(const Tensor &t) {
    CHECK_FAIL(t.defined());
    CHECK_FAIL(t.is_dense()); // (the reference asks its flag, register.cpp:42: it refuses dense views too)
    const char *np = nullptr;
    switch (t.dtype()) {
    case ScalarType::Bool: np = "bool"; break;
    case ScalarType::Byte: np = "uint8"; break;
    case ScalarType::Char: np = "int8"; break;
    case ScalarType::Short: np = "int16"; break;
    case ScalarType::Int: np = "int32"; break;
    case ScalarType::Long: np = "int64"; break;
    case ScalarType::Half: np = "float16"; break;
    case ScalarType::Float: np = "float32"; break;
    case ScalarType::Double: np = "float64"; break;
    // numpy has no bfloat16: the raw bits come back as uint16 (x.view(np.uint32) << 16 is the f32 value; ml_dtypes users can
    // .view(ml_dtypes.bfloat16)). The reference's to_numpy rejects half and bfloat16 altogether (register.cpp:41-57).
    case ScalarType::BFloat16: np = "uint16"; break;
    default: throw std::runtime_error("Unsupported dtype in to_numpy()");
    }
    py::array out(py::dtype(np), t.sizes());
    // a contiguous view may sit inside a larger storage: copy exactly numel elements from data_ptr()
    const size_t bytes = (size_t)t.numel() * (size_t)t.element_size_in_bytes();
    if (bytes) DEV_CALL(kf_memcpy_d2h(out.mutable_data(), t.data_ptr(), bytes, dev::stream(t.device())));
    return out;
}

std::vector<int64_t> args_to_dims(const py::args &args) {
    std::vector<int64_t> dims;
    for (auto a : args) dims.push_back(a.cast<int64_t>());
    return dims;
}

Tensor get_item(Tensor &self, py::object key) {
    Tensor out = self;
    auto apply = [&](py::handle item, int &dim) {
        if (py::isinstance<py::slice>(item)) {
            size_t start, stop, step, len;
            if (!item.cast<py::slice>().compute(out.shape(dim), &start, &stop, &step, &len)) throw py::error_already_set();
            out = out.slice(dim, (int64_t)start, (int64_t)stop, (int64_t)step);
            ++dim;
        } else if (py::isinstance<py::int_>(item)) {
            out = out.select(dim, item.cast<int64_t>()); // the selected dim disappears
        } else {
            throw py::type_error("tensor indices must be ints or slices");
        }
    };
    int dim = 0;
    if (py::isinstance<py::tuple>(key)) {
        auto tup = key.cast<py::tuple>();
        CHECK_FAIL((int)tup.size() <= self.dim());
        for (auto item : tup) apply(item, dim);
    } else {
        apply(key, dim);
    }
    return out;
}

// CPU-only window onto the geometry engine: operands are (shape, strides_in_elements, dtype, defined)
py::dict iter_geometry(std::vector<std::tuple<std::vector<int64_t>, std::vector<int64_t>, ScalarType, bool>> outputs,
                       std::vector<std::tuple<std::vector<int64_t>, std::vector<int64_t>, ScalarType, bool>> inputs, bool is_reduction,
                       int64_t reduce_dim, bool resize_outputs, std::vector<int> aliases, bool check_mem_overlap) {
    IterGeometry g;
    std::vector<int> ids(outputs.size() + inputs.size());
    int slot = 0;
    auto add = [&](const auto &spec, bool is_out, int idx) {
        IterOperand op;
        op.defined = std::get<3>(spec);
        const auto &shape = std::get<0>(spec);
        const auto &stride = std::get<1>(spec);
        op.ndim = (int)shape.size();
        for (int d = 0; d < op.ndim; ++d) {
            op.shape[d] = shape[d];
            op.stride[d] = d < (int)stride.size() ? stride[d] : 0;
        }
        op.dtype = std::get<2>(spec);
        op.device = op.defined ? 0 : -1;
        op.data = reinterpret_cast<char *>(0x10000000ull * (uint64_t)(idx + 1));
        int alias = idx < (int)aliases.size() ? aliases[idx] : -1;
        ids[idx] = alias >= 0 ? alias : idx;
        op.identity = reinterpret_cast<const void *>((uintptr_t)(ids[idx] + 1));
        if (alias >= 0) op.data = reinterpret_cast<char *>(0x10000000ull * (uint64_t)(alias + 1));
        g.add(op, is_out);
    };
    for (auto &o : outputs) { add(o, true, slot); ++slot; }
    for (auto &i : inputs) { add(i, false, slot); ++slot; }
    std::vector<std::vector<int64_t>> allocated(outputs.size());
    auto alloc = [&](int arg, const int64_t *shape, int ndim, ScalarType dtype, int device, IterOperand &op) {
        op.defined = true;
        op.ndim = ndim;
        int64_t run = 1;
        for (int d = ndim - 1; d >= 0; --d) {
            op.shape[d] = shape[d];
            op.stride[d] = run;
            run *= shape[d];
        }
        op.dtype = dtype;
        op.device = device;
        op.data = reinterpret_cast<char *>(0x70000000ull + 0x1000000ull * (uint64_t)arg);
        allocated[arg].assign(shape, shape + ndim);
    };
    g.build(is_reduction, reduce_dim, resize_outputs, check_mem_overlap, alloc);
    py::dict r;
    std::vector<int64_t> shape, perm;
    for (int d = 0; d < g.ndim(); ++d) shape.push_back(g.shape(d));
    std::vector<std::vector<int64_t>> strides;
    for (int t = 0; t < g.ntensors(); ++t) {
        std::vector<int64_t> s;
        for (int d = 0; d < g.ndim(); ++d) s.push_back(g.stride_bytes(t, d));
        strides.push_back(s);
    }
    r["ndim"] = g.ndim();
    r["shape"] = shape;
    r["stride_bytes"] = strides;
    r["common_dtype"] = g.common_dtype();
    r["numel"] = g.numel();
    r["num_output_elements"] = g.num_output_elements();
    r["is_contiguous"] = g.is_contiguous();
    r["can_use_32bit_indexing"] = g.can_use_32bit_indexing();
    r["allocated"] = allocated;
    int pieces = 0;
    int64_t covered = 0;
    g.for_each_32bit([&](const IterGeometry &p) { ++pieces; covered += p.numel(); });
    r["pieces_32bit"] = pieces;
    r["pieces_numel"] = covered;
    return r;
}

} // namespace

PYBIND11_MODULE(_C, m) {
    py::register_exception<utils::Error>(m, "Error", PyExc_RuntimeError);

    m.def("device_info", []() {
        const int n = dev::device_count();
        std::cout << "HIP devices: " << n << "\n";
        for (int i = 0; i < n; ++i) {
            kf_device_props p;
            DEV_CALL(kf_device_props_get(i, &p));
            std::cout << "[" << i << "] " << p.name << " (" << p.arch << "), " << p.compute_units << " CUs, wave " << p.wavefront_size
                      << ", " << p.clock_khz / 1000 << " MHz, LDS/block " << p.lds_per_block << " B, L2 " << p.l2_bytes << " B, HBM "
                      << p.total_mem / (1ull << 30) << " GiB (free " << p.free_mem / (1ull << 30) << " GiB)\n";
        }
    });
    m.def("device_count", &dev::device_count);
    m.def("synchronize", [](int device) { dev::synchronize(device); }, py::arg("device") = 0);
    // HIP graphs over the device's stream (extension; the reference launches kernel by kernel, launcher_cuda.h:330-354): record
    // whatever the operator API enqueues between begin and end - a whole forward + backward step - and replay it with one
    // submission. Replays read and write the addresses captured, so the allocator runs in capture mode meanwhile: every block
    // the recorded step allocates or frees stays in a pool private to the graph (never re-enters the shared cache) until
    // graph_destroy - an eager op, a from_numpy or another step between two replays cannot be handed the graph's scratch.
    // Results the caller still holds when the capture ends are rewritten by every replay (that is the point of replaying).
    struct GraphHandle { void *exec; uint64_t pool; int device; };
    static uint64_t open_capture_pool = 0;
    m.def("graph_begin", [](int device) {
        void *st = dev::stream(device);
        open_capture_pool = utils::memory::DeviceAllocator::GetInstance()->begin_capture(device);
        const int rc = kf_graph_begin_capture(st);
        if (rc != KF_OK) {
            utils::memory::DeviceAllocator::GetInstance()->release_graph(open_capture_pool);
            open_capture_pool = 0;
            DEV_CALL(rc);
        }
    }, py::arg("device") = 0);
    m.def("graph_end", [](int device) {
        void *g = nullptr;
        const uint64_t pool = open_capture_pool;
        open_capture_pool = 0;
        const int rc = kf_graph_end_capture(dev::stream(device), &g);
        utils::memory::DeviceAllocator::GetInstance()->end_capture(pool);
        if (rc != KF_OK) {
            utils::memory::DeviceAllocator::GetInstance()->release_graph(pool);
            DEV_CALL(rc);
        }
        return reinterpret_cast<uintptr_t>(new GraphHandle{g, pool, device});
    }, py::arg("device") = 0);
    m.def("graph_launch", [](uintptr_t g, int device) {
        auto *h = reinterpret_cast<GraphHandle *>(g);
        CHECK_FAIL(h && h->exec, "graph_launch: null graph");
        DEV_CALL(kf_graph_launch(h->exec, dev::stream(device)));
    }, py::arg("graph"), py::arg("device") = 0);
    m.def("graph_destroy", [](uintptr_t g) {
        auto *h = reinterpret_cast<GraphHandle *>(g);
        if (!h) return;
        dev::synchronize(h->device); // no replay may still be running when its scratch goes back to the shared cache
        DEV_CALL(kf_graph_destroy(h->exec));
        utils::memory::DeviceAllocator::GetInstance()->release_graph(h->pool);
        delete h;
    });
    m.def("memstat", []() { utils::memory::DeviceAllocator::GetInstance()->print(); });
    m.def("memstat_dict", [](int device) {
        auto s = utils::memory::DeviceAllocator::GetInstance()->stats(device);
        py::dict d;
        d["active_blocks"] = s.active_blocks;
        d["cached_blocks"] = s.cached_blocks;
        d["active_bytes"] = s.active_bytes;
        d["cached_bytes"] = s.cached_bytes;
        d["driver_allocs"] = s.driver_allocs;
        d["graph_blocks"] = s.graph_blocks;
        d["graph_bytes"] = s.graph_bytes;
        return d;
    }, py::arg("device") = -1);

    py::enum_<ScalarType>(m, "dtype", py::module_local()) // module-local: the reference's own module (oracle/_ref) binds the same C++ names
        .value("byte", ScalarType::Byte)
        .value("char", ScalarType::Char)
        .value("short", ScalarType::Short)
        .value("int", ScalarType::Int)
        .value("long", ScalarType::Long)
        .value("half", ScalarType::Half)
        .value("bfloat16", ScalarType::BFloat16)
        .value("float", ScalarType::Float)
        .value("double", ScalarType::Double)
        .value("bool", ScalarType::Bool)
        .export_values();

    m.def("empty", [](std::vector<int64_t> shape, ScalarType dtype, int device) { return empty(shape, dtype, device); });
    m.def("empty_like", [](const Tensor &self) { return empty_like(self); });
    m.def("from_numpy", &from_numpy);
    m.def("to_numpy", &to_numpy);
    m.def("zeros", [](std::vector<int64_t> shape, ScalarType dtype, int device) { return zeros(shape, dtype, device); });
    m.def("causal_attention", &gpu::causal_attention);
    // the reference's roadmap operators (README.md:28-30)
    // ---- collectives (extension: the reference has no distributed code): one process per GPU, one RCCL communicator
    // per process through the C ABI; the 128-byte id travels out of band (kfunca_amd.parallel uses torch.distributed / gloo for that)
    m.def("comm_unique_id", []() { return py::bytes(gpu::comm_unique_id()); });
    m.def("comm_init", [](py::bytes id, int rank, int world_size, int device) { gpu::comm_init(std::string(id), rank, world_size, device); },
          py::arg("unique_id"), py::arg("rank"), py::arg("world_size"), py::arg("device") = 0);
    m.def("comm_destroy", &gpu::comm_destroy);
    m.def("comm_initialized", &gpu::comm_initialized);
    m.def("comm_rank", &gpu::comm_rank);
    m.def("comm_world_size", &gpu::comm_world_size);
    m.def("all_reduce_", [](py::object x) { // a tensor, or a list of tensors reduced as ONE collective launch; in place, returns its argument
        if (py::isinstance<py::list>(x) || py::isinstance<py::tuple>(x)) {
            std::vector<Tensor> ts = x.cast<std::vector<Tensor>>();
            gpu::all_reduce_(ts);
        } else {
            Tensor t = x.cast<Tensor>();
            gpu::all_reduce_(t);
        }
        return x;
    });
    py::class_<gpu::GradBucket, std::shared_ptr<gpu::GradBucket>>(m, "GradBucket", py::module_local())
        .def(py::init([](const std::vector<Tensor> &params, double cap_mb, bool accum_f32) {
                 return gpu::GradBucket::create(params, (int64_t)(cap_mb * 1048576.0), accum_f32);
             }),
             py::arg("params"), py::arg("cap_mb") = 256.0, py::arg("accum_f32") = false)
        .def("attach", &gpu::GradBucket::attach)
        .def("detach", &gpu::GradBucket::detach)
        .def("wait", &gpu::GradBucket::wait)
        .def("flat", &gpu::GradBucket::flat)
        .def("reduced_bytes", &gpu::GradBucket::reduced_bytes)
        .def("set_collectives", &gpu::GradBucket::set_collectives)
        .def("chunk_ms", &gpu::GradBucket::chunk_ms)
        .def("fired_order", [](const gpu::GradBucket &b) { return b.fired_order(); })
        .def("chunks", [](const gpu::GradBucket &b) {
            py::list out;
            for (const auto &c : b.chunks()) out.append(py::make_tuple(c.first, c.last, c.offset, c.numel));
            return out;
        })
        // the layout arithmetic alone (no device): [(first, last, offset, numel)], chunk 0 = the LAST parameters
        .def_static("plan", [](const std::vector<int64_t> &numels, int64_t cap_elements) {
            py::list out;
            for (const auto &c : gpu::GradBucket::plan(numels, cap_elements)) out.append(py::make_tuple(c.first, c.last, c.offset, c.numel));
            return out;
        })
        // the firing bookkeeping alone (no device; the code arrived() / wait() run): chunk indices in the order their collectives leave
        .def_static("simulate_fired_order", &gpu::GradBucket::simulate_fired_order, py::arg("numels"), py::arg("cap_elements"), py::arg("arrivals"));

    m.def("rms_norm", [](const Tensor &x, py::object w, double eps) { return gpu::rms_norm(x, w.is_none() ? Tensor() : w.cast<Tensor>(), eps); },
          py::arg("x"), py::arg("weight") = py::none(), py::arg("eps") = 1e-5);
    m.def("layer_norm", [](const Tensor &x, py::object w, py::object b, double eps) {
        return gpu::layer_norm(x, w.is_none() ? Tensor() : w.cast<Tensor>(), b.is_none() ? Tensor() : b.cast<Tensor>(), eps);
    }, py::arg("x"), py::arg("weight") = py::none(), py::arg("bias") = py::none(), py::arg("eps") = 1e-5);
    m.def("embedding", &gpu::embedding, py::arg("table"), py::arg("indices"));
    m.def("gemm_fused", [](const Tensor &a, const Tensor &b, float alpha, py::object bias, py::object mul, py::object add) {
        auto opt = [](py::object o) { return o.is_none() ? Tensor() : o.cast<Tensor>(); };
        return gpu::gemm_fused(a, b, alpha, opt(bias), opt(mul), opt(add));
    }, py::arg("a"), py::arg("b"), py::arg("alpha") = 1.0f, py::arg("bias") = py::none(), py::arg("mul") = py::none(), py::arg("add") = py::none());
    // the roadmap's name (README.md:32): the fused QKV projection = one GEMM with the bias in its store; its packed [B*S, 3*H*D] output
    // is what causal_attention_qkv reads in place
    m.def("qkv_linear", [](const Tensor &x, const Tensor &w_qkv, py::object bias) {
        return gpu::gemm_fused(x, w_qkv, 1.0f, bias.is_none() ? Tensor() : bias.cast<Tensor>(), Tensor(), Tensor());
    }, py::arg("x"), py::arg("w_qkv"), py::arg("bias") = py::none());
    m.def("causal_attention_qkv", &gpu::causal_attention_qkv, py::arg("qkv"), py::arg("B"), py::arg("S"), py::arg("H"));
    // from_numpy for bfloat16: uint16 bit patterns in, a BFloat16 tensor out (the inverse of to_numpy's uint16 view)
    m.def("from_numpy_bf16", [](py::array array, int device) {
        CHECK_FAIL(array.dtype().kind() == 'u' && array.dtype().itemsize() == 2, "from_numpy_bf16 expects uint16 bit patterns");
        py::array c = py::array::ensure(array, py::array::c_style);
        std::vector<int64_t> shape(c.shape(), c.shape() + c.ndim());
        Tensor out = empty(shape, ScalarType::BFloat16, device);
        if (out.numel() > 0) out.copy_from_cpu_ptr(const_cast<void *>(c.data()));
        return out;
    }, py::arg("array"), py::arg("device") = 0);
    m.def("gemm", &gpu::gemm);
    m.def("cat", &gpu::concat);
    m.def("_iter_geometry", &iter_geometry, py::arg("outputs"), py::arg("inputs"), py::arg("is_reduction") = false,
          py::arg("reduce_dim") = 0, py::arg("resize_outputs") = true, py::arg("aliases") = std::vector<int>(), py::arg("check_mem_overlap") = true);
    m.def("_promote_types", &promote_types);
    m.def("_pool_index", &utils::memory::DeviceAllocator::pool_index);
    m.def("release_cached", [](int device) { return utils::memory::DeviceAllocator::GetInstance()->release_cached(device); }, py::arg("device") = 0);
    // test hook (tests/test_gpu_host_api.py: a simulated shortage): refuses to arm unless the process says it is a test
    m.def("_alloc_fail_above", [](size_t bytes) {
        CHECK_FAIL(bytes == 0 || std::getenv("KF_TEST_HOOKS") != nullptr, "_alloc_fail_above is a test hook: set KF_TEST_HOOKS=1");
        utils::memory::DeviceAllocator::GetInstance()->debug_fail_above(bytes);
    });
    m.def("_alloc_oom_retries", []() { return utils::memory::DeviceAllocator::GetInstance()->oom_retries(); });

    py::class_<Tensor>(m, "tensor", py::module_local())
        .def("__copy__", [](const Tensor &self) { return Tensor(self); })
        .def("__deepcopy__", [](const Tensor &self, py::dict) { return Tensor(self); })
        .def("__repr__", &Tensor::to_string)
        .def("defined", &Tensor::defined)
        .def("numpy", [](const Tensor &self) { return to_numpy(self); })
        .def("numel", &Tensor::numel)
        .def("dim", &Tensor::dim)
        .def("device", &Tensor::device)
        .def("shape", [](const Tensor &self, int64_t d) { return self.shape((int)d); })
        .def("sizes", [](const Tensor &self) { return self.sizes(); })
        .def("strides", [](const Tensor &self) { return self.strides(); })
        .def("storage_offset", &Tensor::storage_offset)
        .def("is_contiguous", &Tensor::is_contiguous)
        .def("dtype", &Tensor::dtype)
        .def("item", [](Tensor &self, std::vector<int64_t> indices) -> py::object {
            any_t raw = self.item(indices);
            switch (self.dtype()) {
            case ScalarType::Bool: return py::cast(*reinterpret_cast<uint8_t *>(raw.val) != 0);
            case ScalarType::Byte: return py::cast(*reinterpret_cast<uint8_t *>(raw.val));
            case ScalarType::Char: return py::cast(*reinterpret_cast<int8_t *>(raw.val));
            case ScalarType::Short: return py::cast(*reinterpret_cast<int16_t *>(raw.val));
            case ScalarType::Int: return py::cast(*reinterpret_cast<int32_t *>(raw.val));
            case ScalarType::Long: return py::cast(*reinterpret_cast<int64_t *>(raw.val));
            case ScalarType::Half: return py::cast(dtype::f16_bits_to_float(*reinterpret_cast<uint16_t *>(raw.val)));
            case ScalarType::BFloat16: return py::cast(dtype::bf16_bits_to_float(*reinterpret_cast<uint16_t *>(raw.val)));
            case ScalarType::Float: return py::cast(*reinterpret_cast<float *>(raw.val));
            case ScalarType::Double: return py::cast(*reinterpret_cast<double *>(raw.val));
            default: return py::none();
            }
        })
        .def("fill_", [](Tensor &self, double value) { return self.fill_(any_t{value}); })
        .def("data_ptr", [](Tensor &self) -> uintptr_t { return reinterpret_cast<uintptr_t>(self.data_ptr()); })
        .def("storage_ref_count", &Tensor::storage_ref_count)
        .def("impl_ref_count", &Tensor::impl_ref_count)
        .def("contiguous", &Tensor::contiguous)
        .def("permute", [](Tensor &self, py::args args) {
            CHECK_FAIL((int)args.size() == self.dim());
            return self.permute(args_to_dims(args));
        })
        .def("view", [](Tensor &self, py::args args) { return self.view(args_to_dims(args)); })
        .def("split", &Tensor::split)
        .def("sort", &Tensor::sort)
        .def("topk", &Tensor::topk)
        .def("__getitem__", &get_item)
        .def("__add__", &Tensor::operator+)
        .def("__add__", [](const Tensor &self, double s) { return gpu::add(self, s); })
        .def("__iadd__", &Tensor::operator+=)
        .def("__iadd__", [](Tensor &self, double s) { gpu::add_(self, s); return self; })
        .def("__sub__", &Tensor::operator-)
        .def("__sub__", [](const Tensor &self, double s) { return gpu::sub(self, s); })
        .def("__isub__", &Tensor::operator-=)
        .def("__isub__", [](Tensor &self, double s) { gpu::sub_(self, s); return self; })
        .def("__mul__", &Tensor::operator*)
        .def("__mul__", [](const Tensor &self, double s) { return gpu::mul(self, s); })
        .def("__imul__", &Tensor::operator*=)
        .def("__imul__", [](Tensor &self, double s) { gpu::mul_(self, s); return self; })
        .def("__truediv__", &Tensor::operator/)
        .def("__truediv__", [](const Tensor &self, double s) { return gpu::div(self, s); })
        .def("__itruediv__", &Tensor::operator/=)
        .def("__itruediv__", [](Tensor &self, double s) { gpu::div_(self, s); return self; })
        .def("sum", &Tensor::sum)
        .def("mean", &Tensor::mean)
        .def("mean_var", &Tensor::mean_var)
        .def("norm_stat", &Tensor::norm_stat)
        .def("index_put_", &Tensor::index_put_)
        .def("half", &Tensor::_half)
        .def("bfloat16", &Tensor::_bfloat16)
        .def("float", &Tensor::_float)
        .def("requires_grad", &Tensor::requires_grad)
        .def("set_requires_grad", &Tensor::set_requires_grad)
        .def("backward", &Tensor::backward)
        // extension: drop the accumulated gradient (an optimizer's zero_grad(set_to_none)); the next backward's first gradient is
        // then taken over without a copy when nothing else refers to it
        .def("zero_grad", [](Tensor &self) { self.impl()->grad_.reset(); })
        .def("grad", [](Tensor &self) {
            if (self.grad() && self.grad()->defined()) return *self.grad();
            return Tensor();
        });
}
