"""ctypes binding of the device C ABI (include/kfunca_hip.h -> kfunca_amd/libkfunca_hip.so).

This is the thinnest possible host over the drop-in boundary: the parity tests and bench.py call
the hand-written HIP kernels through it with plain device pointers, exactly as a foreign host
(cgo, JNI, ctypes …) would. There is no fallback of any kind: if the library is missing or a call
fails, an exception is raised.
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

import os

PKG = Path(__file__).resolve().parent
# KF_HIP_LIB: load another build of the same library (A/B runs of two kernel variants on one box, tools/ only)
LIB_PATH = Path(os.environ["KF_HIP_LIB"]) if os.environ.get("KF_HIP_LIB") else PKG / "libkfunca_hip.so"

# dtype codes == reference ScalarType order (src/core/include/scalar_type.h:9-27)
BOOL, U8, I8, I16, I32, I64, F16, BF16, F32, F64 = range(10)
DTYPE_SIZE = {BOOL: 1, U8: 1, I8: 1, I16: 2, I32: 4, I64: 8, F16: 2, BF16: 2, F32: 4, F64: 8}
EW_ADD, EW_SUB, EW_MUL, EW_DIV, EW_COPY, EW_FILL, EW_ADD_SCALAR, EW_SUB_SCALAR, EW_MUL_SCALAR, EW_DIV_SCALAR = range(10)
RED_SUM, RED_MEAN = range(2)
MOM_VAR, MOM_STD, MOM_INVSTD = range(3)
EPI_NONE, EPI_BIAS_ROW = range(2)
NORM_RMS, NORM_LAYER = range(2)
MAX_DIMS, MAX_TENSORS = 12, 8
KF_OK, KF_ERR_HIP, KF_ERR_INVALID, KF_ERR_UNSUPPORTED, KF_ERR_INDEX_RANGE, KF_ERR_WORKSPACE, KF_ERR_COMM, KF_ERR_OOM = range(8)
COMM_ID_BYTES = 128

NP2CODE = {np.dtype(np.bool_): BOOL, np.dtype(np.uint8): U8, np.dtype(np.int8): I8, np.dtype(np.int16): I16,
           np.dtype(np.int32): I32, np.dtype(np.int64): I64, np.dtype(np.float16): F16,
           np.dtype(np.float32): F32, np.dtype(np.float64): F64}
CODE2NP = {BOOL: np.bool_, U8: np.uint8, I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64,
           F16: np.float16, BF16: np.uint16, F32: np.float32, F64: np.float64}

EXPORTS = [
    "kf_last_error", "kf_abi_version", "kf_build_source_sha", "kf_device_count", "kf_set_device", "kf_get_device", "kf_malloc", "kf_free",
    "kf_memcpy_h2d", "kf_memcpy_d2h", "kf_memcpy_d2d", "kf_memset_zero", "kf_stream_create", "kf_stream_destroy",
    "kf_stream_sync", "kf_stream_wait_event", "kf_device_sync", "kf_event_create", "kf_event_destroy", "kf_event_record", "kf_event_sync",
    "kf_event_elapsed_ms", "kf_graph_begin_capture", "kf_graph_end_capture", "kf_graph_launch", "kf_graph_destroy", "kf_profile_enable", "kf_profile_reset", "kf_profile_count", "kf_profile_get", "kf_profile_samples", "kf_knobs_reload",
    "kf_device_props_get", "kf_elementwise", "kf_reduce_workspace_bytes", "kf_reduce",
    "kf_reduce_moments_workspace_bytes", "kf_reduce_moments",
    "kf_norm_fwd", "kf_norm_bwd_workspace_bytes", "kf_norm_bwd",
    "kf_index_put", "kf_index_get", "kf_index_add_workspace_bytes", "kf_index_add", "kf_sort_workspace_bytes", "kf_sort", "kf_gemm_workspace_bytes", "kf_gemm", "kf_gemm_ex", "kf_gemm_grouped", "kf_gemm_grouped_single_grid", "kf_attn_fwd", "kf_attn_fwd_scaled", "kf_attn_bwd_workspace_bytes",
    "kf_attn_bwd", "kf_attn_bwd_scaled", "kf_attn_fwd_strided", "kf_attn_bwd_strided", "kf_comm_unique_id", "kf_comm_init", "kf_comm_destroy", "kf_allreduce_sum", "kf_allreduce_sum_multi",
]


class KfError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"kfunca_hip status {code}: {msg}")
        self.code = code


class IterDesc(C.Structure):
    """kf_iter_desc: post-build() TensorIterator state as a POD."""
    _fields_ = [("ndim", C.c_int32), ("ntensors", C.c_int32), ("noutputs", C.c_int32), ("reserved", C.c_int32),
                ("dtype", C.c_int32 * MAX_TENSORS), ("shape", C.c_int64 * MAX_DIMS),
                ("stride_bytes", (C.c_int64 * MAX_DIMS) * MAX_TENSORS), ("data", C.c_void_p * MAX_TENSORS)]


class GemmEpilogue(C.Structure):
    """kf_gemm_epilogue: C = (alpha AB + beta C + bias) o mul + add, aux = the value in brackets."""
    _fields_ = [("bias", C.c_void_p), ("mul", C.c_void_p), ("ldmul", C.c_int64), ("add", C.c_void_p), ("ldadd", C.c_int64),
                ("aux", C.c_void_p), ("ldaux", C.c_int64), ("c_f32", C.c_int32)]


class GemmProblem(C.Structure):
    """kf_gemm_problem: one product of a kf_gemm_grouped call."""
    _fields_ = [("trans_a", C.c_int32), ("trans_b", C.c_int32), ("M", C.c_int64), ("N", C.c_int64), ("K", C.c_int64), ("alpha", C.c_float),
                ("beta", C.c_float), ("A", C.c_void_p), ("lda", C.c_int64), ("B", C.c_void_p), ("ldb", C.c_int64), ("C", C.c_void_p), ("ldc", C.c_int64),
                ("c_f32", C.c_int32)]


class AttnLayout(C.Structure):
    """kf_attn_layout: element strides of the batch, head and row dims of one attention operand."""
    _fields_ = [("batch", C.c_int64), ("head", C.c_int64), ("row", C.c_int64)]


class DeviceProps(C.Structure):
    _fields_ = [("name", C.c_char * 256), ("arch", C.c_char * 64), ("compute_units", C.c_int32),
                ("wavefront_size", C.c_int32), ("max_threads_per_block", C.c_int32), ("clock_khz", C.c_int32),
                ("memory_clock_khz", C.c_int32), ("memory_bus_bits", C.c_int32), ("lds_per_block", C.c_int64),
                ("l2_bytes", C.c_int64), ("total_mem", C.c_uint64), ("free_mem", C.c_uint64)]


_lib = None


def build_source_sha() -> str:
    """The device-source hash linked into the loaded library (kf_build_source_sha): compare with kfunca_amd._build.device_src_sha()."""
    l = lib()
    return l.kf_build_source_sha().decode() if hasattr(l, "kf_build_source_sha") else "unstamped"


def lib():
    """Load libkfunca_hip.so; raises if it has not been built (python -m kfunca_amd._build)."""
    global _lib
    if _lib is None:
        if not LIB_PATH.exists():
            raise ImportError(f"{LIB_PATH} is missing: build it with `python -m kfunca_amd._build` "
                              "(there is no CPU fallback)")
        _lib = C.CDLL(str(LIB_PATH))
        _lib.kf_last_error.restype = C.c_char_p
        if hasattr(_lib, "kf_build_source_sha"):   # (a diagnostic variant library under KF_HIP_LIB is linked without the stamp)
            _lib.kf_build_source_sha.restype = C.c_char_p
            # a library that was NOT built from the sources beside it must not pass for them (a failed rebuild leaves the old .so behind):
            # refuse it unless told otherwise (KF_ALLOW_STALE_LIB=1)
            from . import _build
            if (_build.CSRC / "device").exists() and not os.environ.get("KF_ALLOW_STALE_LIB"):
                built, tree = _lib.kf_build_source_sha().decode(), _build.device_src_sha()
                if built != tree:
                    _lib = None
                    raise ImportError(f"{LIB_PATH} was built from device sources {built}, the tree is {tree}: rebuild it (python -m kfunca_amd._build); "
                                      "KF_ALLOW_STALE_LIB=1 loads it anyway")
        vp, i64, sz = C.c_void_p, C.c_int64, C.c_size_t
        _lib.kf_malloc.argtypes = [C.POINTER(vp), sz]
        _lib.kf_free.argtypes = [vp]
        for n in ("kf_memcpy_h2d", "kf_memcpy_d2h", "kf_memcpy_d2d"):
            getattr(_lib, n).argtypes = [vp, vp, sz, vp]
        _lib.kf_memset_zero.argtypes = [vp, sz, vp]
        _lib.kf_stream_create.argtypes = [C.POINTER(vp)]
        _lib.kf_stream_destroy.argtypes = [vp]
        _lib.kf_stream_sync.argtypes = [vp]
        _lib.kf_stream_wait_event.argtypes = [vp, vp]
        _lib.kf_event_create.argtypes = [C.POINTER(vp)]
        _lib.kf_event_destroy.argtypes = [vp]
        _lib.kf_event_record.argtypes = [vp, vp]
        _lib.kf_event_sync.argtypes = [vp]
        _lib.kf_event_elapsed_ms.argtypes = [vp, vp, C.POINTER(C.c_float)]
        _lib.kf_graph_begin_capture.argtypes = [vp]
        _lib.kf_graph_end_capture.argtypes = [vp, C.POINTER(vp)]
        _lib.kf_graph_launch.argtypes = [vp, vp]
        _lib.kf_graph_destroy.argtypes = [vp]
        _lib.kf_device_props_get.argtypes = [C.c_int, C.POINTER(DeviceProps)]
        _lib.kf_elementwise.argtypes = [C.c_int, C.POINTER(IterDesc), C.c_int, C.c_double, vp]
        _lib.kf_reduce_workspace_bytes.argtypes = [C.POINTER(IterDesc), C.POINTER(sz)]
        _lib.kf_reduce.argtypes = [C.c_int, C.POINTER(IterDesc), vp, sz, vp]
        _lib.kf_reduce_moments_workspace_bytes.argtypes = [C.POINTER(IterDesc), C.POINTER(sz)]
        _lib.kf_reduce_moments.argtypes = [C.c_int, C.POINTER(IterDesc), C.c_double, C.c_double, vp, sz, vp]
        _lib.kf_norm_fwd.argtypes = [C.c_int, C.c_int, i64, i64, i64, vp, vp, vp, C.c_double, vp, vp, vp, vp]
        _lib.kf_norm_bwd_workspace_bytes.argtypes = [C.c_int, C.c_int, i64, i64, i64, C.POINTER(sz)]
        _lib.kf_norm_bwd.argtypes = [C.c_int, C.c_int, i64, i64, i64] + [vp] * 9 + [sz, vp]
        _lib.kf_index_put.argtypes = [C.POINTER(IterDesc), C.c_int, C.POINTER(i64), C.POINTER(i64), vp]
        _lib.kf_index_get.argtypes = [vp, i64, i64, vp, i64, vp, vp]
        _lib.kf_index_add_workspace_bytes.argtypes = [i64]
        _lib.kf_index_add_workspace_bytes.restype = sz
        _lib.kf_index_add.argtypes = [C.c_int, vp, i64, vp, i64, i64, vp, vp, sz, vp]
        _lib.kf_sort_workspace_bytes.argtypes = [C.c_int, i64, i64]
        _lib.kf_sort_workspace_bytes.restype = sz
        _lib.kf_sort.argtypes = [C.c_int, vp, vp, vp, i64, i64, C.c_int, vp, sz, vp]
        _lib.kf_gemm_workspace_bytes.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, C.POINTER(sz)]
        _lib.kf_gemm.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, C.c_float, vp, i64, vp, i64, C.c_float,
                                 vp, i64, C.c_int, vp, vp, sz, vp]
        _lib.kf_gemm_ex.argtypes = [C.c_int, C.c_int, C.c_int, i64, i64, i64, C.c_float, vp, i64, vp, i64, C.c_float, vp, i64,
                                    C.POINTER(GemmEpilogue), vp]
        _lib.kf_gemm_grouped.argtypes = [C.c_int, C.c_int, C.POINTER(GemmProblem), vp]
        _lib.kf_attn_fwd.argtypes = [C.c_int, i64, i64, i64, i64, i64, vp, vp, vp, vp, vp, vp]
        _lib.kf_attn_bwd_workspace_bytes.argtypes = [C.c_int, i64, i64, i64, i64, i64, C.POINTER(sz)]
        _lib.kf_attn_bwd.argtypes = [C.c_int, i64, i64, i64, i64, i64] + [vp] * 10 + [sz, vp]
        _lib.kf_attn_fwd_scaled.argtypes = [C.c_int, i64, i64, i64, i64, i64, C.c_float, vp, vp, vp, vp, vp, vp]
        _lib.kf_attn_bwd_scaled.argtypes = [C.c_int, i64, i64, i64, i64, i64, C.c_float] + [vp] * 10 + [sz, vp]
        lp = C.POINTER(AttnLayout)
        _lib.kf_attn_fwd_strided.argtypes = [C.c_int, i64, i64, i64, i64, i64, C.c_float, vp, lp, vp, lp, vp, lp, vp, lp, vp, vp]
        _lib.kf_attn_bwd_strided.argtypes = [C.c_int, i64, i64, i64, i64, i64, C.c_float, vp, lp, vp, lp, vp, lp, vp, lp, vp, vp, lp, vp, lp, vp, lp,
                                             vp, lp, vp, sz, vp]
        _lib.kf_comm_unique_id.argtypes = [C.c_char_p]
        _lib.kf_comm_init.argtypes = [C.POINTER(vp), C.c_char_p, C.c_int, C.c_int]
        _lib.kf_comm_destroy.argtypes = [vp]
        _lib.kf_allreduce_sum.argtypes = [vp, vp, sz, C.c_int, vp]
    return _lib


def check(rc):
    if rc != KF_OK:
        raise KfError(rc, lib().kf_last_error().decode(errors="replace"))


def device_count() -> int:
    n = C.c_int(0)
    rc = lib().kf_device_count(C.byref(n))
    return n.value if rc == KF_OK else 0


def set_device(i: int):
    check(lib().kf_set_device(int(i)))


def device_props(i: int = 0) -> DeviceProps:
    p = DeviceProps()
    check(lib().kf_device_props_get(int(i), C.byref(p)))
    return p


class Stream:
    def __init__(self):
        h = C.c_void_p()
        check(lib().kf_stream_create(C.byref(h)))
        self.handle = h.value

    def sync(self):
        check(lib().kf_stream_sync(self.handle))

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.kf_stream_destroy(self.handle)
            self.handle = None


class Graph:
    """A captured sequence of library calls on one stream (kf_graph_*): `with Graph.capture(stream) as g: ...` then g.launch()."""

    def __init__(self, stream: Stream):
        self.stream, self.handle = stream, None

    @classmethod
    def capture(cls, stream: Stream):
        return cls(stream)

    def __enter__(self):
        check(lib().kf_graph_begin_capture(self.stream.handle))
        return self

    def __exit__(self, et, ev, tb):
        h = C.c_void_p()
        rc = lib().kf_graph_end_capture(self.stream.handle, C.byref(h))
        if et is None:
            check(rc)
            self.handle = h.value
        return False

    def launch(self, stream: Stream = None):
        check(lib().kf_graph_launch(self.handle, (stream or self.stream).handle))

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.kf_graph_destroy(self.handle)
            self.handle = None


class Event:
    def __init__(self):
        h = C.c_void_p()
        check(lib().kf_event_create(C.byref(h)))
        self.handle = h.value

    def record(self, stream=None):
        check(lib().kf_event_record(self.handle, stream))

    def sync(self):
        check(lib().kf_event_sync(self.handle))

    def elapsed_ms(self, later: "Event") -> float:
        ms = C.c_float(0)
        check(lib().kf_event_elapsed_ms(self.handle, later.handle, C.byref(ms)))
        return ms.value

    def __del__(self):
        if getattr(self, "handle", None) and _lib is not None:
            _lib.kf_event_destroy(self.handle)
            self.handle = None


class DevBuf:
    """A raw device allocation (kf_malloc / kf_free)."""

    def __init__(self, nbytes: int):
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(lib().kf_malloc(C.byref(p), max(self.nbytes, 1)))
        self.ptr = p.value

    @classmethod
    def from_numpy(cls, arr: np.ndarray) -> "DevBuf":
        arr = np.ascontiguousarray(arr)
        b = cls(arr.nbytes)
        if arr.nbytes:
            check(lib().kf_memcpy_h2d(b.ptr, arr.ctypes.data, arr.nbytes, None))
        return b

    def to_numpy(self, shape, dtype) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes, (out.nbytes, self.nbytes)
        if out.nbytes:
            check(lib().kf_memcpy_d2h(out.ctypes.data, self.ptr, out.nbytes, None))
        return out

    def zero(self, stream=None):
        check(lib().kf_memset_zero(self.ptr, self.nbytes, stream))

    def free(self):
        if getattr(self, "ptr", None) and _lib is not None:
            _lib.kf_free(self.ptr)
            self.ptr = None

    def __del__(self):
        self.free()


class View:
    """A strided typed view of device memory (what the host core's Tensor carries)."""

    def __init__(self, ptr: int, shape, strides_elems, code: int):
        self.ptr, self.shape, self.strides, self.code = int(ptr), tuple(shape), tuple(strides_elems), int(code)

    @classmethod
    def of(cls, buf: DevBuf, arr: np.ndarray, code=None, byte_offset=0):
        """View with the geometry of `arr` (possibly a non-contiguous numpy view of the uploaded base)."""
        code = NP2CODE[arr.dtype] if code is None else code
        return cls(buf.ptr + byte_offset, arr.shape, [s // arr.itemsize for s in arr.strides], code)


def make_desc(outputs, inputs, shape=None, coalesce=True) -> IterDesc:
    """Build a kf_iter_desc the way the reference's TensorIterator::build() leaves it for plain
    loops (tensor_iterator.cpp:486-515), restricted to what a test needs: same-ndim broadcast,
    byte strides, dims reversed to fastest-first, adjacent dims coalesced."""
    ops = list(outputs) + list(inputs)
    nd = len(ops[0].shape)
    if shape is None:
        shape = [max(o.shape[i] for o in ops) for i in range(nd)]
    dims = []
    for i in reversed(range(nd)):
        st = []
        for o in ops:
            es = DTYPE_SIZE[o.code]
            st.append(0 if (o.shape[i] == 1 and shape[i] != 1) else o.strides[i] * es)
        dims.append([shape[i], st])
    if coalesce and len(dims) > 1:
        merged = [dims[0]]
        for sz, st in dims[1:]:
            psz, pst = merged[-1]
            if psz == 1:
                merged[-1] = [sz, st]
            elif sz == 1:
                pass
            elif all(psz * a == b for a, b in zip(pst, st)):
                merged[-1][0] = psz * sz
            else:
                merged.append([sz, st])
        dims = merged
    if not dims:
        dims = [[1, [DTYPE_SIZE[o.code] for o in ops]]]
    d = IterDesc()
    d.ndim, d.ntensors, d.noutputs = len(dims), len(ops), len(outputs)
    for t, o in enumerate(ops):
        d.dtype[t] = o.code
        d.data[t] = o.ptr
    for i, (sz, st) in enumerate(dims):
        d.shape[i] = sz
        for t in range(len(ops)):
            d.stride_bytes[t][i] = st[t]
    return d


def make_reduce_desc(out: View, inp: View, dim: int) -> IterDesc:
    """Descriptor as build_for_reduce leaves it (tensor_iterator.cpp:181-244,523-528): the reduced
    dim gets output stride 0 and moves to the front, the rest are ordered by input stride."""
    nd = len(inp.shape)
    es = DTYPE_SIZE[inp.code]
    dims = []
    for i in range(nd):
        o_st = 0 if i == dim and inp.shape[i] != 1 else out.strides[i] * DTYPE_SIZE[out.code]
        dims.append((inp.shape[i], o_st, inp.strides[i] * es, i == dim and inp.shape[i] != 1))
    red = [x for x in dims if x[3]]
    rest = sorted([x for x in dims if not x[3]], key=lambda x: x[2])
    order = red + rest
    merged = []
    for sz, o_st, i_st, r in order:
        if merged and not r and not merged[-1][3]:
            psz, po, pi, _ = merged[-1]
            if psz * po == o_st and psz * pi == i_st:
                merged[-1] = (psz * sz, po, pi, False)
                continue
            if psz == 1:
                merged[-1] = (sz, o_st, i_st, False)
                continue
            if sz == 1:
                continue
        merged.append((sz, o_st, i_st, r))
    d = IterDesc()
    d.ndim, d.ntensors, d.noutputs = len(merged), 2, 1
    d.dtype[0], d.dtype[1] = out.code, inp.code
    d.data[0], d.data[1] = out.ptr, inp.ptr
    for i, (sz, o_st, i_st, _) in enumerate(merged):
        d.shape[i] = sz
        d.stride_bytes[0][i] = o_st
        d.stride_bytes[1][i] = i_st
    return d


def make_moments_desc(out0: View, out1: View, inp: View, dim: int) -> IterDesc:
    """Two-output form of make_reduce_desc (reduce_ops.cpp:24-25: outputs var, mean, then the input)."""
    d1 = make_reduce_desc(out0, inp, dim)
    d2 = make_reduce_desc(out1, inp, dim)
    d = IterDesc()
    d.ndim, d.ntensors, d.noutputs = d1.ndim, 3, 2
    d.dtype[0], d.dtype[1], d.dtype[2] = out0.code, out1.code, inp.code
    d.data[0], d.data[1], d.data[2] = out0.ptr, out1.ptr, inp.ptr
    for i in range(d1.ndim):
        d.shape[i] = d1.shape[i]
        d.stride_bytes[0][i] = d1.stride_bytes[0][i]
        d.stride_bytes[1][i] = d2.stride_bytes[0][i]
        d.stride_bytes[2][i] = d1.stride_bytes[1][i]
    return d


def reduce_moments(mode, desc: IterDesc, correction=1.0, eps=0.0, stream=None):
    need = C.c_size_t(0)
    check(lib().kf_reduce_moments_workspace_bytes(C.byref(desc), C.byref(need)))
    ws = DevBuf(need.value) if need.value else None
    check(lib().kf_reduce_moments(int(mode), C.byref(desc), float(correction), float(eps), ws.ptr if ws else None, need.value, stream))
    return ws  # keep alive until the stream is synchronised


def elementwise(op, desc: IterDesc, compute_dtype=0, scalar=0.0, stream=None):
    check(lib().kf_elementwise(int(op), C.byref(desc), int(compute_dtype), float(scalar), stream))


def reduce(op, desc: IterDesc, stream=None):
    need = C.c_size_t(0)
    check(lib().kf_reduce_workspace_bytes(C.byref(desc), C.byref(need)))
    ws = DevBuf(need.value) if need.value else None
    check(lib().kf_reduce(int(op), C.byref(desc), ws.ptr if ws else None, need.value, stream))
    return ws  # keep alive until the stream is synchronised


def norm_fwd(kind, dtype, rows, cols, x, weight, bias, eps, y, mean=None, rstd=None, ld=None, stream=None):
    check(lib().kf_norm_fwd(kind, dtype, rows, cols, cols if ld is None else ld, x, weight, bias, float(eps), y, mean, rstd, stream))


def norm_bwd(kind, dtype, rows, cols, x, weight, mean, rstd, dy, dx, dweight, dbias, ld=None, stream=None):
    ld = cols if ld is None else ld
    need = C.c_size_t(0)
    check(lib().kf_norm_bwd_workspace_bytes(kind, dtype, rows, cols, ld, C.byref(need)))
    ws = DevBuf(need.value) if need.value else None
    check(lib().kf_norm_bwd(kind, dtype, rows, cols, ld, x, weight, mean, rstd, dy, dx, dweight, dbias, ws.ptr if ws else None, need.value, stream))
    return ws  # keep alive until the stream is synchronised


def index_put(desc: IterDesc, sizes, strides_bytes, stream=None):
    n = len(sizes)
    a = (C.c_int64 * n)(*sizes)
    b = (C.c_int64 * n)(*strides_bytes)
    check(lib().kf_index_put(C.byref(desc), n, a, b, stream))


def index_get(table, nrows, row_bytes, idx, n, out, stream=None):
    check(lib().kf_index_get(table, nrows, row_bytes, idx, n, out, stream))


def index_add(dtype, idx, n, src, cols, nrows, dst, stream=None):
    need = lib().kf_index_add_workspace_bytes(n)
    ws = DevBuf(need) if need else None
    check(lib().kf_index_add(dtype, idx, n, src, cols, nrows, dst, ws.ptr if ws else None, need, stream))
    return ws  # keep alive until the stream is synchronised


def sort_segments(keys: np.ndarray, descending=False, code=None, stream=None):
    """keys [nseg, n] (host) -> (sorted keys, int64 positions) through kf_sort; code overrides the dtype (BF16 as uint16)."""
    keys = np.ascontiguousarray(keys)
    nseg, n = keys.shape
    code = NP2CODE[keys.dtype] if code is None else code
    if keys.size == 0:
        check(lib().kf_sort(code, None, None, None, nseg, n, int(descending), None, 0, stream))
        return keys.copy(), np.zeros(keys.shape, np.int64)
    src = DevBuf.from_numpy(keys)
    dst, pos = DevBuf(keys.nbytes), DevBuf(keys.size * 8)
    need = lib().kf_sort_workspace_bytes(code, nseg, n)
    ws = DevBuf(need) if need else None
    check(lib().kf_sort(code, src.ptr, dst.ptr, pos.ptr, nseg, n, int(descending), ws.ptr if ws else None, need, stream))
    device_sync()
    return dst.to_numpy(keys.shape, keys.dtype), pos.to_numpy(keys.shape, np.int64)


def gemm_workspace_bytes(dtype, trans_a, trans_b, M, N, K) -> int:
    need = C.c_size_t(0)
    check(lib().kf_gemm_workspace_bytes(dtype, int(trans_a), int(trans_b), M, N, K, C.byref(need)))
    return need.value


def gemm(dtype, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, Cptr, ldc, epilogue=EPI_NONE, bias=None,
         workspace=None, workspace_bytes=0, stream=None):
    check(lib().kf_gemm(dtype, int(trans_a), int(trans_b), M, N, K, alpha, A, lda, B, ldb, beta, Cptr, ldc,
                        epilogue, bias, workspace, workspace_bytes, stream))


def gemm_ex(dtype, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, Cptr, ldc, bias=None, mul=None, ldmul=0, add=None, ldadd=0,
            aux=None, ldaux=0, stream=None, c_f32=False):
    e = GemmEpilogue(bias, mul, ldmul, add, ldadd, aux, ldaux, int(c_f32))
    check(lib().kf_gemm_ex(dtype, int(trans_a), int(trans_b), M, N, K, alpha, A, lda, B, ldb, beta, Cptr, ldc, C.byref(e), stream))


def gemm_grouped(dtype, problems, stream=None):
    """problems: tuples (trans_a, trans_b, M, N, K, alpha, beta, A, lda, B, ldb, C, ldc[, c_f32])."""
    arr = (GemmProblem * len(problems))(*[GemmProblem(*p) for p in problems])
    check(lib().kf_gemm_grouped(dtype, len(problems), arr, stream))


def attn_fwd(dtype, B, H, Sq, Skv, D, q, k, v, o, lse=None, stream=None):
    check(lib().kf_attn_fwd(dtype, B, H, Sq, Skv, D, q, k, v, o, lse, stream))


def attn_bwd_workspace_bytes(dtype, B, H, Sq, Skv, D) -> int:
    need = C.c_size_t(0)
    check(lib().kf_attn_bwd_workspace_bytes(dtype, B, H, Sq, Skv, D, C.byref(need)))
    return need.value


def attn_bwd(dtype, B, H, Sq, Skv, D, q, k, v, o, lse, d_o, dq, dk, dv, workspace, workspace_bytes, stream=None):
    check(lib().kf_attn_bwd(dtype, B, H, Sq, Skv, D, q, k, v, o, lse, d_o, dq, dk, dv, workspace, workspace_bytes,
                            stream))


def attn_fwd_strided(dtype, B, H, Sq, Skv, D, scale, q, lq, k, lk, v, lv, o, lo, lse=None, stream=None):
    """lq.. are (batch, head, row) element-stride triples."""
    L = [AttnLayout(*t) for t in (lq, lk, lv, lo)]
    check(lib().kf_attn_fwd_strided(dtype, B, H, Sq, Skv, D, scale, q, C.byref(L[0]), k, C.byref(L[1]), v, C.byref(L[2]), o, C.byref(L[3]), lse, stream))


def attn_bwd_strided(dtype, B, H, Sq, Skv, D, scale, q, lq, k, lk, v, lv, o, lo, lse, d_o, ldo, dq, ldq, dk, ldk, dv, ldv, workspace,
                     workspace_bytes, stream=None):
    L = [AttnLayout(*t) for t in (lq, lk, lv, lo, ldo, ldq, ldk, ldv)]
    check(lib().kf_attn_bwd_strided(dtype, B, H, Sq, Skv, D, scale, q, C.byref(L[0]), k, C.byref(L[1]), v, C.byref(L[2]), o, C.byref(L[3]), lse,
                                    d_o, C.byref(L[4]), dq, C.byref(L[5]), dk, C.byref(L[6]), dv, C.byref(L[7]), workspace, workspace_bytes, stream))


def device_sync():
    check(lib().kf_device_sync())


def stream_wait_event(stream, event: "Event"):
    check(lib().kf_stream_wait_event(stream, event.handle))


def knobs_reload():
    """Re-read the KF_* A/B switches from the environment (they are cached per process)."""
    check(lib().kf_knobs_reload())


class knobs:
    """`with knobs(KF_ATTN_NO_XCD="1"): ...` — set A/B switches for the calls inside the block, restore them after."""

    def __init__(self, **env):
        self.env, self.old = env, {}

    def __enter__(self):
        import os
        for k, v in self.env.items():
            self.old[k] = os.environ.get(k)
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = str(v)
        knobs_reload()
        return self

    def __exit__(self, *exc):
        import os
        for k, v in self.old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        knobs_reload()
        return False


def profile_enable(on: bool):
    check(lib().kf_profile_enable(int(bool(on))))


def profile_reset():
    check(lib().kf_profile_reset())


def profile_samples() -> dict:
    """{kernel name: float32 array of its launches' durations in ms, in launch order} (synchronises)."""
    n = C.c_int(0)
    check(lib().kf_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(64)
        ms, cnt = C.c_double(0), C.c_int64(0)
        check(lib().kf_profile_get(i, name, C.byref(ms), C.byref(cnt)))
        buf = np.empty(min(cnt.value, 65536), dtype=np.float32)
        w = C.c_int(0)
        check(lib().kf_profile_samples(i, buf.ctypes.data_as(C.POINTER(C.c_float)), buf.size, C.byref(w)))
        out[name.value.decode()] = buf[:w.value]
    return out


def profile_results() -> dict:
    """{kernel name: (summed ms, launches)} for launches made while profiling was enabled (synchronises)."""
    n = C.c_int(0)
    check(lib().kf_profile_count(C.byref(n)))
    out = {}
    for i in range(n.value):
        name = C.create_string_buffer(64)
        ms, cnt = C.c_double(0), C.c_int64(0)
        check(lib().kf_profile_get(i, name, C.byref(ms), C.byref(cnt)))
        out[name.value.decode()] = (ms.value, cnt.value)
    return out
