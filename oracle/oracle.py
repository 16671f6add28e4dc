"""ctypes front-end of the CPU oracle (oracle/oracle.c).  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Arrays are numpy; bf16 travels as uint16 with an explicit dtype code (numpy has no bfloat16).
"""
from __future__ import annotations

import ctypes as C
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
LIB_PATH = HERE / "liboracle.so"

BOOL, U8, I8, I16, I32, I64, F16, BF16, F32, F64 = range(10)
ADD, SUB, MUL, DIV = range(4)
SUM, MEAN = range(2)
MAX_DIMS = 12

_NP2CODE = {np.dtype(np.bool_): BOOL, np.dtype(np.uint8): U8, np.dtype(np.int8): I8, np.dtype(np.int16): I16,
            np.dtype(np.int32): I32, np.dtype(np.int64): I64, np.dtype(np.float16): F16,
            np.dtype(np.float32): F32, np.dtype(np.float64): F64}
CODE2NP = {BOOL: np.bool_, U8: np.uint8, I8: np.int8, I16: np.int16, I32: np.int32, I64: np.int64,
           F16: np.float16, BF16: np.uint16, F32: np.float32, F64: np.float64}


class _Tensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("shape", C.c_int64 * MAX_DIMS), ("stride", C.c_int64 * MAX_DIMS)]


def code_of(arr: np.ndarray, code=None) -> int:
    if code is not None:
        return int(code)
    return _NP2CODE[arr.dtype]


# a fixed ISA baseline, not -march=native: the .so is built in one container and loaded on another host (the GPU box);
# -ffp-contract=off keeps a*b+c un-fused unless the source says fmaf, so the bit-exact comparisons do not depend on the host
CFLAGS = ["-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=c11"]


def _build():
    import subprocess
    subprocess.run(["gcc", *CFLAGS, "-o", str(LIB_PATH), str(HERE / "oracle.c"), "-lm"], check=True)


_lib = None


def lib():
    global _lib
    if _lib is None:
        src = HERE / "oracle.c"
        if not LIB_PATH.exists() or LIB_PATH.stat().st_mtime < src.stat().st_mtime:
            _build()
        _lib = C.CDLL(str(LIB_PATH))
        _lib.orc_bf16_to_f32.restype = C.c_float
        _lib.orc_bf16_to_f32.argtypes = [C.c_uint16]
        _lib.orc_f32_to_bf16.restype = C.c_uint16
        _lib.orc_f32_to_bf16.argtypes = [C.c_float]
        _lib.orc_f16_to_f32.restype = C.c_float
        _lib.orc_f16_to_f32.argtypes = [C.c_uint16]
        _lib.orc_f32_to_f16.restype = C.c_uint16
        _lib.orc_f32_to_f16.argtypes = [C.c_float]
    return _lib


def _t(arr: np.ndarray, code=None) -> _Tensor:
    t = _Tensor()
    t.data = arr.ctypes.data
    t.dtype = code_of(arr, code)
    t.ndim = arr.ndim
    for i in range(arr.ndim):
        t.shape[i] = arr.shape[i]
        t.stride[i] = arr.strides[i] // arr.itemsize
    return t


def _check(rc, what):
    if rc != 0:
        raise RuntimeError(f"oracle {what} failed with code {rc}")


# ---- bf16 helpers (vectorised restatement of half.h:195-208) ---------------------------------------
def f32_to_bf16(x: np.ndarray) -> np.ndarray:
    x = np.ascontiguousarray(x, dtype=np.float32)
    u = x.view(np.uint32)
    r = ((u.astype(np.uint64) + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    return np.where(np.isnan(x), np.uint16(0x7FC0), r).astype(np.uint16)


def bf16_to_f32(x: np.ndarray) -> np.ndarray:
    return (np.ascontiguousarray(x, dtype=np.uint16).astype(np.uint32) << 16).view(np.float32)


def to_float(arr: np.ndarray, code=None) -> np.ndarray:
    c = code_of(arr, code)
    return bf16_to_f32(arr) if c == BF16 else arr.astype(np.float64 if c == F64 else np.float32)


def from_float(x: np.ndarray, code: int) -> np.ndarray:
    if code == BF16:
        return f32_to_bf16(x)
    return np.ascontiguousarray(x).astype(CODE2NP[code])


def promote(a: int, b: int) -> int:
    return lib().orc_promote(int(a), int(b))


def binary(op, a, b, out_code=None, a_code=None, b_code=None):
    ca, cb = code_of(a, a_code), code_of(b, b_code)
    co = promote(ca, cb) if out_code is None else out_code
    shape = tuple(max(x, y) for x, y in zip(a.shape, b.shape))
    out = np.empty(shape, dtype=CODE2NP[co])
    ta, tb, to = _t(a, ca), _t(b, cb), _t(out, co)
    _check(lib().orc_binary(int(op), C.byref(ta), C.byref(tb), C.byref(to)), "binary")
    return out


def binary_out(op, a, b, out, a_code=None, b_code=None, out_code=None):
    ta, tb, to = _t(a, a_code), _t(b, b_code), _t(out, out_code)
    _check(lib().orc_binary(int(op), C.byref(ta), C.byref(tb), C.byref(to)), "binary")
    return out


def copy(src, dst, src_code=None, dst_code=None):
    ts, td = _t(src, src_code), _t(dst, dst_code)
    _check(lib().orc_copy(C.byref(ts), C.byref(td)), "copy")
    return dst


def convert(src, dst_code, src_code=None):
    dst = np.empty(src.shape, dtype=CODE2NP[dst_code])
    return copy(src, dst, src_code, dst_code)


def fill(dst, value, dst_code=None):
    td = _t(dst, dst_code)
    lib().orc_fill.argtypes = [C.c_void_p, C.c_double]
    _check(lib().orc_fill(C.byref(td), float(value)), "fill")
    return dst


def reduce(op, x, dim, code=None):
    c = code_of(x, code)
    shape = list(x.shape)
    shape[dim] = 1
    out = np.empty(shape, dtype=CODE2NP[c])
    ti, to = _t(x, c), _t(out, c)
    _check(lib().orc_reduce(int(op), C.byref(ti), int(dim), C.byref(to)), "reduce")
    return out


def moments(mode, x, dim, correction=1.0, eps=0.0, code=None, out_code=None):
    """(variance-like, mean) of x along dim, keepdim; mode 0 var, 1 std, 2 invstd (orc_moments)."""
    c = code_of(x, code)
    oc = c if out_code is None else out_code
    shape = list(x.shape)
    shape[dim] = 1
    o0, o1 = np.empty(shape, dtype=CODE2NP[oc]), np.empty(shape, dtype=CODE2NP[oc])
    ti, t0, t1 = _t(x, c), _t(o0, oc), _t(o1, oc)
    f = lib().orc_moments
    f.argtypes = [C.c_int, C.c_void_p, C.c_int, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    _check(f(int(mode), C.byref(ti), int(dim), float(correction), float(eps), C.byref(t0), C.byref(t1)), "moments")
    return o0, o1


def index_put(self_arr, indices, values, code=None):
    c = code_of(self_arr, code)
    idx = (_Tensor * len(indices))(*[_t(np.ascontiguousarray(i, dtype=np.int64)) for i in indices])
    keep = [np.ascontiguousarray(i, dtype=np.int64) for i in indices]
    idx = (_Tensor * len(keep))(*[_t(i) for i in keep])
    ts, tv = _t(self_arr, c), _t(values, c)
    _check(lib().orc_index_put(C.byref(ts), len(keep), idx, C.byref(tv)), "index_put")
    return self_arr


def gemm(a, b, alpha=1.0, beta=0.0, trans_a=False, trans_b=False, c=None, bias=None, code=None):
    cd = code_of(a, code)
    M = a.shape[1] if trans_a else a.shape[0]
    K = a.shape[0] if trans_a else a.shape[1]
    N = b.shape[0] if trans_b else b.shape[1]
    a = np.ascontiguousarray(a)
    b = np.ascontiguousarray(b)
    out = np.zeros((M, N), dtype=CODE2NP[cd]) if c is None else np.ascontiguousarray(c).copy()
    f = lib().orc_gemm
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_int64,
                  C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    _check(f(cd, int(trans_a), int(trans_b), M, N, K, alpha, a.ctypes.data, a.shape[1], b.ctypes.data, b.shape[1],
             beta, out.ctypes.data, N, None if bias is None else np.ascontiguousarray(bias).ctypes.data), "gemm")
    return out


def attn_fwd(q, k, v, code=None, want_lse=True):
    cd = code_of(q, code)
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    q, k, v = map(np.ascontiguousarray, (q, k, v))
    o = np.empty_like(q)
    lse = np.empty((B, H, Sq), dtype=np.float32)
    f = lib().orc_attn_fwd
    f.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.c_void_p] * 5
    _check(f(cd, B, H, Sq, Skv, D, q.ctypes.data, k.ctypes.data, v.ctypes.data, o.ctypes.data,
             lse.ctypes.data if want_lse else None), "attn_fwd")
    return (o, lse) if want_lse else o


def attn_bwd(q, k, v, d_o, code=None):
    cd = code_of(q, code)
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    q, k, v, d_o = map(np.ascontiguousarray, (q, k, v, d_o))
    dq, dk, dv = np.empty_like(q), np.empty_like(k), np.empty_like(v)
    f = lib().orc_attn_bwd
    f.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.c_void_p] * 7
    _check(f(cd, B, H, Sq, Skv, D, q.ctypes.data, k.ctypes.data, v.ctypes.data, d_o.ctypes.data, dq.ctypes.data,
             dk.ctypes.data, dv.ctypes.data), "attn_bwd")
    return dq, dk, dv


def attn_ref64(q, k, v, d_o=None, code=None):
    """The double-precision evaluation (nothing rounded) on the dtype-rounded inputs + every output element's error scales.
    Returns a dict of float64 arrays: o, lse, mo, qo and, with d_o, dq, dk, dv, mdq, mdk, mdv, qdq, qdk, qdv, bdq
    (m* = sum of |terms|: worst case; q* = terms in quadrature: statistical; bdq = worst case of a rounded delta through dq;
    oracle.c: orc_attn_ref64)."""
    cd = code_of(q, code)
    B, H, Sq, D = q.shape
    Skv = k.shape[2]
    q, k, v = map(np.ascontiguousarray, (q, k, v))
    r = {"o": np.empty(q.shape, np.float64), "lse": np.empty((B, H, Sq), np.float64), "mo": np.empty(q.shape, np.float64), "qo": np.empty(q.shape, np.float64)}
    if d_o is not None:
        d_o = np.ascontiguousarray(d_o)
        for n, like in (("dq", q), ("dk", k), ("dv", v), ("mdq", q), ("mdk", k), ("mdv", v), ("bdq", q), ("qdq", q), ("qdk", k), ("qdv", v)):
            r[n] = np.empty(like.shape, np.float64)
    f = lib().orc_attn_ref64
    f.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.c_void_p] * 18
    ptr = lambda n: r[n].ctypes.data if n in r else None  # noqa: E731
    _check(f(cd, B, H, Sq, Skv, D, q.ctypes.data, k.ctypes.data, v.ctypes.data, None if d_o is None else d_o.ctypes.data,
             ptr("o"), ptr("lse"), ptr("mo"), ptr("dq"), ptr("dk"), ptr("dv"), ptr("mdq"), ptr("mdk"), ptr("mdv"), ptr("bdq"),
             ptr("qo"), ptr("qdq"), ptr("qdk"), ptr("qdv")), "attn_ref64")
    return r


RMS, LAYER = 0, 1


def norm_fwd(kind, x, w=None, b=None, eps=1e-5, code=None):
    """rows = all leading dims flattened; returns (y, mean[rows] f32, rstd[rows] f32)."""
    cd = code_of(x, code)
    x = np.ascontiguousarray(x)
    cols = x.shape[-1]
    rows = x.size // cols
    y = np.empty_like(x)
    mean, rstd = np.zeros(rows, np.float32), np.zeros(rows, np.float32)
    f = lib().orc_norm_fwd
    f.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p, C.c_void_p]
    p = lambda a: None if a is None else np.ascontiguousarray(a).ctypes.data  # noqa: E731
    keep = [None if a is None else np.ascontiguousarray(a) for a in (w, b)]
    _check(f(kind, cd, rows, cols, x.ctypes.data, p(keep[0]), p(keep[1]), float(eps), y.ctypes.data, mean.ctypes.data, rstd.ctypes.data), "norm_fwd")
    return y, mean, rstd


def norm_bwd(kind, x, w, dy, eps=1e-5, code=None, want_db=True):
    """(dx, dw, db) of norm_fwd; db is None for rms."""
    cd = code_of(x, code)
    x, dy = np.ascontiguousarray(x), np.ascontiguousarray(dy)
    cols = x.shape[-1]
    rows = x.size // cols
    dx = np.empty_like(x)
    dw = np.empty(cols, dtype=x.dtype)
    db = np.empty(cols, dtype=x.dtype) if (kind == LAYER and want_db) else None
    wk = None if w is None else np.ascontiguousarray(w)
    f = lib().orc_norm_bwd
    f.argtypes = [C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_void_p, C.c_void_p, C.c_double] + [C.c_void_p] * 4
    _check(f(kind, cd, rows, cols, x.ctypes.data, None if wk is None else wk.ctypes.data, float(eps), dy.ctypes.data, dx.ctypes.data,
             dw.ctypes.data, None if db is None else db.ctypes.data), "norm_bwd")
    return dx, dw, db


def index_get(table, idx):
    """out[n, :] = table[wrap(idx[n]), :] (embedding gather), any dtype: rows are copied as bytes."""
    table = np.ascontiguousarray(table)
    idx = np.ascontiguousarray(idx, dtype=np.int64)
    nrows = table.shape[0]
    row_bytes = table.nbytes // max(nrows, 1)
    out = np.empty(idx.shape + table.shape[1:], dtype=table.dtype)
    f = lib().orc_index_get
    f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_void_p]
    _check(f(table.ctypes.data, nrows, row_bytes, idx.ctypes.data, idx.size, out.ctypes.data), "index_get")
    return out


def num_threads() -> int:
    return lib().orc_num_threads()


# ---- sort (numpy restatement: byte / integer work) ------------------------------------------------------
_KEY_KIND = {U8: "u", I8: "i", I16: "i", I32: "i", I64: "i", F16: "f", BF16: "f", F32: "f", F64: "f"}


def sort_key(x: np.ndarray, code=None) -> np.ndarray:
    """The reference's order-preserving unsigned key (KeyTraits<T>::convert, src/device/utils/sorting_common.h:23-260):
    unsigned: identity; signed: + 2^(bits-1); floating (any width): x ^ (sign ? all ones : sign bit)."""
    c = code_of(x, code)
    if c not in _KEY_KIND:
        raise ValueError("Sort currently does not support bool dtypes.")  # sort_ops_kernel.cu:569-570
    w = x.dtype.itemsize
    ut = {1: np.uint8, 2: np.uint16, 4: np.uint32, 8: np.uint64}[w]
    u = np.ascontiguousarray(x).view(ut)
    sign = ut(1 << (8 * w - 1))
    if _KEY_KIND[c] == "u":
        return u
    if _KEY_KIND[c] == "i":
        return u ^ sign
    return u ^ np.where(u & sign, ut(~ut(0)), sign).astype(ut)


def sort_stable(x: np.ndarray, dim: int, descending: bool, code=None):
    """sort_stable_kernel (src/device/sort_ops_kernel.cu:556-618): least-significant-digit radix sort of the converted keys
    with int64 positions as values (sorting_radix_sort.h); descending walks the digit bins in reverse, i.e. an ascending
    stable sort of the complemented key. Returns (values, indices) shaped like x."""
    key = sort_key(x, code)
    if descending:
        key = ~key
    idx = np.argsort(key, axis=dim, kind="stable").astype(np.int64)
    return np.take_along_axis(x, idx, axis=dim), idx


def topk(x: np.ndarray, k: int, dim: int, largest: bool, code=None):
    """topk_with_sort (sort_ops_kernel.cu:620-632): the first k entries of the stable sort along dim."""
    v, i = sort_stable(x, dim, largest, code)
    sl = [slice(None)] * x.ndim
    sl[dim] = slice(0, k)
    return v[tuple(sl)].copy(), i[tuple(sl)].copy()
