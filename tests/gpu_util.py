"""Helpers for the -m gpu parity tests: everything goes through the C ABI (kfunca_amd.hip_abi)."""
import numpy as np

from kfunca_amd import hip_abi as H
from oracle import oracle as O


class Dev:
    """A numpy array uploaded to the GPU, remembered with its geometry and dtype code."""

    def __init__(self, arr: np.ndarray, code=None, base: np.ndarray = None):
        # `arr` may be a non-contiguous view of `base`; the whole base is uploaded
        self.base = np.ascontiguousarray(arr) if base is None else base
        self.arr = arr if base is not None else self.base
        self.code = H.NP2CODE[self.arr.dtype] if code is None else code
        self.buf = H.DevBuf.from_numpy(self.base)
        off = self.arr.__array_interface__["data"][0] - self.base.__array_interface__["data"][0]
        self.view = H.View.of(self.buf, self.arr, self.code, byte_offset=off)

    @classmethod
    def empty(cls, shape, code):
        return cls(np.zeros(shape, dtype=H.CODE2NP[code]), code)

    def get(self) -> np.ndarray:
        """Download the base buffer and return the view's current contents."""
        host = self.buf.to_numpy(self.base.shape, self.base.dtype)
        off = self.arr.__array_interface__["data"][0] - self.base.__array_interface__["data"][0]
        return np.lib.stride_tricks.as_strided(host.reshape(-1)[off // host.itemsize:], self.arr.shape, self.arr.strides).copy()


def gpu_binary(op, a: Dev, b: Dev, out_code=None, out: Dev = None):
    common = O.promote(a.code, b.code)
    if out is None:
        shape = tuple(max(x, y) for x, y in zip(a.arr.shape, b.arr.shape))
        out = Dev.empty(shape, common if out_code is None else out_code)
    d = H.make_desc([out.view], [a.view, b.view])
    H.elementwise(op, d, common)
    H.device_sync()
    return out


def gpu_copy(src: Dev, dst: Dev):
    d = H.make_desc([dst.view], [src.view])
    H.elementwise(H.EW_COPY, d)
    H.device_sync()
    return dst


def gpu_fill(dst: Dev, value):
    d = H.make_desc([dst.view], [])
    H.elementwise(H.EW_FILL, d, 0, value)
    H.device_sync()
    return dst


def gpu_reduce(op, x: Dev, dim):
    shape = list(x.arr.shape)
    shape[dim] = 1
    out = Dev.empty(shape, x.code)
    d = H.make_reduce_desc(out.view, x.view, dim)
    ws = H.reduce(op, d)
    H.device_sync()
    del ws
    return out


def rand_of(rng, shape, code):
    """Random array of dtype `code` (numpy array; bf16 as uint16 bits)."""
    if code == H.BOOL:
        return rng.integers(0, 2, size=shape).astype(np.bool_)
    if code in (H.U8,):
        return rng.integers(0, 255, size=shape).astype(np.uint8)
    if code in (H.I8, H.I16, H.I32, H.I64):
        return rng.integers(-100, 100, size=shape).astype(H.CODE2NP[code])
    x = rng.uniform(-10, 10, size=shape)
    if code == H.BF16:
        return O.f32_to_bf16(x.astype(np.float32))
    return x.astype(H.CODE2NP[code])


def gpu_moments(mode, x: Dev, dim, correction=1.0, eps=0.0, out_code=None):
    """(variance-like, mean) through kf_reduce_moments; outputs in x's dtype unless out_code is given."""
    shape = list(x.arr.shape)
    shape[dim] = 1
    oc = x.code if out_code is None else out_code
    o0, o1 = Dev.empty(shape, oc), Dev.empty(shape, oc)
    d = H.make_moments_desc(o0.view, o1.view, x.view, dim)
    keep = H.reduce_moments(mode, d, correction, eps)
    H.device_sync()
    del keep
    return o0, o1
