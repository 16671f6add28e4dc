#!/usr/bin/env python3
"""Generic-path elementwise cases (mixed dtypes, strided operands) timed back to back: bytes moved / time."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
n = 1 << 27
a, b, c = H.DevBuf(8 * n), H.DevBuf(8 * n), H.DevBuf(8 * n)
def V(buf, shape, strides, code): return H.View(buf.ptr, shape, strides, code)
cases = []
cases.append(("add f32 + bf16 -> f32 (contiguous, mixed)", H.make_desc([V(c, (n,), (1,), H.F32)], [V(a, (n,), (1,), H.F32), V(b, (n,), (1,), H.BF16)]), H.EW_ADD, H.F32, n * 10))
cases.append(("add bf16 + f32 -> f32", H.make_desc([V(c, (n,), (1,), H.F32)], [V(a, (n,), (1,), H.BF16), V(b, (n,), (1,), H.F32)]), H.EW_ADD, H.F32, n * 10))
cases.append(("add i64 + i64", H.make_desc([V(c, (n,), (1,), H.I64)], [V(a, (n,), (1,), H.I64), V(b, (n,), (1,), H.I64)]), H.EW_ADD, H.I64, n * 24))
cases.append(("add i32 + i32", H.make_desc([V(c, (n,), (1,), H.I32)], [V(a, (n,), (1,), H.I32), V(b, (n,), (1,), H.I32)]), H.EW_ADD, H.I32, n * 12))
m = n // 2
cases.append(("add f32 x[::2] + y (strided input)", H.make_desc([V(c, (m,), (1,), H.F32)], [V(a, (m,), (2,), H.F32), V(b, (m,), (1,), H.F32)]), H.EW_ADD, H.F32, m * 12))
R, C = 8192, 8192
cases.append(("add f32 [8192, 8192] + column [8192, 1]", H.make_desc([V(c, (R, C), (C, 1), H.F32)], [V(a, (R, C), (C, 1), H.F32), V(b, (R, C), (1, 0), H.F32)]), H.EW_ADD, H.F32, R * C * 8))
cases.append(("copy f32 rows sliced [:, :4096] of [8192, 8192]", H.make_desc([V(c, (R, C // 2), (C // 2, 1), H.F32)], [V(a, (R, C // 2), (C, 1), H.F32)]), H.EW_COPY, None, R * C // 2 * 8))
cases.append(("copy f32 -> f64 (convert)", H.make_desc([V(c, (n,), (1,), H.F64)], [V(a, (n,), (1,), H.F32)]), H.EW_COPY, None, n * 12))
cases.append(("copy i32 -> f32 (convert)", H.make_desc([V(c, (n,), (1,), H.F32)], [V(a, (n,), (1,), H.I32)]), H.EW_COPY, None, n * 8))
for name, d, op, cd, nbytes in cases:
    fn = (lambda: H.elementwise(op, d, cd)) if cd is not None else (lambda: H.elementwise(op, d))
    try:
        for _ in range(2): fn()
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(10): fn()
        e1.record(None); H.device_sync()
        ms = e0.elapsed_ms(e1) / 10
        print(f"{name:52s} {ms:8.4f} ms {nbytes / ms / 1e6:8.1f} GB/s", flush=True)
    except Exception as ex:
        print(name, "ERR", str(ex)[:100])
