#!/usr/bin/env python3
"""Reduction cases off the f32 matrix path (other dtypes, middle dims, mean): bytes read / time."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
a, c = H.DevBuf(1 << 30), H.DevBuf(1 << 28)
def contig(shape):
    st, run = [], 1
    for s in reversed(shape):
        st.append(run); run *= s
    return list(reversed(st))
def V(buf, shape, code): return H.View(buf.ptr, shape, contig(shape), code)
R = 16384
cases = [("sum(1) f32 [32Mi, 8] (R = 8)", (1 << 25, 8), 1, H.F32, H.RED_SUM), ("sum(1) f32 [4Mi, 64]", (1 << 22, 64), 1, H.F32, H.RED_SUM), ("sum(1) bf16 [8Mi, 32]", (1 << 23, 32), 1, H.BF16, H.RED_SUM),
         ("sum(0) f32 [16, 16Mi] (R = 16)", (16, 1 << 24), 0, H.F32, H.RED_SUM), ("sum(0) f32 [4, 4096, 4096] (R = 4)", (4, 4096, 4096), 0, H.F32, H.RED_SUM), ("sum(1) f32 [4096, 3, 4096] (middle, R = 3)", (4096, 3, 4096), 1, H.F32, H.RED_SUM),
         ("mean(0) bf16 [2, 128Mi]", (2, 1 << 27), 0, H.BF16, H.RED_MEAN),
         ("sum(1) bf16 [16384,16384]", (R, R), 1, H.BF16, H.RED_SUM), ("sum(0) bf16 [16384,16384]", (R, R), 0, H.BF16, H.RED_SUM),
         ("sum(1) f16 [16384,16384]", (R, R), 1, H.F16, H.RED_SUM),
         ("sum(1) i32 [16384,16384]", (R, R), 1, H.I32, H.RED_SUM), ("sum(0) i64 [8192,16384]", (8192, R), 0, H.I64, H.RED_SUM),
         ("sum(1) f64 [8192,16384]", (8192, R), 1, H.F64, H.RED_SUM), ("mean(1) f32 [16384,16384]", (R, R), 1, H.F32, H.RED_MEAN),
         ("sum(1) f32 [256,4096,256] (middle dim)", (256, 4096, 256), 1, H.F32, H.RED_SUM), ("sum(2) f32 [256,4096,256] (last dim, short rows)", (256, 4096, 256), 2, H.F32, H.RED_SUM),
         ("sum(0) f32 [256,4096,256] (first dim)", (256, 4096, 256), 0, H.F32, H.RED_SUM), ("sum(1) bf16 [512,2048,256] (middle dim)", (512, 2048, 256), 1, H.BF16, H.RED_SUM)]
for name, shape, dim, code, op in cases:
    so = list(shape); so[dim] = 1
    d = H.make_reduce_desc(V(c, tuple(so), code), V(a, shape, code), dim)
    keep = []
    fn = lambda: keep.append(H.reduce(op, d))
    try:
        for _ in range(2): fn()
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(10): fn()
        e1.record(None); H.device_sync()
        ms = e0.elapsed_ms(e1) / 10
        nb = int(np.prod(shape)) * H.DTYPE_SIZE[code]
        print(f"{name:52s} {ms:8.4f} ms {nb / ms / 1e6:8.1f} GB/s", flush=True)
    except Exception as ex:
        print(name, "ERR", str(ex)[:120])
