#!/usr/bin/env python3
"""Randomised stress of the round's new C-ABI paths against numpy / the oracle: sort (every path), ragged kf_gemm, mixed-dtype elementwise on
contiguous and sliced operands, short-row reductions and norms. stress_abi.py SEED SECONDS"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.gpu_util import Dev, gpu_binary, gpu_copy, gpu_reduce
H.set_device(0)
seed, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed)
t_end = time.time() + secs
n_sort = n_gemm = n_ew = n_red = 0
NP = {H.F32: np.float32, H.F64: np.float64, H.I32: np.int32, H.I64: np.int64, H.I16: np.int16, H.U8: np.uint8, H.I8: np.int8, H.F16: np.float16}
def bits(a): return a.view(np.uint8)
while time.time() < t_end:
    kind = rng.integers(0, 4)
    if kind == 0:   # sort
        code = int(rng.choice(list(NP)))
        n = int(rng.choice([1, 2, 7, 33, 64, 65, 127, 128, 129, 300, 512, 513, 1000, 4097, 8192, 8193, 20000, 70000]))
        nseg = int(rng.integers(1, max(2, min(3000, 400000 // n))))
        desc = bool(rng.integers(0, 2))
        keys = (rng.integers(-5, 6, size=(nseg, n)) if rng.integers(0, 3) == 0 else rng.uniform(-100, 100, size=(nseg, n)))
        keys = np.abs(keys).astype(NP[code]) if code == H.U8 else keys.astype(NP[code])
        k, p = H.sort_segments(keys, desc)
        wk, wp = O.sort_stable(keys, 1, desc)
        assert np.array_equal(p, wp) and np.array_equal(bits(k), bits(wk)), ("sort", code, nseg, n, desc)
        n_sort += 1
    elif kind == 1:  # ragged gemm through the C ABI
        code = int(rng.choice([H.BF16, H.F16, H.F32]))
        M, N, K = (int(rng.integers(1, 600)) for _ in range(3))
        if rng.integers(0, 2): M, N, K = M + 300, N + 300, K + 300
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
        b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
        sa, sb = (np.ascontiguousarray(a.T) if ta else a), (np.ascontiguousarray(b.T) if tb else b)
        da, db, dc = H.DevBuf.from_numpy(sa), H.DevBuf.from_numpy(sb), H.DevBuf(M * N * a.itemsize)
        need = H.gemm_workspace_bytes(code, ta, tb, M, N, K)
        ws = H.DevBuf(max(need, 16))
        H.gemm(code, ta, tb, M, N, K, 1.0, da.ptr, sa.shape[1], db.ptr, sb.shape[1], 0.0, dc.ptr, N, 0, None, ws.ptr if need else None, need)
        H.device_sync()
        got = O.to_float(dc.to_numpy((M, N), a.dtype), code).astype(np.float64)
        fa, fb = O.to_float(a, code).astype(np.float64), O.to_float(b, code).astype(np.float64)
        eps = {H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11, H.F32: 2.0 ** -20}[code]
        floor = {H.BF16: 2.0 ** -133, H.F16: 2.0 ** -24, H.F32: 2.0 ** -149}[code]   # the output format's subnormal spacing: a product of two small operands (K = 1) rounds on THAT grid
        err = np.abs(got - fa @ fb)
        bound = 2 * eps * np.abs(fa @ fb) + 2 * eps * (np.abs(fa) @ np.abs(fb)) + floor
        assert (err <= bound).all(), ("gemm", code, M, N, K, ta, tb, float((err / bound).max()), float(err.max()))
        n_gemm += 1
    elif kind == 2:  # mixed-dtype / sliced elementwise
        ca, cb = int(rng.choice([H.F32, H.BF16, H.F16, H.I32, H.F64])), int(rng.choice([H.F32, H.BF16, H.F16, H.I32, H.U8]))
        rows, cols = int(rng.integers(1, 40)), int(rng.choice([8, 64, 200, 512, 1000]))
        def mk(code):
            x = rng.uniform(1, 50, (rows, cols + 8))
            return O.from_float(x.astype(np.float32), code) if code in (H.BF16, H.F16) else x.astype(H.CODE2NP[code])
        a, b = mk(ca), mk(cb)
        oa, ob = int(rng.integers(0, 8)), int(rng.integers(0, 8))
        va, vb = a[:, oa:oa + cols], b[:, ob:ob + cols]
        got = gpu_binary(H.EW_ADD, Dev(va, ca, base=a), Dev(vb, cb, base=b)).get()
        want = O.binary(O.ADD, np.ascontiguousarray(va), np.ascontiguousarray(vb), a_code=ca, b_code=cb)
        assert np.array_equal(bits(got), bits(want)), ("ew", ca, cb, rows, cols, oa, ob)
        n_ew += 1
    else:  # reductions over short extents (integer sums: exact)
        shape = (int(rng.integers(1, 9)), int(rng.choice([64, 1000, 4096, 70000]))) if rng.integers(0, 2) else (int(rng.choice([70000, 300000])), int(rng.choice([4, 8, 16, 64])))
        dim = 0 if shape[0] <= 8 else 1
        x = rng.integers(-1000, 1000, size=shape).astype(np.int32)
        got = gpu_reduce(H.RED_SUM, Dev(x, H.I32), dim).get()
        assert np.array_equal(got.reshape(-1), x.astype(np.int64).sum(axis=dim).astype(np.int32).reshape(-1)), ("reduce", shape, dim)
        n_red += 1
print(f"seed {seed}: {n_sort} sorts, {n_gemm} gemms, {n_ew} elementwise, {n_red} reductions - all agree")
