#!/usr/bin/env python3
"""More operator-API timings: gathers with short rows, strided elementwise, reductions over several dims (ms per call, GB/s of the obvious bytes)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H
rng = np.random.default_rng(0)
def T(shape, bf=True):
    t = kfunca.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32), 0)
    return t.bfloat16() if bf else t
def timeit(name, fn, nbytes=0, reps=5):
    fn(); H.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    H.device_sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:64s} {ms:9.3f} ms" + (f" {nbytes / ms / 1e6:8.0f} GB/s" if nbytes else ""), flush=True)
for D in (32, 64, 128, 256, 1024):
    tab = T((1 << 20, D))
    nidx = (1 << 28) // D
    idx = kfunca.from_numpy(rng.integers(0, 1 << 20, size=(nidx,)).astype(np.int64), 0)
    timeit(f"embedding bf16 table [1Mi, {D}], {nidx} rows", lambda: kfunca.embedding(tab, idx), 2 * nidx * D * 2)
x = T((4096, 8192), False)
y = T((8192, 4096), False)
timeit("x + y.permute(1,0) f32 [4096, 8192]", lambda: x + y.permute(1, 0), 3 * 4 * 4096 * 8192)
z = T((64, 128, 4096), False)
timeit("z.permute(1,0,2).contiguous() f32 [64,128,4096]", lambda: z.permute(1, 0, 2).contiguous(), 2 * 4 * 64 * 128 * 4096)
timeit("z.permute(2,0,1).contiguous()", lambda: z.permute(2, 0, 1).contiguous(), 2 * 4 * 64 * 128 * 4096)
timeit("z.permute(0,2,1).contiguous()", lambda: z.permute(0, 2, 1).contiguous(), 2 * 4 * 64 * 128 * 4096)
q = T((8, 4096, 32, 128))
timeit("q[B,S,H,D].permute(0,2,1,3).contiguous() bf16", lambda: q.permute(0, 2, 1, 3).contiguous(), 2 * 2 * 8 * 4096 * 32 * 128)
w = T((8, 4096, 3 * 4096))
timeit("qkv.split([4096]*3, 2)[1].contiguous() bf16", lambda: w.split([4096, 4096, 4096], 2)[1].contiguous(), 2 * 2 * 8 * 4096 * 4096)
x16 = T((8192, 8192))
timeit("x16.float()", lambda: x16.float(), 6 * 8192 * 8192)
timeit("x16 * x16 (bf16)", lambda: x16 * x16, 6 * 8192 * 8192)
timeit("x16 + x (bf16 + f32)", lambda: x16[:4096] + x, 10 * 4096 * 8192)
xi = kfunca.from_numpy(rng.integers(-100, 100, size=(8192, 8192)).astype(np.int32), 0)
timeit("int32 sum(1)", lambda: xi.sum(1), 4 * 8192 * 8192)
timeit("int32 + int32", lambda: xi + xi, 12 * 8192 * 8192)
