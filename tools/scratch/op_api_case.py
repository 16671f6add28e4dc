#!/usr/bin/env python3
"""Operator-API timings (import kfunca_amd as kfunca) at shapes off the tile grid: wall time per call with a device sync (ms)."""
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
def timeit(name, fn, reps=5):
    fn(); H.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    H.device_sync()
    print(f"{name:60s} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms", flush=True)
a, b = T((4000, 4000)), T((4000, 4000))
timeit("gemm bf16 4000^3", lambda: kfunca.gemm(a, b, 1.0, 0.0))
a, b = T((4096, 4096)), T((4096, 4096))
timeit("gemm bf16 4096^3", lambda: kfunca.gemm(a, b, 1.0, 0.0))
a, b = T((16, 8192)), T((8192, 8192))
timeit("gemm bf16 16 x 8192 x 8192", lambda: kfunca.gemm(a, b, 1.0, 0.0))
a, b = T((4096, 4096), False), T((4096, 50257 // 1 if False else 4097), False)
timeit("gemm f32 4096 x 4097 x 4096", lambda: kfunca.gemm(a, b, 1.0, 0.0))
for S in (4096, 4000, 1000):
    q, k, v = T((8, 32, S, 128)), T((8, 32, S, 128)), T((8, 32, S, 128))
    timeit(f"causal_attention bf16 B8 H32 S{S} D128", lambda: kfunca.causal_attention(q, k, v))
q, k, v = T((8, 32, 2048, 96)), T((8, 32, 2048, 96)), T((8, 32, 2048, 96))
timeit("causal_attention bf16 B8 H32 S2048 D96", lambda: kfunca.causal_attention(q, k, v))
x = T((8192, 8192), False)
timeit("sum(0) f32 8192^2", lambda: x.sum(0))
timeit("sum(1) f32 8192^2", lambda: x.sum(1))
timeit("permute(1,0).contiguous f32 8192^2", lambda: x.permute(1, 0).contiguous())
timeit("sort(1) f32 8192^2", lambda: x.sort(1, False))
timeit("sort(0) f32 8192^2", lambda: x.sort(0, False))
timeit("topk(10, 1) f32 8192^2", lambda: x.topk(10, 1, True))
y = T((8192, 8192), False)
timeit("x + y f32 8192^2", lambda: x + y)
timeit("x + 2.0 f32 8192^2", lambda: x + 2.0)
timeit("cat([x, y], 0)", lambda: kfunca.cat([x, y], 0))
timeit("cat([x, y], 1)", lambda: kfunca.cat([x, y], 1))
timeit("x.half()", lambda: x.half())
timeit("x.split([4096, 4096], 1)[1].contiguous()", lambda: x.split([4096, 4096], 1)[1].contiguous())
timeit("x[:, 1000:5000] contiguous (getitem)", lambda: x[:, 1000:5000].contiguous())
timeit("mean_var(1)", lambda: x.mean_var(1, False))
