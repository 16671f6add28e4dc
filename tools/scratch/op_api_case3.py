#!/usr/bin/env python3
"""Fused GEMM / norms / embedding backward through the operator API at ragged shapes (ms per call)."""
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
    print(f"{name:64s} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms", flush=True)
for n in (4096, 4000):
    a, b, bias, m = T((n, n)), T((n, n)), T((n,)), T((n, n))
    timeit(f"gemm_fused bf16 {n}^3 + bias", lambda: kfunca.gemm_fused(a, b, 0.5, bias, None, None))
    timeit(f"gemm_fused bf16 {n}^3 + bias * mul + add", lambda: kfunca.gemm_fused(a, b, 0.5, bias, m, m))
for cols in (4096, 4000, 5000):
    x, w = T((16384, cols)), T((cols,))
    timeit(f"rms_norm bf16 [16384, {cols}]", lambda: kfunca.rms_norm(x, w, 1e-5))
    timeit(f"layer_norm bf16 [16384, {cols}]", lambda: kfunca.layer_norm(x, w, w, 1e-5))
a = T((4000, 4000)); a.set_requires_grad(True)
b = T((4000, 4000)); b.set_requires_grad(True)
def fb():
    c = kfunca.gemm(a, b, 1.0, 0.0)
    c.backward(c)
timeit("gemm bf16 4000^3 forward + backward (autograd)", fb)
q = T((2, 32, 4000, 128)); q.set_requires_grad(True)
k = T((2, 32, 4000, 128)); k.set_requires_grad(True)
v = T((2, 32, 4000, 128)); v.set_requires_grad(True)
def ab():
    o = kfunca.causal_attention(q, k, v)
    o.backward(o)
timeit("causal_attention bf16 B2 H32 S4000 D128 fwd + bwd", ab)
q = T((2, 32, 4096, 128)); q.set_requires_grad(True)
k = T((2, 32, 4096, 128)); k.set_requires_grad(True)
v = T((2, 32, 4096, 128)); v.set_requires_grad(True)
timeit("causal_attention bf16 B2 H32 S4096 D128 fwd + bwd", ab)
