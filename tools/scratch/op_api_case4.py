#!/usr/bin/env python3
"""Embedding forward + backward, index_put_ with row indices, attention through the fused-qkv entry (ms per call)."""
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
    fn(); fn(); H.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    H.device_sync()
    print(f"{name:70s} {(time.perf_counter() - t0) / reps * 1e3:9.3f} ms", flush=True)
for V, D, n in ((50304, 4096, 32768), (128256, 4096, 32768), (50304, 1024, 262144)):
    tab = T((V, D)); tab.set_requires_grad(True)
    idx = kfunca.from_numpy(rng.integers(0, V, size=(n,)).astype(np.int64), 0)
    timeit(f"embedding fwd bf16 table [{V}, {D}] x {n} tokens", lambda: kfunca.embedding(tab, idx))
    g = T((n, D))
    def fb():
        y = kfunca.embedding(tab, idx)
        y.backward(g)
    timeit(f"embedding fwd + bwd", fb)
x = T((65536, 1024), False)
rows = kfunca.from_numpy(rng.permutation(65536)[:32768].astype(np.int64), 0)
vals = T((32768, 1024), False)
timeit("index_put_ f32 [65536, 1024] <- 32768 rows (1 index)", lambda: x.index_put_([rows], vals))
B, S, Hh, D = 4, 4096, 32, 128
qkv = T((B, S, 3 * Hh * D)); qkv.set_requires_grad(True)
timeit("causal_attention_qkv fwd B4 S4096 H32 D128", lambda: kfunca.causal_attention_qkv(qkv, B, S, Hh))
def qb():
    o = kfunca.causal_attention_qkv(qkv, B, S, Hh)
    o.backward(o)
timeit("causal_attention_qkv fwd + bwd", qb)
