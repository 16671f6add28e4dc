#!/usr/bin/env python3
"""Operator-API latency at config C1 (1024 x 1024 f32): time per call over a long back-to-back loop, with and without a sync per call."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H
rng = np.random.default_rng(0)
a = kfunca.from_numpy(rng.uniform(-1, 1, (1024, 1024)).astype(np.float32), 0)
b = kfunca.from_numpy(rng.uniform(-1, 1, (1024, 1024)).astype(np.float32), 0)
def loop(name, fn, n=2000):
    for _ in range(50): fn()
    H.device_sync()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    t1 = time.perf_counter()
    H.device_sync()
    t2 = time.perf_counter()
    print(f"{name:40s} host {1e6 * (t1 - t0) / n:7.2f} us/call   with drain {1e6 * (t2 - t0) / n:7.2f} us/call", flush=True)
loop("a + b", lambda: a + b)
loop("a += b", lambda: a.__iadd__(b))
loop("a.sum(0)", lambda: a.sum(0))
loop("a.sum(1)", lambda: a.sum(1))
loop("a + 2.0", lambda: a + 2.0)
loop("a.permute(1,0)", lambda: a.permute(1, 0))
loop("a.permute(1,0).contiguous()", lambda: a.permute(1, 0).contiguous())
loop("gemm(a, b)", lambda: kfunca.gemm(a, b, 1.0, 0.0))
loop("a.mean(1)", lambda: a.mean(1))
s = kfunca.from_numpy(rng.uniform(-1, 1, (16, 16)).astype(np.float32), 0)
loop("tiny s + s (16x16)", lambda: s + s)
