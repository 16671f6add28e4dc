#!/usr/bin/env python3
"""Sliced (unaligned) views in elementwise / copies / reductions (ms per call, GB/s of the obvious bytes)."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H
rng = np.random.default_rng(0)
def T(shape, bf=False):
    t = kfunca.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32), 0)
    return t.bfloat16() if bf else t
def timeit(name, fn, nbytes=0, reps=5):
    fn(); fn(); H.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    H.device_sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:64s} {ms:9.3f} ms" + (f" {nbytes / ms / 1e6:8.0f} GB/s" if nbytes else ""), flush=True)
x, y = T((8192, 8200)), T((8192, 8200))
n = 8192 * 4096
timeit("x[:, 0:4096] + y[:, 0:4096] f32 (aligned slices)", lambda: x[:, 0:4096] + y[:, 0:4096], 12 * n)
timeit("x[:, 1:4097] + y[:, 3:4099] f32 (unaligned slices)", lambda: x[:, 1:4097] + y[:, 3:4099], 12 * n)
timeit("x[:, 0:4096] + y[:, 0:4096] f32 (aligned slices) AGAIN", lambda: x[:, 0:4096] + y[:, 0:4096], 12 * n)
timeit("x[:, 1:4097].contiguous()", lambda: x[:, 1:4097].contiguous(), 8 * n)
timeit("x[:, 1:4097].sum(1)", lambda: x[:, 1:4097].sum(1), 4 * n)
timeit("x[:, 1:4097].sum(0)", lambda: x[:, 1:4097].sum(0), 4 * n)
xb, yb = T((8192, 8200), True), T((8192, 8200), True)
timeit("bf16 x[:, 1:4097] + y[:, 3:4099] (unaligned)", lambda: xb[:, 1:4097] + yb[:, 3:4099], 6 * n)
timeit("bf16 x[:, 8:4104] + y[:, 16:4112] (aligned)", lambda: xb[:, 8:4104] + yb[:, 16:4112], 6 * n)
timeit("bf16 x[:, 1:4097].contiguous()", lambda: xb[:, 1:4097].contiguous(), 4 * n)
timeit("x[:, 0:4096].contiguous()", lambda: x[:, 0:4096].contiguous(), 8 * n)
timeit("x[::2].contiguous() f32 (every other row)", lambda: x[::2].contiguous(), 8 * 4096 * 8200)
