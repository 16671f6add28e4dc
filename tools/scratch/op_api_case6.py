#!/usr/bin/env python3
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H
rng = np.random.default_rng(0)
def T(shape):
    return kfunca.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32), 0)
def timeit(name, fn, nbytes=0, reps=10):
    for _ in range(3): fn()
    H.device_sync()
    t0 = time.perf_counter()
    for _ in range(reps): r = fn()
    H.device_sync()
    ms = (time.perf_counter() - t0) / reps * 1e3
    print(f"{name:64s} {ms:9.3f} ms" + (f" {nbytes / ms / 1e6:8.0f} GB/s" if nbytes else ""), flush=True)
for cols in (8200, 8192):
    x = T((8192, cols))
    nb = 8 * 4096 * cols
    timeit(f"[8192,{cols}] x[0:4096].contiguous() (contiguous half)", lambda: x[0:4096].contiguous(), nb)
    timeit(f"[8192,{cols}] x[::2].contiguous() (every other row)", lambda: x[::2].contiguous(), nb)
    timeit(f"[8192,{cols}] x[:, 0:{cols//2}].contiguous() (left half of every row)", lambda: x[:, 0:cols // 2].contiguous(), nb)
    timeit(f"[8192,{cols}] x[1::2] + x[::2]", lambda: x[1::2] + x[::2], nb * 3 // 2)
