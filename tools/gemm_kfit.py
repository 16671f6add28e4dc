#!/usr/bin/env python3
"""bf16 GEMM M = N = 4096 over a sweep of K: a linear fit separates the per-K-tile cost of the main loop from the fixed
cost per launch (prologue fill, epilogue stores, launch)."""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402

H.set_device(0)
rng = np.random.default_rng(0)
n = 4096
KS = [128, 256, 512, 1024, 2048, 4096, 8192]
x = rng.uniform(-1, 1, size=(n, max(KS))).astype(np.float32).view(np.uint32)
bits = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
A, B = H.DevBuf.from_numpy(bits), H.DevBuf.from_numpy(bits[::-1].copy())
C = H.DevBuf(2 * n * n)
for tb, tag in ((1, "NT"), (0, "NN")):
    ts = []
    for K in KS:
        lda = max(KS)
        ldb = max(KS) if tb else n
        times = []
        for r in range(12):
            H.profile_reset()
            H.profile_enable(True)
            H.gemm(H.BF16, 0, tb, n, n, K, 1.0, A.ptr, lda, B.ptr, ldb, 0.0, C.ptr, n, 0, None, None, 0)
            H.device_sync()
            H.profile_enable(False)
            if r >= 2:
                times.append(sum(v[0] for v in H.profile_results().values()))
        ts.append(float(np.median(times)))
        print(f"{tag} K={K:5d}  {ts[-1] * 1e3:8.1f} us   {2.0 * n * n * K / ts[-1] / 1e9:8.1f} TF/s")
    slope, icpt = np.polyfit(np.array(KS[2:]) / 64.0, np.array(ts[2:]) * 1e3, 1)
    print(f"{tag}: {slope:.3f} us per 64-deep K tile, {icpt:.1f} us fixed; main-loop rate {2.0 * n * n * 64 / slope / 1e6:.0f} TF/s")
