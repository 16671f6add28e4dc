#!/usr/bin/env python3
"""Randomised stress of kf_gemm_grouped (round 6): 1..4 problems of random layouts / extents (whole 256-tiles so that the single-grid form runs, and ragged ones so that the
fall-back runs), alpha / beta, 16-bit and float C: every product BIT-identical to kf_gemm alone on the same operands. stress_gemm_grouped.py SEED SECONDS"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
H.set_device(0)
seed, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed)
t_end = time.time() + secs
n = single = 0
while time.time() < t_end:
    code = int(rng.choice([H.BF16, H.F16]))
    count = int(rng.integers(1, 5))
    tiled = rng.random() < 0.6
    pair = rng.random() < 0.35       # the backward pair of a linear layer on large 256-tile shapes: the ONE-grid form (NT then TN)
    if pair:
        count, tiled = 2, True
    probs, keep, want = [], [], []
    for pi in range(count):
        if pair:
            M, N, K = int(rng.choice([2048, 3072, 4096])), int(rng.choice([2048, 2560, 4096])), int(rng.choice([256, 512, 1024, 1280]))
        elif tiled:
            M, N, K = (int(rng.choice([256, 512, 768, 1024])) for _ in range(3))
        else:
            M, N, K = (int(rng.choice([1, 33, 200, 256, 300, 512, 1000])) for _ in range(3))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        if pair:
            ta, tb = (False, True) if pi == 0 else (True, False)
        alpha, beta = float(rng.choice([1.0, 0.5, -2.0])), float(rng.choice([0.0, 0.0, 1.0]))
        cf32 = bool(rng.integers(0, 2)) and tiled
        a = O.from_float(rng.uniform(-1, 1, (K, M) if ta else (M, K)).astype(np.float32), code)
        b = O.from_float(rng.uniform(-1, 1, (N, K) if tb else (K, N)).astype(np.float32), code)
        c = rng.uniform(-1, 1, (M, N)).astype(np.float32) if cf32 else O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
        da, db, dc, dc1 = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(b), H.DevBuf.from_numpy(c), H.DevBuf.from_numpy(c)
        keep.append((da, db, dc, dc1, c, M, N))
        probs.append((int(ta), int(tb), M, N, K, alpha, beta, da.ptr, a.shape[1], db.ptr, b.shape[1], dc.ptr, N, int(cf32)))
        # alone: the same product through kf_gemm_ex (c_f32) / kf_gemm
        need = H.gemm_workspace_bytes(code, ta, tb, M, N, K)
        if cf32:
            H.gemm_ex(code, ta, tb, M, N, K, alpha, da.ptr, a.shape[1], db.ptr, b.shape[1], beta, dc1.ptr, N, c_f32=True)
        else:
            H.gemm(code, ta, tb, M, N, K, alpha, da.ptr, a.shape[1], db.ptr, b.shape[1], beta, dc1.ptr, N, 0, None, None, 0)
    single += int(H.lib().kf_gemm_grouped_single_grid(code, count, (H.GemmProblem * count)(*[H.GemmProblem(*p) for p in probs])))
    H.gemm_grouped(code, probs)
    H.device_sync()
    for i, (da, db, dc, dc1, c, M, N) in enumerate(keep):
        got, alone = dc.to_numpy((M, N), c.dtype), dc1.to_numpy((M, N), c.dtype)
        assert np.array_equal(got.view(np.uint8), alone.view(np.uint8)), ("grouped != alone", code, i, [p[:7] + (p[13],) for p in probs])
        for b_ in (da, db, dc, dc1): b_.free()
    n += 1
print(f"seed {seed}: {n} grouped launches ({single} as one grid): every product bit-identical to kf_gemm alone")
