#!/usr/bin/env python3
"""Experiment (VERDICT round 3 #5a): is the NN hole at 8192^3 the power-of-two row stride of the [K][N] operand? The same product with
ldb = N and with ldb = N + pad (B stored in a wider buffer), per launch and back to back, against NT."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H


def bf16(rng, shape):
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def t(fn, n=20):
    fn(); H.device_sync()
    H.profile_reset(); H.profile_enable(True)
    for _ in range(6): fn()
    H.device_sync(); H.profile_enable(False)
    per = sum(v[0] for v in H.profile_results().values()) / 6
    e0, e1 = H.Event(), H.Event()
    e0.record(None)
    for _ in range(n): fn()
    e1.record(None); H.device_sync()
    return per, e0.elapsed_ms(e1) / n


H.set_device(0)
rng = np.random.default_rng(1)
for (M, N, K) in ((8192, 8192, 8192), (4096, 12288, 4096), (4096, 4096, 4096)):
    fl = 2.0 * M * N * K
    A = H.DevBuf.from_numpy(bf16(rng, (M, K)))
    C = H.DevBuf(2 * M * N)
    for pad in (0, 64, 128, 256, 1024):
        ldb = N + pad
        B = H.DevBuf.from_numpy(bf16(rng, (K, ldb)))
        p, b = t(lambda: H.gemm(H.BF16, 0, 0, M, N, K, 1.0, A.ptr, K, B.ptr, ldb, 0.0, C.ptr, N, 0, None, None, 0))
        print(f"{M}x{N}x{K} NN ldb = N + {pad:5d}: per launch {p:.4f} ms {fl / p / 1e9:7.1f} TF/s | back to back {b:.4f} ms {fl / b / 1e9:7.1f} TF/s", flush=True)
    Bt = H.DevBuf.from_numpy(bf16(rng, (N, K)))
    p, b = t(lambda: H.gemm(H.BF16, 0, 1, M, N, K, 1.0, A.ptr, K, Bt.ptr, K, 0.0, C.ptr, N, 0, None, None, 0))
    print(f"{M}x{N}x{K} NT               : per launch {p:.4f} ms {fl / p / 1e9:7.1f} TF/s | back to back {b:.4f} ms {fl / b / 1e9:7.1f} TF/s", flush=True)
    for gm in (1, 2, 4, 8, 16):
        B = H.DevBuf.from_numpy(bf16(rng, (K, N)))
        with H.knobs(KF_GEMM_GROUP_M=str(gm)):
            p, b = t(lambda: H.gemm(H.BF16, 0, 0, M, N, K, 1.0, A.ptr, K, B.ptr, N, 0.0, C.ptr, N, 0, None, None, 0))
        print(f"{M}x{N}x{K} NN group_m {gm:2d}    : per launch {p:.4f} ms {fl / p / 1e9:7.1f} TF/s | back to back {b:.4f} ms {fl / b / 1e9:7.1f} TF/s", flush=True)
