#!/usr/bin/env python3
"""bf16 GEMM at mid-size grids (100 .. 200 tiles of 256^2), random operands, back to back: TFLOP/s."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
rng = np.random.default_rng(0)
shapes = [(1024, 8192, 8192), (8192, 1024, 8192), (512, 8192, 8192), (768, 8192, 4096), (1280, 5120, 5120), (1536, 6144, 6144), (2560, 2560, 2560), (2816, 2816, 2816), (2048, 4096, 4096), (3072, 3072, 3072), (3072, 3072, 8192), (3328, 3328, 3328), (3584, 3584, 3584), (2048, 2048, 2048), (2304, 2304, 4096)]
mx = max(max(m * k, k * n, m * n) for m, n, k in shapes)
src = rng.uniform(-1, 1, mx).astype(np.float32)
u = src.view(np.uint32)
bf = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
A, B, C = H.DevBuf.from_numpy(bf), H.DevBuf.from_numpy(bf[::-1].copy()), H.DevBuf(2 * mx)
for (m, n, k) in shapes:
    for ta, tb, tag in ((0, 0, "NN"), (0, 1, "NT")):
        lda = k if not ta else m
        ldb = n if not tb else k
        need = H.gemm_workspace_bytes(H.BF16, ta, tb, m, n, k)
        ws = H.DevBuf(max(need, 16))
        fn = lambda: H.gemm(H.BF16, ta, tb, m, n, k, 1.0, A.ptr, lda, B.ptr, ldb, 0.0, C.ptr, n, 0, None, ws.ptr, need)
        for _ in range(5): fn()
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(20): fn()
        e1.record(None); H.device_sync()
        ms = e0.elapsed_ms(e1) / 20
        print(f"{m:6d} {n:6d} {k:6d} {tag} tiles256 {m // 256 * (n // 256):4d} {ms:8.4f} ms {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
