#!/usr/bin/env python3
"""bf16 GEMM at shapes off the square benchmark sizes (back-to-back timing): TFLOP/s."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
shapes = [(4096, 11008, 4096), (4096, 4096, 11008), (8192, 14336, 4096), (8192, 4096, 14336), (4096, 12288, 4096), (32768, 4096, 4096), (4096, 4096, 32768),
          (4000, 4000, 4000), (4100, 4100, 4100), (2048, 8192, 8192), (1024, 8192, 8192), (512, 8192, 8192), (256, 8192, 8192), (128, 8192, 8192), (64, 8192, 8192), (16, 8192, 8192),
          (8192, 8192, 512), (8192, 8192, 128), (3072, 3072, 3072), (6144, 6144, 6144), (5120, 5120, 5120), (16384, 1024, 8192)]
mx = max(max(m * k, k * n, m * n) for m, n, k in shapes)
A, B, C = H.DevBuf(2 * mx), H.DevBuf(2 * mx), H.DevBuf(2 * mx)
for buf in (A, B):
    H.elementwise(H.EW_FILL, H.make_desc([H.View(buf.ptr, (mx,), (1,), H.BF16)], []), 0, 0.5)
for (m, n, k) in shapes:
    for ta, tb, tag in ((0, 0, "NN"), (0, 1, "NT"), (1, 0, "TN")):
        lda = k if not ta else m
        ldb = n if not tb else k
        need = H.gemm_workspace_bytes(H.BF16, ta, tb, m, n, k)
        ws = H.DevBuf(max(need, 16))
        fn = lambda: H.gemm(H.BF16, ta, tb, m, n, k, 1.0, A.ptr, lda, B.ptr, ldb, 0.0, C.ptr, n, 0, None, ws.ptr, need)
        try:
            for _ in range(3): fn()
            H.device_sync()
            e0, e1 = H.Event(), H.Event()
            e0.record(None)
            reps = 10
            for _ in range(reps): fn()
            e1.record(None); H.device_sync()
            ms = e0.elapsed_ms(e1) / reps
            print(f"{m:6d} {n:6d} {k:6d} {tag} {ms:8.4f} ms {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s", flush=True)
        except Exception as ex:
            print(m, n, k, tag, "ERR", str(ex)[:100])
