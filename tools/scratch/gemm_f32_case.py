#!/usr/bin/env python3
"""f32 GEMM at a few shapes through the C ABI (random operands, back to back): TFLOP/s."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
rng = np.random.default_rng(0)
shapes = [(4096, 4096, 4096), (4096, 4097, 4096), (4000, 4000, 4000), (4096, 4224, 4096), (4096, 4160, 4096), (2048, 2048, 2048), (8192, 8192, 1024), (1024, 1024, 1024), (512, 16384, 4096)]
mx = max(max(m * k, k * n, m * n) for m, n, k in shapes)
A, B, C = H.DevBuf.from_numpy(rng.uniform(-1, 1, mx).astype(np.float32)), H.DevBuf.from_numpy(rng.uniform(-1, 1, mx).astype(np.float32)), H.DevBuf(4 * mx)
for (m, n, k) in shapes:
    need = H.gemm_workspace_bytes(H.F32, 0, 0, m, n, k)
    ws = H.DevBuf(max(need, 16))
    fn = lambda: H.gemm(H.F32, 0, 0, m, n, k, 1.0, A.ptr, k, B.ptr, n, 0.0, C.ptr, n, 0, None, ws.ptr, need)
    for _ in range(3): fn()
    H.device_sync()
    H.profile_reset(); H.profile_enable(True); fn(); H.device_sync(); H.profile_enable(False)
    names = sorted(H.profile_results())
    e0, e1 = H.Event(), H.Event()
    e0.record(None)
    for _ in range(10): fn()
    e1.record(None); H.device_sync()
    ms = e0.elapsed_ms(e1) / 10
    print(f"{m:6d} {n:6d} {k:6d} {ms:8.4f} ms {2.0 * m * n * k / ms / 1e9:8.1f} TFLOP/s  {names}", flush=True)
