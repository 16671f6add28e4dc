#!/usr/bin/env python3
"""4-wave vs 8-wave 256-tile kernel on the rectangular products of config C5 and on grids beyond two rounds (same process, interleaved)."""
import os, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
rng = np.random.default_rng(0)
def bf16(shape):
    x = rng.uniform(-1, 1, size=shape).astype(np.float32).view(np.uint32)
    return ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
shapes = [(4096, 12288, 4096), (4096, 16384, 4096), (4096, 4096, 16384), (16384, 4096, 4096), (5120, 5120, 5120), (8192, 8192, 8192), (4096, 4096, 4096)]
for (M, N, K) in shapes:
    big = max(M * K, K * N, M * N)
    A, B, C = H.DevBuf.from_numpy(bf16((big,))), H.DevBuf.from_numpy(bf16((big,))), H.DevBuf(2 * M * N)
    res = {}
    for r in range(5):
        for v in ("KF_GEMM_W8", "KF_GEMM_W4"):
            for e in ("KF_GEMM_W8", "KF_GEMM_W4"):
                os.environ.pop(e, None)
            os.environ[v] = "1"
            H.knobs_reload()
            for tag, ta, tb in (("NN", 0, 0), ("NT", 0, 1), ("TN", 1, 0)):
                lda = M if ta else K
                ldb = K if tb else N
                H.profile_reset(); H.profile_enable(True)
                H.gemm(H.BF16, ta, tb, M, N, K, 1.0, A.ptr, lda, B.ptr, ldb, 0.0, C.ptr, N, 0, None, None, 0)
                H.device_sync(); H.profile_enable(False)
                if r:
                    res.setdefault((v, tag), []).append(sum(x[0] for x in H.profile_results().values()))
    line = f"{M}x{N}x{K}: "
    for tag in ("NN", "NT", "TN"):
        w8, w4 = np.median(res[("KF_GEMM_W8", tag)]), np.median(res[("KF_GEMM_W4", tag)])
        line += f"{tag} w8 {2.0 * M * N * K / w8 / 1e9:6.0f} w4 {2.0 * M * N * K / w4 / 1e9:6.0f} TF | "
    print(line, flush=True)
