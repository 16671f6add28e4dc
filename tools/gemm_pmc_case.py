#!/usr/bin/env python3
"""ONE GEMM case, launched a few times back to back: the program tools/gemm_pmc.sh puts behind `rocprofv3 --pmc ... --` (one process per
case and counter pass, so every dispatch of the pass belongs to the case). Cases: BASELINE configs C2 (f32 4096^3) and C4 (bf16 8192^3,
plain and with the fused alpha / beta / bias-row tail), the layouts of the backward, and the sizes below the 256-tile kernel.
    python3 tools/gemm_pmc_case.py "bf16 8192 NN"        (dtype, n, layout [+epi])"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402

CASES = ["bf16 8192 NN", "bf16 8192 NT", "bf16 8192 TN", "bf16 8192 NN+epi", "bf16 4096 NN", "bf16 4096 NN+epi", "bf16 2048 NN", "bf16 2048 NN+epi",
         "bf16 1024 NN", "f32 4096 NN"]


def main():
    dt, n, lay = sys.argv[1].split()
    n = int(n)
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 6
    H.set_device(0)
    rng = np.random.default_rng(1004)
    code = H.BF16 if dt == "bf16" else H.F32
    if code == H.BF16:
        def mk():
            u = rng.uniform(-1, 1, size=(n, n)).astype(np.float32).view(np.uint32)
            return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    else:
        def mk():
            return rng.uniform(-1, 1, size=(n, n)).astype(np.float32)
    A, B = H.DevBuf.from_numpy(mk()), H.DevBuf.from_numpy(mk())
    C = H.DevBuf.from_numpy(mk())
    bias = H.DevBuf.from_numpy(mk()[0].copy())
    epi = lay.endswith("+epi")
    ta, tb = {"NN": (0, 0), "NT": (0, 1), "TN": (1, 0)}[lay[:2]]
    need = H.gemm_workspace_bytes(code, ta, tb, n, n, n)
    ws = H.DevBuf(max(need, 16))
    for _ in range(rounds + 4):
        if epi:
            H.gemm(code, ta, tb, n, n, n, 0.5, A.ptr, n, B.ptr, n, 2.0, C.ptr, n, H.EPI_BIAS_ROW, bias.ptr, ws.ptr, need)
        else:
            H.gemm(code, ta, tb, n, n, n, 1.0, A.ptr, n, B.ptr, n, 0.0, C.ptr, n, 0, None, ws.ptr, need)
    H.device_sync()


if __name__ == "__main__":
    main()
