#!/usr/bin/env python3
"""In-kernel clock of the bf16 GEMM main loop (MI355X_MICROARCH.md, DVFS give-back item 6): a diagnostic build of the
4-wave kernel stamps s_memtime / s_memrealtime around its K loop; after >= 2 s of back-to-back launches on random data the
median over workgroups of cycles / (100 MHz ticks) x 100 MHz is the clock the chip holds under this load, and cycles per
K tile against the 2048 cycles of MFMA work per SIMD is the matrix-pipe utilisation of the loop at that clock."""
import argparse
import ctypes as C
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--zeros", action="store_true", help="all-zero operands (the clock the chip holds without data toggling)")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    n = args.n
    H.set_device(0)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(n, n)).astype(np.float32).view(np.uint32)
    bits = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
    if args.zeros:
        bits[:] = 0
    A, B = H.DevBuf.from_numpy(bits), H.DevBuf.from_numpy(bits[::-1].copy())
    Cb = H.DevBuf(2 * n * n)
    nblk = (n // 256) ** 2
    diag = H.DevBuf(16 * nblk)
    from kfunca_amd import _build
    f = C.CDLL(str(_build.build_diag())).kfdbg_gemm_clock  # the diagnostic build: not part of libkfunca_hip.so
    f.argtypes = [C.c_int64] * 3 + [C.c_void_p] * 5
    t_end = time.time() + args.seconds
    launches = 0
    while time.time() < t_end:
        for _ in range(50):
            H.check(f(n, n, n, A.ptr, B.ptr, Cb.ptr, diag.ptr, None))
        launches += 50
        H.device_sync()
    d = diag.to_numpy((nblk, 2), np.uint64).astype(np.float64)
    clk = d[:, 0] / d[:, 1] * 0.1  # GHz
    per_tile = d[:, 0] / (n // 64)
    out = {"n": n, "operands": "zeros" if args.zeros else "uniform(-1,1)", "launches": launches, "clock_ghz_median": float(np.median(clk)),
           "clock_ghz_min_max": [float(clk.min()), float(clk.max())], "loop_cycles_per_k_tile_median": float(np.median(per_tile)),
           "mfma_cycles_per_k_tile": 2048, "loop_mfma_utilisation": float(2048 / np.median(per_tile)),
           "loop_us_per_k_tile": float(np.median(d[:, 1]) * 0.01 / (n // 64)),
           "peak_at_this_clock_tflops": float(2.5e3 * np.median(clk) / 2.4)}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
