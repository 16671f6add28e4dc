#!/usr/bin/env python3
"""bf16 GEMM micro-benchmark through the C ABI: forward + the two backward products at n^3, per-kernel HIP-event
times; variants (env switches) interleaved in one process."""
import argparse
import os
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402


def bf16(rng, shape):
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--variants", default="default")
    ap.add_argument("--dtype", default="bf16", choices=["bf16", "f32", "f64"])
    args = ap.parse_args()
    n = args.n
    H.set_device(0)
    rng = np.random.default_rng(0)
    DT = {"bf16": H.BF16, "f32": H.F32, "f64": H.F64}[args.dtype]
    mk = (lambda: bf16(rng, (n, n))) if DT == H.BF16 else (lambda: rng.uniform(-1, 1, size=(n, n)).astype(np.float32 if DT == H.F32 else np.float64))
    A, W, G = (H.DevBuf.from_numpy(mk()) for _ in range(3))
    out = H.DevBuf(8 * n * n)
    need = max(H.gemm_workspace_bytes(DT, ta, tb, n, n, n) for ta, tb in ((0, 0), (0, 1), (1, 0)))
    ws = H.DevBuf(max(need, 16))
    res = {}
    for r in range(args.rounds + 1):
        for v in args.variants.split(","):
            for e in [x for x in os.environ if x.startswith("KF_GEMM")]:
                del os.environ[e]
            if v != "default":
                name, _, val = v.partition("=")
                os.environ[name] = val or "1"
            H.knobs_reload()
            for tag, ta, tb, X, Y in (("NN fwd", 0, 0, A, W), ("NT dA", 0, 1, G, W), ("TN dB", 1, 0, A, G)):
                H.profile_reset()
                H.profile_enable(True)
                H.gemm(DT, ta, tb, n, n, n, 1.0, X.ptr, n, Y.ptr, n, 0.0, out.ptr, n, 0, None, ws.ptr, need)
                H.device_sync()
                H.profile_enable(False)
                if r:
                    for k, (ms, cnt) in H.profile_results().items():
                        res.setdefault(v, {}).setdefault(f"{k} [{tag}]", []).append((ms, cnt))
    for v, d in res.items():
        print(f"== {v} n={n}")
        tot = 0.0
        for k, xs in d.items():
            ms = float(np.median([m for m, _ in xs]))
            cnt = xs[0][1]
            tot += ms
            tf = f"{2.0 * n ** 3 * cnt / (ms * 1e-3) / 1e12:8.1f} TF/s" if ("mfma" in k or "f32" in k or "generic" in k) else ""
            print(f"  {k:22s} {ms / cnt:8.4f} ms x {cnt}  {tf}")
        print(f"  fwd+bwd total {tot:.4f} ms -> {6.0 * n ** 3 / (tot * 1e-3) / 1e12:.1f} TF/s")


if __name__ == "__main__":
    main()
