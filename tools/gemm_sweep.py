#!/usr/bin/env python3
"""BASELINE configs C2 / C4 and the MFMA-utilisation sweep: GEMM through the C ABI at several sizes, layouts and epilogues;
per-launch HIP-event times from the library's profiling mode. Writes one JSON object (profiles/r01_gemm_sweep.json)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402

PEAK = {"bf16": 2500.0, "f32": 157.3}  # dense MFMA TFLOP/s (MI355X_MICROARCH.md)


def bf16(rng, shape):
    x = rng.uniform(-1, 1, size=shape).astype(np.float32)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def timed(fn, rounds):
    for _ in range(8):   # the first case of a size runs behind seconds of host-side operand generation: the clock has to come back up first
        fn()             # (round 3's "21 % NN hole at 8192^3" was mostly this: the same NN product measured fourth ran 12 % faster)
    H.device_sync()
    H.profile_reset()
    H.profile_enable(True)
    for _ in range(rounds):
        fn()
    H.device_sync()
    H.profile_enable(False)
    res = H.profile_results()
    return sum(v[0] for v in res.values()) / rounds, sorted(res)


def timed_b2b(fn, n):
    """n launches behind each other between ONE event pair: what a caller that queues work sees (launch latency hidden behind the
    previous kernel, the clock settled) - the per-launch event times above carry ~10 us of dispatch gap each."""
    fn()
    H.device_sync()
    e0, e1 = H.Event(), H.Event()
    e0.record(None)
    for _ in range(n):
        fn()
    e1.record(None)
    H.device_sync()
    return e0.elapsed_ms(e1) / n


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--b2b", type=int, default=50, help="launches per back-to-back measurement (0 = skip)")
    ap.add_argument("--sizes", default="1024,2048,4096,8192")
    ap.add_argument("--no-f32", action="store_true")
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    H.set_device(0)
    rng = np.random.default_rng(1004)  # seed = 1000 + config number (C4)
    out = {}
    bsizes = tuple(int(x) for x in args.sizes.split(","))
    for dt, code, sizes in (("bf16", H.BF16, bsizes), ("f32", H.F32, () if args.no_f32 else tuple(x for x in bsizes if x <= 4096))):
        for n in sizes:
            mk = (lambda: bf16(rng, (n, n))) if code == H.BF16 else (lambda: rng.uniform(-1, 1, size=(n, n)).astype(np.float32))
            A, B, G = (H.DevBuf.from_numpy(mk()) for _ in range(3))
            es = 2 if code == H.BF16 else 4
            C = H.DevBuf(es * n * n)
            bias = H.DevBuf.from_numpy(mk()[0].copy())
            need = max(H.gemm_workspace_bytes(code, ta, tb, n, n, n) for ta, tb in ((0, 0), (0, 1), (1, 0)))
            ws = H.DevBuf(max(need, 16))
            rounds = args.rounds if n <= 4096 else max(3, args.rounds // 2)
            cases = {
                "NN fwd": lambda: H.gemm(code, 0, 0, n, n, n, 1.0, A.ptr, n, B.ptr, n, 0.0, C.ptr, n, 0, None, ws.ptr, need),
                "NT dA": lambda: H.gemm(code, 0, 1, n, n, n, 1.0, G.ptr, n, B.ptr, n, 0.0, C.ptr, n, 0, None, ws.ptr, need),
                "TN dB": lambda: H.gemm(code, 1, 0, n, n, n, 1.0, A.ptr, n, G.ptr, n, 0.0, C.ptr, n, 0, None, ws.ptr, need),
                "NN alpha*AB + beta*C + bias row (C4 epilogue)":
                    lambda: H.gemm(code, 0, 0, n, n, n, 0.5, A.ptr, n, B.ptr, n, 2.0, C.ptr, n, H.EPI_BIAS_ROW, bias.ptr, ws.ptr, need),
            }
            # every case is measured TWICE, in forward and then in reverse order, and the better run is kept: the first case of a size runs
            # behind seconds of host-side operand generation and the last behind a warm chip - round 4's "NN is 10 % behind NT at 8192^3"
            # was in good part the order (NN was always first); both runs are recorded (ms_runs)
            first = {tag: timed(fn, rounds) for tag, fn in cases.items()}
            second = {tag: timed(cases[tag], rounds) for tag in reversed(list(cases))}
            for tag, fn in cases.items():
                ms, kernels = min(first[tag], second[tag], key=lambda r: r[0])
                tf = 2.0 * n ** 3 / (ms * 1e-3) / 1e12
                out[f"{dt} {n}^3 {tag}"] = {"ms": ms, "ms_runs": [first[tag][0], second[tag][0]], "TFLOP/s": tf, "frac_of_mfma_peak": tf / PEAK[dt], "kernels": kernels}
                b2b = ""
                if args.b2b:
                    mb = timed_b2b(fn, args.b2b if n <= 4096 else max(5, args.b2b // 5))
                    out[f"{dt} {n}^3 {tag}"].update({"ms_back_to_back": mb, "TFLOP/s_back_to_back": 2.0 * n ** 3 / (mb * 1e-3) / 1e12})
                    b2b = f"| back to back {mb:9.4f} ms {2.0 * n ** 3 / (mb * 1e-3) / 1e12:8.1f} TF/s"
                print(f"{dt} {n:5d}^3 {tag:48s} {ms:9.4f} ms {tf:8.1f} TF/s  {tf / PEAK[dt] * 100:5.1f}% of {PEAK[dt]:.0f} {b2b} {kernels}", flush=True)
            fb = sum(out[f"{dt} {n}^3 {t}"]["ms"] for t in ("NN fwd", "NT dA", "TN dB"))
            out[f"{dt} {n}^3 fwd+bwd"] = {"ms": fb, "TFLOP/s": 6.0 * n ** 3 / (fb * 1e-3) / 1e12, "frac_of_mfma_peak": 6.0 * n ** 3 / (fb * 1e-3) / 1e12 / PEAK[dt]}
            if args.b2b:
                fbb = sum(out[f"{dt} {n}^3 {t}"]["ms_back_to_back"] for t in ("NN fwd", "NT dA", "TN dB"))
                out[f"{dt} {n}^3 fwd+bwd"].update({"ms_back_to_back": fbb, "TFLOP/s_back_to_back": 6.0 * n ** 3 / (fbb * 1e-3) / 1e12})
            print(f"{dt} {n:5d}^3 fwd+bwd {fb:9.4f} ms {out[f'{dt} {n}^3 fwd+bwd']['TFLOP/s']:8.1f} TF/s", flush=True)
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
