#!/usr/bin/env python3
"""VERDICT round 4 #7, the experiment: can the HBM-bound half of the attention backward (dQ streams 4.3 GB of dS with the matrix pipes
65 % idle) hide under matrix-bound GEMMs that do not depend on it - the three dW products of config C5's MLP (A^T dC, [4096 x 16384] from
4096 tokens: 0.55 TFLOP each)? Measured at the C ABI, per GPU shapes of C5 (B 1, H 32, S 4096, D 128):
  sequential   attention backward, then the three dW GEMMs, ONE stream
  two streams  attention backward on stream A, the GEMMs on stream B, issued together (the dispatcher is free to co-schedule)
  staggered    the GEMMs start when stream A has been given its work (an event recorded before the backward is issued): same thing with the
               GEMMs queued first
Both kernels families take a whole CU per workgroup (the GEMM: 4 waves x 512 registers; dK/dV the same; dQ 8 waves x 256): whatever
overlap exists is between the tail of one kernel and the head of the next, not side by side on a CU."""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402


def bf16(rng, shape):
    u = rng.uniform(-1, 1, size=shape).astype(np.float32).view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def main():
    H.set_device(0)
    rng = np.random.default_rng(1005)
    B, Hh, S, D, T, DM, F = 1, 32, 4096, 128, 4096, 4096, 16384
    q, k, v, go = (H.DevBuf.from_numpy(bf16(rng, (B, Hh, S, D))) for _ in range(4))
    o, lse = H.DevBuf(2 * B * Hh * S * D), H.DevBuf(4 * B * Hh * S)
    dq, dk, dv = (H.DevBuf(2 * B * Hh * S * D) for _ in range(3))
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    ws = H.DevBuf(need)
    H.attn_fwd(H.BF16, B, Hh, S, S, D, q.ptr, k.ptr, v.ptr, o.ptr, lse.ptr)
    act, dy = H.DevBuf.from_numpy(bf16(rng, (T, F))), H.DevBuf.from_numpy(bf16(rng, (T, DM)))
    dws = [H.DevBuf(2 * F * DM) for _ in range(3)]
    sa, sb = H.Stream(), H.Stream()

    def bwd(st):
        H.attn_bwd(H.BF16, B, Hh, S, S, D, q.ptr, k.ptr, v.ptr, o.ptr, lse.ptr, go.ptr, dq.ptr, dk.ptr, dv.ptr, ws.ptr, need, st)

    def gemms(st):
        for w in dws:  # dW[F, DM] = act^T[F, T] dy[T, DM]
            H.gemm(H.BF16, 1, 0, F, DM, T, 1.0, act.ptr, F, dy.ptr, DM, 0.0, w.ptr, DM, 0, None, None, 0, st)

    def run(mode):
        e0, e1, eb = H.Event(), H.Event(), H.Event()
        e0.record(sa.handle)
        if mode == "sequential":
            bwd(sa.handle)
            gemms(sa.handle)
        else:
            H.stream_wait_event(sb.handle, e0)
            if mode == "staggered":
                gemms(sb.handle)
                bwd(sa.handle)
            else:
                bwd(sa.handle)
                gemms(sb.handle)
            eb.record(sb.handle)
            H.stream_wait_event(sa.handle, eb)
        e1.record(sa.handle)
        e1.sync()
        return e0.elapsed_ms(e1)

    def alone(fn):
        e0, e1 = H.Event(), H.Event()
        e0.record(sa.handle)
        fn(sa.handle)
        e1.record(sa.handle)
        e1.sync()
        return e0.elapsed_ms(e1)

    out = {}
    for _ in range(3):
        run("sequential"); run("two streams")
    for name, f in (("attention backward alone", lambda: alone(bwd)), ("three dW GEMMs alone", lambda: alone(gemms)), ("sequential", lambda: run("sequential")),
                    ("two streams", lambda: run("two streams")), ("staggered", lambda: run("staggered"))):
        ts = sorted(f() for _ in range(12))
        out[name] = {"median_ms": ts[len(ts) // 2], "min_ms": ts[0]}
        print(f"{name:28s} median {ts[len(ts) // 2]:7.3f} ms  min {ts[0]:7.3f} ms", flush=True)
    s = out["sequential"]["median_ms"]
    print(f"two streams / sequential = {out['two streams']['median_ms'] / s:.3f}; staggered / sequential = {out['staggered']['median_ms'] / s:.3f}  "
          f"(< 1 would be overlap; dQ is about 30 % of the attention backward's time)")
    if len(sys.argv) > 1:
        Path(sys.argv[1]).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
