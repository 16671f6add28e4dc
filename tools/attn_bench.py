#!/usr/bin/env python3
"""Micro-benchmark of the attention kernels at the BASELINE C3 shape (or smaller), per-kernel HIP-event
times through the C ABI's profiling mode; variants are interleaved in one process (guide rule 24)."""
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
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--S", type=int, default=4096)
    ap.add_argument("--D", type=int, default=128, help="head size: 128 or 64 (both have native matrix-core kernels)")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--variants", default="default")  # comma list of ENV=1 settings, e.g. default,KF_ATTN_DKV_V2
    ap.add_argument("--no-bwd", action="store_true")
    ap.add_argument("--wall", type=int, default=0, help="also time N back-to-back backward calls per variant with one event pair (no per-kernel events)")
    ap.add_argument("--zeros", action="store_true", help="all-zero operands: the same instruction stream without the power limit (the schedule-bound time)")
    args = ap.parse_args()
    B, Hh, S, D = args.B, args.H, args.S, args.D
    H.set_device(0)
    rng = np.random.default_rng(0)
    per = Hh * S * D * 2
    bufs = {}
    for name in ("q", "k", "v", "do"):
        host = bf16(rng, (Hh, S, D))
        if args.zeros:
            host[:] = 0
        b = H.DevBuf(B * per)
        for i in range(B):
            H.check(H.lib().kf_memcpy_h2d(b.ptr + i * per, host.ctypes.data, per, None))
        bufs[name] = b
    for name in ("o", "dq", "dk", "dv"):
        bufs[name] = H.DevBuf(B * per)
    lse = H.DevBuf(4 * B * Hh * S)
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    ws = H.DevBuf(need)
    pair = B * Hh * S * S * D / 2.0
    flops = {"attn_fwd_mfma_d64": 4 * pair, "attn_bwd_dkv_mfma_d64": 8 * pair, "attn_bwd_dq_mfma_d64": 2 * pair, "attn_fwd_mfma": 4 * pair, "attn_fwd_mfma_v2": 4 * pair, "attn_fwd_mfma_v1": 4 * pair, "attn_bwd_dkv_mfma": 8 * pair, "attn_bwd_dq_mfma": 2 * pair, "attn_bwd_dq_mfma_split": 6 * pair, "attn_bwd_dq_mfma_v1": 6 * pair, "attn_bwd_dkv_mfma_v1": 8 * pair, "attn_bwd_dkv_mfma_v3": 8 * pair, "attn_bwd_dkv_mfma_v2": 8 * pair}
    variants = args.variants.split(",")
    results = {v: {} for v in variants}
    for r in range(args.rounds + 1):
        for v in variants:
            for e in [x for x in os.environ if x.startswith("KF_ATTN")]:
                del os.environ[e]
            if v != "default":
                for kv in v.split("+"):
                    k, _, val = kv.partition("=")
                    os.environ[k] = val or "1"
            H.knobs_reload()
            H.profile_reset()
            H.profile_enable(True)
            H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr)
            if not args.no_bwd:
                H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,
                           bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, need)
            H.device_sync()
            H.profile_enable(False)
            if r == 0:
                continue  # warm-up
            for k, (ms, n) in H.profile_results().items():
                results[v].setdefault(k, []).append(ms / n)
    if args.wall:
        for rep in range(2):
            for v in variants:
                for e in [x for x in os.environ if x.startswith("KF_ATTN")]:
                    del os.environ[e]
                if v != "default":
                    for kv in v.split("+"):
                        k, _, val = kv.partition("=")
                        os.environ[k] = val or "1"
                H.knobs_reload()
                nd = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
                call = lambda: H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,  # noqa: E731
                                          bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, min(nd, need))
                call()
                H.device_sync()
                e0, e1 = H.Event(), H.Event()
                e0.record(None)
                for _ in range(args.wall):
                    call()
                e1.record(None)
                H.device_sync()
                print(f"wall[{rep}] {v:60s} {e0.elapsed_ms(e1) / args.wall:8.3f} ms per backward (workspace {min(nd, need) / 2 ** 20:.0f} MiB)", flush=True)
    for v in variants:
        print(f"== {v}  (B={B} H={Hh} S={S} D={D})")
        for k, xs in results[v].items():
            med, mn = float(np.median(xs)), float(np.min(xs))
            tf = f"{flops[k] / (med * 1e-3) / 1e12:8.1f} TF/s (executed FLOPs)" if k in flops else ""
            print(f"  {k:22s} median {med:8.3f} ms  min {mn:8.3f} ms {tf}")


if __name__ == "__main__":
    main()
