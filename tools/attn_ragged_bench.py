#!/usr/bin/env python3
"""Sequence lengths next to the benchmark shape (missing #2: S = 4000 cost 1.70 ms where 4096 cost 1.21): bf16 causal attention
forward + backward at B 8, H 32, D 128 | 64 for a list of S, interleaved over several rounds in one process; per-kernel HIP-event medians,
the step per token against S = 4096's. Everything through the C ABI with NO padded copies: rows beyond a tensor's end are zero-filled / dropped
by the kernels' buffer descriptors (include/kfunca_hip.h, kf_attn_fwd)."""
import argparse
import json
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
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--S", default="4096,4000,4095,3969,3840,2049,1000")
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    B, Hh, D = args.B, args.H, args.D
    Ss = [int(x) for x in args.S.split(",")]
    H.set_device(0)
    rng = np.random.default_rng(0)
    Smax = max(Ss)
    per = Hh * Smax * D * 2
    host = {n: bf16(rng, (Hh, Smax, D)) for n in ("q", "k", "v", "do")}
    bufs = {}
    for n in ("q", "k", "v", "do", "o", "dq", "dk", "dv"):
        bufs[n] = H.DevBuf(B * per)
    lse = H.DevBuf(4 * B * Hh * Smax)
    need_max = max(H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D) for S in Ss)
    ws = H.DevBuf(need_max)
    res = {S: {} for S in Ss}
    for r in range(args.rounds + 1):
        for S in Ss:
            # the tensors of this S, contiguous [B, H, S, D]: re-packed from the one random head set (only the first round pays the copies' time)
            for n in ("q", "k", "v", "do"):
                x = np.ascontiguousarray(host[n][:, :S])
                for b in range(B):
                    H.check(H.lib().kf_memcpy_h2d(bufs[n].ptr + b * Hh * S * D * 2, x.ctypes.data, x.nbytes, None))
            need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
            H.device_sync()
            H.profile_reset()
            H.profile_enable(True)
            for _ in range(3):
                H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr)
                H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,
                           bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, need)
            H.device_sync()
            H.profile_enable(False)
            if r == 0:
                continue
            for k, (ms, n) in H.profile_results().items():
                res[S].setdefault(k, []).append(ms / n)
    out = {"B": B, "H": Hh, "D": D, "rounds": args.rounds, "cases": []}
    base = None
    print(f"bf16 causal attention fwd + bwd, B {B} H {Hh} D {D}: median ms per kernel over {args.rounds} interleaved rounds x 3 launches")
    for S in Ss:
        med = {k: float(np.median(v)) for k, v in res[S].items()}
        step = sum(med.values())
        us_tok = step * 1e3 / (B * S)
        if S == 4096:
            base = us_tok
        out["cases"].append({"S": S, "kernels_ms": med, "step_ms": step, "us_per_token": us_tok})
    for c in out["cases"]:
        rel = c["us_per_token"] / base if base else float("nan")
        c["per_token_vs_4096"] = rel
        ks = "  ".join(f"{k.replace('attn_', '')} {v:.3f}" for k, v in c["kernels_ms"].items())
        print(f"  S {c['S']:5d}: step {c['step_ms']:.3f} ms = {c['us_per_token']:.4f} us/token ({rel:.3f} x S=4096)   {ks}")
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
