#!/usr/bin/env python3
"""Timing of the f32 attention paths (the reference's dtype: exact-f32 MFMA forward, generic backward) and of 16-bit shapes
the bf16 MFMA kernels do not cover, through the C ABI; HIP-event times from the library's profiling mode."""
import argparse
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=3)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    out = {}
    H.set_device(0)
    rng = np.random.default_rng(0)
    # the C3 shape in the reference's own dtype (f32), f32 at D = 64, and two 16-bit shapes the bf16 MFMA kernels do not take
    for (code, B, Hh, S, D) in ((H.F32, 8, 32, 4096, 128), (H.F32, 8, 32, 4096, 64), (H.BF16, 2, 8, 4096, 64), (H.BF16, 2, 8, 4000, 128)):
        es = 4 if code == H.F32 else 2
        n = B * Hh * S * D
        per = Hh * S * D
        host = rng.uniform(-1, 1, per).astype(np.float32)
        if code == H.BF16:
            host = (host.view(np.uint32) >> 16).astype(np.uint16)
        bufs = {}
        for k in ("q", "k", "v", "do"):
            bufs[k] = H.DevBuf(n * es)
            for i in range(B):
                H.check(H.lib().kf_memcpy_h2d(bufs[k].ptr + i * per * es, host.ctypes.data, per * es, None))
        for k in ("o", "dq", "dk", "dv"):
            bufs[k] = H.DevBuf(n * es)
        lse = H.DevBuf(4 * B * Hh * S)
        need = H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D)
        ws = H.DevBuf(max(need, 16))
        res = {}
        for r in range(args.rounds + 1):
            H.profile_reset()
            H.profile_enable(True)
            H.attn_fwd(code, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr)
            H.attn_bwd(code, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,
                       bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, need)
            H.device_sync()
            H.profile_enable(False)
            if r:
                for k, (ms, cnt) in H.profile_results().items():
                    res.setdefault(k, []).append(ms / cnt)
        pair = B * Hh * S * S * D / 2.0
        print(f"== dtype {code} B={B} H={Hh} S={S} D={D}")
        units = {"fwd": 4, "dkv": 8, "dq": 6}  # executed products (S and dP are recomputed in both backward kernels)
        for k, xs in res.items():
            med = float(np.median(xs))
            u = next((v for kk, v in units.items() if kk in k), 0)
            tf = u * pair / (med * 1e-3) / 1e12 if u else None
            out.setdefault(f"dtype{code}_B{B}_H{Hh}_S{S}_D{D}", {})[k] = {"ms": med, "executed_tflops": tf}
            print(f"  {k:24s} {med:9.3f} ms   " + (f"{tf:7.1f} TF/s executed" if tf else ""))
    if args.json:
        import json
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
