#!/usr/bin/env python3
"""Debug aid for the generated forward (tools/gen_attn_fwd.py): the new kernel against the 8-wave kernel (KF_ATTN_FWD_V3) and the
oracle on a few shapes, with an error map per 32-row group x 64-key... (rows only) to localise a wrong wave / block / tile."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402
from oracle import oracle as O  # noqa: E402


def run(code, q, k, v, v3):
    B, Hh, S, D = q.shape
    bq, bk, bv = (H.DevBuf.from_numpy(x) for x in (q, k, v))
    bo, bl = H.DevBuf(q.nbytes), H.DevBuf(4 * B * Hh * S)
    with H.knobs(KF_ATTN_FWD_V3="1" if v3 else None):
        H.attn_fwd(code, B, Hh, S, k.shape[2], D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, bl.ptr)
        H.device_sync()
    return bo.to_numpy(q.shape, np.uint16), bl.to_numpy((B, Hh, S), np.float32)


def main():
    H.set_device(0)
    rng = np.random.default_rng(3)
    shapes = [(1, 1, 256), (1, 1, 512), (1, 2, 1024), (2, 8, 2048)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for code, name in ((H.BF16, "bf16"), (H.F16, "f16")):
        for (B, Hh, S) in shapes:
            q, k, v = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, 128)).astype(np.float32), code) for _ in range(3))
            t0 = time.time()
            o1, l1 = run(code, q, k, v, False)
            o3, l3 = run(code, q, k, v, True)
            f1, f3 = O.to_float(o1, code).astype(np.float64), O.to_float(o3, code).astype(np.float64)
            d = np.abs(f1 - f3)
            rows = d.max(axis=3)  # [B, H, S]
            grp = rows.reshape(B, Hh, S // 32, 32).max(axis=3)
            print(f"{name} B{B} H{Hh} S{S}: max|O - O_v3| {d.max():.3e}  max|lse - lse_v3| {np.abs(l1 - l3).max():.3e}  nan {np.isnan(f1).sum()}  ({time.time() - t0:.1f} s)")
            if d.max() > 2e-2 or np.isnan(f1).any():
                print("  worst 32-row groups of (b0,h0):", np.round(grp[0, 0], 3).tolist())
                idx = np.argwhere(d > 2e-2)[:12]
                for (b_, h_, r_, c_) in idx:
                    print(f"    [{b_},{h_},{r_},{c_}] w4 {f1[b_, h_, r_, c_]:+.5f} (0x{o1[b_, h_, r_, c_]:04x})  v3 {f3[b_, h_, r_, c_]:+.5f} (0x{o3[b_, h_, r_, c_]:04x})")
                print("    count", int((d > 2e-2).sum()))
                print("  lse diff per 32-row group:", np.round(np.abs(l1 - l3).reshape(B, Hh, S // 32, 32).max(axis=3)[0, 0], 3).tolist())


if __name__ == "__main__":
    main()
