#!/usr/bin/env python3
"""Debug aid for the generated dK/dV (tools/gen_attn_dkv.py): the new kernel against the 32-key kernel (KF_ATTN_DKV_V4) - dK, dV and,
through the stored dS, dQ - with an error map per 32-key group to localise a wrong wave / sub-block / slice."""
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402
from oracle import oracle as O  # noqa: E402


def f32(bits, code):
    return (bits.astype(np.uint32) << 16).view(np.float32).astype(np.float64) if code == H.BF16 else bits.view(np.float16).astype(np.float64)


def run(code, q, k, v, go, v4, split=False):
    B, Hh, S, D = q.shape
    bq, bk, bv, bgo = (H.DevBuf.from_numpy(x) for x in (q, k, v, go))
    bo, bl = H.DevBuf(q.nbytes), H.DevBuf(4 * B * Hh * S)
    H.attn_fwd(code, B, Hh, S, S, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, bl.ptr)
    bdq, bdk, bdv = (H.DevBuf(q.nbytes) for _ in range(3))
    with H.knobs(KF_ATTN_DKV_V4="1" if v4 else None, KF_ATTN_SPLIT_BWD="1" if split else None):
        need = H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D)
        ws = H.DevBuf(need)
        H.attn_bwd(code, B, Hh, S, S, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, bl.ptr, bgo.ptr, bdq.ptr, bdk.ptr, bdv.ptr, ws.ptr, need)
        H.device_sync()
    return [b.to_numpy(q.shape, np.uint16) for b in (bdq, bdk, bdv)]


def main():
    H.set_device(0)
    rng = np.random.default_rng(5)
    shapes = [(1, 1, 256), (1, 1, 512), (1, 2, 1024), (2, 8, 2048)]
    if len(sys.argv) > 1:
        shapes = [tuple(int(x) for x in a.split(",")) for a in sys.argv[1:]]
    for code, name in ((H.BF16, "bf16"), (H.F16, "f16")):
        for (B, Hh, S) in shapes:
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, 128)).astype(np.float32), code) for _ in range(4))
            t0 = time.time()
            new = run(code, q, k, v, go, False)
            old = run(code, q, k, v, go, True)
            msg = []
            for nm, a, b in zip(("dq", "dk", "dv"), new, old):
                fa, fb = f32(a, code), f32(b, code)
                d = np.abs(fa - fb)
                scale = np.abs(fb).max() + 1e-30
                msg.append(f"{nm} {d.max() / scale:.2e} (nan {int(np.isnan(fa).sum())})")
                if d.max() / scale > 3e-2 or np.isnan(fa).any():
                    grp = (np.nan_to_num(d, nan=9.0).max(axis=3)).reshape(B, Hh, S // 32, 32).max(axis=3) / scale
                    print(f"   {nm}: worst per 32-row group of (0,0): {np.round(grp[0, 0], 2).tolist()}")
            print(f"{name} B{B} H{Hh} S{S}: max|new - v4| / max|v4|: " + "  ".join(msg) + f"  ({time.time() - t0:.1f} s)")


if __name__ == "__main__":
    main()
