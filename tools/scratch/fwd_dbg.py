#!/usr/bin/env python3
"""Debug: structured inputs through the generated forward (scores 0 or a constant, v = 1 or the key index)."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402
from oracle import oracle as O  # noqa: E402

H.set_device(0)
S, D = int(sys.argv[1]) if len(sys.argv) > 1 else 256, 128
for case in ("zero", "const", "rand"):
    rng = np.random.default_rng(1)
    if case == "zero":
        q = np.zeros((1, 1, S, D), np.float32); k = np.zeros_like(q)
    elif case == "const":
        q = np.full((1, 1, S, D), 0.5, np.float32); k = np.full_like(q, 0.25)
    else:
        q = rng.uniform(-1, 1, (1, 1, S, D)).astype(np.float32); k = rng.uniform(-1, 1, (1, 1, S, D)).astype(np.float32)
    v = np.ones((1, 1, S, D), np.float32)
    v[0, 0, :, 1] = np.arange(S) / 64.0
    qb, kb, vb = (O.from_float(x, H.BF16) for x in (q, k, v))
    bq, bk, bv = (H.DevBuf.from_numpy(x) for x in (qb, kb, vb))
    bo, bl = H.DevBuf(qb.nbytes), H.DevBuf(4 * S)
    H.attn_fwd(H.BF16, 1, 1, S, S, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, bl.ptr)
    H.device_sync()
    o = O.to_float(bo.to_numpy(qb.shape, np.uint16), H.BF16)[0, 0]
    l = bl.to_numpy((S,), np.float32)
    print(case)
    for r in list(range(0, 4)) + list(range(30, 36)) + list(range(62, 66)) + [S - 1]:
        print(f"  row {r:4d}: O[:,0] {o[r, 0]:+.4f} O[:,1] {o[r, 1]:+.4f} O[:,127] {o[r, 127]:+.4f}  lse {l[r]:+.5f}  (expect lse {np.log(r + 1):.5f} for equal scores)")
    bad = np.where(~np.isfinite(l) | (np.abs(o[:, 0] - 1) > 1e-2))[0]
    print("  bad rows:", bad.tolist())
