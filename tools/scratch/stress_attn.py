#!/usr/bin/env python3
"""Randomised stress of the attention entries at the C ABI on ragged shapes (round 6): random B, H, Sq <= Skv, head size 64 | 128, bf16 | f16,
forward + backward against the double-precision oracle under the suite's scale-aware bounds, guard bands behind every output, the matrix-core labels
asserted. stress_attn.py SEED SECONDS"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import checks as K
from oracle import oracle as O
H.set_device(0)
seed, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed)
t_end = time.time() + secs
GUARD = 16384
def guarded(nbytes):
    b = H.DevBuf(nbytes + GUARD)
    f = np.full(GUARD, 0xAB, dtype=np.uint8)
    H.check(H.lib().kf_memcpy_h2d(b.ptr + nbytes, f.ctypes.data, GUARD, None))
    return b
def guard_ok(b, nbytes):
    t = np.empty(GUARD, dtype=np.uint8)
    H.check(H.lib().kf_memcpy_d2h(t.ctypes.data, b.ptr + nbytes, GUARD, None))
    return bool((t == 0xAB).all())
n = 0
worst = {}
while time.time() < t_end:
    code = int(rng.choice([H.BF16, H.F16]))
    D = int(rng.choice([64, 128]))
    B, Hh = int(rng.integers(1, 3)), int(rng.integers(1, 4))
    Sq = int(rng.choice([1, 2, 31, 32, 33, 63, 64, 65, 127, 129, 200, 255, 256, 257, 300, 511, 513, 700, 777, 1023, 1025]))
    Skv = Sq + int(rng.choice([0, 0, 0, 1, 7, 31, 64, 100, 255, 256, 300]))
    scale_in = float(rng.choice([1.0, 1.0, 3.0]))
    q, k, v, go = (O.from_float((scale_in * rng.uniform(-1, 1, s)).astype(np.float32), code)
                   for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    bq, bk, bv, bgo = (H.DevBuf.from_numpy(x) for x in (q, k, v, go))
    bo, bdq, bdk, bdv = guarded(q.nbytes), guarded(q.nbytes), guarded(k.nbytes), guarded(k.nbytes)
    blse = guarded(4 * B * Hh * Sq)
    need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, D)
    ws = guarded(need)
    H.profile_reset(); H.profile_enable(True)
    H.attn_fwd(code, B, Hh, Sq, Skv, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, blse.ptr)
    H.attn_bwd(code, B, Hh, Sq, Skv, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, blse.ptr, bgo.ptr, bdq.ptr, bdk.ptr, bdv.ptr, ws.ptr, need)
    H.device_sync(); H.profile_enable(False)
    sfx = "_d64" if D == 64 else ""
    case = (code, D, B, Hh, Sq, Skv, scale_in)
    assert {"attn_fwd_mfma" + sfx, "attn_bwd_dkv_mfma" + sfx, "attn_bwd_dq_mfma" + sfx} <= set(H.profile_results()), (case, sorted(H.profile_results()))
    for name, buf, nb in (("o", bo, q.nbytes), ("dq", bdq, q.nbytes), ("dk", bdk, k.nbytes), ("dv", bdv, k.nbytes), ("lse", blse, 4 * B * Hh * Sq), ("ws", ws, need)):
        assert guard_ok(buf, nb), (case, name)
    m = K.attn_check(q, k, v, code, o=bo.to_numpy(q.shape, q.dtype), lse=blse.to_numpy((B, Hh, Sq), np.float32), d_o=go,
                     dq=bdq.to_numpy(q.shape, q.dtype), dk=bdk.to_numpy(k.shape, k.dtype), dv=bdv.to_numpy(k.shape, k.dtype), what=str(case))
    for nm, mm in m.items():
        w = max(mm.get("element", 0), mm.get("row", 0), mm.get("head", 0), mm.get("fraction_of_bound", 0))
        worst[nm] = max(worst.get(nm, 0.0), w)
    n += 1
print(f"seed {seed}: {n} ragged attention cases, all within the bounds; worst fraction of a bound per output: " + ", ".join(f"{k} {v:.2f}" for k, v in worst.items()))
