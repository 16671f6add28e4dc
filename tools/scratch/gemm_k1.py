import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
H.set_device(0)
rng = np.random.default_rng(5)
for code in (H.F16, H.BF16, H.F32):
    for (M, N, K, ta, tb) in ((587, 275, 1, True, False), (587, 275, 1, False, False), (64, 64, 1, False, True), (300, 200, 2, True, True)):
        a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
        b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
        sa, sb = (np.ascontiguousarray(a.T) if ta else a), (np.ascontiguousarray(b.T) if tb else b)
        da, db, dc = H.DevBuf.from_numpy(sa), H.DevBuf.from_numpy(sb), H.DevBuf(M * N * a.itemsize)
        need = H.gemm_workspace_bytes(code, ta, tb, M, N, K)
        ws = H.DevBuf(max(need, 16))
        H.gemm(code, ta, tb, M, N, K, 1.0, da.ptr, sa.shape[1], db.ptr, sb.shape[1], 0.0, dc.ptr, N, 0, None, ws.ptr if need else None, need)
        H.device_sync()
        got = O.to_float(dc.to_numpy((M, N), a.dtype), code).astype(np.float64)
        fa, fb = O.to_float(a, code).astype(np.float64), O.to_float(b, code).astype(np.float64)
        ref = fa @ fb
        err = np.abs(got - ref)
        i = np.unravel_index(np.argmax(err / (np.abs(ref) + 1e-300)), err.shape)
        want = O.to_float(O.from_float(ref.astype(np.float32), code), code).astype(np.float64)   # the product rounded once to the output format
        print(code, M, N, K, ta, tb, "max abs err %.3e" % err.max(), "worst rel at", i, "ref %.6e got %.6e" % (ref[i], got[i]),
              "equal to the once-rounded product:", bool((got == want).all()), "mismatches", int((got != want).sum()), flush=True)
