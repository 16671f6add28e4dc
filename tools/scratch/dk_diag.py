#!/usr/bin/env python3
"""Which dK entries leave the bound on U(-10, 10) inputs at D = 64 (bf16)?"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent / "tests"))
from kfunca_amd import hip_abi as H  # noqa: E402
from oracle import checks as K, oracle as O  # noqa: E402
from test_gpu_attention import fwd, bwd  # noqa: E402
H.set_device(0)
code, D = H.BF16, int(sys.argv[1]) if len(sys.argv) > 1 else 64
rng = np.random.default_rng(100 + code)
q, k, v, go = (O.from_float(rng.uniform(-10, 10, (1, 2, 1024, D)).astype(np.float32), code) for _ in range(4))
ref = O.attn_ref64(q, k, v, go, code=code)
o, lse = fwd(code, q, k, v)
dq, dk, dv = bwd(code, q, k, v, o, lse, go)
fl = K.format_floor(q, k, v, go, code)
got = K.to_f64(dk, code)
r, mag, quad, _ = K.scales(ref, "dk")
err = np.abs(got - r)
bound = K.EPS[code] * (K.C_OUT * np.abs(r) + K.C_SUM * mag) + fl["dk"] + K.ABS_ULP[code]
ratio = err / bound
idx = np.unravel_index(np.argsort(ratio.ravel())[-8:], ratio.shape)
for t in zip(*idx):
    print(t, "ratio %.2f got %.4e ref %.4e mag %.4e floor %.2e" % (ratio[t], got[t], r[t], mag[t], fl["dk"][t]))
rows = np.linalg.norm(got - r, axis=-1) / (K.EPS[code] * (K.C_ROW * np.linalg.norm(r, axis=-1) + K.C_Q * np.linalg.norm(quad, axis=-1)) + np.linalg.norm(fl["dk"], axis=-1))
ri = np.unravel_index(np.argsort(rows.ravel())[-5:], rows.shape)
for t in zip(*ri):
    print("row", t, "ratio %.2f |got| %.3e |ref| %.3e |quad| %.3e" % (rows[t], np.linalg.norm(got[t]), np.linalg.norm(r[t]), np.linalg.norm(quad[t])))
    j = t[2]
    # the attention this key receives: its largest weights
    qf, kf = K.to_f64(q, code)[t[0], t[1]], K.to_f64(k, code)[t[0], t[1]]
    s = qf @ kf[j] / np.sqrt(D)
    p = np.exp(s - ref["lse"][t[0], t[1]])
    p[:j] = 0
    top = np.argsort(p)[-3:]
    print("    largest P_ij over queries i:", [(int(i), float(p[i])) for i in top])
