import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.test_gpu_attention import fwd, bwd, f
code = H.BF16
for (B, Hh, Sq, Skv) in ((1, 1, 512, 128), (1, 1, 128, 128)):
    D = 128
    rng = np.random.default_rng(Sq + Skv + code + D)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    o, lse = fwd(code, q, k, v)
    want = O.attn_bwd(q, k, v, go, code=code)
    got = bwd(code, q, k, v, o, lse, go)
    for nme, g_, w_ in zip(("dq", "dk", "dv"), got, want):
        gf, wf = f(g_, code), f(w_, code)
        bad = ~np.isfinite(gf) | (np.abs(gf - wf) > 3e-2 + 2e-2 * np.abs(wf))
        rows = sorted(set(int(r[2]) for r in np.argwhere(bad)))
        print(Sq, Skv, nme, "bad", int(bad.sum()), "nan", int(np.isnan(gf).sum()), "inf", int(np.isinf(gf).sum()), "rows", rows[:40])
    if Sq == 512:
        gq = f(got[0], code)[0, 0]
        for r in (56, 60, 120, 184, 188, 248, 312, 440, 504):
            print("  dq row", r, gq[r, :6], "want", f(want[0], code)[0, 0, r, :3])
