#!/usr/bin/env python3
"""How the 16-bit attention kernels hold the scale-aware bounds of oracle/checks.py as the logits grow: inputs N(0, s^2), s = 1, 2, 3
(logit std s^2) and U(-10, 10) - the range the reference's own attention test draws from (/root/reference/test/test_nn.py:22-24; logit
std 33: the softmax is one-hot for most rows). Output: fractions of the element / row / head bounds (1 = the bound), per output.
    python tools/attn_large_logits.py [--S 1024] > profiles/rNN_attn_large_logits.txt"""
import argparse
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tests"))
from kfunca_amd import hip_abi as H  # noqa: E402
from oracle import checks as K, oracle as O  # noqa: E402
from test_gpu_attention import fwd, bwd  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--S", type=int, default=1024)
ap.add_argument("--D", type=int, default=128)
ap.add_argument("--forms", default="default,scaled,v3v4")
args = ap.parse_args()
FORMS = {"default": ("default (exact)", {}), "scaled": ("scaled operands", {"KF_ATTN_SCALED_OPERANDS": "1"}),
         "v3v4": ("fwd v3 + dkv v4", {"KF_ATTN_FWD_V3": "1", "KF_ATTN_DKV_V4": "1"})}
H.set_device(0)
B, Hh, S, D = 1, 2, args.S, args.D
print(f"python tools/attn_large_logits.py --S {S} --D {D}   (B {B}, H {Hh}; fractions of the bounds of oracle/checks.py: element / row / head, 1 = the bound)")
for code, cn in ((H.BF16, "bf16"), (H.F16, "f16")):
    for dist in ("n1", "n2", "n3", "u10"):
        rng = np.random.default_rng({"n1": 10, "n2": 20, "n3": 30, "u10": 100}[dist] + code)
        if dist == "u10":
            draw = lambda: rng.uniform(-10, 10, (B, Hh, S, D))  # noqa: E731
            dn = "U(-10,10)"
        else:
            s = float(dist[1])
            draw = lambda: s * rng.standard_normal((B, Hh, S, D))  # noqa: E731
            dn = f"N(0,{s:.0f}^2) "
        q, k, v, go = (O.from_float(draw().astype(np.float32), code) for _ in range(4))
        ref = O.attn_ref64(q, k, v, go, code=code)
        fl = K.format_floor(q, k, v, go, code) if hasattr(K, "format_floor") else {}
        for key in args.forms.split(","):
            name, kn = FORMS[key]
            with H.knobs(**kn):
                o, lse = fwd(code, q, k, v)
                dq, dk, dv = bwd(code, q, k, v, o, lse, go)
            out = {}
            for n, g in (("o", o), ("dq", dq), ("dk", dk), ("dv", dv)):
                g64 = K.to_f64(g, code)
                if not np.isfinite(g64).all():
                    out[n] = "NONFINITE"
                    continue
                r, mag, quad, coh = K.scales(ref, n)
                kw = {"floor": fl[n]} if n in fl else {}
                m = K.margins(g64, r, mag, quad, K.EPS[code], K.ABS_ULP[code], coh, **kw)
                out[n] = f"{m['element']:.2f}/{m['row']:.2f}/{m['head']:.2f}"
            lse_err = np.abs(lse - ref["lse"]) / (1.0 + np.abs(ref["lse"]))
            print(f"{cn} inputs {dn} {name:16s} lse rel err {lse_err.max():.2e}  " + "  ".join(f"{n} {x}" for n, x in out.items()), flush=True)
