import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parents[2]))
from kfunca_amd import hip_abi as H
from oracle import checks as K, oracle as O
from tests.test_gpu_attention import bwd, fwd
for (B, Hh, Sq, Skv) in ((1, 3, 768, 768), (1, 2, 1024, 1024)):
    rng = np.random.default_rng(55 + Sq + Skv)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), H.BF16) for s in ((B, Hh, Sq, 128), (B, Hh, Skv, 128), (B, Hh, Skv, 128), (B, Hh, Sq, 128)))
    o, lse = fwd(H.BF16, q, k, v)
    g = bwd(H.BF16, q, k, v, o, lse, go)
    ref = O.attn_ref64(q, k, v, go, code=O.BF16)
    for name, got in (("dk", g[1]), ("dv", g[2]), ("dq", g[0])):
        gf = K.to_f64(got, O.BF16)
        nerr = np.linalg.norm(gf - ref[name], axis=-1); nref = np.linalg.norm(ref[name], axis=-1); nmag = np.linalg.norm(ref["m" + name], axis=-1)
        ratio = nerr / (K.EPS[O.BF16] * (K.C_ROW * nref + K.C_FLOOR * nmag) + 1e-300)
        idx = np.argsort(ratio.reshape(-1))[-6:]
        print(name, Sq, [(int(i // Sq), int(i % Sq), round(float(ratio.reshape(-1)[i]), 2), float(nerr.reshape(-1)[i]), float(nref.reshape(-1)[i]), float(nmag.reshape(-1)[i])) for i in idx])
