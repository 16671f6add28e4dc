import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle" / "_ref"))
import kfunca as REF
import kfunca_amd as MINE
rng = np.random.default_rng(3)
x = rng.permutation(5 * 6 * 7).reshape(5, 6, 7).astype(np.float32)
cases = [("x[2:3,1,3:4]", lambda t: t[2:3, 1, 3:4], lambda a: a[2:3, 1, 3:4], 0),
         ("x[1:4,2,::2]", lambda t: t[1:4, 2, ::2], lambda a: a[1:4, 2, ::2], 1),
         ("x[1:4,2,::2] dim0", lambda t: t[1:4, 2, ::2], lambda a: a[1:4, 2, ::2], 0),
         ("x.permute(2,0,1)", lambda t: t.permute(2, 0, 1), lambda a: a.transpose(2, 0, 1), 1),
         ("x[3]", lambda t: t[3], lambda a: a[3], 1),
         ("x[:, 2:5]", lambda t: t[:, 2:5], lambda a: a[:, 2:5], 1),
         ("x (whole)", lambda t: t, lambda a: a, 2)]
for name, fv, fn, dim in cases:
    want = np.sort(fn(x), axis=dim)
    wi = np.argsort(fn(x), axis=dim, kind="stable")
    for label, kf in (("reference host", REF), ("this host", MINE)):
        t = fv(kf.from_numpy(x, 0))
        try:
            v, i = t.sort(dim, False)
            v, i = v.contiguous().numpy(), i.contiguous().numpy()
            print(f"{name:22s} sort dim {dim} {label:15s}: values {'OK ' if np.array_equal(v, want) else 'WRONG'} positions {'OK' if np.array_equal(i, wi) else 'WRONG'}", v.reshape(-1)[:4], want.reshape(-1)[:4])
        except Exception as e:
            print(f"{name:22s} {label}: raised {str(e)[:120]}")
        try:
            s = t.sum(dim).contiguous().numpy()
            print(f"{name:22s} sum  dim {dim} {label:15s}: {'OK ' if np.allclose(s, fn(x).sum(axis=dim, keepdims=True)) else 'WRONG'}")
        except Exception as e:
            print(f"{name:22s} sum {label}: raised {str(e)[:120]}")
