import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
rng = np.random.default_rng(1)
for n in (1, 2, 3, 4, 5, 8, 13, 16, 17, 32, 33, 64):
    for nseg in (1, 7, 300):
        for dt in (np.float32, np.int64):
            keys = rng.integers(-50, 50, size=(nseg, n)).astype(dt)
            for desc in (False, True):
                k, p = H.sort_segments(keys, desc)
                o = np.argsort(-keys if desc else keys, axis=1, kind="stable")
                ok = np.array_equal(p, o) and np.array_equal(k, np.take_along_axis(keys, o, 1))
                if not ok:
                    bad = np.argwhere(p != o)
                    print("FAIL", n, nseg, dt.__name__, desc, "first bad", bad[:3].tolist(), "got", p[bad[0][0]][:16], "want", o[bad[0][0]][:16])
print("done")
