import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.gpu_util import Dev, gpu_binary, gpu_copy
H.set_device(0)
rng = np.random.default_rng(0)
code = H.BF16
def rnd(shape): return O.f32_to_bf16(rng.uniform(-1, 1, shape).astype(np.float32))
for (oa, ob, oo) in ((0, 0, 0), (1, 0, 0), (0, 0, 1), (1, 3, 0), (2, 2, 2), (1, 3, 5), (0, 0, 2), (0, 0, 4)):
    a, b, ob_ = rnd((64, 530)), rnd((64, 530)), rnd((64, 530))
    va, vb, vo = a[:, oa:oa + 512], b[:, ob:ob + 512], ob_[:, oo:oo + 512]
    want = ob_.copy()
    dout = Dev(vo, code, base=ob_)
    got = gpu_binary(H.EW_ADD, Dev(va, code, base=a), Dev(vb, code, base=b), out=dout)
    want[:, oo:oo + 512] = O.binary(O.ADD, np.ascontiguousarray(va), np.ascontiguousarray(vb), a_code=code, b_code=code)
    host = got.buf.to_numpy(ob_.shape, ob_.dtype)
    bad = np.argwhere(host != want)
    print((oa, ob, oo), "ok" if len(bad) == 0 else f"BAD {len(bad)} first {bad[:4].tolist()}")
