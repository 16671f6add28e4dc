import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from kfunca_amd import hip_abi as H
H.set_device(0)
n = 4096
rng = np.random.default_rng(0)
x = rng.uniform(-1, 1, size=(n, n)).astype(np.float32).view(np.uint32)
bits = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
A, W, G = H.DevBuf.from_numpy(bits), H.DevBuf.from_numpy(bits[::-1].copy()), H.DevBuf.from_numpy(bits.T.copy())
dA, dW = H.DevBuf(2*n*n), H.DevBuf(2*n*n)
def sep():
    H.gemm(H.BF16, 0, 1, n, n, n, 1.0, G.ptr, n, W.ptr, n, 0.0, dA.ptr, n)
    H.gemm(H.BF16, 1, 0, n, n, n, 1.0, A.ptr, n, G.ptr, n, 0.0, dW.ptr, n)
def grp():
    H.gemm_grouped(H.BF16, [(0, 1, n, n, n, 1.0, 0.0, G.ptr, n, W.ptr, n, dA.ptr, n), (1, 0, n, n, n, 1.0, 0.0, A.ptr, n, G.ptr, n, dW.ptr, n)])
for rep in range(3):
    for name, fn in (("separate", sep), ("grouped", grp)):
        for _ in range(20): fn()
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(200): fn()
        e1.record(None); e1.sync()
        ms = e0.elapsed_ms(e1) / 200
        print(name, round(ms*1e3, 1), "us per pair", round(4*n**3/(ms*1e-3)/1e12), "TF/s")
