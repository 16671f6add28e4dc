import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from kfunca_amd import hip_abi as H
H.set_device(0)
for (M, N, K) in ((256, 4096, 16384), (128, 8192, 8192), (512, 512, 32768), (1024, 1024, 1024), (1024, 1024, 4096), (256, 256, 8192)):
    a, b, c = H.DevBuf(M*K*2), H.DevBuf(K*N*2), H.DevBuf(M*N*2)
    a.zero(); b.zero()
    need = H.gemm_workspace_bytes(H.BF16, 0, 1, M, N, K)
    ws = H.DevBuf(max(need, 16))
    res = {}
    for tag, wsp, wsb in (("split", ws.ptr, need), ("plain", None, 0)):
        for _ in range(3):
            H.gemm(H.BF16, 0, 1, M, N, K, 1.0, a.ptr, K, b.ptr, K, 0.0, c.ptr, N, 0, None, wsp, wsb)
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(20):
            H.gemm(H.BF16, 0, 1, M, N, K, 1.0, a.ptr, K, b.ptr, K, 0.0, c.ptr, N, 0, None, wsp, wsb)
        e1.record(None); e1.sync()
        res[tag] = e0.elapsed_ms(e1) / 20
    print(M, N, K, "slices", need // (M*N*4) if need else 1, {k: round(v*1e3, 1) for k, v in res.items()}, "us; TF", {k: round(2*M*N*K/(v*1e-3)/1e12, 1) for k, v in res.items()})
