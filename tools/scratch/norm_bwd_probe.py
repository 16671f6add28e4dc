import sys; sys.path.insert(0, "/root/repo")
import numpy as np
from kfunca_amd import hip_abi as H
H.set_device(0)
for rows, cols, code in ((1<<16, 8192, H.BF16), (1<<16, 4096, H.F32), (1<<18, 1024, H.BF16)):
    es = 2 if code == H.BF16 else 4
    a, b, c = (H.DevBuf(rows*cols*es) for _ in range(3))
    w, mean, rstd, dw, db = H.DevBuf(cols*es), H.DevBuf(4*rows), H.DevBuf(4*rows), H.DevBuf(cols*es), H.DevBuf(cols*es)
    for buf in (a, b, w, mean, rstd): buf.zero()
    keep = []
    for it in range(3):
        H.profile_reset(); H.profile_enable(True)
        for _ in range(5):
            keep.append(H.norm_bwd(H.NORM_RMS, code, rows, cols, a.ptr, w.ptr, mean.ptr, rstd.ptr, b.ptr, c.ptr, dw.ptr, None))
        H.device_sync(); H.profile_enable(False)
    r = H.profile_results()
    print(rows, cols, code, {k: round(v[0]/v[1], 4) for k, v in r.items()}, "GB/s bwd only", round(3*es*rows*cols / (r["norm_bwd"][0]/r["norm_bwd"][1]*1e-3) / 1e9))
