#!/usr/bin/env python3
"""One sort case a few times (the program behind rocprofv3 --pmc for the radix kernels): sort_case.py NSEG N [ROUNDS]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402
nseg, n = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
H.set_device(0)
rng = np.random.default_rng(0)
keys = rng.standard_normal(nseg * n).astype(np.float32)
a, b, c = H.DevBuf.from_numpy(keys), H.DevBuf(keys.nbytes), H.DevBuf(8 * keys.size)
need = H.lib().kf_sort_workspace_bytes(H.F32, nseg, n)
ws = H.DevBuf(max(need, 16))
for _ in range(rounds):
    H.check(H.lib().kf_sort(H.F32, a.ptr, b.ptr, c.ptr, nseg, n, 0, ws.ptr, need, None))
H.device_sync()
