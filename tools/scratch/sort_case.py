#!/usr/bin/env python3
"""One sort case a few times (the program behind rocprofv3 for the radix kernels): sort_case.py NSEG N [ROUNDS] [f32|f64|i32|i64|u8small]"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402
nseg, n = int(sys.argv[1]), int(sys.argv[2])
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 4
kind = sys.argv[4] if len(sys.argv) > 4 else "f32"
H.set_device(0)
rng = np.random.default_rng(0)
if kind == "f32":
    keys, code = rng.standard_normal(nseg * n).astype(np.float32), H.F32
elif kind == "f64":
    keys, code = rng.standard_normal(nseg * n), H.F64
elif kind == "i64":
    keys, code = rng.integers(-2**62, 2**62, nseg * n, dtype=np.int64), H.I64
elif kind == "i32":
    keys, code = rng.integers(-2**31, 2**31 - 1, nseg * n, dtype=np.int32), H.I32
else:  # small non-negative int32 values: the upper three bytes of every key are the same
    keys, code = rng.integers(0, 200, nseg * n, dtype=np.int32), H.I32
a, b, c = H.DevBuf.from_numpy(keys), H.DevBuf(keys.nbytes), H.DevBuf(8 * keys.size)
need = H.lib().kf_sort_workspace_bytes(code, nseg, n)
ws = H.DevBuf(max(need, 16))
for _ in range(rounds):
    H.check(H.lib().kf_sort(code, a.ptr, b.ptr, c.ptr, nseg, n, 0, ws.ptr, need, None))
H.device_sync()
