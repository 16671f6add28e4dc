#!/usr/bin/env python3
"""Is the dQ kernel's per-process bimodality (0.90 / 0.95 ms inside bench.py's step) a property of where the buffers lie?  One process per run:
MODE=plain (bench.py's own allocations) | arena (ONE allocation, 2 MiB-aligned sub-buffers) | wsfirst (the dS workspace allocated before everything else)
| pad (a 1 GiB allocation in front, freed again, before the rest).  Prints the step and the dQ / dK/dV averages and the buffers' addresses mod 1 GiB."""
import os
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import bench  # noqa: E402
from kfunca_amd import hip_abi as H  # noqa: E402

mode = os.environ.get("MODE", "plain")
H.set_device(0)
Real = H.DevBuf
addrs = []
if mode.startswith("arena"):
    # arena | arena:ALIGN_MB:PAD_MB[:wsfirst]  (every sub-buffer aligned to ALIGN_MB, PAD_MB left free behind each)
    f = mode.split(":")
    align = (int(f[1]) if len(f) > 1 else 2) << 20
    padb = (int(f[2]) if len(f) > 2 else 0) << 20
    arena = Real(40 << 30)
    top = [(arena.ptr + align - 1) // align * align]
    ws_bytes = H.attn_bwd_workspace_bytes(H.BF16, bench.AB, bench.AH, bench.AS, bench.AS, bench.AD)
    ws_slot = None
    if len(f) > 3 and f[3] == "wsfirst":
        ws_slot = top[0]
        top[0] = (top[0] + ws_bytes + padb + align - 1) // align * align

    class Sub(Real):
        def __init__(self, nbytes):
            self.nbytes = int(nbytes)
            if ws_slot is not None and self.nbytes == ws_bytes:
                self.ptr = ws_slot
                return
            self.ptr = top[0]
            top[0] = (top[0] + self.nbytes + padb + align - 1) // align * align
            assert top[0] <= arena.ptr + arena.nbytes

        def free(self):
            self.ptr = None
    H.DevBuf = Sub
elif mode == "wsfirst":
    ws_bytes = H.attn_bwd_workspace_bytes(H.BF16, bench.AB, bench.AH, bench.AS, bench.AS, bench.AD)
    ws = Real(ws_bytes)

    class Pick(Real):
        def __init__(self, nbytes):
            if int(nbytes) == ws_bytes:
                self.nbytes, self.ptr = ws.nbytes, ws.ptr
            else:
                Real.__init__(self, nbytes)
    H.DevBuf = Pick
elif mode == "pad":
    pad = Real(int(os.environ.get("PAD_MB", "1024")) << 20)
wl = bench.Workload(H, 0, 1)
st = H.Stream()
for _ in range(5):
    wl.step(st.handle)
H.device_sync()
H.profile_reset()
H.profile_enable(True)
import time
t0 = time.perf_counter()
for _ in range(30):
    wl.step(st.handle)
H.device_sync()
dt = (time.perf_counter() - t0) / 30 * 1e3
H.profile_enable(False)
prof = H.profile_results()
k = {n: ms / max(c, 1) for n, (ms, c) in prof.items()}
print(mode, "step %.3f" % dt, " ".join("%s %.3f" % (n.replace("attn_", "").replace("_mfma", ""), v) for n, v in k.items()),
      "ws@%x dq@%x k@%x" % (wl.aws.ptr % (1 << 30), wl.dq.ptr % (1 << 30), wl.k.ptr % (1 << 30)), flush=True)
