#!/usr/bin/env python3
"""rms / layer norm forward + backward at short and odd row lengths: bytes / time."""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
tot = 1 << 28  # elements
a, b, c = H.DevBuf(2 * tot), H.DevBuf(2 * tot), H.DevBuf(2 * tot)
for cols in (64, 128, 256, 512, 1024, 2048, 5120, 12288):
    rows = tot // cols
    wbuf, dwb, dbb = H.DevBuf(2 * cols), H.DevBuf(2 * cols), H.DevBuf(2 * cols)
    mean, rstd = H.DevBuf(4 * rows), H.DevBuf(4 * rows)
    for kind, kname in ((H.NORM_RMS, "rms"), (H.NORM_LAYER, "layer")):
        out = []
        for which in ("fwd", "bwd"):
            keep = []
            if which == "fwd":
                fn = lambda: H.norm_fwd(kind, H.BF16, rows, cols, a.ptr, wbuf.ptr, None, 1e-5, c.ptr, mean.ptr, rstd.ptr)
                nb = 4 * rows * cols
            else:
                fn = lambda: keep.append(H.norm_bwd(kind, H.BF16, rows, cols, a.ptr, wbuf.ptr, mean.ptr, rstd.ptr, b.ptr, c.ptr, dwb.ptr, dbb.ptr if kind == H.NORM_LAYER else None))
                nb = 6 * rows * cols
            for _ in range(2): fn()
            H.device_sync()
            e0, e1 = H.Event(), H.Event()
            e0.record(None)
            for _ in range(5): fn()
            e1.record(None); H.device_sync()
            ms = e0.elapsed_ms(e1) / 5
            out.append(f"{which} {ms:7.4f} ms {nb / ms / 1e6:7.0f} GB/s")
        print(f"bf16 [{rows}, {cols}] {kname:5s} " + " | ".join(out), flush=True)
