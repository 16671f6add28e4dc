#!/usr/bin/env python3
"""permute(1,0).contiguous() timing at a few shapes: perm_case.py"""
import sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
H.set_device(0)
for code, es, name in ((H.F32, 4, "f32"), (H.BF16, 2, "bf16")):
    for n0, n1 in ((16384, 16384), (8192, 32768), (4096, 4096), (16384, 4096)):
        a, b = H.DevBuf(es * n0 * n1), H.DevBuf(es * n0 * n1)
        d = H.make_desc([H.View(b.ptr, (n0, n1), (n1, 1), code)], [H.View(a.ptr, (n0, n1), (1, n0), code)])
        def fn(): H.elementwise(H.EW_COPY, d)
        for _ in range(3): fn()
        H.device_sync()
        e0, e1 = H.Event(), H.Event()
        e0.record(None)
        for _ in range(20): fn()
        e1.record(None); H.device_sync()
        ms = e0.elapsed_ms(e1) / 20
        print(f"{name} [{n0},{n1}] {ms:.4f} ms {2 * es * n0 * n1 / ms / 1e6:8.1f} GB/s", flush=True)
