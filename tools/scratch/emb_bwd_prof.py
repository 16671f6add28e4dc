#!/usr/bin/env python3
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H
rng = np.random.default_rng(0)
def T(shape):
    return kfunca.from_numpy(rng.uniform(-1, 1, shape).astype(np.float32), 0).bfloat16()
V, D, n = 128256, 4096, 32768
tab = T((V, D)); tab.set_requires_grad(True)
idx = kfunca.from_numpy(rng.integers(0, V, size=(n,)).astype(np.int64), 0)
g = T((n, D))
def fb():
    y = kfunca.embedding(tab, idx)
    y.backward(g)
fb(); fb(); H.device_sync()
H.profile_reset(); H.profile_enable(True)
fb(); H.device_sync()
H.profile_enable(False)
for k, v in H.profile_results().items(): print(k, v)
