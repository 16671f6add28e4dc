#!/usr/bin/env python3
"""Run programs of the differential fuzz on ONE host only (which host corrupts memory?). diff_fuzz_one_host.py ref|mine FIRST LAST"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle" / "_ref"))
from tests import test_gpu_host_diff_fuzz as F
which, a, b = sys.argv[1], int(sys.argv[2]), int(sys.argv[3])
if which == "ref":
    import kfunca as KF
else:
    import kfunca_amd as KF
for seed in range(a, b):
    if seed % 50 == 0: print("seed", seed, flush=True)
    F.run(KF, F.make_program(1000 + seed, steps=28 + seed % 17))
print("done", which, a, b)
