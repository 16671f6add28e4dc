#!/usr/bin/env python3
"""Run ONE program of the differential fuzz on ONE host, printing every step before it runs (a crash names its step). diff_fuzz_step.py SEED ref|mine"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle" / "_ref"))
from tests import test_gpu_host_diff_fuzz as F
seed, which = int(sys.argv[1]), sys.argv[2]
if which == "ref":
    import kfunca as KF
else:
    import kfunca_amd as KF
prog = F.make_program(1000 + seed, steps=28 + seed % 17)
for n in range(1, len(prog) + 1):
    print("prefix", n, F.origin_args(prog[n - 1]), flush=True)
    st, fin, side = F.run(KF, prog[:n])
    print("   ->", st[-1][:100], flush=True)
print("done")
