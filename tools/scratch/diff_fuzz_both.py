#!/usr/bin/env python3
"""Both hosts in one process, seeds FIRST..LAST (the test's own order: reference host, then this host), progress printed. diff_fuzz_both.py FIRST LAST"""
import sys
from pathlib import Path
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT)); sys.path.insert(0, str(ROOT / "oracle" / "_ref"))
from tests import test_gpu_host_diff_fuzz as F
import os
if os.environ.get('MINE_FIRST'):
    import kfunca_amd as MINE
    import kfunca as REF
else:
    import kfunca as REF
    import kfunca_amd as MINE
a, b = int(sys.argv[1]), int(sys.argv[2])
for seed in range(a, b):
    print("seed", seed, flush=True)
    prog = F.make_program(1000 + seed, steps=28 + seed % 17)
    print("  ref", flush=True); F.run(REF, prog)
    print("  mine", flush=True); F.run(MINE, prog)
print("done")
