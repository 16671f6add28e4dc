"""-m gpu: the caching allocator against the reference's own, differentially: the reference's host half (oracle/_ref: device_allocator.cpp
unmodified, dmalloc / dfree behind docs/seam.cpp) and this host run the SAME random sequence of allocations and frees, each in a fresh child process; what is
compared is the REUSE PATTERN - for every allocation, which earlier allocation's device pointer it received (or none: a driver allocation). Equal patterns mean the same
size classes (eight pools, device_allocator.h:48-57), the same 1 KiB rounding, the same smallest-block-that-fits choice inside a pool (device_allocator.cpp:38-66) and
the same quirk at the class boundaries (a request is looked up in the pool of its RAW size, a freed block is filed under its ROUNDED size)."""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

ROOT = Path(__file__).resolve().parent.parent
REFDIR = ROOT / "oracle" / "_ref"
pytestmark = pytest.mark.gpu
SIZES = [1, 100, 1023, 1024, 1025, 4095, 4096, 4097, 5000, 65535, 65536, 65537, 70000, 262143, 262144, 262145, 300000, 1 << 20, (1 << 20) + 1, 3 << 20, 16 << 20, (16 << 20) + 1,
         40 << 20, 64 << 20, (64 << 20) + 1, 100 << 20, 256 << 20, (256 << 20) + 1, 300 << 20]


def make_sequence(seed, steps=120):
    rng = np.random.default_rng(seed)
    seq, live = [], []
    for _ in range(steps):
        if live and rng.random() < 0.45:
            k = int(rng.integers(0, len(live)))
            seq.append(("free", live.pop(k)))
        else:
            n = int(rng.choice(SIZES)) if rng.random() < 0.8 else int(rng.integers(1, 2 << 20))
            seq.append(("alloc", n))
            live.append(sum(1 for s in seq if s[0] == "alloc") - 1)
    return seq


def child(which, first, last):
    sys.path.insert(0, str(ROOT))
    if which == "ref":
        sys.path.insert(0, str(REFDIR))
        import kfunca as kf
    else:
        import kfunca_amd as kf
    out = {}
    for seed in range(first, last):
        tensors, ptrs, pattern = {}, [], []
        for op, arg in make_sequence(seed):
            if op == "alloc":
                t = kf.empty([arg], kf.byte, 0)
                p = t.data_ptr()
                tensors[len(ptrs)] = t
                pattern.append(max((i for i, q in enumerate(ptrs) if q == p), default=-1))
                ptrs.append(p)
            else:
                del tensors[arg]
        tensors.clear()
        out[seed] = pattern
    print(json.dumps(out))


def test_same_reuse_pattern_as_the_reference_allocator():
    if not list(REFDIR.glob("kfunca*.so")):
        pytest.skip("oracle/_ref/kfunca*.so not built (build container only)")
    n = int(os.environ.get("KF_ALLOC_DIFF_SEEDS", "40"))
    res = {}
    for which in ("ref", "mine"):
        r = subprocess.run([sys.executable, str(Path(__file__).resolve()), which, "0", str(n)], capture_output=True, text=True, timeout=1200, cwd=str(ROOT))
        assert r.returncode == 0, (which, r.returncode, r.stderr[-1500:])
        res[which] = json.loads(r.stdout.strip().splitlines()[-1])
    for seed in range(n):
        a, b = res["ref"][str(seed)], res["mine"][str(seed)]
        assert a == b, f"sequence {seed}: the reference allocator and this one hand out different blocks from allocation {next(i for i, (x, y) in enumerate(zip(a, b)) if x != y)} on: {a} vs {b}; {make_sequence(seed)}"
    assert any(x >= 0 for s in res["ref"].values() for x in s)   # (the sequences do hit the cache)


if __name__ == "__main__":
    child(sys.argv[1], int(sys.argv[2]), int(sys.argv[3]))
