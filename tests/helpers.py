"""Shared test helpers: golden fixtures, seeded regeneration, device upload/download."""
import hashlib
from pathlib import Path

import numpy as np

GOLDEN = Path(__file__).resolve().parent / "golden"


def golden(name):
    return np.load(GOLDEN / f"{name}.npz")


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def uni(rng, shape, dtype=np.float32, lo=-10, hi=10):
    return rng.uniform(lo, hi, size=shape).astype(dtype)


def regen(seed, shapes, expect_sha, lo=-10, hi=10, dtype=np.float32):
    """Re-draw inputs exactly as tests/golden/gen_golden.py did and verify their hash."""
    rng = np.random.default_rng(int(seed))
    arrs = [rng.uniform(lo, hi, size=s) if dtype is np.float64 else uni(rng, s, dtype, lo, hi) for s in shapes]
    assert np.array_equal(sha(*arrs), expect_sha), "RNG stream differs from the one the fixture was made with"
    return arrs


def assert_close(got, want, rtol=1e-3, atol=1e-3, what=""):
    """The reference's assert_allclose (test/common.py:6-11): rtol = atol = 1e-3 by default."""
    got = np.asarray(got, dtype=np.float64)
    want = np.asarray(want, dtype=np.float64)
    assert got.shape == want.shape, (what, got.shape, want.shape)
    bad = ~(np.abs(got - want) <= atol + rtol * np.abs(want))
    if bad.any():
        i = np.unravel_index(np.argmax(np.abs(got - want) * bad), got.shape)
        raise AssertionError(f"{what}: {bad.sum()} / {bad.size} outside rtol={rtol} atol={atol}; "
                             f"worst at {i}: got {got[i]} want {want[i]}")
