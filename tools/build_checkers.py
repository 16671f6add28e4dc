#!/usr/bin/env python3
"""Builds the CHECKERS (test infrastructure; never loaded by kfunca_amd/):

    oracle/liboracle.so    gcc on oracle/oracle.c - the CPU restatement of the reference's algorithms
    oracle/_ref/           the reference's own host half (src/core/*.cpp + src/register.cpp where they lie under /root/reference) linked over
                           this repository's device library: oracle/build_ref_host.py. Build container only - the GPU box has no
                           /root/reference and uses the prebuilt files.

`__graft_entry__.build()` calls build_all() after the product build (building a checker is not using it). Rebuild decisions are made on
content signatures (kfunca_amd/_build.py: _fresh / _mark), not on mtimes.
"""
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
if str(ROOT) not in sys.path:
    sys.path.insert(0, str(ROOT))

ORACLE_LIB = ROOT / "oracle" / "liboracle.so"


def build_oracle(force: bool = False) -> Path:
    from kfunca_amd import _build
    from oracle import oracle as O
    src, hdr = ROOT / "oracle" / "oracle.c", ROOT / "oracle" / "oracle.h"
    # oracle.py owns the flags: a fixed ISA baseline + no implicit fma contraction, so the checker behaves the same on the GPU box's host CPU
    if force or not _build._fresh(ORACLE_LIB, [src, hdr], O.CFLAGS):
        _build._run(["gcc", *O.CFLAGS, "-o", ORACLE_LIB, src, "-lm"])
        _build._mark(ORACLE_LIB, [src, hdr], O.CFLAGS)
    return ORACLE_LIB


def build_ref_host() -> bool:
    """True when oracle/_ref was (re)built; False without the reference mount. A failure is reported, not raised: the tests that need the
    module (tests/test_seam_links.py, tests/test_gpu_reference_host.py) say so themselves."""
    if not Path("/root/reference/src/core/tensor.cpp").exists():
        return False
    try:
        from oracle import build_ref_host as R
        R.build(with_module=True)
        return True
    except Exception as e:  # noqa: BLE001
        print(f"[tools/build_checkers] oracle/_ref NOT built (tests/test_seam_links.py, test_gpu_reference_host.py will fail): {e}", file=sys.stderr)
        return False


def build_all(force: bool = False) -> None:
    build_oracle(force)
    build_ref_host()


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", ORACLE_LIB, "and oracle/_ref" if (ROOT / "oracle" / "_ref").exists() else "")
