"""CPU: rebuild decisions are made on CONTENT hashes, and the device library says which sources it was linked from (VERDICT round 4,
weak #7: staleness was mtime-based and `device_src_sha` on the bench line hashed the sources on the box, not the binary that ran)."""
import ctypes
import os
import time

import bench
from kfunca_amd import _build


def test_signature_follows_content_not_mtime(tmp_path):
    src, inc, obj = tmp_path / "k.hip", tmp_path / "k.inc", tmp_path / "k.o"
    src.write_text("#include \"k.inc\"\n")
    inc.write_text("#define X 1\n")
    flags = ["-O3", f"-I{_build.ROOT}/include"]
    assert not _build._fresh(obj, [src, inc], flags)               # nothing built yet
    obj.write_bytes(b"\x7fELF")
    assert not _build._fresh(obj, [src, inc], flags)               # an object without a signature proves nothing
    _build._mark(obj, [src, inc], flags)
    assert _build._fresh(obj, [src, inc], flags)
    os.utime(inc, (time.time() + 100, time.time() + 100))          # touched: newer than the object, same bytes
    assert _build._fresh(obj, [src, inc], flags)
    old = time.time() - 10_000
    inc.write_text("#define X 2\n")                                # edited .inc - and made to LOOK older than the object
    os.utime(inc, (old, old))
    assert not _build._fresh(obj, [src, inc], flags), "an edited .inc must force the rebuild whatever its mtime says"
    inc.write_text("#define X 1\n")
    assert _build._fresh(obj, [src, inc], flags)
    assert not _build._fresh(obj, [src, inc], flags + ["-DKF_MUTANT"])   # other flags: another object
    # the tree's own location is not part of the signature (the GPU box unpacks the snapshot elsewhere)
    assert _build._sig([src], [f"-I{_build.ROOT}/include"]) == _build._sig([src], ["-I<root>/include"])


def test_library_carries_the_hash_of_the_sources_it_was_built_from():
    _build.build_device()                                          # (a no-op when the tree is built: signatures match)
    lib = ctypes.CDLL(str(_build.DEVICE_LIB))
    lib.kf_build_source_sha.restype = ctypes.c_char_p
    assert lib.kf_build_source_sha().decode() == _build.device_src_sha() == bench.device_src_sha()
    for o in (_build.BUILD / "attention.o", _build.BUILD / "gemm.o", _build.DEVICE_LIB):
        assert (o.parent / (o.name + ".sig")).exists(), o


def test_bench_refuses_profile_figures_for_a_stale_library(monkeypatch):
    sha = bench.device_src_sha()
    got, src = bench.quoted_traffic("attn_bwd_dkv_mfma", sha)
    assert src is None or not str(src).startswith("refused")       # (None, None) when no current profile exists; never a refusal
    got, src = bench.quoted_traffic("attn_bwd_dkv_mfma", "0123456789abcdef")   # a library linked from other sources
    assert got is None and src.startswith("refused") and "0123456789abcdef" in src
