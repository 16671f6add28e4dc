"""CPU: the host core (Tensor / TensorIterator / allocator / autograd / binding) under AddressSanitizer + UBSan. GPU sanitizers are
not available on the pool, so this is the sanitizer coverage the path has: an instrumented build of kfunca_amd/csrc/core + binding
(kfunca_amd/_build.py: build_core_asan) runs the host-only test files in a child process with libasan preloaded."""
import os
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent


def _libasan():
    if shutil.which("gcc") is None:
        return None
    p = subprocess.run(["gcc", "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    return p if p and Path(p).exists() else None


@pytest.mark.skipif(_libasan() is None, reason="no libasan")
def test_host_geometry_and_promotion_under_asan_ubsan():
    import tempfile

    from kfunca_amd import _build
    tmp = tempfile.mkdtemp(prefix="kf_asan_")
    try:
        _run_under_asan(_build.build_core_asan(tmp))  # a shadow package `kfunca_amd` with the instrumented _C
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def _run_under_asan(shadow):
    # libstdc++ goes in with libasan: python itself does not link it, and ASan's __cxa_throw interceptor must find the real one
    stdcpp = subprocess.run(["gcc", "-print-file-name=libstdc++.so"], capture_output=True, text=True).stdout.strip()
    env = dict(os.environ, LD_PRELOAD=f"{_libasan()} {stdcpp}", ASAN_OPTIONS="detect_leaks=0:abort_on_error=1", UBSAN_OPTIONS="halt_on_error=1:print_stacktrace=1",
               PYTHONPATH=f"{shadow}{os.pathsep}{ROOT}")
    code = (f"import sys, kfunca_amd, pytest; assert kfunca_amd.__file__.startswith(r'{shadow}'), kfunca_amd.__file__; "
            f"sys.exit(pytest.main(['-q', '-x', '-p', 'no:cacheprovider', r'{ROOT / 'tests' / 'test_host_geometry.py'}']))")
    res = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=str(shadow), timeout=900)
    assert res.returncode == 0, (res.stdout[-3000:], res.stderr[-3000:])
    assert "AddressSanitizer" not in res.stderr and "runtime error" not in res.stderr, res.stderr[-3000:]
