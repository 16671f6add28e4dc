"""CPU: docs/seam.cpp - the reference's device seam (src/device/include/*.h) implemented over the C ABI - must type-check
against the REAL reference headers. Runs only where the reference tree is mounted (the build container); nothing from the
reference is copied or shipped: the compiler reads the headers where they lie."""
import re
import shutil
import subprocess
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/src")


@pytest.mark.skipif(not (REF / "core" / "include" / "tensor_iterator.h").exists(), reason="reference tree not mounted")
@pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")
def test_seam_type_checks_against_the_reference_headers():
    cmd = ["g++", "-std=c++20", "-fsyntax-only", "-Wall", "-Wno-unknown-pragmas", f"-I{ROOT / 'include'}", f"-I{REF / 'core' / 'include'}",
           f"-I{REF / 'core' / 'utils'}", f"-I{REF / 'core' / 'utils' / 'memory'}", f"-I{REF / 'device' / 'include'}", f"-I{REF / 'device'}", f"-I{REF / 'device' / 'utils'}",
           str(ROOT / "docs" / "seam.cpp")]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-4000:]


def test_seam_defines_every_declaration_of_the_reference_seam():
    """Names only (works without the reference mount): every function the reference's device headers declare - the list is
    SURVEY.md section 8b's - has a definition in docs/seam.cpp."""
    text = (ROOT / "docs" / "seam.cpp").read_text()
    for name in ("dset_device", "dmalloc", "dfree", "dmemcpy_h2d", "dmemcpy_d2h", "dmemset_zeros", "add_kernel", "sub_kernel", "mul_kernel",
                 "div_kernel", "copy_kernel", "fill_kernel", "sum_kernel", "mean_kernel", "mean_var_kernel", "norm_stat_kernel",
                 "index_put_kernel", "gemm_kernel", "causal_attention_kernel", "sort_stable_kernel", "topk_with_sort", "device_info"):
        assert re.search(r"\b" + name + r"\(", text), name
