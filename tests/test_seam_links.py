"""CPU: the drop-in boundary proved by the LINKER.

oracle/build_ref_host.py compiles the reference's own host half - src/core/*.cpp, unmodified, where it lies under /root/reference -
together with docs/seam.cpp (this repository's definitions of src/device/include/*.h over include/kfunca_hip.h) and links the result
against kfunca_amd/libkfunca_hip.so with -Wl,--no-undefined: every symbol the reference's host core expects from its device library
must be resolved by the C ABI, or the link fails. On top of that object src/register.cpp (the reference's pybind11 module) is linked
into oracle/_ref/kfunca*.so, which tests/test_gpu_reference_host.py then drives on the GPU. Skipped where the reference is not mounted."""
import re
import shutil
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/src")

pytestmark = [pytest.mark.skipif(not (REF / "core" / "tensor.cpp").exists(), reason="reference tree not mounted"),
              pytest.mark.skipif(shutil.which("g++") is None, reason="no host compiler")]


@pytest.fixture(scope="module")
def built():
    sys.path.insert(0, str(ROOT))
    from oracle import build_ref_host
    return build_ref_host.build(with_module=True)


def test_reference_core_plus_seam_links_with_no_undefined_symbols(built):
    core = built.parent / "libkfunca_core_on_hip.so"
    assert core.exists() and built.exists()
    # what the linked object still imports: only the C ABI, libstdc++ / libc / libm / libgcc - no dmalloc, no *_kernel(TensorIterator&)
    und = subprocess.run(["nm", "-D", "--undefined-only", str(core)], capture_output=True, text=True, check=True).stdout
    names = [ln.split()[-1].split("@")[0] for ln in und.splitlines() if ln.strip()]
    kf = sorted(n for n in names if n.startswith("kf_"))
    assert len(kf) >= 20, kf
    header = (ROOT / "include" / "kfunca_hip.h").read_text()
    for n in kf:
        assert re.search(r"\b" + n + r"\s*\(", header), f"{n} is imported by the seam but not declared in include/kfunca_hip.h"
    seam_names = ("dset_device", "dmalloc", "dfree", "dmemcpy_h2d", "dmemcpy_d2h", "dmemset_zeros", "add_kernel", "gemm_kernel",
                  "causal_attention_kernel", "sort_stable_kernel", "index_put_kernel", "mean_var_kernel", "device_info")
    left = [n for n in names if any(s in n for s in seam_names)]
    assert not left, f"device-seam symbols left undefined: {left}"
    # and the seam's functions are DEFINED in it (the reference's host core calls them)
    defined = subprocess.run(["nm", "-D", "--defined-only", "-C", str(core)], capture_output=True, text=True, check=True).stdout
    for s in seam_names:
        assert re.search(r"\b" + s + r"\(", defined), s


def test_reference_module_imports_and_fails_loudly_without_a_gpu(built):
    """The reference's own `kfunca` Python module over our device library: importable here; without a GPU every operator raises the
    reference's own error type (utils::Error -> RuntimeError) carrying the C ABI's message - there is no CPU path behind it either."""
    code = ("import sys; sys.path.insert(0, %r); import kfunca, numpy as np\n"
            "names = [n for n in dir(kfunca) if not n.startswith('_')]\n"
            "assert {'from_numpy', 'gemm', 'causal_attention', 'cat', 'tensor', 'dtype', 'device_info', 'memstat'} <= set(names), names\n"
            "import torch\n"
            "if torch.cuda.device_count() == 0:\n"
            "    try:\n"
            "        kfunca.from_numpy(np.zeros((2, 3), np.float32), 0)\n"
            "        raise SystemExit('no error without a GPU')\n"
            "    except RuntimeError as e:\n"
            "        assert 'enforce fail' in str(e), str(e)\n"
            "print('ok')\n") % str(built.parent)
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "ok" in r.stdout, r.stdout + r.stderr
