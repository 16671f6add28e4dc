"""CPU-only: the C-ABI shared library loads without a GPU and exports every function that
include/kfunca_hip.h declares; the ctypes mirror of the descriptor structs has the C layout."""
import ctypes as C
import re
import subprocess
from pathlib import Path

from kfunca_amd import hip_abi as H

ROOT = Path(__file__).resolve().parent.parent


def declared_functions():
    text = (ROOT / "include" / "kfunca_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(kf_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = H.lib()
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/kfunca_hip.h but not exported"
    assert sorted(H.EXPORTS) == names, "hip_abi.EXPORTS out of sync with the header"
    assert lib.kf_abi_version() == 7


def test_struct_layout_matches_c():
    src = r'''
    #include <stdio.h>
    #include <stddef.h>
    #include "kfunca_hip.h"
    int main(void) {
        printf("%zu %zu %zu %zu %zu %zu %zu\n", sizeof(kf_iter_desc), offsetof(kf_iter_desc, dtype), offsetof(kf_iter_desc, shape),
               offsetof(kf_iter_desc, stride_bytes), offsetof(kf_iter_desc, data), sizeof(kf_device_props), offsetof(kf_device_props, total_mem));
        return 0;
    }'''
    exe = Path("/tmp/kf_layout_probe")
    subprocess.run(["gcc", "-x", "c", "-", f"-I{ROOT / 'include'}", "-o", str(exe)], input=src, text=True, check=True)
    got = [int(x) for x in subprocess.run([str(exe)], capture_output=True, text=True, check=True).stdout.split()]
    want = [C.sizeof(H.IterDesc), H.IterDesc.dtype.offset, H.IterDesc.shape.offset, H.IterDesc.stride_bytes.offset,
            H.IterDesc.data.offset, C.sizeof(H.DeviceProps), H.DeviceProps.total_mem.offset]
    assert got == want


def test_no_gpu_calls_fail_loudly_not_silently():
    if H.device_count() > 0:
        return
    # without a device every compute entry point reports an error status; nothing falls back to a CPU path
    a = (C.c_float * 4)()
    v = H.View(C.addressof(a), (4,), (1,), H.F32)
    d = H.make_desc([v], [v, v])
    rc = H.lib().kf_elementwise(H.EW_ADD, C.byref(d), H.F32, 0.0, None)
    assert rc != H.KF_OK and H.lib().kf_last_error()


def test_descriptor_builder_matches_reference_rules():
    # make_desc mirrors tensor_iterator.cpp: reversed dims, byte strides, broadcast -> 0, coalescing
    a = H.View(0x1000, (5, 7, 11), (77, 11, 1), H.F32)
    b = H.View(0x2000, (5, 1, 11), (11, 11, 1), H.F32)
    d = H.make_desc([a], [a, b])
    assert d.ndim == 3 and list(d.shape[:3]) == [11, 7, 5]
    assert list(d.stride_bytes[2][:3]) == [4, 0, 44]
    c = H.View(0x3000, (12, 11, 331), (3641, 331, 1), H.F32)
    d = H.make_desc([c], [c, c])
    assert d.ndim == 1 and d.shape[0] == 12 * 11 * 331  # fully coalesced
    out = H.View(0x4000, (1024, 1), (1, 1), H.F32)
    inp = H.View(0x5000, (1024, 1024), (1024, 1), H.F32)
    r = H.make_reduce_desc(out, inp, 1)  # C1 sum(1): dims [1024 (reduced, out stride 0), 1024]  
    assert r.ndim == 2 and list(r.shape[:2]) == [1024, 1024]
    assert list(r.stride_bytes[0][:2]) == [0, 4] and list(r.stride_bytes[1][:2]) == [4, 4096]
