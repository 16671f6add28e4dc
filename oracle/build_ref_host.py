#!/usr/bin/env python3
"""Recipe: the REFERENCE's own host half on this repository's device library.

    python oracle/build_ref_host.py            -> oracle/_ref/kfunca<EXT_SUFFIX>   (build container only: needs /root/reference)

What is compiled, from the sources where they lie under /root/reference (nothing is copied into the repository; outputs only under
oracle/_ref/, which is git-ignored and travels to the GPU box like every other built .so):
    src/core/*.cpp       the reference's Tensor / TensorImpl / TensorIterator / DeviceAllocator / autograd / operator layer, unmodified
    src/register.cpp     its pybind11 module (PYBIND11_MODULE(kfunca), register.cpp:59-225), unmodified
    docs/seam.cpp        THIS repository's one translation unit: the definitions of src/device/include/*.h over include/kfunca_hip.h
and linked against kfunca_amd/libkfunca_hip.so with -Wl,--no-undefined: every symbol the reference's host core expects from its
device library (libkfunca_device: src/device/*.cu, nvcc + CUTLASS, unbuildable here) is resolved by the C ABI - the drop-in of
SURVEY.md section 8b proved by the linker, and a `kfunca` module whose host logic is the reference's and whose kernels are ours.

No stand-ins: no header, library or tool of the reference is imitated; the .cu files are simply not built (that is the replacement).
tests/test_seam_links.py runs this recipe (CPU, skipped without the reference mount); tests/test_gpu_reference_host.py runs the
reference tests' cases through the resulting module on the GPU (skipped when oracle/_ref/ holds no module).
"""
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
REF = Path("/root/reference/src")
OUT = ROOT / "oracle" / "_ref"


def includes():
    return [f"-I{ROOT / 'include'}", f"-I{REF / 'core' / 'include'}", f"-I{REF / 'core' / 'utils'}", f"-I{REF / 'core' / 'utils' / 'memory'}",
            f"-I{REF / 'device' / 'include'}", f"-I{REF / 'device'}", f"-I{REF / 'device' / 'utils'}", f"-I{REF / 'core'}", f"-I{REF}"]


def run(cmd):
    r = subprocess.run([str(c) for c in cmd], capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("failed: %s\n%s" % (" ".join(map(str, cmd)), r.stderr[-6000:]))


def build(with_module: bool = True) -> Path:
    """Step 1: src/core/*.cpp + docs/seam.cpp -> oracle/_ref/libkfunca_core_on_hip.so, linked with -Wl,--no-undefined against
    libkfunca_hip.so (the linker's proof that the C ABI resolves the whole device seam). Step 2 (with_module): src/register.cpp ->
    oracle/_ref/kfunca<EXT_SUFFIX> on top of it (Python's own symbols stay open, as in any extension module). Returns the last artefact."""
    if not (REF / "core" / "tensor.cpp").exists():
        raise FileNotFoundError("the reference tree is not mounted at /root/reference")
    import pybind11

    from kfunca_amd import _build
    _build.build_device()
    obj = OUT / "obj"
    obj.mkdir(parents=True, exist_ok=True)
    srcs = sorted((REF / "core").glob("*.cpp")) + [ROOT / "docs" / "seam.cpp"] + ([REF / "register.cpp"] if with_module else [])
    py = [f"-I{pybind11.get_include()}", f"-I{sysconfig.get_paths()['include']}"]
    objs = [obj / (s.stem + ".o") for s in srcs]
    # an object is stale when its source OR any header it may include is newer: the reference's headers, the seam's, the C ABI
    hdrs = [h for d in (REF / "core", REF / "device" / "include", ROOT / "include", ROOT / "docs") for h in d.rglob("*.h")]
    newest_hdr = max((h.stat().st_mtime for h in hdrs), default=0.0)
    jobs = [["g++", "-std=c++20", "-O2", "-fPIC", "-w", *includes(), *py, "-c", s, "-o", o] for s, o in zip(srcs, objs)
            if not o.exists() or o.stat().st_mtime < max(s.stat().st_mtime, newest_hdr)]
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(run, jobs))
    core = OUT / "libkfunca_core_on_hip.so"
    # $ORIGIN/../../kfunca_amd: found wherever the repository lies (the GPU box's copy included)
    run(["g++", "-shared", "-o", core, *[o for o in objs if o.stem != "register"], f"-L{ROOT / 'kfunca_amd'}", "-lkfunca_hip", "-Wl,--no-undefined",
         "-Wl,-rpath,$ORIGIN/../../kfunca_amd", "-Wl,-rpath,/opt/rocm/lib"])
    if not with_module:
        return core
    mod = OUT / ("kfunca" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
    run(["g++", "-shared", "-o", mod, obj / "register.o", f"-L{OUT}", "-lkfunca_core_on_hip", f"-L{ROOT / 'kfunca_amd'}", "-lkfunca_hip",
         "-Wl,-rpath,$ORIGIN", "-Wl,-rpath,$ORIGIN/../../kfunca_amd", "-Wl,-rpath,/opt/rocm/lib"])
    return mod


if __name__ == "__main__":
    sys.path.insert(0, str(ROOT))
    print(build())
