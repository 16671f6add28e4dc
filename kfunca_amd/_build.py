"""In-tree builds: the gfx950 device library, the host core + pybind11 module, and the oracle.

hipcc cross-compiles gfx950 without a GPU, so this runs in the build container and the
resulting .so files travel to the GPU box with the source tree (they are git-ignored).
Nothing here falls back to anything: a failed compile raises.
"""
from __future__ import annotations

import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
PKG = ROOT / "kfunca_amd"
CSRC = PKG / "csrc"
INCLUDE = ROOT / "include"
BUILD = PKG / "_build"
ROCM = Path(os.environ.get("ROCM_PATH", "/opt/rocm"))

DEVICE_LIB = PKG / "libkfunca_hip.so"
CORE_MODULE = PKG / ("_C" + (sysconfig.get_config_var("EXT_SUFFIX") or ".so"))
ORACLE_LIB = ROOT / "oracle" / "liboracle.so"

HIP_FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-gpu-rdc",
             "-Wno-unused-result", f"-I{INCLUDE}", f"-I{CSRC / 'device'}"]
CXX_FLAGS = ["-O2", "-std=c++17", "-fPIC", "-fvisibility=hidden", "-Wall", "-Wno-sign-compare",
             f"-I{INCLUDE}", f"-I{CSRC / 'core'}"]


def _run(cmd, **kw):
    res = subprocess.run([str(c) for c in cmd], capture_output=True, text=True, **kw)
    if res.returncode != 0:
        raise RuntimeError("command failed: %s\n%s\n%s" % (" ".join(map(str, cmd)), res.stdout, res.stderr))
    return res


def _newer(target: Path, deps) -> bool:
    if not target.exists():
        return False
    t = target.stat().st_mtime
    return all(Path(d).stat().st_mtime <= t for d in deps)


def _hipcc() -> str:
    h = shutil.which("hipcc") or str(ROCM / "bin" / "hipcc")
    if not Path(h).exists():
        raise RuntimeError("hipcc not found: the device library cannot be built")
    return h


def build_device(force: bool = False) -> Path:
    """hipcc --offload-arch=gfx950: kfunca_amd/csrc/device/*.hip -> kfunca_amd/libkfunca_hip.so"""
    BUILD.mkdir(exist_ok=True)
    srcs = sorted((CSRC / "device").glob("*.hip"))
    hdrs = sorted((CSRC / "device").glob("*.h")) + sorted((CSRC / "device").glob("*.inc")) + sorted(INCLUDE.glob("*.h"))
    objs = []
    jobs = []
    for s in srcs:
        o = BUILD / (s.stem + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            jobs.append([_hipcc(), *HIP_FLAGS, "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    if force or jobs or not _newer(DEVICE_LIB, objs):
        _run([_hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", DEVICE_LIB, *objs,
              f"-L{ROCM / 'lib'}", "-lrccl", f"-Wl,-rpath,{ROCM / 'lib'}"])
    return DEVICE_LIB


def build_core(force: bool = False) -> Path:
    """g++: host core (Tensor / TensorIterator / allocator / autograd / ops) + pybind11 module."""
    import pybind11

    BUILD.mkdir(exist_ok=True)
    build_device(force)
    srcs = sorted((CSRC / "core").glob("*.cpp")) + sorted((CSRC / "binding").glob("*.cpp"))
    hdrs = sorted((CSRC / "core").glob("*.h")) + sorted(INCLUDE.glob("*.h"))
    py_inc = sysconfig.get_paths()["include"]
    objs, jobs = [], []
    for s in srcs:
        o = BUILD / ("core_" + s.stem + ".o")
        objs.append(o)
        if force or not _newer(o, [s] + hdrs):
            jobs.append(["g++", *CXX_FLAGS, f"-I{pybind11.get_include()}", f"-I{py_inc}", "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=min(6, max(1, len(jobs)))) as ex:
        list(ex.map(_run, jobs))
    if force or jobs or not _newer(CORE_MODULE, objs + [DEVICE_LIB]):
        _run(["g++", "-shared", "-fPIC", "-o", CORE_MODULE, *objs, f"-L{PKG}", "-lkfunca_hip",
              "-Wl,-rpath,$ORIGIN", f"-Wl,-rpath,{ROCM / 'lib'}"])
    return CORE_MODULE


def build_oracle(force: bool = False) -> Path:
    """gcc: the CPU restatement (test infrastructure only) -> oracle/liboracle.so"""
    src = ROOT / "oracle" / "oracle.c"
    hdr = ROOT / "oracle" / "oracle.h"
    if force or not _newer(ORACLE_LIB, [src, hdr]):
        # fixed ISA baseline + no implicit fma contraction: the checker must behave the same on the GPU box's host CPU
        _run(["gcc", "-O3", "-march=x86-64-v3", "-ffp-contract=off", "-fopenmp", "-fPIC", "-shared", "-std=c11", "-o", ORACLE_LIB, src, "-lm"])
    return ORACLE_LIB


def build_core_asan(out_root=None) -> Path:
    """The host core + binding compiled with AddressSanitizer + UBSan into <out_root>/kfunca_amd/ (default _build/asan; a shadow package
    beside copies of the package's Python files - 70 MB of instrumented objects, so tests build it in a temporary directory): the CPU-side sanitizer build the GPU pool cannot offer (tests/test_host_asan.py runs the host-only
    tests under it with libasan preloaded). Links the same libkfunca_hip.so."""
    import pybind11

    root = Path(out_root) if out_root else BUILD / "asan"
    out_pkg = root / "kfunca_amd"
    out_pkg.mkdir(parents=True, exist_ok=True)
    build_device()
    srcs = sorted((CSRC / "core").glob("*.cpp")) + sorted((CSRC / "binding").glob("*.cpp"))
    py_inc = sysconfig.get_paths()["include"]
    flags = ["-O1", "-g", "-fno-omit-frame-pointer", "-fsanitize=address,undefined", "-fno-sanitize-recover=undefined", "-std=c++17", "-fPIC",
             "-fvisibility=hidden", f"-I{INCLUDE}", f"-I{CSRC / 'core'}", f"-I{pybind11.get_include()}", f"-I{py_inc}"]
    objs = []
    jobs = []
    for s in srcs:
        o = root / ("core_" + s.stem + ".o")
        objs.append(o)
        jobs.append(["g++", *flags, "-c", s, "-o", o])
    with ThreadPoolExecutor(max_workers=6) as ex:
        list(ex.map(_run, jobs))
    mod = out_pkg / CORE_MODULE.name
    _run(["g++", "-shared", "-fPIC", "-fsanitize=address,undefined", "-o", mod, *objs, f"-L{PKG}", "-lkfunca_hip", f"-Wl,-rpath,{PKG}",
          f"-Wl,-rpath,{ROCM / 'lib'}"])
    for f in PKG.glob("*.py"):
        shutil.copy2(f, out_pkg / f.name)
    shutil.copy2(DEVICE_LIB, out_pkg / DEVICE_LIB.name)
    return out_pkg.parent


def build_diag(force: bool = False) -> Path:
    """The diagnostic library tools/gemm_clock.py loads (clock-stamped GEMM kernel, -DKF_DIAG_BUILD): a separate .so under
    _build/, so that nothing diagnostic is exported from libkfunca_hip.so."""
    BUILD.mkdir(exist_ok=True)
    out = BUILD / "libkfunca_hip_diag.so"
    srcs = [CSRC / "device" / "gemm.hip", CSRC / "device" / "runtime.hip"]
    hdrs = sorted((CSRC / "device").glob("*.h")) + sorted((CSRC / "device").glob("*.inc")) + sorted(INCLUDE.glob("*.h"))
    if force or not _newer(out, srcs + hdrs):
        _run([_hipcc(), *HIP_FLAGS, "-DKF_DIAG_BUILD", "-shared", "-o", out, *srcs])
    return out


def build_mutant(force: bool = False) -> Path:
    """The mutation library tests/test_gpu_attention_mutants.py loads beside the real one (-DKF_MUTANT: attention kernels with
    deliberate single-tile defects behind a run-time selector): a separate .so under _build/, nothing of it in libkfunca_hip.so."""
    BUILD.mkdir(exist_ok=True)
    out = BUILD / "libkfunca_hip_mutant.so"
    srcs = [CSRC / "device" / "attention.hip", CSRC / "device" / "runtime.hip"]
    hdrs = sorted((CSRC / "device").glob("*.h")) + sorted((CSRC / "device").glob("*.inc")) + sorted(INCLUDE.glob("*.h"))
    if force or not _newer(out, srcs + hdrs):
        _run([_hipcc(), *HIP_FLAGS, "-DKF_MUTANT", "-shared", "-o", out, *srcs])
    return out


def build_all(force: bool = False):
    build_device(force)
    if (CSRC / "core").exists() and any((CSRC / "core").glob("*.cpp")):
        build_core(force)
    if (ROOT / "oracle" / "oracle.c").exists():
        build_oracle(force)
    # test-only extras: a failure here must not take the product build down with it (the tests that need them say so)
    try:
        build_mutant(force)
    except Exception as e:  # noqa: BLE001
        print(f"[kfunca_amd._build] mutation library NOT built (tests/test_gpu_attention_mutants.py will fail): {e}", file=sys.stderr)
    if Path("/root/reference/src/core/tensor.cpp").exists():  # build container only: the reference's host half on our device library
        try:
            sys.path.insert(0, str(ROOT))
            from oracle import build_ref_host
            build_ref_host.build(with_module=True)
        except Exception as e:  # noqa: BLE001
            print(f"[kfunca_amd._build] oracle/_ref NOT built (tests/test_seam_links.py, test_gpu_reference_host.py will fail): {e}", file=sys.stderr)


if __name__ == "__main__":
    build_all(force="--force" in sys.argv)
    print("built:", DEVICE_LIB, CORE_MODULE if CORE_MODULE.exists() else "", ORACLE_LIB if ORACLE_LIB.exists() else "")
