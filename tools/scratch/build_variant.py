#!/usr/bin/env python3
"""Build a variant of the device library with extra -D flags on one source: tools/scratch/lib_<name>.so (same-box A/B through KF_HIP_LIB).
usage: build_variant.py NAME SOURCE.hip [-DFLAG ...]"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import _build as B  # noqa: E402

name, src, flags = sys.argv[1], sys.argv[2], sys.argv[3:]
B.build_device()
obj = B.BUILD / f"variant_{name}.o"
B._run([B._hipcc(), *B.HIP_FLAGS, *flags, "-c", B.CSRC / "device" / src, "-o", obj])
stems = [s.stem for s in sorted((B.CSRC / "device").glob("*.hip")) if s.stem != Path(src).stem]
objs = [B.BUILD / (s + ".o") for s in stems] + [obj]
out = Path(__file__).resolve().parent / f"lib_{name}.so"
B._run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{B.ROCM / 'lib'}", "-lrccl", f"-Wl,-rpath,{B.ROCM / 'lib'}"])
print(out)
