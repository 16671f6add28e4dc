#!/usr/bin/env python3
"""Build a variant of the device library from a PATCHED COPY of one source (the tree's own source - and with it the profile stamps - stays untouched):
tools/scratch/lib_<name>.so. usage: build_patched.py NAME SOURCE.hip PATCH.py [-DFLAG ...]   (PATCH.py defines SUBS = [(old text, new text), ...]; every old text must occur exactly once)"""
import sys
from pathlib import Path

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import _build as B  # noqa: E402

name, src, patch, flags = sys.argv[1], sys.argv[2], sys.argv[3], sys.argv[4:]
ns = {}
exec(Path(patch).read_text(), ns)
B.build_device()
text = (B.CSRC / "device" / src).read_text()
for old, new in ns["SUBS"]:
    assert text.count(old) == 1, f"{old!r} occurs {text.count(old)} times"
    text = text.replace(old, new)
here = Path(__file__).resolve().parent
patched = here / f"_patched_{name}.hip"
patched.write_text(text)
obj = B.BUILD / f"variant_{name}.o"
B._run([B._hipcc(), *B.HIP_FLAGS, *flags, f"-I{B.CSRC / 'device'}", "-c", patched, "-o", obj])
stems = [s.stem for s in sorted((B.CSRC / "device").glob("*.hip")) if s.stem != Path(src).stem]
objs = [B.BUILD / (s + ".o") for s in stems] + [obj]
out = here / f"lib_{name}.so"
B._run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", out, *objs, f"-L{B.ROCM / 'lib'}", "-lrccl", f"-Wl,-rpath,{B.ROCM / 'lib'}"])
patched.unlink()
print(out)
