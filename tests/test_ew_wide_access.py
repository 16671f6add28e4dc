"""not gpu: the element-aligned Pack of elementwise.hip / reduce.hip (`packed, aligned(sizeof(T))`) promises 16-byte global accesses at
2- or 4-byte alignment. Nothing in the source forces that: it rests on the compiler still choosing dwordx4 for the relaxed type (and on
gfx950's unaligned access mode at run time, which the GPU tests exercise). This pins the compiler half on the ISA this toolchain
emits: the same-dtype kernels' 16-byte instantiations must contain global_load_dwordx4 / global_store_dwordx4 and no
narrower global load. KF_EW_ALIGNED_ONLY=1 restores the aligned-only dispatch at run time (tests/test_gpu_elementwise.py)."""
import re
import tempfile
from pathlib import Path

from kfunca_amd import _build as B


def _kernels(asm: str):
    """{mangled name: body} of every .globl function in a device-only -S output."""
    out, name, body = {}, None, []
    for line in asm.splitlines():
        m = re.match(r"^(_Z\w+):", line)
        if m:
            if name:
                out[name] = "\n".join(body)
            name, body = m.group(1), []
        elif name:
            body.append(line)
            if line.strip().startswith("s_endpgm"):
                out[name] = "\n".join(body)
                name = None
    return out


def test_relaxed_packs_compile_to_16_byte_accesses():
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "elementwise.s"
        B._run([B._hipcc(), *B.HIP_FLAGS, "-S", "--cuda-device-only", "-o", out, B.CSRC / "device" / "elementwise.hip"])
        kernels = _kernels(out.read_text())
    # ew_same_kernel<T, VEC, NIN, MODE, CONTIG> with VEC = 16 / sizeof(T): the contiguous arithmetic (MODE 0, two inputs) and copy (MODE 1) forms
    wide = {n: b for n, b in kernels.items() if "ew_same_kernel" in n and re.search(r"ew_same_kernelI(\w+?)Li(8|4|2|16)ELi[12]ELi[01]ELb1E", n)}
    assert len(wide) >= 6, sorted(kernels)[:40]
    checked = 0
    for name, body in wide.items():
        m = re.search(r"ew_same_kernelI(\w+?)Li(\d+)ELi", name)
        vec = int(m.group(2))
        loads = re.findall(r"\b(global_load_\w+)", body)
        stores = re.findall(r"\b(global_store_\w+)", body)
        if not loads:
            continue
        # a VEC-element pack of a 16 / VEC-byte type is 16 bytes: every global access of the kernel must be the x4 form
        assert set(loads) == {"global_load_dwordx4"}, (name, vec, sorted(set(loads)))
        assert set(stores) == {"global_store_dwordx4"}, (name, vec, sorted(set(stores)))
        checked += 1
    assert checked >= 6
