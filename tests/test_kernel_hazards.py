"""not gpu: the hazards hipcc cannot see inside the inline-asm MFMAs of the dK/dV kernel, checked on the compiled ISA.

The S and dP chains of attn_bwd_dkv_v4_kernel (both head sizes) are asm MFMAs with VGPR accumulators; the compiler neither pads their
results' first VALU read (GFX940: 11 wait states after an 8-pass MFMA) nor knows that an MFMA keeps reading its A / C operands after
issue. tools/kernel_hazards.py compiles attention.hip to ISA for gfx950 (hipcc cross-compiles without a GPU) and counts the wait
states of every such pair; a code change or a compiler update that lets the scheduler close one of the gaps fails here, not as a
rare wrong gradient on the GPU."""
import subprocess
import sys
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent


def test_asm_mfma_hazard_distances():
    r = subprocess.run([sys.executable, str(ROOT / "tools" / "kernel_hazards.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert " 0 violations" in r.stdout and "384 asm MFMAs" in r.stdout, r.stdout
