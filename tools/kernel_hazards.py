#!/usr/bin/env python3
"""Static check of the hazards hipcc cannot see inside inline-asm MFMAs (dK/dV kernel, both head sizes: S and dP chains on VGPR
accumulators). Compiles kfunca_amd/csrc/device/attention.hip to ISA and, in every attn_bwd_dkv_v4_kernel<*, *, 64 | 128>, verifies for each
`v_mfma_f32_32x32x16_* v[D], A, B, C` (a VGPR destination marks an asm MFMA: the builtin ones write AGPRs there):

  RAW / WAW  the first non-MFMA instruction that reads or writes a register of v[D] after the LAST MFMA of its chain comes at least
             11 wait states later (GFX940: an 8-pass XDL write of a VGPR -> VALU read / write; an instruction counts one wait state,
             `s_nop N` counts N + 1);
  WAR        no instruction writes a VGPR of the MFMA's A or C operand within 8 wait states after it (the MFMA reads them over its passes;
             the compiler thinks an asm statement is done with its inputs when it is issued).

Every check follows ALL paths out of the MFMA: straight on and through each branch to its label (loop back edges included), and an LDS
load's destination and address registers count like any other write / read.
Exit status 0 = clean; prints every violation otherwise.   usage: python tools/kernel_hazards.py [--asm FILE.s]"""
import argparse
import re
import subprocess
import sys
import tempfile
from pathlib import Path

ROOT = Path(__file__).resolve().parent.parent
RAW_STATES, WAR_STATES = 11, 8


def regs(tok):
    """VGPR numbers named by an operand token: v12, v[8:23] (AGPRs, SGPRs, literals: none)."""
    m = re.fullmatch(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.fullmatch(r"v(\d+)", tok)
    return {int(m.group(1))} if m else set()


def parse(line):
    line = line.split(";")[0].strip()
    if not line or line.endswith(":") or line.startswith("."):
        return None
    parts = line.split(None, 1)
    op = parts[0]
    toks = [t.strip() for t in re.split(r",(?![^\[]*\])", parts[1])] if len(parts) > 1 else []
    toks = [t.split()[0] for t in toks if t]  # drop modifiers such as offset:..
    return op, toks


def states(op, toks):
    return int(toks[0]) + 1 if op == "s_nop" and toks else 1


def dst_src(op, toks):
    """(written VGPRs, read VGPRs) of an instruction, by the ISA's operand order (stores / waits / branches write nothing)."""
    if op.startswith(("global_store", "scratch_store", "ds_write", "buffer_store", "s_", "global_load_lds")):
        return set(), set().union(*[regs(t) for t in toks]) if toks else set()
    if not toks:
        return set(), set()
    return regs(toks[0]), set().union(*[regs(t) for t in toks[1:]]) if len(toks) > 1 else set()


def check(name, body):
    """Walks forward from every asm MFMA along EVERY path: straight on, and through each s_branch / s_cbranch_* to its label (loop
    back edges included: a chain-ending MFMA near the bottom of the slice-pair loop is followed into the top of the next iteration)."""
    ins, labels = [], {}
    for l in body:
        t = l.split(";")[0].strip()
        if t.endswith(":"):  # a label (.LBBn_m:): where a branch lands
            labels[t[:-1]] = len(ins)
            continue
        p = parse(l)
        if p:
            ins.append(p)
    bad = []

    def paths(i, budget, visit):
        """Yield (op, toks, wait states so far) along all paths from instruction i, up to `budget` wait states."""
        stack = [(i, 0)]
        seen = set()
        while stack:
            j, w = stack.pop()
            while j < len(ins) and w < budget:
                if (j, w) in seen:
                    break
                seen.add((j, w))
                op2, t2 = ins[j]
                if visit(op2, t2, w) is False:
                    break
                if op2 in ("s_branch",) or op2.startswith("s_cbranch"):
                    tgt = labels.get(t2[0]) if t2 else None
                    if tgt is not None:
                        stack.append((tgt, w + 1))
                    if op2 == "s_branch":
                        break
                w += states(op2, t2)
                j += 1

    for i, (op, toks) in enumerate(ins):
        if not op.startswith("v_mfma_f32_32x32x16") or not toks[0].startswith("v["):
            continue
        d, a_c = regs(toks[0]), regs(toks[1]) | regs(toks[3])
        desc = f"`{op} {', '.join(toks)}`"

        def war(op2, t2, w):  # WAR on A / C operands (an LDS load returning into them counts: its destination is written)
            wr, _ = dst_src(op2, t2)
            if not op2.startswith("v_mfma") and wr & (a_c - d):
                bad.append(f"{name}: WAR {op2} {' '.join(t2)} {w} wait states after {desc}")
                return False
            return True
        paths(i + 1, WAR_STATES, war)

        def raw(op2, t2, w):  # RAW / WAW on the destination: only from the last MFMA of a chain
            wr, rd = dst_src(op2, t2)
            if op2.startswith("v_mfma") and regs(t2[0]) == d:
                return False  # the chain continues: the hardware forwards the accumulator
            if (wr | rd) & d and not op2.startswith("s_waitcnt"):  # ds_read destinations and address registers included
                if w < RAW_STATES:
                    bad.append(f"{name}: RAW/WAW {op2} {' '.join(t2)} only {w} wait states after {desc}")
                return False
            return True
        paths(i + 1, 4 * RAW_STATES, raw)
    bad = sorted(set(bad))
    return bad, sum(1 for op, t in ins if op.startswith("v_mfma_f32_32x32x16") and t[0].startswith("v["))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--asm")
    args = ap.parse_args()
    if args.asm:
        text = Path(args.asm).read_text()
    else:
        sys.path.insert(0, str(ROOT))
        from kfunca_amd import _build as B
        with tempfile.TemporaryDirectory() as td:
            out = Path(td) / "attention.s"
            B._run([B._hipcc(), *B.HIP_FLAGS, "-S", "--cuda-device-only", "-o", out, B.CSRC / "device" / "attention.hip"])
            text = out.read_text()
    total_bad, total = [], 0
    for m in re.finditer(r"^(_ZN2kf22attn_bwd_dkv_v4_kernelILb[01]ELb[01]ELi(?:64|128)EEEvNS_8AttnArgsE):\s", text, re.M):
        body = text[m.end():]
        body = body[:body.index("s_endpgm")].split("\n")
        bad, n = check(m.group(1), body)
        total_bad += bad
        total += n
    print(f"{total} asm MFMAs with VGPR destinations checked, {len(total_bad)} violations")
    for b in total_bad:
        print("  " + b)
    return 1 if total_bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
