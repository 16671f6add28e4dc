#!/usr/bin/env python3
"""Explain a disagreement of tests/test_gpu_host_diff_fuzz.py: for the given seeds, the first tensor (or step) on which the two hosts differ and the chain of
instructions that made it, with both hosts' shape / strides / offset of every tensor on the chain.  diff_fuzz_explain.py SEED [SEED ...]"""
import sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parent.parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "oracle" / "_ref"))
import kfunca as REF  # noqa: E402
import kfunca_amd as MINE  # noqa: E402
from tests import test_gpu_host_diff_fuzz as F  # noqa: E402

MADE = {"new": 1, "permute": 1, "getitem": 1, "contiguous": 1, "view": 1, "binary": 1, "scalar": 1, "reduce": 1, "moments": 2, "convert": 1, "cat": 1, "split": 2,
        "sort": 2, "topk": 2, "gemm": 1, "handle": 1, "zeros": 1}
def makers(prog):
    out, k = {}, 0
    for n, ins in enumerate(prog):
        m = sum(ins[2]) if str(ins[0]) == "autograd" else MADE.get(str(ins[0]), 0)
        for j in range(m): out[k + j] = n
        k += m
    return out
def deps(ins):
    k = str(ins[0])
    if k in ("permute", "getitem", "contiguous", "view", "convert", "split", "sort", "topk"): return [ins[1]]
    if k in ("binary",): return [ins[2], ins[3]]
    if k in ("scalar", "reduce", "moments"): return [ins[2]]
    if k == "cat": return list(ins[1])
    return []
def replay(kf, prog):
    """The tensors after running prog (F.run keeps them to itself)."""
    import types
    keep = {}
    orig = F.run
    src = Path(F.__file__).read_text()
    # run() builds `ts`; re-execute it with a hook: simplest is to call run on the prefix and rebuild the tensors by running the prefix again here
    ts = []
    ops = {"+": lambda a, b: a + b, "-": lambda a, b: a - b, "*": lambda a, b: a * b, "/": lambda a, b: a / b}
    for ins in prog:
        k = str(ins[0])
        try:
            if k == "new": ts.append(kf.from_numpy(ins[1], 0))
            elif k == "permute": ts.append(ts[ins[1]].permute(*ins[2]))
            elif k == "getitem": ts.append(ts[ins[1]][tuple(slice(*x) if isinstance(x, tuple) else x for x in ins[2])])
            elif k == "contiguous": ts.append(ts[ins[1]].contiguous())
            elif k == "view": ts.append(ts[ins[1]].view(*ins[2]))
            elif k == "binary": ts.append(ops[ins[1]](ts[ins[2]], ts[ins[3]]))
            elif k == "scalar": ts.append(ops[ins[1]](ts[ins[2]], ins[3]))
            elif k == "reduce": ts.append(getattr(ts[ins[2]], ins[1])(ins[3]))
            elif k == "moments":
                r = ts[ins[2]].norm_stat(ins[3]) if ins[1] == "norm_stat" else ts[ins[2]].mean_var(ins[3], ins[1] == "mean_std"); ts += [r[0], r[1]]
            elif k == "convert": ts.append(getattr(ts[ins[1]], ins[2])())
            elif k == "cat": ts.append(kf.cat([ts[p] for p in ins[1]], ins[2]))
            elif k == "split": ts += list(ts[ins[1]].split(ins[2], ins[3]))
            elif k == "sort": ts += list(ts[ins[1]].sort(ins[2], ins[3]))
            elif k == "topk": ts += list(ts[ins[1]].topk(ins[2], ins[3], ins[4]))
            elif k == "gemm": ts.append(kf.gemm(kf.from_numpy(ins[1], 0), kf.from_numpy(ins[2], 0), ins[3], 0.0))
            elif k == "autograd": ts += [None] * sum(ins[2])
        except Exception:
            ts += [None] * (F_MADE(ins))
    return ts
def F_MADE(ins):
    return sum(ins[2]) if str(ins[0]) == "autograd" else MADE.get(str(ins[0]), 0)
for seed in map(int, sys.argv[1:]):
    prog = F.make_program(1000 + seed, steps=28 + seed % 17)
    st_r, fin_r, _ = F.run(REF, prog)
    st_m, fin_m, _ = F.run(MINE, prog)
    print(f"==== seed {seed}")
    bad_step = next((n for n, (a, b) in enumerate(zip(st_r, st_m)) if (a == "ok") != (b == "ok")), None)
    if bad_step is not None:
        print("step", bad_step, F.origin_args(prog[bad_step]), "| reference:", st_r[bad_step][:60], "| this host:", st_m[bad_step][:60])
        # the target tensor's geometry in both hosts: replay up to the step
        for label, kf in (("reference", REF), ("this host", MINE)):
            ts = replay(kf, prog[:bad_step])
            t = ts[prog[bad_step][2] if str(prog[bad_step][0]) != "fill" else prog[bad_step][1]]
            print("   ", label, repr(t).split("{")[0].strip()[:200] if "stride" in repr(t) else (t.sizes(),))
        mk = makers(prog)
        chain, todo = [], [prog[bad_step][2] if str(prog[bad_step][0]) != "fill" else prog[bad_step][1]]
        while todo:
            t = todo.pop()
            i = mk[t]
            if i in chain: continue
            chain.append(i)
            todo += deps(prog[i])
        for i in sorted(chain):
            print("      ", i, F.origin_args(prog[i]), prog[i][2] if str(prog[i][0]) in ("getitem", "permute", "view") else "")
        continue
    mk = makers(prog)
    for n, (a, b) in enumerate(zip(fin_r, fin_m)):
        if a is None or a[0] == "raised": continue
        if not np.array_equal(a[2].view(np.uint8), b[2].view(np.uint8)):
            chain, todo = [], [n]
            while todo:
                t = todo.pop()
                i = mk[t]
                if i in chain: continue
                chain.append(i)
                todo += deps(prog[i])
            inplace = [i for i, ins in enumerate(prog) if str(ins[0]) in ("inplace", "inplace_scalar", "fill", "iput")]
            print(f"tensor {n} differs ({a[1]} {a[0]}): max |diff| {np.abs(a[2].astype(np.float64) - b[2].astype(np.float64)).max():.3g}; reference {a[2].reshape(-1)[:4]} this host {b[2].reshape(-1)[:4]}")
            for i in sorted(chain):
                print("   ", i, F.origin_args(prog[i]))
            print("    in-place steps of the program:", [(i, F.origin_args(prog[i])) for i in inplace])
            break
