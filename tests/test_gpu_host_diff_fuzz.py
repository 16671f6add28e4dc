"""-m gpu: randomised DIFFERENTIAL test of the two hosts - the reference's own host half (oracle/_ref/kfunca*.so: src/core/*.cpp + register.cpp,
unmodified, over this repository's device library; tests/test_gpu_reference_host.py explains the build) against this repository's host core.

A seeded generator writes small programs over the module API the reference defines (register.cpp:76-218): tensors from numpy, permute / slicing with steps /
select / view / contiguous, broadcasting binary operators with type promotion (Tensor op Tensor, Tensor op scalar, in place on views), sum / mean / mean_var /
norm_stat along a dimension, half / bfloat16 / float conversions, cat / split, fill_, sort / topk, index_put_, gemm, causal_attention, and the backward pass of a
random DAG of additions with shared nodes. The generator decides from a numpy shadow of every tensor's SHAPE only, so the
two modules execute the same program; after every step both must have raised or both succeeded (the reference's CHECK_FAIL behaviour is part of the API), and at the
end every live tensor must hold the SAME BITS, shape and dtype in both. Equal bits mean our Tensor / TensorIterator restatement (view strides, broadcast
geometry, dimension coalescing and reordering, promotion, output allocation, reduction plans) drives the kernels exactly as the reference's does - on programs nobody
wrote by hand. Rows a1-a3, a6, a13 of SURVEY.md section 8."""
import os
import sys
from pathlib import Path

import numpy as np
import pytest

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))   # (the child process is started by path: the repository root is not on its sys.path)
import kfunca_amd  # noqa: E402

pytestmark = pytest.mark.gpu
REFDIR = Path(__file__).resolve().parent.parent / "oracle" / "_ref"
SKIP_OPS = set(os.environ.get("KF_DIFF_FUZZ_SKIP", "").split(","))   # (bisecting aid: operations the generator leaves out)
FLOATS, INTS = ("f4", "f8"), ("i4", "q")   # ("q": the reference's from_numpy knows int64 as long long only, register.cpp:28-37)


def make_program(seed, steps=28):
    """A list of instructions (plain tuples) + the numpy inputs they name. The shadow tracks shape, numpy-kind ('f' / 'i') and whether the tensor is 16-bit."""
    rng = np.random.default_rng(seed)
    prog, shadow = [], []   # shadow[i] = dict(shape=..., kind='f'|'i', h=bool, contig=bool)

    big = rng.random() < 0.12    # one program in eight works on LARGE extents: the vectorised / tall / wide / multi-block plans of the kernels behind the same geometry

    def rand_shape():
        nd = int(rng.integers(1, 5))
        if big:
            shape = [int(rng.choice([1, 2, 3, 16, 100, 257, 1024, 4099])) for _ in range(nd)]
            cap = 3_000_000
        else:
            shape = [int(rng.choice([1, 2, 3, 4, 5, 7, 8, 16, 33, 70])) for _ in range(nd)]
            cap = 150_000                     # (small tensors: most programs test geometry, not bandwidth)
        while int(np.prod(shape)) > cap:
            shape[int(np.argmax(shape))] = max(1, shape[int(np.argmax(shape))] // 2)
        return tuple(shape)

    def new(shape=None, dt=None):
        shape = rand_shape() if shape is None else shape
        dt = str(rng.choice(FLOATS + INTS)) if dt is None else dt
        if dt in FLOATS:
            arr = rng.uniform(-4, 4, size=shape).astype(dt)
        else:
            arr = rng.integers(1, 9, size=shape).astype(dt) * rng.choice([-1, 1], size=shape).astype(dt)   # never zero: integer division stays defined
        prog.append(("new", arr))
        shadow.append(dict(shape=tuple(shape), kind=np.dtype(dt).kind, h=False, contig=True, off=False, dt=dt))
        return len(shadow) - 1

    def push(shape, kind, h, contig, off=False):
        # off: the tensor may start behind its storage's first element. The reference's permute() drops the storage offset (tensor.cpp:189: as_strided without it -
        # x[2:5].permute(0) is x[0:3] there; index_ops.cpp:22 does the same to index_put_): this host keeps it, so the generator never permutes such a tensor.
        shadow.append(dict(shape=tuple(int(s) for s in shape), kind=kind, h=h, contig=contig, off=off))

    new()
    new()
    for _ in range(steps):
        op = rng.choice(["permute", "getitem", "contiguous", "view", "binary", "scalar", "inplace", "inplace_scalar", "reduce", "moments", "convert", "cat", "split",
                         "fill", "new", "sort", "topk", "iput", "gemm", "attn", "autograd", "invalid", "handle", "zeros"],
                        p=[.07, .11, .05, .05, .12, .05, .09, .04, .08, .04, .05, .04, .03, .02, .03, .03, .02, .02, .01, .01, .01, .01, .01, .01])
        i = int(rng.integers(0, len(shadow)))
        s = shadow[i]
        nd = len(s["shape"])
        if op in SKIP_OPS:
            continue
        if op == "new":
            new()
        elif op == "permute":
            if s["off"]:
                continue
            perm = [int(p) for p in rng.permutation(nd)]
            prog.append(("permute", i, perm))
            push([s["shape"][p] for p in perm], s["kind"], s["h"], False, False)
        elif op == "getitem":
            key, shp, moved = [], [], s["off"]
            for d in range(nd):
                n = s["shape"][d]
                if rng.random() < 0.25 and nd - sum(1 for k in key if isinstance(k, int)) > 1:
                    key.append(int(rng.integers(0, n)))
                    moved = moved or key[-1] > 0
                else:
                    a = int(rng.integers(0, n))
                    b = int(rng.integers(a + 1, n + 1))
                    st = int(rng.choice([1, 1, 2, 3]))
                    key.append((a, b, st))
                    moved = moved or a > 0
                    shp.append(len(range(a, b, st)))
            if not shp:
                continue
            prog.append(("getitem", i, key))
            push(shp, s["kind"], s["h"], False, moved)
        elif op == "contiguous":
            prog.append(("contiguous", i))
            push(s["shape"], s["kind"], s["h"], True)
        elif op == "view":
            if not s["contig"]:
                continue
            n = int(np.prod(s["shape"]))
            divs = [d for d in (1, 2, 3, 4, 5, 7, 8, 16) if n % d == 0]
            a = int(rng.choice(divs))
            rest = n // a
            divs2 = [d for d in (1, 2, 3, 4, 5, 7, 8) if rest % d == 0]
            b = int(rng.choice(divs2))
            shp = [a, b, rest // b] if rng.random() < 0.5 else [a, -1, b]
            prog.append(("view", i, shp))
            push([a, b, rest // b] if shp[1] != -1 else [a, rest // b, b], s["kind"], s["h"], False)   # (the reference's flag: false for whatever as_strided_ made, view() included)
        elif op in ("binary", "inplace"):
            # the other operand: a fresh tensor whose shape broadcasts against tensor i (dims dropped from the front, dims set to 1), any dtype
            shp = list(s["shape"])
            for d in range(nd):
                if rng.random() < 0.3:
                    shp[d] = 1
            drop = int(rng.integers(0, nd)) if rng.random() < 0.3 else 0
            shp = shp[drop:]
            if op == "binary" and rng.random() < 0.3 and nd < 4 and not big:     # ... or the fresh one is the larger
                shp = [int(rng.choice([2, 3, 5]))] + list(s["shape"])
            dt = str(rng.choice(FLOATS + INTS))
            if s["h"]:
                dt = "f4"     # (16-bit tensors meet float partners; the fresh operand may be converted below)
            j = new(tuple(shp), dt)
            if s["h"] and rng.random() < 0.7:
                prog.append(("convert", j, "half" if s["h"] == "half" else "bfloat16"))
                push(shp, "f", s["h"], True)
                j = len(shadow) - 1
            sym = str(rng.choice(["+", "-", "*", "/"]))
            if op == "binary":
                a, b = (i, j) if rng.random() < 0.5 else (j, i)
                prog.append(("binary", sym, a, b))
                out_shape = np.broadcast_shapes(shadow[a]["shape"], shadow[b]["shape"])
                kind = "f" if "f" in (shadow[a]["kind"], shadow[b]["kind"]) else "i"
                push(out_shape, kind, s["h"] if shadow[j]["h"] else False, True)
            else:
                if np.broadcast_shapes(s["shape"], tuple(shp)) != s["shape"]:
                    continue
                prog.append(("inplace", sym, i, j))
        elif op in ("scalar", "inplace_scalar"):
            val = float(rng.choice([2, 3, 0.5, -1.5, 7]))
            if s["kind"] == "i":
                val = float(int(val) or 2)
            sym = str(rng.choice(["+", "-", "*", "/"]))
            prog.append((op, sym, i, val))
            if op == "scalar":
                push(s["shape"], s["kind"], s["h"], True)
        elif op == "reduce":
            d = int(rng.integers(0, nd))
            prog.append(("reduce", str(rng.choice(["sum", "mean"])), i, d))
            shp = list(s["shape"])
            shp[d] = 1
            push(shp, s["kind"], s["h"], True)
        elif op == "moments":
            if s["kind"] != "f" or s["h"]:
                continue
            d = int(rng.integers(0, nd))
            which = str(rng.choice(["mean_var", "mean_std", "norm_stat"]))
            if which == "norm_stat" and (nd != 2 or d != 0):
                which = "mean_var"      # (the reference's norm_stat takes dim 0 of a 2-D tensor only, norm_ops_kernel.cu:8; this host takes any - a superset, not compared)
            prog.append(("moments", which, i, d))
            shp = list(s["shape"])
            shp[d] = 1
            push(shp, "f", False, True)
            push(shp, "f", False, True)
        elif op == "convert":
            to = str(rng.choice(["half", "bfloat16", "float"]))
            prog.append(("convert", i, to))
            push(s["shape"], "f", False if to == "float" else to, False, s["off"])    # (a conversion to the tensor's own type may hand back the tensor itself: never view() it)
        elif op == "cat":
            d = int(rng.integers(0, nd))
            parts = [i]
            for _ in range(int(rng.integers(1, 3))):
                shp = list(s["shape"])
                shp[d] = int(rng.choice([1, 2, 5]))
                dt = {"f": "f4", "i": "q"}[s["kind"]]
                j = new(tuple(shp), dt)
                if s["h"]:
                    prog.append(("convert", j, s["h"]))
                    push(shp, "f", s["h"], True)
                    j = len(shadow) - 1
                parts.append(j)
            prog.append(("cat", parts, d))
            shp = list(s["shape"])
            shp[d] = sum(shadow[p]["shape"][d] for p in parts)
            push(shp, s["kind"], s["h"], True)
        elif op == "split":
            d = int(rng.integers(0, nd))
            n = s["shape"][d]
            if n < 2:
                continue
            a = int(rng.integers(1, n))
            prog.append(("split", i, [a, n - a], d))
            for k, part in enumerate((a, n - a)):
                shp = list(s["shape"])
                shp[d] = part
                push(shp, s["kind"], s["h"], False, s["off"] or k > 0)
        elif op == "sort":
            d = int(rng.integers(0, nd))
            prog.append(("sort", i, d, bool(rng.integers(0, 2))))
            push(s["shape"], s["kind"], s["h"], True)
            push(s["shape"], "i", False, True)
        elif op == "topk":
            d = int(rng.integers(0, nd))
            k = int(rng.integers(1, s["shape"][d] + 1))
            prog.append(("topk", i, k, d, bool(rng.integers(0, 2))))
            shp = list(s["shape"])
            shp[d] = k
            push(shp, s["kind"], s["h"], True)
            push(shp, "i", False, True)
        elif op == "iput":
            # index_put_ into a FRESH tensor (test_tensor.py:262-284): one int64 index array per dimension, distinct targets (a repeated target is a write race in
            # both hosts), values of the tensor's own dtype
            if s.get("dt") is None or s["off"] or not s["contig"]:
                continue
            n = int(np.prod(s["shape"]))
            m = int(rng.integers(1, min(n, 40) + 1))
            flat = rng.choice(n, size=m, replace=False)
            idx = np.unravel_index(flat, s["shape"])
            vals = (rng.uniform(-9, 9, size=m) if s["kind"] == "f" else rng.integers(-9, 9, size=m)).astype(s["dt"])
            prog.append(("iput", i, [np.asarray(x).astype("q") for x in idx], vals))
        elif op == "gemm":
            dt = str(rng.choice(FLOATS))
            M, K, N = (int(rng.choice([1, 3, 16, 33, 64, 100, 130])) for _ in range(3))
            prog.append(("gemm", rng.uniform(-1, 1, (M, K)).astype(dt), rng.uniform(-1, 1, (K, N)).astype(dt), float(rng.choice([1.0, 0.5, -2.0]))))
            push((M, N), "f", False, True)
        elif op == "attn":
            B, H_, Sq = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.choice([1, 5, 16, 33, 64, 65]))
            Skv, D = int(rng.choice([Sq, Sq, Sq + 3, max(1, Sq // 2)])), int(rng.choice([8, 16, 33, 64]))
            # (its output stays out of the pool: the two hosts reach DIFFERENT kernels - see the comparison - and a last-bit difference would travel into everything made from it)
            prog.append(("attn", rng.uniform(-2, 2, (B, H_, Sq, D)).astype("f4"), rng.uniform(-2, 2, (B, H_, Skv, D)).astype("f4"),
                         rng.uniform(-2, 2, (B, H_, Skv, D)).astype("f4")))
        elif op == "autograd":
            # a random DAG of additions over three leaves of one shape (the reference's autograd knows + only: binary_ops.cpp:18-46), nodes reused (fan-in),
            # backward from the root with an explicit gradient; the leaves' accumulated gradients come back
            shp = rand_shape()
            leaves = [rng.uniform(-3, 3, shp).astype("f4") for _ in range(3)]
            req = [bool(rng.integers(0, 2)) for _ in range(3)]
            if not any(req):
                req[0] = True
            # (an INTERMEDIATE node is used once: the reference's first pass walks a shared node once per path and over-counts the fan-in of everything beneath
            #  it - tensor.cpp:91-103 has no visited set - so leaves under a shared sum never become ready and keep no gradient; this host counts edges)
            avail, nodes, edges = [0, 1, 2], 3, []
            for _ in range(int(rng.integers(2, 7))):
                a, b = int(rng.choice(avail)), int(rng.choice(avail))
                if a >= 3 and a == b:
                    b = int(rng.integers(0, 3))
                for x in (a, b):
                    if x >= 3:
                        avail.remove(x)
                edges.append((a, b))
                avail.append(nodes)
                nodes += 1
            while len([x for x in avail if x >= 3]) > 1:      # one root: fold what is left
                xs = [x for x in avail if x >= 3][:2]
                for x in xs:
                    avail.remove(x)
                edges.append((xs[0], xs[1]))
                avail.append(nodes)
                nodes += 1
            prog.append(("autograd", leaves, req, edges, rng.uniform(-1, 1, shp).astype("f4")))
            for r in req:
                if r:
                    push(shp, "f", False, True)
        elif op == "invalid":
            # a call the API must REFUSE (CHECK_FAIL in the reference): both hosts raise, nothing changes. Only refusals the reference makes with a check are drawn -
            # not the ones it leaves to chance (an out-of-range select index reads past the tensor there; sum(dim) with dim > rank is ACCEPTED there - it reads the
            # zero padding of its shape array - and refused here: 89 of 89 such calls in a 3000-program run, the only disagreement among the invalid calls)
            what = str(rng.choice(["permute_dup", "permute_count", "view_numel", "binary_shape", "cat_shape", "inplace_grow", "view_two_neg"]))
            if what == "permute_dup" and nd >= 2:
                prog.append(("bad", "permute", i, [0] * nd))
            elif what == "permute_count":
                prog.append(("bad", "permute", i, list(range(nd + 1))))
            elif what == "view_numel" and s["contig"]:
                prog.append(("bad", "view", i, [int(np.prod(s["shape"])) + 1]))
            elif what == "view_two_neg" and s["contig"]:
                prog.append(("bad", "view", i, [-1, -1]))
            elif what == "binary_shape":
                shp = list(s["shape"])
                shp[-1] = shp[-1] + 1 if shp[-1] > 1 else 2
                if shp[-1] != s["shape"][-1] and s["shape"][-1] != 1:
                    j = new(tuple(shp), "f4")
                    prog.append(("bad", "binary", i, j))
            elif what == "cat_shape" and nd >= 2:
                shp = list(s["shape"])
                shp[0] += 1
                j = new(tuple(shp), {"f": "f4", "i": "q"}[s["kind"]])
                if not s["h"]:
                    prog.append(("bad", "cat", [i, j], nd - 1))
            elif what == "inplace_grow" and any(x == 1 for x in s["shape"]):
                shp = [3 if x == 1 else x for x in s["shape"]]
                j = new(tuple(shp), "f4")
                prog.append(("bad", "inplace", i, j))
        elif op == "handle":
            # copy.copy / copy.deepcopy: BOTH hand back another handle on the same implementation object in the reference (register.cpp:89-90; test_tensor.py:70-84)
            prog.append(("handle", i, bool(rng.integers(0, 2))))
            shadow.append(dict(s))
        elif op == "zeros":
            shp = rand_shape()
            dtn = str(rng.choice(["float", "double", "int", "long", "half", "bfloat16"]))
            prog.append(("zeros", list(shp), dtn, bool(rng.integers(0, 2))))
            push(shp, "i" if dtn in ("int", "long") else "f", dtn if dtn in ("half", "bfloat16") else False, True)
        elif op == "fill":
            prog.append(("fill", i, float(rng.choice([0, 1, -2, 3]))))
    return prog


def run(kf, prog):
    """Execute; returns (per-step status list, final arrays). A step that raises leaves placeholders (None) for the tensors it would have made."""
    ts, status, side = [], [], []
    ops = {"+": lambda a, b: a + b, "-": lambda a, b: a - b, "*": lambda a, b: a * b, "/": lambda a, b: a / b}

    def iop(sym, a, b):
        if sym == "+":
            a += b
        elif sym == "-":
            a -= b
        elif sym == "*":
            a *= b
        else:
            a /= b

    for ins in prog:
        made = {"new": 1, "permute": 1, "getitem": 1, "contiguous": 1, "view": 1, "binary": 1, "scalar": 1, "reduce": 1, "moments": 2, "convert": 1, "cat": 1,
                "split": 2, "inplace": 0, "inplace_scalar": 0, "fill": 0, "sort": 2, "topk": 2, "iput": 0, "gemm": 1, "attn": 0, "bad": 0, "handle": 1, "zeros": 1}.get(str(ins[0]))
        if made is None:
            made = sum(ins[2])   # autograd: one gradient per leaf that requires one
        try:
            k = ins[0]
            if k == "new":
                out = [kf.from_numpy(ins[1], 0)]
            elif k == "permute":
                out = [ts[ins[1]].permute(*ins[2])]
            elif k == "getitem":
                key = tuple(slice(*x) if isinstance(x, tuple) else x for x in ins[2])
                out = [ts[ins[1]][key]]
            elif k == "contiguous":
                out = [ts[ins[1]].contiguous()]
            elif k == "view":
                out = [ts[ins[1]].view(*ins[2])]
            elif k == "binary":
                out = [ops[ins[1]](ts[ins[2]], ts[ins[3]])]
            elif k == "scalar":
                out = [ops[ins[1]](ts[ins[2]], ins[3])]
            elif k == "inplace":
                iop(ins[1], ts[ins[2]], ts[ins[3]])
                out = []
            elif k == "inplace_scalar":
                iop(ins[1], ts[ins[2]], ins[3])
                out = []
            elif k == "reduce":
                out = [getattr(ts[ins[2]], ins[1])(ins[3])]
            elif k == "moments":
                t = ts[ins[2]]
                r = t.norm_stat(ins[3]) if ins[1] == "norm_stat" else t.mean_var(ins[3], ins[1] == "mean_std")
                out = [r[0], r[1]]
            elif k == "convert":
                out = [getattr(ts[ins[1]], ins[2])()]
            elif k == "cat":
                out = [kf.cat([ts[p] for p in ins[1]], ins[2])]
            elif k == "split":
                out = list(ts[ins[1]].split(ins[2], ins[3]))
                assert len(out) == 2
            elif k == "fill":
                ts[ins[1]].fill_(ins[2])
                out = []
            elif k == "handle":
                import copy
                out = [copy.deepcopy(ts[ins[1]]) if ins[2] else copy.copy(ts[ins[1]])]
            elif k == "zeros":
                z = kf.zeros(ins[1], getattr(kf, ins[2]), 0)
                out = [kf.empty_like(z).fill_(0.0) + z if ins[3] and ins[2] in ("float", "double") else z]   # (empty_like: its shape and dtype; its bytes are whatever the allocator had)
            elif k == "bad":
                if ins[1] == "permute":
                    ts[ins[2]].permute(*ins[3])
                elif ins[1] == "view":
                    ts[ins[2]].view(*ins[3])
                elif ins[1] == "binary":
                    ts[ins[2]] + ts[ins[3]]
                elif ins[1] == "cat":
                    kf.cat([ts[p] for p in ins[2]], ins[3])
                elif ins[1] == "reduce":
                    ts[ins[2]].sum(ins[3])
                elif ins[1] == "inplace":
                    t = ts[ins[2]]
                    t += ts[ins[3]]
                out = []
            elif k == "sort":
                out = list(ts[ins[1]].sort(ins[2], ins[3]))
            elif k == "topk":
                out = list(ts[ins[1]].topk(ins[2], ins[3], ins[4]))
            elif k == "iput":
                ts[ins[1]].index_put_([kf.from_numpy(x, 0) for x in ins[2]], kf.from_numpy(ins[3], 0))
                out = []
            elif k == "gemm":
                out = [kf.gemm(kf.from_numpy(ins[1], 0), kf.from_numpy(ins[2], 0), ins[3], 0.0)]
            elif k == "attn":
                side.append(kf.causal_attention(*(kf.from_numpy(x, 0) for x in ins[1:4])).numpy())
                out = []
            elif k == "autograd":
                nodes = [kf.from_numpy(x, 0) for x in ins[1]]
                for t, r in zip(nodes, ins[2]):
                    t.set_requires_grad(r)
                for a, b in ins[3]:
                    nodes.append(nodes[a] + nodes[b])
                nodes[-1].backward(kf.from_numpy(ins[4], 0))
                out = []
                for t, r in zip(nodes[:3], ins[2]):
                    if r:
                        g = t.grad()
                        out.append(g if g.defined() else kf.zeros(list(ins[1][0].shape), t.dtype(), 0))   # (a leaf the root does not reach: no gradient in either host)
            ts.extend(out)
            status.append("ok")
        except Exception as e:  # noqa: BLE001 - whatever the module raises: the OTHER module must raise at the same step
            ts.extend([None] * made)
            status.append(type(e).__name__ + ": " + " ".join(str(e).split())[:240])
    finals = []
    for t in ts:
        if t is None:
            finals.append(None)
            continue
        try:
            dt = str(t.dtype())
            # the text of the tensor ITSELF: shape, STRIDES, storage offset, dtype and the first twelve entries of every dim, as print(t) shows them - small tensors only
            # (every printed element is a device-to-host copy)
            rep = repr(t).replace(",\x08]", "]") if t.numel() <= 150 else None
            # who shares what: the number of tensors on this tensor's storage and on its implementation object (test_tensor.py:70-84) - equal counts on both hosts mean the
            # same aliasing structure (which results are views of which, what contiguous() / a conversion to the own type / split hand back) - and item() of the first element
            first = t.item([0] * t.dim()) if t.numel() > 0 and "half" not in dt.lower() and "bf" not in dt.lower() else None
            finals.append([dt, tuple(t.sizes()), None, rep, (t.storage_ref_count(), t.impl_ref_count(), t.numel(), t.dim(), first)])
        except Exception as e:  # noqa: BLE001
            finals.append(("raised", type(e).__name__))
    for k, t in enumerate(ts):      # (the values last: contiguous() makes copies, which would count as sharers above)
        if t is None or finals[k][0] == "raised":
            continue
        try:
            c = t.contiguous()
            if "half" in finals[k][0].lower() or "bf" in finals[k][0].lower():
                c = c.float()          # (16-bit values are exactly representable: the float image carries the same bits)
            finals[k][2] = c.numpy()
            del c
        except Exception as e:  # noqa: BLE001
            finals[k] = ("raised", type(e).__name__)
    return status, finals, side


def origin_args(ins):
    return (str(ins[0]),) + tuple(x.shape if isinstance(x, np.ndarray) else (x if not isinstance(x, list) else "[...]") for x in ins[1:])


def origin(prog, n):
    """The instruction that made tensor n (for the failure message)."""
    made = {"new": 1, "permute": 1, "getitem": 1, "contiguous": 1, "view": 1, "binary": 1, "scalar": 1, "reduce": 1, "moments": 2, "convert": 1, "cat": 1, "split": 2,
            "sort": 2, "topk": 2, "gemm": 1, "handle": 1, "zeros": 1}
    k = 0
    for ins in prog:
        m = sum(ins[2]) if str(ins[0]) == "autograd" else made.get(str(ins[0]), 0)
        if k <= n < k + m:
            return (str(ins[0]),) + tuple(x.shape if isinstance(x, np.ndarray) else (x if not isinstance(x, list) else "[...]") for x in ins[1:])
        k += m
    return None


def compare(ref_mod, mine_mod, seed, mark=lambda s: None):
    """One program through both hosts; raises AssertionError on the first disagreement. `mark` is told which host is about to run (the child process prints it: a
    crash is then attributable)."""
    prog = make_program(1000 + seed, steps=28 + seed % 17)
    mark("R")
    st_r, fin_r, side_r = run(ref_mod, prog)
    mark("M")
    st_m, fin_m, side_m = run(mine_mod, prog)
    mark("C")
    # causal_attention: the two hosts reach DIFFERENT kernels - the seam hands the reference host's call to kf_attn_fwd as it is (f32, any head size: the generic
    # kernel), this host's operator zero-pads onto the matrix-core kernel: the reference's own tolerance for this operator (test_nn.py:30), not equal bits
    assert len(side_r) == len(side_m)
    for a, b in zip(side_r, side_m):
        assert a.shape == b.shape and np.allclose(a, b, rtol=1e-3, atol=1e-3), "attention outputs differ beyond 1e-3"
    for n, (a, b, ins) in enumerate(zip(st_r, st_m, prog)):
        assert (a == "ok") == (b == "ok"), f"step {n} {origin_args(ins)}: reference host {a}, this host {b}"
    assert len(fin_r) == len(fin_m)
    live = 0
    for n, (a, b) in enumerate(zip(fin_r, fin_m)):
        assert (a is None) == (b is None), n
        if a is None:
            continue
        if a[0] == "raised" or b[0] == "raised":
            assert a[0] == b[0], (n, a[:2], b[:2])
            continue
        assert a[1] == b[1], (n, a[1], b[1])
        assert a[2].dtype == b[2].dtype and a[2].shape == b[2].shape, (n, a[2].dtype, b[2].dtype, a[2].shape, b[2].shape)
        assert a[4][:4] == b[4][:4] and (a[4][4] == b[4][4] or a[4][4] != a[4][4]), f"tensor {n}: (storage_ref_count, impl_ref_count, numel, dim, item(0...)) {a[4]} on the reference host, {b[4]} on this one (made by {origin(prog, n)})"
        assert a[3] == b[3], f"tensor {n}: print(t) differs (made by {origin(prog, n)}):\n{a[3]}\n--- this host:\n{b[3]}"
        if not np.array_equal(a[2].view(np.uint8), b[2].view(np.uint8)):
            bad = np.argwhere(a[2] != b[2])
            i = tuple(bad[0]) if len(bad) else ()
            raise AssertionError(f"tensor {n}: the two hosts disagree (dtype {a[0]}, shape {a[1]}): {len(bad)} of {a[2].size} elements, first at {i}: "
                                 f"reference host {a[2][i] if len(bad) else '?'}, this host {b[2][i] if len(bad) else '?'}; made by {origin(prog, n)}")
        live += 1
    assert live >= 1, st_r   # (a program whose early step both hosts refuse cascades: what is left must still agree)


def child_main(first, last):
    """python tests/test_gpu_host_diff_fuzz.py FIRST LAST: seeds FIRST..LAST-1 in THIS process, one line per event on stdout:
    'S seed' start, 'R' / 'M' / 'C' the reference host / this host / the comparison is about to run, 'OK seed' or 'DIFF seed message'."""
    sys.path.insert(0, str(REFDIR))
    import kfunca as ref_mod
    for seed in range(first, last):
        print("S", seed, flush=True)
        try:
            compare(ref_mod, kfunca_amd, seed, mark=lambda s: print(s, flush=True))
            print("OK", seed, flush=True)
        except AssertionError as e:
            print("DIFF", seed, " ".join(str(e).split())[:900], flush=True)


def test_random_programs_give_the_same_bits_on_both_hosts():
    """KF_DIFF_FUZZ_SEEDS programs (default 1000) through both hosts in a CHILD process. The reference's host half reads uninitialised and freed memory on some of these
    programs (with MALLOC_PERTURB_ set it dies inside the first fifty, alone, whatever the operations; this host runs 8000 of them under MALLOC_CHECK_=3 +
    MALLOC_PERTURB_ - tools/scratch/diff_fuzz_one_host.py, profiles/r06_host_diff_fuzz.txt), and once, with 8000 collected test items, it took the whole pytest process
    with it. So: a child that dies while the REFERENCE host is running costs that one program (at most 5 % of them may go that way) and a fresh child continues behind it;
    a child that dies while THIS host runs, or any disagreement, fails the test."""
    import subprocess
    if not list(REFDIR.glob("kfunca*.so")):
        pytest.skip("oracle/_ref/kfunca*.so not built (python oracle/build_ref_host.py, build container only)")
    total = int(os.environ.get("KF_DIFF_FUZZ_SEEDS", "1000"))
    first, done, lost, diffs = 0, 0, [], []
    while first < total:
        res = subprocess.run([sys.executable, str(Path(__file__).resolve()), str(first), str(total)], capture_output=True, text=True, timeout=1800,
                             cwd=str(Path(__file__).resolve().parent.parent))
        seed, where = None, None
        for ln in res.stdout.splitlines():
            f = ln.split(" ", 2)
            if f[0] == "S":
                seed, where = int(f[1]), "S"
            elif f[0] in ("R", "M", "C"):
                where = f[0]
            elif f[0] == "OK":
                done += 1
                where = None
            elif f[0] == "DIFF":
                diffs.append(ln)
                where = None
        if res.returncode == 0 and where is None:
            break
        assert seed is not None, f"the child did nothing (exit {res.returncode}): {res.stderr[-1500:]}"
        assert where == "R", f"the child died (exit {res.returncode}) at seed {seed} while {'THIS host' if where == 'M' else 'the harness'} was running: {res.stderr[-1500:]}"
        lost.append(seed)
        first = seed + 1
    if diffs and os.environ.get("KF_DIFF_FUZZ_LOG"):
        Path(os.environ["KF_DIFF_FUZZ_LOG"]).write_text("\n".join(diffs) + "\n")
    assert not diffs, f"{len(diffs)} of {total} programs disagree; the first: {diffs[0]}"
    assert len(lost) <= max(2, total // 20), f"the reference host died on {len(lost)} of {total} programs: {lost}"
    assert done + len(lost) == total, (done, lost, total)
    print(f"{done} of {total} random programs: same bits and same refusals on both hosts; the reference host died on {len(lost)}: {lost}")


if __name__ == "__main__":
    child_main(int(sys.argv[1]), int(sys.argv[2]))
