#!/usr/bin/env python3
"""Generate tests/golden/*.npz — golden vectors for the hot path.

The reference (xytpai/kfunca) cannot be built or imported here (nvcc + un-vendored CUTLASS), and its
tests hold no golden files: each test draws unseeded randoms and compares the GPU result with a
numpy / torch-CPU expression at run time. Those EXPRESSIONS are the reference's oracle, so the
fixtures below evaluate exactly them (cited per case) on seeded inputs. Backward cases have no
reference counterpart; they come from torch-CPU autograd.

Run in the build container:  python tests/golden/gen_golden.py
Inputs that are large are not stored: they are regenerated from the seed recorded in the file and
verified against a stored SHA-256 (numpy's PCG64 stream is stable by policy).
"""
import hashlib
import sys
from pathlib import Path

import numpy as np
import torch
import torch.nn.functional as F

OUT = Path(__file__).resolve().parent


def sha(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return np.frombuffer(h.digest(), dtype=np.uint8)


def uni(rng, shape, dtype=np.float32, lo=-10, hi=10):
    return rng.uniform(lo, hi, size=shape).astype(dtype)


def elementwise():
    d = {}
    rng = np.random.default_rng(101)
    # test_tensor.py:15-27  add fp32, and int32 + fp32 promotion (reference result dtype: float)
    for i, shape in enumerate(((2, 3), (1000,), (12, 11, 331))):
        a = uni(rng, shape)
        d[f"add{i}_a"], d[f"add{i}_out"] = a, a + a
        x, y = uni(rng, shape).astype(np.int32), uni(rng, shape)
        d[f"promo{i}_a"], d[f"promo{i}_b"] = x, y
        d[f"promo{i}_out"] = (x + y).astype(np.float32)  # numpy gives f64; reference computes in f32 (exact here)
    # test_tensor.py:29-68  in-place ops with [5,7,11] o [5,1,11] broadcast, then scalar ops
    a, b = uni(rng, (5, 7, 11)), uni(rng, (5, 1, 11))
    d["inpl_a"], d["inpl_b"] = a.copy(), b
    steps = []
    a = a.copy()
    a += b; steps.append(a.copy())
    a -= b; steps.append(a.copy())
    a *= b; steps.append(a.copy())
    a /= b; steps.append(a.copy())
    a += np.float32(2); steps.append(a.copy())
    a -= np.float32(3); steps.append(a.copy())
    a *= np.float32(4); steps.append(a.copy())
    a /= np.float32(5); steps.append(a.copy())
    d["inpl_steps"] = np.stack(steps)
    # test_tensor.py:86-108  broadcast binary, 'easy' shapes (the 1-Gi-element 'hard' ones are run
    # on the GPU box against the oracle by seeded regeneration)
    for i, (s1, s2) in enumerate((((16, 1), (1, 6)), ((62, 1, 45), (62, 6, 1)), ((23, 1, 67), (23, 27, 67)))):
        x, y = uni(rng, s1), uni(rng, s2)
        d[f"bc{i}_a"], d[f"bc{i}_b"] = x, y
        for name, f in (("add", np.add), ("sub", np.subtract), ("mul", np.multiply), ("div", np.divide)):
            d[f"bc{i}_{name}"] = f(x, y)
        xi = uni(rng, s1).astype(np.int32)
        d[f"bc{i}_ai"] = xi
        d[f"bc{i}_imul"] = (xi.astype(np.float32) * y)
    # test_tensor.py:148-160  convert to half / bf16, multiply in the narrow dtype, back to float
    x = rng.uniform(-10, 10, size=(2, 3))
    d["cvt_x"] = x
    t = torch.from_numpy(x)
    h = t.half(); d["cvt_half_bits"] = h.view(torch.int16).numpy().view(np.uint16)
    d["cvt_half_sq"] = (h.float() * h.float()).half().float().numpy()  # computed in float acc, stored as half
    bf = t.bfloat16(); d["cvt_bf16_bits"] = bf.view(torch.int16).numpy().view(np.uint16)
    d["cvt_bf16_sq"] = (bf.float() * bf.float()).bfloat16().float().numpy()
    xs = uni(rng, (4097,))  # full-range conversion check incl. odd length
    d["cvt_big"] = xs
    d["cvt_big_bf16"] = torch.from_numpy(xs).bfloat16().view(torch.int16).numpy().view(np.uint16)
    d["cvt_big_f16"] = torch.from_numpy(xs).half().view(torch.int16).numpy().view(np.uint16)
    np.savez_compressed(OUT / "elementwise.npz", **d)


def shape_ops():
    d = {}
    rng = np.random.default_rng(102)
    # test_tensor.py:162-167 permute(2,1,0,3).contiguous() — bit-exact
    x = rng.uniform(-10, 10, size=(6, 8, 24, 11))
    d["perm_x"], d["perm_out"] = x, np.ascontiguousarray(x.transpose(2, 1, 0, 3))
    # test_tensor.py:233-239 slice [3, 3:8, 4:11:2]
    x = uni(rng, (11, 15, 33, 5), lo=-10000, hi=10000)
    d["slice_x"], d["slice_out"] = x, np.ascontiguousarray(x[3, 3:8, 4:11:2])
    # test_tensor.py:241-247 view(5,-1,23).contiguous() + 1
    x = uni(rng, (5, 2, 11, 23), lo=-10000, hi=10000)
    d["view_x"], d["view_out"] = x, (torch.from_numpy(x).view(5, -1, 23).contiguous() + 1).numpy()
    # test_tensor.py:249-271 cat / split along dim 1
    a, b, c = (uni(rng, s, lo=-10000, hi=10000) for s in ((5, 11, 23), (5, 13, 23), (5, 1, 23)))
    d["cat_a"], d["cat_b"], d["cat_c"] = a, b, c
    d["cat_out"] = torch.cat([torch.from_numpy(a), torch.from_numpy(b), torch.from_numpy(c)], 1).numpy()
    x = uni(rng, (5, 25, 23), lo=-10000, hi=10000)
    d["split_x"] = x
    for i, p in enumerate(torch.from_numpy(x).split([11, 13, 1], 1)):
        d[f"split_{i}"] = p.contiguous().numpy()
    # test_tensor.py:273-284 index_put_
    x = uni(rng, (13, 15), lo=-10000, hi=10000)
    i0, i1 = np.array([0, 5, 1, 2], dtype=np.int64), np.array([0, 11, 1, 0], dtype=np.int64)
    vals = uni(rng, (4,), lo=-10000, hi=10000)
    t = torch.from_numpy(x.copy())
    t.index_put_([torch.from_numpy(i0), torch.from_numpy(i1)], torch.from_numpy(vals))
    d["iput_x"], d["iput_i0"], d["iput_i1"], d["iput_v"], d["iput_out"] = x, i0, i1, vals, t.numpy()
    # negative indices wrap once (tensor_index.h:66-68); 2-D index tensors
    x = uni(rng, (7, 9, 4))
    j0 = np.array([[0, -1], [3, 2]], dtype=np.int64)
    j1 = np.array([[8, -9], [-2, 5]], dtype=np.int64)
    j2 = np.array([[-4, 3], [1, 0]], dtype=np.int64)
    vals = uni(rng, (2, 2))
    t = torch.from_numpy(x.copy())
    t.index_put_([torch.from_numpy(j0), torch.from_numpy(j1), torch.from_numpy(j2)], torch.from_numpy(vals))
    d["iput3_x"], d["iput3_i0"], d["iput3_i1"], d["iput3_i2"], d["iput3_v"], d["iput3_out"] = x, j0, j1, j2, vals, t.numpy()
    # test/core/test_tensor.cpp:10-23 — the reference's one literal known answer: Int [3,5] doubled
    x = np.arange(15, dtype=np.int32).reshape(3, 5)
    d["int_x"], d["int_out"] = x, x + x
    # test_tensor.py:286-309 add-only autograd DAG: a.grad = 3 * grad, b.grad = grad
    g = uni(rng, (2, 3))
    d["ag_grad"], d["ag_a_grad"], d["ag_b_grad"] = g, g * 3, g
    np.savez_compressed(OUT / "shape_ops.npz", **d)


def reductions():
    d = {}
    rng = np.random.default_rng(103)
    # test_tensor.py:110-118 sum / mean over each dim (shape scaled down from [223,23,3213])
    x = uni(rng, (23, 13, 321))
    d["x"] = x
    for dim in range(3):
        d[f"sum{dim}"] = np.sum(x, axis=dim, keepdims=True)
        d[f"mean{dim}"] = np.mean(x, axis=dim, keepdims=True)
    # BASELINE config C1: fp32 1024 x 1024 add + sum — inputs regenerated from the seed
    rng1 = np.random.default_rng(1001)
    a, b = uni(rng1, (1024, 1024)), uni(rng1, (1024, 1024))
    d["c1_seed"] = np.array([1001])
    d["c1_sha"] = sha(a, b)
    d["c1_add_sha"] = sha(a + b)  # fp32 add is exact-rounded: bit-exact target
    d["c1_sum0"] = np.sum(a.astype(np.float64), axis=0, keepdims=True)
    d["c1_sum1"] = np.sum(a.astype(np.float64), axis=1, keepdims=True)
    xi = rng.integers(-100, 100, size=(17, 9, 33)).astype(np.int32)
    d["xi"] = xi
    for dim in range(3):
        d[f"isum{dim}"] = np.sum(xi, axis=dim, keepdims=True, dtype=np.int32)
    np.savez_compressed(OUT / "reductions.npz", **d)


def moments():
    d = {}
    # test_tensor.py:120-132 test_mean_std: f64 (13, 325, 127), dim 1, unbiased variance from the test's expression
    shape, dim = (13, 325, 127), 1
    rng = np.random.default_rng(106)
    arr = rng.uniform(-10, 10, size=shape)
    d["ms_seed"], d["ms_sha"] = np.array([106]), sha(arr)
    mean = np.mean(arr, axis=dim, keepdims=True)
    d["ms_mean"] = mean
    d["ms_var"] = ((arr - mean) * (arr - mean)).sum(axis=dim, keepdims=True) / (shape[dim] - 1)
    # test_tensor.py:134-146 test_norm_stat: f32 [n, m], dim 0, invstd = 1 / sqrt(biased var)
    for i, shp in enumerate(([64, 64], [1024, 2048], [4096, 4096], [4096 * 4 + 3, 4096 * 4 + 3])):
        rng = np.random.default_rng(107 + i)
        arr = uni(rng, shp)
        d[f"ns{i}_seed"], d[f"ns{i}_sha"], d[f"ns{i}_shape"] = np.array([107 + i]), sha(arr), np.array(shp)
        mean = np.mean(arr, axis=0, keepdims=True, dtype=np.float64)  # the test's f32 pairwise mean, stated exactly
        var = np.sum((arr - mean) * (arr - mean), axis=0, keepdims=True)
        d[f"ns{i}_mean"] = mean
        d[f"ns{i}_invstd"] = 1.0 / np.sqrt(var / shp[0])
    np.savez_compressed(OUT / "moments.npz", **d)


def gemm():
    d = {}
    rng = np.random.default_rng(104)
    # test_gemm.py:9-17  float64 [123,457] @ [457,234], alpha 1 beta 0 vs np.matmul
    a, b = rng.uniform(-10, 10, size=(123, 457)), rng.uniform(-10, 10, size=(457, 234))
    d["f64_seed"] = np.array([104])
    d["f64_sha"] = sha(a, b)
    d["f64_out"] = np.matmul(a, b)
    # fp32, tile-aligned and ragged, all four layouts, alpha/beta (extension of the same expression)
    a, b, c = uni(rng, (128, 64), lo=-1, hi=1), uni(rng, (64, 256), lo=-1, hi=1), uni(rng, (128, 256), lo=-1, hi=1)
    d["f32_a"], d["f32_b"], d["f32_c"] = a, b, c
    d["f32_out"] = np.matmul(a.astype(np.float64), b.astype(np.float64))
    d["f32_out_ab"] = 0.5 * d["f32_out"] + 2.0 * c.astype(np.float64)
    # backward (no reference counterpart): torch-CPU autograd of C = A @ B
    ta, tb = torch.from_numpy(a).double().requires_grad_(), torch.from_numpy(b).double().requires_grad_()
    g = uni(rng, (128, 256), lo=-1, hi=1)
    (ta @ tb).backward(torch.from_numpy(g).double())
    d["f32_g"], d["f32_da"], d["f32_db"] = g, ta.grad.numpy(), tb.grad.numpy()
    np.savez_compressed(OUT / "gemm.npz", **d)


def attention():
    d = {}
    # test_nn.py:11-33  fp32 U(-10,10) vs torch SDPA(is_causal=True); inputs regenerated from seeds
    cases = ((2, 4, 32, 256, 128), (3, 5, 64, 32, 64), (5, 16, 65, 33, 123))
    for i, (B, H, Sq, Skv, D) in enumerate(cases):
        rng = np.random.default_rng(1050 + i)
        q, k, v = uni(rng, (B, H, Sq, D)), uni(rng, (B, H, Skv, D)), uni(rng, (B, H, Skv, D))
        out = F.scaled_dot_product_attention(torch.from_numpy(q), torch.from_numpy(k), torch.from_numpy(v), is_causal=True)
        d[f"fwd{i}_dims"] = np.array([B, H, Sq, Skv, D])
        d[f"fwd{i}_sha"] = sha(q, k, v)
        d[f"fwd{i}_out"] = out.numpy()
    # backward (no reference counterpart): torch-CPU autograd of SDPA, U(-1,1)
    for i, (B, H, Sq, Skv, D) in enumerate(((2, 3, 48, 48, 64), (1, 2, 128, 128, 128), (1, 2, 40, 72, 32))):
        rng = np.random.default_rng(1060 + i)
        q, k, v, g = (uni(rng, s, lo=-1, hi=1) for s in ((B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D), (B, H, Sq, D)))
        tq, tk, tv = (torch.from_numpy(t).double().requires_grad_() for t in (q, k, v))
        out = F.scaled_dot_product_attention(tq, tk, tv, is_causal=True)
        out.backward(torch.from_numpy(g).double())
        d[f"bwd{i}_dims"] = np.array([B, H, Sq, Skv, D])
        d[f"bwd{i}_sha"] = sha(q, k, v, g)
        d[f"bwd{i}_out"] = out.detach().numpy().astype(np.float32)
        d[f"bwd{i}_dq"], d[f"bwd{i}_dk"], d[f"bwd{i}_dv"] = (t.grad.numpy().astype(np.float32) for t in (tq, tk, tv))
    np.savez_compressed(OUT / "attention.npz", **d)


def sort():
    d = {}
    # test_tensor.py:169-192 test_sort_small_slice: torch.sort(arr, dim, descending, stable=True), U(-1000, 1000) cast to
    # f32 / f64 / i32 (the i32 cast makes many duplicates: the stability cases). Small shapes stored, larger ones by digest.
    shapes = [[2, 3, 4], [23, 11, 23], [11, 23, 64], [13, 65, 1049], [5, 11, 22223]]
    n = 0
    for dt in (np.float32, np.float64, np.int32):
        for desc in (False, True):
            for dim in (2, 1, 0):
                for shape in shapes:
                    seed = 300 + n
                    arr = np.random.default_rng(seed).uniform(-1000, 1000, size=shape).astype(dt)
                    res, ind = torch.sort(torch.from_numpy(arr), dim=dim, descending=desc, stable=True)
                    res, ind = res.numpy(), ind.numpy()
                    tag = f"s{n}"
                    d[tag + "_meta"] = np.array([seed, dim, int(desc)] + shape)
                    d[tag + "_dtype"] = np.array(np.dtype(dt).str)
                    d[tag + "_sha_in"] = sha(arr)
                    if arr.size <= 6000:
                        d[tag + "_res"], d[tag + "_ind"] = res, ind
                    else:
                        d[tag + "_sha_out"] = sha(res, ind)
                    n += 1
    d["n_sort"] = np.array([n])
    # test_tensor.py:194-201 test_sort_large_slice: np.sort / np.argsort(kind='stable') of f32 (4, 1024000), axis 1
    arr = np.random.default_rng(290).uniform(-1000, 1000, size=(4, 1024000)).astype(np.float32)
    d["large_seed"], d["large_sha_in"] = np.array([290]), sha(arr)
    d["large_sha_out"] = sha(np.sort(arr, axis=1), np.argsort(arr, axis=1, kind="stable").astype(np.int64))
    # test_tensor.py:203-222 test_topk_small: torch.topk(arr, 8, dim, largest) VALUES (the test does not compare indices)
    n = 0
    for dt in (np.float32, np.float64, np.int32):
        for largest in (False, True):
            for dim in (2, 1, 0):
                for shape in ([13, 65, 1049], [33, 22, 22223]):
                    seed = 500 + n
                    arr = np.random.default_rng(seed).uniform(-100000, 100000, size=shape).astype(dt)
                    res, _ = torch.topk(torch.from_numpy(arr), 8, dim=dim, largest=largest)
                    tag = f"t{n}"
                    d[tag + "_meta"] = np.array([seed, dim, int(largest)] + shape)
                    d[tag + "_dtype"] = np.array(np.dtype(dt).str)
                    d[tag + "_sha_in"] = sha(arr)
                    d[tag + "_sha_out"] = sha(res.numpy())
                    n += 1
    d["n_topk"] = np.array([n])
    # test_tensor.py:224-231 test_topk_large: f32 (4, 1024000), k in {2049, 22223}, largest, values only
    for i, k in enumerate((2049, 22223)):
        arr = np.random.default_rng(291 + i).uniform(-10000, 10000, size=(4, 1024000)).astype(np.float32)
        res, _ = torch.topk(torch.from_numpy(arr), k, dim=1, largest=True)
        d[f"tl{i}_meta"], d[f"tl{i}_sha_in"], d[f"tl{i}_sha_out"] = np.array([291 + i, k]), sha(arr), sha(res.numpy())
    np.savez_compressed(OUT / "sort.npz", **d)


def norms():
    """rms_norm / layer_norm (README.md:28 roadmap item; no reference test exists): torch-CPU F.rms_norm / F.layer_norm in float64
    on seeded f32 inputs, forward + autograd backward. Shapes: one wave per row (cols 64..1024), one block per row (4096), a
    ragged row length (331: generic kernel) and a 3-D input (leading dims flatten into rows)."""
    d = {}
    for i, shp in enumerate(((37, 64), (5, 7, 256), (16, 1024), (5, 4096), (23, 331))):
        rng = np.random.default_rng(130 + i)
        x, g = uni(rng, shp, lo=-3, hi=3), uni(rng, shp, lo=-1, hi=1)
        w, b = uni(rng, shp[-1:], lo=0.5, hi=1.5), uni(rng, shp[-1:], lo=-1, hi=1)
        d[f"n{i}_x"], d[f"n{i}_g"], d[f"n{i}_w"], d[f"n{i}_b"] = x, g, w, b
        for kind in ("rms", "layer"):
            tx, tw, tb = (torch.from_numpy(a).double().requires_grad_() for a in (x, w, b))
            if kind == "rms":
                y = F.rms_norm(tx, (shp[-1],), tw, eps=1e-5)
            else:
                y = F.layer_norm(tx, (shp[-1],), tw, tb, eps=1e-5)
            y.backward(torch.from_numpy(g).double())
            f32 = lambda t: t.detach().numpy().astype(np.float32)  # noqa: E731  (evaluated in f64, stored in f32: 6e-8 relative)
            d[f"n{i}_{kind}_y"], d[f"n{i}_{kind}_dx"], d[f"n{i}_{kind}_dw"] = f32(y), f32(tx.grad), f32(tw.grad)
            if kind == "layer":
                d[f"n{i}_layer_db"] = f32(tb.grad)
    np.savez_compressed(OUT / "norms.npz", **d)


if __name__ == "__main__":
    torch.manual_seed(0)
    for f in (elementwise, shape_ops, reductions, moments, gemm, attention, sort, norms):
        if len(sys.argv) > 1 and f.__name__ not in sys.argv[1:]:
            continue
        f()
        print("wrote", f.__name__)
    sys.exit(0)
