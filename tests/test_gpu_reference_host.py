"""-m gpu: the REFERENCE's own host half running on this repository's device library.

oracle/_ref/kfunca*.so is built in the build container by oracle/build_ref_host.py from the reference's sources where they lie
(src/core/*.cpp + src/register.cpp, unmodified) + docs/seam.cpp, linked against kfunca_amd/libkfunca_hip.so - the reference's
Tensor / TensorIterator / allocator / autograd / pybind11 module with our HIP kernels behind its device seam.
It travels to the GPU box as a built object (the reference tree does not); without it these tests are skipped.

Every case below is one of the reference's own test cases (test/test_tensor.py, test_gemm.py, test_nn.py - cited per case), restated
on seeded inputs, written ONCE against the module API the reference defines and run through BOTH modules:
  * reference host + our kernels must meet the reference's own bar (its numpy / torch-CPU expressions, committed as tests/golden/*.npz);
  * our host core (kfunca_amd) must give the SAME BITS: both hosts build their own TensorIterator geometry, choose their own output
    allocation and view strides, and hand the device library a kf_iter_desc - equal bits mean our TensorIterator / Tensor restatement
    drives the kernels exactly as the reference's does (rows a1-a3, a6 of SURVEY.md section 8 pinned against the reference itself).
"""
import copy
import sys
from pathlib import Path

import numpy as np
import pytest

import kfunca_amd
from tests.helpers import assert_close, golden, regen, uni

pytestmark = pytest.mark.gpu
REFDIR = Path(__file__).resolve().parent.parent / "oracle" / "_ref"


@pytest.fixture(scope="module")
def ref():
    if not list(REFDIR.glob("kfunca*.so")):
        pytest.skip("oracle/_ref/kfunca*.so not built (python oracle/build_ref_host.py, build container only)")
    sys.path.insert(0, str(REFDIR))
    import kfunca
    assert Path(kfunca.__file__).parent == REFDIR
    return kfunca


def npy(t):
    return t.contiguous().numpy()


def both(ref, case):
    """Run `case(module)` through the reference's module and ours; returns (reference-host results, ours)."""
    return case(ref), case(kfunca_amd)


def same_bits(a, b, what):
    assert len(a) == len(b)
    for i, (x, y) in enumerate(zip(a, b)):
        assert x.shape == y.shape and x.dtype == y.dtype and np.array_equal(x.view(np.uint8), y.view(np.uint8)), f"{what}[{i}]: the two hosts disagree"


def test_add_and_promotion(ref):  # test_tensor.py:15-27, test/core/test_tensor.cpp:10-23
    g, s = golden("elementwise"), golden("shape_ops")

    def case(kf):
        out = []
        for i in range(3):
            a = kf.from_numpy(g[f"add{i}_a"], 0)
            out.append(npy(a + a))
            out.append(npy(kf.from_numpy(g[f"promo{i}_a"], 0) + kf.from_numpy(g[f"promo{i}_b"], 0)))
        t = kf.from_numpy(s["int_x"], 0)
        out.append(npy(t + t))
        return out
    r, m = both(ref, case)
    for i in range(3):
        assert np.array_equal(r[2 * i], g[f"add{i}_out"]) and np.array_equal(r[2 * i + 1], g[f"promo{i}_out"])  # fp32 add: bit-exact vs numpy
    assert np.array_equal(r[6], s["int_out"])
    same_bits(r, m, "add")


def test_inplace_ops_with_broadcast_and_scalars(ref):  # test_tensor.py:29-68
    g = golden("elementwise")

    def case(kf):
        a, b = kf.from_numpy(g["inpl_a"], 0), kf.from_numpy(g["inpl_b"], 0)
        addr, out = a.data_ptr(), []
        for op in ("+=b", "-=b", "*=b", "/=b", "+=2", "-=3", "*=4", "/=5"):
            rhs = b if op.endswith("b") else int(op[2:])
            if op[0] == "+": a += rhs
            elif op[0] == "-": a -= rhs
            elif op[0] == "*": a *= rhs
            else: a /= rhs
            assert a.data_ptr() == addr
            out.append(npy(a))
        return out
    r, m = both(ref, case)
    for got, want in zip(r, g["inpl_steps"]):
        assert np.array_equal(got, want)
    same_bits(r, m, "in-place")


def test_refcounts(ref):  # test_tensor.py:70-84
    for kf in (ref, kfunca_amd):
        arr = np.random.default_rng(2).uniform(-10, 10, size=(3, 4)).astype(np.float32)
        x = kf.from_numpy(arr, 0)
        x_ref = x
        x_deep = copy.deepcopy(x)
        assert x.data_ptr() == x_ref.data_ptr() == x_deep.data_ptr()
        assert x.storage_ref_count() == x_deep.storage_ref_count() == 1 and x.impl_ref_count() == x_deep.impl_ref_count() == 2
        del x, x_ref
        assert x_deep.impl_ref_count() == 1


def test_broadcast_binary(ref):  # test_tensor.py:86-108
    g = golden("elementwise")

    def case(kf):
        out = []
        for i in range(3):
            a, b = kf.from_numpy(g[f"bc{i}_a"], 0), kf.from_numpy(g[f"bc{i}_b"], 0)
            out += [npy(a + b), npy(a - b), npy(a * b), npy(a / b), npy(kf.from_numpy(g[f"bc{i}_ai"], 0) * b)]
        return out
    r, m = both(ref, case)
    for i in range(3):
        for j, op in enumerate(("add", "sub", "mul", "div", "imul")):
            assert np.array_equal(r[5 * i + j], g[f"bc{i}_{op}"]), (i, op)
    same_bits(r, m, "broadcast")


def test_reduce_and_moments(ref):  # test_tensor.py:110-146
    g, gm = golden("reductions"), golden("moments")
    arr = uni(np.random.default_rng(4), (223, 23, 3213))
    (ms,) = regen(gm["ms_seed"][0], [(13, 325, 127)], gm["ms_sha"], dtype=np.float64)
    (ns,) = regen(gm["ns0_seed"][0], [tuple(int(v) for v in gm["ns0_shape"])], gm["ns0_sha"])

    def case(kf):
        x, t, out = kf.from_numpy(g["x"], 0), kf.from_numpy(arr, 0), []
        for dim in range(3):
            out += [npy(x.sum(dim)), npy(x.mean(dim)), npy(t.sum(dim)), npy(t.mean(dim))]
        mv = kf.from_numpy(ms, 0).mean_var(1, False)
        st = kf.from_numpy(ns, 0).norm_stat(0)
        return out + [npy(mv[0]), npy(mv[1]), npy(st[0]), npy(st[1])]
    r, m = both(ref, case)
    for dim in range(3):
        assert_close(r[4 * dim], g[f"sum{dim}"], rtol=1e-2, atol=1e-2)
        assert_close(r[4 * dim + 1], g[f"mean{dim}"], rtol=1e-2, atol=1e-2)
        assert_close(r[4 * dim + 2], np.sum(arr, axis=dim, keepdims=True), rtol=1e-2, atol=1e-2)
        assert_close(r[4 * dim + 3], np.mean(arr, axis=dim, keepdims=True), rtol=1e-2, atol=1e-2)
    assert_close(r[12], gm["ms_mean"], rtol=1e-2, atol=1e-2)
    assert_close(r[13], gm["ms_var"], rtol=1e-2, atol=1e-2)
    assert_close(r[14], gm["ns0_mean"])
    assert_close(r[15], gm["ns0_invstd"])
    same_bits(r, m, "reduce")  # same split plan, same fold order: reproducible sums agree to the bit


def test_convert(ref):  # test_tensor.py:148-160
    g = golden("elementwise")

    def case(kf):
        t = kf.from_numpy(g["cvt_x"], 0)
        h = t.half()
        h *= h
        bf = t.bfloat16()
        bf *= bf
        return [npy(h.float()), npy(bf.float())]
    r, m = both(ref, case)
    assert np.array_equal(r[0], g["cvt_half_sq"]) and np.array_equal(r[1], g["cvt_bf16_sq"])
    same_bits(r, m, "convert")


def test_views_cat_split_index_put(ref):  # test_tensor.py:162-167, 233-284 - bit-exact
    s = golden("shape_ops")

    def case(kf):
        out = [npy(kf.from_numpy(s["perm_x"], 0).permute(2, 1, 0, 3)), npy(kf.from_numpy(s["slice_x"], 0)[3, 3:8, 4:11:2]),
               npy(kf.from_numpy(s["view_x"], 0).view(5, -1, 23).contiguous() + 1),
               npy(kf.cat([kf.from_numpy(s[f"cat_{k}"], 0) for k in "abc"], 1))]
        out += [npy(p) for p in kf.from_numpy(s["split_x"], 0).split([11, 13, 1], 1)]
        t = kf.from_numpy(s["iput_x"], 0)
        t.index_put_([kf.from_numpy(s["iput_i0"].astype("q"), 0), kf.from_numpy(s["iput_i1"].astype("q"), 0)], kf.from_numpy(s["iput_v"], 0))
        return out + [npy(t)]
    r, m = both(ref, case)
    for got, key in zip(r, ("perm_out", "slice_out", "view_out", "cat_out", "split_0", "split_1", "split_2", "iput_out")):
        assert np.array_equal(got, s[key]), key
    same_bits(r, m, "shape ops")


def test_backward_add_dag(ref):  # test_tensor.py:286-309
    s = golden("shape_ops")

    def case(kf):
        rng = np.random.default_rng(5)
        grad = kf.from_numpy(s["ag_grad"], 0)
        a, b, c = (kf.from_numpy(uni(rng, (2, 3)), 0) for _ in range(3))
        a.set_requires_grad(True)
        b.set_requires_grad(True)
        (((c + a) + (a + b)) + a).backward(grad)
        assert not c.grad().defined()
        return [npy(a.grad()), npy(b.grad())]
    r, m = both(ref, case)
    assert_close(r[0], s["ag_a_grad"])
    assert_close(r[1], s["ag_b_grad"])
    same_bits(r, m, "autograd")


def test_gemm(ref):  # test_gemm.py:9-17 (f64 123 x 457 x 234) + an f32 case on the matrix cores
    g = golden("gemm")
    a, b = regen(g["f64_seed"][0], [(123, 457), (457, 234)], g["f64_sha"], dtype=np.float64)

    def case(kf):
        return [npy(kf.gemm(kf.from_numpy(a, 0), kf.from_numpy(b, 0), 1.0, 0.0)), npy(kf.gemm(kf.from_numpy(g["f32_a"], 0), kf.from_numpy(g["f32_b"], 0), 1.0, 0.0))]
    r, m = both(ref, case)
    assert_close(r[0], g["f64_out"])
    assert_close(r[1], g["f32_out"], rtol=1e-4, atol=1e-4)
    same_bits(r[1:], m[1:], "gemm f32")  # (the ragged f64 case: our operator zero-pads onto the MFMA kernel, the seam calls the generic one)
    assert_close(m[0], r[0], rtol=1e-12, atol=1e-9)


def test_causal_attention(ref):  # test_nn.py:11-33: the reference's three cases, its dtype, its tolerance
    g = golden("attention")
    ins = []
    for i in range(3):
        B, H, Sq, Skv, D = (int(x) for x in g[f"fwd{i}_dims"])
        ins.append(regen(1050 + i, [(B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D)], g[f"fwd{i}_sha"]))

    def case(kf):
        return [npy(kf.causal_attention(*(kf.from_numpy(x, 0) for x in qkv))) for qkv in ins]
    r, m = both(ref, case)
    for i in range(3):
        assert_close(r[i], g[f"fwd{i}_out"], what=f"reference host, case {i}")  # rtol = atol = 1e-3 (test/common.py:6-11)
    same_bits(r[:1], m[:1], "attention")  # case 0 runs the same exact-f32 kernel under both hosts (the others: our operator pads, the seam does not)
    for i in (1, 2):
        assert_close(m[i], r[i], rtol=1e-4, atol=1e-4)


def test_sort_and_topk(ref):  # test_tensor.py:169-231
    rng = np.random.default_rng(8)
    x = rng.integers(-50, 50, (7, 33, 5)).astype(np.float32)

    def case(kf):
        t, out = kf.from_numpy(x, 0), []
        for dim in (0, 1, 2):
            for desc in (False, True):
                v, i = t.sort(dim, desc)
                out += [npy(v), npy(i)]
        v, i = t.topk(4, 1, True)
        return out + [npy(v), npy(i)]
    r, m = both(ref, case)
    k = 0
    for dim in (0, 1, 2):
        for desc in (False, True):
            order = np.argsort(-x if desc else x, axis=dim, kind="stable")
            assert np.array_equal(r[k + 1], order) and np.array_equal(r[k], np.take_along_axis(x, order, axis=dim))
            k += 2
    same_bits(r, m, "sort")


def test_device_info_and_memstat_run(ref, capfd):  # register.cpp:60-62: the two diagnostics the reference's CI calls (.github/workflows/ci.yml:39)
    ref.device_info()
    ref.memstat()
    out = capfd.readouterr().out
    assert "GBPS" in out and "TFLOPS" in out
