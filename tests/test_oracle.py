"""Pins the CPU oracle (oracle/oracle.c) against the golden vectors in tests/golden, which hold the
reference's own test-oracle expressions (numpy / torch-CPU) evaluated on seeded inputs.
No GPU needed."""
import numpy as np
import pytest

from oracle import oracle as O
from tests.helpers import assert_close, golden, regen, uni

OPS = {"add": O.ADD, "sub": O.SUB, "mul": O.MUL, "div": O.DIV}


def test_promotion_table():
    # tensor_iterator.cpp:32-44
    assert O.promote(O.I32, O.F32) == O.F32
    assert O.promote(O.F16, O.BF16) == O.BF16       # reference quirk: Half + BFloat16 -> BFloat16
    assert O.promote(O.U8, O.I8) == O.I8
    assert O.promote(O.BOOL, O.U8) == O.U8
    assert O.promote(O.I64, O.F16) == O.F16
    assert O.promote(O.F32, O.F64) == O.F64


def test_add_and_promotion_bit_exact():
    g = golden("elementwise")
    for i in range(3):
        a = g[f"add{i}_a"]
        assert np.array_equal(O.binary(O.ADD, a, a), g[f"add{i}_out"])
        out = O.binary(O.ADD, g[f"promo{i}_a"], g[f"promo{i}_b"])
        assert out.dtype == np.float32
        assert np.array_equal(out, g[f"promo{i}_out"])


def test_inplace_chain():
    g = golden("elementwise")
    a, b = g["inpl_a"].copy(), g["inpl_b"]
    steps = g["inpl_steps"]
    for k, op in enumerate((O.ADD, O.SUB, O.MUL, O.DIV)):
        O.binary_out(op, a, b, a)
        assert np.array_equal(a, steps[k]), k
    for k, (op, s) in enumerate(((O.ADD, 2), (O.SUB, 3), (O.MUL, 4), (O.DIV, 5))):
        # register.cpp:172-206: scalar ops materialise empty_like(self).fill_(s)
        sc = O.fill(np.empty_like(a), s)
        O.binary_out(op, a, sc, a)
        assert np.array_equal(a, steps[4 + k]), k


def test_broadcast_binary():
    g = golden("elementwise")
    for i in range(3):
        a, b = g[f"bc{i}_a"], g[f"bc{i}_b"]
        for name, op in OPS.items():
            assert np.array_equal(O.binary(op, a, b), g[f"bc{i}_{name}"]), (i, name)
        assert np.array_equal(O.binary(O.MUL, g[f"bc{i}_ai"], b), g[f"bc{i}_imul"])


def test_convert_half_bf16():
    g = golden("elementwise")
    x = g["cvt_x"]
    h = O.convert(x, O.F16)
    assert np.array_equal(h.view(np.uint16), g["cvt_half_bits"])
    assert np.array_equal(O.convert(O.binary(O.MUL, h, h), O.F32), g["cvt_half_sq"])
    bf = O.convert(x, O.BF16)
    assert np.array_equal(bf, g["cvt_bf16_bits"])
    sq = O.binary(O.MUL, bf, bf, a_code=O.BF16, b_code=O.BF16)
    assert np.array_equal(O.convert(sq, O.F32, src_code=O.BF16), g["cvt_bf16_sq"])
    assert np.array_equal(O.convert(g["cvt_big"], O.BF16), g["cvt_big_bf16"])
    assert np.array_equal(O.convert(g["cvt_big"], O.F16).view(np.uint16), g["cvt_big_f16"])


def test_integer_and_bool_semantics():
    a = np.array([127, -128, 5, -7], dtype=np.int8)
    b = np.array([1, -1, 2, 2], dtype=np.int8)
    assert np.array_equal(O.binary(O.ADD, a, b), np.array([-128, 127, 7, -5], dtype=np.int8))  # wraps
    assert np.array_equal(O.binary(O.DIV, a, b), np.array([127, -128, 2, -3], dtype=np.int8))  # truncates; -128/-1 wraps
    t, f = np.array([True, True, False, False]), np.array([True, False, True, False])
    assert np.array_equal(O.binary(O.ADD, t, f), t | f)
    assert np.array_equal(O.binary(O.SUB, t, f), t ^ f)
    assert np.array_equal(O.binary(O.MUL, t, f), t & f)
    assert np.array_equal(O.fill(np.empty(3, dtype=np.int32), -2.7), np.array([-2, -2, -2], dtype=np.int32))


def test_views_copy_bit_exact():
    g = golden("shape_ops")
    x = g["perm_x"]
    dst = np.empty(g["perm_out"].shape, dtype=x.dtype)
    assert np.array_equal(O.copy(x.transpose(2, 1, 0, 3), dst), g["perm_out"])
    x = g["slice_x"]
    v = x[3, 3:8, 4:11:2]
    assert np.array_equal(O.copy(v, np.empty(v.shape, dtype=x.dtype)), g["slice_out"])
    x = g["view_x"].reshape(5, -1, 23)
    one = O.fill(np.empty_like(x), 1)
    assert np.array_equal(O.binary(O.ADD, x, one), g["view_out"])
    # cat = empty + narrow().copy_() per input (tensor_shape.cpp:41-70)
    out = np.empty(g["cat_out"].shape, dtype=np.float32)
    off = 0
    for k in "abc":
        t = g[f"cat_{k}"]
        O.copy(t, out[:, off:off + t.shape[1], :])
        off += t.shape[1]
    assert np.array_equal(out, g["cat_out"])
    x, off = g["split_x"], 0
    for i, n in enumerate((11, 13, 1)):
        v = x[:, off:off + n, :]
        assert np.array_equal(O.copy(v, np.empty(v.shape, dtype=x.dtype)), g[f"split_{i}"])
        off += n
    assert np.array_equal(O.binary(O.ADD, g["int_x"], g["int_x"]), g["int_out"])  # test/core/test_tensor.cpp:10-23


def test_index_put():
    g = golden("shape_ops")
    x = g["iput_x"].copy()
    assert np.array_equal(O.index_put(x, [g["iput_i0"], g["iput_i1"]], g["iput_v"]), g["iput_out"])
    x = g["iput3_x"].copy()
    assert np.array_equal(O.index_put(x, [g["iput3_i0"], g["iput3_i1"], g["iput3_i2"]], g["iput3_v"]), g["iput3_out"])


def test_reductions():
    g = golden("reductions")
    x = g["x"]
    for dim in range(3):
        # reference tolerance for reductions: 1e-2 (test_tensor.py:118); the oracle is far inside it
        assert_close(O.reduce(O.SUM, x, dim), g[f"sum{dim}"], rtol=1e-5, atol=1e-4, what=f"sum{dim}")
        assert_close(O.reduce(O.MEAN, x, dim), g[f"mean{dim}"], rtol=1e-5, atol=1e-5, what=f"mean{dim}")
        assert np.array_equal(O.reduce(O.SUM, g["xi"], dim), g[f"isum{dim}"])
    # integer mean: factor = nout / numel in the integer dtype == 0 (reduce_ops_kernel.cu:49-53)
    assert not O.reduce(O.MEAN, g["xi"], 1).any()
    a, b = regen(g["c1_seed"][0], [(1024, 1024), (1024, 1024)], g["c1_sha"])
    from tests.helpers import sha
    assert np.array_equal(sha(O.binary(O.ADD, a, b)), g["c1_add_sha"])  # config C1 add: bit-exact
    assert_close(O.reduce(O.SUM, a, 0), g["c1_sum0"], rtol=1e-6, atol=1e-3, what="c1 sum0")
    assert_close(O.reduce(O.SUM, a, 1), g["c1_sum1"], rtol=1e-6, atol=1e-3, what="c1 sum1")


def test_moments():
    """orc_moments vs the reference tests' expressions: test_mean_std (test_tensor.py:120-132, f64, unbiased) and
    test_norm_stat (test_tensor.py:134-146, f32, invstd of the biased variance)."""
    g = golden("moments")
    (arr,) = regen(g["ms_seed"][0], [(13, 325, 127)], g["ms_sha"], dtype=np.float64)
    var, mean = O.moments(0, arr, 1)
    assert_close(mean, g["ms_mean"], rtol=1e-12, atol=1e-13, what="mean")
    assert_close(var, g["ms_var"], rtol=1e-12, atol=0, what="var")
    std, _ = O.moments(1, arr, 1)
    assert_close(std, np.sqrt(g["ms_var"]), rtol=1e-12, atol=0, what="std")
    for i in range(3):  # the 16387^2 case is regenerated in the -m gpu suite only (1 GiB)
        shp = tuple(int(v) for v in g[f"ns{i}_shape"])
        (x,) = regen(g[f"ns{i}_seed"][0], [shp], g[f"ns{i}_sha"])
        inv, mean = O.moments(2, x, 0, eps=1e-12)
        assert mean.dtype == np.float32 and mean.shape == (1, shp[1])
        assert_close(mean, g[f"ns{i}_mean"], rtol=1e-6, atol=1e-6, what="norm_stat mean")
        assert_close(inv, g[f"ns{i}_invstd"], rtol=1e-5, atol=0, what="norm_stat invstd")


def test_gemm():
    g = golden("gemm")
    a, b = regen(g["f64_seed"][0], [(123, 457), (457, 234)], g["f64_sha"], dtype=np.float64)
    assert_close(O.gemm(a, b), g["f64_out"], what="f64 gemm (test_gemm.py:9-17)")
    a, b, c = g["f32_a"], g["f32_b"], g["f32_c"]
    assert_close(O.gemm(a, b), g["f32_out"], rtol=1e-4, atol=1e-4, what="f32 NN")
    assert_close(O.gemm(a.T.copy(), b, trans_a=True), g["f32_out"], rtol=1e-4, atol=1e-4, what="f32 TN")
    assert_close(O.gemm(a, b.T.copy(), trans_b=True), g["f32_out"], rtol=1e-4, atol=1e-4, what="f32 NT")
    assert_close(O.gemm(a, b, alpha=0.5, beta=2.0, c=c), g["f32_out_ab"], rtol=1e-4, atol=1e-4, what="alpha/beta")
    gg = g["f32_g"]  # dA = dC B^T, dB = A^T dC
    assert_close(O.gemm(gg, b, trans_b=True), g["f32_da"], rtol=1e-4, atol=1e-4, what="dA")
    assert_close(O.gemm(a, gg, trans_a=True), g["f32_db"], rtol=1e-4, atol=1e-4, what="dB")
    # bf16 inputs: products exact in f32, result rounded once
    ab, bb = O.f32_to_bf16(a), O.f32_to_bf16(b)
    want = O.bf16_to_f32(ab).astype(np.float64) @ O.bf16_to_f32(bb).astype(np.float64)
    assert_close(O.bf16_to_f32(O.gemm(ab, bb, code=O.BF16)), want, rtol=2 ** -8, atol=1e-2, what="bf16 gemm")


def test_attention_forward():
    g = golden("attention")
    for i in range(3):
        B, H, Sq, Skv, D = (int(v) for v in g[f"fwd{i}_dims"])
        q, k, v = regen(1050 + i, [(B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D)], g[f"fwd{i}_sha"])
        o, lse = O.attn_fwd(q, k, v)
        assert_close(o, g[f"fwd{i}_out"], what=f"attention fwd case {i} (test_nn.py:11-33)")
        assert np.isfinite(lse).all()


def test_attention_backward():
    g = golden("attention")
    for i in range(3):
        B, H, Sq, Skv, D = (int(v) for v in g[f"bwd{i}_dims"])
        q, k, v, go = regen(1060 + i, [(B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D), (B, H, Sq, D)], g[f"bwd{i}_sha"], lo=-1, hi=1)
        o, _ = O.attn_fwd(q, k, v)
        assert_close(o, g[f"bwd{i}_out"], rtol=1e-4, atol=1e-5, what="fwd")
        dq, dk, dv = O.attn_bwd(q, k, v, go)
        for n, got in (("dq", dq), ("dk", dk), ("dv", dv)):
            assert_close(got, g[f"bwd{i}_{n}"], rtol=1e-4, atol=1e-5, what=f"bwd case {i} {n}")


# ---- sort / topk: the oracle against the reference tests' torch.sort(stable=True) / np.argsort / torch.topk -------
def _sort_cases(g, prefix, count_key):
    for n in range(int(g[count_key][0])):
        meta = g[f"{prefix}{n}_meta"]
        seed, dim, flag, shape = int(meta[0]), int(meta[1]), bool(meta[2]), [int(v) for v in meta[3:]]
        yield n, seed, dim, flag, shape, np.dtype(str(g[f"{prefix}{n}_dtype"]))


def test_sort_oracle_matches_reference_test_expressions():
    from tests.helpers import sha
    g = golden("sort")
    for n, seed, dim, desc, shape, dt in _sort_cases(g, "s", "n_sort"):
        arr = np.random.default_rng(seed).uniform(-1000, 1000, size=shape).astype(dt)
        assert np.array_equal(sha(arr), g[f"s{n}_sha_in"])
        res, ind = O.sort_stable(arr, dim, desc)
        assert ind.dtype == np.int64
        if f"s{n}_res" in g:
            assert np.array_equal(res, g[f"s{n}_res"]) and np.array_equal(ind, g[f"s{n}_ind"]), (n, shape, dim, desc, dt)
        else:
            assert np.array_equal(sha(res, ind), g[f"s{n}_sha_out"]), (n, shape, dim, desc, dt)


def test_topk_oracle_matches_reference_test_expressions():
    from tests.helpers import sha
    g = golden("sort")
    for n, seed, dim, largest, shape, dt in _sort_cases(g, "t", "n_topk"):
        if np.prod(shape) > 2_000_000 and n % 3:  # keep the CPU suite short: every third of the 16 M-element cases
            continue
        arr = np.random.default_rng(seed).uniform(-100000, 100000, size=shape).astype(dt)
        assert np.array_equal(sha(arr), g[f"t{n}_sha_in"])
        res, _ = O.topk(arr, 8, dim, largest)
        assert np.array_equal(sha(res), g[f"t{n}_sha_out"]), (n, shape, dim, largest, dt)


def test_sort_key_order_special_values():
    # KeyTraits order (sorting_common.h:40-55): -NaN < -inf < ... < -0.0 < +0.0 < ... < +inf < +NaN
    x = np.array([np.nan, np.inf, 1.0, 0.0, -0.0, -1.0, -np.inf, -np.nan], dtype=np.float32)
    x[7] = np.frombuffer(np.uint32(0xFFC00000).tobytes(), np.float32)[0]
    v, i = O.sort_stable(x[None, :], 1, False)
    assert i[0].tolist() == [7, 6, 5, 4, 3, 2, 1, 0]
    v, i = O.sort_stable(x[None, :], 1, True)
    assert i[0].tolist() == [0, 1, 2, 3, 4, 5, 6, 7]
    with pytest.raises(ValueError):
        O.sort_key(np.zeros(3, np.bool_))


def test_norms_vs_torch_fixtures():
    """rms_norm / layer_norm forward + backward of the oracle against torch-CPU F.rms_norm / F.layer_norm + autograd (float64)
    on the committed fixtures (tests/golden/gen_golden.py: norms)."""
    g = golden("norms")
    for i in range(5):
        x, go, w, b = g[f"n{i}_x"], g[f"n{i}_g"], g[f"n{i}_w"], g[f"n{i}_b"]
        for kind, name in ((O.RMS, "rms"), (O.LAYER, "layer")):
            y, mean, rstd = O.norm_fwd(kind, x, w, b if kind == O.LAYER else None, eps=1e-5)
            assert_close(y, g[f"n{i}_{name}_y"], rtol=1e-5, atol=1e-5, what=f"{name} fwd {i}")
            dx, dw, db = O.norm_bwd(kind, x, w, go, eps=1e-5)
            assert_close(dx, g[f"n{i}_{name}_dx"], rtol=1e-4, atol=1e-5, what=f"{name} dx {i}")
            assert_close(dw, g[f"n{i}_{name}_dw"], rtol=1e-4, atol=1e-4, what=f"{name} dw {i}")
            if kind == O.LAYER:
                assert_close(db, g[f"n{i}_layer_db"], rtol=1e-4, atol=1e-4, what=f"db {i}")
                x2 = x.reshape(-1, x.shape[-1]).astype(np.float64)
                assert_close(mean, x2.mean(1), rtol=1e-5, atol=1e-6, what="mean")
                assert_close(rstd, 1.0 / np.sqrt(x2.var(1) + 1e-5), rtol=1e-5, atol=1e-6, what="rstd")


def test_index_get_is_numpy_take():
    rng = np.random.default_rng(140)
    for dt in (np.float32, np.uint16, np.int64, np.uint8):
        table = rng.integers(0, 200, size=(50, 24)).astype(dt)
        idx = rng.integers(-50, 50, size=(7, 9))
        assert np.array_equal(O.index_get(table, idx), table[idx])
