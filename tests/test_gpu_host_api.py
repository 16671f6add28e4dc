"""-m gpu: the Python operator API (`import kfunca_amd as kfunca`) — the cases of the reference's own
test suite (test/test_tensor.py, test_gemm.py, test_nn.py), re-derived with fixed seeds and checked
against the golden vectors / the CPU oracle, plus what the reference lacks (GEMM / attention backward)."""
import copy

import numpy as np
import pytest

import kfunca_amd as kfunca
from oracle import checks as K
from oracle import oracle as O
from tests.helpers import assert_close, golden, regen, uni

pytestmark = pytest.mark.gpu


def close(a, b, atol=1e-3, rtol=1e-3):  # test/common.py:6-11
    a = a if isinstance(a, np.ndarray) else a.contiguous().numpy()
    b = b if isinstance(b, np.ndarray) else b.contiguous().numpy()
    assert_close(a, b, rtol=rtol, atol=atol)


def test_tensor_impl_roundtrip():  # test_tensor.py:10-13
    rng = np.random.default_rng(1)
    for dt in (np.float64, np.float32, np.float16, np.int64, np.int32, np.int16, np.int8, np.uint8, np.bool_):
        arr = (rng.uniform(-10, 10, size=(2, 3)) * 3).astype(dt)
        t = kfunca.from_numpy(arr, 0)
        assert t.sizes() == [2, 3] and np.array_equal(t.numpy(), arr)
    nc = np.arange(24, dtype=np.float32).reshape(4, 6).T  # non-contiguous numpy input is honoured
    assert np.array_equal(kfunca.from_numpy(nc, 0).numpy(), nc)


def test_tensor_add_golden():  # test_tensor.py:15-27
    g = golden("elementwise")
    for i in range(3):
        a = kfunca.from_numpy(g[f"add{i}_a"], 0)
        assert np.array_equal((a + a).numpy(), g[f"add{i}_out"])
        out = kfunca.from_numpy(g[f"promo{i}_a"], 0) + kfunca.from_numpy(g[f"promo{i}_b"], 0)
        assert out.dtype() == kfunca.float and np.array_equal(out.numpy(), g[f"promo{i}_out"])
    s = golden("shape_ops")
    t = kfunca.from_numpy(s["int_x"], 0)
    assert np.array_equal((t + t).numpy(), s["int_out"])  # test/core/test_tensor.cpp:10-23


def test_inplace_op_golden():  # test_tensor.py:29-68
    g = golden("elementwise")
    a, b = kfunca.from_numpy(g["inpl_a"], 0), kfunca.from_numpy(g["inpl_b"], 0)
    addr = a.data_ptr()
    steps = g["inpl_steps"]
    a += b; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[0])
    a -= b; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[1])
    a *= b; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[2])
    a /= b; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[3])
    a += 2; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[4])
    a -= 3; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[5])
    a *= 4; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[6])
    a /= 5; assert a.data_ptr() == addr and np.array_equal(a.numpy(), steps[7])


def test_scalar_operands_no_temporary_and_small_dtypes():  # register.cpp:172-206 semantics on every path
    rng = np.random.default_rng(9)
    a = uni(rng, (33, 65))
    t = kfunca.from_numpy(a, 0)
    before = kfunca.memstat_dict(0)["driver_allocs"]
    r = ((t + 2.0) * 0.5 - 1.0) / 4.0
    assert np.array_equal(r.numpy(), ((a + np.float32(2.0)) * np.float32(0.5) - np.float32(1.0)) / np.float32(4.0))
    i8 = kfunca.from_numpy(np.arange(-20, 20, dtype=np.int8).reshape(4, 10), 0)  # small ints: 1-element broadcast operand
    assert np.array_equal((i8 * 3.0).numpy(), (np.arange(-20, 20).reshape(4, 10) * 3).astype(np.int8))
    h = t.bfloat16()
    want = O.binary(O.ADD, O.f32_to_bf16(a), O.fill(np.empty(a.shape, dtype=np.uint16), 0.3, dst_code=7), a_code=7, b_code=7)
    assert np.array_equal((h + 0.3).float().numpy(), O.bf16_to_f32(want))
    x = kfunca.from_numpy(a, 0)
    x.set_requires_grad(True)
    y = (x + 1.0) + x  # d/dx = 2
    y.backward(kfunca.from_numpy(np.ones_like(a), 0))
    assert np.array_equal(x.grad().numpy(), np.full_like(a, 2.0))
    assert before <= kfunca.memstat_dict(0)["driver_allocs"]


def test_data_ptr_and_refcounts():  # test_tensor.py:70-84
    arr = np.random.default_rng(2).uniform(-10, 10, size=(3, 4)).astype(np.float32)
    x = kfunca.from_numpy(arr, 0)
    x_ref = kfunca.from_numpy(arr, 0)
    x_ref = x
    x_deep = copy.deepcopy(x)
    assert x.data_ptr() == x_ref.data_ptr() == x_deep.data_ptr()
    assert x.storage_ref_count() == x_ref.storage_ref_count() == x_deep.storage_ref_count() == 1
    assert x.impl_ref_count() == x_ref.impl_ref_count() == x_deep.impl_ref_count() == 2
    del x
    assert x_deep.impl_ref_count() == 2 and x_ref.impl_ref_count() == 2
    del x_ref
    assert x_deep.impl_ref_count() == 1
    v = x_deep.permute(1, 0)  # a view shares the storage, not the impl
    assert v.storage_ref_count() == 2 and v.impl_ref_count() == 1 and v.data_ptr() == x_deep.data_ptr()


def test_broadcast_binary_golden():  # test_tensor.py:86-108 ('easy' shapes)
    g = golden("elementwise")
    for i in range(3):
        a, b = kfunca.from_numpy(g[f"bc{i}_a"], 0), kfunca.from_numpy(g[f"bc{i}_b"], 0)
        assert np.array_equal((a + b).numpy(), g[f"bc{i}_add"])
        assert np.array_equal((a - b).numpy(), g[f"bc{i}_sub"])
        assert np.array_equal((a * b).numpy(), g[f"bc{i}_mul"])
        assert np.array_equal((a / b).numpy(), g[f"bc{i}_div"])
        assert np.array_equal((kfunca.from_numpy(g[f"bc{i}_ai"], 0) * b).numpy(), g[f"bc{i}_imul"])


@pytest.mark.slow
def test_broadcast_binary_hard():  # test_tensor.py:91-92: 2^30 elements, forces the 32-bit split
    rng = np.random.default_rng(3)
    small = uni(rng, (2, 1024, 1, 512))
    big = kfunca.empty([2, 1024, 1024, 512], kfunca.float, 0)
    big.fill_(1.5)
    out = big + kfunca.from_numpy(small, 0)
    assert out.sizes() == [2, 1024, 1024, 512]
    for (i, j) in ((0, 0), (0, 1023), (1, 512), (1, 1023)):  # windows on both sides of the 2^31-byte boundary
        got = out[i, j].contiguous().numpy()
        assert np.array_equal(got, np.broadcast_to(small[i, j] + np.float32(1.5), (1024, 512)))
    same = big + big
    assert np.array_equal(same[1, 1000, 17:19].contiguous().numpy(), np.full((2, 512), 3.0, dtype=np.float32))


def test_reduce():  # test_tensor.py:110-118 at the reference's shape
    g = golden("reductions")
    x = kfunca.from_numpy(g["x"], 0)
    for dim in range(3):
        close(x.sum(dim), g[f"sum{dim}"], atol=1e-2, rtol=1e-2)
        close(x.mean(dim), g[f"mean{dim}"], atol=1e-2, rtol=1e-2)
        assert x.sum(dim).sizes() == list(g[f"sum{dim}"].shape)  # keepdim
    rng = np.random.default_rng(4)
    arr = uni(rng, (223, 23, 3213))
    t = kfunca.from_numpy(arr, 0)
    for dim in range(3):
        close(t.sum(dim), np.sum(arr, axis=dim, keepdims=True), atol=1e-2, rtol=1e-2)
        close(t.mean(dim), np.mean(arr, axis=dim, keepdims=True), atol=1e-2, rtol=1e-2)
    a, _ = regen(g["c1_seed"][0], [(1024, 1024), (1024, 1024)], g["c1_sha"])  # BASELINE C1
    ta = kfunca.from_numpy(a, 0)
    close(ta.sum(0), g["c1_sum0"], atol=1e-2, rtol=1e-5)
    close(ta.sum(1), g["c1_sum1"], atol=1e-2, rtol=1e-5)


def test_mean_std():  # test_tensor.py:120-132, the same expressions on the GPU tensors, plus the golden values
    g = golden("moments")
    shape, dim = (13, 325, 127), 1
    (arr,) = regen(g["ms_seed"][0], [shape], g["ms_sha"], dtype=np.float64)
    arr_ = kfunca.from_numpy(arr, 0)
    divisor = shape[dim] - 1
    mean = arr_.mean(dim)
    var = ((arr_ - mean) * (arr_ - mean)).sum(dim)
    var = var / divisor
    mean_var = arr_.mean_var(dim, False)
    close(mean, mean_var[0], atol=1e-2, rtol=1e-2)
    close(var, mean_var[1], atol=1e-2, rtol=1e-2)
    close(mean_var[0], g["ms_mean"], atol=1e-12, rtol=1e-11)
    close(mean_var[1], g["ms_var"], atol=0, rtol=1e-11)
    assert mean_var[0].sizes() == [13, 1, 127] and mean_var[1].dtype() == kfunca.double
    close(arr_.mean_var(dim, True)[1], np.sqrt(g["ms_var"]), atol=0, rtol=1e-11)
    kfunca.memstat()


def test_norm_stat():  # test_tensor.py:134-146
    g = golden("moments")
    for i in range(4):
        shape = [int(v) for v in g[f"ns{i}_shape"]]
        (arr,) = regen(g[f"ns{i}_seed"][0], [tuple(shape)], g[f"ns{i}_sha"])
        arr_ = kfunca.from_numpy(arr, 0)
        mean_invstd = arr_.norm_stat(0)
        close(g[f"ns{i}_mean"], mean_invstd[0])
        close(g[f"ns{i}_invstd"], mean_invstd[1])
        assert mean_invstd[0].sizes() == [1, shape[1]] and mean_invstd[0].dtype() == kfunca.float
    h = kfunca.from_numpy(uni(np.random.default_rng(5), (512, 384)), 0).bfloat16()  # 16-bit input: f32 statistics
    m, inv = h.norm_stat(1)
    ref = h.float().numpy().astype(np.float64)
    assert m.dtype() == kfunca.float and m.sizes() == [512, 1]
    close(m, ref.mean(axis=1, keepdims=True), atol=1e-5, rtol=1e-5)
    close(inv, 1.0 / np.sqrt(ref.var(axis=1, keepdims=True)), atol=0, rtol=1e-4)


def test_convert():  # test_tensor.py:148-160
    g = golden("elementwise")
    t = kfunca.from_numpy(g["cvt_x"], 0)
    h = t.half()
    assert np.array_equal(h.numpy().view(np.uint16), g["cvt_half_bits"])
    h *= h
    assert np.array_equal(h.float().numpy(), g["cvt_half_sq"])
    bf = t.bfloat16()
    bf *= bf
    assert np.array_equal(bf.float().numpy(), g["cvt_bf16_sq"])
    t *= t
    close(t, bf.float(), atol=1e-1, rtol=1e-1)


def test_permute_slice_view_cat_split_golden():  # test_tensor.py:162-167, 233-271 — bit-exact
    s = golden("shape_ops")
    t = kfunca.from_numpy(s["perm_x"], 0)
    p = t.permute(2, 1, 0, 3)
    assert not p.is_contiguous() and p.data_ptr() == t.data_ptr()
    assert np.array_equal(p.contiguous().numpy(), s["perm_out"])
    t = kfunca.from_numpy(s["slice_x"], 0)
    assert np.array_equal(t[3, 3:8, 4:11:2].contiguous().numpy(), s["slice_out"])
    t = kfunca.from_numpy(s["view_x"], 0)
    assert np.array_equal((t.view(5, -1, 23).contiguous() + 1).numpy(), s["view_out"])
    parts = [kfunca.from_numpy(s[f"cat_{k}"], 0) for k in "abc"]
    assert np.array_equal(kfunca.cat(parts, 1).numpy(), s["cat_out"])
    t = kfunca.from_numpy(s["split_x"], 0)
    for i, part in enumerate(t.split([11, 13, 1], 1)):
        assert np.array_equal(part.contiguous().numpy(), s[f"split_{i}"])
    with pytest.raises(RuntimeError):
        t.split([11, 13], 1)  # sizes must cover the dim (tensor_shape.cpp:87)
    with pytest.raises(RuntimeError):
        t.permute(0, 0, 1)


def test_index_put_golden():  # test_tensor.py:273-284
    s = golden("shape_ops")
    t = kfunca.from_numpy(s["iput_x"], 0)
    idx = [kfunca.from_numpy(s["iput_i0"].astype("q"), 0), kfunca.from_numpy(s["iput_i1"].astype("q"), 0)]
    t.index_put_(idx, kfunca.from_numpy(s["iput_v"], 0))
    assert np.array_equal(t.numpy(), s["iput_out"])
    t = kfunca.from_numpy(s["iput3_x"], 0)
    idx = [kfunca.from_numpy(s[f"iput3_i{k}"], 0) for k in range(3)]
    t.index_put_(idx, kfunca.from_numpy(s["iput3_v"], 0))
    assert np.array_equal(t.numpy(), s["iput3_out"])
    with pytest.raises(RuntimeError, match="Long"):
        t.index_put_([kfunca.from_numpy(np.zeros((2, 2), dtype=np.int32), 0)] * 3, kfunca.from_numpy(s["iput3_v"], 0))


def test_basic_backward_add_dag():  # test_tensor.py:286-309
    s = golden("shape_ops")
    rng = np.random.default_rng(5)
    grad = kfunca.from_numpy(s["ag_grad"], 0)
    a, b, c = (kfunca.from_numpy(uni(rng, (2, 3)), 0) for _ in range(3))
    a.set_requires_grad(True)
    b.set_requires_grad(True)
    ca = c + a
    ab = a + b
    accb = ca + ab
    accba = accb + a
    accba.backward(grad)
    close(a.grad(), s["ag_a_grad"])
    close(b.grad(), s["ag_b_grad"])
    assert not c.grad().defined()


def test_one_gradient_tensor_for_two_leaves_is_shared_by_neither():
    """add's backward hands the SAME tensor to both inputs (binary_ops.cpp:16-33). The engine adopts a first gradient without a copy only
    when the handle it holds is the tensor's only one (update_grad: impl_ref_count() == 1 - it moves its accumulator slot in), so two
    leaves fed by one tensor end up with gradients of their own: a second backward adds into each in place and must not be seen by
    the other; the caller's grad_output is never adopted either; a fresh weight gradient (gemm's dW) still is, without a copy."""
    rng = np.random.default_rng(11)
    a, b = (kfunca.from_numpy(uni(rng, (4, 8)), 0) for _ in range(2))
    a.set_requires_grad(True)
    b.set_requires_grad(True)
    g1 = kfunca.from_numpy(uni(rng, (4, 8)), 0)
    (a + b).backward(g1)
    assert len({a.grad().data_ptr(), b.grad().data_ptr(), g1.data_ptr()}) == 3
    # only a is in the second graph: its gradient doubles in place, b's stays, the caller's tensor is untouched
    g1_host = g1.numpy().copy()
    (a + kfunca.from_numpy(uni(rng, (4, 8)), 0)).backward(g1)
    assert np.array_equal(a.grad().numpy(), g1_host + g1_host)
    assert np.array_equal(b.grad().numpy(), g1_host)
    assert np.array_equal(g1.numpy(), g1_host)
    # a leaf that is its own root: the caller's grad_output is copied, not adopted
    c = kfunca.from_numpy(uni(rng, (4, 8)), 0)
    c.set_requires_grad(True)
    c.backward(g1)
    assert c.grad().data_ptr() != g1.data_ptr() and np.array_equal(c.grad().numpy(), g1_host)
    # the same leaf twice in one node: one gradient, the sum of both
    d = kfunca.from_numpy(uni(rng, (4, 8)), 0)
    d.set_requires_grad(True)
    (d + d).backward(g1)
    assert np.array_equal(d.grad().numpy(), g1_host + g1_host) and np.array_equal(g1.numpy(), g1_host)


def test_gemm_golden_and_backward():  # test_gemm.py:9-17 + backward (no reference counterpart)
    g = golden("gemm")
    a, b = regen(g["f64_seed"][0], [(123, 457), (457, 234)], g["f64_sha"], dtype=np.float64)
    out = kfunca.gemm(kfunca.from_numpy(a, 0), kfunca.from_numpy(b, 0), 1.0, 0.0)
    close(out, g["f64_out"])
    ta, tb = kfunca.from_numpy(g["f32_a"], 0), kfunca.from_numpy(g["f32_b"], 0)
    ta.set_requires_grad(True)
    tb.set_requires_grad(True)
    c = kfunca.gemm(ta, tb, 1.0, 0.0)
    close(c, g["f32_out"], atol=1e-4, rtol=1e-4)
    c.backward(kfunca.from_numpy(g["f32_g"], 0))
    close(ta.grad(), g["f32_da"], atol=1e-4, rtol=1e-4)
    close(tb.grad(), g["f32_db"], atol=1e-4, rtol=1e-4)
    # leading dims of A flatten into M (gemm_kernel.cu:10-15)
    a3 = kfunca.from_numpy(g["f32_a"].reshape(4, 32, 64), 0)
    c3 = kfunca.gemm(a3, tb, 1.0, 0.0)
    assert c3.sizes() == [4, 32, 256]
    close(c3, g["f32_out"].reshape(4, 32, 256), atol=1e-4, rtol=1e-4)
    with pytest.raises(RuntimeError):
        kfunca.gemm(ta, kfunca.from_numpy(g["f32_a"], 0), 1.0, 0.0)  # K mismatch
    with pytest.raises(RuntimeError):
        kfunca.gemm(ta.permute(1, 0), tb, 1.0, 0.0)  # operands must be contiguous


def test_causal_attention_golden_and_backward():  # test_nn.py:11-33 + backward
    g = golden("attention")
    for i in range(3):
        B, H, Sq, Skv, D = (int(x) for x in g[f"fwd{i}_dims"])
        q, k, v = regen(1050 + i, [(B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D)], g[f"fwd{i}_sha"])
        out = kfunca.causal_attention(*(kfunca.from_numpy(x, 0) for x in (q, k, v))).numpy()
        close(out, g[f"fwd{i}_out"])
    for i in range(3):
        B, H, Sq, Skv, D = (int(x) for x in g[f"bwd{i}_dims"])
        q, k, v, go = regen(1060 + i, [(B, H, Sq, D), (B, H, Skv, D), (B, H, Skv, D), (B, H, Sq, D)], g[f"bwd{i}_sha"], lo=-1, hi=1)
        tq, tk, tv = (kfunca.from_numpy(x, 0) for x in (q, k, v))
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(kfunca.from_numpy(go, 0))
        close(tq.grad(), g[f"bwd{i}_dq"], atol=1e-5, rtol=1e-4)
        close(tk.grad(), g[f"bwd{i}_dk"], atol=1e-5, rtol=1e-4)
        close(tv.grad(), g[f"bwd{i}_dv"], atol=1e-5, rtol=1e-4)
    # bf16 MFMA path through the operator API (tolerance: tests/test_gpu_attention.py header)
    rng = np.random.default_rng(6)
    q, k, v, go = (rng.uniform(-1, 1, (1, 2, 256, 128)).astype(np.float32) for _ in range(4))
    tq, tk, tv = (kfunca.from_numpy(x, 0).bfloat16() for x in (q, k, v))
    for t in (tq, tk, tv):
        t.set_requires_grad(True)
    out = kfunca.causal_attention(tq, tk, tv)
    out.backward(kfunca.from_numpy(go, 0).bfloat16())
    qb, kb, vb, gb = (O.f32_to_bf16(x) for x in (q, k, v, go))
    K.attn_check(qb, kb, vb, O.BF16, o=out.numpy(), d_o=gb, dq=tq.grad().numpy(), dk=tk.grad().numpy(), dv=tv.grad().numpy(), what="operator bf16")


def test_ragged_16bit_attention_takes_the_mfma_kernels():
    """Sequence lengths that are not multiples of 128 and head sizes below 128 (16-bit, Skv >= Sq): the operator pads with
    zeros, which leaves every result unchanged (zero columns add nothing to Q K^T or P V and the softmax scale stays
    1 / sqrt(D); padded keys are above every real query's diagonal; padded queries carry q = 0, dO = 0), and the MFMA kernels
    run instead of the generic vector-ALU one."""
    from kfunca_amd import hip_abi as H
    rng = np.random.default_rng(16)
    for (B, Hh, Sq, Skv, D) in ((2, 2, 200, 200, 128), (1, 2, 130, 300, 128), (1, 1, 1, 1, 128), (2, 3, 256, 256, 64), (1, 2, 100, 140, 80), (1, 2, 90, 90, 40)):
        q, k, v, go = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
        tq, tk, tv = (kfunca.from_numpy(x, 0).bfloat16() for x in (q, k, v))
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        H.profile_reset()
        H.profile_enable(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(kfunca.from_numpy(go, 0).bfloat16())
        kfunca.synchronize()
        H.profile_enable(False)
        names = set(H.profile_results())
        sfx = "_d64" if D <= 64 else ""  # head sizes up to 64 are padded to the native 64-wide kernels, the rest to 128
        assert {"attn_fwd_mfma" + sfx, "attn_bwd_dkv_mfma" + sfx, "attn_bwd_dq_mfma" + sfx} <= names and not any("generic" in n for n in names), names
        assert out.sizes() == [B, Hh, Sq, D]
        qb, kb, vb, gb = (O.f32_to_bf16(x) for x in (q, k, v, go))
        for t, like in zip((tq, tk, tv), (qb, kb, vb)):
            assert t.grad().sizes() == list(like.shape)
        K.attn_check(qb, kb, vb, O.BF16, o=out.numpy(), d_o=gb, dq=tq.grad().numpy(), dk=tk.grad().numpy(), dv=tv.grad().numpy(),
                     what=f"ragged {Sq}x{Skv} D{D}")


def test_ragged_f32_attention_takes_the_f32_mfma_kernels():
    """The same padding for f32 tensors (rows to multiples of 32, head size to 64 or 128): ragged shapes of the reference's own
    dtype - its third test shape, 65 x 33 keys... with Skv >= Sq - run on the exact-f32 MFMA kernels, forward and backward."""
    from kfunca_amd import hip_abi as H
    rng = np.random.default_rng(17)
    for (B, Hh, Sq, Skv, D) in ((2, 2, 65, 65, 123), (1, 2, 33, 100, 64), (1, 3, 200, 200, 128), (1, 1, 1, 7, 16)):
        q, k, v, go = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
        tq, tk, tv = (kfunca.from_numpy(x, 0) for x in (q, k, v))
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        H.profile_reset()
        H.profile_enable(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(kfunca.from_numpy(go, 0))
        kfunca.synchronize()
        H.profile_enable(False)
        names = set(H.profile_results())
        assert {"attn_fwd_f32_mfma", "attn_bwd_dkv_f32_mfma", "attn_bwd_dq_f32_mfma"} <= names and not any("generic" in n for n in names), names
        o_ref, _ = O.attn_fwd(q, k, v)
        close(out, o_ref, atol=2e-5, rtol=2e-5)
        for t, ref in zip((tq, tk, tv), O.attn_bwd(q, k, v, go)):
            assert t.grad().sizes() == list(ref.shape)
            close(t.grad(), ref, atol=5e-5, rtol=1e-4)


def test_allocator_reuse_and_scope():
    base = kfunca.memstat_dict(0)
    t = kfunca.empty([1 << 20], kfunca.float, 0)
    p = t.data_ptr()
    del t
    t2 = kfunca.empty([1 << 20], kfunca.float, 0)  # best-fit reuse of the cached 4 MiB block
    assert t2.data_ptr() == p
    st = kfunca.memstat_dict(0)
    assert st["driver_allocs"] <= base["driver_allocs"] + 1
    z = kfunca.zeros([7, 5], kfunca.int, 0)
    assert not z.numpy().any()
    with pytest.raises(RuntimeError, match="does not support bool"):  # sort_ops_kernel.cu:569-570
        kfunca.zeros([4], kfunca.bool, 0).sort(0, False)
    with pytest.raises(RuntimeError, match="out of range"):
        z.topk(9, 0, True)
    for call in (lambda: z.norm_stat(0), lambda: z.mean_var(0, False)):  # floating dtypes only, as the reference's dispatch
        with pytest.raises(RuntimeError, match="Unsupported ScalarType"):
            call()
    assert "tensor(shape=[7,5]" in repr(z)
    assert z.item([1, 2]) == 0


def test_sort_small_slice():  # test_tensor.py:169-192 (shapes, dims, dtypes, both directions; seeds fixed)
    g = golden("sort")
    from tests.helpers import sha
    for n in range(int(g["n_sort"][0])):
        meta = g[f"s{n}_meta"]
        seed, dim, desc, shape = int(meta[0]), int(meta[1]), bool(meta[2]), [int(v) for v in meta[3:]]
        arr = np.random.default_rng(seed).uniform(-1000, 1000, size=shape).astype(np.dtype(str(g[f"s{n}_dtype"])))
        res, ind = kfunca.from_numpy(arr, 0).sort(dim, desc)
        assert res.sizes() == shape and ind.sizes() == shape and ind.dtype() == kfunca.long
        res, ind = res.numpy(), ind.numpy()
        if f"s{n}_res" in g:
            assert np.array_equal(res, g[f"s{n}_res"]) and np.array_equal(ind, g[f"s{n}_ind"]), (n, shape, dim, desc)
        else:
            assert np.array_equal(sha(res, ind), g[f"s{n}_sha_out"]), (n, shape, dim, desc)


def test_sort_views_and_negative_dim():
    rng = np.random.default_rng(801)
    arr = rng.integers(-50, 50, size=(6, 40, 9)).astype(np.int32)
    t = kfunca.from_numpy(arr, 0).permute(2, 0, 1)  # non-contiguous input: the dense dim-last copy path
    ref = arr.transpose(2, 0, 1)
    for dim in (-1, 0, 1):
        for desc in (False, True):
            v, i = t.sort(dim, desc)
            wv, wi = O.sort_stable(np.ascontiguousarray(ref), dim % 3, desc)
            assert np.array_equal(v.numpy(), wv) and np.array_equal(i.numpy(), wi)
    e = kfunca.empty([3, 0, 5], kfunca.float, 0)
    v, i = e.sort(1, False)
    assert v.sizes() == [3, 0, 5] and i.sizes() == [3, 0, 5]


def test_topk_small_and_large():  # test_tensor.py:203-231: values of torch.topk (the reference does not compare indices)
    g = golden("sort")
    from tests.helpers import sha
    for n in range(int(g["n_topk"][0])):
        meta = g[f"t{n}_meta"]
        seed, dim, largest, shape = int(meta[0]), int(meta[1]), bool(meta[2]), [int(v) for v in meta[3:]]
        arr = np.random.default_rng(seed).uniform(-100000, 100000, size=shape).astype(np.dtype(str(g[f"t{n}_dtype"])))
        res, ind = kfunca.from_numpy(arr, 0).topk(8, dim, largest)
        want = list(shape)
        want[dim] = 8
        assert res.sizes() == want and ind.sizes() == want
        assert np.array_equal(sha(res.numpy()), g[f"t{n}_sha_out"]), (n, shape, dim, largest)
        assert np.array_equal(np.take_along_axis(arr, ind.numpy(), axis=dim), res.numpy())
    for i in range(2):
        seed, k = (int(v) for v in g[f"tl{i}_meta"])
        arr = np.random.default_rng(seed).uniform(-10000, 10000, size=(4, 1024000)).astype(np.float32)
        res, ind = kfunca.from_numpy(arr, 0).topk(k, 1, True)
        assert np.array_equal(sha(res.numpy()), g[f"tl{i}_sha_out"])


def test_ragged_gemm_runs_on_the_matrix_cores():
    """M, N, K off the kernels' tile multiples: the operator zero-pads (big enough problems) and the MFMA kernels run instead
    of the scalar fallback - forward and both backward products; results as for aligned shapes."""
    from kfunca_amd import hip_abi as H
    rng = np.random.default_rng(21)
    for dt_name, (M, K, N), tol in (("bf16", (1000, 700, 1030), 2e-2), ("f32", (333, 250, 517), 1e-4)):
        a = rng.uniform(-1, 1, (M, K)).astype(np.float32)
        b = (rng.uniform(-1, 1, (K, N)) / np.sqrt(K)).astype(np.float32)
        g = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        if dt_name == "bf16":
            a, b, g = (O.bf16_to_f32(O.f32_to_bf16(x)) for x in (a, b, g))
        up = (lambda x: kfunca.from_numpy(x, 0).bfloat16()) if dt_name == "bf16" else (lambda x: kfunca.from_numpy(x, 0))
        ta, tb = up(a), up(b)
        ta.set_requires_grad(True)
        tb.set_requires_grad(True)
        H.profile_reset()
        H.profile_enable(True)
        c = kfunca.gemm(ta, tb, 1.0, 0.0)
        c.backward(up(g))
        kfunca.synchronize()
        H.profile_enable(False)
        names = set(H.profile_results())
        assert any("mfma" in n for n in names) and "gemm_generic" not in names, names
        out = (lambda t: t.float().numpy()) if dt_name == "bf16" else (lambda t: t.numpy())
        want = a.astype(np.float64) @ b.astype(np.float64)
        assert c.sizes() == [M, N] and np.abs(out(c) - want).max() <= tol * np.abs(want).max()
        wa, wb = g.astype(np.float64) @ b.astype(np.float64).T, a.astype(np.float64).T @ g.astype(np.float64)
        assert np.abs(out(ta.grad()) - wa).max() <= tol * np.abs(wa).max()
        assert np.abs(out(tb.grad()) - wb).max() <= tol * np.abs(wb).max()


def test_out_of_memory_is_recoverable_and_the_attention_backward_falls_back():
    """ADVICE round 3 (medium). (1) kf_malloc reports out-of-memory as KF_ERR_OOM -> utils::OutOfMemory, having read HIP's sticky
    last-error (ROCm >= 7.0 keeps it until read): the next kernel launch check is clean. (2) The allocator hands its idle cache back to
    the driver and retries once before giving up. (3) causal_attention's backward scratch asks for less when - and only when - the failure
    is out-of-memory, down to the statistics alone (the recomputing dQ kernel): forced here with the allocator's test hook, results
    checked against the ordinary run (dK / dV the same bits, dQ under the oracle's bounds)."""
    rng = np.random.default_rng(61)
    # (1) a real out-of-memory through the allocator, then a kernel: the launch check must not see a stale error
    with pytest.raises(RuntimeError, match="out of device memory"):
        kfunca.empty([1 << 40], kfunca.byte, 0)  # 1 TiB on a 288 GB device
    a = rng.uniform(-1, 1, (300, 7)).astype(np.float32)
    assert np.array_equal((kfunca.from_numpy(a, 0) + kfunca.from_numpy(a, 0)).numpy(), a + a)
    # (2) that attempt took the release-and-retry path (the idle cache went back to the driver before the second try)
    assert kfunca._alloc_oom_retries() >= 1
    # (3) the attention backward under a simulated shortage
    q, k, v, go = (rng.uniform(-1, 1, (1, 2, 512, 128)).astype(np.float32) for _ in range(4))
    qb, kb, vb, gb = (O.f32_to_bf16(x) for x in (q, k, v, go))

    def run():
        tq, tk, tv = (kfunca.from_numpy(x, 0).bfloat16() for x in (q, k, v))
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(kfunca.from_numpy(go, 0).bfloat16())
        return out.numpy(), tq.grad().numpy(), tk.grad().numpy(), tv.grad().numpy()
    ref = run()
    kfunca.synchronize(0)
    kfunca.release_cached(0)  # so that the scratch request reaches the driver (and the hook) instead of a cached block
    before = kfunca._alloc_oom_retries()
    import os
    with pytest.raises(RuntimeError, match="test hook"):
        os.environ.pop("KF_TEST_HOOKS", None)
        kfunca._alloc_fail_above(600 << 10)          # the hook does not arm in a process that has not declared itself a test
    os.environ["KF_TEST_HOOKS"] = "1"
    kfunca._alloc_fail_above(600 << 10)  # dS of the two heads is 1 MiB, every tensor of this problem 256 KiB
    try:
        got = run()
    finally:
        kfunca._alloc_fail_above(0)
    assert kfunca._alloc_oom_retries() > before
    assert np.array_equal(got[0], ref[0]) and np.array_equal(got[2], ref[2]) and np.array_equal(got[3], ref[3])
    K.attn_check(qb, kb, vb, O.BF16, o=got[0], d_o=gb, dq=got[1], dk=got[2], dv=got[3], what="backward after the out-of-memory fallback")
    # anything that is not out-of-memory still propagates (the fallback catches utils::OutOfMemory only): a bad device index
    with pytest.raises(RuntimeError, match="out of range"):
        kfunca.empty([4], kfunca.float, 99)
