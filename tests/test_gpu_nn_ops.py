"""-m gpu: the reference's roadmap operators (README.md:28-30) through the Python operator API with autograd: rms_norm,
layer_norm, embedding; and the bfloat16 numpy bridge. Expected values: the torch-CPU fixtures (tests/golden/norms.npz) and
numpy (gather = table[idx], its backward = np.add.at in f64)."""
import numpy as np
import pytest

import kfunca_amd as kfunca
from oracle import checks as K
from oracle import oracle as O
from tests.helpers import assert_close, golden

pytestmark = pytest.mark.gpu


def leaf(a, bf16=False):
    t = kfunca.from_numpy(a, 0)
    if bf16:
        t = t.bfloat16()
    t.set_requires_grad(True)
    return t


def test_norms_forward_backward_through_autograd():
    g = golden("norms")
    for i in range(5):
        x, go, w, b = g[f"n{i}_x"], g[f"n{i}_g"], g[f"n{i}_w"], g[f"n{i}_b"]
        for name in ("rms", "layer"):
            tx, tw, tb = leaf(x), leaf(w), leaf(b)
            y = kfunca.rms_norm(tx, tw, 1e-5) if name == "rms" else kfunca.layer_norm(tx, tw, tb, 1e-5)
            assert y.sizes() == list(x.shape)
            assert_close(y.numpy(), g[f"n{i}_{name}_y"], rtol=1e-5, atol=1e-5, what=f"{name} y {i}")
            y.backward(kfunca.from_numpy(go, 0))
            assert_close(tx.grad().numpy(), g[f"n{i}_{name}_dx"], rtol=1e-4, atol=1e-5, what=f"{name} dx {i}")
            assert_close(tw.grad().numpy(), g[f"n{i}_{name}_dw"], rtol=1e-4, atol=1e-4, what=f"{name} dw {i}")
            if name == "layer":
                assert_close(tb.grad().numpy(), g[f"n{i}_layer_db"], rtol=1e-4, atol=1e-4, what=f"db {i}")
    # weight-less forms and a frozen weight (no dw requested)
    x = g["n1_x"]
    tx = leaf(x)
    y = kfunca.rms_norm(tx)
    ref, _, _ = O.norm_fwd(O.RMS, x, None, None)
    assert_close(y.numpy(), ref, rtol=1e-5, atol=1e-5)
    y.backward(kfunca.from_numpy(g["n1_g"], 0))
    assert_close(tx.grad().numpy(), O.norm_bwd(O.RMS, x, None, g["n1_g"])[0], rtol=1e-4, atol=1e-5)


def test_norm_bf16_chain_with_gemm():
    """bf16 rms_norm feeding a GEMM, backward through both (what a block does): vs f64 numpy on the bf16-rounded inputs."""
    rng = np.random.default_rng(180)
    x = O.bf16_to_f32(O.f32_to_bf16(rng.uniform(-2, 2, (256, 512)).astype(np.float32)))
    w = O.bf16_to_f32(O.f32_to_bf16(rng.uniform(0.5, 1.5, (512,)).astype(np.float32)))
    m = O.bf16_to_f32(O.f32_to_bf16(rng.uniform(-1, 1, (512, 128)).astype(np.float32) / 16))
    go = O.bf16_to_f32(O.f32_to_bf16(rng.uniform(-1, 1, (256, 128)).astype(np.float32)))
    tx, tw, tm = leaf(x, True), leaf(w, True), leaf(m, True)
    y = kfunca.gemm(kfunca.rms_norm(tx, tw, 1e-5), tm, 1.0, 0.0)
    y.backward(kfunca.from_numpy(go, 0).bfloat16())
    x64, w64, m64, g64 = (a.astype(np.float64) for a in (x, w, m, go))
    rstd = 1.0 / np.sqrt((x64 ** 2).mean(1, keepdims=True) + 1e-5)
    h = x64 * rstd * w64
    assert_close(y.float().numpy(), h @ m64, rtol=3e-2, atol=3e-2, what="y")
    dh = g64 @ m64.T
    gg = dh * w64
    xh = x64 * rstd
    dx = rstd * (gg - xh * (gg * xh).mean(1, keepdims=True))
    assert_close(tx.grad().float().numpy(), dx, rtol=5e-2, atol=5e-2, what="dx")
    assert_close(tw.grad().float().numpy(), (dh * xh).sum(0), rtol=5e-2, atol=0.5, what="dw")
    assert_close(tm.grad().float().numpy(), h.T @ g64, rtol=5e-2, atol=0.5, what="dm")


def test_embedding_forward_backward():
    rng = np.random.default_rng(181)
    vocab, dim = 500, 96
    table = rng.uniform(-1, 1, (vocab, dim)).astype(np.float32)
    idx = rng.integers(-vocab, vocab, size=(4, 37)).astype(np.int64)
    idx[0, :20] = 3
    tt = leaf(table)
    out = kfunca.embedding(tt, kfunca.from_numpy(idx, 0))
    assert out.sizes() == [4, 37, dim] and np.array_equal(out.numpy(), table[idx])
    go = rng.uniform(-1, 1, (4, 37, dim)).astype(np.float32)
    out.backward(kfunca.from_numpy(go, 0))
    want = np.zeros((vocab, dim), dtype=np.float64)
    np.add.at(want, np.where(idx < 0, idx + vocab, idx).reshape(-1), go.reshape(-1, dim).astype(np.float64))
    assert_close(tt.grad().numpy(), want, rtol=1e-5, atol=1e-5, what="dTable")
    # any dtype gathers bit-exactly (no gradient): int64 rows
    ti = rng.integers(-9, 9, size=(50, 5)).astype(np.int64)
    assert np.array_equal(kfunca.embedding(kfunca.from_numpy(ti, 0), kfunca.from_numpy(idx[:1] % 50, 0)).numpy(), ti[idx[:1] % 50])


def test_bfloat16_numpy_bridge():
    """to_numpy of a bfloat16 tensor returns the raw bits as uint16 (the reference rejects the dtype, register.cpp:41-57);
    from_numpy_bf16 is the inverse; the bits are the oracle's round-to-nearest-even of the f32 values (half.h:195-208)."""
    rng = np.random.default_rng(182)
    x = rng.uniform(-10, 10, (7, 33)).astype(np.float32)
    t = kfunca.from_numpy(x, 0).bfloat16()
    bits = t.numpy()
    assert bits.dtype == np.uint16 and np.array_equal(bits, O.f32_to_bf16(x))
    back = kfunca.from_numpy_bf16(bits, 0)
    assert back.dtype() == kfunca.bfloat16 and np.array_equal(back.numpy(), bits)
    assert np.array_equal(back.float().numpy(), O.bf16_to_f32(bits))


def test_qkv_linear_feeds_attention_in_place():
    """README.md:32's qkv_linear: x W_qkv + b in one kernel, its packed output consumed in place by causal_attention_qkv; forward and
    the gradients of x, W and b against f64 numpy / the oracle on the bf16-rounded inputs."""
    rng = np.random.default_rng(183)
    B, S, Hh, D = 1, 128, 2, 128
    d = Hh * D
    r = lambda shp, sc=1.0: O.bf16_to_f32(O.f32_to_bf16((rng.uniform(-1, 1, shp) * sc).astype(np.float32)))  # noqa: E731
    x, w, b, g = r((B * S, d)), r((d, 3 * d), 1 / 16), r((3 * d,), 0.1), r((B * S, d))
    tx, tw, tb = leaf(x, True), leaf(w, True), leaf(b, True)
    qkv = kfunca.qkv_linear(tx, tw, tb)
    out = kfunca.causal_attention_qkv(qkv, B, S, Hh)
    out.backward(kfunca.from_numpy(g, 0).bfloat16())
    eps = 2.0 ** -8
    x64, w64 = x.astype(np.float64), w.astype(np.float64)
    qkv_ref = x64 @ w64 + b
    got = qkv.float().numpy().astype(np.float64)
    assert (np.abs(got - qkv_ref) <= eps * np.abs(qkv_ref) + 1e-6 * (np.abs(x64) @ np.abs(w64) + np.abs(b))).all(), "qkv"  # the GEMM bound of test_gpu_gemm.py
    # attention is checked on the projection the device produced (its bits), under the scale-aware bounds of oracle/checks.py
    heads = lambda x2: np.ascontiguousarray(x2.reshape(B, S, Hh, D).transpose(0, 2, 1, 3))  # noqa: E731
    flat = lambda x4: x4.transpose(0, 2, 1, 3).reshape(B * S, d)  # noqa: E731
    qkv_bits, g_bits = qkv.numpy(), O.f32_to_bf16(g)
    q, k, v = heads(qkv_bits[:, :d]), heads(qkv_bits[:, d:2 * d]), heads(qkv_bits[:, 2 * d:])
    ref = O.attn_ref64(q, k, v, heads(g_bits), code=O.BF16)
    K.attn_check(q, k, v, O.BF16, o=heads(out.numpy()), ref=ref, what="attention on the packed projection")
    # the gradients behind it: d(qkv) = [dq | dk | dv] reaches the projection rounded to bf16 once; sums of it bounded by eps x sum |terms|
    dqkv = np.concatenate([flat(ref["dq"]), flat(ref["dk"]), flat(ref["dv"])], axis=1)
    mqkv = np.concatenate([flat(ref["mdq"]), flat(ref["mdk"]), flat(ref["mdv"])], axis=1)  # the error scale of every d(qkv) element
    edqkv = eps * (1.5 * np.abs(dqkv) + 0.75 * mqkv)                                         # ... and its bound (oracle/checks.py, element form)
    for name, t, want, err in (("db", tb, dqkv.sum(0), edqkv.sum(0)), ("dx", tx, dqkv @ w64.T, edqkv @ np.abs(w64).T),
                               ("dW", tw, x64.T @ dqkv, np.abs(x64).T @ edqkv)):
        gotg = t.grad().float().numpy().astype(np.float64)
        assert np.isfinite(gotg).all() and (np.abs(gotg - want) <= err + eps * np.abs(want)).all(), name
        assert np.linalg.norm(gotg - want) <= 3 * eps * np.linalg.norm(want), name  # and 1 % of the whole gradient's norm
