"""-m gpu: seeded random-shape sweeps through the operator API against the oracle - the paths that pick kernels by shape
(sort: LDS / radix, views; GEMM: matrix-core kernels behind zero-padding; attention: MFMA kernels behind zero-padding or the
generic kernels) over shapes nobody wrote down by hand."""
import os

import numpy as np
import pytest

import kfunca_amd as kfunca
from oracle import checks as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu
SEED = int(os.environ.get("KF_FUZZ_SEED", "0"))  # `KF_FUZZ_SEED=n pytest tests/test_gpu_fuzz.py`: other shapes (0 = the committed sweep)


def test_sort_topk_random_shapes_views_dtypes():
    rng = np.random.default_rng(2025 + 1000 * SEED)
    dts = [np.float32, np.float64, np.int32, np.int64, np.int16, np.uint8, np.int8, np.float16]
    for _ in range(60):
        nd = int(rng.integers(1, 4))
        shape = [int(rng.choice([1, 2, 3, 5, 17, 64, 100, 257, 1000, 9000, 20000])) for _ in range(nd)]
        while np.prod(shape) > 2_000_000:
            shape[int(rng.integers(0, nd))] = 3
        dt = dts[int(rng.integers(0, len(dts)))]
        dim, desc = int(rng.integers(-nd, nd)), bool(rng.integers(0, 2))
        x = rng.integers(0, 255, size=shape).astype(dt) if dt == np.uint8 else rng.uniform(-50, 50, size=shape).astype(dt)
        t = kfunca.from_numpy(x, 0)
        if nd >= 2 and rng.integers(0, 2):
            perm = [int(p) for p in rng.permutation(nd)]
            t, x = t.permute(*perm), np.ascontiguousarray(x.transpose(perm))
        v, i = t.sort(dim, desc)
        wv, wi = O.sort_stable(x, dim % nd, desc)
        assert np.array_equal(v.numpy().view(np.uint8), wv.view(np.uint8)) and np.array_equal(i.numpy(), wi), (shape, dt, dim, desc)
        k = int(rng.integers(0, x.shape[dim % nd] + 1))
        tv, ti = t.topk(k, dim, desc)
        wv2, wi2 = O.topk(x, k, dim % nd, desc)
        assert np.array_equal(tv.numpy().view(np.uint8), wv2.view(np.uint8)) and np.array_equal(ti.numpy(), wi2), (shape, dt, dim, desc, k)


def test_gemm_random_ragged_shapes():
    rng = np.random.default_rng(2026 + 1000 * SEED)
    for _ in range(30):
        M, N, K_ = (int(rng.integers(1, 700)) for _ in range(3))
        dt = ["f32", "f64", "bf16"][int(rng.integers(0, 3))]
        a, b = rng.uniform(-1, 1, (M, K_)), rng.uniform(-1, 1, (K_, N))
        if dt == "f32":
            a, b, tol = a.astype(np.float32), b.astype(np.float32), 1e-4
            ta, tb = kfunca.from_numpy(a, 0), kfunca.from_numpy(b, 0)
        elif dt == "f64":
            tol = 1e-11
            ta, tb = kfunca.from_numpy(a, 0), kfunca.from_numpy(b, 0)
        else:
            a16, b16 = (O.f32_to_bf16(x.astype(np.float32)) for x in (a, b))
            a, b = O.bf16_to_f32(a16), O.bf16_to_f32(b16)
            ta, tb = kfunca.from_numpy(a, 0).bfloat16(), kfunca.from_numpy(b, 0).bfloat16()
            c = kfunca.gemm(ta, tb, 1.0, 0.0)
            # THE 16-bit GEMM bound of the suite (oracle/checks.py gemm_ok: eps |c| + 1e-6 sum |a||b| per element; VERDICT round 5 weak #8: this
            # test accepted 2e-2 of the largest magnitude)
            ok, worst = K.gemm_ok(O.f32_to_bf16(c.float().numpy()), a16, b16, O.BF16)
            assert ok, (dt, M, N, K_, worst)
            continue
        c = kfunca.gemm(ta, tb, 1.0, 0.0)
        got = c.numpy().astype(np.float64)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        assert np.abs(got - a64 @ b64).max() <= tol * (np.abs(a64) @ np.abs(b64)).max() + 1e-12, (dt, M, N, K_)


def test_attention_random_shapes_dtypes_forward_backward():
    """Random shapes and dtypes through the operator API (matrix-core kernels with operator padding, generic kernels for the rest).
    f32: 3e-5 / 1e-4 of the output's scale against the f32 oracle; bf16 / f16: the scale-aware bounds of oracle/checks.py against the
    double-precision oracle (per element, per row, per head; nothing absolute)."""
    rng = np.random.default_rng(77 + 1000 * SEED)
    for _ in range(24):
        B, Hh = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        Sq = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 128, 129, 256, 300, 512]))
        Skv = int(rng.choice([1, 5, 32, 33, 64, 100, 128, 200, 256, 384, 512]))
        D = int(rng.choice([16, 32, 64, 80, 123, 128, 160]))
        dt = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
        q, k, v, g = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
        if dt == "f32":
            conv = lambda x: kfunca.from_numpy(x, 0)  # noqa: E731
        elif dt == "bf16":
            conv, code, bits = (lambda x: kfunca.from_numpy(x, 0).bfloat16()), O.BF16, tuple(O.f32_to_bf16(x) for x in (q, k, v, g))
        else:
            conv, code, bits = (lambda x: kfunca.from_numpy(x, 0).half()), O.F16, tuple(x.astype(np.float16) for x in (q, k, v, g))
        tq, tk, tv = conv(q), conv(k), conv(v)
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(conv(g))
        if dt == "f32":
            o_ref, _ = O.attn_fwd(q, k, v)
            assert np.isfinite(out.numpy()).all() and np.abs(out.numpy() - o_ref).max() <= 3e-5 * max(1.0, np.abs(o_ref).max()), (dt, B, Hh, Sq, Skv, D)
            for t, r in zip((tq, tk, tv), O.attn_bwd(q, k, v, g)):
                gr = t.grad().numpy()
                assert np.isfinite(gr).all() and np.abs(gr - r).max() <= 1e-4 * max(1.0, np.abs(r).max()), (dt, B, Hh, Sq, Skv, D)
        else:
            K.attn_check(*bits[:3], code, o=out.numpy(), d_o=bits[3], dq=tq.grad().numpy(), dk=tk.grad().numpy(), dv=tv.grad().numpy(),
                         what=f"{dt} B{B} H{Hh} {Sq}x{Skv} D{D}")


def test_round2_operators_random_shapes():
    """rms_norm / layer_norm, gemm_fused, causal_attention_qkv and embedding over seeded random shapes and dtypes, forward + backward
    through autograd, against f64 numpy on the dtype-rounded inputs (the kernels behind them are picked by shape: register-tile plans
    and generic norm kernels, every GEMM family incl. split-K, strided attention or its fall-back composition)."""
    rng = np.random.default_rng(2027 + 1000 * SEED)

    def mk(shape, dt, scale=1.0):
        x = (rng.uniform(-1, 1, shape) * scale).astype(np.float32)
        if dt == "bf16":
            x = O.bf16_to_f32(O.f32_to_bf16(x))
        t = kfunca.from_numpy(x, 0)
        t = t.bfloat16() if dt == "bf16" else t
        t.set_requires_grad(True)
        return t, x.astype(np.float64)

    back = lambda t, dt: (t.float() if dt == "bf16" else t).numpy().astype(np.float64)  # noqa: E731
    close = lambda got, want, tol: np.abs(got - want).max() <= tol * max(1.0, np.abs(want).max())  # noqa: E731
    for _ in range(16):  # norms
        dt = ["f32", "bf16"][int(rng.integers(0, 2))]
        rows, cols = int(rng.integers(1, 200)), int(rng.choice([8, 24, 100, 512, 1000, 2048, 4096, 8200, 12288]))
        tol = 1e-4 if dt == "f32" else 4e-2
        tx, x = mk((rows, cols), dt, 2.0)
        tw, w = mk((cols,), dt)
        tb, b = mk((cols,), dt)
        tg, g = mk((rows, cols), dt)
        layer = bool(rng.integers(0, 2))
        y = kfunca.layer_norm(tx, tw, tb, 1e-5) if layer else kfunca.rms_norm(tx, tw, 1e-5)
        y.backward(tg)
        mean = x.mean(1, keepdims=True) if layer else 0.0
        rstd = 1.0 / np.sqrt(((x - mean) ** 2).mean(1, keepdims=True) + 1e-5)
        xh = (x - mean) * rstd
        assert close(back(y, dt), xh * w + (b if layer else 0.0), tol), (dt, rows, cols, layer)
        gg = g * w
        dx = rstd * (gg - (gg.mean(1, keepdims=True) if layer else 0.0) - xh * (gg * xh).mean(1, keepdims=True))
        assert close(back(tx.grad(), dt), dx, tol) and close(back(tw.grad(), dt), (g * xh).sum(0), tol * np.sqrt(rows)), (dt, rows, cols, layer)
        if layer:
            assert close(back(tb.grad(), dt), g.sum(0), tol * np.sqrt(rows))
    for it in range(14):  # gemm_fused
        dt = ["f32", "bf16"][int(rng.integers(0, 2))]
        M, N = (int(rng.choice([1, 7, 64, 128, 200, 256, 384])) for _ in range(2))
        K = int(rng.choice([16, 64, 100, 512, 4096]))
        if it < 2:  # ragged AND big enough for the zero-padded route onto the tile kernels, every tail operand in use (round 5)
            dt, M, N, K = ("bf16", 200, 384, 520) if it == 0 else ("f32", 130, 200, 1000)
        tol = 1e-4 if dt == "f32" else 4e-2
        ta, a = mk((M, K), dt)
        tb, b = mk((K, N), dt, 1.0 / np.sqrt(K))
        tbias, bias = mk((N,), dt)
        tm, m = mk((M, N), dt)
        tadd, add = mk((M, N), dt)
        tg, g = mk((M, N), dt)
        use = [bool(rng.integers(0, 2)) for _ in range(3)]
        if it < 2:
            use = [True, True, True]
        y = kfunca.gemm_fused(ta, tb, 0.5, tbias if use[0] else None, tm if use[1] else None, tadd if use[2] else None)
        y.backward(tg)
        raw = 0.5 * (a @ b) + (bias if use[0] else 0.0)
        assert close(back(y, dt), raw * (m if use[1] else 1.0) + (add if use[2] else 0.0), tol), (dt, M, N, K, use)
        draw = g * (m if use[1] else 1.0)
        assert close(back(ta.grad(), dt), 0.5 * draw @ b.T, tol) and close(back(tb.grad(), dt), 0.5 * a.T @ draw, tol * np.sqrt(M)), (dt, M, N, K, use)
        if use[0]:
            assert close(back(tbias.grad(), dt), draw.sum(0), tol * np.sqrt(M))
        if use[1]:
            assert close(back(tm.grad(), dt), g * raw, tol)
        if use[2]:
            assert close(back(tadd.grad(), dt), g, tol)
    for _ in range(8):  # causal_attention_qkv: strided kernels (bf16, D 128, S % 128 == 0) and the fall-back composition
        dt = ["f32", "bf16"][int(rng.integers(0, 2))]
        B, Hh, D = int(rng.integers(1, 3)), int(rng.integers(1, 4)), int(rng.choice([64, 128]))
        S = int(rng.choice([128, 256, 96, 130]))
        d = Hh * D
        tol = 1e-4 if dt == "f32" else 4e-2
        tqkv, qkv = mk((B * S, 3 * d), dt)
        tg, g = mk((B * S, d), dt)
        out = kfunca.causal_attention_qkv(tqkv, B, S, Hh)
        out.backward(tg)
        heads = lambda x2: np.ascontiguousarray(x2.reshape(B, S, Hh, D).transpose(0, 2, 1, 3)).astype(np.float32)  # noqa: E731
        q, k, v, go = heads(qkv[:, :d]), heads(qkv[:, d:2 * d]), heads(qkv[:, 2 * d:]), heads(g)
        o_ref, _ = O.attn_fwd(q, k, v)
        flat = lambda x4: x4.transpose(0, 2, 1, 3).reshape(B * S, d).astype(np.float64)  # noqa: E731
        assert close(back(out, dt), flat(o_ref), tol), (dt, B, Hh, S, D)
        rq, rk, rv = O.attn_bwd(q, k, v, go)
        want = np.concatenate([flat(rq), flat(rk), flat(rv)], axis=1)
        assert close(back(tqkv.grad(), dt), want, tol), (dt, B, Hh, S, D)
    for _ in range(6):  # embedding
        dt = ["f32", "bf16"][int(rng.integers(0, 2))]
        vocab, dim, n = int(rng.integers(1, 400)), int(rng.choice([1, 8, 33, 256])), int(rng.integers(1, 3000))
        tt, table = mk((vocab, dim), dt)
        idx = rng.integers(-vocab, vocab, size=(n,)).astype(np.int64)
        tg, g = mk((n, dim), dt)
        out = kfunca.embedding(tt, kfunca.from_numpy(idx, 0))
        out.backward(tg)
        assert np.array_equal(back(out, dt), table[idx])
        want = np.zeros((vocab, dim))
        np.add.at(want, np.where(idx < 0, idx + vocab, idx), g)
        assert close(back(tt.grad(), dt), want, 1e-5 if dt == "f32" else 2e-2), (dt, vocab, dim, n)


def test_abi_stress_slice():
    """A 20-second seeded slice of tools/scratch/stress_abi.py (the round's 16 000-case randomised check was not in the
    suite): sorts over every path, ragged kf_gemm in all four layouts and three dtypes, mixed-dtype element-wise on sliced operands and short
    reductions at the C ABI, against numpy / the oracle - bit-exact where the work is integer or a copy, the 16-bit GEMM bound elsewhere."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "scratch" / "stress_abi.py"), str(600 + SEED), "20"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0 and "all agree" in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]


@pytest.mark.parametrize("script,seconds,says", [("stress_abi2.py", 12, "all agree, every guard byte intact"), ("stress_attn.py", 15, "all within the bounds"),
                                                 ("stress_attn_strided.py", 10, "bit-identical to the contiguous entries"), ("stress_gemm_grouped.py", 12, "bit-identical to kf_gemm alone")])
def test_round6_stress_slices(script, seconds, says):
    """Seeded slices of the round's other randomised stresses (their long runs are in profiles/r06_stress_*.txt): norms with leading dimensions + gather / scatter-add + fused
    GEMM tails with guard bytes; ragged attention forward + backward against the f64 oracle's bounds; attention operands in random strided layouts against the contiguous entries, bit for
    bit; grouped GEMM launches (the one-grid backward pair among them) against kf_gemm alone, bit for bit."""
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    r = subprocess.run([sys.executable, str(root / "tools" / "scratch" / script), str(900 + SEED), str(seconds)], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and says in r.stdout, r.stdout[-1500:] + r.stderr[-3000:]
