"""-m gpu: seeded random-shape sweeps through the operator API against the oracle - the paths that pick kernels by shape
(sort: LDS / radix, views; GEMM: matrix-core kernels behind zero-padding; attention: MFMA kernels behind zero-padding or the
generic kernels) over shapes nobody wrote down by hand."""
import numpy as np
import pytest

import kfunca_amd as kfunca
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_sort_topk_random_shapes_views_dtypes():
    rng = np.random.default_rng(2025)
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
    rng = np.random.default_rng(2026)
    for _ in range(30):
        M, N, K = (int(rng.integers(1, 700)) for _ in range(3))
        dt = ["f32", "f64", "bf16"][int(rng.integers(0, 3))]
        a, b = rng.uniform(-1, 1, (M, K)), rng.uniform(-1, 1, (K, N))
        if dt == "f32":
            a, b, tol = a.astype(np.float32), b.astype(np.float32), 1e-4
            ta, tb = kfunca.from_numpy(a, 0), kfunca.from_numpy(b, 0)
        elif dt == "f64":
            tol = 1e-11
            ta, tb = kfunca.from_numpy(a, 0), kfunca.from_numpy(b, 0)
        else:
            a, b = (O.bf16_to_f32(O.f32_to_bf16(x.astype(np.float32))) for x in (a, b))
            tol = 2e-2
            ta, tb = kfunca.from_numpy(a, 0).bfloat16(), kfunca.from_numpy(b, 0).bfloat16()
        c = kfunca.gemm(ta, tb, 1.0, 0.0)
        got = (c.float() if dt == "bf16" else c).numpy().astype(np.float64)
        a64, b64 = a.astype(np.float64), b.astype(np.float64)
        assert np.abs(got - a64 @ b64).max() <= tol * (np.abs(a64) @ np.abs(b64)).max() + 1e-12, (dt, M, N, K)


def test_attention_random_shapes_dtypes_forward_backward():
    rng = np.random.default_rng(77)
    for _ in range(24):
        B, Hh = int(rng.integers(1, 3)), int(rng.integers(1, 4))
        Sq = int(rng.choice([1, 7, 31, 32, 33, 64, 100, 128, 129, 256, 300, 512]))
        Skv = int(rng.choice([1, 5, 32, 33, 64, 100, 128, 200, 256, 384, 512]))
        D = int(rng.choice([16, 32, 64, 80, 123, 128, 160]))
        dt = ["f32", "bf16", "f16"][int(rng.integers(0, 3))]
        q, k, v, g = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
        if dt == "f32":
            conv, back, tolf, tolb = (lambda x: kfunca.from_numpy(x, 0)), (lambda t: t.numpy()), 3e-5, 1e-4
            ref_in = (q, k, v, g)
        elif dt == "bf16":
            conv, back, tolf, tolb = (lambda x: kfunca.from_numpy(x, 0).bfloat16()), (lambda t: t.float().numpy()), 2e-2, 4e-2
            ref_in = tuple(O.bf16_to_f32(O.f32_to_bf16(x)) for x in (q, k, v, g))
        else:
            conv, back, tolf, tolb = (lambda x: kfunca.from_numpy(x, 0).half()), (lambda t: t.float().numpy()), 4e-3, 1e-2
            ref_in = tuple(x.astype(np.float16).astype(np.float32) for x in (q, k, v, g))
        tq, tk, tv = conv(q), conv(k), conv(v)
        for t in (tq, tk, tv):
            t.set_requires_grad(True)
        out = kfunca.causal_attention(tq, tk, tv)
        out.backward(conv(g))
        o_ref, _ = O.attn_fwd(*ref_in[:3])
        assert np.isfinite(back(out)).all() and np.abs(back(out) - o_ref).max() <= tolf * max(1.0, np.abs(o_ref).max()), (dt, B, Hh, Sq, Skv, D)
        for t, r in zip((tq, tk, tv), O.attn_bwd(*ref_in)):
            gr = back(t.grad())
            assert np.isfinite(gr).all() and np.abs(gr - r).max() <= tolb * max(1.0, np.abs(r).max()), (dt, B, Hh, Sq, Skv, D)
