"""-m gpu: causal attention forward/backward through the C ABI vs the CPU oracle and golden vectors.

Tolerances (stated):
  f32 (the reference's dtype): rtol = atol = 1e-3 on U(-10,10) inputs — the reference's own bound
      (test_nn.py:11-33 via test/common.py:6-11).
  bf16 / f16 MFMA path, U(-1,1) inputs (|O| <= 1): P and dS are rounded to the 16-bit type before
      the second contraction and outputs are rounded once: |err| <= atol + rtol*|ref| with
      bf16: rtol = 2e-2, atol = 2e-2 (fwd) / 3e-2 (bwd);  f16: rtol = 4e-3, atol = 4e-3 / 8e-3.
      The oracle runs f32 math on the same 16-bit-rounded inputs.
"""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.helpers import assert_close, golden, regen

pytestmark = pytest.mark.gpu
TOL = {H.BF16: dict(rtol=2e-2, atol=2e-2), H.F16: dict(rtol=4e-3, atol=4e-3), H.F32: dict(rtol=1e-3, atol=1e-3)}
TOL_BWD = {H.BF16: dict(rtol=2e-2, atol=3e-2), H.F16: dict(rtol=4e-3, atol=8e-3), H.F32: dict(rtol=1e-3, atol=1e-3)}


def fwd(code, q, k, v):
    B, Hh, Sq, D = q.shape
    Skv = k.shape[2]
    dq_, dk_, dv_ = (H.DevBuf.from_numpy(x) for x in (q, k, v))
    o = H.DevBuf(q.nbytes)
    lse = H.DevBuf(4 * B * Hh * Sq)
    H.attn_fwd(code, B, Hh, Sq, Skv, D, dq_.ptr, dk_.ptr, dv_.ptr, o.ptr, lse.ptr)
    H.device_sync()
    return o.to_numpy(q.shape, q.dtype), lse.to_numpy((B, Hh, Sq), np.float32)


def bwd(code, q, k, v, o, lse, go):
    B, Hh, Sq, D = q.shape
    Skv = k.shape[2]
    bufs = [H.DevBuf.from_numpy(x) for x in (q, k, v, o, lse, go)]
    dq, dk, dv = H.DevBuf(q.nbytes), H.DevBuf(k.nbytes), H.DevBuf(v.nbytes)
    need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, D)
    ws = H.DevBuf(need)
    H.attn_bwd(code, B, Hh, Sq, Skv, D, *[b.ptr for b in bufs], dq.ptr, dk.ptr, dv.ptr, ws.ptr, need)
    H.device_sync()
    return dq.to_numpy(q.shape, q.dtype), dk.to_numpy(k.shape, k.dtype), dv.to_numpy(v.shape, v.dtype)


def f(x, code):
    return O.to_float(x, code)


def test_golden_fp32_reference_cases():
    g = golden("attention")
    for i in range(3):  # (2,4,32,256,128), (3,5,64,32,64), (5,16,65,33,123): test_nn.py:13-17
        B, Hh, Sq, Skv, D = (int(x) for x in g[f"fwd{i}_dims"])
        q, k, v = regen(1050 + i, [(B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D)], g[f"fwd{i}_sha"])
        o, lse = fwd(H.F32, q, k, v)
        assert_close(o, g[f"fwd{i}_out"], what=f"fp32 fwd case {i}")
        _, lse_ref = O.attn_fwd(q, k, v)
        assert_close(lse, lse_ref, rtol=1e-4, atol=1e-3, what="lse")


def test_golden_fp32_backward():
    g = golden("attention")
    for i in range(3):
        B, Hh, Sq, Skv, D = (int(x) for x in g[f"bwd{i}_dims"])
        q, k, v, go = regen(1060 + i, [(B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)], g[f"bwd{i}_sha"], lo=-1, hi=1)
        o, lse = fwd(H.F32, q, k, v)
        assert_close(o, g[f"bwd{i}_out"], rtol=1e-4, atol=1e-5, what="fwd")
        dq, dk, dv = bwd(H.F32, q, k, v, o, lse, go)
        for n, got in (("dq", dq), ("dk", dk), ("dv", dv)):
            assert_close(got, g[f"bwd{i}_{n}"], rtol=1e-4, atol=1e-5, what=f"fp32 bwd case {i} {n}")


@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_uniform_scores_exact_structure(code):
    """Q = 0 makes every visible key equally likely: O[m] = mean(V[0..m]). With V holding small integers
    and row counts that are powers of two the expected value is exact — this pins the causal mask,
    the transposed V reads and the accumulator-as-operand k order, independent of exp/rounding."""
    B, Hh, S, D = 1, 2, 256, 128
    rng = np.random.default_rng(3)
    q = np.zeros((B, Hh, S, D), dtype=np.float32)
    k = rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)
    v = rng.integers(-8, 9, (B, Hh, S, D)).astype(np.float32)
    o, lse = fwd(code, *(O.from_float(x, code) for x in (q, k, v)))
    want = np.cumsum(v.astype(np.float64), axis=2) / np.arange(1, S + 1)[None, None, :, None]
    assert_close(f(o, code), want, rtol=2 ** -7, atol=2 ** -6, what="uniform attention")
    for m in (0, 1, 3, 7, 15, 31, 63, 127, 255):  # 1/(m+1) exact: result must match to one final rounding
        assert_close(f(o, code)[:, :, m], want[:, :, m], rtol=2 ** -8, atol=1e-6, what=f"row {m}")
    assert_close(lse, np.log(np.arange(1, S + 1, dtype=np.float64))[None, None, :].repeat(Hh, 1), rtol=1e-5, atol=1e-5, what="lse = log(m+1)")


@pytest.mark.parametrize("code", [H.BF16, H.F16])
@pytest.mark.parametrize("D", [128, 64])  # the reference's two fast head sizes (causal_attention_kernel.cu:25-60), each with native kernels
@pytest.mark.parametrize("B,Hh,Sq,Skv", [(1, 2, 256, 256), (2, 3, 128, 384), (1, 2, 384, 128), (1, 1, 512, 512)])
def test_mfma_path_vs_oracle(code, D, B, Hh, Sq, Skv):
    rng = np.random.default_rng(Sq + Skv + code + D)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                   for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    H.profile_reset()
    H.profile_enable(True)
    o, lse = fwd(code, q, k, v)
    o_ref, lse_ref = O.attn_fwd(q, k, v, code=code)
    assert_close(f(o, code), f(o_ref, code), **TOL[code], what="fwd")
    assert_close(lse, lse_ref, rtol=1e-3, atol=2e-3, what="lse")
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    H.profile_enable(False)
    sfx = "_d64" if D == 64 else ""
    for label in ("attn_fwd_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dq_mfma"):  # the matrix-core kernels of THIS head size ran, nothing padded
        assert label + sfx in H.profile_results(), (label + sfx, sorted(H.profile_results()))
    rq, rk, rv = O.attn_bwd(q, k, v, go, code=code)
    for n, got, want in (("dq", dq, rq), ("dk", dk, rk), ("dv", dv, rv)):
        assert_close(f(got, code), f(want, code), **TOL_BWD[code], what=f"bwd {n}")


def test_head_size_64_longer_sequences_and_batches():
    """Native D = 64 kernels at sizes where every schedule feature is live: paired blocks (S >= 1024), heads pinned to XCDs
    (B H % 8 == 0), many slice pairs per key block, Sq != Skv."""
    code = H.BF16
    for (B, Hh, Sq, Skv) in ((1, 8, 1024, 1024), (2, 1, 768, 1280), (1, 2, 2048, 2048)):
        rng = np.random.default_rng(64 + Sq + Skv)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                       for s in ((B, Hh, Sq, 64), (B, Hh, Skv, 64), (B, Hh, Skv, 64), (B, Hh, Sq, 64)))
        o, lse = fwd(code, q, k, v)
        o_ref, lse_ref = O.attn_fwd(q, k, v, code=code)
        assert_close(f(o, code), f(o_ref, code), **TOL[code], what=f"d64 fwd {Sq}x{Skv}")
        assert_close(lse, lse_ref, rtol=1e-3, atol=2e-3, what="d64 lse")
        got = bwd(code, q, k, v, o, lse, go)
        again = bwd(code, q, k, v, o, lse, go)
        for n, g_, g2, want in zip(("dq", "dk", "dv"), got, again, O.attn_bwd(q, k, v, go, code=code)):
            assert_close(f(g_, code), f(want, code), **TOL_BWD[code], what=f"d64 bwd {n} {Sq}x{Skv}")
            assert np.array_equal(g_.view(np.uint16), g2.view(np.uint16)), f"d64 {n} not reproducible"


def test_rescale_branch_is_exercised():
    """Online-softmax hazard test (guide rule 26): spike one key per tile against every query so the
    running max jumps at a chosen tile and every earlier contribution must be rescaled exactly once."""
    code, B, Hh, S, D = H.BF16, 1, 1, 512, 128
    rng = np.random.default_rng(12)
    q = rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)
    k = rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)
    v = rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)
    q[..., 0] = 4.0
    for j, n in enumerate((70, 200, 330, 460)):  # one spike in four different 64-key tiles, growing
        k[0, 0, n, 0] = 8.0 * (j + 1)
    qb, kb, vb = (O.f32_to_bf16(x) for x in (q, k, v))
    o, lse = fwd(code, qb, kb, vb)
    o_ref, lse_ref = O.attn_fwd(qb, kb, vb, code=code)
    assert_close(f(o, code), f(o_ref, code), **TOL[code], what="spiked fwd")
    assert_close(lse, lse_ref, rtol=1e-3, atol=2e-3, what="spiked lse")


def test_generic_path_16bit_ragged():
    rng = np.random.default_rng(13)
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv, D) in ((2, 2, 65, 33, 64), (1, 3, 40, 72, 80), (1, 1, 17, 17, 256)):
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
            o, lse = fwd(code, q, k, v)
            o_ref, _ = O.attn_fwd(q, k, v, code=code)
            assert_close(f(o, code), f(o_ref, code), **TOL[code], what="generic fwd")
            dq, dk, dv = bwd(code, q, k, v, o, lse, go)
            for got, want in zip((dq, dk, dv), O.attn_bwd(q, k, v, go, code=code)):
                assert_close(f(got, code), f(want, code), **TOL_BWD[code], what="generic bwd")


def test_full_size_properties():
    """BASELINE config C3 shape per head (S = 4096, D = 128), fewer heads: size-independent properties.
    (1) causal prefix: the first 256 output rows equal attention over the 256-token prefix (oracle);
    (2) the last rows vs an f64 numpy evaluation; (3) V = const => O = const, dQ = dK = 0 and
    dV[n] = sum over visible queries of P — columns of dV sum to sum(dO)."""
    code, B, Hh, S, D = H.BF16, 1, 2, 4096, 128
    rng = np.random.default_rng(14)
    q, k, v, go = (O.f32_to_bf16(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)) for _ in range(4))
    o, lse = fwd(code, q, k, v)
    o_pre, lse_pre = O.attn_fwd(q[:, :, :256], k[:, :, :256], v[:, :, :256], code=code)
    assert_close(f(o, code)[:, :, :256], f(o_pre, code), **TOL[code], what="causal prefix")
    assert_close(lse[:, :, :256], lse_pre, rtol=1e-3, atol=2e-3, what="prefix lse")
    qf, kf, vf = (f(x, code).astype(np.float64) for x in (q, k, v))
    for m in (4095, 4032, 2049):
        s = (qf[0, 1, m] @ kf[0, 1, :m + 1].T) / np.sqrt(D)
        p = np.exp(s - s.max())
        p /= p.sum()
        assert_close(f(o, code)[0, 1, m], p @ vf[0, 1, :m + 1], **TOL[code], what=f"row {m}")
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    dq_pre = None
    # dV column sums: sum_n dV[n] = sum_m dO[m] (rows of P sum to 1)
    want = f(go, code).astype(np.float64).sum(axis=2)
    assert_close(f(dv, code).astype(np.float64).sum(axis=2), want, rtol=2e-2, atol=0.5, what="sum dV == sum dO")
    # dQ of the last 64 rows vs f64 numpy for one head
    gf, of = f(go, code).astype(np.float64), f(o, code).astype(np.float64)
    for m in (4095, 3000):
        s = (qf[0, 0, m] @ kf[0, 0, :m + 1].T) / np.sqrt(D)
        p = np.exp(s - s.max())
        p /= p.sum()
        dp = gf[0, 0, m] @ vf[0, 0, :m + 1].T
        ds = p * (dp - (p * dp).sum())
        assert_close(f(dq, code)[0, 0, m], (ds @ kf[0, 0, :m + 1]) / np.sqrt(D), **TOL_BWD[code], what=f"dq row {m}")
    # dK, dV of the LAST key (only the last query sees it)
    m = n = S - 1
    s = (qf[0, 0, m] @ kf[0, 0].T) / np.sqrt(D)
    p = np.exp(s - s.max())
    p /= p.sum()
    dp = gf[0, 0, m] @ vf[0, 0].T
    ds = p * (dp - (p * dp).sum())
    assert_close(f(dv, code)[0, 0, n], p[n] * gf[0, 0, m], **TOL_BWD[code], what="dv last key")
    assert_close(f(dk, code)[0, 0, n], ds[n] * qf[0, 0, m] / np.sqrt(D), **TOL_BWD[code], what="dk last key")


def test_errors():
    a = H.DevBuf(4096)
    with pytest.raises(H.KfError) as e:
        H.attn_fwd(H.I32, 1, 1, 4, 4, 8, a.ptr, a.ptr, a.ptr, a.ptr)
    assert e.value.code == H.KF_ERR_UNSUPPORTED
    with pytest.raises(H.KfError) as e:
        H.attn_fwd(H.F32, 1, 1, 4, 4, 512, a.ptr, a.ptr, a.ptr, a.ptr)
    assert e.value.code == H.KF_ERR_UNSUPPORTED


def test_backward_is_bitwise_reproducible():
    """No atomics anywhere in the backward: dQ, dK, dV must come out bit-identical run after run (this also catches
    LDS-ring races and too-weak counted waits, which show up as run-to-run differences or NaNs in whole waves)."""
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv) in ((2, 4, 1024, 1024), (1, 2, 384, 128), (2, 3, 128, 384)):
            rng = np.random.default_rng(77 + Sq + code)
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, 128), (B, Hh, Skv, 128), (B, Hh, Skv, 128), (B, Hh, Sq, 128)))
            o, lse = fwd(code, q, k, v)
            first = bwd(code, q, k, v, o, lse, go)
            assert all(np.isfinite(f(x, code)).all() for x in first)
            for rep in range(6):
                junk = H.DevBuf(1 << (16 + rep))  # perturb the allocator / timing a little
                again = bwd(code, q, k, v, o, lse, go)
                for nme, a0, a1 in zip(("dq", "dk", "dv"), first, again):
                    assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), (code, Sq, Skv, rep, nme)
                del junk


def test_causal_pairing_is_a_schedule_not_arithmetic():
    """A workgroup that takes a block and its causal mirror computes exactly what two workgroups compute: forward, LSE and all
    three gradients are bit-identical with the pairing switched off (`KF_ATTN_NO_PAIR`), on paired shapes (even block counts),
    unpaired ones (odd counts) and Sq != Skv."""
    for (B, Hh, Sq, Skv) in ((2, 8, 2048, 2048), (1, 8, 1536, 1536), (1, 4, 1024, 2048), (1, 3, 768, 768)):
        rng = np.random.default_rng(55 + Sq + Skv)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), H.BF16)
                       for s in ((B, Hh, Sq, 128), (B, Hh, Skv, 128), (B, Hh, Skv, 128), (B, Hh, Sq, 128)))
        with H.knobs(KF_ATTN_NO_PAIR=None):
            o1, l1 = fwd(H.BF16, q, k, v)
            g1 = bwd(H.BF16, q, k, v, o1, l1, go)
        with H.knobs(KF_ATTN_NO_PAIR="1"):
            o0, l0 = fwd(H.BF16, q, k, v)
            g0 = bwd(H.BF16, q, k, v, o0, l0, go)
        assert np.array_equal(o0, o1) and np.array_equal(l0.view(np.uint32), l1.view(np.uint32)), (Sq, Skv)
        for a0, a1 in zip(g0, g1):
            assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), (Sq, Skv)
        want = O.attn_fwd(q, k, v, code=O.BF16)[0]
        assert np.abs(f(o1, H.BF16) - f(want, H.BF16)).max() < 2e-2


def test_scaled_entry_points_equal_the_default_scale():
    """kf_attn_fwd_scaled / kf_attn_bwd_scaled with scale = 1 / sqrt(D) are the plain entries, bit for bit; another scale
    behaves like pre-scaling the scores (checked against the oracle on Q * (scale * sqrt(D)) in f32)."""
    import ctypes as C
    code, B, Hh, S, D = H.BF16, 1, 2, 256, 128
    rng = np.random.default_rng(91)
    q, k, v, go = (O.f32_to_bf16(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)) for _ in range(4))
    o, lse = fwd(code, q, k, v)
    bufs = [H.DevBuf.from_numpy(x) for x in (q, k, v)]
    o2, l2 = H.DevBuf(q.nbytes), H.DevBuf(4 * B * Hh * S)
    H.check(H.lib().kf_attn_fwd_scaled(code, B, Hh, S, S, D, C.c_float(1.0 / np.sqrt(np.float32(D))), bufs[0].ptr, bufs[1].ptr, bufs[2].ptr,
                                       o2.ptr, l2.ptr, None))
    H.device_sync()
    assert np.array_equal(o2.to_numpy(q.shape, q.dtype), o) and np.array_equal(l2.to_numpy((B, Hh, S), np.float32), lse)
    with pytest.raises(H.KfError):
        H.check(H.lib().kf_attn_fwd_scaled(code, B, Hh, S, S, D, C.c_float(0.0), bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o2.ptr, l2.ptr, None))
    # f32 generic path with half the default scale == default scale on Q / 2
    qf, kf_, vf = (rng.uniform(-1, 1, (1, 1, 40, 32)).astype(np.float32) for _ in range(3))
    b2 = [H.DevBuf.from_numpy(x) for x in (qf, kf_, vf)]
    o3, l3 = H.DevBuf(qf.nbytes), H.DevBuf(4 * 40)
    H.check(H.lib().kf_attn_fwd_scaled(H.F32, 1, 1, 40, 40, 32, C.c_float(0.5 / np.sqrt(np.float32(32))), b2[0].ptr, b2[1].ptr, b2[2].ptr, o3.ptr, l3.ptr, None))
    H.device_sync()
    want, _ = O.attn_fwd(qf * np.float32(0.5), kf_, vf)
    assert_close(o3.to_numpy(qf.shape, np.float32), want, what="explicit scale")


@pytest.mark.parametrize("B,Hh,Sq,Skv,D", [(1, 2, 128, 128, 128), (2, 3, 96, 160, 64), (1, 1, 512, 512, 128), (1, 2, 32, 256, 64), (1, 1, 288, 64, 128)])
def test_f32_mfma_forward_vs_oracle(B, Hh, Sq, Skv, D):
    """The reference's own fast path (f32, D in {64, 128}) on the exact-f32 MFMA: reference tolerance 1e-3 on U(-10, 10)
    inputs (test_nn.py:11-33), and 2e-5 on U(-1, 1) where no logit saturates."""
    rng = np.random.default_rng(Sq * 7 + Skv + D)
    for lo, hi, tol in ((-10, 10, 1e-3), (-1, 1, 2e-5)):
        q, k, v = (rng.uniform(lo, hi, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D)))
        H.profile_reset()
        H.profile_enable(True)
        o, lse = fwd(H.F32, q, k, v)
        H.profile_enable(False)
        assert "attn_fwd_f32_mfma" in H.profile_results()
        o_ref, lse_ref = O.attn_fwd(q, k, v)
        assert_close(o, o_ref, rtol=tol, atol=tol, what=f"f32 mfma fwd {lo}")
        assert_close(lse, lse_ref, rtol=1e-5, atol=1e-3 if hi == 10 else 1e-4, what="lse")


@pytest.mark.parametrize("B,Hh,Sq,Skv,D", [(1, 2, 128, 128, 128), (2, 3, 96, 160, 64), (1, 2, 512, 512, 128), (1, 2, 1024, 1024, 64), (1, 1, 288, 64, 128),
                                           (1, 2, 640, 640, 128), (1, 1, 256, 768, 64)])
def test_f32_mfma_backward_vs_oracle(B, Hh, Sq, Skv, D):
    """f32 backward on the exact-f32 MFMA (no reference counterpart: pinned to the double-precision oracle). Paired and
    unpaired block counts, Sq != Skv both ways, both head sizes; 5e-5 of each gradient's scale on U(-1, 1)."""
    rng = np.random.default_rng(Sq * 3 + Skv + D)
    q, k, v, go = (rng.uniform(-1, 1, s).astype(np.float32) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    o, lse = fwd(H.F32, q, k, v)
    H.profile_reset()
    H.profile_enable(True)
    got = bwd(H.F32, q, k, v, o, lse, go)
    H.profile_enable(False)
    assert {"attn_bwd_dkv_f32_mfma", "attn_bwd_dq_f32_mfma"} <= set(H.profile_results())
    want = O.attn_bwd(q, k, v, go)
    for name, g, w in zip(("dq", "dk", "dv"), got, want):
        scale = np.abs(w).max() + 1e-30
        assert np.isfinite(g).all() and np.abs(g - w).max() <= 5e-5 * scale, (name, float(np.abs(g - w).max() / scale))
    again = bwd(H.F32, q, k, v, o, lse, go)  # no atomics: bitwise reproducible
    for a0, a1 in zip(got, again):
        assert np.array_equal(a0.view(np.uint32), a1.view(np.uint32))


@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_backward_forms_stored_ds_and_recomputing_split(code):
    """Two backward forms behind one entry. Default: the dK/dV kernel stores dS = P o (dP - delta) in 16 bits and the dQ kernel
    computes dQ = scale dS K from it (kernel label attn_bwd_dq_mfma, workspace grows by B H Sq Skv 2 bytes). KF_ATTN_SPLIT_BWD:
    the recomputing dQ kernel (label attn_bwd_dq_mfma_split, small workspace). Both against the oracle; dK and dV are the same
    kernel arithmetic in both forms and must be bit-identical; shapes cover Sq != Skv and a half-filled last query block."""
    for (B, Hh, Sq, Skv) in ((1, 2, 512, 512), (2, 1, 384, 640), (1, 3, 640, 256), (1, 2, 128, 128)):
        rng = np.random.default_rng(21 + Sq + Skv + code)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                       for s in ((B, Hh, Sq, 128), (B, Hh, Skv, 128), (B, Hh, Skv, 128), (B, Hh, Sq, 128)))
        o, lse = fwd(code, q, k, v)
        want = O.attn_bwd(q, k, v, go, code=code)
        res = {}
        for form, env in (("ds", None), ("split", "1")):
            with H.knobs(KF_ATTN_SPLIT_BWD=env):
                small = 3 * ((B * Hh * Sq * 4 + 255) // 256 * 256)
                need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, 128)
                assert need == (small if env else small + B * Hh * ((Sq + 255) // 256) * 256 * Skv * 2), (form, need)
                H.profile_reset()
                H.profile_enable(True)
                res[form] = bwd(code, q, k, v, o, lse, go)
                H.profile_enable(False)
                assert ("attn_bwd_dq_mfma_split" if env else "attn_bwd_dq_mfma") in H.profile_results(), (form, H.profile_results())
            for nme, got, ref in zip(("dq", "dk", "dv"), res[form], want):
                assert_close(f(got, code), f(ref, code), **TOL_BWD[code], what=f"{form} {nme} {Sq}x{Skv}")
        assert np.array_equal(res["ds"][1].view(np.uint16), res["split"][1].view(np.uint16))
        assert np.array_equal(res["ds"][2].view(np.uint16), res["split"][2].view(np.uint16))


@pytest.mark.parametrize("code", [H.BF16, H.F16])
@pytest.mark.parametrize("D", [128, 64])
def test_strided_layouts_packed_qkv_are_the_same_arithmetic(code, D):
    """kf_attn_*_strided on q / k / v living inside one packed [B S, 3 H D] projection output, o as [B S, H D], and dq / dk / dv
    written into a packed gradient: BIT-identical to the contiguous [B,H,S,D] entries on the same values (the layout changes
    addresses, not arithmetic), the bytes between the strided outputs untouched."""
    B, Hh, S = 2, 4, 512
    d = Hh * D
    rng = np.random.default_rng(33 + code + D)
    qkv = O.from_float(rng.uniform(-1, 1, (B * S, 3 * d)).astype(np.float32), code)
    gout = O.from_float(rng.uniform(-1, 1, (B * S, d)).astype(np.float32), code)
    heads = lambda x2: np.ascontiguousarray(x2.reshape(B, S, Hh, D).transpose(0, 2, 1, 3))  # noqa: E731  [B S, H D] -> [B, H, S, D]
    q, k, v, go = heads(qkv[:, :d]), heads(qkv[:, d:2 * d]), heads(qkv[:, 2 * d:]), heads(gout)
    o_ref, lse_ref = fwd(code, q, k, v)
    dq_ref, dk_ref, dv_ref = bwd(code, q, k, v, o_ref, lse_ref, go)
    es = 2
    packed, flat = (S * 3 * d, D, 3 * d), (S * d, D, d)
    bqkv, bgo = H.DevBuf.from_numpy(qkv), H.DevBuf.from_numpy(gout)
    bo, blse = H.DevBuf(B * S * d * es), H.DevBuf(4 * B * Hh * S)
    scale = 1.0 / np.sqrt(D)
    H.attn_fwd_strided(code, B, Hh, S, S, D, scale, bqkv.ptr, packed, bqkv.ptr + d * es, packed, bqkv.ptr + 2 * d * es, packed, bo.ptr, flat, blse.ptr)
    H.device_sync()
    o2 = bo.to_numpy((B * S, d), qkv.dtype)
    assert np.array_equal(heads(o2), o_ref)
    assert np.array_equal(blse.to_numpy((B, Hh, S), np.float32).view(np.uint32), lse_ref.view(np.uint32))
    dqkv = H.DevBuf.from_numpy(np.full((B * S, 3 * d + 8), 0x1234, dtype=np.uint16))  # a wider gradient buffer: 8 pad columns per row
    wide = (S * (3 * d + 8), D, 3 * d + 8)
    need = H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D)
    ws = H.DevBuf(need)
    H.attn_bwd_strided(code, B, Hh, S, S, D, scale, bqkv.ptr, packed, bqkv.ptr + d * es, packed, bqkv.ptr + 2 * d * es, packed, bo.ptr, flat, blse.ptr,
                       bgo.ptr, flat, dqkv.ptr, wide, dqkv.ptr + d * es, wide, dqkv.ptr + 2 * d * es, wide, ws.ptr, need)
    H.device_sync()
    g = dqkv.to_numpy((B * S, 3 * d + 8), np.uint16)
    assert (g[:, 3 * d:] == 0x1234).all()
    for name, got, ref in (("dq", g[:, :d], dq_ref), ("dk", g[:, d:2 * d], dk_ref), ("dv", g[:, 2 * d:3 * d], dv_ref)):
        assert np.array_equal(heads(got).view(np.uint16), ref.view(np.uint16)), name
    # refused off the matrix-core path (D = 96) and for misaligned strides
    with pytest.raises(H.KfError) as e:
        H.attn_fwd_strided(code, B, Hh, S, S, 96, scale, bqkv.ptr, packed, bqkv.ptr, packed, bqkv.ptr, packed, bo.ptr, flat, blse.ptr)
    assert e.value.code == H.KF_ERR_UNSUPPORTED
    with pytest.raises(H.KfError) as e:
        H.attn_fwd_strided(code, B, Hh, S, S, D, scale, bqkv.ptr, (S * 3 * d, D, 3 * d + 1), bqkv.ptr, packed, bqkv.ptr, packed, bo.ptr, flat, blse.ptr)
    assert e.value.code == H.KF_ERR_INVALID
