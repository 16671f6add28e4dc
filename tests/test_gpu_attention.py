"""-m gpu: causal attention forward/backward through the C ABI vs the CPU oracle and golden vectors.

Tolerances (stated):
  f32 (the reference's dtype): rtol = atol = 1e-3 on U(-10,10) inputs — the reference's own bound
      (test_nn.py:11-33 via test/common.py:6-11).
  bf16 / f16 (MFMA and generic kernels): the SCALE-AWARE bounds of oracle/checks.py against the double-precision evaluation
      on the same 16-bit inputs (oracle.attn_ref64; the constants live in oracle/checks.py and only there) - per element
      eps (C_OUT |ref| + C_SUM sum|terms| + coh) + floor, per row ||err|| <= eps (C_ROW ||ref|| + C_Q ||quad||) + ||floor||, per head
      ||err||_F <= eps (C_HEAD ||ref||_F + C_QH ||quad||_F) + ||floor||_F with C_OUT = C_SUM = 2, C_ROW = C_Q = 2.5, C_HEAD = C_QH = 1.25,
      eps = 2^-8 (bf16) | 2^-11 (f16), floor = what the element format cannot hold (format_floor: one absolute rounding of every P / dS
      entry - 1e-5 of the outputs and less; it matters on the reference tests' own U(-10, 10) inputs only); LSE within 2e-6 (1 + |lse|).
      No absolute tolerance anywhere: an output that is all zeros, or short of one 64-key tile,
      fails (tests/test_attention_bounds.py on CPU, tests/test_gpu_attention_mutants.py on the kernels themselves).
"""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import checks as K
from oracle import oracle as O
from tests.helpers import assert_close, golden, regen

pytestmark = pytest.mark.gpu


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
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    H.profile_enable(False)
    sfx = "_d64" if D == 64 else ""
    for label in ("attn_fwd_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dq_mfma"):  # the matrix-core kernels of THIS head size ran, nothing padded
        assert label + sfx in H.profile_results(), (label + sfx, sorted(H.profile_results()))
    K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what=f"{Sq}x{Skv} D{D}")


def _guarded(nbytes, guard=64 * 1024):
    """A device buffer with `guard` bytes of 0xAB behind its `nbytes`: what a kernel must not touch."""
    buf = H.DevBuf(nbytes + guard)
    fill = np.full(guard, 0xAB, dtype=np.uint8)
    H.check(H.lib().kf_memcpy_h2d(buf.ptr + nbytes, fill.ctypes.data, guard, None))
    return buf


def _guard_ok(buf, nbytes, guard=64 * 1024):
    tail = np.empty(guard, dtype=np.uint8)
    H.check(H.lib().kf_memcpy_d2h(tail.ctypes.data, buf.ptr + nbytes, guard, None))
    return bool((tail == 0xAB).all())


# round 6: ANY sequence lengths with Skv >= Sq on the generated streams - a last query block of fewer than 256 rows,
# a last key tile of fewer than 64 keys, a last key block of fewer than 256 keys, a last slice of fewer than 32 queries - at the C ABI itself, no
# padded copies: rows beyond a tensor's end are zero-filled / dropped by the kernels' buffer descriptors. Three heads: the rows "beyond the
# end" of heads 0 and 1 are the NEXT head's rows (they must read as zeros and must not be written), those of head 2 lie behind the tensor.
@pytest.mark.parametrize("code", [H.BF16, H.F16])
@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("Sq,Skv", [(4000, 4000), (1000, 1000), (320, 320), (257, 257), (257, 320), (320, 1000), (1000, 4000), (257, 4000),
                                    (1, 1), (33, 65), (255, 256), (2049, 2049)])
def test_ragged_lengths_run_the_generated_streams(code, D, Sq, Skv):
    B, Hh = 1, 3
    rng = np.random.default_rng(Sq * 7 + Skv + code + D)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                   for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    bq, bk, bv, bgo = (H.DevBuf.from_numpy(x) for x in (q, k, v, go))
    bo, bdq, bdk, bdv = _guarded(q.nbytes), _guarded(q.nbytes), _guarded(k.nbytes), _guarded(k.nbytes)
    blse = _guarded(4 * B * Hh * Sq)
    need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, D)
    ws = _guarded(need)
    H.profile_reset()
    H.profile_enable(True)
    H.attn_fwd(code, B, Hh, Sq, Skv, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, blse.ptr)
    H.attn_bwd(code, B, Hh, Sq, Skv, D, bq.ptr, bk.ptr, bv.ptr, bo.ptr, blse.ptr, bgo.ptr, bdq.ptr, bdk.ptr, bdv.ptr, ws.ptr, need)
    H.device_sync()
    H.profile_enable(False)
    sfx = "_d64" if D == 64 else ""
    for label in ("attn_fwd_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dq_mfma"):
        assert label + sfx in H.profile_results(), (label + sfx, sorted(H.profile_results()))
    for name, buf, n in (("o", bo, q.nbytes), ("dq", bdq, q.nbytes), ("dk", bdk, k.nbytes), ("dv", bdv, k.nbytes), ("lse", blse, 4 * B * Hh * Sq), ("workspace", ws, need)):
        assert _guard_ok(buf, n), f"{name}: bytes behind the tensor were written ({Sq}x{Skv} D{D})"
    o, lse = bo.to_numpy(q.shape, q.dtype), blse.to_numpy((B, Hh, Sq), np.float32)
    dq, dk, dv = bdq.to_numpy(q.shape, q.dtype), bdk.to_numpy(k.shape, k.dtype), bdv.to_numpy(k.shape, k.dtype)
    K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what=f"ragged {Sq}x{Skv} D{D}")
    assert not dk[:, :, Sq:].any() and not dv[:, :, Sq:].any()   # keys no query sees


def test_ragged_lengths_equal_the_zero_padded_problem_bit_for_bit():
    """The descriptors' zero fill IS zero padding: S = 1000 must give the bits of the same tensors padded with zero rows to S = 1024 (the
    padded problem's extra rows dropped) - forward, LSE and all three gradients, both head sizes."""
    code, B, Hh, S, Sp = H.BF16, 2, 2, 1000, 1024
    for D in (128, 64):
        rng = np.random.default_rng(600 + D)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32), code) for _ in range(4))
        o, lse = fwd(code, q, k, v)
        got = bwd(code, q, k, v, o, lse, go)
        pad = lambda x: np.concatenate([x, np.zeros((B, Hh, Sp - S, D), dtype=x.dtype)], axis=2)  # noqa: E731
        op, lsep = fwd(code, pad(q), pad(k), pad(v))
        # (the padded problem's backward wants the padded rows' O / lse as its own forward left them, and dO = 0 there)
        gp = bwd(code, pad(q), pad(k), pad(v), op, lsep, pad(go))
        assert np.array_equal(o.view(np.uint16), op[:, :, :S].view(np.uint16)) and np.array_equal(lse, lsep[:, :, :S])
        for n, a, b in zip(("dq", "dk", "dv"), got, gp):
            assert np.array_equal(a.view(np.uint16), b[:, :, :S].view(np.uint16)), (n, D)


def test_head_size_64_longer_sequences_and_batches():
    """Native D = 64 kernels at sizes where every schedule feature is live: paired blocks (S >= 1024), heads pinned to XCDs
    (B H % 8 == 0), many slice pairs per key block, Sq != Skv."""
    code = H.BF16
    for (B, Hh, Sq, Skv) in ((1, 8, 1024, 1024), (2, 1, 768, 1280), (1, 2, 2048, 2048)):
        rng = np.random.default_rng(64 + Sq + Skv)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                       for s in ((B, Hh, Sq, 64), (B, Hh, Skv, 64), (B, Hh, Skv, 64), (B, Hh, Sq, 64)))
        o, lse = fwd(code, q, k, v)
        got = bwd(code, q, k, v, o, lse, go)
        again = bwd(code, q, k, v, o, lse, go)
        K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=got[0], dk=got[1], dv=got[2], what=f"d64 {Sq}x{Skv}")
        for n, g_, g2 in zip(("dq", "dk", "dv"), got, again):
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
    K.attn_check(qb, kb, vb, code, o=o, lse=lse, what="spiked")


def test_generic_path_16bit_ragged():
    rng = np.random.default_rng(13)
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv, D) in ((2, 2, 65, 33, 64), (1, 3, 40, 72, 80), (1, 1, 17, 17, 256)):
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
            o, lse = fwd(code, q, k, v)
            dq, dk, dv = bwd(code, q, k, v, o, lse, go)
            K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what=f"generic {Sq}x{Skv} D{D}")


def test_full_size_properties():
    """BASELINE config C3 shape per head (S = 4096, D = 128), two heads, every output element: forward, LSE, dQ, dK, dV against the
    double-precision oracle under the scale-aware bounds, plus size-independent properties: (1) causal prefix: the first 256 output
    rows are BIT-identical to attention over the 256-token prefix alone; (2) columns of dV sum to the column sums of dO (rows of P
    sum to 1), within the rounding of the 4096 outputs summed."""
    code, B, Hh, S, D = H.BF16, 1, 2, 4096, 128
    rng = np.random.default_rng(14)
    q, k, v, go = (O.f32_to_bf16(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32)) for _ in range(4))
    o, lse = fwd(code, q, k, v)
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    m = K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what="S 4096")
    assert all(m[n]["row_rel_l2"] < 1e-2 for n in ("o", "dq", "dk", "dv")), m  # every row within 1 % of its own norm
    o_pre, lse_pre = fwd(code, q[:, :, :256].copy(), k[:, :, :256].copy(), v[:, :, :256].copy())
    assert np.array_equal(o[:, :, :256], o_pre) and np.array_equal(lse[:, :, :256].view(np.uint32), lse_pre.view(np.uint32)), "causal prefix"
    want = f(go, code).astype(np.float64).sum(axis=2)
    got = f(dv, code).astype(np.float64).sum(axis=2)
    bound = 2.0 ** -8 * np.abs(f(dv, code).astype(np.float64)).sum(axis=2) + 2.0 ** -8 * np.abs(f(go, code).astype(np.float64)).sum(axis=2) / np.sqrt(S)
    assert (np.abs(got - want) <= bound).all(), "sum dV == sum dO"


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
        K.attn_check(q, k, v, H.BF16, o=o1, lse=l1, d_o=go, dq=g1[0], dk=g1[1], dv=g1[2], what=f"paired {Sq}x{Skv}")


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
        ref = O.attn_ref64(q, k, v, go, code=code)
        res = {}
        for form, env in (("ds", None), ("split", "1")):
            with H.knobs(KF_ATTN_SPLIT_BWD=env):
                small = 3 * ((B * Hh * Sq * 4 + 255) // 256 * 256)   # (Sq is a multiple of 32 here: the row-constant arrays need no pad rows)
                need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, 128)
                # dS as full rows (the default while the cap holds them for every pair): ceil(Sq / 256) x ceil(Skv / 256) squares of 64 tiles of 2 KiB
                nqb, nkb = (Sq + 255) // 256, (Skv + 255) // 256
                assert need == (small if env else small + B * Hh * nqb * nkb * 64 * 2048), (form, need)
                H.profile_reset()
                H.profile_enable(True)
                res[form] = bwd(code, q, k, v, o, lse, go)
                H.profile_enable(False)
                assert ("attn_bwd_dq_mfma_split" if env else "attn_bwd_dq_mfma") in H.profile_results(), (form, H.profile_results())
            K.attn_check(q, k, v, code, d_o=go, dq=res[form][0], dk=res[form][1], dv=res[form][2], ref=ref, what=f"{form} {Sq}x{Skv}")
        assert np.array_equal(res["ds"][1].view(np.uint16), res["split"][1].view(np.uint16))
        assert np.array_equal(res["ds"][2].view(np.uint16), res["split"][2].view(np.uint16))


@pytest.mark.parametrize("S", [512, 500])   # 500: a ragged length on the packed layout (round 6): the rows behind a head's last one are OTHER tensors' bytes there
@pytest.mark.parametrize("code", [H.BF16, H.F16])
@pytest.mark.parametrize("D", [128, 64])
def test_strided_layouts_packed_qkv_are_the_same_arithmetic(code, D, S):
    """kf_attn_*_strided on q / k / v living inside one packed [B S, 3 H D] projection output, o as [B S, H D], and dq / dk / dv
    written into a packed gradient: BIT-identical to the contiguous [B,H,S,D] entries on the same values (the layout changes
    addresses, not arithmetic), the bytes between the strided outputs untouched."""
    B, Hh = 2, 4
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


def _bwd_ws(code, q, k, v, o, lse, go, ws_bytes):
    """kf_attn_bwd with a caller-chosen workspace size."""
    B, Hh, Sq, D = q.shape
    Skv = k.shape[2]
    bufs = [H.DevBuf.from_numpy(x) for x in (q, k, v, o, lse, go)]
    dq, dk, dv = H.DevBuf(q.nbytes), H.DevBuf(k.nbytes), H.DevBuf(v.nbytes)
    ws = H.DevBuf(ws_bytes)
    H.attn_bwd(code, B, Hh, Sq, Skv, D, *[b.ptr for b in bufs], dq.ptr, dk.ptr, dv.ptr, ws.ptr, ws_bytes)
    H.device_sync()
    return dq.to_numpy(q.shape, q.dtype), dk.to_numpy(k.shape, k.dtype), dv.to_numpy(v.shape, v.dtype)


@pytest.mark.parametrize("D", [128, 64])
def test_backward_workspace_is_bounded_any_size_above_the_statistics_is_accepted(D):
    """The backward's workspace contract (include/kfunca_hip.h): three rows of statistics are the minimum, O(B H S); whatever lies beyond
    holds dS for as many (batch, head) pairs at a time as fit. Groups of 3 (no XCD map), 8 and 16 pairs (XCD map on) and the full
    workspace give BIT-identical gradients (the group is a schedule); the minimum alone runs the recomputing dQ kernel - for head size
    64 as well (round 2 refused it) - whose dK / dV are the same bits and whose dQ meets the same bounds. KF_ATTN_DS_CAP_MB caps what
    kf_attn_bwd_workspace_bytes recommends."""
    code, B, Hh, S = H.BF16, 2, 12, 512
    rng = np.random.default_rng(300 + D)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32), code) for _ in range(4))
    o, lse = fwd(code, q, k, v)
    stats = 3 * ((B * Hh * S * 4 + 255) // 256 * 256)
    nb = (S + 255) // 256
    # dS of one pair in its two layouts (include/kfunca_hip.h): FULL ROWS (nb x nb squares of 128 KiB: what the query recommends while the cap holds it
    # for every pair - the dQ kernel streams faster from 2 MiB-aligned rows) and, round 6, THE CAUSAL HALF (nb (nb + 1) / 2 squares), which the library
    # takes whenever the workspace it is given does not hold full rows for all pairs, or under KF_ATTN_DS_TRI=1
    one, half = nb * nb * 64 * 2048, 32 * nb * (nb + 1) * 2048
    full = H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D)
    assert full == stats + B * Hh * one
    with H.knobs(KF_ATTN_DS_TRI="1"):
        assert H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D) == stats + B * Hh * half
        # config C3 (B 8, H 32, S 4096): 8 GiB of dS as full rows, at most 0.55 of that as the causal half
        c3 = H.attn_bwd_workspace_bytes(code, 8, 32, 4096, 4096, D) - 3 * 8 * 32 * 4096 * 4
        assert c3 <= 0.55 * (8 * 32 * 4096 * 4096 * 2), c3
    ref = _bwd_ws(code, q, k, v, o, lse, go, full)
    with H.knobs(KF_ATTN_DS_TRI="1"):   # the causal half for every pair at once: the same bits
        tri = _bwd_ws(code, q, k, v, o, lse, go, stats + B * Hh * half)
    for n, a0, a1 in zip(("dq", "dk", "dv"), ref, tri):
        assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), ("causal half", n)
    sfx = "_d64" if D == 64 else ""
    for pairs in (3, 8, 16, 23):
        H.profile_reset()
        H.profile_enable(True)
        got = _bwd_ws(code, q, k, v, o, lse, go, stats + pairs * one + 100)   # less than full rows for all 24 pairs: groups of the causal half
        H.profile_enable(False)
        res = H.profile_results()
        g = (pairs * one + 100) // half
        g = min(B * Hh, g if g < 8 else g - g % 8)
        groups = -(-B * Hh // g)
        assert res["attn_bwd_dkv_mfma" + sfx][1] == groups and res["attn_bwd_dq_mfma" + sfx][1] == groups, (pairs, res)
        for n, a0, a1 in zip(("dq", "dk", "dv"), ref, got):
            assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), (pairs, n)
    H.profile_reset()
    H.profile_enable(True)
    small = _bwd_ws(code, q, k, v, o, lse, go, stats)  # no room for dS at all: the recomputing form
    H.profile_enable(False)
    assert "attn_bwd_dq_mfma_split" + sfx in H.profile_results() and "attn_bwd_dq_mfma" + sfx not in H.profile_results()
    assert np.array_equal(small[1].view(np.uint16), ref[1].view(np.uint16)) and np.array_equal(small[2].view(np.uint16), ref[2].view(np.uint16))
    K.attn_check(q, k, v, code, d_o=go, dq=small[0], dk=small[1], dv=small[2], what=f"recomputing dQ, D {D}")
    with pytest.raises(H.KfError) as e:
        _bwd_ws(code, q, k, v, o, lse, go, stats - 256)
    assert e.value.code == H.KF_ERR_WORKSPACE
    with H.knobs(KF_ATTN_DS_CAP_MB="4"):  # 4 MiB: not full rows for all 24 pairs -> groups of the causal half, 10 pairs of 384 KiB fit: 8 (a multiple of 8)
        assert H.attn_bwd_workspace_bytes(code, B, Hh, S, S, D) == stats + 8 * half
        capped = bwd(code, q, k, v, o, lse, go)
    for n, a0, a1 in zip(("dq", "dk", "dv"), ref, capped):
        assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), n


@pytest.mark.parametrize("D", [128, 64])
def test_key_blocks_beyond_the_last_query_touch_nothing_behind_the_tensors(D):
    """Skv >> Sq: most 256-key blocks of the dK / dV pass have no query that sees them. Their gradients are zero, and the pass must not
    fetch Q / dO / row-constant slices that lie behind the tensors' last row on their behalf (round 4: the first two slices a block
    requests were not saturated for such blocks - 2 MiB behind a 64 KiB Q here; whether that faulted depended on what the allocator had
    mapped next to it). Values against the oracle, zeros where no query reaches, for both dtypes, paired and unpaired schedules."""
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv) in ((1, 1, 256, 8192), (1, 3, 512, 4096)):
            rng = np.random.default_rng(Sq + Skv + code)
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
            for nopair in (None, "1"):
                with H.knobs(KF_ATTN_NO_PAIR=nopair):
                    o, lse = fwd(code, q, k, v)
                    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
                assert not dk[:, :, Sq:].any() and not dv[:, :, Sq:].any(), (code, D, Sq, Skv, nopair)
            K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what=f"keys beyond the queries {Sq}x{Skv}")


@pytest.mark.parametrize("D", [128, 64])
def test_rescale_path_on_every_tile(D):
    """KF_ATTN_NO_DEFER makes the forward adopt every tile's maximum: the inline rare path of the generated stream (new maximum, the
    copies of -max the score chains start from, this tile's exponents shifted, O and the row sums scaled) then runs on EVERY tile of
    every block instead of almost never. Same bounds as the default schedule; the two agree to the rounding of P."""
    for code in (H.BF16, H.F16):
        for scale_in in (1.0, 3.0):
            B, Hh, S = 1, 2, 1024
            rng = np.random.default_rng(int(7 * scale_in) + code)
            q, k, v = (O.from_float((scale_in * rng.standard_normal((B, Hh, S, D))).astype(np.float32), code) for _ in range(3))
            with H.knobs(KF_ATTN_NO_DEFER=None):
                o0, l0 = fwd(code, q, k, v)
            with H.knobs(KF_ATTN_NO_DEFER="1"):
                o1, l1 = fwd(code, q, k, v)
            K.attn_check(q, k, v, code, o=o1, lse=l1, what=f"rescale on every tile, inputs x{scale_in}")
            K.attn_check(q, k, v, code, o=o0, lse=l0, what=f"deferred maximum, inputs x{scale_in}")
            assert np.abs(l0 - l1).max() <= 1e-5 * (1 + np.abs(l0).max())   # the maximum in use cancels out of lse up to f32 rounding
            d = np.abs(f(o0, code).astype(np.float64) - f(o1, code).astype(np.float64))
            assert d.max() <= 4 * K.EPS[code] * np.abs(f(o0, code)).max()


def _draw_large(dist, rng, shape):
    if dist == "u10":       # the reference's own attention test range: test/test_nn.py:22-24
        return rng.uniform(-10, 10, shape)
    return float(dist[1]) * rng.standard_normal(shape)


@pytest.mark.parametrize("D", [128, 64])
@pytest.mark.parametrize("dist", ["n2", "n3", "u10"])
@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_exact_scores_hold_the_bounds_at_large_logits(code, dist, D):
    """Why the default streams keep q and k as they are, and (round 5) both 16-bit types on the inputs that stress them:
    N(0, 2^2) / N(0, 3^2) (logit std 4 / 9, a peaked softmax) and U(-10, 10), the range the reference's own attention test draws from
    (test/test_nn.py:22-24; logit std 33: one-hot rows, most of P below what f16 - or f32 - can hold). The default kernels (exact f32
    scores; f16: P carried as P 2^14 into the dV product) stay inside every bound on all of them, through the generated streams and
    through the 8-wave forward + 32-key dK/dV kernels; profiles/r05_attn_large_logits.txt has the table (tools/attn_large_logits.py)."""
    B, Hh, S = 1, 2, 1024
    rng = np.random.default_rng({"n2": 20, "n3": 30, "u10": 100}[dist] + code)
    q, k, v, go = (O.from_float(_draw_large(dist, rng, (B, Hh, S, D)).astype(np.float32), code) for _ in range(4))
    ref = O.attn_ref64(q, k, v, go, code=code)
    for knobs in ({}, {"KF_ATTN_FWD_V3": "1", "KF_ATTN_DKV_V4": "1"}):
        with H.knobs(**knobs):
            o, lse = fwd(code, q, k, v)
            dq, dk, dv = bwd(code, q, k, v, o, lse, go)
        m = K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, ref=ref, what=f"exact scores, D {D}, inputs {dist} {knobs}")
        assert max(max(m[n][a] for a in ("element", "row", "head")) for n in K.NAMES) <= 1.0, m   # (attn_check has asserted it per output)


def test_scaled_operands_leave_the_bounds_at_large_logits():
    """The opt-in scaled-operand streams (KF_ATTN_SCALED_OPERANDS: c q and c k rounded to 16 bits once) move every score by
    eps * scale * sum |q k| and leave the bounds at logit std 9 - measured, recorded here so that the trade stays visible."""
    code, B, Hh, S = H.BF16, 1, 2, 1024
    rng = np.random.default_rng(30 + code)
    q, k, v, go = (O.from_float((3.0 * rng.standard_normal((B, Hh, S, 128))).astype(np.float32), code) for _ in range(4))
    ref = O.attn_ref64(q, k, v, go, code=code)
    o, lse = fwd(code, q, k, v)
    with H.knobs(KF_ATTN_SCALED_OPERANDS="1"):
        o2, lse2 = fwd(code, q, k, v)
    assert np.abs(lse2 - ref["lse"]).max() > 100 * np.abs(lse - ref["lse"]).max()    # the price of the faster form at this logit scale
    with pytest.raises(AssertionError):
        K.check_one("o", o2, ref, code, "scaled query, logit std 9")


def test_scaled_operand_streams():
    """KF_ATTN_SCALED_OPERANDS selects the faster forms of both generated kernels (c q / c k rounded to the element type once per
    block): on the operands the parity suite uses they hold the same bounds (lse against that rounding's own bound). The DEFAULT dK / dV
    stream is the arithmetic of the 32-key kernel (both exact-score now): bit-identical gradients."""
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv) in ((1, 2, 512, 512), (2, 8, 2048, 2048), (1, 2, 256, 768)):
            rng = np.random.default_rng(Sq + Skv + code)
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, 128), (B, Hh, Skv, 128), (B, Hh, Skv, 128), (B, Hh, Sq, 128)))
            with H.knobs(KF_ATTN_SCALED_OPERANDS="1"):
                o, lse = fwd(code, q, k, v)
                g = bwd(code, q, k, v, o, lse, go)
                K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=g[0], dk=g[1], dv=g[2], what=f"scaled operands {Sq}x{Skv}", scaled_query=True)
            o, lse = fwd(code, q, k, v)
            g = bwd(code, q, k, v, o, lse, go)
            with H.knobs(KF_ATTN_DKV_V4="1"):
                g4 = bwd(code, q, k, v, o, lse, go)
            for n, a, b in zip(("dq", "dk", "dv"), g, g4):
                if code == H.F16 and n != "dv":
                    # f16, round 5: the generated stream forms dS = p dP' with ONE rounding (v_fma_mix*_f16), the 32-key kernel multiplies
                    # in f32 and converts: two roundings that differ in about one dS entry in 2^12 - dQ / dK agree to an ulp, not to the bit
                    # (an element is a sum of many terms: one term's last bit against the ROW's largest element, not against the element itself)
                    fa, fb = f(a, code).astype(np.float64), f(b, code).astype(np.float64)
                    assert (np.abs(fa - fb) <= 2.0 ** -9 * np.abs(fa).max(axis=-1, keepdims=True) + 2.0 ** -24).all(), (code, n, Sq, Skv)
                    assert (a.view(np.uint16) == b.view(np.uint16)).mean() > 0.9, (code, n, Sq, Skv)
                else:
                    assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), (code, n, Sq, Skv)


@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_head_size_64_generated_streams_against_the_hand_kernels(code):
    """Round 5: head size 64 - the reference's second fast size (causal_attention_kernel.cu:40-52) - runs the generated
    streams too (gen_attn_fwd.py / gen_attn_dkv.py with D = 64) wherever their shape conditions hold. Both the streams and the round-2/3 hand
    kernels (KF_ATTN_FWD_V3 / KF_ATTN_DKV_V4) must hold the bounds against the oracle, and they must agree with each other: the forward to
    the rounding of P (different tile shapes, different moments of adopting a new maximum), dK / dV / dQ of the bf16 stream to the bit (the
    same arithmetic in the same order; f16: dV to the bit, dK / dQ to one rounding of dS - see test_scaled_operand_streams)."""
    for (B, Hh, Sq, Skv) in ((1, 2, 512, 512), (2, 8, 2048, 2048), (1, 2, 256, 768)):
        rng = np.random.default_rng(64 + Sq + Skv + code)
        q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                       for s in ((B, Hh, Sq, 64), (B, Hh, Skv, 64), (B, Hh, Skv, 64), (B, Hh, Sq, 64)))
        ref = O.attn_ref64(q, k, v, go, code=code)
        o, lse = fwd(code, q, k, v)
        g = bwd(code, q, k, v, o, lse, go)
        K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=g[0], dk=g[1], dv=g[2], ref=ref, what=f"D64 streams {Sq}x{Skv}")
        with H.knobs(KF_ATTN_FWD_V3="1", KF_ATTN_DKV_V4="1"):
            o3, lse3 = fwd(code, q, k, v)
            g4 = bwd(code, q, k, v, o, lse, go)          # (the same O and lse: the backward kernels alone are compared)
        K.attn_check(q, k, v, code, o=o3, lse=lse3, d_o=go, dq=g4[0], dk=g4[1], dv=g4[2], ref=ref, what=f"D64 hand kernels {Sq}x{Skv}")
        assert np.abs(lse - lse3).max() <= 1e-5 * (1 + np.abs(lse).max())
        d = np.abs(f(o, code).astype(np.float64) - f(o3, code).astype(np.float64))
        assert d.max() <= 4 * K.EPS[code] * np.abs(f(o, code)).max()
        for n, a, b in zip(("dq", "dk", "dv"), g, g4):
            if code == H.F16 and n != "dv":
                fa, fb = f(a, code).astype(np.float64), f(b, code).astype(np.float64)
                assert (np.abs(fa - fb) <= 2.0 ** -9 * np.abs(fa).max(axis=-1, keepdims=True) + 2.0 ** -24).all(), (code, n, Sq, Skv)
            else:
                assert np.array_equal(a.view(np.uint16), b.view(np.uint16)), (code, n, Sq, Skv)


@pytest.mark.parametrize("D", [128, 64])
def test_generated_dkv_stream_on_ragged_query_counts(D):
    """The dK / dV stream needs Skv % 256 == 0 and nothing of Sq but a multiple of 32: passes of 1, 2, 3, 9 and 25 slices, key blocks with fewer
    slices than the requests the prologue sends ahead (2 at D = 128, 3 at D = 64: clamped re-fetches), queries that end inside a key block.
    (The forward falls back to the 8-wave kernel for Sq % 256 != 0: the pairing of the two is part of the case.)"""
    for code in (H.BF16, H.F16):
        for (B, Hh, Sq, Skv) in ((1, 2, 32, 256), (1, 1, 64, 256), (2, 2, 96, 512), (1, 2, 288, 512), (1, 2, 800, 1024)):
            rng = np.random.default_rng(D + Sq + Skv + code)
            q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code)
                           for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
            o, lse = fwd(code, q, k, v)
            dq, dk, dv = bwd(code, q, k, v, o, lse, go)
            K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=dq, dk=dk, dv=dv, what=f"ragged Sq, D {D} {Sq}x{Skv}")
            if Sq % 128 or Skv % 128:
                continue   # (the 32-key hand kernel wants whole 128-row tiles: off them the knob selects the generic kernels, another arithmetic)
            with H.knobs(KF_ATTN_DKV_V4="1"):
                g4 = bwd(code, q, k, v, o, lse, go)
            assert np.array_equal(dv.view(np.uint16), g4[2].view(np.uint16)), (code, D, Sq, Skv)
            if code == H.BF16:
                assert np.array_equal(dk.view(np.uint16), g4[1].view(np.uint16)) and np.array_equal(dq.view(np.uint16), g4[0].view(np.uint16)), (D, Sq, Skv)


def test_forward_without_an_lse_buffer():
    """kf_attn_fwd's lse pointer may be NULL (a caller that wants no backward): every forward kernel - the generated stream, the 8-wave
    kernel, D = 64 - must skip the store and write the same O."""
    for code in (H.BF16, H.F16):
        for (B, Hh, S, D) in ((1, 2, 512, 128), (1, 2, 384, 128), (1, 2, 256, 64)):
            rng = np.random.default_rng(S + D + code)
            q, k, v = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32), code) for _ in range(3))
            o_ref, _ = fwd(code, q, k, v)
            dq_, dk_, dv_ = (H.DevBuf.from_numpy(x) for x in (q, k, v))
            o = H.DevBuf(q.nbytes)
            H.attn_fwd(code, B, Hh, S, S, D, dq_.ptr, dk_.ptr, dv_.ptr, o.ptr, None)
            H.device_sync()
            assert np.array_equal(o.to_numpy(q.shape, q.dtype), o_ref), (code, S, D)
