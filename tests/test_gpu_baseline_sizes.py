"""-m gpu: parity at BASELINE.json's own sizes, on the kernels those sizes dispatch to.

Every test (1) runs the configuration through the C ABI at its full size, (2) ASSERTS the kernel that ran by its profile
label (kf_profile_*: the label names the kernel variant, kfunca_amd/csrc/device/gemm.hip / attention.hip), and (3) compares
sampled output rows / sampled (batch, head) pairs with the CPU oracle — the oracle takes a row subset of a GEMM and a head
subset of attention, so the check costs seconds while the launch is the real one.

  C2  fp32 4096^3 fwd + bwd            -> gemm_f32_mfma (the 128-tile form of gemm_f32_kernel), bit-exact vs the fma chain
  C4  bf16 8192 x 8192 x K + epilogue  -> gemm_bf16_mfma (the 4-wave 256-tile kernel: every grid since round 3) and, forced, gemm_bf16_mfma_w8; all four layouts, alpha / beta / bias
  C3  bf16 attention B8 H32 S4096 D128 -> forward and the backward kernels with the XCD block map on
Reference bars: test/test_gemm.py:9-17, test/test_nn.py:11-33, test/common.py:6-11.
"""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import checks as K
from oracle import oracle as O
from tests.helpers import assert_close
from tests.test_gpu_attention import bwd, f, fwd

pytestmark = pytest.mark.gpu


def gemm_dev(code, da, db, dc, M, N, K, ta, tb, lda, ldb, alpha=1.0, beta=0.0, bias=None):
    """One profiled kf_gemm launch on resident buffers; returns the set of kernel labels that ran."""
    H.profile_reset()
    H.profile_enable(True)
    H.gemm(code, ta, tb, M, N, K, alpha, da.ptr, lda, db.ptr, ldb, beta, dc.ptr, N,
           H.EPI_BIAS_ROW if bias is not None else H.EPI_NONE, bias.ptr if bias is not None else None, None, 0)
    H.device_sync()
    H.profile_enable(False)
    return set(H.profile_results())


def rows_of(buf, rows, N, dtype):
    """Download only the sampled rows of a row-major [*, N] device matrix."""
    out = np.empty((len(rows), N), dtype=dtype)
    for i, r in enumerate(rows):
        H.check(H.lib().kf_memcpy_d2h(out[i].ctypes.data, buf.ptr + int(r) * N * out.itemsize, N * out.itemsize, None))
    return out


def test_c2_f32_4096_fwd_bwd_runs_the_128_tile_kernel_bit_exact():
    """C = A W, dA = dC W^T, dW = A^T dC at 4096^3 in f32: the 128-tile form (>= 192 tiles) is the one that runs, and with
    alpha = 1, beta = 0 it is the oracle's k-ordered fma chain bit for bit (block_utils.h:46-77)."""
    n = 4096
    rng = np.random.default_rng(1002)  # seed = 1000 + config number
    a, w, g = (rng.uniform(-1, 1, (n, n)).astype(np.float32) for _ in range(3))
    da, dw, dg, out = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(w), H.DevBuf.from_numpy(g), H.DevBuf(4 * n * n)
    rows = np.sort(rng.choice(n, 48, replace=False))
    rows[0], rows[-1] = 0, n - 1
    cases = (("C = A W (NN)", da, dw, 0, 0, lambda: O.gemm(a[rows], w)),
             ("dA = dC W^T (NT)", dg, dw, 0, 1, lambda: O.gemm(g[rows], w, trans_b=True)),
             ("dW = A^T dC (TN)", da, dg, 1, 0, lambda: O.gemm(np.ascontiguousarray(a[:, rows]), g, trans_a=True)))
    for what, x, y, ta, tb, want_fn in cases:
        out.zero()
        ran = gemm_dev(H.F32, x, y, out, n, n, n, ta, tb, n, n)
        assert ran == {"gemm_f32_mfma"}, (what, ran)  # not gemm_f32_mfma_t64, not gemm_generic
        got, want = rows_of(out, rows, n, np.float32), want_fn()
        assert np.array_equal(got, want), what
    # the reference's own bar (np.matmul in f64, rtol = atol = 1e-3) on the forward rows
    out.zero()
    gemm_dev(H.F32, da, dw, out, n, n, n, 0, 0, n, n)
    assert_close(rows_of(out, rows, n, np.float32), a[rows].astype(np.float64) @ w.astype(np.float64), what="f32 4096^3 vs f64 matmul")


def _bf16_rows_check(got, want64, mag, eps=2.0 ** -8, scale=1.0):
    assert (np.abs(got - want64) <= scale * eps * np.abs(want64) + scale * 1e-6 * mag + scale * eps * 1e-3).all()


@pytest.mark.parametrize("K,w8", [(512, False), (8192, False), (512, True)])
def test_c4_bf16_8192_epilogue_on_both_256_tile_kernels(K, w8):
    """8192 x 8192 x K bf16 with alpha / beta and the fused bias row: 1024 tiles of 256^2 -> the 4-wave kernel (every grid since
    round 3: it is ahead of the 8-wave form at every size measured), and the 8-wave kernel forced with KF_GEMM_W8. K = 512 sweeps
    all four layouts; K = 8192 is config C4 itself (NN and NT). Sampled rows against the oracle (bf16 inputs, f32
    accumulation, one rounding) and against f64 numpy with the bound of tests/test_gpu_gemm.py."""
    with H.knobs(KF_GEMM_W8="1" if w8 else None, KF_GEMM_W4=None):
        _c4_case(K, "gemm_bf16_mfma_w8" if w8 else "gemm_bf16_mfma")


def _c4_case(K, label):
    M = N = 8192
    rng = np.random.default_rng(1004 + K)
    bits = lambda shape: O.f32_to_bf16(rng.uniform(-1, 1, shape).astype(np.float32))
    a, b, c, bias = bits((M, K)), bits((K, N)), bits((M, N)), bits((N,))
    af, bf_, cf, biasf = (O.bf16_to_f32(x).astype(np.float64) for x in (a, b, c, bias))
    rows = np.sort(rng.choice(M, 32, replace=False))
    rows[0], rows[-1] = 0, M - 1
    want_plain = af[rows] @ bf_
    mag = np.abs(af[rows]) @ np.abs(bf_)
    want_epi = 0.5 * want_plain + 2.0 * cf[rows] + biasf[None, :]
    dbias, dc = H.DevBuf.from_numpy(bias), H.DevBuf(2 * M * N)
    layouts = ((0, 0), (0, 1), (1, 0), (1, 1)) if K == 512 else ((0, 0), (0, 1))
    for ta, tb in layouts:
        sa, sb = (np.ascontiguousarray(a.T) if ta else a), (np.ascontiguousarray(b.T) if tb else b)
        da, db = H.DevBuf.from_numpy(sa), H.DevBuf.from_numpy(sb)
        dc.zero()
        ran = gemm_dev(H.BF16, da, db, dc, M, N, K, ta, tb, sa.shape[1], sb.shape[1])
        assert ran == {label}, (ta, tb, ran)
        got = O.bf16_to_f32(rows_of(dc, rows, N, np.uint16)).astype(np.float64)
        _bf16_rows_check(got, want_plain, mag)
        orc_a = np.ascontiguousarray(sa[:, rows]) if ta else sa[rows]
        orc = O.bf16_to_f32(O.gemm(orc_a, sb, trans_a=bool(ta), trans_b=bool(tb), code=O.BF16)).astype(np.float64)
        _bf16_rows_check(got, orc, mag, scale=2.0)
        # alpha / beta / bias row epilogue on the same launch shape
        H.check(H.lib().kf_memcpy_h2d(dc.ptr, c.ctypes.data, c.nbytes, None))
        ran = gemm_dev(H.BF16, da, db, dc, M, N, K, ta, tb, sa.shape[1], sb.shape[1], alpha=0.5, beta=2.0, bias=dbias)
        assert ran == {label}, (ta, tb, ran)
        got = O.bf16_to_f32(rows_of(dc, rows, N, np.uint16)).astype(np.float64)
        assert (np.abs(got - want_epi) <= 2 * 2.0 ** -8 * np.abs(want_epi) + 2e-6 * mag + 2 * 2.0 ** -8).all(), ("epilogue", ta, tb)
        orc = O.bf16_to_f32(O.gemm(orc_a, sb, alpha=0.5, beta=2.0, trans_a=bool(ta), trans_b=bool(tb), c=c[rows], bias=bias,
                                   code=O.BF16)).astype(np.float64)
        assert (np.abs(got - orc) <= 2 * 2.0 ** -8 * np.abs(want_epi) + 2e-6 * mag + 2 * 2.0 ** -8).all(), ("epilogue vs oracle", ta, tb)
        del da, db


@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_8_wave_kernel_forced_at_an_oracle_sized_shape(code):
    """KF_GEMM_W8 forces the 8-wave kernel on a 160-tile grid (2560 x 4096, K = 192): the WHOLE output against the oracle for
    every layout, plus small-integer operands (exact in 16 bits and in the f32 accumulation: any fragment / lane /
    transposed-read mistake is a wrong integer)."""
    eps = 2.0 ** -8 if code == H.BF16 else 2.0 ** -11
    label = "gemm_bf16_mfma_w8" if code == H.BF16 else "gemm_f16_mfma_w8"
    rng = np.random.default_rng(88 + code)
    M, N, K = 2560, 4096, 192
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    ai = rng.integers(-3, 4, (M, K)).astype(np.float32)
    bi = (rng.integers(-2, 3, (K, N)) + (np.arange(N)[None, :] % 3 == 0)).astype(np.float32)
    af, bf_ = O.to_float(a, code).astype(np.float64), O.to_float(b, code).astype(np.float64)
    want, mag, want_i = af @ bf_, np.abs(af) @ np.abs(bf_), ai.astype(np.float64) @ bi.astype(np.float64)
    dc = H.DevBuf(2 * M * N)
    with H.knobs(KF_GEMM_W8="1", KF_GEMM_W4=None):
        for ta in (0, 1):
            for tb in (0, 1):
                for x, y, ref, exact in ((a, b, want, False), (O.from_float(ai, code), O.from_float(bi, code), want_i, True)):
                    sa, sb = (np.ascontiguousarray(x.T) if ta else x), (np.ascontiguousarray(y.T) if tb else y)
                    da, db = H.DevBuf.from_numpy(sa), H.DevBuf.from_numpy(sb)
                    ran = gemm_dev(code, da, db, dc, M, N, K, ta, tb, sa.shape[1], sb.shape[1])
                    assert ran == {label}, (ta, tb, ran)
                    got = O.to_float(dc.to_numpy((M, N), x.dtype), code).astype(np.float64)
                    if exact:
                        ok = np.abs(ref) <= (256 if code == H.BF16 else 2048)
                        assert np.array_equal(got[ok], ref[ok]), (code, ta, tb)
                    else:
                        assert (np.abs(got - ref) <= eps * np.abs(ref) + 1e-6 * mag + 1e-30).all(), (code, ta, tb)
                        orc = O.to_float(O.gemm(sa, sb, trans_a=bool(ta), trans_b=bool(tb), code=code), code).astype(np.float64)
                        assert (np.abs(got - orc) <= 2 * eps * np.abs(ref) + 2e-6 * mag + 1e-30).all(), "vs oracle"


def _rand16(rng, shape, code):
    return O.from_float((rng.random(shape, dtype=np.float32) * 2.0 - 1.0), code)


def test_backward_with_the_xcd_block_map_vs_oracle_and_without_it():
    """B * H = 8: every attention kernel remaps its blocks so that one (b, h) lands on one XCD (a.xcd_map). Forward, dQ, dK, dV
    against the oracle with the map ON, and bit-identical results with KF_ATTN_NO_XCD (the map is a schedule, not arithmetic)."""
    code, B, Hh, S, D = H.BF16, 2, 4, 1024, 128
    rng = np.random.default_rng(1003)
    q, k, v, go = (_rand16(rng, (B, Hh, S, D), code) for _ in range(4))
    with H.knobs(KF_ATTN_NO_XCD=None):
        o, lse = fwd(code, q, k, v)
        grads = bwd(code, q, k, v, o, lse, go)
    K.attn_check(q, k, v, code, o=o, lse=lse, d_o=go, dq=grads[0], dk=grads[1], dv=grads[2], what="xcd map on")  # scale-aware bounds
    with H.knobs(KF_ATTN_NO_XCD="1"):
        o0, lse0 = fwd(code, q, k, v)
        grads0 = bwd(code, q, k, v, o0, lse0, go)
    assert np.array_equal(o0, o) and np.array_equal(lse0.view(np.uint32), lse.view(np.uint32))
    for name, a0, a1 in zip(("dq", "dk", "dv"), grads0, grads):
        assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), name


@pytest.mark.parametrize("D", [128, 64])
def test_generated_streams_are_schedule_independent_and_repeatable(D):
    """The generated streams (both head sizes) keep their data in LDS rings with counted waits and one barrier per tile / slice: a
    synchronisation slip would show as run-to-run differences or as a dependence on the block schedule. A chip-filling problem (B * H = 64,
    S = 2048: paired blocks, 512 workgroups) three times, then with the XCD map off and with unpaired blocks: every output bit-identical."""
    code, B, Hh, S = H.BF16, 4, 16, 2048
    rng = np.random.default_rng(2048 + D)
    q, k, v, go = (_rand16(rng, (B, Hh, S, D), code) for _ in range(4))
    o, lse = fwd(code, q, k, v)
    grads = bwd(code, q, k, v, o, lse, go)
    sl = (slice(1, 2), slice(5, 6))
    K.attn_check(q[sl], k[sl], v[sl], code, o=o[sl], lse=lse[sl], d_o=go[sl], dq=grads[0][sl], dk=grads[1][sl], dv=grads[2][sl], what=f"D{D} (1,5)")
    for knobs in ({}, {}, {"KF_ATTN_NO_XCD": "1"}, {"KF_ATTN_NO_PAIR": "1"}):
        with H.knobs(**knobs):
            o1, lse1 = fwd(code, q, k, v)
            g1 = bwd(code, q, k, v, o, lse, go)
        assert np.array_equal(o1, o) and np.array_equal(lse1.view(np.uint32), lse.view(np.uint32)), (D, knobs)
        for name, a0, a1 in zip(("dq", "dk", "dv"), grads, g1):
            assert np.array_equal(a0.view(np.uint16), a1.view(np.uint16)), (D, knobs, name)


@pytest.mark.parametrize("Hh,D", [(32, 128), (16, 64)])
def test_c3_full_config_sampled_heads_vs_oracle(Hh, D):
    """Config C3 itself: bf16, B = 8, H = 32, S = 4096, D = 128 (B * H = 256: XCD map and causal pairing on, as in bench.py), and its
    twin on the head-size-64 kernels (B = 8, H = 16, S = 4096, D = 64: the same schedule features at a quarter of the bytes).
    Distinct random data in every (b, h); forward, LSE, dQ, dK and dV of sampled (b, h) pairs - every element of them - against the
    double-precision oracle run on exactly those heads under the scale-aware bounds of oracle/checks.py (no absolute tolerance: a
    dropped key tile or an all-zero gradient fails), and a checksum over ALL heads: sum_n dV[b,h,n,:] = sum_m dO[b,h,m,:] (rows of
    P sum to 1) within the rounding of the 4096 summed outputs."""
    code, B, S = H.BF16, 8, 4096
    rng = np.random.default_rng(1003 + D)
    q, k, v, go = (_rand16(rng, (B, Hh, S, D), code) for _ in range(4))
    H.profile_reset()
    H.profile_enable(True)
    o, lse = fwd(code, q, k, v)
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    H.profile_enable(False)
    ran = set(H.profile_results())
    sfx = "_d64" if D == 64 else ""
    assert {"attn_fwd_mfma" + sfx, "attn_bwd_dkv_mfma" + sfx, "attn_bwd_dq_mfma" + sfx} <= ran and not any("generic" in x for x in ran), ran
    pairs = [(0, 0), (7, Hh - 1), (3, Hh // 2 + 1), (5, 8)]  # first, last, and two in the middle (different XCDs)
    for b, h in pairs:
        sl = (slice(b, b + 1), slice(h, h + 1))
        m = K.attn_check(q[sl], k[sl], v[sl], code, o=o[sl], lse=lse[sl], d_o=go[sl], dq=dq[sl], dk=dk[sl], dv=dv[sl], what=f"({b},{h})")
        assert all(m[n]["row_rel_l2"] < 1e-2 for n in ("o", "dq", "dk", "dv")), m  # every row within 1 % of its own norm
    want = f(go, code).astype(np.float64).sum(axis=2)
    got = f(dv, code).astype(np.float64).sum(axis=2)
    bound = 2.0 ** -8 * (np.abs(f(dv, code).astype(np.float64)).sum(axis=2) + np.abs(f(go, code).astype(np.float64)).sum(axis=2) / np.sqrt(S))
    assert (np.abs(got - want) <= bound).all(), "sum dV == sum dO, all heads"
    assert np.isfinite(f(dq, code)).all() and np.isfinite(f(dk, code)).all()


def test_c3_at_a_ragged_length_full_batch():
    """Config C3's batch and heads at S = 4000 (round 6: ragged lengths on the generated streams, no padded copies): B = 8, H = 32, D = 128 -
    XCD map on, 16 query / key blocks per head of which the last is 160 rows, 125 slices. Two sampled (b, h) pairs against the oracle, the
    checksum sum_n dV = sum_m dO over all 256 heads, and the causal-prefix property: the first 1000 rows of O / LSE are those of the S = 1000
    problem on the same tensors' prefixes, bit for bit (a row's result depends on nothing behind it)."""
    code, B, Hh, S, D = H.BF16, 8, 32, 4000, 128
    rng = np.random.default_rng(4000)
    q, k, v, go = (_rand16(rng, (B, Hh, S, D), code) for _ in range(4))
    H.profile_reset()
    H.profile_enable(True)
    o, lse = fwd(code, q, k, v)
    dq, dk, dv = bwd(code, q, k, v, o, lse, go)
    H.profile_enable(False)
    ran = set(H.profile_results())
    assert {"attn_fwd_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dq_mfma"} <= ran and not any("generic" in x for x in ran), ran
    for b, h in ((0, 0), (7, 31)):
        sl = (slice(b, b + 1), slice(h, h + 1))
        K.attn_check(q[sl], k[sl], v[sl], code, o=o[sl], lse=lse[sl], d_o=go[sl], dq=dq[sl], dk=dk[sl], dv=dv[sl], what=f"S=4000 ({b},{h})")
    want = f(go, code).astype(np.float64).sum(axis=2)
    got = f(dv, code).astype(np.float64).sum(axis=2)
    bound = 2.0 ** -8 * (np.abs(f(dv, code).astype(np.float64)).sum(axis=2) + np.abs(f(go, code).astype(np.float64)).sum(axis=2) / np.sqrt(S))
    assert (np.abs(got - want) <= bound).all(), "sum dV == sum dO, all heads"
    cut = 1000
    sub = (slice(0, 2), slice(0, 4))
    o1, lse1 = fwd(code, *(np.ascontiguousarray(x[sub][:, :, :cut]) for x in (q, k, v)))
    assert np.array_equal(o1, o[sub][:, :, :cut]) and np.array_equal(lse1.view(np.uint32), lse[sub][:, :, :cut].view(np.uint32))
