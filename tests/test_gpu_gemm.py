"""-m gpu: GEMM through the C ABI vs the CPU oracle and the golden vectors.

Tolerances (stated, per dtype):
  f64  : rtol = atol = 1e-3, the reference's own (test_gemm.py:9-17 via test/common.py:6-11); we also
         assert 1e-10 relative, the f64 accumulation bound.
  f32  : the MFMA f32 path is a k-ordered fma chain, the same arithmetic as the oracle's restatement of
         fma_dot_ref (block_utils.h:46-77): we assert |err| <= 4e-7 * sum_k|a||b| (a few f32 ulps of
         the accumulated magnitude) and the reference's 1e-3 on top.
  bf16 / f16: inputs are exact in f32 and products exact; error = f32 accumulation order + one final
         rounding: |err| <= 2^-8 * |c| + 1e-6 * sum_k|a||b| for bf16 (2^-11 for f16).
"""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import checks as K
from oracle import oracle as O
from tests.helpers import assert_close, golden, regen

pytestmark = pytest.mark.gpu


def run_gemm(code, a, b, ta=False, tb=False, alpha=1.0, beta=0.0, c=None, bias=None):
    """a, b are the STORED arrays (already transposed if ta/tb)."""
    M = a.shape[1] if ta else a.shape[0]
    K = a.shape[0] if ta else a.shape[1]
    N = b.shape[0] if tb else b.shape[1]
    da, db = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(b)
    out = np.zeros((M, N), dtype=a.dtype) if c is None else c
    dc = H.DevBuf.from_numpy(out)
    dbias = H.DevBuf.from_numpy(bias) if bias is not None else None
    need = H.gemm_workspace_bytes(code, ta, tb, M, N, K)
    ws = H.DevBuf(need) if need else None
    H.gemm(code, ta, tb, M, N, K, alpha, da.ptr, a.shape[1], db.ptr, b.shape[1], beta, dc.ptr, N,
           H.EPI_BIAS_ROW if bias is not None else H.EPI_NONE, dbias.ptr if dbias else None,
           ws.ptr if ws else None, need)
    H.device_sync()
    return dc.to_numpy((M, N), a.dtype)


def f64(x, code):
    return O.to_float(x, code).astype(np.float64)


def test_golden_f64_reference_case():
    g = golden("gemm")
    a, b = regen(g["f64_seed"][0], [(123, 457), (457, 234)], g["f64_sha"], dtype=np.float64)
    got = run_gemm(H.F64, a, b)
    assert_close(got, g["f64_out"], what="f64 123x457x234 (test_gemm.py:9-17)")
    assert_close(got, g["f64_out"], rtol=1e-10, atol=1e-9, what="f64 tight")


def test_golden_f32_layouts_alpha_beta_backward():
    g = golden("gemm")
    a, b, c, gg = g["f32_a"], g["f32_b"], g["f32_c"], g["f32_g"]
    tol = dict(rtol=1e-4, atol=1e-4)
    assert_close(run_gemm(H.F32, a, b), g["f32_out"], **tol, what="NN")
    assert_close(run_gemm(H.F32, a.T.copy(), b, ta=True), g["f32_out"], **tol, what="TN")
    assert_close(run_gemm(H.F32, a, b.T.copy(), tb=True), g["f32_out"], **tol, what="NT")
    assert_close(run_gemm(H.F32, a.T.copy(), b.T.copy(), ta=True, tb=True), g["f32_out"], **tol, what="TT")
    assert_close(run_gemm(H.F32, a, b, alpha=0.5, beta=2.0, c=c.copy()), g["f32_out_ab"], **tol, what="alpha/beta")
    # backward of C = A B (no reference counterpart; torch autograd fixture): dA = dC B^T, dB = A^T dC
    assert_close(run_gemm(H.F32, gg, b, tb=True), g["f32_da"], **tol, what="dA")
    assert_close(run_gemm(H.F32, a, gg, ta=True), g["f32_db"], **tol, what="dB")


@pytest.mark.parametrize("M,N,K", [(128, 128, 16), (256, 384, 272), (512, 128, 1024), (100, 130, 70), (1, 1, 1), (129, 257, 33)])
def test_f32_vs_oracle(M, N, K):
    rng = np.random.default_rng(M * 7 + N * 3 + K)
    a, b = rng.uniform(-1, 1, (M, K)).astype(np.float32), rng.uniform(-1, 1, (K, N)).astype(np.float32)
    mag = np.abs(a).astype(np.float64) @ np.abs(b).astype(np.float64)
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            got = run_gemm(H.F32, sa, sb, ta, tb)
            want = O.gemm(sa, sb, trans_a=ta, trans_b=tb)
            assert (np.abs(got.astype(np.float64) - want) <= 4e-7 * mag + 1e-30).all(), (ta, tb)
            assert_close(got, want, what=f"f32 {ta} {tb}")
    bias = rng.uniform(-1, 1, (N,)).astype(np.float32)
    c = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    got = run_gemm(H.F32, a, b, alpha=0.75, beta=-1.5, c=c.copy(), bias=bias)
    want = O.gemm(a, b, alpha=0.75, beta=-1.5, c=c.copy(), bias=bias)
    assert_close(got, want, rtol=1e-5, atol=1e-5, what="alpha/beta/bias epilogue")


@pytest.mark.parametrize("M,N,K", [(64, 64, 16), (128, 192, 80), (256, 64, 1024), (320, 448, 272)])
def test_f64_mfma_vs_numpy(M, N, K):
    """f64 on v_mfma_f64_16x16x4_f64 (the reference's GEMM dtypes are f32 and f64; its one GEMM test is f64): every layout,
    alpha / beta / bias, against numpy f64 at the accumulation bound 1e-13 * sum |a||b| (reference bar: 1e-3)."""
    rng = np.random.default_rng(M + N * 5 + K)
    a, b = rng.uniform(-1, 1, (M, K)), rng.uniform(-1, 1, (K, N))
    mag = np.abs(a) @ np.abs(b)
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            H.profile_reset()
            H.profile_enable(True)
            got = run_gemm(H.F64, sa, sb, ta, tb)
            H.profile_enable(False)
            assert "gemm_f64_mfma" in H.profile_results()
            assert (np.abs(got - a @ b) <= 1e-13 * mag + 1e-300).all(), (ta, tb)
    bias, c = rng.uniform(-1, 1, (N,)), rng.uniform(-1, 1, (M, N))
    got = run_gemm(H.F64, a, b, alpha=0.75, beta=-1.5, c=c.copy(), bias=bias)
    assert (np.abs(got - (0.75 * (a @ b) - 1.5 * c + bias[None, :])) <= 1e-13 * (mag + np.abs(c) + 1)).all()


def test_f32_mfma_is_the_fma_chain():
    """v_mfma_f32_32x32x2_f32 accumulates as a k-ordered f32 fma chain (guide: 'bit-for-bit'), which
    is what the oracle restates: with alpha = 1, beta = 0 the two agree exactly."""
    rng = np.random.default_rng(5)
    a, b = rng.uniform(-1, 1, (256, 512)).astype(np.float32), rng.uniform(-1, 1, (512, 128)).astype(np.float32)
    got, want = run_gemm(H.F32, a, b), O.gemm(a, b)
    assert np.array_equal(got, want)


@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_16bit_exact_integers_catch_layout_bugs(code):
    """Small integers are exact in bf16/f16 and in f32 accumulation: any fragment/lane/transposed-read
    mistake shows up as a wrong integer. B is asymmetric, A is not the identity."""
    rng = np.random.default_rng(6)
    M, N, K = (256, 384, 192) if code == H.F16 else (512, 768, 448)  # both run the 128-tile kernel (the 256-tile kernels: test_256_tile_kernel_* below and tests/test_gpu_baseline_sizes.py)
    a = rng.integers(-3, 4, (M, K)).astype(np.float32)
    b = (rng.integers(-2, 3, (K, N)) + (np.arange(N)[None, :] % 3 == 0)).astype(np.float32)
    want = a.astype(np.float64) @ b.astype(np.float64)
    assert np.abs(want).max() < 2048 if code == H.F16 else True
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            got = run_gemm(code, O.from_float(sa, code), O.from_float(sb, code), ta, tb)
            exact = np.abs(want) <= 256  # representable without rounding in bf16
            assert np.array_equal(f64(got, code)[exact], want[exact]), (code, ta, tb)
            assert_close(f64(got, code), want, rtol=2 ** -8, atol=0, what=f"int {code} {ta} {tb}")


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -8), (H.F16, 2.0 ** -11)])
@pytest.mark.parametrize("M,N,K", [(128, 128, 64), (256, 512, 320), (384, 128, 1024), (100, 130, 70), (64, 64, 64), (512, 256, 64), (256, 256, 128), (768, 512, 2048)])
def test_16bit_vs_oracle(code, eps, M, N, K):
    rng = np.random.default_rng(M + N + K + code)
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    af, bf = f64(a, code), f64(b, code)
    mag = np.abs(af) @ np.abs(bf)
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            got = f64(run_gemm(code, sa, sb, ta, tb), code)
            want = af @ bf
            assert (np.abs(got - want) <= eps * np.abs(want) + 1e-6 * mag + 1e-30).all(), (code, ta, tb, M, N, K)
            orc = f64(O.gemm(sa, sb, trans_a=ta, trans_b=tb, code=code), code)
            assert (np.abs(got - orc) <= 2 * eps * np.abs(want) + 2e-6 * mag + 1e-30).all(), "vs oracle"
    bias = O.from_float(rng.uniform(-1, 1, (N,)).astype(np.float32), code)
    c = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    got = f64(run_gemm(code, a, b, alpha=0.5, beta=2.0, c=c.copy(), bias=bias), code)
    want = 0.5 * (af @ bf) + 2.0 * f64(c, code) + f64(bias, code)[None, :]
    assert (np.abs(got - want) <= 2 * eps * np.abs(want) + 2e-6 * mag + 2 * eps).all(), "epilogue"


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -8), (H.F16, 2.0 ** -11)])
def test_256_tile_kernel_layouts_epilogue_and_unaligned_c(code, eps):
    """The 256-tile kernels only run on grids of >= 160 tiles: 2560 x 4096 (K small keeps the oracle cheap) takes the 4-wave
    form (asserted; the 8-wave form: tests/test_gpu_baseline_sizes.py). Every operand
    layout, alpha / beta / row bias, and a C whose rows are not 16-byte aligned (the narrow-store epilogue)."""
    rng = np.random.default_rng(77 + code)
    M, N, K = 2560, 4096, 192
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    af, bf = f64(a, code), f64(b, code)
    want, mag = af @ bf, np.abs(af) @ np.abs(bf)
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            H.profile_reset()
            H.profile_enable(True)
            got = f64(run_gemm(code, sa, sb, ta, tb), code)
            H.profile_enable(False)
            assert set(H.profile_results()) == {"gemm_bf16_mfma" if code == H.BF16 else "gemm_f16_mfma"}
            assert (np.abs(got - want) <= eps * np.abs(want) + 1e-6 * mag + 1e-30).all(), (code, ta, tb)
    bias = O.from_float(rng.uniform(-1, 1, (N,)).astype(np.float32), code)
    c = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    want_e = 0.5 * want + 2.0 * f64(c, code) + f64(bias, code)[None, :]
    for tb in (False, True):
        sb = b.T.copy() if tb else b
        got = f64(run_gemm(code, a, sb, tb=tb, alpha=0.5, beta=2.0, c=c.copy(), bias=bias), code)
        assert (np.abs(got - want_e) <= 2 * eps * np.abs(want_e) + 2e-6 * mag + 2 * eps).all(), ("epilogue", tb)
    # C with ldc = N + 4 (rows 8-byte aligned only): columns beyond N must stay untouched
    ldc = N + 4
    cp = np.full((M, ldc), 7.0, dtype=np.float32)
    cp[:, :N] = f64(c, code)
    cpad = O.from_float(cp, code)
    da, db, dc, dbias = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(b), H.DevBuf.from_numpy(cpad), H.DevBuf.from_numpy(bias)
    assert H.gemm_workspace_bytes(code, False, False, M, N, K) == 0
    H.gemm(code, False, False, M, N, K, 0.5, da.ptr, K, db.ptr, N, 2.0, dc.ptr, ldc, H.EPI_BIAS_ROW, dbias.ptr, None, 0)
    H.device_sync()
    got = f64(dc.to_numpy((M, ldc), a.dtype), code)
    assert (got[:, N:] == 7.0).all()
    assert (np.abs(got[:, :N] - want_e) <= 2 * eps * np.abs(want_e) + 2e-6 * mag + 2 * eps).all(), "unaligned C"


def test_linearity_at_full_size():
    """Size-independent property at BASELINE size 4096^3 bf16: (A1 + A2) B == A1 B + A2 B within rounding,
    and a column-sparse probe: B = one-hot columns selects columns of A exactly."""
    rng = np.random.default_rng(11)
    n = 4096
    a = O.f32_to_bf16(rng.integers(-4, 5, (n, n)).astype(np.float32))
    sel = rng.permutation(n)
    b = np.zeros((n, n), dtype=np.float32)
    b[sel, np.arange(n)] = 1.0  # C[:, j] = A[:, sel[j]]
    got = O.bf16_to_f32(run_gemm(H.BF16, a, O.f32_to_bf16(b)))
    assert np.array_equal(got, O.bf16_to_f32(a)[:, sel])
    got_t = O.bf16_to_f32(run_gemm(H.BF16, a, O.f32_to_bf16(b.T.copy()), tb=True))
    assert np.array_equal(got_t, O.bf16_to_f32(a)[:, sel])
    got_ta = O.bf16_to_f32(run_gemm(H.BF16, a, O.f32_to_bf16(b), ta=True))  # A^T B
    assert np.array_equal(got_ta, O.bf16_to_f32(a).T[:, sel])


def test_errors():
    a = H.DevBuf(1024)
    with pytest.raises(H.KfError) as e:
        H.gemm(H.I32, False, False, 4, 4, 4, 1.0, a.ptr, 4, a.ptr, 4, 0.0, a.ptr, 4)
    assert e.value.code == H.KF_ERR_UNSUPPORTED
    with pytest.raises(H.KfError) as e:
        H.gemm(H.F32, False, False, 4, 4, 4, 1.0, a.ptr, 2, a.ptr, 4, 0.0, a.ptr, 4)
    assert e.value.code == H.KF_ERR_INVALID
    # every kernel reads every operand layout in place: no shape, dtype or layout asks for scratch
    for code in (H.BF16, H.F16, H.F32, H.F64):
        for ta in (False, True):
            for tb in (False, True):
                for (M, N, K) in ((128, 128, 64), (2048, 2048, 2048), (4096, 4096, 4096), (100, 130, 70)):
                    assert H.gemm_workspace_bytes(code, ta, tb, M, N, K) == 0


@pytest.mark.parametrize("code,M,N,K,label", [
    (H.F32, 256, 384, 64, "gemm_f32_mfma_t64"), (H.F32, 2048, 2048, 32, "gemm_f32_mfma"), (H.F64, 128, 192, 48, "gemm_f64_mfma"),
    (H.BF16, 256, 384, 128, "gemm_bf16_mfma_128"), (H.F16, 384, 256, 64, "gemm_f16_mfma_128"),
    (H.BF16, 2560, 4096, 128, "gemm_bf16_mfma"), (H.F16, 2560, 4096, 64, "gemm_f16_mfma"),
    (H.BF16, 2560, 4096, 128, "gemm_bf16_mfma_w8"), (H.F16, 2560, 4096, 64, "gemm_f16_mfma_w8"), (H.BF16, 100, 130, 70, "gemm_generic"),
    (H.F32, 33, 65, 17, "gemm_generic")])
def test_fused_elementwise_tail(code, M, N, K, label):
    with H.knobs(KF_GEMM_W8="1" if label.endswith("_w8") else None):  # the 8-wave 256-tile kernel runs only when asked for (round 3)
        _fused_tail_case(code, M, N, K, label)


def _fused_tail_case(code, M, N, K, label):
    """kf_gemm_ex: C = (alpha A B + beta C + bias) o mul + add with aux = the bracket, on every kernel family (label asserted).
    aux must equal what kf_gemm alone stores (bit for bit: same arithmetic, same rounding); C against f64 numpy on the
    dtype-rounded inputs with the GEMM bound of this file plus one rounding of the result. A wide (ld > N) mul operand and a
    tight add operand exercise both access forms of the 16-bit tail."""
    eps = {H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11, H.F32: 1e-6, H.F64: 1e-13}[code]
    rng = np.random.default_rng(M + N + K + code)
    mk = lambda shp: O.from_float(rng.uniform(-1, 1, shp).astype(np.float32), code) if code != H.F64 else rng.uniform(-1, 1, shp)  # noqa: E731
    a, b, c, bias, mul_w, add = mk((M, K)), mk((K, N)), mk((M, N)), mk((N,)), mk((M, N + 8)), mk((M, N))
    f = lambda x: (O.to_float(x, code) if code != H.F64 else x).astype(np.float64)  # noqa: E731
    da, db, dbias, dmul, dadd = (H.DevBuf.from_numpy(x) for x in (a, b, bias, mul_w, add))
    dc, daux, dplain = H.DevBuf.from_numpy(c), H.DevBuf(c.nbytes), H.DevBuf.from_numpy(c)
    H.gemm(code, 0, 0, M, N, K, 0.5, da.ptr, K, db.ptr, N, 2.0, dplain.ptr, N, H.EPI_BIAS_ROW, dbias.ptr, None, 0)
    H.profile_reset()
    H.profile_enable(True)
    H.gemm_ex(code, 0, 0, M, N, K, 0.5, da.ptr, K, db.ptr, N, 2.0, dc.ptr, N, bias=dbias.ptr, mul=dmul.ptr, ldmul=N + 8, add=dadd.ptr, ldadd=N,
              aux=daux.ptr, ldaux=N)
    H.device_sync()
    H.profile_enable(False)
    assert set(H.profile_results()) == {label}, H.profile_results()
    plain, aux, got = dplain.to_numpy((M, N), a.dtype), daux.to_numpy((M, N), a.dtype), dc.to_numpy((M, N), a.dtype)
    # aux = what kf_gemm alone stores, bit for bit: the tail runs in the same kernel's epilogue (the 4-wave 256-tile kernel's tail form is
    # the same loop; under KF_GEMM_W8 the plain product above ran the 8-wave kernel too)
    assert np.array_equal(aux.view(np.uint8), plain.view(np.uint8)), "aux differs from the plain product"
    raw = 0.5 * (f(a) @ f(b)) + 2.0 * f(c) + f(bias)[None, :]
    mag = np.abs(f(a)) @ np.abs(f(b)) + 2.0 * np.abs(f(c)) + 1.0
    assert (np.abs(f(aux) - raw) <= 2 * eps * np.abs(raw) + 2e-6 * mag * (1 if code != H.F64 else 1e-7) + 2 * eps).all(), "aux"
    want = raw * f(mul_w)[:, :N] + f(add)
    assert (np.abs(f(got) - want) <= 4 * eps * (np.abs(want) + np.abs(raw)) + 2e-6 * mag * (1 if code != H.F64 else 1e-7) + 4 * eps).all(), "C"
    # each operand alone
    for kw, expect in ((dict(mul=dmul.ptr, ldmul=N + 8), raw * f(mul_w)[:, :N]), (dict(add=dadd.ptr, ldadd=N), raw + f(add))):
        H.check(H.lib().kf_memcpy_h2d(dc.ptr, c.ctypes.data, c.nbytes, None))
        H.gemm_ex(code, 0, 0, M, N, K, 0.5, da.ptr, K, db.ptr, N, 2.0, dc.ptr, N, bias=dbias.ptr, **kw)
        H.device_sync()
        got1 = f(dc.to_numpy((M, N), a.dtype))
        assert (np.abs(got1 - expect) <= 4 * eps * (np.abs(expect) + np.abs(raw)) + 2e-6 * mag * (1 if code != H.F64 else 1e-7) + 4 * eps).all(), kw.keys()


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -8), (H.F16, 2.0 ** -11)])
def test_split_k_skinny_products(code, eps):
    """Split-K (kernel label ..._128_splitk): a skinny product - few output tiles, a long contraction - is cut into K slices whose f32
    partial tiles a fold kernel adds in slice order. Every layout, alpha / beta / bias and the element-wise tail, against the oracle
    and f64; bit-identical run to run; without the workspace the same call runs unsplit and agrees within the accumulation bound."""
    rng = np.random.default_rng(99 + code)
    M, N, K = 256, 384, 4096
    need = H.gemm_workspace_bytes(code, False, False, M, N, K)
    assert need == 8 * M * N * 4, need  # 6 tiles x 64 K tiles -> 8 slices of 8 K tiles
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    af, bf_ = f64(a, code), f64(b, code)
    want, mag = af @ bf_, np.abs(af) @ np.abs(bf_)
    label = "gemm_bf16_mfma_128_splitk" if code == H.BF16 else "gemm_f16_mfma_128_splitk"
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (a.T.copy() if ta else a), (b.T.copy() if tb else b)
            H.profile_reset()
            H.profile_enable(True)
            got = run_gemm(code, sa, sb, ta, tb)
            H.profile_enable(False)
            assert set(H.profile_results()) == {label}, H.profile_results()
            assert np.array_equal(got, run_gemm(code, sa, sb, ta, tb)), "split-K result differs run to run"
            gf = f64(got, code)
            assert (np.abs(gf - want) <= eps * np.abs(want) + 1e-6 * mag + 1e-30).all(), (code, ta, tb)
            orc = f64(O.gemm(sa, sb, trans_a=ta, trans_b=tb, code=code), code)
            assert (np.abs(gf - orc) <= 2 * eps * np.abs(want) + 2e-6 * mag + 1e-30).all(), "vs oracle"
    # epilogue + tail through the fold, and the unsplit form of the same call (no workspace passed)
    bias = O.from_float(rng.uniform(-1, 1, (N,)).astype(np.float32), code)
    c = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    x = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    da, db, dbias, dx = (H.DevBuf.from_numpy(t) for t in (a, b, bias, x))
    ws = H.DevBuf(need)
    outs = []
    for wsp, wsb in ((ws.ptr, need), (None, 0)):
        dc = H.DevBuf.from_numpy(c)
        H.profile_reset()
        H.profile_enable(True)
        H.gemm(code, 0, 0, M, N, K, 0.5, da.ptr, K, db.ptr, N, 2.0, dc.ptr, N, H.EPI_BIAS_ROW, dbias.ptr, wsp, wsb)
        H.device_sync()
        H.profile_enable(False)
        assert set(H.profile_results()) == {label if wsp else label[:-7]}, H.profile_results()
        outs.append(f64(dc.to_numpy((M, N), a.dtype), code))
    want_e = 0.5 * want + 2.0 * f64(c, code) + f64(bias, code)[None, :]
    for o in outs:
        assert (np.abs(o - want_e) <= 2 * eps * np.abs(want_e) + 2e-6 * mag + 2 * eps).all()
    with H.knobs(KF_GEMM_NO_SPLITK="1"):
        assert H.gemm_workspace_bytes(code, False, False, M, N, K) == 0


@pytest.mark.parametrize("Kp", [2560, 1792])  # 16 x 10 = 160 tiles each, and 16 x 7 = 112: above the 100-tile line of the 256-tile kernels since round 5
@pytest.mark.parametrize("code", [H.BF16, H.F16])
def test_grouped_backward_pair_is_the_two_products(code, Kp):
    """kf_gemm_grouped on the backward pair of a linear layer (dA = dC W^T, dW = A^T dC) at a shape of the 4-wave 256-tile kernel: ONE
    launch (label ..._pair), results BIT-identical to the two separate kf_gemm launches (same per-tile arithmetic, another grid), with
    alpha / beta; any other list of problems is the per-problem calls in order."""
    rng = np.random.default_rng(123 + code)
    M, N = 4096, 4096  # dA: [M, K'] = dC[M, N] W[K', N]^T -> 16 x (K' / 256) tiles; dW: [K', N] = A^T dC -> the same count
    a = O.from_float(rng.uniform(-1, 1, (M, Kp)).astype(np.float32), code)     # A  [M, K']
    w = O.from_float(rng.uniform(-1, 1, (Kp, N)).astype(np.float32), code)     # W  [K', N]
    g = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)      # dC [M, N]
    c0 = O.from_float(rng.uniform(-1, 1, (M, Kp)).astype(np.float32), code)
    c1 = O.from_float(rng.uniform(-1, 1, (Kp, N)).astype(np.float32), code)
    da, dw, dg = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(w), H.DevBuf.from_numpy(g)
    label = "gemm_bf16_mfma_pair" if code == H.BF16 else "gemm_f16_mfma_pair"
    for alpha, beta in ((1.0, 0.0), (0.5, 2.0)):
        sep0, sep1 = H.DevBuf.from_numpy(c0), H.DevBuf.from_numpy(c1)
        H.gemm(code, 0, 1, M, Kp, N, alpha, dg.ptr, N, dw.ptr, N, beta, sep0.ptr, Kp)   # dA = dC W^T
        H.gemm(code, 1, 0, Kp, N, M, alpha, da.ptr, Kp, dg.ptr, N, beta, sep1.ptr, N)   # dW = A^T dC
        grp0, grp1 = H.DevBuf.from_numpy(c0), H.DevBuf.from_numpy(c1)
        H.profile_reset()
        H.profile_enable(True)
        H.gemm_grouped(code, [(0, 1, M, Kp, N, alpha, beta, dg.ptr, N, dw.ptr, N, grp0.ptr, Kp),
                              (1, 0, Kp, N, M, alpha, beta, da.ptr, Kp, dg.ptr, N, grp1.ptr, N)])
        H.device_sync()
        H.profile_enable(False)
        assert H.profile_results()[label][1] == 1 and len(H.profile_results()) == 1, H.profile_results()
        assert np.array_equal(grp0.to_numpy((M, Kp), a.dtype), sep0.to_numpy((M, Kp), a.dtype)), (alpha, beta)
        assert np.array_equal(grp1.to_numpy((Kp, N), a.dtype), sep1.to_numpy((Kp, N), a.dtype)), (alpha, beta)
    want = f64(g, code) @ f64(w, code).T
    got = f64(H.DevBuf.to_numpy(grp0, (M, Kp), a.dtype), code) if False else None  # (values already pinned through the separate launches' own tests)
    del got, want
    # not the pair pattern (NN + NN): two ordinary launches, same results as kf_gemm
    o0, o1 = H.DevBuf(M * N * 2), H.DevBuf(M * N * 2)
    H.profile_reset()
    H.profile_enable(True)
    H.gemm_grouped(code, [(0, 0, M, N, Kp, 1.0, 0.0, da.ptr, Kp, dw.ptr, N, o0.ptr, N), (0, 0, M, N, Kp, 1.0, 0.0, da.ptr, Kp, dw.ptr, N, o1.ptr, N)])
    H.device_sync()
    H.profile_enable(False)
    assert sum(v[1] for v in H.profile_results().values()) == 2 and label not in H.profile_results()
    assert np.array_equal(o0.to_numpy((M, N), a.dtype), o1.to_numpy((M, N), a.dtype))


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -8), (H.F16, 2.0 ** -11)])
def test_headline_backward_pair_vs_oracle_at_4096(code, eps):
    """The headline GEMM's BACKWARD at its own size: the one-grid pair launch dA = dC W^T, dW = A^T dC at
    4096^3 (label gemm_*_mfma_pair, what bench.py times) against the oracle on sampled rows of both outputs, under the bound the
    forward rows are held to. Bar: test/test_gemm.py:9-17 (the reference's only GEMM test: forward, f64); the backward has no
    reference counterpart (binary_ops.cpp:16-33 is its only GradFunction)."""
    rng = np.random.default_rng(4096 + code)
    n = 4096
    a, w, g = (O.from_float(rng.uniform(-1, 1, (n, n)).astype(np.float32), code) for _ in range(3))
    da, dw, dg = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(w), H.DevBuf.from_numpy(g)
    o_da, o_dw = H.DevBuf(2 * n * n), H.DevBuf(2 * n * n)
    H.profile_reset()
    H.profile_enable(True)
    H.gemm_grouped(code, [(0, 1, n, n, n, 1.0, 0.0, dg.ptr, n, dw.ptr, n, o_da.ptr, n),
                          (1, 0, n, n, n, 1.0, 0.0, da.ptr, n, dg.ptr, n, o_dw.ptr, n)])
    H.device_sync()
    H.profile_enable(False)
    label = "gemm_bf16_mfma_pair" if code == H.BF16 else "gemm_f16_mfma_pair"
    assert set(H.profile_results()) == {label}, H.profile_results()
    rows = [0, 1, 127, 128, 255, 256, 2047, 3000, 4094, 4095]
    # one bound for the suite and for bench.py's spot check (oracle/checks.py gemm_ok): eps |c| + 1e-6 sum |a||b| against f64 mathematics
    a_cols = np.ascontiguousarray(a[:, rows])
    ok_da, frac_da = K.gemm_ok(o_da.to_numpy((n, n), a.dtype)[rows], g[rows], w, code, trans_b=True)
    ok_dw, frac_dw = K.gemm_ok(o_dw.to_numpy((n, n), a.dtype)[rows], a_cols, g, code, trans_a=True)
    assert ok_da, f"dA at {frac_da:.2f} of the bound"
    assert ok_dw, f"dW at {frac_dw:.2f} of the bound"
    # and against the oracle's own (rounded) result: two roundings of nearly the same sum differ by at most one ulp
    got_da, got_dw = f64(o_da.to_numpy((n, n), a.dtype)[rows], code), f64(o_dw.to_numpy((n, n), a.dtype)[rows], code)
    want_da = f64(O.gemm(g[rows], w, trans_b=True, code=code), code)
    want_dw = f64(O.gemm(a_cols, g, trans_a=True, code=code), code)
    mag_da = np.abs(f64(g[rows], code)) @ np.abs(f64(w, code)).T
    mag_dw = np.abs(f64(a_cols, code)).T @ np.abs(f64(g, code))
    assert (np.abs(got_da - want_da) <= 2 * eps * np.abs(want_da) + 2e-6 * mag_da + 1e-30).all(), "dA vs oracle"
    assert (np.abs(got_dw - want_dw) <= 2 * eps * np.abs(want_dw) + 2e-6 * mag_dw + 1e-30).all(), "dW vs oracle"


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -8), (H.F16, 2.0 ** -11)])
def test_float_output_behind_16bit_operands(code, eps):
    """kf_gemm_epilogue.c_f32 / kf_gemm_problem.c_f32 (ABI 6): 16-bit operands, FLOAT C - the f32 accumulators leave unrounded (a weight
    gradient that will be summed over ranks or micro-batches). Every 16-bit kernel family: the 4-wave 256-tile kernel, the 128-tile kernel,
    the generic one, the backward pair as one grid; beta = 1 accumulates in float. Bar: against f64 mathematics within f32 accumulation
    noise (2e-6 sum |a||b|), i.e. ~2^8 times tighter than a 16-bit C allows; rounding the float result gives the 16-bit kernel's bits."""
    rng = np.random.default_rng(640 + code)
    for (M, N, K, label) in ((2048, 4096, 512, "mfma"), (256, 384, 128, "mfma_128"), (100, 130, 70, "generic")):
        a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
        b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
        c0 = rng.uniform(-1, 1, (M, N)).astype(np.float32)
        da, db = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(b)
        want = f64(a, code) @ f64(b, code)
        mag = np.abs(f64(a, code)) @ np.abs(f64(b, code))
        for ta, tb in ((False, False), (True, False), (False, True)):
            sa = np.ascontiguousarray(a.T) if ta else a
            sb = np.ascontiguousarray(b.T) if tb else b
            dsa, dsb = H.DevBuf.from_numpy(sa), H.DevBuf.from_numpy(sb)
            dc = H.DevBuf.from_numpy(c0)
            H.profile_reset()
            H.profile_enable(True)
            H.gemm_ex(code, ta, tb, M, N, K, 0.5, dsa.ptr, sa.shape[1], dsb.ptr, sb.shape[1], 1.0, dc.ptr, N, c_f32=True)
            H.device_sync()
            H.profile_enable(False)
            assert any(label in k for k in H.profile_results()), (label, H.profile_results())
            got = dc.to_numpy((M, N), np.float32).astype(np.float64)
            assert (np.abs(got - (0.5 * want + c0)) <= 2e-6 * mag + 1e-6).all(), (M, N, K, ta, tb)
        # the float result, rounded once, is the 16-bit kernel's output
        dc = H.DevBuf(4 * M * N)
        H.gemm_ex(code, False, False, M, N, K, 1.0, da.ptr, K, db.ptr, N, 0.0, dc.ptr, N, c_f32=True)
        H.device_sync()
        rounded = O.from_float(dc.to_numpy((M, N), np.float32), code)
        assert np.array_equal(rounded, run_gemm(code, a, b))
    # the backward pair as ONE grid with a float dW (256 tiles per product: the pair kernel's shape), sampled rows against f64
    n = 4096
    a, w, g = (O.from_float(rng.uniform(-1, 1, (n, n)).astype(np.float32), code) for _ in range(3))
    da, dw, dg = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(w), H.DevBuf.from_numpy(g)
    o_da, o_dw = H.DevBuf(2 * n * n), H.DevBuf(4 * n * n)
    H.profile_reset()
    H.profile_enable(True)
    H.gemm_grouped(code, [(0, 1, n, n, n, 1.0, 0.0, dg.ptr, n, dw.ptr, n, o_da.ptr, n), (1, 0, n, n, n, 1.0, 0.0, da.ptr, n, dg.ptr, n, o_dw.ptr, n, 1)])
    H.device_sync()
    H.profile_enable(False)
    assert any(k.endswith("_pair") for k in H.profile_results()), H.profile_results()
    rows = [0, 1, 255, 256, 2047, 3000, 4095]
    got_dw = o_dw.to_numpy((n, n), np.float32)[rows].astype(np.float64)
    a_cols = f64(np.ascontiguousarray(a[:, rows]), code)
    want_dw, mag_dw = a_cols.T @ f64(g, code), np.abs(a_cols).T @ np.abs(f64(g, code))
    assert (np.abs(got_dw - want_dw) <= 2e-6 * mag_dw + 1e-6).all()
    sep = H.DevBuf(2 * n * n)
    H.gemm(code, 0, 1, n, n, n, 1.0, dg.ptr, n, dw.ptr, n, 0.0, sep.ptr, n)
    H.device_sync()
    assert np.array_equal(o_da.to_numpy((n, n), a.dtype), sep.to_numpy((n, n), a.dtype))   # dA of the pair: untouched by its neighbour's float output
    with pytest.raises(H.KfError) as e:  # float operands have nothing to gain: refused, not ignored
        H.gemm_ex(H.F32, False, False, 64, 64, 64, 1.0, da.ptr, 64, dw.ptr, 64, 0.0, o_dw.ptr, 64, c_f32=True)
    assert e.value.code == H.KF_ERR_INVALID


@pytest.mark.parametrize("code,M,N,K", [(H.BF16, 1000, 1000, 1000), (H.F16, 300, 520, 700), (H.BF16, 16, 2048, 2048), (H.BF16, 129, 257, 4100), (H.F32, 500, 300, 260),
                                        (H.F32, 33, 2000, 1025)])
def test_ragged_extents_run_on_the_matrix_kernels(code, M, N, K):
    """Extents that are not whole tiles: with caller scratch the product runs on zero-padded images through the tile kernels (round 5) -
    every layout, alpha / beta, bias row, a C with a wider leading dimension whose pad columns must stay untouched; same bounds as whole tiles."""
    rng = np.random.default_rng(M + 3 * N + 7 * K + code)
    eps = {H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11, H.F32: 2.0 ** -20}[code]
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    c = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    bias = O.from_float(rng.uniform(-1, 1, (N,)).astype(np.float32), code)
    want = f64(a, code) @ f64(b, code)
    mag = np.abs(f64(a, code)) @ np.abs(f64(b, code))
    assert H.gemm_workspace_bytes(code, False, False, M, N, K) > 0
    for ta in (False, True):
        for tb in (False, True):
            sa, sb = (np.ascontiguousarray(a.T) if ta else a), (np.ascontiguousarray(b.T) if tb else b)
            H.profile_reset()
            H.profile_enable(True)
            got = f64(run_gemm(code, sa, sb, ta, tb), code)
            H.profile_enable(False)
            names = set(H.profile_results())
            assert "gemm_pad" in names and "gemm_generic" not in names and any("mfma" in n for n in names), names
            assert (np.abs(got - want) <= 2 * eps * np.abs(want) + 2 * eps * mag + 1e-30).all(), (ta, tb)
    want_e = 0.5 * want + 2.0 * f64(c, code) + f64(bias, code)[None, :]
    got = f64(run_gemm(code, a, b, alpha=0.5, beta=2.0, c=c.copy(), bias=bias), code)
    assert (np.abs(got - want_e) <= 2 * eps * np.abs(want_e) + 2 * eps * (mag + 4) + 1e-30).all()
    ldc = N + 3
    cp = np.full((M, ldc), 7.0, dtype=np.float32)
    cpad = O.from_float(cp, code)
    da, db, dc = H.DevBuf.from_numpy(a), H.DevBuf.from_numpy(b), H.DevBuf.from_numpy(cpad)
    need = H.gemm_workspace_bytes(code, False, False, M, N, K)
    ws = H.DevBuf(need)
    H.gemm(code, False, False, M, N, K, 1.0, da.ptr, K, db.ptr, N, 0.0, dc.ptr, ldc, H.EPI_NONE, None, ws.ptr, need)
    H.device_sync()
    out = f64(dc.to_numpy((M, ldc), cpad.dtype), code)
    assert (out[:, N:] == 7.0).all()
    assert (np.abs(out[:, :N] - want) <= 2 * eps * np.abs(want) + 2 * eps * mag + 1e-30).all()
    with H.knobs(KF_GEMM_NO_PAD="1"):
        assert H.gemm_workspace_bytes(code, False, False, M, N, K) == 0


@pytest.mark.parametrize("code,M,N,K", [(H.BF16, 256, 384, 1000), (H.BF16, 250, 384, 1024), (H.F32, 192, 128, 1001), (H.F16, 384, 250, 512)])
def test_ragged_in_one_extent_only(code, M, N, K):
    """Only K ragged (C is written in place, with beta and a bias row), only M or only N ragged (the aligned operand is read where it lies)."""
    rng = np.random.default_rng(M * 5 + N * 11 + K)
    eps = {H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11, H.F32: 2.0 ** -20}[code]
    a = O.from_float(rng.uniform(-1, 1, (M, K)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (K, N)).astype(np.float32), code)
    c = O.from_float(rng.uniform(-1, 1, (M, N)).astype(np.float32), code)
    bias = O.from_float(rng.uniform(-1, 1, (N,)).astype(np.float32), code)
    assert H.gemm_workspace_bytes(code, False, False, M, N, K) > 0
    want = 0.5 * (f64(a, code) @ f64(b, code)) + 2.0 * f64(c, code) + f64(bias, code)[None, :]
    mag = np.abs(f64(a, code)) @ np.abs(f64(b, code)) + 4
    for tb in (False, True):
        sb = np.ascontiguousarray(b.T) if tb else b
        H.profile_reset()
        H.profile_enable(True)
        got = f64(run_gemm(code, a, sb, tb=tb, alpha=0.5, beta=2.0, c=c.copy(), bias=bias), code)
        H.profile_enable(False)
        assert "gemm_generic" not in H.profile_results(), H.profile_results()
        assert (np.abs(got - want) <= 2 * eps * np.abs(want) + 2 * eps * mag + 1e-30).all(), tb


@pytest.mark.parametrize("code", [H.F16, H.BF16, H.F32])
def test_one_term_products_are_the_once_rounded_product(code):
    """K = 1: every output is ONE product, rounded once to the element type - bit for bit, subnormal f16 results included (found by the randomised
    stress, whose bound had no absolute floor: the kernels were right). Every operand layout, ragged M and N, extents of 1."""
    rng = np.random.default_rng(77 + code)
    for (M, N) in ((587, 275), (64, 64), (1, 300), (257, 1), (1, 1)):
        a = O.from_float(rng.uniform(-1, 1, (M, 1)).astype(np.float32), code)
        b = O.from_float(rng.uniform(-1, 1, (1, N)).astype(np.float32), code)
        a[:: 7] = O.from_float(np.full((len(a[::7]), 1), 3e-4, np.float32), code)    # products of two small operands: f16 subnormals
        want = O.from_float((f64(a, code) @ f64(b, code)).astype(np.float32), code)  # (the f32 product of two 16-bit values is exact; of two floats: one rounding)
        for ta in (False, True):
            for tb in (False, True):
                got = run_gemm(code, np.ascontiguousarray(a.T) if ta else a, np.ascontiguousarray(b.T) if tb else b, ta=ta, tb=tb)
                assert np.array_equal(got.view(np.uint8), want.view(np.uint8)), (code, M, N, ta, tb)
