"""-m gpu: rms_norm / layer_norm forward + backward and the embedding gather / scatter-add through the C ABI, against the CPU
oracle and the torch-CPU fixtures (tests/golden/norms.npz).

Tolerances (stated): f32 - the kernels hold a row in registers and use exact two-pass statistics in f32: forward rtol = atol =
1e-5 vs the double oracle, backward 1e-4 (column sums over up to thousands of rows in f32). 16-bit - one rounding of the output
on top of f32 arithmetic on the rounded inputs: rtol = 2^-7 (bf16) / 2^-10 (f16), atol scaled by the magnitude of the terms.
Gather: bit-exact (byte mover). Scatter-add: f32 sums in input order - exact equality with the same sums taken in that order.
"""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.helpers import assert_close, golden

pytestmark = pytest.mark.gpu


def run_fwd(kind, x, w, b, code, eps=1e-5):
    cols = x.shape[-1]
    rows = x.size // cols
    dx_, dy_ = H.DevBuf.from_numpy(x), H.DevBuf(x.nbytes)
    dw_ = H.DevBuf.from_numpy(w) if w is not None else None
    db_ = H.DevBuf.from_numpy(b) if b is not None else None
    mean, rstd = H.DevBuf(4 * rows), H.DevBuf(4 * rows)
    H.norm_fwd(kind, code, rows, cols, dx_.ptr, dw_.ptr if dw_ else None, db_.ptr if db_ else None, eps, dy_.ptr, mean.ptr, rstd.ptr)
    H.device_sync()
    return dy_.to_numpy(x.shape, x.dtype), mean.to_numpy((rows,), np.float32), rstd.to_numpy((rows,), np.float32)


def run_bwd(kind, x, w, mean, rstd, g, code, want_db):
    cols = x.shape[-1]
    rows = x.size // cols
    bx, bg, bm, br = H.DevBuf.from_numpy(x), H.DevBuf.from_numpy(g), H.DevBuf.from_numpy(mean), H.DevBuf.from_numpy(rstd)
    bw = H.DevBuf.from_numpy(w) if w is not None else None
    dx, dw = H.DevBuf(x.nbytes), H.DevBuf(cols * x.itemsize)
    db = H.DevBuf(cols * x.itemsize) if want_db else None
    ws = H.norm_bwd(kind, code, rows, cols, bx.ptr, bw.ptr if bw else None, bm.ptr, br.ptr, bg.ptr, dx.ptr, dw.ptr, db.ptr if db else None)
    H.device_sync()
    del ws
    return dx.to_numpy(x.shape, x.dtype), dw.to_numpy((cols,), x.dtype), db.to_numpy((cols,), x.dtype) if db else None


def test_golden_f32_forward_backward():
    g = golden("norms")
    for i in range(5):
        x, go, w, b = g[f"n{i}_x"], g[f"n{i}_g"], g[f"n{i}_w"], g[f"n{i}_b"]
        for kind, name in ((H.NORM_RMS, "rms"), (H.NORM_LAYER, "layer")):
            y, mean, rstd = run_fwd(kind, x, w, b if kind == H.NORM_LAYER else None, H.F32)
            assert_close(y, g[f"n{i}_{name}_y"], rtol=1e-5, atol=1e-5, what=f"{name} fwd {i}")
            dx, dw, db = run_bwd(kind, x, w, mean, rstd, go, H.F32, kind == H.NORM_LAYER)
            assert_close(dx, g[f"n{i}_{name}_dx"], rtol=1e-4, atol=1e-5, what=f"{name} dx {i}")
            assert_close(dw, g[f"n{i}_{name}_dw"], rtol=1e-4, atol=1e-4, what=f"{name} dw {i}")
            if db is not None:
                assert_close(db, g[f"n{i}_layer_db"], rtol=1e-4, atol=1e-4, what=f"db {i}")


@pytest.mark.parametrize("code,eps", [(H.BF16, 2.0 ** -7), (H.F16, 2.0 ** -10), (H.F32, 1e-5)])
@pytest.mark.parametrize("rows,cols", [(3, 8), (70, 512), (33, 2048), (17, 4096), (9, 8192), (5, 16384), (2, 32768), (11, 1000), (6, 12288),
                                       (1030, 16), (50, 32), (1000, 64), (333, 128), (77, 256), (41, 1536)])  # (8 / 16 / 32 lanes per row: several rows per wave)
def test_vs_oracle_every_plan(code, eps, rows, cols):
    """Every register-tile plan (one wave per row with 1 / 2 / 4 packs, one block per row with 2..16 packs), the generic kernels
    (row length not a multiple of the pack; rows beyond the backward's tile) and no-weight / no-bias forms, vs the oracle."""
    rng = np.random.default_rng(rows * 31 + cols + code)
    x = O.from_float(rng.uniform(-3, 3, (rows, cols)).astype(np.float32), code)
    go = O.from_float(rng.uniform(-1, 1, (rows, cols)).astype(np.float32), code)
    w = O.from_float(rng.uniform(0.5, 1.5, (cols,)).astype(np.float32), code)
    b = O.from_float(rng.uniform(-1, 1, (cols,)).astype(np.float32), code)
    f = lambda a: O.to_float(a, code).astype(np.float64)  # noqa: E731
    for kind in (H.NORM_RMS, H.NORM_LAYER):
        for ww, bb in ((w, b if kind == H.NORM_LAYER else None), (None, None)):
            y, mean, rstd = run_fwd(kind, x, ww, bb, code)
            y_ref, mean_ref, rstd_ref = O.norm_fwd(kind, x, ww, bb, code=code)
            assert_close(rstd, rstd_ref, rtol=1e-5, atol=1e-7, what="rstd")
            if kind == H.NORM_LAYER:
                assert_close(mean, mean_ref, rtol=1e-5, atol=1e-6, what="mean")
            assert_close(f(y), f(y_ref), rtol=2 * eps, atol=2 * eps, what=f"fwd kind {kind}")
            dx, dw, db = run_bwd(kind, x, ww, mean, rstd, go, code, kind == H.NORM_LAYER)
            rx, rw, rb = O.norm_bwd(kind, x, ww, go, code=code)
            assert_close(f(dx), f(rx), rtol=4 * eps, atol=4 * eps, what=f"dx kind {kind}")
            scale = np.sqrt(rows)  # column sums of `rows` terms of magnitude <= ~2
            assert_close(f(dw), f(rw), rtol=4 * eps, atol=4 * eps * scale, what=f"dw kind {kind}")
            if db is not None:
                assert_close(f(db), f(rb), rtol=4 * eps, atol=4 * eps * scale, what="db")


def test_backward_is_bitwise_reproducible_and_strided_rows():
    """dw / db are folded in a fixed order (no atomics): two runs agree bit for bit. Rows with a leading dimension > cols (a
    column slice of a wider matrix) leave the bytes between rows untouched."""
    rng = np.random.default_rng(150)
    rows, cols, ld = 300, 1024, 1536
    big = O.f32_to_bf16(rng.uniform(-2, 2, (rows, ld)).astype(np.float32))
    x = np.ascontiguousarray(big[:, :cols])
    w = O.f32_to_bf16(rng.uniform(0.5, 1.5, (cols,)).astype(np.float32))
    go = O.f32_to_bf16(rng.uniform(-1, 1, (rows, cols)).astype(np.float32))
    y, mean, rstd = run_fwd(H.NORM_RMS, x, w, None, H.BF16)
    a1 = run_bwd(H.NORM_RMS, x, w, mean, rstd, go, H.BF16, False)
    a2 = run_bwd(H.NORM_RMS, x, w, mean, rstd, go, H.BF16, False)
    assert np.array_equal(a1[0], a2[0]) and np.array_equal(a1[1], a2[1])
    bx, bw = H.DevBuf.from_numpy(big), H.DevBuf.from_numpy(w)
    out = H.DevBuf.from_numpy(np.full((rows, ld), 0x4242, dtype=np.uint16))
    H.norm_fwd(H.NORM_RMS, H.BF16, rows, cols, bx.ptr, bw.ptr, None, 1e-5, out.ptr, None, None, ld=ld)
    H.device_sync()
    got = out.to_numpy((rows, ld), np.uint16)
    assert np.array_equal(got[:, :cols], y) and (got[:, cols:] == 0x4242).all()


def test_errors():
    a = H.DevBuf(4096)
    with pytest.raises(H.KfError) as e:
        H.norm_fwd(H.NORM_RMS, H.I32, 4, 8, a.ptr, None, None, 1e-5, a.ptr)
    assert e.value.code == H.KF_ERR_UNSUPPORTED
    with pytest.raises(H.KfError) as e:
        H.norm_fwd(H.NORM_RMS, H.F32, 4, 8, a.ptr, None, a.ptr, 1e-5, a.ptr)  # rms takes no bias
    assert e.value.code == H.KF_ERR_INVALID
    with pytest.raises(H.KfError) as e:
        H.norm_fwd(7, H.F32, 4, 8, a.ptr, None, None, 1e-5, a.ptr)
    assert e.value.code == H.KF_ERR_INVALID


@pytest.mark.parametrize("dt,cols", [(np.float32, 128), (np.uint16, 4096), (np.int64, 3), (np.uint8, 7), (np.float64, 1), (np.uint16, 33)])
def test_index_get_bit_exact(dt, cols):
    """Embedding gather: every unit width (16 / 8 / 4 / 2 / 1 bytes by alignment), negative indices, repeated rows."""
    rng = np.random.default_rng(160 + cols)
    nrows, n = 1000, 5000
    table = rng.integers(0, 250, size=(nrows, cols)).astype(dt)
    idx = rng.integers(-nrows, nrows, size=(n,)).astype(np.int64)
    bt, bi, out = H.DevBuf.from_numpy(table), H.DevBuf.from_numpy(idx), H.DevBuf(n * cols * table.itemsize)
    H.index_get(bt.ptr, nrows, cols * table.itemsize, bi.ptr, n, out.ptr)
    H.device_sync()
    got = out.to_numpy((n, cols), dt)
    assert np.array_equal(got, O.index_get(table, idx)) and np.array_equal(got, table[idx])


@pytest.mark.parametrize("cols", [200, 100, 4104])  # 16-byte packs (one block of columns), element-wise rows, packs in two column blocks
@pytest.mark.parametrize("code", [H.F32, H.BF16, H.F16])
def test_index_add_is_the_gathers_backward(code, cols):
    """dTable[r] = sum of the gradient rows whose index is r, in input order, f32 accumulation; heavy duplicates, negative
    indices naming the same rows as positive ones, rows nobody names stay untouched; two runs bit-identical."""
    rng = np.random.default_rng(170 + code)
    nrows, n = 300, 4000 if cols < 1000 else 1500
    idx = rng.integers(-nrows, nrows, size=(n,)).astype(np.int64)
    idx[:500] = 7  # one very popular row
    idx[idx % nrows == 11] = 12  # row 11 is named by nobody
    src = O.from_float(rng.uniform(-1, 1, (n, cols)).astype(np.float32), code)
    bi, bs = H.DevBuf.from_numpy(idx), H.DevBuf.from_numpy(src)
    sentinel = O.from_float(np.full((nrows, cols), 5.0, dtype=np.float32), code)
    outs = []
    for _ in range(2):
        dst = H.DevBuf.from_numpy(sentinel)
        ws = H.index_add(code, bi.ptr, n, bs.ptr, cols, nrows, dst.ptr)
        H.device_sync()
        del ws
        outs.append(dst.to_numpy((nrows, cols), src.dtype))
    assert np.array_equal(outs[0], outs[1])
    want = np.full((nrows, cols), 5.0, dtype=np.float32)
    srcf = O.to_float(src, code)
    wrapped = np.where(idx < 0, idx + nrows, idx)
    for r in np.unique(wrapped):
        acc = np.zeros(cols, dtype=np.float32)
        for j in np.nonzero(wrapped == r)[0]:  # input order, f32 adds: what the kernel does
            acc = acc + srcf[j]
        want[r] = acc
    assert np.array_equal(O.to_float(outs[0], code), O.to_float(O.from_float(want, code), code))
    assert (O.to_float(outs[0], code)[11] == 5.0).all()


@pytest.mark.parametrize("nrows,n,cols", [(300, 4000, 200), (256, 4000, 100), (70000, 20000, 64)])  # nrows a power of two: the sentinel key needs one more bit
def test_index_add_drops_out_of_range_indices(nrows, n, cols):
    """an index outside [-nrows, nrows) used to be truncated to a valid-looking row (and broke the radix sort's promise).
    Now it names no row: its gradient row is dropped, every in-range sum is what it is without those entries, nothing is written
    outside dst (a guard band behind dst stays untouched)."""
    rng = np.random.default_rng(190 + cols)
    idx = rng.integers(-nrows, nrows, size=(n,)).astype(np.int64)
    bad = rng.choice(n, size=n // 10, replace=False)
    idx[bad] = rng.choice(np.array([nrows, nrows + 5, -nrows - 1, 2 ** 31 + 3, -2 ** 40, 2 ** 62], dtype=np.int64), size=bad.size)
    src = rng.uniform(-1, 1, (n, cols)).astype(np.float32)
    bi, bs = H.DevBuf.from_numpy(idx), H.DevBuf.from_numpy(src)
    guard = 64
    dst = H.DevBuf.from_numpy(np.full((nrows + guard, cols), 5.0, dtype=np.float32))
    ws = H.index_add(H.F32, bi.ptr, n, bs.ptr, cols, nrows, dst.ptr)
    H.device_sync()
    del ws
    got = dst.to_numpy((nrows + guard, cols), np.float32)
    want = np.full((nrows + guard, cols), 5.0, dtype=np.float32)
    ok = (idx >= -nrows) & (idx < nrows)
    wrapped = np.where(idx < 0, idx + nrows, idx)
    for r in np.unique(wrapped[ok]):
        acc = np.zeros(cols, dtype=np.float32)
        for j in np.nonzero(ok & (wrapped == r))[0]:
            acc = acc + src[j]
        want[r] = acc
    assert np.array_equal(got, want)


def test_index_add_long_index_list_takes_the_global_sort_with_skipped_passes():
    """20000 indices (beyond one block's radix sort) into 70000 rows: row numbers need 17 bits, so the sort behind the add runs three of the
    four radix passes of its 32-bit keys (round 5) - same sums, in input order, bit for bit."""
    rng = np.random.default_rng(181)
    nrows, n, cols = 70000, 20000, 64
    idx = rng.integers(-nrows, nrows, size=(n,)).astype(np.int64)
    idx[::7] = 69999
    src = rng.uniform(-1, 1, (n, cols)).astype(np.float32)
    bi, bs = H.DevBuf.from_numpy(idx), H.DevBuf.from_numpy(src)
    dst = H.DevBuf.from_numpy(np.full((nrows, cols), 5.0, dtype=np.float32))
    ws = H.index_add(H.F32, bi.ptr, n, bs.ptr, cols, nrows, dst.ptr)
    H.device_sync()
    del ws
    got = dst.to_numpy((nrows, cols), np.float32)
    want = np.full((nrows, cols), 5.0, dtype=np.float32)
    wrapped = np.where(idx < 0, idx + nrows, idx)
    order = np.argsort(wrapped, kind="stable")
    starts = np.flatnonzero(np.r_[True, wrapped[order][1:] != wrapped[order][:-1]])
    ends = np.r_[starts[1:], n]
    for s0, e0 in zip(starts, ends):
        acc = np.zeros(cols, dtype=np.float32)
        for j in order[s0:e0]:
            acc = acc + src[j]
        want[wrapped[order[s0]]] = acc
    assert np.array_equal(got, want)
