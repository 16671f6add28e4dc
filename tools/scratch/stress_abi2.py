#!/usr/bin/env python3
"""Second randomised stress at the C ABI (round 6): what stress_abi.py does not draw - rms / layer norm forward + backward on random row counts, row
lengths and leading dimensions (guard bytes between and behind the rows), the embedding gather / scatter-add with wild index lists, kf_gemm_ex with random
tails (bias / mul / add / aux, wide and tight leading dimensions, transposed operands, ragged extents, c_f32). Bounds: the GPU suite's own
(tests/test_gpu_norm.py, tests/test_gpu_gemm.py). stress_abi2.py SEED SECONDS"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
H.set_device(0)
seed, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed)
t_end = time.time() + secs
n_norm = n_idx = n_gex = 0
EPS = {H.BF16: 2.0 ** -7, H.F16: 2.0 ** -10, H.F32: 1e-5}
def close(got, want, rtol, atol, what):
    got, want = np.asarray(got, np.float64), np.asarray(want, np.float64)
    bad = ~(np.abs(got - want) <= atol + rtol * np.abs(want))
    assert not bad.any(), (what, int(bad.sum()), float(np.abs(got - want).max()))
def f64(a, code): return O.to_float(a, code).astype(np.float64)
while time.time() < t_end:
    kind = rng.integers(0, 3)
    if kind == 0:   # norms
        code = int(rng.choice([H.BF16, H.F16, H.F32]))
        eps = EPS[code]
        rows = int(rng.choice([1, 2, 3, 7, 33, 64, 100, 257, 1000, 3000]))
        cols = int(rng.choice([1, 2, 3, 7, 8, 15, 16, 17, 31, 33, 64, 100, 128, 255, 256, 257, 500, 1000, 1024, 1536, 2047, 2048, 4096, 5000, 8192, 12288, 16384, 20000]))
        if rows * cols > 6_000_000: rows = max(1, 6_000_000 // cols)
        ld = cols + int(rng.choice([0, 0, 1, 8, 24, 100]))
        nk = int(rng.choice([H.NORM_RMS, H.NORM_LAYER]))
        has_w = bool(rng.integers(0, 2))
        has_b = has_w and nk == H.NORM_LAYER and bool(rng.integers(0, 2))
        big = O.from_float(rng.uniform(-3, 3, (rows, ld)).astype(np.float32), code)
        gbig = O.from_float(rng.uniform(-1, 1, (rows, ld)).astype(np.float32), code)
        x, go = np.ascontiguousarray(big[:, :cols]), np.ascontiguousarray(gbig[:, :cols])
        w = O.from_float(rng.uniform(0.5, 1.5, (cols,)).astype(np.float32), code) if has_w else None
        b = O.from_float(rng.uniform(-1, 1, (cols,)).astype(np.float32), code) if has_b else None
        fill = 0x4242 if big.itemsize == 2 else 0x42424242
        ut = np.uint16 if big.itemsize == 2 else np.uint32
        bx, bg = H.DevBuf.from_numpy(big), H.DevBuf.from_numpy(gbig)
        bw = H.DevBuf.from_numpy(w) if has_w else None
        bb = H.DevBuf.from_numpy(b) if has_b else None
        by, bdx = (H.DevBuf.from_numpy(np.full((rows + 1, ld), fill, dtype=ut)) for _ in range(2))
        bmean, brstd = H.DevBuf.from_numpy(np.full(rows + 8, 7.0, np.float32)), H.DevBuf.from_numpy(np.full(rows + 8, 7.0, np.float32))
        bdw, bdb = (H.DevBuf.from_numpy(np.full(cols + 8, fill, dtype=ut)) for _ in range(2))
        H.norm_fwd(nk, code, rows, cols, bx.ptr, bw.ptr if bw else None, bb.ptr if bb else None, 1e-5, by.ptr, bmean.ptr, brstd.ptr, ld=ld)
        ws = H.norm_bwd(nk, code, rows, cols, bx.ptr, bw.ptr if bw else None, bmean.ptr, brstd.ptr, bg.ptr, bdx.ptr, bdw.ptr, bdb.ptr if nk == H.NORM_LAYER else None, ld=ld)
        H.device_sync()
        del ws
        tag = ("norm", nk, code, rows, cols, ld, has_w, has_b)
        yy, dxx = by.to_numpy((rows + 1, ld), ut), bdx.to_numpy((rows + 1, ld), ut)
        assert (yy[:rows, cols:] == fill).all() and (yy[rows] == fill).all(), tag + ("y guard",)
        assert (dxx[:rows, cols:] == fill).all() and (dxx[rows] == fill).all(), tag + ("dx guard",)
        mean, rstd = bmean.to_numpy((rows + 8,), np.float32), brstd.to_numpy((rows + 8,), np.float32)
        assert (mean[rows:] == 7.0).all() and (rstd[rows:] == 7.0).all(), tag + ("stat guard",)
        dw_raw, db_raw = bdw.to_numpy((cols + 8,), ut), bdb.to_numpy((cols + 8,), ut)
        assert (dw_raw[cols:] == fill).all() and (db_raw[cols:] == fill).all(), tag + ("dw guard",)
        y = np.ascontiguousarray(yy[:rows, :cols]).view(big.dtype)
        dx = np.ascontiguousarray(dxx[:rows, :cols]).view(big.dtype)
        y_ref, mean_ref, rstd_ref = O.norm_fwd(nk, x, w, b, code=code)
        close(rstd[:rows], rstd_ref, 1e-5, 1e-7, tag + ("rstd",))
        if nk == H.NORM_LAYER: close(mean[:rows], mean_ref, 1e-5, 1e-6, tag + ("mean",))
        close(f64(y, code), f64(y_ref, code), 2 * eps, 2 * eps * max(1.0, float(np.max(rstd_ref)) / 8), tag + ("y",))
        rx, rw, rb = O.norm_bwd(nk, x, w, go, code=code)
        # (a row of two or three nearly equal values has a huge 1 / sigma: xhat and dx amplify the f32 rounding of x - mean by it; the bound follows)
        amp = max(1.0, float(np.max(rstd_ref)))
        close(f64(dx, code), f64(rx, code), 4 * eps, 4 * eps * amp, tag + ("dx",))
        sc = np.sqrt(rows) * amp
        close(f64(dw_raw[:cols].view(big.dtype), code), f64(rw, code), 4 * eps, 4 * eps * sc, tag + ("dw",))
        if nk == H.NORM_LAYER: close(f64(db_raw[:cols].view(big.dtype), code), f64(rb, code), 4 * eps, 4 * eps * sc, tag + ("db",))
        n_norm += 1
    elif kind == 1:  # gather / scatter-add
        nrows = int(rng.choice([1, 2, 17, 256, 300, 1000, 65536, 70000]))
        n = int(rng.choice([1, 2, 63, 64, 65, 1000, 4096, 5000, 20000]))
        dt = [np.float32, np.uint16, np.int64, np.uint8, np.float64][int(rng.integers(0, 5))]
        cols = int(rng.choice([1, 3, 7, 8, 33, 64, 128, 200, 1000]))
        if nrows * cols > 4_000_000: cols = max(1, 4_000_000 // nrows)
        table = rng.integers(0, 250, size=(nrows, cols)).astype(dt)
        idx = rng.integers(-nrows, nrows, size=(n,)).astype(np.int64)
        if rng.integers(0, 2): idx[: n // 2] = idx[0]
        bt, bi = H.DevBuf.from_numpy(table), H.DevBuf.from_numpy(idx)
        out = H.DevBuf.from_numpy(np.full(n * cols * table.itemsize + 64, 0xAB, np.uint8))
        H.index_get(bt.ptr, nrows, cols * table.itemsize, bi.ptr, n, out.ptr)
        H.device_sync()
        raw = out.to_numpy((n * cols * table.itemsize + 64,), np.uint8)
        assert (raw[-64:] == 0xAB).all(), ("get guard", nrows, n, cols, dt)
        assert np.array_equal(raw[:-64].view(dt).reshape(n, cols), table[idx]), ("get", nrows, n, cols, dt)
        code = int(rng.choice([H.F32, H.BF16, H.F16]))
        n2 = min(n, 3000)
        idx2 = idx[:n2].copy()
        bad = rng.choice(n2, size=n2 // 8, replace=False)
        idx2[bad] = rng.choice(np.array([nrows, -nrows - 1, 2 ** 31 + 3, -2 ** 40, 2 ** 62], dtype=np.int64), size=bad.size)
        src = O.from_float(rng.uniform(-1, 1, (n2, cols)).astype(np.float32), code)
        bi2, bs = H.DevBuf.from_numpy(idx2), H.DevBuf.from_numpy(src)
        sentinel = O.from_float(np.full((nrows + 4, cols), 5.0, dtype=np.float32), code)
        dst = H.DevBuf.from_numpy(sentinel)
        ws = H.index_add(code, bi2.ptr, n2, bs.ptr, cols, nrows, dst.ptr)
        H.device_sync()
        del ws
        got = O.to_float(dst.to_numpy((nrows + 4, cols), src.dtype), code)
        want = np.full((nrows + 4, cols), 5.0, dtype=np.float32)
        ok = (idx2 >= -nrows) & (idx2 < nrows)
        wrapped = np.where(idx2 < 0, idx2 + nrows, idx2)
        srcf = O.to_float(src, code)
        order = np.argsort(wrapped[ok], kind="stable")
        rows_ok, src_ok = wrapped[ok][order], srcf[ok][order]
        starts = np.flatnonzero(np.r_[True, rows_ok[1:] != rows_ok[:-1]])
        for s0, s1 in zip(starts, np.r_[starts[1:], len(rows_ok)]):
            acc = np.zeros(cols, dtype=np.float32)
            for j in range(s0, s1): acc = acc + src_ok[j]      # input order, f32 adds
            want[rows_ok[s0]] = acc
        assert np.array_equal(got, O.to_float(O.from_float(want, code), code)), ("add", code, nrows, n2, cols)
        n_idx += 1
    else:           # fused GEMM tails
        code = int(rng.choice([H.BF16, H.F16, H.F32]))
        eps = {H.BF16: 2.0 ** -8, H.F16: 2.0 ** -11, H.F32: 1e-6}[code]
        M, N, K = (int(rng.choice([1, 8, 33, 64, 128, 200, 256, 300, 512, 640, 1000])) for _ in range(3))
        ta, tb = bool(rng.integers(0, 2)), bool(rng.integers(0, 2))
        mk = lambda shp: O.from_float(rng.uniform(-1, 1, shp).astype(np.float32), code)  # noqa: E731
        ldm, lda_, ldx = N + int(rng.choice([0, 8, 24])), N + int(rng.choice([0, 8])), N + int(rng.choice([0, 16]))
        a, b, c, bias, mul_w, add = mk((M, K)), mk((K, N)), mk((M, N)), mk((N,)), mk((M, ldm)), mk((M, lda_))
        use = {k: bool(rng.integers(0, 2)) for k in ("bias", "mul", "add", "aux")}
        alpha, beta = float(rng.choice([1.0, 0.5, -2.0])), float(rng.choice([0.0, 0.0, 1.0, 2.0]))
        sa, sb = (np.ascontiguousarray(a.T) if ta else a), (np.ascontiguousarray(b.T) if tb else b)
        da, db, dbias, dmul, dadd = (H.DevBuf.from_numpy(x) for x in (sa, sb, bias, mul_w, add))
        dc = H.DevBuf.from_numpy(np.concatenate([c.reshape(-1).view(np.uint8), np.full(256, 0xAB, np.uint8)]))
        daux = H.DevBuf.from_numpy(np.full(M * ldx * c.itemsize + 256, 0xAB, np.uint8))
        H.gemm_ex(code, ta, tb, M, N, K, alpha, da.ptr, sa.shape[1], db.ptr, sb.shape[1], beta, dc.ptr, N, bias=dbias.ptr if use["bias"] else None,
                  mul=dmul.ptr if use["mul"] else None, ldmul=ldm, add=dadd.ptr if use["add"] else None, ldadd=lda_, aux=daux.ptr if use["aux"] else None, ldaux=ldx)
        H.device_sync()
        tag = ("gemm_ex", code, M, N, K, ta, tb, alpha, beta, tuple(use.items()), ldm, lda_, ldx)
        craw = dc.to_numpy((M * N * c.itemsize + 256,), np.uint8)
        assert (craw[-256:] == 0xAB).all(), tag + ("C guard",)
        got = f64(craw[:-256].view(c.dtype).reshape(M, N), code)
        fa, fb = f64(a, code), f64(b, code)
        raw = alpha * (fa @ fb) + beta * f64(c, code) + (f64(bias, code)[None, :] if use["bias"] else 0.0)
        mag = abs(alpha) * (np.abs(fa) @ np.abs(fb)) + abs(beta) * np.abs(f64(c, code)) + 1.0
        want = raw * (f64(mul_w, code)[:, :N] if use["mul"] else 1.0) + (f64(add, code)[:, :N] if use["add"] else 0.0)
        assert (np.abs(got - want) <= 4 * eps * (np.abs(want) + np.abs(raw)) + 2e-6 * mag + 4 * eps).all(), tag + ("C", float(np.abs(got - want).max()))
        if use["aux"]:
            xraw = daux.to_numpy((M * ldx * c.itemsize + 256,), np.uint8)
            assert (xraw[-256:] == 0xAB).all(), tag + ("aux guard",)
            xr = xraw[:-256].view(c.dtype).reshape(M, ldx)
            assert (xr[:, N:].view(np.uint8) == 0xAB).all(), tag + ("aux row gap",)
            assert (np.abs(f64(np.ascontiguousarray(xr[:, :N]), code) - raw) <= 2 * eps * np.abs(raw) + 2e-6 * mag + 2 * eps).all(), tag + ("aux",)
        n_gex += 1
print(f"seed {seed}: {n_norm} norms (fwd + bwd), {n_idx} gathers + scatter-adds, {n_gex} fused GEMMs - all agree, every guard byte intact")
