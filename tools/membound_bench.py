#!/usr/bin/env python3
"""Achieved HBM bandwidth of the memory-bound half of the hot path (elementwise, copy/permute, convert, fill,
reductions, index_put_) through the C ABI, per-kernel HIP-event times from the library's profiling mode.
Algorithmic bytes per element follow SURVEY.md §8(d): binary 3*sizeof(T), copy 2*sizeof(T), reduce sizeof(T) (+ output).
Prints one JSON object; `--json` path saves it (profiles/)."""
import argparse
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402

PEAK = 8000.0  # GB/s, HBM3E spec (MI355X_MICROARCH.md); 6.3 TB/s is the measured float4-copy ceiling


MARK = {"on": False, "buf": None, "order": []}


def timed(name, fn, rounds):
    if MARK["on"]:  # a one-workgroup f64 fill in front of every case: tools/membound_prof.py cuts the rocprofv3 dispatch list at these
        if MARK["buf"] is None:
            MARK["buf"] = H.DevBuf(512)
        H.elementwise(H.EW_FILL, H.make_desc([H.View(MARK["buf"].ptr, (64,), (1,), H.F64)], []), 0, float(len(MARK["order"])))
        MARK["order"].append(name)
    fn()
    H.device_sync()
    H.profile_reset()
    H.profile_enable(True)
    for _ in range(rounds):
        fn()
    H.device_sync()
    H.profile_enable(False)
    res = H.profile_results()
    ms = sum(v[0] for v in res.values()) / rounds
    return ms, list(res)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--rounds", type=int, default=10)
    ap.add_argument("--json", default="")
    ap.add_argument("--markers", action="store_true", help="a one-workgroup f64 fill before every case (for tools/membound_prof.py)")
    args = ap.parse_args()
    MARK["on"] = args.markers
    H.set_device(0)
    out = {}

    def view(buf, shape, code, strides=None):
        if strides is None:
            strides, run = [], 1
            for s in reversed(shape):
                strides.append(run)
                run *= s
            strides = list(reversed(strides))
        return H.View(buf.ptr, shape, strides, code)

    def record(name, ms, nbytes, kernels):
        gbs = nbytes / (ms * 1e-3) / 1e9
        out[name] = {"case_index": len(out), "rounds": args.rounds, "ms": ms, "algorithmic_bytes": nbytes, "GB/s": gbs, "frac_of_8TB/s": gbs / PEAK, "kernels": kernels}
        print(f"{name:44s} {ms:8.4f} ms  {gbs:9.1f} GB/s  {gbs / PEAK * 100:5.1f}% of HBM peak  {kernels}", flush=True)

    big = 1 << 28  # 256 Mi elements
    a, b, c = (H.DevBuf(4 * big) for _ in range(3))
    for buf, val in ((a, 1.5), (b, -0.25)):
        H.elementwise(H.EW_FILL, H.make_desc([view(buf, (big,), H.F32)], []), 0, val)
    H.device_sync()

    # BASELINE config C1: fp32 add 1024 x 1024 (launch-latency regime) and the same op at HBM-resident size
    for n, tag in ((1024 * 1024, "C1 add f32 1024x1024"), (big, "add f32 256Mi")):
        va, vb, vc = (view(x, (n,), H.F32) for x in (a, b, c))
        d = H.make_desc([vc], [va, vb])
        ms, k = timed(tag, lambda: H.elementwise(H.EW_ADD, d, H.F32), args.rounds)
        record(tag, ms, 12 * n, k)
    n16 = 2 * big
    va, vb, vc = (view(x, (n16,), H.BF16) for x in (a, b, c))
    d = H.make_desc([vc], [va, vb])
    ms, k = timed("add bf16", lambda: H.elementwise(H.EW_ADD, d, H.BF16), args.rounds)
    record("add bf16 512Mi (16-B lanes; scalar in the reference)", ms, 6 * n16, k)
    # broadcast add [N,C] + [1,C] (residual / bias shape)
    N_, C_ = 1 << 16, 4096
    d = H.make_desc([view(c, (N_, C_), H.F32)], [view(a, (N_, C_), H.F32), view(b, (1, C_), H.F32)])
    ms, k = timed("bcast", lambda: H.elementwise(H.EW_ADD, d, H.F32), args.rounds)
    record("add f32 [65536,4096] + [1,4096]", ms, 8 * N_ * C_, k)
    # copy, permute (transpose), convert, fill
    d = H.make_desc([view(c, (big,), H.F32)], [view(a, (big,), H.F32)])
    ms, k = timed("copy", lambda: H.elementwise(H.EW_COPY, d), args.rounds)
    record("copy f32 256Mi", ms, 8 * big, k)
    R = 16384
    d = H.make_desc([view(c, (R, R), H.F32)], [view(a, (R, R), H.F32, [1, R])])
    ms, k = timed("permute", lambda: H.elementwise(H.EW_COPY, d), args.rounds)
    record("permute(1,0).contiguous() f32 16384^2", ms, 8 * R * R, k)
    d = H.make_desc([view(c, (big,), H.BF16)], [view(a, (big,), H.F32)])
    ms, k = timed("convert", lambda: H.elementwise(H.EW_COPY, d), args.rounds)
    record("convert f32 -> bf16 256Mi", ms, 6 * big, k)
    half = big // 2  # mixed dtypes (the runtime-cast kernel; eight elements per lane on contiguous operands since round 5)
    d = H.make_desc([view(c, (half,), H.F32)], [view(a, (half,), H.F32), view(b, (half,), H.BF16)])
    ms, k = timed("add mixed", lambda: H.elementwise(H.EW_ADD, d, H.F32), args.rounds)
    record("add f32 + bf16 -> f32 128Mi (mixed dtypes)", ms, 10 * half, k)
    d = H.make_desc([view(c, (half,), H.F32)], [view(a, (half,), H.I32)])
    ms, k = timed("convert i32", lambda: H.elementwise(H.EW_COPY, d), args.rounds)
    record("convert i32 -> f32 128Mi", ms, 8 * half, k)
    d = H.make_desc([view(c, (big,), H.F32)], [])
    ms, k = timed("fill", lambda: H.elementwise(H.EW_FILL, d, 0, 3.0), args.rounds)
    record("fill f32 256Mi", ms, 4 * big, k)
    # reductions: C1 shapes and HBM-resident ones
    for (rows, cols, dim, tag) in ((1024, 1024, 1, "C1 sum(1) f32 1024x1024"), (1024, 1024, 0, "C1 sum(0) f32 1024x1024"),
                                   (R, R, 1, "sum(1) f32 16384^2"), (R, R, 0, "sum(0) f32 16384^2"), (64, 1 << 22, 1, "sum(1) f32 [64, 4Mi]")):
        shape_o = [rows, cols]
        shape_o[dim] = 1
        vi = view(a, (rows, cols), H.F32)
        vo = view(c, tuple(shape_o), H.F32)
        d = H.make_reduce_desc(vo, vi, dim)
        keep = []
        ms, k = timed(tag, lambda: keep.append(H.reduce(H.RED_SUM, d)), args.rounds)
        record(tag, ms, 4 * rows * cols + 4 * (rows * cols // (cols if dim == 1 else rows)), k)
    # statistics (mean_var / norm_stat): same traffic as sum, two small outputs
    for (rows, cols, dim, code, tag) in ((R, R, 1, H.F32, "mean_var(1) f32 16384^2"), (R, R, 0, H.F32, "norm_stat(0) f32 16384^2"),
                                         (1 << 16, 8192, 1, H.BF16, "row statistics bf16 [65536, 8192] (f32 outputs)")):
        shape_o = [rows, cols]
        shape_o[dim] = 1
        es = 2 if code == H.BF16 else 4
        vi = view(a, (rows, cols), code)
        nout = rows * cols // shape_o[1 - dim] if False else (rows if dim == 1 else cols)
        o0 = H.View(c.ptr, tuple(shape_o), (shape_o[1], 1), H.F32)
        o1 = H.View(c.ptr + 4 * nout, tuple(shape_o), (shape_o[1], 1), H.F32)
        d = H.make_moments_desc(o0, o1, vi, dim)
        keep = []
        ms, k = timed(tag, lambda: keep.append(H.reduce_moments(H.MOM_INVSTD if dim == 0 else H.MOM_VAR, d, 1.0, 1e-12)), args.rounds)
        record(tag, ms, es * rows * cols + 8 * nout, k)
    # rms_norm / layer_norm rows: forward reads x, writes y; backward reads x and dy, writes dx (+ dw, db)
    for (rows, cols, code, tag) in ((1 << 16, 8192, H.BF16, "bf16 [65536, 8192]"), (1 << 16, 4096, H.F32, "f32 [65536, 4096]"),
                                    (1 << 18, 1024, H.BF16, "bf16 [262144, 1024] (one wave per row)")):
        es = 2 if code == H.BF16 else 4
        wbuf, mean, rstd, dwb, dbb = H.DevBuf(cols * es), H.DevBuf(4 * rows), H.DevBuf(4 * rows), H.DevBuf(cols * es), H.DevBuf(cols * es)
        H.elementwise(H.EW_FILL, H.make_desc([view(wbuf, (cols,), code)], []), 0, 1.0)
        for kind, kname in ((H.NORM_RMS, "rms_norm"), (H.NORM_LAYER, "layer_norm")):
            ms, k = timed(kname, lambda: H.norm_fwd(kind, code, rows, cols, a.ptr, wbuf.ptr, None, 1e-5, c.ptr, mean.ptr, rstd.ptr), args.rounds)
            record(f"{kname} fwd {tag}", ms, 2 * es * rows * cols, k)
            keep = []
            ms, k = timed(kname, lambda: keep.append(H.norm_bwd(kind, code, rows, cols, a.ptr, wbuf.ptr, mean.ptr, rstd.ptr, b.ptr, c.ptr, dwb.ptr,
                                                                 dbb.ptr if kind == H.NORM_LAYER else None)), args.rounds)
            record(f"{kname} bwd {tag}", ms, 3 * es * rows * cols, k)
    # embedding gather: 1 Mi rows of 4096 bf16 (8 KiB) out of a 64 Ki-row table
    nidx, ecols = 1 << 17, 4096
    eidx = H.DevBuf.from_numpy(np.random.default_rng(1).integers(0, 1 << 16, size=nidx).astype(np.int64))
    ms, k = timed("gather", lambda: H.index_get(a.ptr, 1 << 16, ecols * 2, eidx.ptr, nidx, c.ptr), args.rounds)
    record("embedding gather bf16 128Ki rows x 4096 (table 512 MiB)", ms, 2 * nidx * ecols * 2, k)
    # index_put_: 16 Mi scattered 4-byte values into a 256 Mi-element tensor (2 int64 indices each)
    m = 1 << 24
    rng = np.random.default_rng(0)
    flat = rng.permutation(1 << 24).astype(np.int64)
    i0 = H.DevBuf.from_numpy(flat // 4096 * 16 % 65536)
    i1 = H.DevBuf.from_numpy(flat % 4096)
    sv = H.View(c.ptr, (m,), (0,), H.F32)
    d = H.make_desc([sv], [view(a, (m,), H.F32), view(i0, (m,), H.I64), view(i1, (m,), H.I64)])
    ms, k = timed("index_put", lambda: H.index_put(d, [65536, 4096], [4096 * 4, 4]), args.rounds)
    record("index_put_ 16Mi x f32 (2 int64 indices)", ms, m * (16 + 8), k)
    # stable sort with int64 positions: algorithmic bytes = keys in + keys out + 8-byte positions out (16 B per f32 key);
    # the radix path really moves ~4 passes x (2 key reads + key write + position read + write)
    keys = rng.standard_normal(1 << 26).astype(np.float32)
    H.check(H.lib().kf_memcpy_h2d(a.ptr, keys.ctypes.data, keys.nbytes, None))
    H.device_sync()
    for (nseg, n, tag) in ((4, 1024000, "sort f32 [4, 1024000] (reference's large case)"), (1, 1 << 26, "sort f32 [1, 64Mi]"),
                           (16384, 4096, "sort f32 [16384, 4096] (radix passes in LDS)"), (1 << 17, 512, "sort f32 [128Ki, 512] (register bitonic, 8 slots per lane)"), (1 << 20, 64, "sort f32 [1Mi, 64] (register bitonic)")):
        need = H.lib().kf_sort_workspace_bytes(H.F32, nseg, n)
        ws = H.DevBuf(max(need, 16))
        ms, k = timed(tag, lambda: H.check(H.lib().kf_sort(H.F32, a.ptr, b.ptr, c.ptr, nseg, n, 0, ws.ptr, need, None)), args.rounds)
        record(tag, ms, 16 * nseg * n, k)
        out[tag]["Mkeys/s"] = nseg * n / (ms * 1e-3) / 1e6
        print(f"{'':44s} {out[tag]['Mkeys/s']:.0f} Mkeys/s")
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
