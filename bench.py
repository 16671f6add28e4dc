#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W [--check]

N > 1 runs one process per GPU. Started by torch.distributed.run (RANK / LOCAL_RANK / WORLD_SIZE in the environment) each
process is one rank; started plainly (`python bench.py --gpus 8`) this process touches no GPU: it starts the N rank
processes itself (children, never an exec), relays rank 0's JSON line and exits with the worst child status.

One "step" = one pass of the hot path over one batch of synthetic bf16 tensors, everything through the C ABI
(include/kfunca_hip.h) with inputs already resident in HBM:
    GEMM 4096^3 forward + backward   C = A W,  {dA = dC W^T, dW = A^T dC} as one grouped launch     (BASELINE target GEMM)
    causal attention forward + backward, B=8 H=32 S=4096 D=128               (BASELINE configs[2] / C3)
At N > 1 the batch is sharded (weak scaling: every rank runs the per-GPU batch above) and the weight gradient dW is
sum-all-reduced over RCCL/xGMI (kfunca_amd.parallel.ProcessGroup) on a second stream, overlapped with the attention pass.

Rank 0 prints ONE JSON line: tokens/s over the whole job for the K timed steps, the same over a >= 2 s sustained loop
(the chip needs ~2 s of back-to-back work to settle its clock), per-kernel HIP-event durations, the all-reduce time and bus
bandwidth, the `roofline` of the dominant kernel and a `cpu_baseline` (the CPU oracle timed on a bounded sample).
--check (and always at N = 1): after timing, outputs of the timed buffers are compared with the oracle; with a
communicator the all-reduced dW is compared with the sum of the ranks' own dW taken over gloo on the host.
"""
import argparse
import ctypes as C
import hashlib
import json
import math
import os

# the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle: invalid argument).
# Must be in the environment before HIP initialises; children inherit it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import subprocess
import sys
import threading
import time
from contextlib import contextmanager
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

GEMM_N = 4096
AB, AH, AS, AD = 8, 32, 4096, 128
PEAK_MFMA_BF16 = 2500.0  # TFLOP/s dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM = 8000.0        # GB/s spec, same guide
XGMI_PEAK = 7 * 153.0    # GB/s per GPU, all links

# algorithmic FLOPs per launch: causal attention counts S^2/2 score entries
_PAIR = AB * AH * AS * AS * AD / 2.0
KERNEL_FLOPS = {
    "gemm_bf16_mfma": 2.0 * GEMM_N ** 3,
    "gemm_bf16_mfma_pair": 4.0 * GEMM_N ** 3,  # dA = dC W^T and dW = A^T dC in one grid (kf_gemm_grouped)
    "attn_fwd_mfma": 4.0 * _PAIR,       # QK^T + PV
    "attn_bwd_dkv_mfma": 8.0 * _PAIR,   # S, dP, dV, dK
    "attn_bwd_dq_mfma": 2.0 * _PAIR,    # dQ = dS K
}
GEMM_FLOPS_STEP = 6.0 * GEMM_N ** 3
ATTN_FLOPS_STEP = 14.0 * _PAIR
TOKENS_STEP = AB * AS


RANK_TIMEOUT_S = float(os.environ.get("KF_BENCH_RANK_TIMEOUT_S", "600"))   # the parent's patience with its rank processes
PHASE_TIMEOUT_S = float(os.environ.get("KF_BENCH_PHASE_TIMEOUT_S", "180"))  # a rank's own patience with one collective phase


@contextmanager
def deadline(what: str, seconds: float = None):
    """Bounds a phase that can wait on OTHER ranks (communicator construction, a barrier, a timed loop with collectives): if it has not
    ended after `seconds`, this process says so on stderr and exits with status 124 - a fresh exit, never a re-exec - so that a rank stuck in
    ncclCommInitRank or in a collective takes the job down (torch.distributed.run and launch_ranks both end the other ranks when one
    fails) instead of holding the GPUs until somebody else's limit fires."""
    seconds = PHASE_TIMEOUT_S if seconds is None else seconds
    done = threading.Event()

    def watch():
        if not done.wait(seconds):
            print(f"[bench.py] rank {os.environ.get('RANK', '0')}: '{what}' did not finish within {seconds:.0f} s - giving up", file=sys.stderr, flush=True)
            os._exit(124)
    t = threading.Thread(target=watch, daemon=True)
    t.start()
    try:
        yield
    finally:
        done.set()


def bf16_random(rng, shape):
    x = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


def device_src_sha() -> str:
    """Content hash of the device sources (kfunca_amd/_build.py owns the definition: the same value is linked into libkfunca_hip.so as
    kf_build_source_sha): profiles/*.json written by tools/pmc_*.py carry it, and a profile whose stamp differs from the tree being
    benchmarked is not quoted (the GPU box has no .git, so a commit id is not available)."""
    from kfunca_amd import _build
    return _build.device_src_sha()


def quoted_traffic(dom: str, lib_sha: str):
    """(bytes per launch, file name) of the newest committed PMC traffic profile of kernel `dom` - quoted ONLY when (a) the profile
    was taken on THESE device sources and (b) the library that just ran was built from them too (lib_sha = kf_build_source_sha(); a stale
    prebuilt .so next to fresh sources would otherwise print fresh-looking hashes: VERDICT round 4, weak #7). Else (None, reason)."""
    if lib_sha != device_src_sha():
        return None, f"refused: libkfunca_hip.so was built from sources {lib_sha}, the tree is {device_src_sha()}"
    for tfile in sorted((ROOT / "profiles").glob("r*_pmc_traffic.json"), reverse=True):
        tj = json.loads(tfile.read_text())
        if stamp_is_current(tj, BENCH_SOURCES):
            return tj.get(dom, {}).get("bytes_per_launch"), tfile.name
    return None, None


# which device sources each kind of committed profile depends on (tests/test_profiles_fresh.py): a profile is STALE - and must be
# re-collected before it is quoted - when any of its files has changed since it was taken
BENCH_SOURCES = ("attention.hip", "attn_fwd_w4.inc", "attn_dkv_w4.inc", "gemm.hip", "common.h", "runtime.hip")
MEMBOUND_SOURCES = ("elementwise.hip", "reduce.hip", "norm.hip", "index.hip", "sort.hip", "common.h", "offset_calc.h", "runtime.hip")


def device_src_shas(names=None) -> dict:
    """Per-file content hashes of the device sources (all of them, or the `names` subset)."""
    out = {}
    for p in sorted((ROOT / "kfunca_amd" / "csrc" / "device").glob("*")):
        if p.suffix in (".hip", ".h", ".inc") and (names is None or p.name in names):
            out[p.name] = hashlib.sha256(p.read_bytes()).hexdigest()[:16]
    return out


def stamp(names) -> dict:
    """What a profile-writing tool puts into its output: the tree-wide hash (as before) and the per-file hashes it depends on."""
    return {"device_src_sha": device_src_sha(), "device_src_files": device_src_shas(names)}


def stamp_is_current(obj: dict, names) -> bool:
    """True when a profile's stamp matches THIS tree on the files it depends on (an old stamp without per-file hashes must match tree-wide)."""
    files = obj.get("device_src_files")
    if files:
        now = device_src_shas(names)
        if all(n in files for n in now):
            return all(files[n] == h for n, h in now.items())
        # a dependency the stamp does not list (the list grew after it was taken): only the tree-wide hash can vouch for it
        return all(files[n] == h for n, h in now.items() if n in files) and obj.get("device_src_sha") == device_src_sha()
    return obj.get("device_src_sha") == device_src_sha()


def grad_f32_default(world: int) -> bool:
    """Which gradient format the dW all-reduce uses. bf16 sums round once per addition: the check's bound N 2^-8 sum|dW_r| is 3 % of sum|dW| at
    N = 8 - so from 8 ranks on the float path (the dW GEMM keeps its f32 accumulators, RCCL adds floats; twice the message bytes) is the
    DEFAULT; below it bf16 is. KF_BENCH_GRAD_F32=1 / KF_BENCH_GRAD_BF16=1 force either."""
    if os.environ.get("KF_BENCH_GRAD_F32"):
        return True
    if os.environ.get("KF_BENCH_GRAD_BF16"):
        return False
    return world >= 8


class Workload:
    def __init__(self, H, rank, world=1):
        self.H = H
        rng = np.random.default_rng(1003 + 7919 * rank)  # seed = 1000 + config number (C3), rank-offset data
        n = GEMM_N
        self.A_host = bf16_random(rng, (n, n))
        self.A = H.DevBuf.from_numpy(self.A_host)
        wrng = np.random.default_rng(1003)                # weights: identical on every rank
        self.W_host = bf16_random(wrng, (n, n))
        self.W = H.DevBuf.from_numpy(self.W_host)
        self.dC = H.DevBuf.from_numpy(bf16_random(rng, (n, n)))
        self.grad_f32 = grad_f32_default(world)  # dW leaves the GEMM as float and is all-reduced as float
        self.Cc, self.dA, self.dW = H.DevBuf(2 * n * n), H.DevBuf(2 * n * n), H.DevBuf((4 if self.grad_f32 else 2) * n * n)
        nbytes = AB * AH * AS * AD * 2
        self.q, self.k, self.v, self.o, self.do = (H.DevBuf(nbytes) for _ in range(5))
        self.dq, self.dk, self.dv = (H.DevBuf(nbytes) for _ in range(3))
        per_b = nbytes // AB
        self.host = {}
        for name, buf in (("q", self.q), ("k", self.k), ("v", self.v), ("do", self.do)):  # one random batch element, replicated on device
            host = bf16_random(rng, (AH, AS, AD))
            self.host[name] = host[0].copy()  # head (0, 0), kept for the after-timing spot check
            H.check(H.lib().kf_memcpy_h2d(buf.ptr, host.ctypes.data, per_b, None))
            for b in range(1, AB):
                H.check(H.lib().kf_memcpy_d2d(buf.ptr + b * per_b, buf.ptr, per_b, None))
        self.lse = H.DevBuf(4 * AB * AH * AS)
        self.aws_bytes = H.attn_bwd_workspace_bytes(H.BF16, AB, AH, AS, AS, AD)
        self.aws = H.DevBuf(self.aws_bytes)
        H.device_sync()

    def step(self, stream, pg=None, comm_stream=None, ev_grad=None, ev_comm=None, comm_events=None, no_comm=False):
        H, n, s = self.H, GEMM_N, stream
        if no_comm:  # the same step with the collective left out (the "off" arm of exposed_comm_ms)
            pg = None
        if pg is not None and ev_comm.recorded:  # dW of the previous step must be fully reduced before it is rewritten
            H.stream_wait_event(s, ev_comm)
        H.gemm(H.BF16, 0, 0, n, n, n, 1.0, self.A.ptr, n, self.W.ptr, n, 0.0, self.Cc.ptr, n, 0, None, None, 0, s)
        # the backward pair dA = dC W^T, dW = A^T dC: one grid (the second product starts under the first one's last tiles)
        H.gemm_grouped(H.BF16, [(0, 1, n, n, n, 1.0, 0.0, self.dC.ptr, n, self.W.ptr, n, self.dA.ptr, n),
                                (1, 0, n, n, n, 1.0, 0.0, self.A.ptr, n, self.dC.ptr, n, self.dW.ptr, n, int(self.grad_f32))], s)
        if pg is not None:  # gradient all-reduce on its own stream, overlapped with the attention pass
            ev_grad.record(s)
            H.stream_wait_event(comm_stream, ev_grad)
            if comm_events is not None:
                e0, e1 = H.Event(), H.Event()
                e0.record(comm_stream)
            pg.allreduce_sum_device(self.dW.ptr, n * n, H.F32 if self.grad_f32 else H.BF16, comm_stream)
            if comm_events is not None:
                e1.record(comm_stream)
                comm_events.append((e0, e1))
            ev_comm.record(comm_stream)
            ev_comm.recorded = True
        H.attn_fwd(H.BF16, AB, AH, AS, AS, AD, self.q.ptr, self.k.ptr, self.v.ptr, self.o.ptr, self.lse.ptr, s)
        H.attn_bwd(H.BF16, AB, AH, AS, AS, AD, self.q.ptr, self.k.ptr, self.v.ptr, self.o.ptr, self.lse.ptr, self.do.ptr,
                   self.dq.ptr, self.dk.ptr, self.dv.ptr, self.aws.ptr, self.aws_bytes, s)


CHECK_NOTES = {}


def spot_check(H, wl, check_dw=True):
    """After timing: outputs left in the timed buffers against the oracle (the checker; never inside the timed region).
    GEMM: 8 rows of C = A W and of both products of the backward pair, dA = dC W^T and dW = A^T dC (bf16 inputs, f32 accumulation,
    one rounding). Attention: all of head (0, 0) under the scale-aware
    bounds (see below) and the checksum sum_n dV[n] = sum_m dO[m]."""
    from oracle import oracle as O
    n = GEMM_N
    rows = [0, 1, 127, 128, 2047, 3000, 4094, 4095]
    f64 = lambda x: O.bf16_to_f32(x).astype(np.float64)

    from oracle import checks as K

    def rows_ok(buf, a16, b16, **kw):
        """The suite's GEMM bound (oracle/checks.py gemm_ok: eps |c| + 1e-6 sum |a||b| against the f64 product), on the sampled rows."""
        got = np.empty((len(rows), n), dtype=np.uint16)
        for i, r in enumerate(rows):
            H.check(H.lib().kf_memcpy_d2h(got[i].ctypes.data, buf.ptr + r * n * 2, n * 2, None))
        return K.gemm_ok(got, a16, b16, O.BF16, **kw)[0]

    dC_host = wl.dC.to_numpy((n, n), np.uint16)
    gemm_ok = rows_ok(wl.Cc, wl.A_host[rows], wl.W_host)
    # the backward pair of the same launch (gemm_bf16_mfma_pair): rows of dA = dC W^T and of dW = A^T dC under the same bound
    # (bar: the reference's own GEMM test, test/test_gemm.py:9-17, is a forward-only f64 case; the backward is this repository's)
    da_ok = rows_ok(wl.dA, dC_host[rows], wl.W_host, trans_b=True)
    a_cols = np.ascontiguousarray(wl.A_host[:, rows])
    dw_ok = rows_ok(wl.dW, a_cols, dC_host, trans_a=True) if (check_dw and not wl.grad_f32) else None
    if check_dw and wl.grad_f32:  # the float dW: against the f64 product, f32 accumulation noise only
        got = np.empty((len(rows), n), dtype=np.float32)
        for i, r in enumerate(rows):
            H.check(H.lib().kf_memcpy_d2h(got[i].ctypes.data, wl.dW.ptr + r * n * 4, n * 4, None))
        want, mag = f64(a_cols).T @ f64(dC_host), np.abs(f64(a_cols)).T @ np.abs(f64(dC_host))
        dw_ok = bool((np.abs(got - want) <= 2e-6 * mag + 1e-6).all())
    # attention: EVERY element of head (0, 0) - O, LSE, dQ, dK, dV at S = 4096 - against the double-precision oracle under the
    # scale-aware bounds of oracle/checks.py (per element, per row, per head; no absolute tolerance), and the same head of the last
    # batch element (the batch is one element replicated) bit-identical to it
    q, k, v, go = (wl.host[x][None, None] for x in ("q", "k", "v", "do"))
    per_b = AH * AS * AD * 2

    def head(buf, b, shape=(1, 1, AS, AD), dt=np.uint16, per_batch=per_b):
        out = np.empty(shape, dtype=dt)
        H.check(H.lib().kf_memcpy_d2h(out.ctypes.data, buf.ptr + b * per_batch, out.nbytes, None))
        return out

    got = {n: head(b, 0) for n, b in (("o", wl.o), ("dq", wl.dq), ("dk", wl.dk), ("dv", wl.dv))}
    lse_got = head(wl.lse, 0, (1, 1, AS), np.float32, AH * AS * 4)
    try:
        margins = K.attn_check(q, k, v, O.BF16, o=got["o"], lse=lse_got, d_o=go, dq=got["dq"], dk=got["dk"], dv=got["dv"], what="bench head (0,0)")
        attn_ok, attn_note = True, {n: round(max(m.get("element", 0), m.get("row", 0), m.get("head", 0)), 3) for n, m in margins.items() if n != "lse"}
        attn_note["lse"] = round(margins["lse"]["fraction_of_bound"], 3)   # the forward rounds c q to bf16: lse against that rounding's worst case (oracle/checks.py)
    except AssertionError as e:
        attn_ok, attn_note = False, str(e)
    same = all(np.array_equal(got[n], head(b, AB - 1)) for n, b in (("o", wl.o), ("dq", wl.dq), ("dk", wl.dk), ("dv", wl.dv)))
    dv_sum, do_sum = O.bf16_to_f32(got["dv"][0, 0]).astype(np.float64).sum(0), O.bf16_to_f32(wl.host["do"]).astype(np.float64).sum(0)
    bound = 2.0 ** -8 * (np.abs(O.bf16_to_f32(got["dv"][0, 0]).astype(np.float64)).sum(0) + np.abs(O.bf16_to_f32(wl.host["do"]).astype(np.float64)).sum(0) / math.sqrt(AS))
    bwd_ok = bool((np.abs(dv_sum - do_sum) <= bound).all())
    CHECK_NOTES["attn_head00_worst_fraction_of_bound"] = attn_note  # per output: max of the element / row / head figures (1 = at the bound)
    out = {"gemm_rows_vs_oracle": gemm_ok, "gemm_dA_rows_vs_oracle": da_ok, "attn_head00_vs_oracle_scale_aware": attn_ok,
           "attn_last_batch_bit_identical": bool(same), "attn_dv_checksum": bwd_ok}
    if dw_ok is not None:  # with a communicator dW holds the SUM over ranks: that is check_allreduce's business
        out["gemm_dW_rows_vs_oracle"] = dw_ok
    return out


def check_allreduce(H, wl, pg, stream):
    """Section 8e's parity sentence on the live buffers: the RCCL-reduced dW must equal the sum over ranks of each rank's own
    dW = A_r^T dC_r (recomputed here without the collective, summed in f32 over gloo on the host). RCCL adds bf16 values, so
    each of its N - 1 additions rounds once: |err| <= N 2^-8 sum_r |dW_r| elementwise (N = 1: the all-reduce is the identity
    and the two must be bit-identical)."""
    n = GEMM_N
    if wl.grad_f32:
        # the float gradient path: each rank's dW is its f32 accumulators, RCCL adds floats. Against the sum of the ranks' own float dW
        # taken in f64 on the host: f32 addition noise only - 2^-22 sum_r |dW_r| per addition level, NOTHING that scales like the 16-bit
        # format's 2^-8 (the bound of the 16-bit path below grows with the number of ranks; this one does not in any way that matters)
        reduced = wl.dW.to_numpy((n, n), np.float32)
        local = H.DevBuf(4 * n * n)
        H.gemm_ex(H.BF16, 1, 0, n, n, n, 1.0, wl.A.ptr, n, wl.dC.ptr, n, 0.0, local.ptr, n, stream=stream, c_f32=True)
        H.device_sync()
        mine = local.to_numpy((n, n), np.float32).astype(np.float64)
        total = pg.allreduce_sum_host(mine.copy().reshape(-1)).reshape(n, n)
        mag = pg.allreduce_sum_host(np.abs(mine).reshape(-1).copy()).reshape(n, n)
        if pg.world == 1:
            return bool(np.array_equal(reduced, local.to_numpy((n, n), np.float32)))
        return bool((np.abs(reduced - total) <= 2.0 ** -20 * mag + 1e-6).all())
    reduced = wl.dW.to_numpy((n, n), np.uint16)
    local = H.DevBuf(2 * n * n)
    H.gemm(H.BF16, 1, 0, n, n, n, 1.0, wl.A.ptr, n, wl.dC.ptr, n, 0.0, local.ptr, n, 0, None, None, 0, stream)
    H.device_sync()
    mine = local.to_numpy((n, n), np.uint16)
    u = (mine.astype(np.uint32) << 16).view(np.float32)
    total = pg.allreduce_sum_host(u.copy().reshape(-1)).reshape(n, n)
    mag = pg.allreduce_sum_host(np.abs(u).reshape(-1).copy()).reshape(n, n)
    got = (reduced.astype(np.uint32) << 16).view(np.float32)
    if pg.world == 1:
        return bool(np.array_equal(reduced, mine))
    return bool((np.abs(got - total) <= pg.world * 2.0 ** -8 * mag + 1e-6).all())


def cpu_baseline():
    """The CPU oracle (kind "port": the reference has no CPU path) on a bounded sample of
    the same step: attention fwd+bwd for `heads` of the 256 (b,h) pairs at full S, and `rows` of the 4096
    output rows of each of the three GEMMs; scaled linearly to the whole step."""
    from oracle import oracle as O
    cores = O.num_threads()
    rng = np.random.default_rng(1003)
    heads, rows, n = max(1, min(cores, 128)), 512, GEMM_N  # ~10-20 s of CPU work on 8..256 cores
    q, k, v, go = (bf16_random(rng, (1, heads, AS, AD)) for _ in range(4))
    t0 = time.perf_counter()
    O.attn_fwd(q, k, v, code=O.BF16)
    O.attn_bwd(q, k, v, go, code=O.BF16)
    t_attn = time.perf_counter() - t0
    a, w, g = (bf16_random(rng, (n, n)) for _ in range(3))
    f = O.lib().orc_gemm
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_int64,
                  C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    out = np.empty((rows, n), dtype=np.uint16)
    t0 = time.perf_counter()
    f(O.BF16, 0, 0, rows, n, n, 1.0, a.ctypes.data, n, w.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # C  = A W
    f(O.BF16, 0, 1, rows, n, n, 1.0, g.ctypes.data, n, w.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # dA = dC W^T
    f(O.BF16, 1, 0, rows, n, n, 1.0, a.ctypes.data, n, g.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # dW = A^T dC (rows of dW)
    t_gemm = time.perf_counter() - t0
    t_step = t_attn * (AB * AH / heads) + t_gemm * (n / rows)
    return {"value": TOKENS_STEP / t_step, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"attention fwd+bwd of {heads}/256 (b,h) pairs at S=4096 D=128 ({t_attn:.1f} s) + {rows}/4096 output rows "
                      f"of each of the 3 GEMMs ({t_gemm:.1f} s), scaled linearly to one step"}


def mfma_ceiling(stream, warm_s=0.7, timed_s=0.3):
    """TFLOP/s of a bare back-to-back MFMA loop on random bf16 operands on THIS box, now: {"32x32x16": .., "16x16x32": ..} - what the
    matrix pipes hold under the power cap when nothing else is asked of them. None when the diagnostic library is not there."""
    lib = ROOT / "kfunca_amd" / "_build" / "libkfunca_diag.so"
    if not lib.exists():
        return None
    f = C.CDLL(str(lib)).kf_diag_mfma_ceiling
    f.argtypes = [C.c_int, C.c_int, C.c_double, C.c_double, C.c_void_p, C.POINTER(C.c_double)]
    out = {}
    for shape, name in ((0, "32x32x16"), (1, "16x16x32")):
        t = C.c_double(0.0)
        if f(shape, 0, warm_s, timed_s, stream, C.byref(t)) != 0:
            return None
        out[name] = t.value
    return out


# the MFMA instruction shape of each timed kernel (which ceiling it is priced against)
KERNEL_MFMA_SHAPE = {"gemm_bf16_mfma": "16x16x32", "gemm_bf16_mfma_pair": "16x16x32", "attn_fwd_mfma": "32x32x16",
                     "attn_bwd_dkv_mfma": "32x32x16", "attn_bwd_dq_mfma": "32x32x16"}


def launch_ranks(args) -> int:
    """`python bench.py --gpus N` without a torchrun environment: start the N ranks as CHILD processes (this process has
    not touched, and never touches, a GPU), pass rank 0's output through and return the worst exit status."""
    # rendezvous over a FILE store in a fresh temporary directory (kfunca_amd/parallel.py, KF_RDZV_FILE): round 4 picked a TCP port by
    # bind-then-close, which any other job on the box could take in between
    import tempfile
    with tempfile.TemporaryDirectory(prefix="kf_rdzv_") as tmp:   # removed with its store file whatever happens to the ranks
        rdzv, t_job = Path(tmp) / "store", time.time()
        procs = []
        for r in range(args.gpus):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), KF_RDZV_FILE=str(rdzv), KF_RDZV_T0=repr(t_job))
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
        out0, codes = supervise(procs, RANK_TIMEOUT_S)
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)  # library banners etc.: THE line is the last one
    if lines and all(c == 0 for c in codes):
        print(lines[-1], flush=True)
    return max(abs(c) for c in codes)


def supervise(procs, timeout_s):
    """Waits for the rank processes: returns (rank 0's stdout, exit codes). When one of them fails, or the whole job outlives
    `timeout_s`, the OTHERS are ended (these exact children, by handle) - one rank's hang must not keep the rest, and their GPUs, waiting."""
    out0 = []
    reader = threading.Thread(target=lambda: out0.append(procs[0].stdout.read() if procs[0].stdout else ""), daemon=True)
    reader.start()
    t_end = time.monotonic() + timeout_s
    why = None
    while True:
        codes = [p.poll() for p in procs]
        if all(c is not None for c in codes):
            break
        if any(c not in (None, 0) for c in codes):
            why = f"rank {next(i for i, c in enumerate(codes) if c not in (None, 0))} exited with status {next(c for c in codes if c not in (None, 0))}"
        elif time.monotonic() > t_end:
            why = f"the job did not finish within {timeout_s:.0f} s"
        if why:
            print(f"[bench.py] {why}: ending the remaining ranks", file=sys.stderr, flush=True)
            for p in procs:
                if p.poll() is None:
                    p.kill()
            codes = [p.wait() for p in procs]  # the killed ones report -9: the job's status is non-zero
            break
        time.sleep(0.05)
    reader.join(timeout=5)
    return (out0[0] if out0 else ""), [c if c is not None else 1 for c in codes]


def dry_run_cpu(args, rank, world):
    """The N > 1 plumbing without a GPU (tests/test_parallel_gloo.py): rendezvous, gradient bucket, sum all-reduce over gloo,
    max-over-ranks timing and the one-line report — the same ProcessGroup calls the GPU path makes, host buffers instead.
    KF_BENCH_DRY_FAULT="<rank>:exit" / "<rank>:hang" makes that rank exit with status 3 / never reach the collective: what the parent's
    supervise() and the ranks' deadline() watchdogs are for."""
    from kfunca_amd import parallel
    fault = os.environ.get("KF_BENCH_DRY_FAULT", "")
    with deadline("rendezvous"):
        pg = parallel.ProcessGroup(backend="gloo")
    if fault and int(fault.split(":")[0]) == rank:
        if fault.endswith("exit"):
            os._exit(3)
        time.sleep(10 ** 6)   # "hang": this rank holds everybody's collective until a watchdog ends the job
    bucket = parallel.GradBucket([(64, 64)], dtype=np.float32)
    flat = np.full(bucket.numel, float(rank + 1), dtype=np.float32)
    t0 = time.perf_counter()
    with deadline("dry-run all-reduce"):
        pg.allreduce_sum_host(flat)
        elapsed = pg.max_over_ranks(time.perf_counter() - t0)
    ok = bool((flat == world * (world + 1) / 2).all())
    pg.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "allreduce_check": ok, "elapsed_s": elapsed,
                          "grad_dtype": "f32" if grad_f32_default(world) else "bf16"}), flush=True)
    pg.close()
    return 0 if ok else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--sustain-seconds", type=float, default=2.0, help="length of the sustained loop after the timed steps (0: skip)")
    ap.add_argument("--check", action="store_true", help="after timing, verify outputs (oracle spot checks; all-reduced dW vs the gloo sum)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-ceiling", action="store_true", help="skip the bare-MFMA ceiling probe (roofline.frac_of_capped_mfma)")
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world == 1:
        return launch_ranks(args)  # before anything touches the GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)

    from kfunca_amd import hip_abi as H
    from kfunca_amd import parallel
    if H.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    H.set_device(local_rank)

    pg = None
    force_comm = bool(os.environ.get("KF_BENCH_FORCE_COMM"))  # exercises the collective path on one GPU
    if world > 1 or force_comm:
        if rank == 0:  # RCCL's own warnings of rank 0 go to stderr (stdout carries THE line)
            os.environ.setdefault("NCCL_DEBUG", "WARN")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        with deadline("rendezvous + ncclCommInitRank"):
            pg = parallel.ProcessGroup(backend="rccl")  # gloo rendezvous (plumbing) + RCCL communicator through the C ABI
        C.CDLL(None).fflush(None)  # RCCL leaves its version banner in the C stdout buffer: out with it now

    wl = Workload(H, rank, world)
    stream = H.Stream()
    comm_stream = H.Stream() if pg else None
    ev_grad, ev_comm = (H.Event(), H.Event()) if pg else (None, None)
    if pg:
        ev_comm.recorded = False

    no_comm_env = bool(os.environ.get("KF_BENCH_NO_COMM"))  # A/B by hand: the main loop without the collective

    def barrier():
        with deadline("device sync + barrier"):
            H.device_sync()
            if pg is not None:
                pg.barrier()

    def run_step(comm_events=None, no_comm=no_comm_env):
        wl.step(stream.handle, pg, comm_stream.handle if pg else None, ev_grad, ev_comm, comm_events, no_comm=no_comm)

    def timed(steps, comm_events=None, no_comm=no_comm_env):
        """EXACTLY `steps` steps between two barrier + device-sync brackets; the maximum over ranks."""
        barrier()
        t0 = time.perf_counter()
        with deadline(f"{steps} timed steps", PHASE_TIMEOUT_S + 0.05 * steps):
            for _ in range(steps):
                run_step(comm_events, no_comm)
        barrier()
        dt = time.perf_counter() - t0
        return pg.max_over_ranks(dt) if pg is not None else dt

    for _ in range(args.warmup):
        run_step()
    barrier()
    # attribution at N > 1: the SAME loop with the collective left out, BEFORE the measured one (the no-comm steps
    # leave each rank's local dW behind: whatever runs last must be a step with the collective). The difference to the measured
    # loop is the communication time the step could not hide (exposed); what the all-reduce took on its own stream is `allreduce.ms`.
    # Both arms run with per-launch profiling OFF and interleaved (off, on, off, on): round 4's figure compared a profiled loop with an
    # unprofiled one (0.4 ms of event records, not of communication); `exposed_comm_noise_floor_ms` is the A/B's own repeatability.
    elapsed_off, ab = None, None
    if pg is not None and not no_comm_env:
        ab = [timed(args.steps, None, no_comm=(i % 2 == 0)) / args.steps * 1e3 for i in range(4)]   # off, on, off, on
        elapsed_off = (ab[0] + ab[2]) / 2 * 1e-3 * args.steps
    # THE measured loop runs with per-launch profiling OFF (the event records were 1.6 % of `value`); the per-kernel
    # table comes from a SECOND loop of the same length with profiling on, reported as `ms_per_step_profiled`.
    elapsed = timed(args.steps)
    H.profile_reset()
    H.profile_enable(True)
    comm_events = [] if (pg and not no_comm_env) else None
    elapsed_prof = timed(args.steps, comm_events)
    H.profile_enable(False)
    prof = H.profile_results()
    samples = H.profile_samples()  # every launch's own HIP-event duration: percentiles, run-to-run spread

    # sustained loop: the chip settles its clock only after ~2 s of back-to-back work (profiles/r01_gemm_clock.json)
    sustained = None
    if args.sustain_seconds > 0:
        n_sus = max(args.steps, int(math.ceil(args.sustain_seconds / max(elapsed / args.steps, 1e-6))))
        sustained = (n_sus, timed(n_sus))
    # this box's own matrix-pipe ceiling under its power cap, right behind the sustained loop (the chip is warm): a bare MFMA loop of the
    # dominant kernels' instruction shape on random bf16 register operands (kfunca_amd/csrc/diag/mfma_ceiling.hip -> _build/libkfunca_diag.so;
    # not part of libkfunca_hip.so). ~1 s per shape.
    ceiling = mfma_ceiling(stream.handle) if rank == 0 and not args.no_ceiling else None

    checks = {}
    if pg is not None and (args.check or force_comm):
        # the check reads dW as the LAST step left it: make that a step with the collective, whatever loops ran before
        run_step(None, no_comm=False)
        barrier()
        checks["allreduce_dw_vs_gloo_sum"] = check_allreduce(H, wl, pg, stream.handle)
    rc = 0
    if rank == 0:
        if world == 1 or args.check:
            checks.update(spot_check(H, wl, check_dw=(pg is None or world == 1)))
        ms_step = elapsed / args.steps * 1e3
        kern = {k: {"avg_ms": ms / max(cnt, 1), "launches": cnt} for k, (ms, cnt) in prof.items()}
        for k, v in kern.items():
            if k in samples and len(samples[k]):
                a = np.sort(samples[k].astype(np.float64))
                v.update({"p50_ms": float(np.percentile(a, 50)), "p95_ms": float(np.percentile(a, 95)), "min_ms": float(a[0]), "max_ms": float(a[-1])})
            if k in KERNEL_FLOPS:
                v["tflops"] = KERNEL_FLOPS[k] / (v["avg_ms"] * 1e-3) / 1e12
                if "p50_ms" in v:
                    v["tflops_p50"] = KERNEL_FLOPS[k] / (v["p50_ms"] * 1e-3) / 1e12
        per_step = {k: ms / args.steps for k, (ms, cnt) in prof.items()}
        gemm_ms = sum(v for k, v in per_step.items() if k.startswith("gemm"))
        attn_ms = sum(v for k, v in per_step.items() if k.startswith("attn"))
        dom = max((k for k in per_step if k in KERNEL_FLOPS), key=lambda k: per_step[k])
        achieved = kern[dom]["tflops"]
        # traffic beyond L2 per launch of the dominant kernel: rocprofv3 PMC passes cannot run inside this process; the
        # committed measurement of the same kernel on the same shape is quoted only if it was taken on THESE device sources
        lib_sha = H.build_source_sha()
        traffic, traffic_src = quoted_traffic(dom, lib_sha)
        out = {
            "metric": "bf16 GEMM TFLOP/s + causal-attn fwd+bwd tokens/s",
            "value": world * TOKENS_STEP / (elapsed / args.steps),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "ms_per_step_profiled": elapsed_prof / args.steps * 1e3,   # the second loop, per-launch HIP events on: where `kernels` comes from
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "bf16 GEMM 4096x4096x4096 fwd+bwd + bf16 causal attention fwd+bwd B=8 H=32 S=4096 D=128 per GPU",
                       "global_batch": AB * world, "seq_len": AS, "parallelism": f"dp{world}"},
            "gemm_tflops": GEMM_FLOPS_STEP / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
            "gemm_ms_per_step": gemm_ms,
            "attn_tokens_per_s": TOKENS_STEP / (attn_ms * 1e-3) if attn_ms else None,
            "attn_tflops": ATTN_FLOPS_STEP / (attn_ms * 1e-3) / 1e12 if attn_ms else None,
            "attn_ms_per_step": attn_ms,
            "kernels": kern,
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_MFMA_BF16, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_MFMA_BF16, "traffic": traffic, "traffic_source": traffic_src,
                         "algorithmic_flops_per_launch": KERNEL_FLOPS[dom]},
            "grad_dtype": "f32" if wl.grad_f32 else "bf16",   # what the dW GEMM writes and RCCL sums (f32 from 8 ranks on: grad_f32_default)
            "device_src_sha": device_src_sha(),
            "device_lib_sha": lib_sha,      # what the library that ran was linked from (kf_build_source_sha): equal to device_src_sha, or the build is stale
            "data_note": "uniform(-1, 1) operands; each attention operand is ONE random [H, S, D] batch element replicated over the batch at distinct "
                         "addresses (tests/test_gpu_baseline_sizes.py runs the same size with distinct data per head)",
            "attention_scores": "scaled operands (KF_ATTN_SCALED_OPERANDS)" if os.environ.get("KF_ATTN_SCALED_OPERANDS") else "exact f32",
            "checks": checks,
            "check_notes": CHECK_NOTES,
        }
        if ceiling:
            shape = KERNEL_MFMA_SHAPE.get(dom, "32x32x16")
            out["roofline"]["capped_mfma_tflops"] = ceiling      # bare MFMA loops on this box, this process, after the sustained loop
            out["roofline"]["frac_of_capped_mfma"] = achieved / ceiling[shape] if ceiling.get(shape) else None
            out["roofline"]["capped_mfma_shape"] = shape
            for k, v in kern.items():
                if "tflops" in v and ceiling.get(KERNEL_MFMA_SHAPE.get(k, "")):
                    v["frac_of_capped_mfma"] = v["tflops"] / ceiling[KERNEL_MFMA_SHAPE[k]]
        if sustained:
            out["ms_per_step_sustained"] = sustained[1] / sustained[0] * 1e3
            out["value_sustained"] = world * TOKENS_STEP / (sustained[1] / sustained[0])
            out["sustained_steps"] = sustained[0]
        if comm_events:
            ms = [e0.elapsed_ms(e1) for e0, e1 in comm_events]
            nbytes = GEMM_N * GEMM_N * (4 if wl.grad_f32 else 2)
            t = sum(ms) / len(ms) * 1e-3
            out["allreduce"] = {"ms": t * 1e3, "ms_p50": float(np.percentile(ms, 50)), "ms_max": float(max(ms)), "message_bytes": nbytes, "dtype": "f32" if wl.grad_f32 else "bf16",
                                "busbw_GBps": 2.0 * (world - 1) / world * nbytes / t / 1e9 if world > 1 else 0.0,
                                "xgmi_peak_GBps": XGMI_PEAK, "overlapped_with": "attention forward + backward"}
        if elapsed_off is not None:  # where the step time of an N-GPU job goes: compute alone, compute + collective, the collective alone
            ms_off, ms_on = (ab[0] + ab[2]) / 2, (ab[1] + ab[3]) / 2    # both arms unprofiled (the measured loop above carries ~0.1 ms of event records)
            exposed = ms_on - ms_off
            ar = out.get("allreduce", {}).get("ms")
            out["ms_per_step_no_comm"] = ms_off
            out["ms_per_step_comm_ab_off_on_off_on"] = ab
            out["exposed_comm_ms"] = exposed
            out["exposed_comm_noise_floor_ms"] = max(abs(ab[0] - ab[2]), abs(ab[1] - ab[3]))
            out["overlap_efficiency"] = (max(0.0, min(1.0, 1.0 - max(0.0, exposed) / ar)) if ar else None)
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
        if not all(checks.values()):
            rc = 1
    if pg is not None and rank != 0 and not all(checks.values()):
        rc = 1
    C.CDLL(None).fflush(None)
    if pg is not None:
        pg.barrier()  # every rank's library output is out before rank 0 prints THE line
    if rank == 0:
        print(json.dumps(out), flush=True)
    if pg is not None:
        pg.close()
    return rc


if __name__ == "__main__":
    sys.exit(main())
