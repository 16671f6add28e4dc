"""-m gpu: runtime half of the C ABI — device query, streams/events, cross-stream ordering, the
per-launch profiling mode, and the RCCL entry points (single-rank communicator: the binding, not the fabric)."""
import ctypes as C

import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O

pytestmark = pytest.mark.gpu


def test_device_props():
    p = H.device_props(0)
    assert p.arch.decode().startswith("gfx950"), p.arch
    assert p.compute_units == 256 and p.wavefront_size == 64
    assert p.total_mem > 200 * (1 << 30)  # 288 GB HBM3E


def test_streams_events_and_profile():
    s1, s2 = H.Stream(), H.Stream()
    n = 1 << 22
    a = H.DevBuf.from_numpy(np.ones(n, dtype=np.float32))
    b = H.DevBuf(4 * n)
    va, vb = H.View(a.ptr, (n,), (1,), H.F32), H.View(b.ptr, (n,), (1,), H.F32)
    ev = H.Event()
    H.profile_reset()
    H.profile_enable(True)
    H.elementwise(H.EW_ADD, H.make_desc([vb], [va, va]), H.F32, stream=s1.handle)   # b = 2 on s1
    ev.record(s1.handle)
    H.stream_wait_event(s2.handle, ev)
    H.elementwise(H.EW_MUL, H.make_desc([vb], [vb, vb]), H.F32, stream=s2.handle)   # b = 4 on s2, ordered behind s1
    s2.sync()
    H.profile_enable(False)
    assert np.array_equal(b.to_numpy((n,), np.float32), np.full(n, 4.0, dtype=np.float32))
    prof = H.profile_results()
    assert prof["ew_arith"][1] == 2 and prof["ew_arith"][0] > 0
    e0, e1 = H.Event(), H.Event()
    e0.record(s1.handle)
    H.elementwise(H.EW_ADD, H.make_desc([vb], [va, va]), H.F32, stream=s1.handle)
    e1.record(s1.handle)
    e1.sync()
    assert 0 < e0.elapsed_ms(e1) < 100


def test_rccl_single_rank_allreduce():
    ident = C.create_string_buffer(H.COMM_ID_BYTES)
    H.check(H.lib().kf_comm_unique_id(ident))
    comm = C.c_void_p()
    H.check(H.lib().kf_comm_init(C.byref(comm), ident.raw, 0, 1))
    x = np.arange(1 << 16, dtype=np.float32)
    d = H.DevBuf.from_numpy(x)
    s = H.Stream()
    H.check(H.lib().kf_allreduce_sum(comm, d.ptr, x.size, H.F32, s.handle))
    s.sync()
    assert np.array_equal(d.to_numpy(x.shape, np.float32), x)  # sum over one rank
    rc = H.lib().kf_allreduce_sum(comm, d.ptr, x.size, H.BOOL, s.handle)
    assert rc == H.KF_ERR_UNSUPPORTED
    H.check(H.lib().kf_comm_destroy(comm))


def test_graph_capture_of_a_launch_bound_sequence():
    """C1-size ops (1024 x 1024 fp32 add + sum) are launch-bound. The compute entries only enqueue on the caller's stream and
    own no memory, so a chain of them records into a HIP graph and replays with one submission: same bits, less host time."""
    import time
    s = H.Stream()
    n = 1024
    rng = np.random.default_rng(3)
    a_h, b_h = rng.uniform(-10, 10, (n, n)).astype(np.float32), rng.uniform(-10, 10, (n, n)).astype(np.float32)
    a, b = H.DevBuf.from_numpy(a_h), H.DevBuf.from_numpy(b_h)
    c, r = H.DevBuf(4 * n * n), H.DevBuf(4 * n)
    va, vb, vc = (H.View(x.ptr, (n, n), (n, 1), H.F32) for x in (a, b, c))
    vr = H.View(r.ptr, (n, 1), (1, 1), H.F32)
    d_add = H.make_desc([vc], [va, vb])
    d_acc = H.make_desc([vc], [vc, vb])
    d_sum = H.make_reduce_desc(vr, vc, 1)
    need = C.c_size_t(0)
    H.check(H.lib().kf_reduce_workspace_bytes(C.byref(d_sum), C.byref(need)))
    ws = H.DevBuf(max(need.value, 16))
    reps = 40

    def chain():  # c = a + b; c += b (reps times); r = sum(c, 1)
        H.elementwise(H.EW_ADD, d_add, H.F32, stream=s.handle)
        for _ in range(reps):
            H.elementwise(H.EW_ADD, d_acc, H.F32, stream=s.handle)
        H.check(H.lib().kf_reduce(H.RED_SUM, C.byref(d_sum), ws.ptr, need.value, s.handle))

    chain()
    s.sync()
    want_c, want_r = c.to_numpy((n, n), np.float32), r.to_numpy((n, 1), np.float32)
    ref = a_h + b_h
    for _ in range(reps):
        ref = ref + b_h
    assert np.array_equal(want_c, ref)  # fp32 adds are exactly rounded: bit-exact against numpy
    t0 = time.perf_counter()
    for _ in range(5):
        chain()
    s.sync()
    eager = (time.perf_counter() - t0) / 5
    with H.Graph.capture(s) as g:
        chain()
    H.check(H.lib().kf_memset_zero(c.ptr, 4 * n * n, s.handle))
    g.launch()
    s.sync()
    assert np.array_equal(c.to_numpy((n, n), np.float32), want_c) and np.array_equal(r.to_numpy((n, 1), np.float32), want_r)
    t0 = time.perf_counter()
    for _ in range(5):
        g.launch()
    s.sync()
    graph = (time.perf_counter() - t0) / 5
    print(f"chain of {reps + 2} launches: eager {eager * 1e6:.0f} us, graph {graph * 1e6:.0f} us")
    assert graph < eager * 1.2
    with pytest.raises(H.KfError):
        H.check(H.lib().kf_graph_begin_capture(None))


def test_graph_capture_of_a_training_step_sequence():
    """The bench's per-step sequence in miniature - the three GEMMs on the 256-tile kernels, attention forward + backward (MFMA
    kernels: LDS-DMA, large dynamic LDS) and a sort - captured into one HIP graph and replayed: bit-identical to the eager run."""
    s = H.Stream()
    rng = np.random.default_rng(5)
    M, N, K = 2560, 4096, 128

    def bf(shape):
        return O.f32_to_bf16(rng.uniform(-1, 1, shape).astype(np.float32))

    A, Bm, G = H.DevBuf.from_numpy(bf((M, K))), H.DevBuf.from_numpy(bf((K, N))), H.DevBuf.from_numpy(bf((M, N)))
    Cc, dA, dB = H.DevBuf(2 * M * N), H.DevBuf(2 * M * K), H.DevBuf(2 * K * N)
    Bh, Hh, S, D = 1, 2, 256, 128
    q, k, v, do = (H.DevBuf.from_numpy(bf((Bh, Hh, S, D))) for _ in range(4))
    o, dq, dk, dv = (H.DevBuf(2 * Bh * Hh * S * D) for _ in range(4))
    lse = H.DevBuf(4 * Bh * Hh * S)
    need = H.attn_bwd_workspace_bytes(H.BF16, Bh, Hh, S, S, D)
    ws = H.DevBuf(need)
    keys = H.DevBuf.from_numpy(rng.standard_normal((3, 20000)).astype(np.float32))
    skeys, spos = H.DevBuf(4 * 60000), H.DevBuf(8 * 60000)
    sneed = H.lib().kf_sort_workspace_bytes(H.F32, 3, 20000)
    sws = H.DevBuf(sneed)
    assert H.gemm_workspace_bytes(H.BF16, False, False, M, N, K) == 0  # the 256-tile kernels read every layout in place
    gneed = max(H.gemm_workspace_bytes(H.BF16, False, True, M, K, N), H.gemm_workspace_bytes(H.BF16, True, False, K, N, M), 16)
    gws = H.DevBuf(gneed)  # the small backward products take the 128-tile kernel and its re-layout scratch

    def step():
        H.gemm(H.BF16, False, False, M, N, K, 1.0, A.ptr, K, Bm.ptr, N, 0.0, Cc.ptr, N, stream=s.handle)      # C = A B
        H.gemm(H.BF16, False, True, M, K, N, 1.0, G.ptr, N, Bm.ptr, N, 0.0, dA.ptr, K, workspace=gws.ptr, workspace_bytes=gneed, stream=s.handle)  # dA = G B^T
        H.gemm(H.BF16, True, False, K, N, M, 1.0, A.ptr, K, G.ptr, N, 0.0, dB.ptr, N, workspace=gws.ptr, workspace_bytes=gneed, stream=s.handle)  # dB = A^T G
        H.attn_fwd(H.BF16, Bh, Hh, S, S, D, q.ptr, k.ptr, v.ptr, o.ptr, lse.ptr, stream=s.handle)
        H.attn_bwd(H.BF16, Bh, Hh, S, S, D, q.ptr, k.ptr, v.ptr, o.ptr, lse.ptr, do.ptr, dq.ptr, dk.ptr, dv.ptr, ws.ptr, need, stream=s.handle)
        H.check(H.lib().kf_sort(H.F32, keys.ptr, skeys.ptr, spos.ptr, 3, 20000, 1, sws.ptr, sneed, s.handle))

    outs = [(Cc, 2 * M * N), (dA, 2 * M * K), (dB, 2 * K * N), (o, 2 * Bh * Hh * S * D), (dq, 2 * Bh * Hh * S * D), (dk, 2 * Bh * Hh * S * D),
            (dv, 2 * Bh * Hh * S * D), (skeys, 4 * 60000), (spos, 8 * 60000)]
    step()
    s.sync()
    want = [b.to_numpy((n,), np.uint8).copy() for b, n in outs]
    with H.Graph.capture(s) as g:
        step()
    for b, n in outs:
        H.check(H.lib().kf_memset_zero(b.ptr, n, s.handle))
    g.launch()
    s.sync()
    for (b, n), w in zip(outs, want):
        assert np.array_equal(b.to_numpy((n,), np.uint8), w)


def test_bench_collective_path_and_checks_on_one_gpu():
    """`KF_BENCH_FORCE_COMM=1 python bench.py --gpus 1 --check`: the whole N > 1 code path of bench.py on one GPU — gloo rendezvous
    + RCCL communicator through kfunca_amd.parallel.ProcessGroup, the all-reduce of dW on its own stream behind an event, and
    after timing the checks: all-reduced dW == sum over ranks of each rank's own dW (here: bit-identical, one rank), sampled GEMM
    rows against the oracle, every element of attention head (0, 0) - O, LSE, dQ, dK, dV at S = 4096 - under the scale-aware bounds,
    sum dV == sum dO."""
    import json
    import os
    import subprocess
    import sys
    from pathlib import Path
    env = dict({k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE")}, KF_BENCH_FORCE_COMM="1")
    res = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--gpus", "1", "--check", "--steps", "3",
                          "--warmup", "1", "--sustain-seconds", "0.2", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert line["checks"] == {"allreduce_dw_vs_gloo_sum": True, "gemm_rows_vs_oracle": True, "gemm_dA_rows_vs_oracle": True, "gemm_dW_rows_vs_oracle": True,
                              "attn_head00_vs_oracle_scale_aware": True, "attn_last_batch_bit_identical": True, "attn_dv_checksum": True}, line["checks"]
    # round 4: where an N-GPU step's time goes - the same loop without the collective, the exposed communication, the overlap
    assert line["ms_per_step_no_comm"] > 0 and "exposed_comm_ms" in line and "overlap_efficiency" in line and line["allreduce"]["ms_max"] >= line["allreduce"]["ms_p50"] > 0
    worst = line["check_notes"]["attn_head00_worst_fraction_of_bound"]
    assert set(worst) == {"o", "dq", "dk", "dv", "lse"} and all(0 < v < 1 for v in worst.values()), worst
    assert line["allreduce"]["message_bytes"] == 4096 * 4096 * 2 and line["allreduce"]["ms"] > 0
    assert line["n_gpus"] == 1 and line["ms_per_step_sustained"] > 0 and line["roofline"]["frac"] > 0
    # the float gradient path (KF_BENCH_GRAD_F32): dW leaves the pair launch as float, RCCL sums floats, the same checks hold
    env["KF_BENCH_GRAD_F32"] = "1"
    res = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--gpus", "1", "--check", "--steps", "2",
                          "--warmup", "1", "--sustain-seconds", "0", "--no-cpu-baseline"], capture_output=True, text=True, env=env, timeout=900)
    assert res.returncode == 0, (res.stdout[-1500:], res.stderr[-1500:])
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert all(line["checks"].values()) and line["checks"]["gemm_dW_rows_vs_oracle"] and line["allreduce"]["dtype"] == "f32" and line["allreduce"]["message_bytes"] == 4096 * 4096 * 4
