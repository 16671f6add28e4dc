"""-m gpu: runtime half of the C ABI — device query, streams/events, cross-stream ordering, the
per-launch profiling mode, and the RCCL entry points (single-rank communicator: the binding, not the fabric)."""
import ctypes as C

import numpy as np
import pytest

from kfunca_amd import hip_abi as H

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
