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
