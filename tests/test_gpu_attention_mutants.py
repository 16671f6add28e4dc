"""-m gpu: mutation test of the attention parity bounds ON THE KERNELS.

kfunca_amd/_build/libkfunca_hip_mutant.so is attention.hip compiled with -DKF_MUTANT (kfunca_amd/_build.py: build_mutant; never part
of libkfunca_hip.so): the production kernels plus four deliberate single-tile defects behind kfmut_select() (KF_MUT in attention.hip):
  1  the forward's last 256-query block skips key tile 1          2  queries 128-159 give nothing to the first 128-key block
  3  the stored-dS dQ kernel's last block skips key step 0        4  the recomputing dQ kernel's last block skips key step 0
With the selector at 0 the mutant library must PASS the same scale-aware bounds as the product (and agree with it bit for bit);
with a defect switched on the bounds must FAIL on exactly the outputs that defect reaches - at config C3's own sequence length,
where the round-2 absolute tolerances could not see a missing tile."""
import ctypes as C

import numpy as np
import pytest

from kfunca_amd import _build
from kfunca_amd import hip_abi as H
from oracle import checks as K
from oracle import oracle as O

pytestmark = pytest.mark.gpu
MUT = _build.BUILD / "libkfunca_hip_mutant.so"


@pytest.fixture(scope="module")
def mut():
    if not MUT.exists():
        _build.build_mutant()
    lib = C.CDLL(str(MUT))
    lib.kf_attn_fwd.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.c_void_p] * 6
    lib.kf_attn_bwd.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.c_void_p] * 10 + [C.c_size_t, C.c_void_p]
    lib.kf_attn_bwd_workspace_bytes.argtypes = [C.c_int] + [C.c_int64] * 5 + [C.POINTER(C.c_size_t)]
    yield lib
    lib.kfmut_select(0)


def run(lib, code, q, k, v, go, which, split=False):
    B, Hh, S, D = q.shape
    bufs = [H.DevBuf.from_numpy(x) for x in (q, k, v, go)]
    o, lse = H.DevBuf(q.nbytes), H.DevBuf(4 * B * Hh * S)
    dq, dk, dv = (H.DevBuf(q.nbytes) for _ in range(3))
    with H.knobs(KF_ATTN_SPLIT_BWD="1" if split else None):
        H.check(lib.kf_knobs_reload())  # the mutant library keeps its own copy of the A/B switches
        need = C.c_size_t(0)
        H.check(lib.kf_attn_bwd_workspace_bytes(code, B, Hh, S, S, D, C.byref(need)))
        ws = H.DevBuf(max(need.value, 256))
        H.check(lib.kfmut_select(which))
        H.check(lib.kf_attn_fwd(code, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr, None))
        H.check(lib.kf_attn_bwd(code, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr, bufs[3].ptr, dq.ptr, dk.ptr, dv.ptr,
                                ws.ptr, need.value, None))
        H.device_sync()
        H.check(lib.kfmut_select(0))
    H.check(lib.kf_knobs_reload())
    return dict(o=o.to_numpy(q.shape, q.dtype), lse=lse.to_numpy((B, Hh, S), np.float32), dq=dq.to_numpy(q.shape, q.dtype),
                dk=dk.to_numpy(q.shape, q.dtype), dv=dv.to_numpy(q.shape, q.dtype))


def verdicts(code, q, k, v, go, ref, r):
    """Which outputs pass the scale-aware bounds."""
    out = {}
    for n in K.NAMES:
        try:
            K.check_one(n, r[n], ref, code, what="mutant")
            out[n] = True
        except AssertionError:
            out[n] = False
    return out


@pytest.mark.parametrize("D", [128, 64])
def test_every_defect_is_caught_and_the_clean_build_passes(mut, D):
    code, B, Hh, S = H.BF16, 1, 2, 4096
    rng = np.random.default_rng(40 + D)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32), code) for _ in range(4))
    ref = O.attn_ref64(q, k, v, go, code=code)
    clean = run(mut, code, q, k, v, go, 0)
    assert verdicts(code, q, k, v, go, ref, clean) == dict(o=True, dq=True, dk=True, dv=True)
    # the mutant build with nothing switched on IS the product: bit-identical to libkfunca_hip.so
    from tests.test_gpu_attention import bwd, fwd
    o, lse = fwd(code, q, k, v)
    prod = dict(zip(("dq", "dk", "dv"), bwd(code, q, k, v, o, lse, go)), o=o)
    for n in K.NAMES:
        assert np.array_equal(clean[n].view(np.uint16), prod[n].view(np.uint16)), n
    # 1: a forward tile dropped -> O wrong (and, through O / LSE, every gradient of those rows: only O is asserted)
    assert verdicts(code, q, k, v, go, ref, run(mut, code, q, k, v, go, 1))["o"] is False
    # 2: one slice x key block dropped in dK/dV -> dK, dV of those keys and dQ of that slice (through the stored dS) wrong, O untouched
    assert verdicts(code, q, k, v, go, ref, run(mut, code, q, k, v, go, 2)) == dict(o=True, dq=False, dk=False, dv=False)
    # 3: one key step dropped in the stored-dS dQ kernel -> only dQ wrong
    assert verdicts(code, q, k, v, go, ref, run(mut, code, q, k, v, go, 3)) == dict(o=True, dq=False, dk=True, dv=True)
    if D == 128:  # 4: the same in the recomputing dQ kernel (the form behind KF_ATTN_SPLIT_BWD)
        assert verdicts(code, q, k, v, go, ref, run(mut, code, q, k, v, go, 4, split=True)) == dict(o=True, dq=False, dk=True, dv=True)
        assert verdicts(code, q, k, v, go, ref, run(mut, code, q, k, v, go, 0, split=True)) == dict(o=True, dq=True, dk=True, dv=True)


def test_round2_tolerances_would_have_passed_the_defects(mut):
    """For the record: the absolute tolerances this round replaced (rtol 2e-2, atol 3e-2) accept defect 3 - a dQ short of 64 keys."""
    code, B, Hh, S, D = H.BF16, 1, 1, 4096, 128
    rng = np.random.default_rng(7)
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, (B, Hh, S, D)).astype(np.float32), code) for _ in range(4))
    bad = run(mut, code, q, k, v, go, 3)
    want = O.attn_bwd(q, k, v, go, code=code)[0]
    assert np.allclose(K.to_f64(bad["dq"], code), K.to_f64(want, code), rtol=2e-2, atol=3e-2)  # the old test: green on a wrong gradient
    ref = O.attn_ref64(q, k, v, go, code=code)
    with pytest.raises(AssertionError):
        K.check_one("dq", bad["dq"], ref, code)
