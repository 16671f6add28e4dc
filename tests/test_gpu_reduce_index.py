"""-m gpu: sum / mean and index_put_ through the C ABI vs the CPU oracle and the golden vectors."""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.gpu_util import Dev, gpu_reduce, rand_of
from tests.helpers import assert_close, golden, regen

pytestmark = pytest.mark.gpu
# Floating sums: the reference's own tolerance is 1e-2 (test_tensor.py:118). We hold ourselves to
# f32 tree-summation accuracy against the double-accumulated oracle: |err| <= 2e-6 * sum|x|.


def check_float_reduce(x, code, dim, rtol=1e-2, atol=1e-2):
    for hop, oop in ((H.RED_SUM, O.SUM), (H.RED_MEAN, O.MEAN)):
        got = gpu_reduce(hop, Dev(x, code), dim).get()
        want = O.reduce(oop, x, dim, code=code)
        g, w = O.to_float(got, code), O.to_float(want, code)
        assert_close(g, w, rtol=rtol, atol=atol, what=f"reduce op={hop} dim={dim} shape={x.shape}")
        if code == H.F32:  # tight bound, scaled by the magnitude actually summed
            mag = np.sum(np.abs(x.astype(np.float64)), axis=dim, keepdims=True) * (1.0 if hop == H.RED_SUM else 1.0 / x.shape[dim])
            assert (np.abs(g.astype(np.float64) - w) <= 2e-6 * mag + 1e-30).all(), (hop, dim, x.shape)


def test_golden_reductions():
    g = golden("reductions")
    x = g["x"]
    for dim in range(3):
        assert_close(gpu_reduce(H.RED_SUM, Dev(x), dim).get(), g[f"sum{dim}"], rtol=1e-2, atol=1e-2, what=f"sum{dim}")
        assert_close(gpu_reduce(H.RED_MEAN, Dev(x), dim).get(), g[f"mean{dim}"], rtol=1e-2, atol=1e-2, what=f"mean{dim}")
        assert np.array_equal(gpu_reduce(H.RED_SUM, Dev(g["xi"]), dim).get(), g[f"isum{dim}"])  # integers: exact
        assert not gpu_reduce(H.RED_MEAN, Dev(g["xi"]), dim).get().any()  # integer mean factor is 0 in the reference
    a, _ = regen(g["c1_seed"][0], [(1024, 1024), (1024, 1024)], g["c1_sha"])  # BASELINE config C1
    assert_close(gpu_reduce(H.RED_SUM, Dev(a), 0).get(), g["c1_sum0"], rtol=1e-5, atol=1e-2, what="C1 sum(0)")
    assert_close(gpu_reduce(H.RED_SUM, Dev(a), 1).get(), g["c1_sum1"], rtol=1e-5, atol=1e-2, what="C1 sum(1)")


def test_reference_shape_every_dim():
    rng = np.random.default_rng(31)
    x = rng.uniform(-10, 10, size=(223, 23, 3213)).astype(np.float32)  # test_tensor.py:110-118
    for dim in range(3):
        check_float_reduce(x, H.F32, dim)


@pytest.mark.parametrize("shape,dim", [((4, 1 << 20), 1), ((1 << 20, 4), 0), ((3, 1 << 18, 5), 1), ((1, 70001), 1),
                                       ((70001, 1), 0), ((1000, 1000), 0), ((1000, 1000), 1), ((129, 1), 1), ((1, 1), 0),
                                       ((64, 48, 40), 0), ((64, 48, 40), 1), ((64, 48, 40), 2), ((5, 2048, 16), 1)])
def test_float_shapes_paths(shape, dim):
    rng = np.random.default_rng(hash(shape) % 1000)
    check_float_reduce(rng.uniform(-10, 10, size=shape).astype(np.float32), H.F32, dim)


def test_other_dtypes():
    rng = np.random.default_rng(33)
    for code, rtol in ((H.F64, 1e-12), (H.F16, 1e-2), (H.BF16, 5e-2)):
        for shape, dim in (((37, 515), 1), ((37, 515), 0), ((9, 64, 33), 1)):
            x = rand_of(rng, shape, code)
            for hop, oop in ((H.RED_SUM, O.SUM), (H.RED_MEAN, O.MEAN)):
                got = gpu_reduce(hop, Dev(x, code), dim).get()
                want = O.reduce(oop, x, dim, code=code)
                assert_close(O.to_float(got, code), O.to_float(want, code), rtol=rtol, atol=rtol * 10, what=f"{code} {shape} {dim}")
    for code in (H.U8, H.I8, H.I16, H.I32, H.I64, H.BOOL):  # exact, with in-dtype wraparound
        for shape, dim in (((37, 515), 1), ((37, 515), 0), ((9, 64, 33), 1), ((4, 1 << 16), 1)):
            x = rand_of(rng, shape, code)
            got = gpu_reduce(H.RED_SUM, Dev(x, code), dim).get()
            assert np.array_equal(got, O.reduce(O.SUM, x, dim, code=code)), (code, shape, dim)


def test_non_contiguous_input_generic_path():
    rng = np.random.default_rng(34)
    base = rng.uniform(-10, 10, size=(12, 20, 36)).astype(np.float32)
    for view in (base.transpose(2, 0, 1), base[:, ::2, 1:30:3], base.transpose(1, 2, 0)[::3]):
        for dim in range(3):
            for hop, oop in ((H.RED_SUM, O.SUM), (H.RED_MEAN, O.MEAN)):
                got = gpu_reduce(hop, Dev(view, base=base), dim).get()
                assert_close(got, O.reduce(oop, view, dim), rtol=1e-5, atol=1e-4, what=f"{view.shape} {dim}")


def test_workspace_contract():
    x = Dev(np.ones((2, 1 << 20), dtype=np.float32))
    out = Dev.empty((2, 1), H.F32)
    d = H.make_reduce_desc(out.view, x.view, 1)
    import ctypes as C
    need = C.c_size_t(0)
    H.check(H.lib().kf_reduce_workspace_bytes(C.byref(d), C.byref(need)))
    assert need.value > 0
    rc = H.lib().kf_reduce(H.RED_SUM, C.byref(d), None, 0, None)
    assert rc == H.KF_ERR_WORKSPACE
    # reproducible run to run (no atomics): identical bits
    a = gpu_reduce(H.RED_SUM, x, 1).get()
    b = gpu_reduce(H.RED_SUM, x, 1).get()
    assert np.array_equal(a, b) and a[0, 0] == float(1 << 20)


def test_golden_index_put():
    g = golden("shape_ops")

    def run(x, idx, vals, code=None):
        self_ = Dev(x.copy(), code)
        vd = Dev(vals, code)
        ids = [Dev(np.ascontiguousarray(i, dtype=np.int64)) for i in idx]
        # self viewed with stride 0 over the index shape (index_ops.cpp:23-25)
        sv = H.View(self_.buf.ptr, vals.shape, (0,) * vals.ndim, self_.code)
        d = H.make_desc([sv], [vd.view] + [i.view for i in ids])
        es = x.itemsize
        H.index_put(d, list(x.shape), [s * es for s in (np.array(x.strides) // es)])
        H.device_sync()
        return self_.get()

    assert np.array_equal(run(g["iput_x"], [g["iput_i0"], g["iput_i1"]], g["iput_v"]), g["iput_out"])
    assert np.array_equal(run(g["iput3_x"], [g["iput3_i0"], g["iput3_i1"], g["iput3_i2"]], g["iput3_v"]), g["iput3_out"])
    # larger scatter without duplicates, every element size, vs the oracle
    rng = np.random.default_rng(35)
    for code in (H.U8, H.I16, H.F32, H.F64, H.BF16):
        x = rand_of(rng, (300, 70), code)
        flat = rng.permutation(300 * 70)[:5000]
        i0, i1 = (flat // 70).astype(np.int64), (flat % 70).astype(np.int64)
        i0[::7] -= 300  # negative indices wrap once
        vals = rand_of(rng, (5000,), code)
        want = O.index_put(x.copy(), [i0, i1], vals, code=code)
        assert np.array_equal(run(x, [i0, i1], vals, code), want), code
