"""-m gpu: mean_var / norm_stat statistics (kf_reduce_moments) through the C ABI vs the CPU oracle and the golden
vectors made from the reference tests' own expressions (test_tensor.py:120-146).

Tolerances: the reference asserts 1e-2 for mean_var (test_tensor.py:130-131) and its default 1e-3 for norm_stat
(test_tensor.py:145-146 via test/common.py:6-11). We hold the kernels to the accumulate type's accuracy against the
double two-pass oracle: f32 statistics rel 2e-5 (+ 1e-5 * |x|max absolute on the mean), f64 rel 1e-11; 16-bit
inputs accumulate in f32 and round once on store (bf16 2^-8, f16 2^-11 relative)."""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.gpu_util import Dev, gpu_moments, rand_of
from tests.helpers import assert_close, golden, regen

pytestmark = pytest.mark.gpu


def check(x, code, dim, mode, correction=1.0, eps=0.0, out_code=None, rel=2e-5, base=None):
    v, m = gpu_moments(mode, Dev(x, code, base=base), dim, correction, eps, out_code)
    wv, wm = O.moments(mode, x, dim, correction, eps, code=code, out_code=H.F64)
    oc = code if out_code is None else out_code
    gv, gm = O.to_float(v.get(), oc).astype(np.float64), O.to_float(m.get(), oc).astype(np.float64)
    amax = float(np.abs(O.to_float(x, code)).max()) if x.size else 0.0
    what = f"moments mode={mode} code={code} shape={x.shape} dim={dim}"
    assert_close(gm, wm, rtol=rel, atol=rel * amax, what=what + " mean")
    assert_close(gv, wv, rtol=4 * rel, atol=1e-30, what=what + " var-like")


def test_golden_mean_std_reference_case():
    g = golden("moments")
    (arr,) = regen(g["ms_seed"][0], [(13, 325, 127)], g["ms_sha"], dtype=np.float64)
    v, m = gpu_moments(H.MOM_VAR, Dev(arr), 1)
    assert_close(m.get(), g["ms_mean"], rtol=1e-2, atol=1e-2, what="mean (test_tensor.py:130)")
    assert_close(v.get(), g["ms_var"], rtol=1e-2, atol=1e-2, what="var (test_tensor.py:131)")
    assert_close(m.get(), g["ms_mean"], rtol=1e-11, atol=1e-12, what="mean f64 tight")
    assert_close(v.get(), g["ms_var"], rtol=1e-11, atol=0, what="var f64 tight")
    s, _ = gpu_moments(H.MOM_STD, Dev(arr), 1)
    assert_close(s.get(), np.sqrt(g["ms_var"]), rtol=1e-11, atol=0, what="take_sqrt")


@pytest.mark.parametrize("i", [0, 1, 2, 3])
def test_golden_norm_stat_reference_cases(i):
    g = golden("moments")
    shp = tuple(int(v) for v in g[f"ns{i}_shape"])
    (arr,) = regen(g[f"ns{i}_seed"][0], [shp], g[f"ns{i}_sha"])
    inv, m = gpu_moments(H.MOM_INVSTD, Dev(arr), 0, eps=1e-12)
    assert_close(m.get(), g[f"ns{i}_mean"], what=f"norm_stat mean {shp} (test_tensor.py:145)")
    assert_close(inv.get(), g[f"ns{i}_invstd"], what=f"norm_stat invstd {shp} (test_tensor.py:146)")
    assert_close(m.get(), g[f"ns{i}_mean"], rtol=2e-5, atol=2e-5, what="mean tight")
    assert_close(inv.get(), g[f"ns{i}_invstd"], rtol=5e-5, atol=0, what="invstd tight")


@pytest.mark.parametrize("shape,dim", [((4, 1 << 20), 1), ((1 << 20, 4), 0), ((3, 1 << 18, 5), 1), ((1, 70001), 1), ((70001, 1), 0),
                                       ((1000, 1000), 0), ((1000, 1000), 1), ((64, 48, 40), 0), ((64, 48, 40), 1), ((64, 48, 40), 2),
                                       ((5, 2048, 16), 1), ((2, 3), 1), ((7, 2), 0)])
def test_f32_shapes_paths(shape, dim):
    rng = np.random.default_rng(abs(hash(shape)) % 1000)
    x = rng.uniform(-10, 10, size=shape).astype(np.float32)
    for mode in (H.MOM_VAR, H.MOM_STD):
        check(x, H.F32, dim, mode)
    check(x, H.F32, dim, H.MOM_INVSTD, eps=1e-12)


def test_large_offset_is_stable():
    """Welford / Chan updates must not cancel: mean 1e4, spread 1 -> var ~ 1/12 (a sum-of-squares formula loses it in f32)."""
    rng = np.random.default_rng(41)
    x = (1e4 + rng.uniform(-0.5, 0.5, size=(64, 1 << 16))).astype(np.float32)
    for dim in (0, 1):
        v, m = gpu_moments(H.MOM_VAR, Dev(x), dim)
        wv, wm = O.moments(0, x, dim, out_code=H.F64)
        assert_close(m.get(), wm, rtol=1e-6, atol=0, what="mean")
        # ulp(1e4) = 1e-3 in f32, so every (x - mean) carries ~1e-3 relative rounding: 1e-2 is the f32 Welford bound,
        # a sum-of-squares formula would be off by O(10)
        assert_close(v.get(), wv, rtol=1e-2, atol=0, what="var with large offset")


def test_other_dtypes_and_f32_statistics_for_16bit():
    rng = np.random.default_rng(42)
    for shape, dim in (((37, 515), 1), ((37, 515), 0), ((9, 64, 33), 1), ((128, 4096), 1), ((4096, 128), 0)):
        check(rand_of(rng, shape, H.F64), H.F64, dim, H.MOM_VAR, rel=1e-11)
        for code, rel in ((H.BF16, 2.0 ** -7), (H.F16, 2.0 ** -10)):
            x = rand_of(rng, shape, code)
            check(x, code, dim, H.MOM_VAR, rel=rel)
            check(x, code, dim, H.MOM_INVSTD, correction=0.0, eps=1e-5, out_code=H.F32, rel=2e-5)  # layernorm-style statistics


def test_non_contiguous_input_generic_path():
    rng = np.random.default_rng(43)
    base = rng.uniform(-10, 10, size=(12, 20, 36)).astype(np.float32)
    for view in (base.transpose(2, 0, 1), base[:, ::2, 1:30:3], base.transpose(1, 2, 0)[::3]):
        for dim in range(3):
            check(view, H.F32, dim, H.MOM_VAR, base=base)


def test_degenerate_counts_follow_the_reference_projection():
    """n = 1 with correction 1: divisor 0 -> m2 / 0 = nan (0/0), exactly WelfordOps::project (reduce_ops_kernel.cu:122-128)."""
    x = np.arange(6, dtype=np.float32).reshape(6, 1)
    v, m = gpu_moments(H.MOM_VAR, Dev(x), 1)
    assert np.isnan(v.get()).all() and np.array_equal(m.get(), x)
    v0, _ = gpu_moments(H.MOM_VAR, Dev(x), 1, correction=0.0)
    assert np.array_equal(v0.get(), np.zeros_like(x))


def test_errors():
    x = Dev(np.zeros((4, 4), dtype=np.int32))
    o = Dev.empty((4, 1), H.I32)
    with pytest.raises(H.KfError) as e:
        H.reduce_moments(H.MOM_VAR, H.make_moments_desc(o.view, o.view, x.view, 1))
    assert e.value.code == H.KF_ERR_UNSUPPORTED  # floating types only, as the reference's dispatch
    xf, of, o64 = Dev(np.zeros((4, 4), dtype=np.float32)), Dev.empty((4, 1), H.F32), Dev.empty((4, 1), H.F64)
    with pytest.raises(H.KfError) as e:
        H.reduce_moments(H.MOM_VAR, H.make_moments_desc(o64.view, o64.view, xf.view, 1))
    assert e.value.code == H.KF_ERR_UNSUPPORTED
    with pytest.raises(H.KfError) as e:
        H.reduce_moments(7, H.make_moments_desc(of.view, of.view, xf.view, 1))
    assert e.value.code == H.KF_ERR_INVALID
