"""-m gpu: stable segmented sort through the C ABI (kf_sort) vs the CPU oracle and the golden vectors. Bit-exact."""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.helpers import golden, sha

pytestmark = pytest.mark.gpu

NP_OF = {H.U8: np.uint8, H.I8: np.int8, H.I16: np.int16, H.I32: np.int32, H.I64: np.int64, H.F16: np.float16, H.BF16: np.uint16,
         H.F32: np.float32, H.F64: np.float64}


def draw(rng, shape, code, lo=-1000, hi=1000):
    if code == H.BF16:
        return O.f32_to_bf16(rng.uniform(lo, hi, size=shape).astype(np.float32))
    if code == H.U8:
        return rng.integers(0, 256, size=shape).astype(np.uint8)
    if code == H.I8:
        return rng.integers(-128, 128, size=shape).astype(np.int8)
    if code == H.I64:
        return rng.integers(-2 ** 62, 2 ** 62, size=shape)
    return rng.uniform(lo, hi, size=shape).astype(NP_OF[code])


def check(keys, code, desc):
    got_k, got_p = H.sort_segments(keys, desc, code=code)
    want_k, want_p = O.sort_stable(keys, 1, desc, code=code)
    assert np.array_equal(got_p, want_p), (code, keys.shape, desc)
    assert np.array_equal(got_k.view(np.uint8), want_k.view(np.uint8)), (code, keys.shape, desc)  # bit-exact incl. NaN payloads


def _cases(g, prefix, count_key):
    for n in range(int(g[count_key][0])):
        meta = g[f"{prefix}{n}_meta"]
        yield n, int(meta[0]), int(meta[1]), bool(meta[2]), [int(v) for v in meta[3:]], np.dtype(str(g[f"{prefix}{n}_dtype"]))


def test_golden_reference_sort_cases():
    # test_tensor.py:169-192; the device entry sorts [nseg, n] segments, so the sort dim is moved last on the host here
    # (the operator does that on the device: tests/test_gpu_host_api.py)
    g = golden("sort")
    for n, seed, dim, desc, shape, dt in _cases(g, "s", "n_sort"):
        arr = np.random.default_rng(seed).uniform(-1000, 1000, size=shape).astype(dt)
        assert np.array_equal(sha(arr), g[f"s{n}_sha_in"])
        moved = np.ascontiguousarray(np.moveaxis(arr, dim, -1))
        k, p = H.sort_segments(moved.reshape(-1, shape[dim]), desc)
        res = np.ascontiguousarray(np.moveaxis(k.reshape(moved.shape), -1, dim))
        ind = np.ascontiguousarray(np.moveaxis(p.reshape(moved.shape), -1, dim))
        if f"s{n}_res" in g:
            assert np.array_equal(res, g[f"s{n}_res"]) and np.array_equal(ind, g[f"s{n}_ind"]), (n, shape, dim, desc, dt)
        else:
            assert np.array_equal(sha(res, ind), g[f"s{n}_sha_out"]), (n, shape, dim, desc, dt)


def test_golden_large_slice():
    g = golden("sort")  # test_tensor.py:194-201: (4, 1024000) f32 vs np.sort / np.argsort(kind="stable")
    arr = np.random.default_rng(int(g["large_seed"][0])).uniform(-1000, 1000, size=(4, 1024000)).astype(np.float32)
    assert np.array_equal(sha(arr), g["large_sha_in"])
    k, p = H.sort_segments(arr, False)
    assert np.array_equal(sha(k, p), g["large_sha_out"])


def test_golden_topk_large_values():
    g = golden("sort")  # test_tensor.py:224-231: the first k of the descending sort
    for i in range(2):
        seed, k = (int(v) for v in g[f"tl{i}_meta"])
        arr = np.random.default_rng(seed).uniform(-10000, 10000, size=(4, 1024000)).astype(np.float32)
        assert np.array_equal(sha(arr), g[f"tl{i}_sha_in"])
        keys, _ = H.sort_segments(arr, True)
        assert np.array_equal(sha(keys[:, :k]), g[f"tl{i}_sha_out"])


@pytest.mark.parametrize("code", [H.U8, H.I8, H.I16, H.I32, H.I64, H.F16, H.BF16, H.F32, H.F64])
def test_every_dtype_every_path(code):
    rng = np.random.default_rng(700 + code)
    for nseg, n in ((1, 1), (1000, 1), (257, 2), (3, 2), (70, 3), (1000, 13), (333, 17), (100, 33), (50, 100), (37, 128), (20, 200), (9, 300), (11, 511), (3, 65), (100000, 3), (5, 514), (5, 64), (7, 65), (3, 512), (3, 513), (3, 1000), (2, 1024), (2, 1025), (2, 2048), (3, 2049), (2, 4096), (2, 4097), (2, 8192),
                    (3, 8193), (5, 22223), (2, 100000),
                    (2, 16384), (3, 73733), (1, 200003)):   # (whole 8192-key tiles: the 16-byte histogram loads; nine tiles and a ragged one: the XCD tile map)
        for desc in (False, True):
            check(draw(rng, (nseg, n), code), code, desc)


def test_stability_with_heavy_duplicates():
    rng = np.random.default_rng(711)
    for code, shape in ((H.I32, (3, 300000)), (H.F32, (2, 50000)), (H.I64, (2, 20000)), (H.I16, (64, 500)), (H.I32, (4, 3000)), (H.I64, (3, 6000)), (H.I8, (5, 700))):   # (the last three: the block-local radix path)
        keys = rng.integers(-3, 4, size=shape).astype(NP_OF[code])
        for desc in (False, True):
            check(keys, code, desc)
    check(np.zeros((2, 10000), np.float32), H.F32, True)  # all equal: positions must come out as iota


def test_special_float_values():
    # KeyTraits order (sorting_common.h:40-55, 186-202): -NaN < -inf < ... < -0.0 < +0.0 < ... < +inf < +NaN
    for dt, code, ut in ((np.float32, H.F32, np.uint32), (np.float64, H.F64, np.uint64), (np.float16, H.F16, np.uint16)):
        base = np.array([np.nan, np.inf, 1.0, 0.0, -0.0, -1.0, -np.inf, -np.nan, 0.0, -0.0, np.nan], dtype=dt)
        bits = base.view(ut).copy()
        bits[7] |= ut(1) << ut(8 * base.itemsize - 1)  # a NaN with the sign bit set, whatever np.nan's sign was
        base = bits.view(dt)
        for reps in (1, 100, 500, 1000):  # bitonic, block-local radix (1100 and 5500 keys), global radix
            keys = np.tile(base, reps)[None, :]
            for desc in (False, True):
                check(keys, code, desc)


def test_sorted_permutation_properties_large():
    # size-independent properties on a long segment: output ordered, positions a permutation, keys_out = keys_in[pos]
    rng = np.random.default_rng(712)
    keys = rng.standard_normal((2, 5_000_000)).astype(np.float32)
    for desc in (False, True):
        k, p = H.sort_segments(keys, desc)
        d = np.diff(k, axis=1)
        assert (d <= 0).all() if desc else (d >= 0).all()
        for s in range(2):
            assert np.array_equal(np.sort(p[s]), np.arange(keys.shape[1]))
            assert np.array_equal(k[s], keys[s][p[s]])
            ties = k[s][1:] == k[s][:-1]
            assert (np.diff(p[s])[ties] > 0).all()  # equal keys keep input order in both directions


def test_empty_and_errors():
    k, p = H.sort_segments(np.zeros((0, 5), np.float32))
    assert k.shape == (0, 5) and p.shape == (0, 5)
    k, p = H.sort_segments(np.zeros((4, 0), np.float32))
    assert k.shape == (4, 0)
    buf, pos = H.DevBuf(64), H.DevBuf(128)
    with pytest.raises(H.KfError, match="bool"):
        H.check(H.lib().kf_sort(H.BOOL, buf.ptr, pos.ptr, pos.ptr, 1, 16, 0, None, 0, None))
    with pytest.raises(H.KfError, match="different buffers"):
        H.check(H.lib().kf_sort(H.F32, buf.ptr, buf.ptr, pos.ptr, 1, 16, 0, None, 0, None))
    big = H.DevBuf(4 * 20000)
    out, pos2 = H.DevBuf(4 * 20000), H.DevBuf(8 * 20000)
    assert H.lib().kf_sort_workspace_bytes(H.F32, 1, 20000) > 0
    with pytest.raises(H.KfError, match="workspace"):
        H.check(H.lib().kf_sort(H.F32, big.ptr, out.ptr, pos2.ptr, 1, 20000, 0, None, 0, None))


def test_many_tiles_and_segments():
    # a few hundred tiles per segment (two scan levels), several segments in one launch, ragged last tiles, every key width
    rng = np.random.default_rng(714)
    for code, shape in ((H.F32, (1, 3_000_000)), (H.I32, (5, 400_003)), (H.F64, (2, 700_001)), (H.U8, (3, 300_000)), (H.BF16, (2, 1_000_000)), (H.I64, (7, 70_000))):
        for desc in (False, True):
            check(draw(rng, shape, code), code, desc)
