"""-m gpu: degenerate shapes through the operator API and the C ABI — empty tensors, extents of 1, K = 0, one-element
reductions — the cases the reference's iterator handles by early-outs (tensor_iterator.cpp:181-244, gemm_kernel.cu:9-25)."""
import numpy as np
import pytest

import kfunca_amd as kfunca
from kfunca_amd import hip_abi as H

pytestmark = pytest.mark.gpu


def test_empty_tensors_through_the_operator_api():
    e = kfunca.empty([0, 5], kfunca.float, 0)
    assert e.numel() == 0 and e.sizes() == [0, 5]
    assert (e + e).sizes() == [0, 5]
    assert (e * 2.0).sizes() == [0, 5]
    assert e.contiguous().numpy().shape == (0, 5)
    assert e.permute(1, 0).contiguous().numpy().shape == (5, 0)
    z = kfunca.zeros([4, 0, 3], kfunca.int, 0)
    assert z.numpy().shape == (4, 0, 3)
    s = kfunca.from_numpy(np.zeros((3, 0), dtype=np.float32), 0).sum(0)
    assert s.sizes() == [1, 0]
    parts = kfunca.cat([kfunca.empty([0, 4], kfunca.float, 0), kfunca.from_numpy(np.ones((2, 4), dtype=np.float32), 0)], 0)
    assert np.array_equal(parts.numpy(), np.ones((2, 4), dtype=np.float32))


def test_extent_one_and_scalar_like_shapes():
    rng = np.random.default_rng(50)
    a = rng.uniform(-10, 10, (1, 1, 1)).astype(np.float32)
    t = kfunca.from_numpy(a, 0)
    assert np.array_equal((t + t).numpy(), a + a)
    assert np.array_equal(t.sum(1).numpy(), a) and np.array_equal(t.mean(2).numpy(), a)
    col = rng.uniform(-10, 10, (7, 1)).astype(np.float32)
    row = rng.uniform(-10, 10, (1, 9)).astype(np.float32)
    assert np.array_equal((kfunca.from_numpy(col, 0) * kfunca.from_numpy(row, 0)).numpy(), col * row)  # outer product by broadcasting
    big = rng.uniform(-10, 10, (1, 300001)).astype(np.float32)
    np.testing.assert_allclose(kfunca.from_numpy(big, 0).sum(1).numpy(), big.sum(axis=1, keepdims=True, dtype=np.float64), rtol=1e-5, atol=1e-2)


def test_gemm_degenerate_extents_c_abi():
    a = H.DevBuf.from_numpy(np.ones((4, 4), dtype=np.float32))
    c_h = np.full((4, 4), 3.0, dtype=np.float32)
    c = H.DevBuf.from_numpy(c_h)
    H.gemm(H.F32, False, False, 0, 4, 4, 1.0, a.ptr, 4, a.ptr, 4, 0.0, c.ptr, 4)  # M = 0: nothing to do
    H.gemm(H.F32, False, False, 4, 0, 4, 1.0, a.ptr, 4, a.ptr, 4, 0.0, c.ptr, 4)  # N = 0
    H.device_sync()
    assert np.array_equal(c.to_numpy((4, 4), np.float32), c_h)
    H.gemm(H.F32, False, False, 4, 4, 0, 1.0, a.ptr, 4, a.ptr, 4, 2.0, c.ptr, 4)  # K = 0: C = beta * C
    H.device_sync()
    assert np.array_equal(c.to_numpy((4, 4), np.float32), 2.0 * c_h)
    H.gemm(H.F32, False, False, 4, 4, 0, 1.0, a.ptr, 4, a.ptr, 4, 0.0, c.ptr, 4)  # K = 0, beta = 0: zeros, C is not read
    H.device_sync()
    assert not c.to_numpy((4, 4), np.float32).any()


def test_attention_degenerate_extents_c_abi():
    q = H.DevBuf(1024)
    lse = H.DevBuf(64)
    H.attn_fwd(H.BF16, 0, 4, 128, 128, 128, q.ptr, q.ptr, q.ptr, q.ptr, lse.ptr)  # B = 0
    H.attn_fwd(H.F32, 1, 1, 0, 8, 16, q.ptr, q.ptr, q.ptr, q.ptr, lse.ptr)        # Sq = 0
    with pytest.raises(H.KfError) as e:
        H.attn_fwd(H.F32, 1, 1, 4, 0, 16, q.ptr, q.ptr, q.ptr, q.ptr, lse.ptr)    # no keys: softmax over nothing
    assert e.value.code == H.KF_ERR_INVALID
    # one query, one key: the output is V, lse = s
    one = np.array([[[[0.5, -1.0, 2.0, 4.0]]]], dtype=np.float32)
    d = H.DevBuf.from_numpy(one)
    o, l = H.DevBuf(16), H.DevBuf(4)
    H.attn_fwd(H.F32, 1, 1, 1, 1, 4, d.ptr, d.ptr, d.ptr, o.ptr, l.ptr)
    H.device_sync()
    assert np.allclose(o.to_numpy((4,), np.float32), one.ravel())
    assert np.allclose(l.to_numpy((1,), np.float32), (one ** 2).sum() / 2.0)
