"""CPU-only: the host core's TensorIterator geometry engine (no device memory involved), dtype
promotion, allocator size classes and loud failure without a GPU.
Expected geometries are the ones SURVEY.md §8(a6) / §3.3 derive from the reference
(src/core/tensor_iterator.cpp:486-515)."""
import pytest

import kfunca_amd as kf

F, D, I, L, H, BF, B8, U8 = kf.dtype.float, kf.dtype.double, kf.dtype.int, kf.dtype.long, kf.dtype.half, kf.dtype.bfloat16, kf.dtype.bool, kf.dtype.byte


def cont(shape):
    st, run = [], 1
    for s in reversed(shape):
        st.append(run)
        run *= s
    return list(reversed(st))


def op(shape, dtype=F, strides=None, defined=True):
    return (list(shape), cont(shape) if strides is None else list(strides), dtype, defined)


UNDEF = ([], [], F, False)


def test_c1_add_coalesces_to_one_dim():
    r = kf._iter_geometry([UNDEF], [op([1024, 1024]), op([1024, 1024])])
    assert r["ndim"] == 1 and r["shape"] == [1048576]
    assert r["stride_bytes"] == [[4], [4], [4]] and r["is_contiguous"]
    assert r["allocated"] == [[1024, 1024]]


def test_c1_sum_geometry():
    r = kf._iter_geometry([UNDEF], [op([1024, 1024])], is_reduction=True, reduce_dim=1, resize_outputs=False)
    assert r["shape"] == [1024, 1024] and r["stride_bytes"] == [[0, 4], [4, 4096]]  # reduced dim first, out stride 0
    assert r["num_output_elements"] == 1024 and r["allocated"] == [[1024, 1]]
    r = kf._iter_geometry([UNDEF], [op([1024, 1024])], is_reduction=True, reduce_dim=0, resize_outputs=False)
    assert r["stride_bytes"] == [[0, 4], [4096, 4]] and r["allocated"] == [[1, 1024]]
    r = kf._iter_geometry([UNDEF], [op([223, 23, 3213])], is_reduction=True, reduce_dim=1, resize_outputs=False)
    assert r["shape"] == [23, 3213, 223] and r["stride_bytes"][0] == [0, 4, 3213 * 4]


def test_broadcast_inplace_and_promotion():
    a, b = op([5, 7, 11]), op([5, 1, 11])
    r = kf._iter_geometry([a], [a, b], aliases=[-1, 0, -1])
    assert r["shape"] == [11, 7, 5] and r["stride_bytes"][2] == [4, 0, 44] and r["allocated"] == [[]]
    r = kf._iter_geometry([UNDEF], [op([2, 3], I), op([2, 3], F)])
    assert r["common_dtype"] == F and r["stride_bytes"] == [[4], [4], [4]]
    r = kf._iter_geometry([UNDEF], [op([16, 1]), op([1, 6])])
    assert r["shape"] == [6, 16] and r["stride_bytes"] == [[4, 24], [0, 4], [4, 0]] and r["allocated"] == [[16, 6]]
    with pytest.raises(RuntimeError):
        kf._iter_geometry([UNDEF], [op([2, 3]), op([3])])          # same-ndim rule (no rank broadcasting)
    with pytest.raises(RuntimeError):
        kf._iter_geometry([UNDEF], [op([2, 3]), op([2, 4])])       # not broadcastable
    with pytest.raises(RuntimeError):
        kf._iter_geometry([op([2, 1])], [op([2, 3]), op([2, 3])], resize_outputs=False)  # outputs cannot broadcast


def test_permuted_copy_and_slices():
    # permute(2,1,0,3) of [16,8,64,11] copied into a fresh contiguous tensor
    src = op([64, 8, 16, 11], D, [11, 64 * 11, 8 * 64 * 11, 1])
    r = kf._iter_geometry([op([64, 8, 16, 11], D)], [src], resize_outputs=False)
    assert r["ndim"] == 4 and r["shape"][0] == 11 and r["stride_bytes"][0][0] == 8 and r["stride_bytes"][1][0] == 8
    assert r["numel"] == 64 * 8 * 16 * 11
    # narrow() window of a cat result: dense inner block, strided outer
    r = kf._iter_geometry([op([5, 11, 23], F, [25 * 23, 23, 1])], [op([5, 11, 23])], resize_outputs=False, check_mem_overlap=False)
    assert r["shape"] == [253, 5] and r["stride_bytes"] == [[4, 2300], [4, 1012]]


def test_32bit_split_of_the_hard_shapes():
    big, bc = [2, 1024, 1024, 512], [2, 1024, 1, 512]
    r = kf._iter_geometry([UNDEF], [op(big), op(bc)])
    assert not r["can_use_32bit_indexing"] and r["pieces_32bit"] == 2 and r["pieces_numel"] == 2 ** 30
    r = kf._iter_geometry([UNDEF], [op(big), op(big)])
    assert r["is_contiguous"] and r["ndim"] == 1 and r["numel"] == 2 ** 30
    r = kf._iter_geometry([UNDEF], [op([8, 1024, 1024, 512], D), op([8, 1024, 1, 512], D)])
    assert r["pieces_32bit"] >= 16 and r["pieces_numel"] == 8 * 2 ** 29


def test_promotion_table_and_pools():
    p = kf._promote_types
    assert p(I, F) == F and p(H, BF) == BF and p(U8, kf.dtype.char) == kf.dtype.char and p(B8, U8) == U8
    assert p(L, H) == H and p(F, D) == D and p(I, L) == L
    # size classes <4K,64K,256K,1M,4M,64M,256M,inf (device_allocator.h:48-57)
    idx = kf._pool_index
    assert [idx(1), idx(4096), idx(4097), idx(1 << 20), idx((1 << 20) + 1), idx(1 << 28), idx((1 << 28) + 1)] == [0, 0, 1, 3, 4, 6, 7]


def test_loud_without_gpu():
    if kf.device_count() > 0:
        pytest.skip("GPU present")
    with pytest.raises(RuntimeError, match="no HIP device"):
        kf.empty([2, 3], F, 0)


def _walk(desc, bufs):
    """Execute out = in1 + in2 (or a copy) exactly as a kernel would: over the geometry's index space, operand t at byte offset
    sum_d idx[d] * stride_bytes[t][d]. Pure Python on numpy float32 buffers - small cases only."""
    import itertools

    import numpy as np
    shape, sb = desc["shape"], desc["stride_bytes"]
    for idx in itertools.product(*[range(s) for s in shape]):
        offs = [sum(i * s for i, s in zip(idx, st)) // 4 for st in sb]
        vals = [bufs[t][offs[t]] for t in range(1, len(bufs))]
        bufs[0][offs[0]] = np.float32(sum(vals))


def test_random_geometries_address_the_same_elements_as_numpy():
    """Adversarial property test of the 13 build stages (reorder with ambiguous strides, coalescing, broadcast strides, sliced and
    permuted operands, size-1 dims): whatever geometry comes out, walking it must compute exactly what numpy computes on the
    strided views - for 300 seeded random cases of rank 1..5. (The three fixed cases above pin the geometry itself; this pins
    its MEANING, which is what the kernels consume.)"""
    import numpy as np
    rng = np.random.default_rng(4242)
    for case in range(300):
        nd = int(rng.integers(1, 6))
        shape = [int(rng.integers(1, 5)) for _ in range(nd)]
        views, bufs = [], []
        for t in range(3):  # 0 = output, 1..2 = inputs
            # a dense parent in a random dim order, optionally wider than the view (slicing with a step), inputs optionally broadcast
            order = list(rng.permutation(nd))
            step = [int(rng.integers(1, 3)) for _ in range(nd)]
            vshape = list(shape)
            if t > 0:
                for d in range(nd):
                    if rng.random() < 0.25:
                        vshape[d] = 1
            pshape = [vshape[d] * step[d] for d in range(nd)]
            strides, run = [0] * nd, 1
            for d in reversed(order):
                strides[d] = run
                run *= pshape[d]
            buf = rng.integers(-50, 50, size=run).astype(np.float32)
            vstr = [strides[d] * step[d] for d in range(nd)]
            views.append((vshape, vstr))
            bufs.append(buf)
        two = rng.random() < 0.7
        ins = views[1:3] if two else views[1:2]
        try:
            r = kf._iter_geometry([op(views[0][0], F, views[0][1])], [op(v[0], F, v[1]) for v in ins], resize_outputs=False, check_mem_overlap=False)
        except RuntimeError:
            continue  # e.g. an output that would have to broadcast: rejected, as in the reference
        as_np = lambda b, v: np.lib.stride_tricks.as_strided(b, v[0], [s * 4 for s in v[1]])  # noqa: E731
        want = np.broadcast_to(as_np(bufs[1], ins[0]), shape).astype(np.float32)
        if two:
            want = want + np.broadcast_to(as_np(bufs[2], ins[1]), shape)
        out = bufs[0].copy()
        before = out.copy()
        _walk(r, [out] + bufs[1:1 + len(ins)])
        got = as_np(out, views[0])
        assert np.array_equal(got, want), (case, shape, views)
        # nothing outside the output view was written
        mask = np.zeros(out.shape, dtype=np.float32)  # same item size as the data: as_np's strides are in 4-byte elements
        as_np(mask, views[0])[...] = 1.0
        assert np.array_equal(out[mask == 0], before[mask == 0]), case
        assert r["numel"] == int(np.prod(shape)) and int(np.prod(r["shape"])) == r["numel"]
