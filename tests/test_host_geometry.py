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
