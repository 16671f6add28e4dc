"""-m gpu: elementwise loops through the C ABI vs the CPU oracle and the golden vectors.
Bit-exact everywhere: IEEE add/sub/mul/div are correctly rounded on both sides, integer and byte
work is exact, copies move bits."""
import numpy as np
import pytest

from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.gpu_util import Dev, gpu_binary, gpu_copy, gpu_fill, rand_of
from tests.helpers import golden

pytestmark = pytest.mark.gpu
OPS = {"add": (H.EW_ADD, O.ADD), "sub": (H.EW_SUB, O.SUB), "mul": (H.EW_MUL, O.MUL), "div": (H.EW_DIV, O.DIV)}
ALL = [H.BOOL, H.U8, H.I8, H.I16, H.I32, H.I64, H.F16, H.BF16, H.F32, H.F64]


def bits(x):
    return np.ascontiguousarray(x).view(np.uint8)


def test_golden_add_promotion_int():
    g = golden("elementwise")
    for i in range(3):
        a = Dev(g[f"add{i}_a"])
        assert np.array_equal(gpu_binary(H.EW_ADD, a, a).get(), g[f"add{i}_out"])
        out = gpu_binary(H.EW_ADD, Dev(g[f"promo{i}_a"]), Dev(g[f"promo{i}_b"]))
        assert out.code == H.F32 and np.array_equal(out.get(), g[f"promo{i}_out"])
    s = golden("shape_ops")
    x = Dev(s["int_x"])
    assert np.array_equal(gpu_binary(H.EW_ADD, x, x).get(), s["int_out"])  # test/core/test_tensor.cpp:10-23


def test_golden_inplace_chain():
    g = golden("elementwise")
    a, b = Dev(g["inpl_a"].copy()), Dev(g["inpl_b"])
    for k, op in enumerate((H.EW_ADD, H.EW_SUB, H.EW_MUL, H.EW_DIV)):
        gpu_binary(op, a, b, out=a)
        assert np.array_equal(a.get(), g["inpl_steps"][k]), k
    for k, (op, s) in enumerate(((H.EW_ADD, 2), (H.EW_SUB, 3), (H.EW_MUL, 4), (H.EW_DIV, 5))):
        sc = gpu_fill(Dev.empty(a.arr.shape, H.F32), s)
        gpu_binary(op, a, sc, out=a)
        assert np.array_equal(a.get(), g["inpl_steps"][4 + k]), k


def test_golden_broadcast():
    g = golden("elementwise")
    for i in range(3):
        a, b = Dev(g[f"bc{i}_a"]), Dev(g[f"bc{i}_b"])
        for name, (hop, _) in OPS.items():
            assert np.array_equal(gpu_binary(hop, a, b).get(), g[f"bc{i}_{name}"]), (i, name)
        assert np.array_equal(gpu_binary(H.EW_MUL, Dev(g[f"bc{i}_ai"]), b).get(), g[f"bc{i}_imul"])


def test_golden_convert():
    g = golden("elementwise")
    x = Dev(g["cvt_x"])
    h = gpu_copy(x, Dev.empty(x.arr.shape, H.F16))
    assert np.array_equal(h.get().view(np.uint16), g["cvt_half_bits"])
    sq = gpu_binary(H.EW_MUL, h, h)
    assert np.array_equal(gpu_copy(sq, Dev.empty(x.arr.shape, H.F32)).get(), g["cvt_half_sq"])
    bf = gpu_copy(x, Dev.empty(x.arr.shape, H.BF16))
    assert np.array_equal(bf.get(), g["cvt_bf16_bits"])
    sq = gpu_binary(H.EW_MUL, bf, bf)
    assert np.array_equal(gpu_copy(sq, Dev.empty(x.arr.shape, H.F32)).get(), g["cvt_bf16_sq"])
    big = Dev(g["cvt_big"])
    assert np.array_equal(gpu_copy(big, Dev.empty(big.arr.shape, H.BF16)).get(), g["cvt_big_bf16"])
    assert np.array_equal(gpu_copy(big, Dev.empty(big.arr.shape, H.F16)).get().view(np.uint16), g["cvt_big_f16"])


@pytest.mark.parametrize("code", ALL)
def test_same_dtype_ops_every_dtype(code):
    rng = np.random.default_rng(200 + code)
    for shape in ((1,), (3,), (1000,), (64, 257), (7, 33, 65)):
        a, b = rand_of(rng, shape, code), rand_of(rng, shape, code)
        if code in (H.U8, H.I8, H.I16, H.I32, H.I64):
            b = np.where(b == 0, 3, b).astype(b.dtype)  # x / 0 is UB in the reference
        da, db = Dev(a, code), Dev(b, code)
        for name, (hop, oop) in OPS.items():
            if code == H.BOOL and name == "div":
                b2 = np.ones_like(b)
                got = gpu_binary(hop, da, Dev(b2, code)).get()
                want = O.binary(oop, a, b2, a_code=code, b_code=code)
            else:
                got = gpu_binary(hop, da, db).get()
                want = O.binary(oop, a, b, a_code=code, b_code=code)
            assert np.array_equal(bits(got), bits(want)), (code, shape, name)


def test_mixed_dtypes_and_output_cast():
    rng = np.random.default_rng(7)
    pairs = [(H.I32, H.F32), (H.U8, H.I8), (H.BOOL, H.I32), (H.F16, H.BF16), (H.I64, H.F16), (H.F32, H.F64),
             (H.BF16, H.F32), (H.I16, H.F64), (H.BOOL, H.U8)]
    for ca, cb in pairs:
        a, b = rand_of(rng, (17, 5, 9), ca), rand_of(rng, (17, 1, 9), cb)
        for name, (hop, oop) in (("add", OPS["add"]), ("mul", OPS["mul"]), ("sub", OPS["sub"])):
            got = gpu_binary(hop, Dev(a, ca), Dev(b, cb))
            want = O.binary(oop, a, b, a_code=ca, b_code=cb)
            assert got.code == O.promote(ca, cb)
            assert np.array_equal(bits(got.get()), bits(want)), (ca, cb, name)
    for ca, cb in pairs + [(H.F32, H.BF16), (H.F64, H.I8), (H.I32, H.I64)]:  # contiguous, whole groups of eight: the wide mixed-dtype kernel
        a, b = rand_of(rng, (40, 8, 5), ca), rand_of(rng, (40, 8, 5), cb)
        for name, (hop, oop) in (("add", OPS["add"]), ("mul", OPS["mul"]), ("sub", OPS["sub"])):
            got = gpu_binary(hop, Dev(a, ca), Dev(b, cb))
            assert np.array_equal(bits(got.get()), bits(O.binary(oop, a, b, a_code=ca, b_code=cb))), (ca, cb, name, "wide")
    # provided output of another dtype: result is cast on store (tensor_memory_access.h:26-37)
    a, b = rand_of(rng, (100,), H.F32), rand_of(rng, (100,), H.F32)
    out = gpu_binary(H.EW_ADD, Dev(a), Dev(b), out=Dev.empty((100,), H.I32))
    want = O.binary_out(O.ADD, a, b, np.empty(100, dtype=np.int32))
    assert np.array_equal(out.get(), want)


def test_binary_with_a_transposed_operand():
    """x (op) y.T and y.T (op) x through the LDS-tiled binary kernel (whole 64-tiles, with and without batch dims, non-commutative
    operators in both operand orders), bit-exact; ragged shapes and a transposed operand that is also the output keep the strided kernel."""
    rng = np.random.default_rng(81)
    for code in (H.F32, H.BF16, H.F16, H.I32):
        for shape, perm in (((512, 1024), (1, 0)), ((3, 256, 128), (0, 2, 1)), ((2, 2, 128, 192), (0, 1, 3, 2)), ((192, 3, 128), (2, 1, 0)), ((70, 130), (1, 0))):
            yb = rand_of(rng, shape, code)
            yt = yb.transpose(perm)
            x = rand_of(rng, yt.shape, code)
            if code == H.I32:
                x = np.where(x == 0, 3, x).astype(x.dtype)
                yb = np.where(yb == 0, 5, yb).astype(yb.dtype)
                yt = yb.transpose(perm)
            for name in ("add", "sub", "div"):
                hop, oop = OPS[name]
                got = gpu_binary(hop, Dev(x, code), Dev(yt, code, base=yb)).get()
                assert np.array_equal(bits(got), bits(O.binary(oop, x, np.ascontiguousarray(yt), a_code=code, b_code=code))), (code, shape, perm, name, "x op yT")
                got = gpu_binary(hop, Dev(yt, code, base=yb), Dev(x, code)).get()
                assert np.array_equal(bits(got), bits(O.binary(oop, np.ascontiguousarray(yt), x, a_code=code, b_code=code))), (code, shape, perm, name, "yT op x")


def test_in_place_with_a_transposed_operand_and_unaligned_row_slices():
    """x += y.T (the output IS the straight operand) through the tiled kernel; row slices that start at an odd element keep the 16-byte
    kernels (round 5) - all bit-exact, bytes outside the slices untouched."""
    rng = np.random.default_rng(82)
    for code in (H.F32, H.BF16):
        yb = rand_of(rng, (256, 192), code)
        x = rand_of(rng, (192, 256), code)
        dx = Dev(x, code)
        got = gpu_binary(OPS["add"][0], dx, Dev(yb.T, code, base=yb), out=dx).get()
        assert np.array_equal(bits(got), bits(O.binary(OPS["add"][1], x, np.ascontiguousarray(yb.T), a_code=code, b_code=code))), code
        a, b = rand_of(rng, (64, 530), code), rand_of(rng, (64, 530), code)
        out_base = rand_of(rng, (64, 530), code)
        va, vb, vo = a[:, 1:513], b[:, 3:515], out_base[:, 5:517]
        want = out_base.copy()
        dout = Dev(vo, code, base=out_base)
        got = gpu_binary(OPS["add"][0], Dev(va, code, base=a), Dev(vb, code, base=b), out=dout)
        want[:, 5:517] = O.binary(OPS["add"][1], np.ascontiguousarray(va), np.ascontiguousarray(vb), a_code=code, b_code=code)
        host = got.buf.to_numpy(out_base.shape, out_base.dtype)
        assert np.array_equal(bits(host), bits(want)), code


def test_vectorised_broadcast_paths():
    rng = np.random.default_rng(8)
    for code in (H.F32, H.BF16, H.F64):
        for sa, sb in (((128, 256), (1, 256)), ((128, 256), (128, 1)), ((4, 64, 128), (4, 1, 128)), ((4, 64, 128), (1, 64, 1)),
                       ((2, 1024, 64, 32), (2, 1024, 1, 32))):
            a, b = rand_of(rng, sa, code), rand_of(rng, sb, code)
            for hop, oop in (OPS["add"], OPS["div"]):
                got = gpu_binary(hop, Dev(a, code), Dev(b, code)).get()
                assert np.array_equal(bits(got), bits(O.binary(oop, a, b, a_code=code, b_code=code))), (code, sa, sb)


def test_copy_views_bit_exact():
    s = golden("shape_ops")
    x = s["perm_x"]
    src = Dev(x.transpose(2, 1, 0, 3), base=x)
    assert np.array_equal(gpu_copy(src, Dev.empty(src.arr.shape, H.F64)).get(), s["perm_out"])
    x = s["slice_x"]
    v = Dev(x[3, 3:8, 4:11:2], base=x)
    assert np.array_equal(gpu_copy(v, Dev.empty(v.arr.shape, H.F32)).get(), s["slice_out"])
    # cat = narrow().copy_() per input (tensor_shape.cpp:41-70): copy INTO a strided destination
    out_host = np.zeros(s["cat_out"].shape, dtype=np.float32)
    off = 0
    out_dev = Dev(out_host)
    for k in "abc":
        t = s[f"cat_{k}"]
        dst = Dev.__new__(Dev)
        dst.base, dst.arr, dst.code, dst.buf = out_dev.base, out_dev.base[:, off:off + t.shape[1], :], H.F32, out_dev.buf
        dst.view = H.View.of(out_dev.buf, dst.arr, H.F32, byte_offset=off * 23 * 4)
        gpu_copy(Dev(t), dst)
        off += t.shape[1]
    assert np.array_equal(out_dev.get(), s["cat_out"])
    x, off = s["split_x"], 0
    for i, n in enumerate((11, 13, 1)):
        v = Dev(x[:, off:off + n, :], base=x)
        assert np.array_equal(gpu_copy(v, Dev.empty(v.arr.shape, H.F32)).get(), s[f"split_{i}"])
        off += n
    # every element size, permuted + sliced, including NaN payload preservation
    rng = np.random.default_rng(9)
    for code in ALL:
        x = rand_of(rng, (6, 10, 12), code)
        v = x.transpose(2, 0, 1)[1:11:3]
        got = gpu_copy(Dev(v, code, base=x), Dev.empty(v.shape, code)).get()
        assert np.array_equal(bits(got), bits(np.ascontiguousarray(v))), code
    nan = np.array([0x7FC00001, 0xFFC12345, 0x7F800001, 0x00000001], dtype=np.uint32).view(np.float32)
    assert np.array_equal(bits(gpu_copy(Dev(nan), Dev.empty((4,), H.F32)).get()), bits(nan))


def test_convert_matrix_and_fill():
    rng = np.random.default_rng(10)
    for cs, shape in [(c, (33, 7)) for c in ALL] + [(c, (37, 8, 3)) for c in ALL]:  # (the second shape: whole groups of eight, the wide kernel)
        x0 = rand_of(rng, shape, cs)
        for cd in ALL:
            x = x0
            if cd == H.U8 and cs in (H.F16, H.F32, H.F64):
                x = np.abs(x0)  # negative float -> unsigned is UB in C++ (and in the reference)
            elif cd == H.U8 and cs == H.BF16:
                x = x0 & np.uint16(0x7FFF)
            got = gpu_copy(Dev(x, cs), Dev.empty(x.shape, cd)).get()
            want = O.convert(x, cd, src_code=cs)
            assert np.array_equal(bits(got), bits(want)), (cs, cd)
    for code in ALL:
        for val in (0.0, 1.0, -2.7, 3.999, 1e-3):
            if code in (H.U8, H.BOOL) and val < 0:
                continue
            for shape in ((5,), (1024,), (3, 5, 7)):
                got = gpu_fill(Dev.empty(shape, code), val).get()
                want = O.fill(np.empty(shape, dtype=H.CODE2NP[code]), val, dst_code=code)
                assert np.array_equal(bits(got), bits(want)), (code, val)
    # fill a strided view leaves the rest untouched
    base = np.arange(60, dtype=np.float32).reshape(6, 10)
    d = Dev(base[1:5:2, 2:9:3], base=base)
    gpu_fill(d, -1)
    want = base.copy()
    want[1:5:2, 2:9:3] = -1
    assert np.array_equal(d.buf.to_numpy(base.shape, base.dtype), want)


def test_empty_and_error_codes():
    d = H.make_desc([H.View(0x1000, (0,), (1,), H.F32)], [H.View(0x1000, (0,), (1,), H.F32), H.View(0x1000, (0,), (1,), H.F32)])
    H.elementwise(H.EW_ADD, d, H.F32)  # numel 0: no-op, no error
    a = Dev(np.zeros(4, dtype=np.float32))
    d = H.make_desc([a.view], [a.view, a.view])
    with pytest.raises(H.KfError) as e:
        H.elementwise(99, d, H.F32)
    assert e.value.code == H.KF_ERR_INVALID
    # a strided descriptor whose byte extent exceeds int32 must be refused, not mis-indexed
    big = H.View(a.buf.ptr, (3, 1 << 20), (1 << 30, 1), H.F32)
    small = H.View(a.buf.ptr, (3, 1 << 20), (0, 1), H.F32)
    d = H.make_desc([big], [small, small])
    with pytest.raises(H.KfError) as e:
        H.elementwise(H.EW_ADD, d, H.F32)
    assert e.value.code == H.KF_ERR_INDEX_RANGE


@pytest.mark.slow
def test_contiguous_beyond_int32_bytes():
    """The reference's 'hard' shape [2,1024,1024,512] (test_tensor.py:91-92): 2^30 fp32 elements, 4 GiB per
    operand. Built on-device with fill; checked by sampled windows, one straddling the 2^31-byte boundary."""
    n = 2 * 1024 * 1024 * 512
    a, b, c = H.DevBuf(4 * n), H.DevBuf(4 * n), H.DevBuf(4 * n)
    va, vb, vc = (H.View(x.ptr, (n,), (1,), H.F32) for x in (a, b, c))
    H.elementwise(H.EW_FILL, H.make_desc([va], []), 0, 1.25)
    H.elementwise(H.EW_FILL, H.make_desc([vb], []), 0, -3.5)
    # overwrite windows of b with a ramp so every sampled element is distinct
    ramp = np.arange(1 << 16, dtype=np.float32)
    offs = [0, (1 << 29) - (1 << 15), n - (1 << 16)]
    for o in offs:
        H.check(H.lib().kf_memcpy_h2d(b.ptr + 4 * o, ramp.ctypes.data, ramp.nbytes, None))
    H.elementwise(H.EW_ADD, H.make_desc([vc], [va, vb]), H.F32)
    H.device_sync()
    for o in offs:
        got = np.empty(1 << 16, dtype=np.float32)
        H.check(H.lib().kf_memcpy_d2h(got.ctypes.data, c.ptr + 4 * o, got.nbytes, None))
        assert np.array_equal(got, ramp + np.float32(1.25)), o
    got = np.empty(1024, dtype=np.float32)
    H.check(H.lib().kf_memcpy_d2h(got.ctypes.data, c.ptr + 4 * (1 << 28), got.nbytes, None))
    assert np.array_equal(got, np.full(1024, -2.25, dtype=np.float32))


def test_tiled_transpose_and_vector_convert_paths():
    """permute(...).contiguous() through the LDS-tiled transpose kernel (ragged tiles, batch dims, every element
    size) and the 8-wide float-family convert kernel — both must stay bit-exact."""
    rng = np.random.default_rng(11)
    for code in (H.U8, H.I16, H.F32, H.F64, H.BF16):
        for shape, perm in (((70, 130), (1, 0)), ((3, 65, 129), (0, 2, 1)), ((5, 33, 4, 70), (3, 1, 2, 0)), ((2, 256, 64), (2, 0, 1)),
                            ((17, 16), (1, 0)), ((128, 192), (1, 0)), ((3, 128, 64), (0, 2, 1)), ((2, 3, 64, 128), (1, 0, 3, 2))):  # whole tiles: 16-byte form
            x = rand_of(rng, shape, code)
            v = x.transpose(perm)
            got = gpu_copy(Dev(v, code, base=x), Dev.empty(v.shape, code)).get()
            assert np.array_equal(bits(got), bits(np.ascontiguousarray(v))), (code, shape, perm)
    x = rand_of(rng, (4096, 4096), H.F32)  # multi-tile, exact
    got = gpu_copy(Dev(x.T, base=x), Dev.empty((4096, 4096), H.F32)).get()
    assert np.array_equal(got, x.T)
    for code, shape, perm in ((H.BF16, (2048, 2048), (1, 0)), (H.F32, (2, 1024, 512), (0, 2, 1)), (H.BF16, (2, 2048, 1024), (0, 2, 1)), (H.I16, (1024, 4096), (1, 0))):
        x = rand_of(rng, shape, code)  # tile counts that take the XCD super-tile order (4 x 4 tiles of 4-byte, 8 x 8 of 2-byte elements), with a batch dim too
        v = x.transpose(perm)
        got = gpu_copy(Dev(v, code, base=x), Dev.empty(v.shape, code)).get()
        assert np.array_equal(bits(got), bits(np.ascontiguousarray(v))), (code, shape, perm)
    for cs in (H.F32, H.F16, H.BF16):
        for n in (8, 1 << 16, (1 << 16) + 8):
            x = rand_of(rng, (n,), cs)
            for cd in (H.F32, H.F16, H.BF16):
                got = gpu_copy(Dev(x, cs), Dev.empty((n,), cd)).get()
                assert np.array_equal(bits(got), bits(O.convert(x, cd, src_code=cs))), (cs, cd, n)


@pytest.mark.parametrize("code", [H.F32, H.F64, H.BF16, H.F16, H.I32, H.I64])
def test_scalar_operand_equals_fill_then_op(code):
    """`tensor (op) python-float` (register.cpp:172-206): the reference fills a same-dtype tensor with the scalar and runs
    the binary kernel; KF_EW_*_SCALAR must give the same bits without the temporary — contiguous, ragged, strided, in place."""
    rng = np.random.default_rng(300 + code)
    scal = {"add": H.EW_ADD_SCALAR, "sub": H.EW_SUB_SCALAR, "mul": H.EW_MUL_SCALAR, "div": H.EW_DIV_SCALAR}
    for shape in ((1,), (1000,), (64, 257), (7, 33, 64)):
        a = rand_of(rng, shape, code)
        for s in (2.0, -3.0, 0.3, 1e-3, 7.75):
            if code in (H.I32, H.I64) and int(s) == 0:
                continue  # x / 0
            filled = O.fill(np.empty(shape, dtype=a.dtype), s, dst_code=code)
            for name, (_, oop) in OPS.items():
                want = O.binary(oop, a, filled, a_code=code, b_code=code)
                da, out = Dev(a, code), Dev.empty(shape, code)
                H.elementwise(scal[name], H.make_desc([out.view], [da.view]), 0, s)
                H.device_sync()
                assert np.array_equal(bits(out.get()), bits(want)), (code, shape, s, name)
                H.elementwise(scal[name], H.make_desc([da.view], [da.view]), 0, s)  # in place
                H.device_sync()
                assert np.array_equal(bits(da.get()), bits(want)), (code, shape, s, name, "in place")
    base = rand_of(rng, (12, 20, 36), code)
    for view in (base.transpose(2, 0, 1), base[:, ::2, 1:30:3]):
        filled = O.fill(np.empty(view.shape, dtype=base.dtype), 1.5, dst_code=code)
        want = O.binary(O.MUL, np.ascontiguousarray(view), filled, a_code=code, b_code=code)
        out = Dev.empty(view.shape, code)
        H.elementwise(H.EW_MUL_SCALAR, H.make_desc([out.view], [Dev(view, code, base=base).view]), 0, 1.5)
        H.device_sync()
        assert np.array_equal(bits(out.get()), bits(want)), (code, view.shape)
    with pytest.raises(H.KfError) as e:  # not covered: the host broadcasts a 1-element tensor instead
        x = Dev(np.zeros(8, dtype=np.int8))
        H.elementwise(H.EW_ADD_SCALAR, H.make_desc([x.view], [x.view]), 0, 1.0)
    assert e.value.code == H.KF_ERR_UNSUPPORTED
