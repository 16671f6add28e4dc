"""CPU restatement of BASELINE config C5's transformer block (forward + backward) for one batch shard. TEST INFRASTRUCTURE ONLY
(tests/test_gpu_block.py, tools/block_bench.py --check).

The block is built from operators the reference API has: gemm (QKV, output projection, gated-MLP up / gate /
down: gemm_ops.cpp:6-16), causal_attention (nn_ops.cpp:6-8, semantics causal_attention_ref.h:25-64), add (residuals) and mul (the
gate), binary_ops.cpp:6-91 - optionally with the two rms_norms of a pre-norm block (README.md:28 roadmap):

    n1 = rms_norm(x, g1) | x          qkv = n1 Wqkv        a = attention(qkv)         h = a Wo + x
    n2 = rms_norm(h, g2) | h          up = n2 Wup          t = (n2 Wgate) o up        y = t Wdown + h

Every tensor an operator of the device path STORES is 16-bit, so the restatement rounds the same tensors to bf16 (round-to-nearest-even,
half.h:195-208) and evaluates everything between two roundings in float32 BLAS / the attention oracle: what is left between this and
the device result is accumulation order and the occasional rounding flip of an intermediate, i.e. a few bf16 ulps of each output's
scale - the bound the callers hold it to is relative to each tensor's norm, and stated there."""
import numpy as np

from . import oracle as O


def r16(x):
    """float32 -> bf16 -> float32 (round-to-nearest-even)."""
    u = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32)
    return (((u + 0x7FFF + ((u >> 16) & 1)) >> 16) << 16).astype(np.uint32).view(np.float32)


def _heads(x2, B, S, H, D):  # [B S, H D] -> [B, H, S, D]
    return np.ascontiguousarray(x2.reshape(B, S, H, D).transpose(0, 2, 1, 3))


def _flat(x4):  # [B, H, S, D] -> [B S, H D]
    B, H, S, D = x4.shape
    return np.ascontiguousarray(x4.transpose(0, 2, 1, 3).reshape(B * S, H * D))


def _rms_fwd(x, gain, eps):
    rstd = 1.0 / np.sqrt((x.astype(np.float64) ** 2).mean(axis=1, keepdims=True) + eps)
    return r16((x * rstd * gain).astype(np.float32)), rstd


def _rms_bwd(x, gain, rstd, dy):
    x64, dy64 = x.astype(np.float64), dy.astype(np.float64)
    gdy = dy64 * gain
    dx = rstd * (gdy - x64 * rstd * rstd * (gdy * x64).mean(axis=1, keepdims=True))
    dgain = (dy64 * x64 * rstd).sum(axis=0)
    return r16(dx.astype(np.float32)), r16(dgain.astype(np.float32))


def block_fwd_bwd(x, w, g, B, S, H, D, gains=None, eps=1e-5):
    """x [B S, d], w = [Wqkv [d, 3d], Wo [d, d], Wgate [d, f], Wup [d, f], Wdown [f, d]], g = dL/dy [B S, d]; all float32 arrays holding
    bf16 values. Returns (y, dx, [dWqkv, dWo, dWgate, dWup, dWdown], [dgain1, dgain2] | None) as float32 arrays of bf16 values."""
    d = H * D
    wq, wo, wg, wu, wd = w
    if gains is not None:
        n1, rstd1 = _rms_fwd(x, gains[0], eps)
    else:
        n1 = x
    qkv = r16(n1 @ wq)
    q, k, v = (O.f32_to_bf16(_heads(qkv[:, i * d:(i + 1) * d], B, S, H, D)) for i in range(3))
    a4, _ = O.attn_fwd(q, k, v, code=O.BF16)                      # f32 math on the bf16 inputs, one rounding of the output
    a = _flat(O.bf16_to_f32(a4))
    h = r16(a @ wo + x)
    if gains is not None:
        n2, rstd2 = _rms_fwd(h, gains[1], eps)
    else:
        n2 = h
    up = r16(n2 @ wu)
    raw = n2 @ wg
    t = r16(raw * up)
    raw = r16(raw)                                                # the gate's product is kept in 16 bits for its backward
    y = r16(t @ wd + h)
    # backward
    dt = r16(g @ wd.T)
    dwd = r16(t.T @ g)
    dup = r16(dt * raw)
    draw = r16(dt * up)
    dwg = r16(n2.T @ draw)
    dwu = r16(n2.T @ dup)
    dn2 = r16(r16(draw @ wg.T) + r16(dup @ wu.T))
    if gains is not None:
        dh_n, dg2 = _rms_bwd(h, gains[1], rstd2, dn2)
    else:
        dh_n, dg2 = dn2, None
    dh = r16(g + dh_n)
    da = r16(dh @ wo.T)
    dwo = r16(a.T @ dh)
    go4 = O.f32_to_bf16(_heads(da, B, S, H, D))
    dq4, dk4, dv4 = O.attn_bwd(q, k, v, go4, code=O.BF16)
    dqkv = np.concatenate([_flat(O.bf16_to_f32(t4)) for t4 in (dq4, dk4, dv4)], axis=1)
    dwq = r16(n1.T @ dqkv)
    dn1 = r16(dqkv @ wq.T)
    if gains is not None:
        dx_n, dg1 = _rms_bwd(x, gains[0], rstd1, dn1)
    else:
        dx_n, dg1 = dn1, None
    dx = r16(dh + dx_n)
    return y, dx, [dwq, dwo, dwg, dwu, dwd], (None if gains is None else [dg1, dg2])


def rel_fro(got, want):
    got, want = np.asarray(got, dtype=np.float64), np.asarray(want, dtype=np.float64)
    return float(np.linalg.norm(got - want) / (np.linalg.norm(want) + 1e-300))
