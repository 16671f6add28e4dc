#!/usr/bin/env python3
"""Randomised stress of kf_attn_fwd_strided / kf_attn_bwd_strided (round 6): every operand in its own random layout - a [B, S, H, D]-ordered buffer (head stride D, row stride
H D + pad) or a [B, H, S, D]-ordered one with padded rows and a padded head stride, outputs into buffers prefilled with a pattern - on random ragged shapes (Skv >= Sq, or whole
128-row tiles), both head sizes, both 16-bit types. The layout changes addresses, not arithmetic: every output must be BIT-identical to the contiguous entries' on the same values, every
byte between and behind the strided outputs untouched. stress_attn_strided.py SEED SECONDS"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from kfunca_amd import hip_abi as H
from oracle import oracle as O
from tests.test_gpu_attention import fwd, bwd
H.set_device(0)
seed, secs = int(sys.argv[1]), float(sys.argv[2])
rng = np.random.default_rng(seed)
t_end = time.time() + secs
FILL = 0x7A7A

def place(x, pattern=None, like=None):
    """x [B, H, S, D] -> (host buffer of uint16, element offset, (batch, head, row) strides) in a random layout; pattern: prefill instead of x's values (an output);
    like = (pad, order): the row padding and dim order of another operand (K and V, dK and dV must share a row stride: kfunca_hip.h)."""
    B, Hh, S, D = x.shape
    pad = int(rng.choice([0, 0, 8, 24, 64])) if like is None else like[0]
    lead = int(rng.choice([0, 8, 128]))
    bshd = (rng.random() < 0.5) if like is None else like[1]
    place.last = (pad, bshd)
    if bshd:      # [B, S, H, D] order
        row = Hh * D + pad
        lay = (S * row + int(rng.choice([0, 8 * row])), D, row)
    else:                       # [B, H, S, D] order, padded rows / heads
        row = D + pad
        head = S * row + int(rng.choice([0, 8, 8 * row]))
        lay = (Hh * head + int(rng.choice([0, 16])), head, row)
    n = lead + (B - 1) * lay[0] + (Hh - 1) * lay[1] + (S - 1) * lay[2] + D + 64
    buf = np.full(n, FILL, dtype=np.uint16)
    idx = (lead + np.arange(B)[:, None, None, None] * lay[0] + np.arange(Hh)[None, :, None, None] * lay[1] + np.arange(S)[None, None, :, None] * lay[2]
           + np.arange(D)[None, None, None, :])
    if pattern is None:
        buf[idx] = x.view(np.uint16)
    return buf, lead, lay, idx

n = 0
while time.time() < t_end:
    code = int(rng.choice([H.BF16, H.F16]))
    D = int(rng.choice([64, 128]))
    B, Hh = int(rng.integers(1, 3)), int(rng.integers(1, 5))
    if rng.random() < 0.7:
        Sq = int(rng.choice([1, 31, 64, 100, 128, 200, 256, 257, 500, 640, 1000]))
        Skv = Sq + int(rng.choice([0, 0, 1, 28, 128, 300]))
    else:
        Sq, Skv = int(rng.choice([128, 256, 512])), int(rng.choice([128, 256]))   # whole tiles, fewer keys than queries allowed
    q, k, v, go = (O.from_float(rng.uniform(-1, 1, s).astype(np.float32), code) for s in ((B, Hh, Sq, D), (B, Hh, Skv, D), (B, Hh, Skv, D), (B, Hh, Sq, D)))
    o_ref, lse_ref = fwd(code, q, k, v)
    dq_ref, dk_ref, dv_ref = bwd(code, q, k, v, o_ref, lse_ref, go)
    scale = 1.0 / np.sqrt(D)
    ins = {"q": place(q), "go": place(go), "k": place(k)}
    ins["v"] = place(v, like=place.last)
    outs = {"o": place(q, pattern=True), "dq": place(q, pattern=True), "dk": place(k, pattern=True)}
    outs["dv"] = place(v, pattern=True, like=place.last)
    dev = {name: H.DevBuf.from_numpy(t[0]) for name, t in {**ins, **outs}.items()}
    P = lambda name: dev[name].ptr + 2 * ({**ins, **outs}[name][1])  # noqa: E731
    L = lambda name: {**ins, **outs}[name][2]                        # noqa: E731
    blse = H.DevBuf.from_numpy(np.full(B * Hh * Sq + 16, 7.0, np.float32))
    tag = (code, B, Hh, Sq, Skv, D, {n_: L(n_) for n_ in ("q", "k", "v", "o", "go", "dq", "dk", "dv")})
    H.attn_fwd_strided(code, B, Hh, Sq, Skv, D, scale, P("q"), L("q"), P("k"), L("k"), P("v"), L("v"), P("o"), L("o"), blse.ptr)
    H.device_sync()
    got_o = dev["o"].to_numpy((len(outs["o"][0]),), np.uint16)
    assert np.array_equal(got_o[outs["o"][3]], o_ref.view(np.uint16)), ("o",) + tag
    mask = np.ones(len(got_o), bool); mask[outs["o"][3].reshape(-1)] = False
    assert (got_o[mask] == FILL).all(), ("o gaps",) + tag
    lse = blse.to_numpy((B * Hh * Sq + 16,), np.float32)
    assert np.array_equal(lse[:B * Hh * Sq].view(np.uint32), lse_ref.reshape(-1).view(np.uint32)) and (lse[B * Hh * Sq:] == 7.0).all(), ("lse",) + tag
    need = H.attn_bwd_workspace_bytes(code, B, Hh, Sq, Skv, D)
    ws = H.DevBuf(need + 256)
    H.attn_bwd_strided(code, B, Hh, Sq, Skv, D, scale, P("q"), L("q"), P("k"), L("k"), P("v"), L("v"), P("o"), L("o"), blse.ptr, P("go"), L("go"),
                       P("dq"), L("dq"), P("dk"), L("dk"), P("dv"), L("dv"), ws.ptr, need)
    H.device_sync()
    for name, ref in (("dq", dq_ref), ("dk", dk_ref), ("dv", dv_ref)):
        got = dev[name].to_numpy((len(outs[name][0]),), np.uint16)
        assert np.array_equal(got[outs[name][3]], ref.view(np.uint16)), (name,) + tag
        mask = np.ones(len(got), bool); mask[outs[name][3].reshape(-1)] = False
        assert (got[mask] == FILL).all(), (name + " gaps",) + tag
    for b in dev.values(): b.free()
    blse.free(); ws.free()
    n += 1
print(f"seed {seed}: {n} strided attention cases (forward + backward): every output bit-identical to the contiguous entries', every byte between and behind the outputs untouched")
