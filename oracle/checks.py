"""Scale-aware acceptance bounds for the 16-bit attention kernels. TEST INFRASTRUCTURE (tests/, __graft_entry__.smoke(), bench.py's
spot check) - never the product path.

Why not rtol / atol: at S = 4096 the median |dQ| on U(-1, 1) operands is 0.003 and the median |O| 0.01; an absolute tolerance of
2e-2 / 3e-2 accepted an all-zero gradient in 99 % of its entries. The bounds here are relative to what each output
element is a SUM OF. oracle.attn_ref64 returns, per element, the double-precision value `ref`, the worst-case error scale `mag`
(sum of |terms|: every rounding pushing the same way) and the statistical one `quad` (the terms in quadrature: independent roundings
add like a random walk; a sum of one term has quad = mag, a long sum has quad close to |ref| itself):

  element   |got - ref| <= eps (C_OUT |ref| + C_SUM mag + coh)                       one output rounding + the worst case of the inner ones
  row       ||got - ref||_2 <= eps (C_ROW ||ref||_2 + C_Q ||quad||_2)                   over each row's D entries
  head      ||got - ref||_F <= eps (C_HEAD ||ref||_F + C_QH ||quad||_F)                 per (batch, head)

eps = 2^-8 (bf16: 8 significant bits, half an ulp) or 2^-11 (f16). The element bound is a worst case and cannot see a missing tile in
a long row; the row and head bounds can: rounding errors are incoherent, so a row's error norm sits near 0.6 eps ||ref||, while a
dropped 64-key tile at S = 4096 is 0.1-0.5 ||ref||. `coh` (dQ only, the oracle's bdq): the backward forms delta = rowsum(dO o O) from
the ROUNDED O, an error of up to eps sum |dO O| that moves every dS of a query row the same way; its worst case through dQ = scale dS K.
`floor` (round 5; format_floor below): what the 16-bit FORMAT itself cannot hold. A relative bound presumes the intermediates P and dS are
rounded with RELATIVE precision eps; below the element type's smallest normal number (f16: 2^-14 = 6.1e-5; bf16: 2^-126) a stored value
has ABSOLUTE spacing instead (f16: 2^-24) or is flushed to zero. With the reference tests' own inputs, U(-10, 10) (test/test_nn.py:22-24:
logit std 33, the softmax one-hot), most entries of P lie far below either limit and whole key columns of dV / dK are sums of such terms:
the double-precision oracle returns 1e-50 there, no 16-bit kernel can. The floor adds, per output element, one absolute rounding of every
term's intermediate times the other factor: dV_j += u_P sum_i |dO_i|, dK_j += u_dS scale sum_i |Q_i|, dQ_i += u_dS scale sum_j |K_j|
(sums over the causally visible partners; u = half the subnormal spacing for f16 - 2^-25, and 2^-25 2^-14 for P, which the backward kernels
carry as P 2^14 - and the flush limit 2^-126 for bf16; plus, for both types, the entries of P that f32's exp2 itself returns as 0: 2^-126 times
the |dP - delta| they are multiplied by). Relative to the outputs themselves these floors are 1e-5 and less on every tested
input: a dropped tile, a wrong mask row or a zeroed slice is as visible as before (tests/test_attention_bounds.py).
The constants are >= 2x the largest figures measured on MI355X over the shapes of tools/attn_parity_margins.py
(profiles/r03_attn_parity_margins.json); tests/test_attention_bounds.py shows on CPU that the same bounds reject a dropped tile, a
zeroed slice, a wrong mask row and an all-zero gradient, tests/test_gpu_attention_mutants.py does so on the kernels themselves.
"""
import os

import numpy as np

from . import oracle as O

EPS = {O.BF16: 2.0 ** -8, O.F16: 2.0 ** -11}
C_OUT, C_SUM, C_ROW, C_Q, C_HEAD, C_QH = 2.0, 2.0, 2.5, 2.5, 1.25, 1.25
ABS_ULP = {O.BF16: 2.0 ** -126, O.F16: 2.0 ** -24}  # f16 outputs below 2^-14 are subnormal: one absolute ulp there, not a relative one; bf16 outputs below 2^-126 may be flushed
P_SHIFT_F16 = 14            # the f16 backward kernels carry P as P 2^14 into the dV product (tools/gen_attn_dkv.py P_SHIFT, attention.hip kPShiftF16)
C_FLOOR = 2.0
NAMES = ("o", "dq", "dk", "dv")


def to_f64(x, code):
    return O.to_float(x, code).astype(np.float64)


def gemm_ok(got16, a16, b16, code, trans_a=False, trans_b=False):
    """THE acceptance bound of a 16-bit GEMM output (tests/test_gpu_gemm.py and bench.py's spot check share it): against the
    double-precision product of the 16-bit operands, |got - c| <= eps |c| + 1e-6 sum_k |a||b| + half a subnormal ulp - one rounding of the result to the
    element type (eps = half an ulp) plus f32 accumulation noise relative to the sum of the terms' magnitudes. Returns (ok, worst
    fraction of the bound)."""
    a, b = to_f64(a16, code), to_f64(b16, code)
    a = a.T if trans_a else a
    b = b.T if trans_b else b
    want, mag = a @ b, np.abs(a) @ np.abs(b)
    # (+ half an absolute ulp of the format's subnormal grid: at K = 1 a product of two small f16 operands has nothing else to hide behind)
    frac = np.abs(to_f64(got16, code) - want) / (EPS[code] * np.abs(want) + 1e-6 * mag + 0.5 * ABS_ULP[code])
    return bool((frac <= 1.0).all()), float(frac.max())


def format_floor(q, k, v, d_o, code, scale=None):
    """{name: per-element absolute floor} for dq, dk, dv (see the module docstring): what one absolute rounding (f16: half the subnormal
    spacing; bf16: the flush-to-zero limit) of every P and dS entry can move the gradients by. Causal, top-left aligned like the
    reference's mask (causal_attention_ref.h:40-48: query i sees keys j <= i)."""
    qa, ka, da = (np.abs(to_f64(x, code)) for x in (q, k, d_o))
    Sq, Skv, D = qa.shape[2], ka.shape[2], qa.shape[3]
    scale = 1.0 / np.sqrt(D) if scale is None else scale
    if code == O.F16:
        u_p, u_ds = 2.0 ** -25 * 2.0 ** -P_SHIFT_F16, 2.0 ** -25
    else:
        u_p = u_ds = 2.0 ** -126
    n = min(Sq, Skv)

    def suffix_over_queries(x):   # key j <- sum over queries i >= j (none for a key beyond the last query)
        out = np.zeros(x.shape[:2] + (Skv, D))
        out[:, :, :n] = np.cumsum(x[:, :, ::-1], axis=2)[:, :, ::-1][:, :, :n]
        return out

    def prefix_over_keys(x):      # query i <- sum over keys j <= i (all keys for a query beyond the last key)
        c = np.cumsum(x, axis=2)
        return c[:, :, np.minimum(np.arange(Sq), Skv - 1)]

    # ... and f32 itself: exp2 returns 0 below 2^-126, so an entry of P that small takes its whole dS = P (dP - delta) with it, whatever
    # |dP - delta| <= 2 ||dO_i|| ||V_j|| is (U(-10, 10) at D = 64: P 1e-39 times 4000 is a dK entry of 6e-37 - representable in bf16, and 0 in
    # every kernel that forms P in f32). Both element types.
    p32 = 2.0 ** -126
    a_i = 2.0 * np.linalg.norm(to_f64(d_o, code), axis=-1, keepdims=True)       # [B, H, Sq, 1]
    vn_j = np.linalg.norm(to_f64(v, code), axis=-1, keepdims=True)              # [B, H, Skv, 1]
    return {"dv": C_FLOOR * u_p * suffix_over_queries(da),
            "dk": C_FLOOR * scale * (u_ds * suffix_over_queries(qa) + p32 * vn_j * suffix_over_queries(a_i * qa)),
            "dq": C_FLOOR * scale * (u_ds * prefix_over_keys(ka) + p32 * a_i * prefix_over_keys(vn_j * ka))}


def margins(got, ref, mag, quad, eps, ulp=0.0, coh=None, floor=None):
    """The three normalised error figures of one output (each must be <= 1 for the bound to hold): worst element, worst row, worst head;
    and the worst row-relative L2 error over the LATER HALF of the rows (the long rows, where a missing tile is smallest)."""
    err = np.abs(got - ref)
    tiny = np.finfo(np.float64).tiny
    coh = np.zeros_like(ref) if coh is None else coh
    floor = np.zeros_like(ref) if floor is None else floor
    el = err / (eps * (C_OUT * np.abs(ref) + C_SUM * mag + coh) + floor + ulp + tiny)
    nerr, nref = np.linalg.norm(got - ref, axis=-1), np.linalg.norm(ref, axis=-1)
    rn = nerr / (eps * (C_ROW * nref + C_Q * np.linalg.norm(quad, axis=-1)) + np.linalg.norm(floor, axis=-1) + ulp * np.sqrt(got.shape[-1]) + tiny)
    fro = lambda x: np.linalg.norm(x.reshape(*x.shape[:2], -1), axis=-1)  # noqa: E731
    hd = fro(got - ref) / (eps * (C_HEAD * fro(ref) + C_QH * fro(quad)) + fro(floor) + ulp * np.sqrt(got[0, 0].size) + tiny)
    half = got.shape[2] // 2
    return {"element": float(el.max()), "row": float(rn.max()), "head": float(hd.max()),
            "row_rel_l2": float((nerr[:, :, half:] / (nref[:, :, half:] + tiny)).max())}


def scales(ref, name):
    """(ref, mag, quad, coh) of one output out of an oracle.attn_ref64 result."""
    return ref[name], ref["m" + name], ref["q" + name], ref["bdq"] if name == "dq" else None


def check_one(name, got16, ref, code, what="", rows=None, floor=None):
    """Assert one 16-bit output against the oracle.attn_ref64 result `ref` (rows: a slice of the S axis both are restricted to; floor:
    this output's entry of format_floor); returns the margins."""
    r, mag, quad, coh = scales(ref, name)
    if rows is not None:
        r, mag, quad, coh = r[:, :, rows], mag[:, :, rows], quad[:, :, rows], None if coh is None else coh[:, :, rows]
        floor = None if floor is None else floor[:, :, rows]
    got = to_f64(got16, code)
    assert got.shape == r.shape, (what, name, got.shape, r.shape)
    assert np.isfinite(got).all(), f"{what} {name}: non-finite values"
    m = margins(got, r, mag, quad, EPS[code], ABS_ULP[code], coh, floor)
    bad = [k for k in ("element", "row", "head") if not m[k] <= 1.0]
    if bad:
        raise AssertionError(f"{what} {name}: outside the scale-aware bound ({', '.join(f'{k} {m[k]:.2f}x' for k in bad)}; "
                             f"worst row relative L2 error {m['row_rel_l2']:.3e}, eps {EPS[code]:.2e})")
    return m


def lse_scaled_query_bound(q, k, code):
    """Worst-case |delta lse| when the kernel rounds c * q (c = scale * log2 e) to the element type ONCE per query block and lets
    the score MFMAs deliver exponents (the opt-in scaled-operand form of the round-4 forward, KF_ATTN_SCALED_OPERANDS): every product q_i k_i of a score moves by at most
    eps |q_i k_i|, a score by eps * scale * sum_i |q_i k_i|, and lse - a p-weighted mean of the scores - by no more than the
    worst visible score of its row. [B, H, Sq] (natural-log units, like lse)."""
    qa, ka = np.abs(to_f64(q, code)), np.abs(to_f64(k, code))
    B, Hh, Sq, D = qa.shape
    Skv = ka.shape[2]
    vis = np.arange(Skv)[None, :] <= np.arange(Sq)[:, None]          # the reference's mask: row >= col, absolute (causal_attention_ref.h:40-48)
    out = np.empty((B, Hh, Sq))
    for b in range(B):
        for h in range(Hh):
            w = qa[b, h] @ ka[b, h].T
            out[b, h] = np.where(vis, w, 0.0).max(axis=1)
    return EPS[code] * out / np.sqrt(D)


def attn_check(q, k, v, code, o=None, lse=None, d_o=None, dq=None, dk=None, dv=None, what="", ref=None, lse_tol=None, scaled_query=None):
    """Check whatever outputs are given (16-bit arrays [B,H,S,D]; lse f32 [B,H,Sq]) of causal attention on the 16-bit inputs q, k, v
    (and d_o for the gradients). Returns {name: margins}. `ref` = a precomputed oracle.attn_ref64 result. `scaled_query`: the forward
    that produced `lse` is the one that rounds c * q to the element type (None: KF_ATTN_SCALED_OPERANDS is set and the shape is one
    kf_attn_fwd gives that stream - D = 128, Sq % 256 == 0, Skv >= Sq)."""
    if ref is None:
        ref = O.attn_ref64(q, k, v, d_o if (dq is not None or dk is not None or dv is not None) else None, code=code)
    out = {}
    floors = format_floor(q, k, v, d_o, code) if d_o is not None else {}
    for name, got in (("o", o), ("dq", dq), ("dk", dk), ("dv", dv)):
        if got is not None:
            out[name] = check_one(name, got, ref, code, what, floor=floors.get(name))
    if lse is not None:
        # LSE is f32 out. The kernels that keep q as it is: the f32 score chain's error + the exp2 / log arithmetic, 2e-6 (1 + |lse|)
        # (measured 1.7e-7; the default). The opt-in forward that scales q in 16 bits: + the rigorous bound of that one rounding per row (measured on
        # MI355X: <= 0.1 of it on U(-1, 1) operands, where the roundings of a score's 128 products average out, up to 0.5 on
        # spiked operands where one product carries the score; `fraction_of_bound` in the returned record).
        tol = (lse_tol if lse_tol is not None else 2e-6) * (1.0 + np.abs(ref["lse"]))
        d = np.abs(lse.astype(np.float64) - ref["lse"])
        if scaled_query is None:   # the opt-in form of the generated forward (KF_ATTN_SCALED_OPERANDS), on the shapes kf_attn_fwd gives it
            scaled_query = bool(os.environ.get("KF_ATTN_SCALED_OPERANDS")) and q.shape[-1] == 128 and q.shape[2] % 256 == 0 and k.shape[2] >= q.shape[2]
        if scaled_query and lse_tol is None:
            tol = tol + lse_scaled_query_bound(q, k, code)
        assert np.isfinite(lse).all() and (d <= tol).all(), f"{what} lse: max error {d.max():.3e}"
        out["lse"] = {"max_abs": float(d.max()), "fraction_of_bound": float((d / (tol + np.finfo(np.float64).tiny)).max())}
    return out
