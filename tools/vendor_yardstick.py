#!/usr/bin/env python3
"""A same-box vendor yardstick: context for this repository's rates, NOT part of the product, of bench.py's timed region or of any test.

On the GPU box's own PyTorch (ROCm): torch.matmul in bf16 at 4096^3 and 8192^3 (hipBLASLt / rocBLAS behind it) and
torch.nn.functional.scaled_dot_product_attention(is_causal=True) forward + backward at config C3 (B 8, H 32, S 4096, D 128; whatever
flash-attention backend this torch build dispatches to), each timed with device events over ~2 s of back-to-back launches after ~1 s
of warm-up, beside the SAME shapes through this repository's C ABI in the same process, interleaved (vendor, ours, vendor, ours), with
board power and clock sampled during each run (tools/power_trace.py: amd-smi / rocm-smi). Uniform(-1, 1) operands everywhere.

    python tools/vendor_yardstick.py --json profiles/r06_vendor_yardstick.json

kfunca_amd/ never imports torch for compute and never imports this file."""
import argparse
import json
import sys
import threading
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
sys.path.insert(0, str(ROOT / "tools"))


class Sampler:
    """Board power / clock every ~100 ms while a run is in flight."""

    def __init__(self):
        from power_trace import sample
        self.sample, self.samples, self.stop = sample, [], threading.Event()

    def __enter__(self):
        def poll():
            while not self.stop.is_set():
                s = self.sample()
                s["t"] = time.time()
                self.samples.append(s)
                time.sleep(0.1)
        self.th = threading.Thread(target=poll, daemon=True)
        self.th.start()
        return self

    def __exit__(self, *a):
        self.stop.set()
        self.th.join()

    def summary(self, t0, t1):
        busy = [s for s in self.samples if t0 <= s["t"] <= t1]
        avg = lambda k: float(np.mean([s[k] for s in busy if k in s])) if any(k in s for s in busy) else None  # noqa: E731
        return {"power_w": avg("power_w"), "sclk_mhz": avg("sclk_mhz"), "samples": len(busy)}


def timed(fn, sync, make_event, elapsed_ms, warm_s, timed_s):
    """ms per call of fn(): ~warm_s untimed, then ~timed_s between two device events (no host sync inside)."""
    fn(); sync()
    t = time.time(); fn(); sync()
    per = max(time.time() - t, 1e-5)
    for _ in range(max(1, int(warm_s / per))):
        fn()
    sync()
    n = max(3, int(timed_s / per))
    t0 = time.time()
    e0, e1 = make_event(), make_event()
    e0.record()
    for _ in range(n):
        fn()
    e1.record()
    sync()
    return elapsed_ms(e0, e1) / n, n, t0 + 0.3, time.time()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--json", default="")
    ap.add_argument("--warm", type=float, default=1.0)
    ap.add_argument("--seconds", type=float, default=2.0)
    ap.add_argument("--rounds", type=int, default=2)
    args = ap.parse_args()
    import torch
    import torch.nn.functional as F

    from kfunca_amd import hip_abi as H
    H.set_device(0)
    dev = torch.device("cuda:0")
    t_ev = lambda: torch.cuda.Event(enable_timing=True)  # noqa: E731
    t_ms = lambda a, b: a.elapsed_time(b)  # noqa: E731

    class KEv:  # this repository's events on its own stream (None = the legacy default stream, where these launches go)
        def __init__(self):
            self.e = H.Event()

        def record(self):
            self.e.record(None)

    k_ms = lambda a, b: a.e.elapsed_ms(b.e)  # noqa: E731
    out = {"torch": torch.__version__, "hip": getattr(torch.version, "hip", None), "device": torch.cuda.get_device_name(0),
           "note": "ms per call, uniform(-1,1) bf16 operands, interleaved vendor / ours, device events over ~%.0f s after ~%.0f s warm" % (args.seconds, args.warm),
           "cases": []}
    g = torch.Generator(device=dev).manual_seed(0)

    def u(*shape):
        return (torch.rand(*shape, device=dev, generator=g, dtype=torch.float32) * 2 - 1).to(torch.bfloat16)

    def run_case(name, flops, vendor_fn, ours_fn, extra=None):
        case = {"case": name, "flop_per_call": flops, "vendor": [], "ours": []}
        with Sampler() as smp:
            for _ in range(args.rounds):
                for who, fn, sync, mk, el in (("vendor", vendor_fn, torch.cuda.synchronize, t_ev, t_ms), ("ours", ours_fn, H.device_sync, KEv, k_ms)):
                    ms, n, t0, t1 = timed(fn, sync, mk, el, args.warm, args.seconds)
                    case[who].append({"ms": ms, "tflops": flops / (ms * 1e-3) / 1e12, "calls": n, **smp.summary(t0, t1)})
        case["vendor_best_ms"] = min(r["ms"] for r in case["vendor"])
        case["ours_best_ms"] = min(r["ms"] for r in case["ours"])
        case["ours_over_vendor_time"] = case["ours_best_ms"] / case["vendor_best_ms"]
        if extra:
            case.update(extra)
        out["cases"].append(case)
        print(json.dumps({k: case[k] for k in ("case", "vendor_best_ms", "ours_best_ms", "ours_over_vendor_time")}), file=sys.stderr, flush=True)

    # ---- GEMM: C = A B, bf16, row-major, both through their own buffers
    for n in (4096, 8192):
        a, b = u(n, n), u(n, n)
        c = torch.empty(n, n, device=dev, dtype=torch.bfloat16)
        ka, kb, kc = H.DevBuf(2 * n * n), H.DevBuf(2 * n * n), H.DevBuf(2 * n * n)
        torch.cuda.synchronize()
        H.check(H.lib().kf_memcpy_d2d(ka.ptr, a.data_ptr(), 2 * n * n, None))
        H.check(H.lib().kf_memcpy_d2d(kb.ptr, b.data_ptr(), 2 * n * n, None))
        H.device_sync()
        run_case(f"gemm bf16 NN {n}^3", 2.0 * n ** 3, lambda: torch.matmul(a, b, out=c),
                 lambda: H.gemm(H.BF16, 0, 0, n, n, n, 1.0, ka.ptr, n, kb.ptr, n, 0.0, kc.ptr, n))
        # the backward pair of a linear layer: dA = dC B^T, dB = A^T dC (two vendor calls against one grouped launch)
        dc = u(n, n)
        da, db = torch.empty_like(c), torch.empty_like(c)
        kdc, kda, kdb = H.DevBuf(2 * n * n), H.DevBuf(2 * n * n), H.DevBuf(2 * n * n)
        H.check(H.lib().kf_memcpy_d2d(kdc.ptr, dc.data_ptr(), 2 * n * n, None))
        H.device_sync()

        def vendor_bwd():
            torch.matmul(dc, b.t(), out=da)
            torch.matmul(a.t(), dc, out=db)

        def ours_bwd():
            H.gemm_grouped(H.BF16, [(0, 1, n, n, n, 1.0, 0.0, kdc.ptr, n, kb.ptr, n, kda.ptr, n), (1, 0, n, n, n, 1.0, 0.0, ka.ptr, n, kdc.ptr, n, kdb.ptr, n)], None)
        run_case(f"gemm bf16 backward pair (NT + TN) {n}^3", 4.0 * n ** 3, vendor_bwd, ours_bwd)
        del a, b, c, dc, da, db, ka, kb, kc, kdc, kda, kdb

    # ---- causal attention, config C3
    B, Hh, S, D = 8, 32, 4096, 128
    pair = B * Hh * S * S * D / 2.0
    q, k, v, go = (u(B, Hh, S, D).requires_grad_(x) for x in (True, True, True, False))
    nb = B * Hh * S * D * 2
    kq, kk, kv, kgo, ko, kdq, kdk, kdv = (H.DevBuf(nb) for _ in range(8))
    torch.cuda.synchronize()
    for dst, src in ((kq, q), (kk, k), (kv, v), (kgo, go)):
        H.check(H.lib().kf_memcpy_d2d(dst.ptr, src.data_ptr(), nb, None))
    klse = H.DevBuf(4 * B * Hh * S)
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    kws = H.DevBuf(need)
    H.device_sync()
    backends = {}
    try:
        from torch.backends.cuda import flash_sdp_enabled, mem_efficient_sdp_enabled, math_sdp_enabled
        backends = {"flash": flash_sdp_enabled(), "mem_efficient": mem_efficient_sdp_enabled(), "math": math_sdp_enabled()}
    except Exception:  # noqa: BLE001
        pass

    def vendor_fwd():
        with torch.no_grad():
            F.scaled_dot_product_attention(q, k, v, is_causal=True)

    def vendor_fwd_bwd():
        o = F.scaled_dot_product_attention(q, k, v, is_causal=True)
        q.grad = k.grad = v.grad = None
        o.backward(go)

    def ours_fwd():
        H.attn_fwd(H.BF16, B, Hh, S, S, D, kq.ptr, kk.ptr, kv.ptr, ko.ptr, klse.ptr)

    def ours_fwd_bwd():
        ours_fwd()
        H.attn_bwd(H.BF16, B, Hh, S, S, D, kq.ptr, kk.ptr, kv.ptr, ko.ptr, klse.ptr, kgo.ptr, kdq.ptr, kdk.ptr, kdv.ptr, kws.ptr, need)

    run_case("causal attention forward bf16 B8 H32 S4096 D128", 4.0 * pair, vendor_fwd, ours_fwd, {"torch_sdp_backends_enabled": backends})
    run_case("causal attention forward + backward bf16 B8 H32 S4096 D128", 14.0 * pair, vendor_fwd_bwd, ours_fwd_bwd,
             {"torch_sdp_backends_enabled": backends, "tokens_per_call": B * S})
    # agreement of the two forwards on the same inputs (context, not a parity claim: both are bf16 kernels)
    o_t = F.scaled_dot_product_attention(q.detach(), k.detach(), v.detach(), is_causal=True)
    ours_fwd()
    H.device_sync()
    torch.cuda.synchronize()
    mine = torch.empty_like(o_t)
    H.check(H.lib().kf_memcpy_d2d(mine.data_ptr(), ko.ptr, nb, None))
    H.device_sync()
    out["forward_max_abs_difference_vendor_vs_ours"] = float((mine.float() - o_t.float()).abs().max())
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
