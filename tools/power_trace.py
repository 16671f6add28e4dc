#!/usr/bin/env python3
"""Board power and clocks beside a back-to-back bf16 GEMM loop (prove or drop the power-wall reading of the GEMM's
clock). A sampler thread polls `amd-smi metric` (falls back to `rocm-smi`) every ~100 ms while 4096^3 GEMMs run for a few seconds, on
uniform(-1, 1) operands and on all-zero operands; TFLOP/s comes from HIP events around the same loop. Writes one JSON object."""
import argparse
import json
import re
import subprocess
import sys
import threading
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from kfunca_amd import hip_abi as H  # noqa: E402


def sample():
    """One reading: {power_w, sclk_mhz, ...} from whichever CLI answers."""
    out = {}
    try:
        r = subprocess.run(["amd-smi", "metric", "-g", "0", "--power", "--clock", "--json"], capture_output=True, text=True, timeout=5)
        if r.returncode == 0 and r.stdout.strip():
            j = json.loads(r.stdout)
            j = j[0] if isinstance(j, list) else j
            p = j.get("power", {})
            for key in ("socket_power", "current_socket_power", "average_socket_power"):
                v = p.get(key)
                if isinstance(v, dict) and isinstance(v.get("value"), (int, float)):
                    out["power_w"] = float(v["value"])
                    break
            clk = j.get("clock", {})
            g = clk.get("gfx_0") or clk.get("gfx") or {}
            v = g.get("clk") if isinstance(g, dict) else None
            if isinstance(v, dict) and isinstance(v.get("value"), (int, float)):
                out["sclk_mhz"] = float(v["value"])
            if out:
                return out
    except Exception:
        pass
    try:
        r = subprocess.run(["rocm-smi", "-d", "0", "--showpower", "--showclocks"], capture_output=True, text=True, timeout=5)
        m = re.search(r"Power \(W\):\s*([0-9.]+)", r.stdout)
        if m:
            out["power_w"] = float(m.group(1))
        m = re.search(r"sclk clock level:.*\((\d+)Mhz\)", r.stdout)
        if m:
            out["sclk_mhz"] = float(m.group(1))
    except Exception:
        pass
    return out


def run(n, seconds, zeros, layout="NT"):
    ta, tb = {"NN": (0, 0), "NT": (0, 1), "TN": (1, 0)}[layout]
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(n, n)).astype(np.float32).view(np.uint32)
    bits = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
    if zeros:
        bits[:] = 0
    A, B, Cb = H.DevBuf.from_numpy(bits), H.DevBuf.from_numpy(bits[::-1].copy()), H.DevBuf(2 * n * n)
    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            s = sample()
            s["t"] = time.time()
            samples.append(s)
            time.sleep(0.1)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    time.sleep(0.5)  # idle readings first
    t_start = time.time()
    e0, e1 = H.Event(), H.Event()
    launches = 0
    e0.record(None)
    while time.time() < t_start + seconds:
        for _ in range(100):
            H.gemm(H.BF16, ta, tb, n, n, n, 1.0, A.ptr, n, B.ptr, n, 0.0, Cb.ptr, n)
        launches += 100
        H.device_sync()
    e1.record(None)
    e1.sync()
    t_end = time.time()
    stop.set()
    th.join()
    ms = e0.elapsed_ms(e1)
    busy = [s for s in samples if t_start + 1.0 <= s["t"] <= t_end]
    idle = [s for s in samples if s["t"] < t_start]
    avg = lambda xs, k: float(np.mean([s[k] for s in xs if k in s])) if any(k in s for s in xs) else None  # noqa: E731
    return {"operands": "zeros" if zeros else "uniform(-1,1)", "launches": launches, "tflops": 2.0 * n ** 3 * launches / (ms * 1e-3) / 1e12,
            "power_w_idle": avg(idle, "power_w"), "power_w_loaded": avg(busy, "power_w"), "sclk_mhz_loaded": avg(busy, "sclk_mhz"),
            "samples_loaded": len(busy)}


def run_attention(seconds, zeros):
    """The same measurement beside the C3 attention step (forward + backward, B 8, H 32, S 4096, D 128)."""
    B, Hh, S, D = 8, 32, 4096, 128
    rng = np.random.default_rng(1)
    per = Hh * S * D * 2
    bufs = {}
    for name in ("q", "k", "v", "do"):
        x = rng.uniform(-1, 1, size=(Hh, S, D)).astype(np.float32).view(np.uint32)
        host = ((x + 0x7FFF + ((x >> 16) & 1)) >> 16).astype(np.uint16)
        if zeros:
            host[:] = 0
        b = H.DevBuf(B * per)
        for i in range(B):
            H.check(H.lib().kf_memcpy_h2d(b.ptr + i * per, host.ctypes.data, per, None))
        bufs[name] = b
    for name in ("o", "dq", "dk", "dv"):
        bufs[name] = H.DevBuf(B * per)
    lse = H.DevBuf(4 * B * Hh * S)
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    ws = H.DevBuf(need)

    def step():
        H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr)
        H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,
                   bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, need)

    samples, stop = [], threading.Event()

    def poll():
        while not stop.is_set():
            s = sample()
            s["t"] = time.time()
            samples.append(s)
            time.sleep(0.1)

    th = threading.Thread(target=poll, daemon=True)
    th.start()
    for _ in range(3):
        step()
    H.device_sync()
    t_start = time.time()
    steps = 0
    H.profile_reset()
    H.profile_enable(True)
    while time.time() < t_start + seconds:
        for _ in range(20):
            step()
        steps += 20
        H.device_sync()
    H.profile_enable(False)
    t_end = time.time()
    stop.set()
    th.join()
    busy = [s for s in samples if t_start + 1.0 <= s["t"] <= t_end]
    avg = lambda xs, k: float(np.mean([s[k] for s in xs if k in s])) if any(k in s for s in xs) else None  # noqa: E731
    return {"operands": "zeros" if zeros else "uniform(-1,1)", "steps": steps, "kernel_ms": {k: v[0] / v[1] for k, v in H.profile_results().items()},
            "power_w_loaded": avg(busy, "power_w"), "sclk_mhz_loaded": avg(busy, "sclk_mhz"), "samples_loaded": len(busy)}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--attention", action="store_true", help="trace the C3 attention step instead of the GEMM loop")
    ap.add_argument("--n", type=int, default=4096)
    ap.add_argument("--layout", default="NT", choices=["NN", "NT", "TN"], help="operand layout of the GEMM loop (NT = the round-1/2 traces)")
    ap.add_argument("--seconds", type=float, default=4.0)
    ap.add_argument("--json", default="")
    args = ap.parse_args()
    H.set_device(0)
    if args.attention:
        out = {"workload": "attention fwd+bwd B8 H32 S4096 D128 bf16", "first_sample": sample(),
               "runs": [run_attention(args.seconds, False), run_attention(args.seconds, True)]}
    else:
        out = {"n": args.n, "layout": args.layout, "first_sample": sample(),
               "runs": [run(args.n, args.seconds, False, args.layout), run(args.n, args.seconds, True, args.layout)]}
    print(json.dumps(out, indent=1))
    if args.json:
        Path(args.json).write_text(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
