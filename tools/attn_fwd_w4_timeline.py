#!/usr/bin/env python3
"""Where a wave of the generated attention forward (tools/gen_attn_fwd.py, attn_fwd_w4_kernel) spends its cycles: a diagnostic build
of the device library (generator --stamps, -DKF_FWD_W4_STAMPS -DKF_ATTN_TIMELINE=3 -> tools/scratch/lib_w4_stamps.so; nothing of it is in
libkfunca_hip.so) takes s_memtime at the four slot boundaries of the steady tile body, once per iteration of the other variants and
around the prologue and the epilogue; every wave writes eight sums per query block.

    python tools/attn_fwd_w4_timeline.py [--B 8 --H 32 --S 4096] [--zeros] [--build-only]
"""
import argparse
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "tools" / "scratch" / "lib_w4_stamps.so"
INC = ROOT / "kfunca_amd" / "_build" / "attn_fwd_w4_stamps.inc"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--S", type=int, default=4096)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--name", default="", help="suffix of the diagnostic library (placement experiments: KF_GEN_* set, --name X --build-only, then --name X)")
    args = ap.parse_args()
    global LIB, INC
    if args.name:
        LIB = LIB.with_name(LIB.stem + "_" + args.name + ".so")
        INC = INC.with_name(INC.stem + "_" + args.name + ".inc")
    if not LIB.exists() or args.build_only:
        INC.parent.mkdir(exist_ok=True)
        subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_fwd.py"), "--stamps", "--out", str(INC)], check=True)
        subprocess.run([sys.executable, str(ROOT / "tools" / "scratch" / "build_variant.py"), LIB.stem[4:], "attention.hip", "-DKF_ATTN_TIMELINE=3",
                        "-DKF_FWD_W4_STAMPS", f'-DKF_FWD_W4_INC="{INC}"'], check=True)
        if args.build_only:
            return
    os.environ["KF_HIP_LIB"] = str(LIB)
    sys.path.insert(0, str(ROOT))
    from kfunca_amd import hip_abi as H

    B, Hh, S, D = args.B, args.H, args.S, 128
    H.set_device(0)
    rng = np.random.default_rng(0)
    x = rng.uniform(-1, 1, size=(Hh, S, D)).astype(np.float32)
    u = x.view(np.uint32)
    host = ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)
    if args.zeros:
        host[:] = 0
    per = host.nbytes
    bufs = []
    for _ in range(3):
        b = H.DevBuf(B * per)
        for i in range(B):
            H.check(H.lib().kf_memcpy_h2d(b.ptr + i * per, host.ctypes.data, per, None))
        bufs.append(b)
    o, lse = H.DevBuf(B * per), H.DevBuf(4 * B * Hh * S)
    nwg = B * Hh * (S // 512)
    tl = H.DevBuf.from_numpy(np.zeros((nwg, 2, 4, 8), dtype=np.uint32))
    fn = H.lib().kfdbg_attn_timeline
    fn.argtypes = [ctypes.c_void_p]
    H.check(fn(tl.ptr))
    for _ in range(3):
        H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr)
    H.device_sync()
    e0, e1 = H.Event(), H.Event()
    e0.record(None)
    H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr)
    e1.record(None)
    H.device_sync()
    H.check(fn(None))
    t = tl.to_numpy((nwg, 2, 4, 8), np.uint32).astype(np.float64)   # [workgroup, pass, wave, sums]
    cal = t[..., 7].mean()
    steady = t[..., 1:5].sum() / 1.0
    n_steady = None
    # s91 is not stored: the steady iteration count follows from the shape (wave w of block x runs 4 x + w - 1 steady iterations, x = 0 .. S/256 - 1)
    nxb = S // 256
    n_steady = sum(max(0, 4 * x + w - 1) for x in range(nxb) for w in range(4)) * B * Hh
    print(f"B {B} H {Hh} S {S} D {D} {'zeros' if args.zeros else 'uniform(-1,1)'}: kernel {e0.elapsed_ms(e1):.3f} ms under the stamps; one stamp costs {cal:.0f} cycles "
          f"(taken off each bucket below per stamp)")
    names = ["slot A: S(b0) 16 MFMA + b1's exp chain", "slot B: PV(b1) + b0 max / decision + DMA + bookkeeping", "slot C: S(b1) + b0's exp chain + V reads",
             "slot D: PV(b0) + b1 max / decision + K reads"]
    tot = 0.0
    for i, nm in enumerate(names):
        c = t[..., 1 + i].sum() / n_steady - cal
        tot += c
        print(f"  {nm:58s} {c:7.0f} cycles / steady iteration   (16 MFMAs = 512 of matrix pipe)")
    print(f"  steady iteration {tot:.0f} cycles (2048 of matrix pipe: {100 * 2048 / tot:.0f} % busy)")
    nblocks = nwg * 2 * 4
    other_iters = sum((5 if 4 * x + w - 1 >= 0 else 4) - 0 for x in range(nxb) for w in range(4))  # first, masked, drain + idle ones: T + 1 - steady
    other_iters = sum((4 * (x + 1) + 1) - max(0, 4 * x + w - 1) for x in range(nxb) for w in range(4)) * B * Hh
    print(f"  per block and wave: prologue {t[..., 0].sum() / nblocks - cal:.0f} cycles, epilogue {t[..., 6].sum() / nblocks - cal:.0f}, "
          f"other iterations (first / diagonal / drain / idle) {t[..., 5].sum() / other_iters - cal:.0f} cycles each, {other_iters / nblocks:.1f} of them per block")
    per_block = (t[..., 0] + t[..., 1:7].sum(axis=-1)).max(axis=2)   # the slowest wave of a block
    print(f"  a block's pass: {per_block.mean():.0f} cycles on average; two passes per workgroup, {nwg / 256:.0f} workgroups per CU -> "
          f"{per_block.sum() / 256:.0f} cycles per CU = {per_block.sum() / 256 / (e0.elapsed_ms(e1) * 1e3):.0f} MHz x kernel time (the rest: launch gaps, clock)")


if __name__ == "__main__":
    main()
