#!/usr/bin/env python3
"""Where a wave of the attention forward spends its cycles: a diagnostic build of the device library (-DKF_ATTN_TIMELINE, built here
into tools/scratch/lib_timeline.so; nothing of it is in libkfunca_hip.so) stamps s_memtime at the phase boundaries of the tile loop
and every wave writes its seven sums. Prints, per phase, the mean share of a wave's loop time and cycles per tile iteration.

    python tools/attn_timeline.py [--B 8 --H 32 --S 4096 --D 128] [--zeros]
"""
import argparse
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "tools" / "scratch" / "lib_timeline.so"
PHASES = ["wait own DMA (vmcnt)", "barrier", "issue 4 LDS-DMA", "Q K^T (16 MFMA issue)", "softmax (+ MFMA drain)", "P V (16 MFMA)", "loop overhead"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--B", type=int, default=8)
    ap.add_argument("--H", type=int, default=32)
    ap.add_argument("--S", type=int, default=4096)
    ap.add_argument("--D", type=int, default=128)
    ap.add_argument("--zeros", action="store_true")
    ap.add_argument("--build-only", action="store_true")
    ap.add_argument("--outside", action="store_true", help="forward: cycles of a pass before / in / behind the tile loop (build -DKF_ATTN_TIMELINE=2)")
    ap.add_argument("--dkv", action="store_true", help="the dK/dV kernel's slice phases instead of the forward's tile loop (D = 128)")
    args = ap.parse_args()
    global LIB
    if args.outside:
        LIB = ROOT / "tools" / "scratch" / "lib_timeline2.so"
    if not LIB.exists() or args.build_only:
        subprocess.run([sys.executable, str(ROOT / "tools" / "scratch" / "build_variant.py"), "timeline2" if args.outside else "timeline", "attention.hip",
                        "-DKF_ATTN_TIMELINE=2" if args.outside else "-DKF_ATTN_TIMELINE"], check=True)
        if args.build_only:
            return
    os.environ["KF_HIP_LIB"] = str(LIB)
    sys.path.insert(0, str(ROOT))
    from kfunca_amd import hip_abi as H

    B, Hh, S, D = args.B, args.H, args.S, args.D
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
    nblk = B * Hh * (S // 256)
    tl = H.DevBuf.from_numpy(np.zeros((nblk, 2, 8, 8), dtype=np.uint64))
    fn = H.lib().kfdbg_attn_timeline
    fn.argtypes = [ctypes.c_void_p]
    for _ in range(3):
        H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr)
    H.device_sync()
    if args.dkv:
        return dkv_timeline(H, args, bufs, o, lse, fn)
    H.check(fn(tl.ptr))
    H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr)
    H.device_sync()
    H.check(fn(None))
    t = tl.to_numpy((nblk, 2, 8, 8), np.uint64).astype(np.float64)
    if args.outside:
        blocks = t[t[..., 7].max(axis=-1) > 0]  # [passes, 8 waves, 8]
        per = blocks.max(axis=1)                # the slowest wave of a pass sets its length
        nt = per[:, 7]
        tot = per[:, 0] + per[:, 1] + per[:, 2]
        print(f"forward passes {len(per)}: mean cycles before the tile loop {per[:, 0].mean():.0f}, in it {per[:, 1].mean():.0f} ({(per[:, 1] / nt).mean():.0f} per tile), "
              f"behind it {per[:, 2].mean():.0f}; outside the loop {100 * (per[:, 0] + per[:, 2]).sum() / tot.sum():.1f} % of a pass")
        a_, b_ = np.polyfit(nt, per[:, 1], 1)
        print(f"  loop cycles = {a_:.0f} x tiles + {b_:.0f}  (the intercept is the diagonal tiles' excess)")
        for lo in (4, 16, 32, 64):
            sel = (nt >= lo) & (nt < lo * 2)
            if sel.any():
                print(f"  passes of {lo}..{2 * lo - 1} tiles: before {per[sel, 0].mean():.0f}, loop {per[sel, 1].mean():.0f}, behind {per[sel, 2].mean():.0f}")
        return
    t = t[t[..., 7] > 0]  # (passes x waves) that ran: [n, 8]
    tiles = t[:, 7].sum()
    tot = t[:, :7].sum()
    print(f"B {B} H {Hh} S {S} D {D} {'zeros' if args.zeros else 'uniform(-1,1)'}: {len(t)} wave passes, {tot / tiles:.0f} cycles per tile iteration of a wave")
    for i, name in enumerate(PHASES):
        print(f"  {name:28s} {100 * t[:, i].sum() / tot:5.1f} %   {t[:, i].sum() / tiles:7.0f} cycles / iteration")
    early, late = t.reshape(-1, 8, 8)[:, :4].reshape(-1, 8), t.reshape(-1, 8, 8)[:, 4:].reshape(-1, 8)
    for nm, g in (("early waves", early), ("late waves", late)):
        print(f"  {nm}: " + ", ".join(f"{100 * g[:, i].sum() / g[:, :7].sum():.0f}%" for i in range(7)))


DKV_PHASES = ["q0 S k0-3", "q1 S k4-7 (+ dP constants)", "q2 dP k0-3 (+ 8 exp)", "q3 dP k4-7 (+ 8 exp, pack)", "q4 dV k0 (+ dS)", "q5 dV k1 (+ dS, pack)",
              "q6 vmcnt + barrier (pair ends)", "q6 dK k0 (+ DMA issue at pair ends)", "q7 dK k1 (+ constants, DMA issue)", "between slices (dS stores, stamp collection)"]


def dkv_timeline(H, args, bufs, o, lse, fn):
    B, Hh, S, D = args.B, args.H, args.S, args.D
    per = Hh * S * D * 2
    go = H.DevBuf(B * per)
    H.check(H.lib().kf_memcpy_d2d(go.ptr, bufs[0].ptr, B * per, None)) if hasattr(H.lib(), "kf_memcpy_d2d") else None
    dq, dk, dv = H.DevBuf(B * per), H.DevBuf(B * per), H.DevBuf(B * per)
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    ws = H.DevBuf(need)
    nblk = B * Hh * (S // 128)
    tl = H.DevBuf.from_numpy(np.zeros((nblk, 2, 4, 16), dtype=np.uint64))
    run = lambda: H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs[0].ptr, bufs[1].ptr, bufs[2].ptr, o.ptr, lse.ptr, go.ptr, dq.ptr, dk.ptr, dv.ptr, ws.ptr, need)  # noqa: E731
    run()
    H.device_sync()
    H.check(fn(tl.ptr))
    run()
    H.device_sync()
    H.check(fn(None))
    t = tl.to_numpy((nblk, 2, 4, 16), np.uint64).astype(np.float64).reshape(-1, 16)
    t = t[t[:, 10] > 0]
    slices = t[:, 10].sum()
    tot = t[:, :10].sum()
    print(f"dK/dV, B {B} H {Hh} S {S} D {D}: {len(t)} wave passes, {slices:.0f} slices, {tot / slices:.0f} cycles per slice of a wave (32 MFMAs = 1024 cycles of pipe time)")
    for i, name in enumerate(DKV_PHASES):
        print(f"  {name:48s} {100 * t[:, i].sum() / tot:5.1f} %   {t[:, i].sum() / slices:6.0f} cycles / slice")
    print(f"  per pass (one key block of a workgroup): slice loop {tot / len(t):.0f} cycles, before it (K / V fragments, three slice pairs staged, "
          f"first wait + barrier) {t[:, 11].mean():.0f}, after it (drain, dK / dV through LDS to memory) {t[:, 12].mean():.0f}")


if __name__ == "__main__":
    main()
