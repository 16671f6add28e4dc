#!/usr/bin/env python3
"""Where a wave of the generated dK / dV kernel (tools/gen_attn_dkv.py, attn_bwd_dkv_w4_kernel) spends its cycles: a diagnostic build of
the device library (generator --stamps, -DKF_DKV_W4_STAMPS -DKF_ATTN_TIMELINE=3 -> tools/scratch/lib_dkv_stamps.so; nothing of it is in
libkfunca_hip.so) takes s_memtime at every slice barrier (the value is consumed behind the wait the barrier has anyway: the LDS reads in
flight are not drained, the schedule measured is the product's), at the end of the prologue and around the epilogue; every wave writes
eight sums per key-block pass.

    python tools/attn_dkv_w4_timeline.py [--B 8 --H 32 --S 4096] [--zeros] [--build-only]
"""
import argparse
import ctypes
import os
import subprocess
import sys
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
LIB = ROOT / "tools" / "scratch" / "lib_dkv_stamps.so"
INC = ROOT / "kfunca_amd" / "_build" / "attn_dkv_w4_stamps.inc"


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
        subprocess.run([sys.executable, str(ROOT / "tools" / "gen_attn_dkv.py"), "--stamps", "--out", str(INC)], check=True)
        subprocess.run([sys.executable, str(ROOT / "tools" / "scratch" / "build_variant.py"), LIB.stem[4:], "attention.hip", "-DKF_ATTN_TIMELINE=3",
                        "-DKF_DKV_W4_STAMPS", f'-DKF_DKV_W4_INC="{INC}"'], check=True)
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
    bufs = {}
    for name in ("q", "k", "v", "do"):
        b = H.DevBuf(B * per)
        for i in range(B):
            H.check(H.lib().kf_memcpy_h2d(b.ptr + i * per, host.ctypes.data, per, None))
        bufs[name] = b
    for name in ("o", "dq", "dk", "dv"):
        bufs[name] = H.DevBuf(B * per)
    lse = H.DevBuf(4 * B * Hh * S)
    need = H.attn_bwd_workspace_bytes(H.BF16, B, Hh, S, S, D)
    ws = H.DevBuf(need)
    nkb = S // 256
    nwg = B * Hh * (nkb // 2)
    tl = H.DevBuf.from_numpy(np.zeros((nwg, 2, 4, 16), dtype=np.uint32))
    fn = H.lib().kfdbg_attn_timeline
    fn.argtypes = [ctypes.c_void_p]
    H.attn_fwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr)
    bwd = lambda: H.attn_bwd(H.BF16, B, Hh, S, S, D, bufs["q"].ptr, bufs["k"].ptr, bufs["v"].ptr, bufs["o"].ptr, lse.ptr, bufs["do"].ptr,  # noqa: E731
                             bufs["dq"].ptr, bufs["dk"].ptr, bufs["dv"].ptr, ws.ptr, need)
    H.check(fn(tl.ptr))
    for _ in range(2):
        bwd()
    H.device_sync()
    H.profile_reset()
    H.profile_enable(True)
    bwd()
    H.device_sync()
    H.profile_enable(False)
    H.check(fn(None))
    ms = {k: v[0] / v[1] for k, v in H.profile_results().items()}["attn_bwd_dkv_mfma"]
    t = tl.to_numpy((nwg, 2, 4, 16), np.uint32).astype(np.float64)   # [workgroup, pass, wave, sums]
    # slices per wave and pass by kind, from the shape: key block x of nkb runs slices 8 x .. S / 32 - 1; wave w: idle 2 w, diag0 1, diag1 1, steady the rest
    tot = S // 32
    n_pass = nwg * 2
    steady_per_cu = t[..., 7].sum() / (nwg * 2 * 4)
    print(f"B {B} H {Hh} S {S} D {D} {'zeros' if args.zeros else 'uniform(-1,1)'}: dK/dV kernel {ms:.3f} ms under the stamps "
          f"({nwg} workgroups of 2 key-block passes, 4 waves; {tot * (nkb + 1) // 2 / nkb:.1f} slices per pass on average)")
    n_steady = t[..., 7].sum()
    print(f"  steady slice (barrier to barrier)      {t[..., 1].sum() / n_steady:7.0f} cycles   (64 MFMAs = 2048 of matrix pipe: {100 * 2048 * n_steady / t[..., 1].sum():.0f} % busy), "
          f"{steady_per_cu:.1f} per wave and pass")
    sA, sB, sC = (t[..., 8 + i].sum() / n_steady for i in range(3))
    print(f"    of which slot S {sA:.0f} | slot dP {sB:.0f} | slot dV {sC:.0f} | barrier wait + barrier + slot dK {t[..., 1].sum() / n_steady - sA - sB - sC:.0f}   (16 MFMAs = 512 of matrix pipe each)")
    for i, (nm, per_wave) in enumerate((("diag1 slice (sub-block 1 on the diagonal)", 1), ("diag0 slice (sub-block 0 on the diagonal, 32 MFMAs)", 1), ("idle slice (keys above the slice: no MFMAs)", 3))):
        n = n_pass * 4 * (per_wave if i < 2 else 0) if i < 2 else n_pass * (0 + 2 + 4 + 6)
        print(f"  {nm:52s} {t[..., 2 + i].sum() / max(n, 1):7.0f} cycles each, {n / (n_pass * 4):.1f} per wave and pass")
    print(f"  per pass and wave: block start -> first slice's reads {t[..., 0].sum() / (n_pass * 4):.0f} cycles, last barrier -> epilogue {t[..., 5].sum() / (n_pass * 4):.0f}, "
          f"epilogue (dV, dK through LDS to memory) {t[..., 6].sum() / (n_pass * 4):.0f}")
    per_pass = t[..., 0:7].sum(axis=-1).max(axis=2)
    print(f"  a pass: {per_pass.mean():.0f} cycles on average (slowest wave); {nwg / 256:.0f} workgroups per CU -> {per_pass.sum() / 256:.0f} cycles per CU = "
          f"{per_pass.sum() / 256 / (ms * 1e3):.0f} MHz x kernel time")


if __name__ == "__main__":
    main()
