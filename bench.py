#!/usr/bin/env python3
"""bench.py — the hot path's headline benchmark on MI355X.

    python bench.py --gpus N --steps K --warmup W
    (N > 1: launched by torch.distributed.run, one rank per GPU; see the driver contract)

One "step" = one pass of the hot path over one batch of synthetic bf16 tensors, everything through the
C ABI (include/kfunca_hip.h) with inputs already resident in HBM:
    GEMM 4096^3 forward + backward   C = A W,  dA = dC W^T,  dW = A^T dC       (BASELINE target GEMM)
    causal attention forward + backward, B=8 H=32 S=4096 D=128               (BASELINE configs[2] / C3)
At N > 1 the batch is sharded (weak scaling: every rank runs the per-GPU batch above) and the weight
gradient dW is sum-all-reduced over RCCL/xGMI on a second stream, overlapped with the attention pass.

Prints ONE JSON line on rank 0: tokens/s over the whole job, per-kernel HIP-event durations, the
`roofline` of the dominant kernel and a `cpu_baseline` (the CPU oracle timed on a bounded sample).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent
sys.path.insert(0, str(ROOT))

from kfunca_amd import hip_abi as H  # noqa: E402

GEMM_N = 4096
AB, AH, AS, AD = 8, 32, 4096, 128
PEAK_MFMA_BF16 = 2500.0  # TFLOP/s dense, MI355X_MICROARCH.md "Peak BF16/FP16 MFMA"
PEAK_HBM = 8000.0        # GB/s spec, same guide

# algorithmic FLOPs per launch: causal attention counts S^2/2 score entries
_PAIR = AB * AH * AS * AS * AD / 2.0
KERNEL_FLOPS = {
    "gemm_bf16_mfma": 2.0 * GEMM_N ** 3,
    "attn_fwd_mfma": 4.0 * _PAIR,       # QK^T + PV
    "attn_bwd_dkv_mfma": 8.0 * _PAIR,   # S, dP, dV, dK
    "attn_bwd_dq_mfma": 2.0 * _PAIR,    # dQ (its S / dP recomputation is overhead, not algorithmic work)
}
GEMM_FLOPS_STEP = 6.0 * GEMM_N ** 3
ATTN_FLOPS_STEP = 14.0 * _PAIR
TOKENS_STEP = AB * AS


def bf16_random(rng, shape):
    x = rng.uniform(-1.0, 1.0, size=shape).astype(np.float32)
    u = x.view(np.uint32)
    return ((u + 0x7FFF + ((u >> 16) & 1)) >> 16).astype(np.uint16)


class Workload:
    def __init__(self, rank):
        rng = np.random.default_rng(1003 + 7919 * rank)  # seed = 1000 + config number (C3), rank-offset data
        n = GEMM_N
        self.A = H.DevBuf.from_numpy(bf16_random(rng, (n, n)))
        wrng = np.random.default_rng(1003)                # weights: identical on every rank
        self.W = H.DevBuf.from_numpy(bf16_random(wrng, (n, n)))
        self.dC = H.DevBuf.from_numpy(bf16_random(rng, (n, n)))
        self.Cc, self.dA, self.dW = H.DevBuf(2 * n * n), H.DevBuf(2 * n * n), H.DevBuf(2 * n * n)
        need = max(H.gemm_workspace_bytes(H.BF16, ta, tb, n, n, n) for ta, tb in ((0, 0), (0, 1), (1, 0)))
        self.gws_bytes = need
        self.gws = H.DevBuf(need)
        nbytes = AB * AH * AS * AD * 2
        self.q, self.k, self.v, self.o, self.do = (H.DevBuf(nbytes) for _ in range(5))
        self.dq, self.dk, self.dv = (H.DevBuf(nbytes) for _ in range(3))
        per_b = nbytes // AB
        for buf in (self.q, self.k, self.v, self.do):  # one random batch element, replicated on device
            host = bf16_random(rng, (AH, AS, AD))
            H.check(H.lib().kf_memcpy_h2d(buf.ptr, host.ctypes.data, per_b, None))
            for b in range(1, AB):
                H.check(H.lib().kf_memcpy_d2d(buf.ptr + b * per_b, buf.ptr, per_b, None))
        self.lse = H.DevBuf(4 * AB * AH * AS)
        self.aws_bytes = H.attn_bwd_workspace_bytes(H.BF16, AB, AH, AS, AS, AD)
        self.aws = H.DevBuf(self.aws_bytes)
        H.device_sync()

    def step(self, stream, comm=None, comm_stream=None, ev_grad=None, ev_comm=None):
        n, s = GEMM_N, stream
        if comm is not None and ev_comm.recorded:  # dW of the previous step must be fully reduced before it is rewritten
            H.stream_wait_event(s, ev_comm)
        H.gemm(H.BF16, 0, 0, n, n, n, 1.0, self.A.ptr, n, self.W.ptr, n, 0.0, self.Cc.ptr, n, 0, None, self.gws.ptr, self.gws_bytes, s)
        H.gemm(H.BF16, 0, 1, n, n, n, 1.0, self.dC.ptr, n, self.W.ptr, n, 0.0, self.dA.ptr, n, 0, None, self.gws.ptr, self.gws_bytes, s)
        H.gemm(H.BF16, 1, 0, n, n, n, 1.0, self.A.ptr, n, self.dC.ptr, n, 0.0, self.dW.ptr, n, 0, None, self.gws.ptr, self.gws_bytes, s)
        if comm is not None:  # gradient all-reduce on its own stream, overlapped with the attention pass
            ev_grad.record(s)
            H.stream_wait_event(comm_stream, ev_grad)
            H.check(H.lib().kf_allreduce_sum(comm, self.dW.ptr, n * n, H.BF16, comm_stream))
            ev_comm.record(comm_stream)
            ev_comm.recorded = True
        H.attn_fwd(H.BF16, AB, AH, AS, AS, AD, self.q.ptr, self.k.ptr, self.v.ptr, self.o.ptr, self.lse.ptr, s)
        H.attn_bwd(H.BF16, AB, AH, AS, AS, AD, self.q.ptr, self.k.ptr, self.v.ptr, self.o.ptr, self.lse.ptr, self.do.ptr,
                   self.dq.ptr, self.dk.ptr, self.dv.ptr, self.aws.ptr, self.aws_bytes, s)


def cpu_baseline():
    """The CPU oracle (kind "port": the reference has no CPU path) on a bounded sample of
    the same step: attention fwd+bwd for `heads` of the 256 (b,h) pairs at full S, and `rows` of the 4096
    output rows of each of the three GEMMs; scaled linearly to the whole step."""
    from oracle import oracle as O
    cores = O.num_threads()
    rng = np.random.default_rng(1003)
    heads, rows, n = max(1, min(cores, 128)), 512, GEMM_N  # ~10-20 s of CPU work on 8..256 cores
    q, k, v, go = (bf16_random(rng, (1, heads, AS, AD)) for _ in range(4))
    t0 = time.perf_counter()
    O.attn_fwd(q, k, v, code=O.BF16)
    O.attn_bwd(q, k, v, go, code=O.BF16)
    t_attn = time.perf_counter() - t0
    a, w, g = (bf16_random(rng, (n, n)) for _ in range(3))
    f = O.lib().orc_gemm
    f.argtypes = [C.c_int, C.c_int, C.c_int, C.c_int64, C.c_int64, C.c_int64, C.c_float, C.c_void_p, C.c_int64,
                  C.c_void_p, C.c_int64, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]
    out = np.empty((rows, n), dtype=np.uint16)
    t0 = time.perf_counter()
    f(O.BF16, 0, 0, rows, n, n, 1.0, a.ctypes.data, n, w.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # C  = A W
    f(O.BF16, 0, 1, rows, n, n, 1.0, g.ctypes.data, n, w.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # dA = dC W^T
    f(O.BF16, 1, 0, rows, n, n, 1.0, a.ctypes.data, n, g.ctypes.data, n, 0.0, out.ctypes.data, n, None)   # dW = A^T dC (rows of dW)
    t_gemm = time.perf_counter() - t0
    t_step = t_attn * (AB * AH / heads) + t_gemm * (n / rows)
    return {"value": TOKENS_STEP / t_step, "unit": "tokens/s", "cores": cores, "kind": "port",
            "sample": f"attention fwd+bwd of {heads}/256 (b,h) pairs at S=4096 D=128 ({t_attn:.1f} s) + {rows}/4096 output rows "
                      f"of each of the 3 GEMMs ({t_gemm:.1f} s), scaled linearly to one step"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if args.gpus != 1 or world != 1:
            raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    if H.device_count() == 0:
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    H.set_device(local_rank)

    dist = None
    comm = None
    if world > 1:
        import torch.distributed as dist  # plumbing only: rendezvous, barrier, max-reduce of the timing
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)
    if world > 1 or os.environ.get("KF_BENCH_FORCE_COMM"):  # the env knob exercises the collective path on one GPU
        ident = [None]
        if rank == 0:
            buf = C.create_string_buffer(H.COMM_ID_BYTES)
            H.check(H.lib().kf_comm_unique_id(buf))
            ident[0] = buf.raw
        if dist is not None:
            dist.broadcast_object_list(ident, src=0)
        h = C.c_void_p()
        H.check(H.lib().kf_comm_init(C.byref(h), ident[0], rank, world))
        comm = h.value
        C.CDLL(None).fflush(None)  # RCCL leaves its version banner in the C stdout buffer: out with it now, so the JSON line is the last line

    wl = Workload(rank)
    stream = H.Stream()
    comm_stream = H.Stream() if comm else None
    ev_grad, ev_comm = (H.Event(), H.Event()) if comm else (None, None)
    if comm:
        ev_comm.recorded = False
    lib = H.lib()

    def barrier():
        H.device_sync()
        if dist is not None:
            dist.barrier()

    def run_step():
        wl.step(stream.handle, comm, comm_stream.handle if comm else None, ev_grad, ev_comm)

    for _ in range(args.warmup):
        run_step()
    barrier()
    H.profile_reset()
    H.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_step()
    H.device_sync()
    if dist is not None:
        dist.barrier()
    t1 = time.perf_counter()
    H.profile_enable(False)
    elapsed = t1 - t0
    if dist is not None:
        import torch
        t = torch.tensor([elapsed], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t[0])
    prof = H.profile_results()

    if rank == 0:
        ms_step = elapsed / args.steps * 1e3
        kern = {k: {"avg_ms": ms / max(cnt, 1), "launches": cnt} for k, (ms, cnt) in prof.items()}
        for k, v in kern.items():
            if k in KERNEL_FLOPS:
                v["tflops"] = KERNEL_FLOPS[k] / (v["avg_ms"] * 1e-3) / 1e12
        per_step = {k: ms / args.steps for k, (ms, cnt) in prof.items()}
        gemm_ms = sum(v for k, v in per_step.items() if k.startswith("gemm"))
        attn_ms = sum(v for k, v in per_step.items() if k.startswith("attn"))
        dom = max((k for k in per_step if k in KERNEL_FLOPS), key=lambda k: per_step[k])
        achieved = kern[dom]["tflops"]
        # traffic beyond L2 per launch of the dominant kernel: rocprofv3 PMC passes cannot run inside this process;
        # the committed measurement of the same kernel on the same shape is reported (profiles/r01_pmc_traffic.json)
        traffic = None
        tfile = ROOT / "profiles" / "r01_pmc_traffic.json"
        if tfile.exists():
            traffic = json.loads(tfile.read_text()).get(dom, {}).get("bytes_per_launch")
        out = {
            "metric": "bf16 GEMM TFLOP/s + causal-attn fwd+bwd tokens/s",
            "value": world * TOKENS_STEP / (elapsed / args.steps),
            "unit": "tokens/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": ms_step,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
            "config": {"workload": "bf16 GEMM 4096x4096x4096 fwd+bwd + bf16 causal attention fwd+bwd B=8 H=32 S=4096 D=128 per GPU",
                       "global_batch": AB * world, "seq_len": AS, "parallelism": f"dp{world}"},
            "gemm_tflops": GEMM_FLOPS_STEP / (gemm_ms * 1e-3) / 1e12 if gemm_ms else None,
            "gemm_ms_per_step": gemm_ms,
            "attn_tokens_per_s": TOKENS_STEP / (attn_ms * 1e-3) if attn_ms else None,
            "attn_tflops": ATTN_FLOPS_STEP / (attn_ms * 1e-3) / 1e12 if attn_ms else None,
            "attn_ms_per_step": attn_ms,
            "kernels": kern,
            "roofline": {"kernel": dom, "bound": "mfma", "achieved": achieved, "peak": PEAK_MFMA_BF16, "unit": "TFLOP/s",
                         "frac": achieved / PEAK_MFMA_BF16, "traffic": traffic,
                         "algorithmic_flops_per_launch": KERNEL_FLOPS[dom]},
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
    if comm is not None:
        H.check(lib.kf_comm_destroy(comm))
    C.CDLL(None).fflush(None)
    if dist is not None:
        dist.barrier()  # every rank's library output is out before rank 0 prints THE line
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
