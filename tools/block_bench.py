#!/usr/bin/env python3
"""BASELINE config C5: a bf16 causal-attention transformer block forward + backward, batch-sharded over N GPUs of one node with the
weight gradients sum-all-reduced over RCCL / xGMI - through the Python operator API, Tensor.backward and the C++ gradient bucket.

    python tools/block_bench.py [--gpus N] [--steps K] [--warmup W] [--form reference|fused|fused-norm] [--check] [--graph]

One process per GPU. `--gpus N` without a torchrun environment starts the N ranks itself (children of a parent that never touches a
GPU); under `python -m torch.distributed.run --nproc-per-node N tools/block_bench.py --gpus N` each process is one rank. Every rank
holds the same weights (same seed) and its own batch element (rank-offset seed): per-GPU batch 1, S 4096, d_model 4096 = 32 heads
x 128, gated MLP of width 16384; weak scaling. The five weight gradients (and the two norm gains of
--form fused-norm) live in ONE flat bucket (kfunca.GradBucket) cut into chunks of --bucket-mb; a chunk's all-reduce is issued on the
bucket's communication stream the moment the backward pass has produced its last gradient, so it runs under the remaining backward;
the step ends with bucket.wait(). With one rank and no communicator the bucket still collects the gradients (no collective).

Rank 0 prints ONE JSON line: ms per step (max over ranks), tokens/s over the whole job, matrix TFLOP/s, the bucket plan, the
all-reduce bytes per step and - with --check - the verdicts:
  allreduce_vs_gloo_sum   every rank reruns the step WITHOUT the collective, the ranks' flat gradients are summed in f32 over gloo on
                          the host, and the RCCL-reduced bucket must lie within N 2^-8 sum_r |dW_r| of it (one bf16 rounding per add);
  shard_vs_oracle         rank 0's own block, at full size: sampled rows of y and dx and sampled rows of all five dW of the
                          NOT-reduced step against oracle/block_ref.py (float32 BLAS + the attention oracle on the same bf16 inputs, rounding
                          to bf16 where the device path stores a tensor), and the kernel labels that ran (matrix-core attention + MFMA GEMMs,
                          nothing generic).
--dry-run-cpu: the launcher, the rendezvous and the bucket plan without a GPU (tests/test_parallel_gloo.py)."""
import argparse
import json
import os

# the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle: invalid argument).
# Must be in the environment before HIP initialises; children inherit it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import subprocess
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

B, S, HH, D, F = 1, 4096, 32, 128, 16384
DM, T = HH * D, B * S
WSHAPES = ((DM, 3 * DM), (DM, DM), (DM, F), (DM, F), (F, DM))  # qkv, out-proj, gate, up, down


def launch_ranks(n: int) -> int:
    """Start the n ranks as CHILD processes (this process has not touched, and never touches, a GPU), pass rank 0's line through and
    return the worst exit status."""
    # rendezvous over a FILE store in a fresh temporary directory (kfunca_amd/parallel.py, KF_RDZV_FILE): round 4 picked a TCP port by
    # bind-then-close, which any other job on the box could take in between
    import tempfile
    from bench import RANK_TIMEOUT_S, supervise  # bounded: a rank that fails or hangs ends the others (bench.py)
    with tempfile.TemporaryDirectory(prefix="kf_rdzv_") as tmp:   # removed with its store file whatever happens to the ranks
        rdzv, t_job = Path(tmp) / "store", time.time()
        procs = []
        for r in range(n):
            env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), KF_RDZV_FILE=str(rdzv), KF_RDZV_T0=repr(t_job))
            procs.append(subprocess.Popen([sys.executable, str(Path(__file__).resolve()), *sys.argv[1:]], env=env,
                                          stdout=subprocess.PIPE if r == 0 else sys.stderr, stderr=sys.stderr, text=True))
        out0, codes = supervise(procs, RANK_TIMEOUT_S)
    lines = [ln for ln in (out0 or "").splitlines() if ln.strip()]
    for ln in lines[:-1]:
        print(ln, file=sys.stderr)
    if lines and all(c == 0 for c in codes):
        print(lines[-1], flush=True)
    return max(abs(c) for c in codes)


def param_numels(with_gains: bool):
    numels = [a * b for a, b in WSHAPES]
    if with_gains:
        numels = [DM, numels[0], numels[1], DM, numels[2], numels[3], numels[4]]
    return numels


def bucket_plan(cap_mb: float, with_gains: bool, elem_bytes: int = 2):
    """[(first, last, offset, numel)] for the block's parameters in forward order (chunk 0 = the last ones): pure arithmetic, no GPU."""
    import kfunca_amd as kfunca
    return [tuple(int(v) for v in c) for c in kfunca.GradBucket.plan(param_numels(with_gains), int(cap_mb * 1048576) // elem_bytes)]


def grad_f32(args, world: int) -> bool:
    """The bucket's element type: --grad f32 | bf16 | auto (default). auto = float from 8 ranks on (a 16-bit bucket rounds once per
    addition: N 2^-8 sum|dW_r| = 3 % of sum|dW| at N = 8), bf16 below (half the bytes on the wire). DESIGN.md section 6."""
    return args.grad == "f32" or (args.grad == "auto" and world >= 8)


def dry_run_cpu(args, rank, world):
    import kfunca_amd as kfunca
    from kfunca_amd import parallel
    pg = parallel.ProcessGroup(backend="gloo")
    with_gains = args.form == "fused-norm"
    es = 4 if grad_f32(args, world) else 2
    plan = bucket_plan(args.bucket_mb, with_gains)   # --bucket-mb counts 16-bit elements: a float bucket keeps the chunking and doubles the bytes
    total = sum(c[3] for c in plan)
    nparams = len(WSHAPES) + (2 if with_gains else 0)
    # the order the chunks' collectives leave in when the gradients arrive as the backward produces them (reverse of the forward's use) -
    # run through the C++ core's own bookkeeping (GradBucket::Tracker, the code arrived() / wait() run on the GPU path). Every rank must
    # get the SAME order: RCCL matches collectives by sequence number
    numels = param_numels(with_gains)
    fired = [int(c) for c in kfunca.GradBucket.simulate_fired_order(numels, int(args.bucket_mb * 1048576) // 2, list(range(nparams - 1, -1, -1)))]
    flat = np.full(1024, float(rank + 1), dtype=np.float32)  # a stand-in message: the plan is what is under test here
    pg.allreduce_sum_host(flat)
    ok = bool((flat == world * (world + 1) / 2).all()) and plan[0][1] == nparams - 1 and plan[-1][0] == 0
    key = float(hash((tuple(plan), tuple(fired))) % (1 << 40))
    same = pg.max_over_ranks(float(total)) == float(total) and pg.max_over_ranks(key) == key and -pg.max_over_ranks(-key) == key
    pg.barrier()
    if rank == 0:
        print(json.dumps({"dry_run": True, "n_gpus": world, "bucket_chunks": plan, "bucket_elements": total, "fired_order": fired,
                          "grad_dtype": "f32" if es == 4 else "bf16", "allreduce_check": ok and same}), flush=True)
    pg.close()
    return 0 if ok and same else 1


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--json", default="")
    ap.add_argument("--graph", action="store_true", help="one GPU only: capture one step into a HIP graph and time its replays")
    ap.add_argument("--form", choices=["reference", "fused", "fused-norm"], default="fused",
                    help="reference: only operators the reference API has; fused: attention on the packed QKV projection + GEMM tails "
                         "(same math); fused-norm: that plus the two rms_norms of a real pre-norm block")
    ap.add_argument("--bucket-mb", type=float, default=136.0,
                    help="largest chunk of the gradient bucket reduced by one collective (default: [down] [up] [gate] [out-proj + qkv], 128 MiB each)")
    ap.add_argument("--grad", choices=["auto", "bf16", "f32"], default="auto",
                    help="element type of the gradient bucket and of the RCCL sum: auto = f32 from 8 ranks on, bf16 below")
    ap.add_argument("--check", action="store_true")
    ap.add_argument("--force-comm", action="store_true", help="one rank: still create the RCCL communicator and run the collectives")
    ap.add_argument("--dry-run-cpu", action="store_true", help=argparse.SUPPRESS)
    args = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world == 1:
        return launch_ranks(args.gpus)  # before anything touches the GPU
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    if args.dry_run_cpu:
        return dry_run_cpu(args, rank, world)

    import kfunca_amd as kfunca
    from kfunca_amd import hip_abi as H
    from kfunca_amd import parallel
    if H.device_count() == 0:
        raise SystemExit("block_bench.py needs a GPU: the HIP path has no CPU fallback")
    dev = local_rank
    H.set_device(dev)
    from bench import deadline
    pg = None
    if world > 1 or args.force_comm:
        if rank == 0:  # RCCL's own warnings of rank 0 go to stderr (stdout carries THE line)
            os.environ.setdefault("NCCL_DEBUG", "WARN")
            os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")
        with deadline("rendezvous + ncclCommInitRank"):
            pg = parallel.ProcessGroup(backend="gloo")  # rendezvous + host-side sums for --check; the data path is the C++ core's communicator
            ident = [kfunca.comm_unique_id() if rank == 0 else None]
            pg.dist.broadcast_object_list(ident, src=0)
            kfunca.comm_init(ident[0], rank, world, dev)
        import ctypes
        ctypes.CDLL(None).fflush(None)  # RCCL's version banner

    wrng = np.random.default_rng(1005)            # weights: identical on every rank
    drng = np.random.default_rng(1005 + 7919 * (rank + 1))  # data: one batch element per rank

    def dev_bf16(a, grad):
        t = kfunca.from_numpy(np.ascontiguousarray(a, dtype=np.float32), dev).bfloat16()
        t.set_requires_grad(grad)
        return t

    w_host = [(wrng.uniform(-1, 1, s) / np.sqrt(s[0])).astype(np.float32) for s in WSHAPES]
    gain_host = [np.ones(DM, np.float32) + 0.1 * wrng.uniform(-1, 1, DM).astype(np.float32) for _ in range(2)]
    x_host = drng.uniform(-1, 1, (T, DM)).astype(np.float32)
    g_host = drng.uniform(-1, 1, (T, DM)).astype(np.float32)
    x = dev_bf16(x_host, True)
    w = [dev_bf16(a, True) for a in w_host]
    gains = [dev_bf16(a, True) for a in gain_host]
    g = dev_bf16(g_host, False)
    # the bucket takes the parameters in the order the forward uses them (gradients arrive in about the reverse of it)
    if args.form == "fused-norm":
        params, pnames = [gains[0], w[0], w[1], gains[1], w[2], w[3], w[4]], ["dgain1", "dWqkv", "dWo", "dgain2", "dWgate", "dWup", "dWdown"]
    else:
        params, pnames = list(w), ["dWqkv", "dWo", "dWgate", "dWup", "dWdown"]
    f32_bucket = grad_f32(args, world)
    bucket = kfunca.GradBucket(params, args.bucket_mb * (2 if f32_bucket else 1), f32_bucket)   # the same chunking in elements either way
    bucket.attach()

    def body():
        if args.form == "reference":
            qkv = kfunca.gemm(x, w[0], 1.0, 0.0)
            q, k, v = (t.contiguous().view(B, S, HH, D).permute(0, 2, 1, 3).contiguous() for t in qkv.split([DM, DM, DM], 1))
            a = kfunca.causal_attention(q, k, v).permute(0, 2, 1, 3).contiguous().view(T, DM)
            h = x + kfunca.gemm(a, w[1], 1.0, 0.0)
            return h + kfunca.gemm(kfunca.gemm(h, w[2], 1.0, 0.0) * kfunca.gemm(h, w[3], 1.0, 0.0), w[4], 1.0, 0.0)
        n1 = kfunca.rms_norm(x, gains[0], 1e-5) if args.form == "fused-norm" else x
        a = kfunca.causal_attention_qkv(kfunca.gemm(n1, w[0], 1.0, 0.0), B, S, HH)
        h = kfunca.gemm_fused(a, w[1], 1.0, None, None, x)
        n2 = kfunca.rms_norm(h, gains[1], 1e-5) if args.form == "fused-norm" else h
        up = kfunca.gemm(n2, w[3], 1.0, 0.0)
        return kfunca.gemm_fused(kfunca.gemm_fused(n2, w[2], 1.0, None, up, None), w[4], 1.0, None, None, h)

    def step():  # one training step's shape: gradients start empty, forward, backward (chunks leave as they complete), join
        for t in [x] + params:
            t.zero_grad()
        y = body()
        y.backward(g)
        bucket.wait()
        return y

    def barrier():
        with deadline("device sync + barrier"):
            kfunca.synchronize(dev)
            if pg is not None:
                pg.barrier()

    for _ in range(args.warmup):
        step()
    barrier()
    graph = None
    if args.graph and world == 1 and not kfunca.comm_initialized():
        kfunca.graph_begin(dev)
        step()
        graph = kfunca.graph_end(dev)
        kfunca.graph_launch(graph, dev)
        kfunca.synchronize(dev)
    H.profile_reset()
    H.profile_enable(graph is None)
    def timed():
        barrier()
        t0 = time.perf_counter()
        with deadline(f"{args.steps} timed steps"):
            for _ in range(args.steps):
                if graph is not None:
                    kfunca.graph_launch(graph, dev)
                else:
                    step()
        barrier()
        dt = time.perf_counter() - t0
        return pg.max_over_ranks(dt) if pg is not None else dt
    elapsed = timed()
    H.profile_enable(False)
    ms = elapsed / args.steps * 1e3
    prof = {k: {"ms_per_step": v[0] / args.steps, "launches_per_step": v[1] / args.steps} for k, v in H.profile_results().items()}
    fired = list(bucket.fired_order())
    # attribution: what each chunk's collective took on the communication stream (last step), and the same timed
    # loop with the collectives left out - the difference is the communication the backward could not hide
    chunk_ms = [float(v) for v in bucket.chunk_ms()]
    # Round 5: the A/B is INTERLEAVED (on, off, on, off) and both arms run in the SAME state - per-launch profiling
    # off. Round 4 compared the profiled main loop (two extra event records around each of ~50 launches per step) with an unprofiled
    # no-comm loop: 0.40 ms of "exposed communication" at world 1, where a collective takes 4.6 us. The noise floor of the A/B itself
    # (the larger difference between the two runs of one arm) is printed beside the figure.
    ms_off, ab, ab_noise = None, None, None
    if kfunca.comm_initialized() and graph is None:
        ab = []
        for arm in (True, False, True, False):
            bucket.set_collectives(arm)
            ab.append(timed() / args.steps * 1e3)
        bucket.set_collectives(True)
        ms_on_ab, ms_off = (ab[0] + ab[2]) / 2, (ab[1] + ab[3]) / 2
        ab_noise = max(abs(ab[0] - ab[2]), abs(ab[1] - ab[3]))

    checks = {}
    if args.check:
        step()                                   # the reduced gradients of one more step ...
        kfunca.synchronize(dev)
        reduced = (bucket.flat().numpy() if f32_bucket else bits_to_f32(bucket.flat().numpy())).copy()
        H.profile_reset()
        H.profile_enable(True)
        y_bits, dx_bits, dw_bits = local_gradients(kfunca, bucket, params, x, body, g, dev)  # ... and the same step with nothing reduced
        H.profile_enable(False)
        labels = set(H.profile_results())
        checks.update(check_allreduce(kfunca, pg, bucket, params, reduced, dw_bits, world if kfunca.comm_initialized() else 1, f32_bucket))
        if rank == 0:
            checks.update(check_shard(H, args.form, w_host, gain_host, x_host, g_host, y_bits, dx_bits, dict(zip(pnames, dw_bits)), labels))

    rc = 0
    if rank == 0:
        flops = 6.0 * T * DM * (4 * DM + 3 * F) + 14.0 * B * S * S * DM / 2
        nbytes = bucket.reduced_bytes()
        out = {"config": "C5: bf16 causal-attention block fwd+bwd, per-GPU B=1 S=4096 d=4096 H=32 D=128 f=16384, batch-sharded, dW all-reduced over RCCL",
               "form": args.form, "mode": "hip graph replay" if graph is not None else "eager", "n_gpus": world, "steps": args.steps,
               "ms_per_step": ms, "tokens_per_s": world * T / (ms * 1e-3), "matrix_tflops_per_gpu": flops / (ms * 1e-3) / 1e12,
               "scaling": "weak", "bucket": {"chunks_first_last_offset_numel": [list(c) for c in bucket.chunks()], "fired_order_last_step": fired,
                                               "bytes_reduced_per_step": nbytes, "grad_dtype": "f32" if f32_bucket else "bf16", "collective": "RCCL" if kfunca.comm_initialized() else "none (one rank)"},
               "allreduce_busbw_lower_bound_GBps": (2.0 * (world - 1) / world * nbytes / (ms * 1e-3) / 1e9) if world > 1 else 0.0,
               "allreduce_chunk_ms": chunk_ms,
               "allreduce_chunk_busbw_GBps": [(2.0 * (world - 1) / world * c[3] * (4 if f32_bucket else 2) / (t * 1e-3) / 1e9) if (world > 1 and t > 0) else 0.0
                                              for c, t in zip(bucket.chunks(), chunk_ms)],
               "ms_per_step_no_comm": ms_off, "ms_per_step_comm_ab_on_off_on_off": ab,
               "exposed_comm_ms": (ms_on_ab - ms_off) if ms_off is not None else None, "exposed_comm_noise_floor_ms": ab_noise,
               "overlap_efficiency": (max(0.0, min(1.0, 1.0 - max(0.0, ms_on_ab - ms_off) / sum(chunk_ms))) if (ms_off is not None and sum(chunk_ms) > 0) else None),
               "device_ms_per_step": sum(v["ms_per_step"] for v in prof.values()),
               "elementwise_launches_per_step": sum(v["launches_per_step"] for k, v in prof.items() if k.startswith("ew_")),
               "kernels": prof, "checks": checks}
        if args.json:
            Path(args.json).write_text(json.dumps(out, indent=1))
        rc = 0 if all(v is True for v in checks.values() if isinstance(v, bool)) else 1
    elif checks and not all(v is True for v in checks.values() if isinstance(v, bool)):
        rc = 1
    if pg is not None:
        pg.barrier()
    if rank == 0:
        print(json.dumps(out), flush=True)
    bucket.detach()
    if kfunca.comm_initialized():
        kfunca.comm_destroy()
    if pg is not None:
        pg.close()
    return rc


def bits_to_f32(bits):
    return (bits.astype(np.uint32) << 16).view(np.float32)


def slot_offsets(numels):
    off, o = [], 0
    for n in numels:
        off.append(o)
        o += (n + 63) // 64 * 64
    return off


def local_gradients(kfunca, bucket, params, x, body, g, dev):
    """The same forward + backward with the bucket DETACHED: every rank's own gradients, nothing reduced (returns y, dx, [dW] as bf16 bits)."""
    bucket.detach()
    for t in [x] + params:
        t.zero_grad()
    y = body()
    y.backward(g)
    kfunca.synchronize(dev)
    out = (y.numpy(), x.grad().numpy(), [p.grad().numpy() for p in params])
    for t in [x] + params:
        t.zero_grad()
    bucket.attach()
    return out


def check_allreduce(kfunca, pg, bucket, params, reduced_flat, local, world, f32_bucket=False):
    """SURVEY.md section 8e's parity sentence on the live bucket: the RCCL-reduced flat gradient == the sum over ranks of each rank's own
    gradients (taken in f32 over gloo on the host), within N 2^-8 sum_r |dW_r| (RCCL adds bf16 values: one rounding per addition);
    with one rank the collective is the identity and the two must be bit-identical."""
    offs = slot_offsets([int(np.prod(p.sizes())) for p in params])
    ok = True
    worst = 0.0
    for p, off, mine_bits in zip(params, offs, local):
        n = mine_bits.size
        got = reduced_flat[off:off + n]
        mine = bits_to_f32(mine_bits).reshape(-1)
        if world == 1 and not f32_bucket:
            ok = ok and bool(np.array_equal(got.view(np.uint32), mine.view(np.uint32)))
            continue
        total = pg.allreduce_sum_host(mine.copy()) if pg is not None else mine.copy()
        mag = pg.allreduce_sum_host(np.abs(mine)) if pg is not None else np.abs(mine)
        err = np.abs(got.astype(np.float64) - total)
        # a float bucket sums the GEMMs' unrounded f32 accumulators: against the sum of the ranks' bf16-ROUNDED local gradients that is one
        # rounding of each rank's term (2^-9 |dW_r|; 2^-8 leaves room for the f32 additions) - nothing that grows with the number of ranks
        bound = (1 if f32_bucket else world) * 2.0 ** -8 * mag + 1e-30
        worst = max(worst, float((err / bound).max()))
        ok = ok and bool((err <= bound).all())
    return {"allreduce_vs_gloo_sum": ok, "allreduce_worst_fraction_of_bound": worst}


# the kernels a full-size shard must run on: matrix-core attention (forward, dK/dV, dQ from the stored dS) and the 16-bit MFMA GEMM
# families (256-tile kernels, the backward pair as one grid, split-K for the skinny products); nothing generic
MUST_RUN = ("attn_fwd_mfma", "attn_bwd_dkv_mfma", "attn_bwd_dq_mfma", "gemm_bf16_mfma")


def check_shard(H, form, w_host, gain_host, x_host, g_host, y_bits, dx_bits, dw_bits, labels):
    """Rank 0's own block at FULL size against oracle/block_ref.py (float32 BLAS + the attention oracle on the bf16 inputs, rounding to
    bf16 exactly where the device path stores a tensor): 64 sampled rows of y and dx, 64 sampled rows of each dW and each whole tensor's
    relative Frobenius error; bound 1.5e-2 of the tensor's norm (measured: tools/block_bench.py --check prints the figures)."""
    from oracle import block_ref as R
    r = R.r16
    gains = [r(a) for a in gain_host] if form == "fused-norm" else None
    y, dx, dw, dg = R.block_fwd_bwd(r(x_host), [r(a) for a in w_host], r(g_host), B, S, HH, D, gains=gains)
    rng = np.random.default_rng(5)
    figs = {}

    def cmp(name, got_bits, want):
        got = bits_to_f32(got_bits)
        rows = np.sort(rng.choice(got.shape[0], size=min(64, got.shape[0]), replace=False)) if got.ndim == 2 else slice(None)
        figs[name] = {"rel_fro": R.rel_fro(got, want), "rel_fro_sampled_rows": R.rel_fro(got[rows], want[rows]),
                      "max_abs_over_max": float(np.abs(got - want).max() / (np.abs(want).max() + 1e-30))}
        return bool(np.isfinite(got).all()) and figs[name]["rel_fro"] < 1.5e-2 and figs[name]["rel_fro_sampled_rows"] < 1.5e-2
    ok = cmp("y", y_bits, y) & cmp("dx", dx_bits, dx)
    for nme, wb in zip(("dWqkv", "dWo", "dWgate", "dWup", "dWdown"), dw):
        ok &= cmp(nme, dw_bits[nme], wb)
    if dg is not None:
        for nme, wb in zip(("dgain1", "dgain2"), dg):
            ok &= cmp(nme, dw_bits[nme], wb)
    ran = sorted(labels)
    kernels_ok = all(any(k.startswith(m) for k in ran) for m in MUST_RUN) and not any("generic" in k for k in ran)
    return {"shard_vs_oracle": bool(ok), "shard_kernels_are_the_matrix_core_ones": kernels_ok, "shard_figures": figs, "shard_kernels": ran}


if __name__ == "__main__":
    sys.exit(main())
