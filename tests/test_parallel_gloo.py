"""CPU-only, world_size 2 over gloo: the N > 1 path of the hot path is batch sharding with ONE exchange
step, the sum all-reduce of the weight gradient. The per-rank math is done by the CPU oracle (this is a
test: the oracle is the checker); what is under test is kfunca_amd.parallel — shard ranges, the flat
gradient bucket and the all-reduce — against the single-process full-batch result."""
import os
import socket

import numpy as np
import pytest
import torch.multiprocessing as mp

from kfunca_amd import parallel
from oracle import oracle as O


def test_shard_range_partitions_exactly():
    for n in (0, 1, 7, 8, 9, 64, 1000):
        for world in (1, 2, 3, 4, 8):
            spans = [parallel.shard_range(n, r, world) for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            sizes = [hi - lo for lo, hi in spans]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        parallel.shard_range(8, 2, 2)


def test_bucket_layout():
    b = parallel.GradBucket([(4096, 4096), (7,), (3, 5)], dtype=np.float32)
    assert b.slots[0].offset == 0 and b.slots[1].offset == 4096 * 4096 and b.slots[2].offset == 4096 * 4096 + 64
    assert b.numel % 64 == 0 and all(b.byte_offset(i) % 16 == 0 for i in range(3))
    flat = np.arange(b.numel, dtype=np.float32)
    assert b.view(flat, 2).shape == (3, 5) and b.view(flat, 2)[0, 0] == b.slots[2].offset


def _worker(rank, world, port, out):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    pg = parallel.ProcessGroup(backend="gloo")
    rng = np.random.default_rng(77)           # identical on every rank: full batch + replicated weights
    M, K, N = 48, 32, 40
    A = rng.uniform(-1, 1, (M, K)).astype(np.float32)
    dC = rng.uniform(-1, 1, (M, N)).astype(np.float32)
    extra = rng.uniform(-1, 1, (M, 5)).astype(np.float32)
    lo, hi = parallel.shard_range(M, rank, world)
    bucket = parallel.GradBucket([(K, N), (K, 5)], dtype=np.float32)
    flat = np.zeros(bucket.numel, dtype=np.float32)
    # per-rank weight gradients of its batch shard: dW_r = A_r^T dC_r
    bucket.view(flat, 0)[...] = O.gemm(A[lo:hi], dC[lo:hi], trans_a=True)
    bucket.view(flat, 1)[...] = O.gemm(A[lo:hi], extra[lo:hi], trans_a=True)
    pg.allreduce_sum_host(flat)
    worst = pg.max_over_ranks(float(rank))
    full0, full1 = O.gemm(A, dC, trans_a=True), O.gemm(A, extra, trans_a=True)
    ok = (np.allclose(bucket.view(flat, 0), full0, rtol=1e-5, atol=1e-5) and np.allclose(bucket.view(flat, 1), full1, rtol=1e-5, atol=1e-5)
          and worst == world - 1)
    out[rank] = bool(ok)
    pg.close()


def test_gradient_allreduce_world2_matches_full_batch():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


def test_bench_launches_its_own_ranks_cpu_dry_run():
    """`python bench.py --gpus 2` with no torchrun environment starts the two rank processes itself (children of a parent that
    never touches a GPU), the ranks rendezvous on 127.0.0.1, all-reduce a gradient bucket through kfunca_amd.parallel, and the
    parent relays rank 0's ONE JSON line and the worst exit status. --dry-run-cpu swaps the RCCL data path for gloo."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "bench.py"), "--gpus", "2", "--dry-run-cpu"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line == {"dry_run": True, "n_gpus": 2, "allreduce_check": True, "elapsed_s": line["elapsed_s"], "grad_dtype": "bf16"}


def test_gradient_bucket_plan_is_pure_arithmetic_and_cuts_from_the_end():
    """kfunca.GradBucket.plan (C++ core, csrc/core/comm.cpp): slots 64-element aligned in parameter order, chunks of at most `cap`
    elements cut from the END of the list (the last parameters' gradients arrive first), every parameter in exactly one chunk, an
    oversized parameter alone in its chunk. No device needed."""
    import kfunca_amd as kfunca
    plan = kfunca.GradBucket.plan([100, 7, 4096, 64, 1], 4200)
    assert plan == [(3, 4, 4288, 128), (1, 2, 128, 4160), (0, 0, 0, 128)]  # slots 128 | 64 | 4096 | 64 | 64 elements at 0, 128, 192, 4288, 4352
    assert kfunca.GradBucket.plan([10, 20, 30], 1) == [(2, 2, 128, 64), (1, 1, 64, 64), (0, 0, 0, 64)]  # smaller than any slot: one chunk each
    assert kfunca.GradBucket.plan([10, 20, 30], 10 ** 9) == [(0, 2, 0, 192)]
    d, f = 4096, 16384  # config C5's block with the default 136 MiB cap of tools/block_bench.py: [down] [up] [gate] [out-proj + qkv]
    numels = [d * 3 * d, d * d, d * f, d * f, f * d]
    chunks = kfunca.GradBucket.plan(numels, 136 * 1048576 // 2)
    assert [(c[0], c[1]) for c in chunks] == [(4, 4), (3, 3), (2, 2), (0, 1)]
    assert sum(c[3] for c in chunks) == sum(numels) and all(c[2] % 64 == 0 for c in chunks)


def test_block_bench_launches_its_own_ranks_cpu_dry_run():
    """`python tools/block_bench.py --gpus 2` (config C5's multi-rank runner) with no torchrun environment: the parent starts both
    ranks, they rendezvous over gloo, agree on the gradient bucket's plan and the parent relays rank 0's one line."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    res = subprocess.run([sys.executable, str(Path(__file__).resolve().parent.parent / "tools" / "block_bench.py"), "--gpus", "2", "--dry-run-cpu"],
                         capture_output=True, text=True, env=env, timeout=300)
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["allreduce_check"] is True
    assert [c[:2] for c in line["bucket_chunks"]] == [[4, 4], [3, 3], [2, 2], [0, 1]]


def _block_worker(rank, world, port, out):
    """Config C5's exchange step, numerically: every rank runs ITS batch element of the block restatement (oracle/block_ref.py),
    lays the five dW (and the two norm gains) into the flat gradient bucket and all-reduces it over gloo."""
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from oracle import block_ref as R
    pg = parallel.ProcessGroup(backend="gloo")
    B, S, H, D, f = world, 64, 2, 32, 128
    d = H * D
    rng = np.random.default_rng(505)             # identical on every rank: the full batch + the replicated weights
    x = R.r16(rng.uniform(-1, 1, (B * S, d)).astype(np.float32))
    g = R.r16(rng.uniform(-1, 1, (B * S, d)).astype(np.float32))
    shapes = ((d, 3 * d), (d, d), (d, f), (d, f), (f, d))
    w = [R.r16((rng.uniform(-1, 1, s) / np.sqrt(s[0])).astype(np.float32)) for s in shapes]
    gains = [R.r16(np.ones(d, np.float32) + 0.1 * rng.uniform(-1, 1, d).astype(np.float32)) for _ in range(2)]
    lo, hi = parallel.shard_range(B, rank, world)   # batch elements of this rank
    rows = slice(lo * S, hi * S)
    _, dx_r, dw_r, dg_r = R.block_fwd_bwd(x[rows], w, g[rows], hi - lo, S, H, D, gains=gains)
    bucket = parallel.GradBucket([*shapes, (d,), (d,)], dtype=np.float32)
    flat = np.zeros(bucket.numel, dtype=np.float32)
    for i, a in enumerate([*dw_r, *dg_r]):
        bucket.view(flat, i)[...] = a
    mag = np.abs(flat)
    pg.allreduce_sum_host(flat)
    pg.allreduce_sum_host(mag)                    # sum_r |dW_r|: the scale the per-rank roundings are relative to
    ok = True
    if rank == 0:  # the full batch in ONE process: its gradients are the sum of the shards' (attention and norms are per row / per batch element)
        _, dx_full, dw_full, dg_full = R.block_fwd_bwd(x, w, g, B, S, H, D, gains=gains)
        for i, a in enumerate([*dw_full, *dg_full]):
            got = bucket.view(flat, i)
            # every rank rounds ITS dW to bf16 once (the device path stores 16-bit gradients) and the full batch rounds the sum once:
            # |difference| <= 2^-9 (sum_r |dW_r| + |dW|) elementwise (+ f32 BLAS blocking noise), nothing that grows with the
            # number of additions: the all-reduce itself adds in f32
            bound = 2.0 ** -8 * (bucket.view(mag, i) + np.abs(a)) + 1e-6
            ok = ok and bool((np.abs(got - a) <= bound).all()) and bool(np.isfinite(got).all()) and float(np.abs(a).max()) > 0
        # activations' gradients need no exchange at all: the shard's rows ARE the full batch's rows (up to BLAS blocking at another M:
        # an occasional bf16 rounding flip)
        ok = ok and bool((np.abs(dx_full[rows] - dx_r) <= 2.0 ** -7 * np.abs(dx_full[rows]) + 1e-6).all())
    out[rank] = bool(ok)
    pg.close()


def test_block_gradients_of_two_batch_shards_sum_to_the_full_batch():
    """a world-2 gloo NUMERICAL test of config C5 - two ranks each run half the batch of oracle/block_ref.py, the
    all-reduced bucket equals the full-batch gradients (SURVEY.md section 8e's parity sentence, on CPU)."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_block_worker, args=(2, port, out), nprocs=2, join=True)
    assert dict(out) == {0: True, 1: True}


# ---- N = 8 readiness without an 8-GPU node: the launcher, the file store, supervise() and the C5 chunk plan with
# EIGHT ranks on CPU. 8 interpreters importing torch at once need ~2.5 GiB and a minute on 8 cores.
def _run_tool(script, *argv, env_extra=None, timeout=600):
    import subprocess
    import sys
    from pathlib import Path
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "KF_RDZV_FILE", "KF_RDZV_T0")}
    env.update(env_extra or {})
    root = Path(__file__).resolve().parent.parent
    return subprocess.run([sys.executable, str(root / script), *argv], capture_output=True, text=True, env=env, timeout=timeout)


def test_bench_world8_cpu_dry_run_selects_float_gradients():
    """`python bench.py --gpus 8 --dry-run-cpu`: eight self-launched ranks over the file store; from 8 ranks on the gradient all-reduce is
    float by default (bench.grad_f32_default), bf16 when KF_BENCH_GRAD_BF16=1 asks for it."""
    import json
    res = _run_tool("bench.py", "--gpus", "8", "--dry-run-cpu")
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] and line["n_gpus"] == 8 and line["allreduce_check"] is True and line["grad_dtype"] == "f32"
    import bench
    assert [bench.grad_f32_default(n) for n in (1, 2, 4, 8)] == [False, False, False, True]
    os.environ["KF_BENCH_GRAD_BF16"] = "1"
    try:
        assert bench.grad_f32_default(8) is False
    finally:
        del os.environ["KF_BENCH_GRAD_BF16"]


@pytest.mark.parametrize("mode", ["exit", "hang"])
def test_world8_job_ends_when_rank5_fails(mode):
    """One rank of eight exits with status 3 / never reaches the collective: supervise() (exit) or the ranks' own deadline() watchdogs and
    the parent's limit (hang) end the WHOLE job with a non-zero status and no JSON line - nobody waits for somebody else's limit."""
    import time
    t0 = time.monotonic()
    res = _run_tool("bench.py", "--gpus", "8", "--dry-run-cpu",
                    env_extra={"KF_BENCH_DRY_FAULT": f"5:{mode}", "KF_BENCH_PHASE_TIMEOUT_S": "20", "KF_BENCH_RANK_TIMEOUT_S": "240"})
    took = time.monotonic() - t0
    assert res.returncode != 0
    assert not [ln for ln in res.stdout.splitlines() if ln.strip().startswith("{")], res.stdout
    assert "ending the remaining ranks" in res.stderr, res.stderr[-2000:]
    if mode == "exit":
        assert "rank 5 exited with status 3" in res.stderr, res.stderr[-2000:]
    assert took < 240, took


def test_block_bench_world8_plan_and_fired_order():
    """Config C5's runner with eight ranks on CPU: every rank computes the bucket's chunk plan and - through the C++ core's own Tracker, the
    code GradBucket::arrived / wait run - the order the chunks' collectives leave in; all eight agree (RCCL matches collectives by sequence),
    the bucket is float at 8 ranks and the chunking in ELEMENTS is the one the bf16 bucket has."""
    import json
    res = _run_tool("tools/block_bench.py", "--gpus", "8", "--dry-run-cpu")
    assert res.returncode == 0, res.stderr[-2000:]
    line = json.loads([ln for ln in res.stdout.splitlines() if ln.strip()][-1])
    assert line["dry_run"] and line["n_gpus"] == 8 and line["allreduce_check"] is True and line["grad_dtype"] == "f32"
    assert [c[:2] for c in line["bucket_chunks"]] == [[4, 4], [3, 3], [2, 2], [0, 1]] and line["fired_order"] == [0, 1, 2, 3]


def test_tracker_fires_each_chunk_once_whatever_the_arrival_order():
    """GradBucket::Tracker (device-free): a chunk fires when its LAST parameter arrives, a repeated arrival (a second backward before wait())
    fires nothing, chunks nobody completed are appended by finish() in index order - so every rank issues every collective exactly once."""
    import itertools
    import kfunca_amd as kfunca
    numels, cap = [100, 7, 4096, 64, 1], 4200       # chunks (3,4) (1,2) (0,0): tests above
    sim = kfunca.GradBucket.simulate_fired_order
    assert sim(numels, cap, [4, 3, 2, 1, 0]) == [0, 1, 2]
    assert sim(numels, cap, [0, 1, 2, 3, 4]) == [2, 1, 0]
    assert sim(numels, cap, [4, 4, 3, 3]) == [0, 1, 2]          # repeats fire nothing; the rest leaves at finish()
    assert sim(numels, cap, []) == [0, 1, 2]
    assert sim(numels, cap, [2, 4, 0]) == [2, 0, 1]             # chunk 2 completes first; 0 and 1 are incomplete: finish() in index order
    for order in itertools.permutations(range(5)):
        assert sorted(sim(numels, cap, list(order))) == [0, 1, 2]
    with pytest.raises(RuntimeError):
        sim(numels, cap, [5])


def test_stale_rendezvous_file_is_refused(tmp_path):
    """a store file left behind by a crashed job must not be joined. ProcessGroup refuses a KF_RDZV_FILE older than the job."""
    import time
    store = tmp_path / "store"
    store.write_bytes(b"left behind")
    old = time.time() - 3600
    os.utime(store, (old, old))
    saved = {k: os.environ.get(k) for k in ("KF_RDZV_FILE", "KF_RDZV_T0", "RANK", "WORLD_SIZE")}
    os.environ.update(KF_RDZV_FILE=str(store), KF_RDZV_T0=repr(time.time()), RANK="0", WORLD_SIZE="1")
    try:
        with pytest.raises(RuntimeError, match="stale store"):
            parallel.ProcessGroup(backend="gloo")
    finally:
        for k, v in saved.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def test_bench_under_the_drivers_own_launch_line_cpu_dry_run():
    """The driver starts N > 1 as `python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P
    bench.py --gpus N ...`: the ranks find RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* in their environment, must NOT start children of their
    own, rendezvous over torchrun's TCP store and rank 0 alone prints the one JSON line."""
    import json
    import socket
    import subprocess
    import sys
    from pathlib import Path
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "KF_RDZV_FILE", "KF_RDZV_T0")}
    root = Path(__file__).resolve().parent.parent
    res = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
                          "--master-port", str(port), str(root / "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1", "--dry-run-cpu"],
                         capture_output=True, text=True, env=env, timeout=300, cwd=str(root))
    assert res.returncode == 0, res.stderr[-2000:]
    lines = [ln for ln in res.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, res.stdout
    line = json.loads(lines[0])
    assert line["dry_run"] and line["n_gpus"] == 2 and line["allreduce_check"] is True
