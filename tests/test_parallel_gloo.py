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
    assert line == {"dry_run": True, "n_gpus": 2, "allreduce_check": True, "elapsed_s": line["elapsed_s"]}


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
