"""Batch-sharded data parallelism for one node of 1..8 MI355X.

The hot path shards along the batch: GEMM's M (flattened leading dims) and attention's B have no
cross-unit dependence, so forward and backward need NO collective. The only exchange step is the sum
all-reduce of the weight gradients dW = sum_r A_r^T dC_r, done once per step on a flat bucket:

  * on GPUs: RCCL over xGMI through the C ABI (kf_comm_init / kf_allreduce_sum), on its own HIP stream so
    it overlaps the rest of the backward; the 128-byte RCCL unique id travels over torch.distributed's
    store (gloo) — plumbing only;
  * on CPU (tests, world_size 2): the same bucket logic over torch.distributed's gloo backend.

One process per GPU; ranks come from RANK / LOCAL_RANK / WORLD_SIZE (torch.distributed.run).
"""
from __future__ import annotations

import ctypes as C
import os

# the pool's host driver only supports dmabuf IPC: without this RCCL's cross-process buffer sharing fails (hipIpcGetMemHandle: invalid argument).
# Must be in the environment before HIP initialises; children inherit it.
os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
import time
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np


def _job_start() -> float:
    """When this job began: the launcher's own clock at the moment it made the rendezvous directory (KF_RDZV_T0), else this process's
    creation time. A store file older than that belongs to somebody else."""
    t0 = os.environ.get("KF_RDZV_T0")
    if t0:
        return float(t0)
    try:
        import psutil
        return psutil.Process().create_time()
    except Exception:  # noqa: BLE001
        return time.time()


def shard_range(n_items: int, rank: int, world: int) -> Tuple[int, int]:
    """[lo, hi) of the batch items rank `rank` owns: contiguous, sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    base, extra = divmod(n_items, world)
    lo = rank * base + min(rank, extra)
    return lo, lo + base + (1 if rank < extra else 0)


@dataclass
class BucketSlot:
    offset: int  # elements
    shape: Tuple[int, ...]


class GradBucket:
    """Flat gradient bucket: many weight gradients, one collective (bigger, fewer messages — the ring /
    direct algorithms over 7 x 153 GB/s xGMI links only reach their plateau on messages of tens of MiB)."""

    def __init__(self, shapes: Sequence[Sequence[int]], dtype=np.float32):
        self.dtype = np.dtype(dtype)
        self.slots: List[BucketSlot] = []
        off = 0
        for s in shapes:
            n = int(np.prod(s)) if len(s) else 1
            self.slots.append(BucketSlot(off, tuple(int(x) for x in s)))
            off += (n + 63) // 64 * 64  # 64-element alignment keeps every slot 16-byte aligned for any dtype
        self.numel = off

    def nbytes(self) -> int:
        return self.numel * self.dtype.itemsize

    def view(self, flat: np.ndarray, i: int) -> np.ndarray:
        s = self.slots[i]
        n = int(np.prod(s.shape)) if s.shape else 1
        return flat[s.offset:s.offset + n].reshape(s.shape)

    def byte_offset(self, i: int) -> int:
        return self.slots[i].offset * self.dtype.itemsize


class ProcessGroup:
    """Rendezvous + (on CPU) collectives over torch.distributed; on GPU the data path is RCCL via the C ABI."""

    def __init__(self, backend: str = "auto"):
        import torch.distributed as dist
        self.dist = dist
        self.rank = int(os.environ.get("RANK", "0"))
        self.world = int(os.environ.get("WORLD_SIZE", "1"))
        self.local_rank = int(os.environ.get("LOCAL_RANK", str(self.rank)))
        if not dist.is_initialized():
            rdzv = os.environ.get("KF_RDZV_FILE")
            if rdzv:
                # ranks this repository's own launchers started (bench.py / tools/block_bench.py --gpus N without torchrun): a FILE store in a
                # fresh temporary directory of the parent - no TCP port to guess, nothing another job on the box can race for. The store must
                # be NEW: one left behind by a crashed job at the same path would hand this rendezvous that job's keys. The
                # launcher's directory is fresh, so the file can only exist once a rank of THIS job has created it: younger than the job.
                if os.path.exists(rdzv) and os.path.getmtime(rdzv) < _job_start() - 2.0:
                    raise RuntimeError(f"KF_RDZV_FILE={rdzv} exists and is older than this job: a stale store of another job "
                                       "(remove it, or let the launcher make a fresh directory)")
                dist.init_process_group("gloo", init_method=f"file://{rdzv}", rank=self.rank, world_size=self.world)
            else:   # under torch.distributed.run: its MASTER_ADDR / MASTER_PORT
                os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
                os.environ.setdefault("MASTER_PORT", "29517")
                dist.init_process_group("gloo", rank=self.rank, world_size=self.world)
        self.backend = backend
        self.comm = None
        if backend == "auto":
            from . import hip_abi
            self.backend = "rccl" if hip_abi.device_count() > 0 else "gloo"
        if self.backend == "rccl":
            self._init_rccl()

    def _init_rccl(self):
        from . import hip_abi as H
        H.set_device(self.local_rank)
        ident = [None]
        if self.rank == 0:
            buf = C.create_string_buffer(H.COMM_ID_BYTES)
            H.check(H.lib().kf_comm_unique_id(buf))
            ident[0] = buf.raw
        self.dist.broadcast_object_list(ident, src=0)
        h = C.c_void_p()
        H.check(H.lib().kf_comm_init(C.byref(h), ident[0], self.rank, self.world))
        self.comm = h.value

    def barrier(self):
        self.dist.barrier()

    def allreduce_sum_host(self, flat: np.ndarray) -> np.ndarray:
        """gloo path: in-place sum all-reduce of a host bucket (CPU tests of the N > 1 logic)."""
        import torch
        t = torch.from_numpy(flat)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return flat

    def allreduce_sum_device(self, ptr: int, count: int, dtype_code: int, stream=None):
        """RCCL path: in-place sum all-reduce of `count` elements at device pointer `ptr` on `stream`."""
        from . import hip_abi as H
        if self.comm is None:
            raise RuntimeError("RCCL communicator not initialised (backend is %r)" % self.backend)
        H.check(H.lib().kf_allreduce_sum(self.comm, ptr, count, dtype_code, stream))

    def max_over_ranks(self, value: float) -> float:
        import torch
        t = torch.tensor([value], dtype=torch.float64)
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t[0])

    def close(self):
        if self.comm is not None:
            from . import hip_abi as H
            H.check(H.lib().kf_comm_destroy(self.comm))
            self.comm = None
        if self.dist.is_initialized():
            self.dist.destroy_process_group()
