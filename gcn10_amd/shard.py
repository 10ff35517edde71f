"""Block distribution and rank plumbing across the GPUs of one node.

The reference deals block ids round-robin over MPI ranks
(``for (i = rank; i < n_blocks; i += size)``, src/main.c:171) and exchanges no
raster data; its only collective is the closing MPI_Barrier (src/main.c:187).
Here one process drives one GPU; blocks are independent, so the data path has
no collective at all and nothing crosses xGMI.  What MPI was used for -- rank
identity, a barrier and, for benchmarking, gathering each rank's timing -- is
done with files in a directory private to the job (``FileGroup``): no RCCL,
no torch, nothing that can fail on a node where the GPUs themselves work.

``TorchGroup`` keeps the same three operations on ``torch.distributed`` (gloo
or nccl = RCCL) for callers that already live inside a torch job; it is used
only when ``GCN10_DIST_BACKEND`` asks for it.
"""
from __future__ import annotations

import json
import os
import shutil
import time
from typing import Any, List, Optional, Sequence


def world_from_env():
    """(rank, local_rank, world_size) as torch.distributed.run (or bench.py's own launcher) exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def blocks_for_rank(block_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Static round-robin share of a block list, exactly src/main.c:171."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %d of %d" % (rank, world))
    return [block_ids[i] for i in range(rank, len(block_ids), world)]


def _proc_start_ticks(pid: int) -> str:
    """Start time of a process (field 22 of /proc/<pid>/stat): tells two processes that were given the same pid apart."""
    try:
        with open("/proc/%d/stat" % pid) as f:
            return f.read().rsplit(")", 1)[1].split()[19]
    except (OSError, IndexError):
        return "0"


def default_rendezvous_dir() -> str:
    """A directory every rank of ONE launch computes alike and no other launch does.

    ``GCN10_RDV_DIR`` when the launcher made one (bench.py --gpus N does); otherwise named after
    the launcher process all ranks are children of (torch.distributed.run's agent): its pid, its
    start time and MASTER_PORT.
    """
    d = os.environ.get("GCN10_RDV_DIR")
    if d:
        return d
    ppid = os.getppid()
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    # an elastic agent that restarts its workers keeps its pid: the run id and the restart count keep a new
    # attempt out of the previous attempt's directory (whose s000000_r*.json files would complete its barriers)
    attempt = "%s_%s" % (os.environ.get("TORCHELASTIC_RUN_ID", "none"), os.environ.get("TORCHELASTIC_RESTART_COUNT", "0"))
    attempt = "".join(c if c.isalnum() or c in "-_" else "_" for c in attempt)[:64]
    return os.path.join(base, "gcn10_rdv_%d_%d_%s_%s_%s" % (os.getuid(), ppid, _proc_start_ticks(ppid),
                                                             os.environ.get("MASTER_PORT", "0"), attempt))


class SoloGroup:
    """World size 1: every operation is the identity."""

    def __init__(self):
        self.rank, self.local_rank, self.world = 0, 0, 1
        self.backend = "none"

    def barrier(self):
        pass

    def all_gather(self, obj: Any) -> List[Any]:
        return [obj]

    def max(self, value: float) -> float:
        return float(value)

    def sum(self, value: float) -> float:
        return float(value)

    def close(self):
        pass


class FileGroup:
    """Barrier and all-gather through small files in one directory of this node.

    Every collective step has a sequence number; a rank publishes ``s<seq>_r<rank>.json``
    (written to a temporary name, then renamed: readers never see half a file) and polls for the
    other ranks' files of the same step.  A rank that dies leaves the others with a TimeoutError
    after ``timeout_s``, not a hang.
    """

    backend = "file"

    def __init__(self, rank: int, world: int, directory: Optional[str] = None, local_rank: Optional[int] = None,
                 timeout_s: float = 600.0):
        if world < 1 or not (0 <= rank < world):
            raise ValueError("bad rank %d of %d" % (rank, world))
        self.rank, self.world = rank, world
        self.local_rank = rank if local_rank is None else local_rank
        self.dir = directory or default_rendezvous_dir()
        self.timeout_s = timeout_s
        self._seq = 0
        self._closed = False
        os.makedirs(self.dir, exist_ok=True)
        # a directory that already holds THIS rank's files belongs to an earlier attempt: they would let the
        # other ranks' barriers complete on stale data
        for name in os.listdir(self.dir):
            if name.endswith("_r%d.json" % rank) or name == "fin_r%d" % rank:
                try:
                    os.unlink(os.path.join(self.dir, name))
                except OSError:
                    pass

    def _path(self, seq: int, rank: int) -> str:
        return os.path.join(self.dir, "s%06d_r%d.json" % (seq, rank))

    def _wait_for(self, path: str):
        deadline = time.monotonic() + self.timeout_s
        spin_until = time.monotonic() + 0.002
        while not os.path.exists(path):
            now = time.monotonic()
            if now > deadline:
                raise TimeoutError("rank %d: no %s after %.0f s (a rank died?)" % (self.rank, path, self.timeout_s))
            if now > spin_until:
                time.sleep(0.0002)

    def all_gather(self, obj: Any) -> List[Any]:
        seq = self._seq
        self._seq += 1
        mine = self._path(seq, self.rank)
        tmp = mine + ".tmp%d" % os.getpid()
        with open(tmp, "w") as f:
            json.dump(obj, f)
        os.rename(tmp, mine)
        out = []
        for r in range(self.world):
            p = self._path(seq, r)
            self._wait_for(p)
            with open(p) as f:
                out.append(json.load(f))
        return out

    def barrier(self):
        self.all_gather(None)

    def max(self, value: float) -> float:
        return max(float(v) for v in self.all_gather(float(value)))

    def sum(self, value: float) -> float:
        return sum(float(v) for v in self.all_gather(float(value)))

    def close(self):
        """Last step: rank 0 removes the directory once every rank has said it is done reading."""
        if self._closed:
            return
        self._closed = True
        try:
            self.all_gather("fin")
            fin = os.path.join(self.dir, "fin_r%d" % self.rank)
            open(fin, "w").close()
            if self.rank == 0:
                for r in range(self.world):
                    self._wait_for(os.path.join(self.dir, "fin_r%d" % r))
                if not os.environ.get("GCN10_RDV_DIR"):     # a launcher-made directory is the launcher's to remove
                    shutil.rmtree(self.dir, ignore_errors=True)
        except (TimeoutError, OSError):
            pass


class TorchGroup:
    """The same operations on torch.distributed (``gloo`` on CPUs, ``nccl`` = RCCL on GPUs)."""

    def __init__(self, backend: str, device_index: Optional[int] = None):
        import torch
        import torch.distributed as dist
        self.rank, self.local_rank, self.world = world_from_env()
        self.backend = backend
        if backend == "nccl":
            idx = self.local_rank if device_index is None else device_index
            torch.cuda.set_device(idx)
            self._device = torch.device("cuda", idx)
        else:
            self._device = torch.device("cpu")
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
        self._dist, self._torch = dist, torch

    def barrier(self):
        self._dist.barrier()

    def all_gather(self, obj: Any) -> List[Any]:
        out: List[Any] = [None] * self.world
        self._dist.all_gather_object(out, obj)
        return out

    def _reduce(self, value: float, op) -> float:
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self._device)
        self._dist.all_reduce(t, op=op)
        return float(t.item())

    def max(self, value: float) -> float:
        return self._reduce(value, self._dist.ReduceOp.MAX)

    def sum(self, value: float) -> float:
        return self._reduce(value, self._dist.ReduceOp.SUM)

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None


def Group(backend: Optional[str] = None, device_index: Optional[int] = None):
    """The group of this process: solo at world size 1, files by default, torch.distributed on request
    (``backend`` or ``GCN10_DIST_BACKEND`` = ``gloo`` | ``nccl``)."""
    rank, local_rank, world = world_from_env()
    if world == 1:
        return SoloGroup()
    backend = backend or os.environ.get("GCN10_DIST_BACKEND") or "file"
    if backend == "file":
        # the file rendezvous is node-local: a job that spans nodes would wait out the time-out instead
        local_world = os.environ.get("LOCAL_WORLD_SIZE")
        if local_world and int(local_world) != world and not os.environ.get("GCN10_RDV_DIR"):
            raise RuntimeError("WORLD_SIZE %d != LOCAL_WORLD_SIZE %s: the default file rendezvous works on one node "
                               "only; set GCN10_DIST_BACKEND=gloo (or nccl), or GCN10_RDV_DIR to a directory all "
                               "nodes share" % (world, local_world))
        return FileGroup(rank, world, local_rank=local_rank)
    if backend in ("gloo", "nccl"):
        return TorchGroup(backend, device_index)
    raise ValueError("unknown GCN10_DIST_BACKEND %r (file | gloo | nccl)" % backend)
