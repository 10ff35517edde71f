"""Block distribution across the GPUs of one node.

The reference deals block ids round-robin over MPI ranks
(``for (i = rank; i < n_blocks; i += size)``, src/main.c:171) and exchanges no
raster data; its only collective is the closing MPI_Barrier (src/main.c:187).
Here one process drives one GPU; blocks are independent, so the data path has
no collective at all.  ``torch.distributed`` (RCCL on GPUs, gloo on CPUs) is
used only for what MPI was used for: rank identity, a barrier and, for
benchmarking, a max-reduction of the elapsed time.
"""
from __future__ import annotations

import os
from typing import List, Sequence


def world_from_env():
    """(rank, local_rank, world_size) as torch.distributed.run exports them."""
    return (int(os.environ.get("RANK", "0")), int(os.environ.get("LOCAL_RANK", "0")),
            int(os.environ.get("WORLD_SIZE", "1")))


def blocks_for_rank(block_ids: Sequence[int], rank: int, world: int) -> List[int]:
    """Static round-robin share of a block list, exactly src/main.c:171."""
    if world < 1 or not (0 <= rank < world):
        raise ValueError("bad rank %d of %d" % (rank, world))
    return [block_ids[i] for i in range(rank, len(block_ids), world)]


class Group:
    """A thin, optional torch.distributed wrapper (no torch import at world size 1)."""

    def __init__(self, backend: str | None = None, device_index: int | None = None):
        self.rank, self.local_rank, self.world = world_from_env()
        self._dist = None
        self._device = None
        if self.world > 1:
            import torch
            import torch.distributed as dist
            # GCN10_DIST_BACKEND=gloo lets several ranks rehearse on one GPU (tests)
            backend = backend or os.environ.get("GCN10_DIST_BACKEND") or \
                ("nccl" if torch.cuda.is_available() else "gloo")
            if backend == "nccl":
                idx = self.local_rank if device_index is None else device_index
                torch.cuda.set_device(idx)
                self._device = torch.device("cuda", idx)
            else:
                self._device = torch.device("cpu")
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group(backend=backend, rank=self.rank, world_size=self.world)
            self._dist = dist
            self._torch = torch

    def barrier(self):
        if self._dist is not None:
            self._dist.barrier()

    def max(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.MAX)
        return float(t.item())

    def sum(self, value: float) -> float:
        if self._dist is None:
            return float(value)
        t = self._torch.tensor([value], dtype=self._torch.float64, device=self._device)
        self._dist.all_reduce(t, op=self._dist.ReduceOp.SUM)
        return float(t.item())

    def close(self):
        if self._dist is not None:
            self._dist.destroy_process_group()
            self._dist = None
