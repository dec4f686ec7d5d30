"""Replica sharding of independent sequences over the GPUs of one node.

The reference has no multi-GPU code: "batch" = independent sequences with separate state slices
(src/runtime/v7.rs:519-521).  They shard with NO data-path collective: every rank holds a full weight
replica and owns the streams `b` with `b % world_size == rank`; the only communication is the timing
barrier / max-reduce used by bench.py and the gather of generated tokens for callers that want them.
Backend: torch.distributed ("nccl" == RCCL on ROCm for GPU runs, "gloo" in the CPU tests).
"""
from __future__ import annotations

from typing import List, Optional, Sequence


def partition_streams(num_streams: int, world_size: int) -> List[List[int]]:
    """stream b -> rank b % world_size (contiguous per rank after sorting)."""
    if world_size < 1:
        raise ValueError("world_size must be >= 1")
    return [list(range(r, num_streams, world_size)) for r in range(world_size)]


class ReplicaGroup:
    """Thin wrapper over torch.distributed for the replica pattern; a no-op when world_size == 1."""

    def __init__(self, dist=None, device: Optional[str] = None):
        self.dist = dist
        self.device = device
        self.rank = dist.get_rank() if dist is not None else 0
        self.world_size = dist.get_world_size() if dist is not None else 1

    def my_streams(self, num_streams: int) -> List[int]:
        return partition_streams(num_streams, self.world_size)[self.rank]

    def barrier(self) -> None:
        if self.dist is not None:
            self.dist.barrier()

    def max_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
        return float(t.item())

    def sum_over_ranks(self, value: float) -> float:
        if self.dist is None:
            return float(value)
        import torch
        t = torch.tensor([float(value)], dtype=torch.float64, device=self.device or "cpu")
        self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM)
        return float(t.item())

    def gather_tokens(self, tokens: Sequence[Sequence[int]], num_streams: int) -> Optional[List[List[int]]]:
        """Collect per-stream token lists on rank 0 in global stream order (host side, not timed)."""
        mine = dict(zip(self.my_streams(num_streams), [list(map(int, t)) for t in tokens]))
        if self.dist is None:
            return [mine[b] for b in range(num_streams)]
        out = [None] * self.world_size if self.rank == 0 else None
        self.dist.gather_object(mine, out, dst=0)
        if self.rank != 0:
            return None
        merged = {}
        for d in out:
            merged.update(d)
        return [merged[b] for b in range(num_streams)]
