"""One process per GPU: shard the unique reads, replicate the target database, gather hits.

The path has no exchange step: every read is matched independently against the replicated
database, so the only collective is the concatenation of the per-rank hit
lists on rank 0 (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" on CPU
for the tests).  Because the reference orders results.txt by read sequence first
(cmd/muscato/main.go:657-659) and shards are contiguous ranges of the sorted unique reads,
concatenating shards in rank order keeps the global order.
"""
from __future__ import annotations

from typing import List, Optional, Tuple

import torch
import torch.distributed as dist


def shard_range(n: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous slice [lo, hi) of n sorted unique reads owned by `rank`."""
    lo = (n * rank) // world
    hi = (n * (rank + 1)) // world
    return lo, hi


def gather_hits(local: torch.Tensor, read_base: int, dst: int = 0, group=None) -> Optional[torch.Tensor]:
    """local: int32 [n_local, 4] hits with shard-local read_idx (any device).  Adds
    `read_base` to column 0 and concatenates all ranks' hits on rank `dst` in rank order.
    Returns the [n_total, 4] tensor on `dst`, None elsewhere.

    One all_gather of the counts (8 B per rank) and one padded gather of the tuples."""
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    local = local.reshape(-1, 4)
    if read_base:
        local = local.clone()
        local[:, 0] += read_base
    if world == 1:
        return local
    dev = local.device
    cnt = torch.tensor([local.shape[0]], dtype=torch.int64, device=dev)
    cnts = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(cnts, cnt, group=group)
    counts = [int(c.item()) for c in cnts]
    mx = max(max(counts), 1)
    padded = torch.zeros((mx, 4), dtype=local.dtype, device=dev)
    padded[:local.shape[0]] = local
    if rank == dst:
        bufs: List[torch.Tensor] = [torch.empty((mx, 4), dtype=local.dtype, device=dev) for _ in range(world)]
        dist.gather(padded, gather_list=bufs, dst=dst, group=group)
        return torch.cat([b[:c] for b, c in zip(bufs, counts)], dim=0)
    dist.gather(padded, gather_list=None, dst=dst, group=group)
    return None
