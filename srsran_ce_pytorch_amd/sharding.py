"""Slot-batch sharding across the GPUs of a node: slots are independent, so each rank
owns a contiguous range of the slot axis (all Rx ports of a slot stay together) and the data path needs no
collective.  `torch.distributed` is used for the bench's barrier / max-over-ranks timing only."""
from __future__ import annotations

from typing import Tuple

import torch
import torch.distributed as dist


def shard_slots(n_slots: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [start, stop) of the slot axis owned by `rank`; sizes differ by at most one."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    base, extra = divmod(n_slots, world)
    start = rank * base + min(rank, extra)
    return start, start + base + (1 if rank < extra else 0)


def max_over_ranks(values, device="cpu"):
    """Elementwise MAX of a list of floats over all ranks (measurement only, never data path)."""
    t = torch.tensor(list(values), dtype=torch.float64, device=device)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return [float(x) for x in t]


def aggregate_slots_per_second(slots_per_rank: int, steps: int, elapsed_s: float, world: int) -> float:
    """Whole-job throughput: slots all ranks processed / max-over-ranks wall time."""
    return world * slots_per_rank * steps / elapsed_s
