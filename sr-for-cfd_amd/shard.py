"""Sample sharding for multi-GPU inference: independent samples, no data-path
collective.  One process per GPU (`torch.distributed`, backend "nccl" = RCCL on ROCm; "gloo"
in the CPU tests); the only exchange is the scalar reduction of the timed region."""
from __future__ import annotations

from typing import Tuple


def shard_range(n_total: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous [lo, hi) block of `n_total` samples for `rank`; sizes differ by at most one."""
    if world <= 0 or not (0 <= rank < world) or n_total < 0:
        raise ValueError("bad shard arguments")
    base, rem = divmod(n_total, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX-reduce a per-rank duration; identity when no process group is up."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return float(seconds)
    t = torch.tensor([seconds], dtype=torch.float64, device=device or "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def aggregate_throughput(units_per_rank: int, world: int, seconds_max: float) -> float:
    """Whole-job units/s for weak scaling: every rank processed `units_per_rank` in `seconds_max`."""
    return units_per_rank * world / seconds_max
