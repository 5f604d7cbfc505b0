"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

Round 1 ships the *replica* mode of BASELINE.json config 5: independent clips, one per rank, no data-path
collective — `shard_clips` assigns clips, `max_over_ranks` produces the whole-job wall time bench.py reports.
`frame_shards` is the partition the frame-sharded single-clip mode (config 4) will use: the 2·T CFG x frame
instances are split contiguously, so that a rank's shard is one contiguous row range of every
frames-outermost activation matrix (halo / all-gather exchanges are contiguous slabs).
"""
from __future__ import annotations

from typing import List, Tuple

import torch


def shard_clips(n_clips: int, rank: int, world: int) -> List[int]:
    """Round-robin clip ids of this rank (clip i -> rank i % world), as the reference script's chunk loop would."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world {world}")
    return list(range(rank, n_clips, world))


def frame_shards(n_instances: int, world: int) -> List[Tuple[int, int]]:
    """Contiguous [start, stop) ranges of the B*T frame instances; sizes differ by at most one."""
    base, rem = divmod(n_instances, world)
    out, s = [], 0
    for r in range(world):
        n = base + (1 if r < rem else 0)
        out.append((s, s + n))
        s += n
    return out


def sharding_efficiency(n_instances: int, world: int) -> float:
    """Upper bound on per-GPU efficiency of whole-frame sharding: mean shard / max shard."""
    sizes = [b - a for a, b in frame_shards(n_instances, world)]
    return (sum(sizes) / len(sizes)) / max(sizes)


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX all-reduce of a wall-clock interval (identity when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())
