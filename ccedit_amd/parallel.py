"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

Round 1 ships the *replica* mode of BASELINE.json config 5: independent clips, one per rank, no data-path
collective — `shard_clips` assigns clips, `max_over_ranks` produces the whole-job wall time bench.py reports.
`frame_shards` is the partition the frame-sharded single-clip mode (config 4) will use: the 2·T CFG x frame
instances are split contiguously, so that a rank's shard is one contiguous row range of every
frames-outermost activation matrix (halo / all-gather exchanges are contiguous slabs).
"""
from __future__ import annotations

from typing import List, Optional, Tuple

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


class FrameShard:
    """One clip's T keyframes split contiguously over the ranks of a process group (BASELINE.json config 4).

    Every rank holds frames [t0, t1) of EVERY clip of the (CFG-doubled) batch, i.e. a (B, t_local, H, W, C) slab of
    each frames-outermost activation.  Spatial work is frame-local.  The three kinds of temporal work exchange:
      * Conv1d k=3 over T     -> `halo`: one boundary frame to/from each neighbour rank (point-to-point)
      * GroupNorm over (C/32 x T) -> `allreduce`: per-(clip, pixel, group) sum / sum-of-squares (fp32)
      * temporal attention    -> `gather_frames`: all-gather of the K/V rows (RCCL all-gather over xGMI)
    Collectives go through torch.distributed: backend "nccl" (= RCCL on ROCm) moves device tensors directly;
    with "gloo" (CPU tests, or several ranks sharing one GPU) tensors are staged through host memory.
    """

    def __init__(self, t_glob: int, rank: Optional[int] = None, world: Optional[int] = None, group=None):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        if self.world > t_glob:
            raise ValueError(f"cannot shard {t_glob} keyframes over {self.world} ranks")
        self.t_glob = t_glob
        self.bounds = frame_shards(t_glob, self.world)
        self.t0, self.t1 = self.bounds[self.rank]
        self.t_local = self.t1 - self.t0
        self.t_max = max(b - a for a, b in self.bounds)
        self.staged = dist.get_backend(group) != "nccl"
        self.bytes_sent = 0

    # -- helpers --------------------------------------------------------------------------------
    def _out(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().cpu().contiguous() if (self.staged and t.is_cuda) else t.contiguous()

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the ranks."""
        if self.staged and t.is_cuda:
            h = t.detach().cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)
        self.bytes_sent += t.numel() * t.element_size()
        return t

    def halo(self, first: torch.Tensor, last: torch.Tensor):
        """Send my first local frame to rank-1 and my last to rank+1; return (frame before my first, frame after my
        last) — None at the clip ends.  first/last: contiguous tensors of identical shape."""
        dist = self.dist
        dev = first.device
        ops_, prev_buf, next_buf = [], None, None
        f_out, l_out = self._out(first), self._out(last)
        if self.rank > 0:
            prev_buf = torch.empty_like(f_out)
            peer = self._global_rank(self.rank - 1)      # P2POp peers are GLOBAL ranks, also inside a sub-group
            ops_.append(dist.P2POp(dist.isend, f_out, peer, self.group))
            ops_.append(dist.P2POp(dist.irecv, prev_buf, peer, self.group))
        if self.rank < self.world - 1:
            next_buf = torch.empty_like(l_out)
            peer = self._global_rank(self.rank + 1)
            ops_.append(dist.P2POp(dist.isend, l_out, peer, self.group))
            ops_.append(dist.P2POp(dist.irecv, next_buf, peer, self.group))
        if ops_:
            for r in dist.batch_isend_irecv(ops_):
                r.wait()
        self.bytes_sent += (int(self.rank > 0) + int(self.rank < self.world - 1)) * first.numel() * first.element_size()
        prev = None if prev_buf is None else prev_buf.to(dev)
        nxt = None if next_buf is None else next_buf.to(dev)
        return prev, nxt

    def owner_of(self, t_global: int) -> int:
        """Rank holding global keyframe t_global."""
        for r, (lo, hi) in enumerate(self.bounds):
            if lo <= t_global < hi:
                return r
        raise ValueError(f"keyframe {t_global} outside [0, {self.t_glob})")

    def broadcast(self, t: torch.Tensor, src: int) -> torch.Tensor:
        """In-place broadcast of a contiguous tensor from rank `src` (TVI2V: the centre keyframe's K/V rows)."""
        if self.staged and t.is_cuda:
            h = t.detach().cpu()
            self.dist.broadcast(h, src=self._global_rank(src), group=self.group)
            if self.rank != src:
                t.copy_(h)
        else:
            self.dist.broadcast(t, src=self._global_rank(src), group=self.group)
        if self.rank == src:
            self.bytes_sent += t.numel() * t.element_size()
        return t

    def _global_rank(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def gather_frames(self, x: torch.Tensor, b: int) -> torch.Tensor:
        """x: (b * t_local, ...) local frames of every clip -> (b * t_glob, ...) with all ranks' frames in order."""
        rest = x.shape[1:]
        xl = x.reshape(b, self.t_local, *rest)
        if self.t_local < self.t_max:          # equal-size all-gather: pad short shards
            pad = torch.zeros((b, self.t_max - self.t_local, *rest), dtype=x.dtype, device=x.device)
            xl = torch.cat([xl, pad], dim=1)
        send = self._out(xl)
        bufs = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(bufs, send, group=self.group)
        self.bytes_sent += send.numel() * send.element_size()
        parts = [bufs[r][:, : (hi - lo)] for r, (lo, hi) in enumerate(self.bounds)]
        full = torch.cat(parts, dim=1).to(x.device)
        return full.reshape(b * self.t_glob, *rest)
