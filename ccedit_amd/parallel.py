"""Multi-GPU plumbing (one process per GPU, torch.distributed; backend "nccl" == RCCL on ROCm, "gloo" in CPU tests).

Two modes:
  * replicas (BASELINE.json config 5): independent clips, one per rank, no data-path collective — `shard_clips`
    assigns clips, `max_over_ranks` produces the whole-job wall time bench.py reports;
  * frame sharding (config 4): ONE clip, its T keyframes split contiguously over the ranks (`FrameShard`).  Spatial
    work is frame-local.  Temporal work (Conv1d / GroupNorm / attention over T, every pixel independent) runs in the
    TRANSPOSED layout — all T frames of 1/world of the pixels — reached by one all-to-all and left by another
    (mode "a2a", the default): xGMI is a full mesh, so an all-to-all drives all 7 links of a GPU at once where a
    neighbour halo drives 2, and the temporal attention moves C values per token instead of all-gathering 2C x T.
    Mode "halo" is the round-1 scheme (halo p2p + statistics all-reduce + K/V all-gather), kept for comparison.
  The two CFG halves of a step can run on `cfg_pair()` shards whose uneven remainders sit at opposite ends of the
  rank list, so 2 x 17 frame instances spread 5/4/4/4/4/4/4/5 over 8 ranks (ceiling 0.85 instead of 0.71).
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


def cfg_pair_efficiency(t_glob: int, world: int) -> float:
    """The same bound for FrameShard.cfg_pair: per rank, shard of half 0 + mirrored shard of half 1."""
    sizes = [b - a for a, b in frame_shards(t_glob, world)]
    both = [a + b for a, b in zip(sizes, sizes[::-1])]
    return (sum(both) / len(both)) / max(both)


def row_sharding_efficiency(latent_rows: int, world: int) -> float:
    """Load-balance ceiling of RowShard: every rank holds latent_rows / world rows of every frame at the top level and 1/2, 1/4, 1/8
    of that below — 1.0 whenever the deepest level (latent_rows / 8) still divides, which is the only case RowShard accepts."""
    if latent_rows % (8 * world):
        raise ValueError(f"{latent_rows} latent rows: the deepest level has {latent_rows // 8}, not divisible by {world} ranks")
    return 1.0


def max_over_ranks(seconds: float, device=None) -> float:
    """MAX all-reduce of a wall-clock interval (identity when torch.distributed is not initialised)."""
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=device if device is not None else "cpu")
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


class _ShardComm:
    """Communicator plumbing shared by the two single-clip decompositions (FrameShard: keyframes, RowShard: latent rows): rank / world
    of a process group, host staging for non-RCCL backends, byte / collective counters, device events around every exchange and the
    host-issue-order log the tests compare across ranks."""

    issue_log = None       # class-wide: a list collects (partition, kind, elements) of every collective in HOST ISSUE ORDER — all
    #                        shards of the process append to the same list, so the interleaving of two communicators is visible.
    #                        Ranks whose logs differ would deadlock on real links; tests compare them across ranks.

    def _init_comm(self, rank, world, group):
        import torch.distributed as dist
        self.dist = dist
        self.group = group
        self.rank = dist.get_rank(group) if rank is None else rank
        self.world = dist.get_world_size(group) if world is None else world
        self.staged = dist.get_backend(group) != "nccl"
        self.bytes_sent = 0
        self.n_collectives = 0
        self.timing = None                     # a list: (start, stop) device events around every exchange (bench.py)

    def _log_partition(self) -> int:
        return 0

    def _tick(self, t: torch.Tensor, kind: str = "a2a"):
        self.n_collectives += 1
        log = self.issue_log                   # (class attribute of the shard's class, or of the base)
        if log is not None:
            log.append((self._log_partition(), kind, int(t.numel())))
        if self.timing is None or not t.is_cuda:
            return None
        ev = torch.cuda.Event(enable_timing=True)
        ev.record()
        return ev

    def _tock(self, ev):
        if ev is not None:
            stop = torch.cuda.Event(enable_timing=True)
            stop.record()                      # on the stream that waits for the collective's result
            self.timing.append((ev, stop))

    def comm_ms(self) -> float:
        """Sum of the recorded exchange durations (call after a device synchronize)."""
        return sum(a.elapsed_time(b) for a, b in (self.timing or []))

    def reset_counters(self):
        self.bytes_sent, self.n_collectives = 0, 0
        if self.timing is not None:
            self.timing = []

    def all_agree(self, ok: bool) -> bool:
        """True iff `ok` holds on EVERY rank of the communicator (a MIN all-reduce of one flag through host memory, not counted as a
        data-path exchange).  Used where ranks must take the same branch before issuing further collectives — e.g. whether the HIP-graph
        capture of a sharded evaluation succeeded everywhere (network._forward_graphed)."""
        flag = torch.tensor([1 if ok else 0], dtype=torch.int32)
        if not self.staged:
            flag = flag.cuda()
        self.dist.all_reduce(flag, op=self.dist.ReduceOp.MIN, group=self.group)
        return bool(int(flag.item()) == 1)

    def _out(self, t: torch.Tensor) -> torch.Tensor:
        return t.detach().cpu().contiguous() if (self.staged and t.is_cuda) else t.contiguous()

    def _global_rank(self, r: int) -> int:
        return r if self.group is None else self.dist.get_global_rank(self.group, r)

    def allreduce(self, t: torch.Tensor) -> torch.Tensor:
        """In-place SUM over the ranks."""
        ev = self._tick(t, "allreduce")
        if self.staged and t.is_cuda:
            h = t.detach().cpu()
            self.dist.all_reduce(h, group=self.group)
            t.copy_(h)
        else:
            self.dist.all_reduce(t, group=self.group)
        self._tock(ev)
        self.bytes_sent += t.numel() * t.element_size()
        return t

    def _neighbours(self, to_prev: torch.Tensor, to_next: torch.Tensor, kind: str, from_next: bool = True):
        """Send `to_prev` to rank - 1 and `to_next` to rank + 1; return (received from rank - 1, received from rank + 1), None at
        the ends of the rank list.  Contiguous tensors of identical shape; point-to-point, both directions in one batch.
        from_next=False: only the downward direction (every rank sends `to_next` and receives from rank - 1 — what a stride-2
        convolution needs); `to_prev` is then neither sent nor counted."""
        dist = self.dist
        dev = to_next.device
        ops_, prev_buf, next_buf = [], None, None
        ev = self._tick(to_next, kind)
        f_out, l_out = (self._out(to_prev) if from_next else None), self._out(to_next)
        if self.rank > 0:
            prev_buf = torch.empty_like(l_out)
            peer = self._global_rank(self.rank - 1)      # P2POp peers are GLOBAL ranks, also inside a sub-group
            if from_next:
                ops_.append(dist.P2POp(dist.isend, f_out, peer, self.group))
            ops_.append(dist.P2POp(dist.irecv, prev_buf, peer, self.group))
        if self.rank < self.world - 1:
            peer = self._global_rank(self.rank + 1)
            ops_.append(dist.P2POp(dist.isend, l_out, peer, self.group))
            if from_next:
                next_buf = torch.empty_like(l_out)
                ops_.append(dist.P2POp(dist.irecv, next_buf, peer, self.group))
        if ops_:
            for r in dist.batch_isend_irecv(ops_):
                r.wait()
        self._tock(ev)
        self.bytes_sent += ((int(self.rank > 0) if from_next else 0) + int(self.rank < self.world - 1)) * to_next.numel() * to_next.element_size()
        prev = None if prev_buf is None else prev_buf.to(dev)
        nxt = None if next_buf is None else next_buf.to(dev)
        return prev, nxt


class RowShard(_ShardComm):
    """ONE clip, the latent ROWS of every frame split contiguously over the ranks (BASELINE.json config 4 — the balanced decomposition:
    the reference has no multi-GPU path, scripts/sampling/sampling_tv2v.py:106 is a single .to("cuda")).

    Every rank holds all B * T frames of rows [r h / N, (r + 1) h / N) at every level of the networks (h = 64, 32, 16, 8 latent rows
    at 512 x 768: any N in {2, 4, 8} divides them all), i.e. a (B * T, h / N, w, C) slab of each frames-outermost activation — 1 / N of
    the work of EVERY kernel, whatever T is (whole keyframes of T = 17 over 8 ranks cap at 0.71 / 0.85, `sharding_efficiency`).
      * temporal operators (Conv1d over T, GroupNorm over C/32 x T, temporal attention) see all T frames of their pixels: LOCAL, no
        exchange — the 272 transpositions per step of FrameShard's pair mode do not exist here;
      * 3x3 convolutions need the neighbour ranks' boundary rows: `halo_exchange` receives one row from above and one from below
        (point-to-point) into two small tensors the conv kernel reads IN PLACE (CcGemmDesc.vpad = 2, halo_top / halo_bot: no
        extended copy of the slab; a stride-2 convolution only needs — and only exchanges — the row above);
      * spatial GroupNorm: local (sum, sum of squares) per (frame, group), `gn_stats` all-reduces the 32 doubles per frame;
      * spatial self-attention, attn = "heads" (default): the 8 heads are split over the ranks instead of the rows — one
        all-to-all brings every rank the WHOLE frame of q, k and v for its 8 / N heads (`to_heads`), it runs the full-frame attention
        for them, and one all-to-all returns the outputs to the ranks that own the rows (`from_heads`): q, k, v, o each cross the
        links once, (N - 1) / N of them — a quarter of what all-gathering K and V costs.  attn = "gather": the RCCL all-gather of
        the K / V rows BASELINE.json names (`gather_rows`), local queries against the gathered frame; also the fallback when the
        heads do not divide by N.  Text cross-attention is local in both;
      * the TVI2V anchor frame is a frame like any other — its K / V rows arrive with everything else, no broadcast.
    Collectives go through torch.distributed as in FrameShard (nccl = RCCL, gloo staged through the host).  Pack / unpack around the
    exchanges are HIP copy kernels with index plans cached per shape (ccedit_copy_2d_blocks / ccedit_copy_row_blocks) writing into
    recycled slabs: no torch.cat, no allocation per exchange; with RCCL the whole evaluation is captured into a HIP graph like the
    single-GPU one."""

    mode = "rows"

    def __init__(self, rank: Optional[int] = None, world: Optional[int] = None, group=None, attn: str = "heads"):
        if attn not in ("heads", "gather"):
            raise ValueError(f"RowShard attn {attn!r}: 'heads' or 'gather'")
        self.attn = attn
        self._init_comm(rank, world, group)
        self._plans = {}
        self._slabs = {}

    def rows(self, h: int) -> Tuple[int, int]:
        """[r0, r1) of this rank at a level with h rows."""
        if h % self.world:
            raise ValueError(f"{h} rows cannot be split evenly over {self.world} ranks (the frame height must be a multiple of "
                             f"{8 * 8 * self.world} pixels: three stride-2 levels below the latent)")
        n = h // self.world
        return self.rank * n, (self.rank + 1) * n

    def check_latent(self, h: int):
        """The latent height must split evenly at EVERY level of the networks (h, h/2, h/4, h/8 rows): h % (8 * world) == 0.  `rows`
        alone would accept 48 rows over 4 ranks and fail two levels down with an odd local height."""
        if h % (8 * self.world):
            raise ValueError(f"{h} latent rows cannot be row-sharded over {self.world} ranks: the deepest level has {h // 8} rows "
                             f"(the frame height must be a multiple of {64 * self.world} pixels)")
        return row_sharding_efficiency(h, self.world)

    _sibling = None

    @staticmethod
    def cfg_pair(groups=(None, None), attn: str = "heads"):
        """Two RowShards over the same ranks, one per CFG half of a step (uc, c): the wrapper evaluates the halves as two B = 1 passes
        on two HIP streams (network._forward_eager), each with its own communicator, so the exchanges of one half are hidden behind
        the kernels of the other.  `groups`: one process group per half; the second is created here over the same ranks when it is
        not given (`dist.new_group` is collective: every rank calls cfg_pair at the same point)."""
        g0, g1 = groups
        if g1 is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(g0) > 1:
                ranks = list(range(dist.get_world_size())) if g0 is None else dist.get_process_group_ranks(g0)
                g1 = dist.new_group(ranks)
        a, b = RowShard(group=g0, attn=attn), RowShard(group=g1, attn=attn)
        b._log_part = 2
        return (a, b)

    def sibling(self) -> "RowShard":
        """A second RowShard over the SAME ranks on its own communicator, for work issued on a side stream (the ControlNet beside the
        UNet encoder): a communicator serialises its collectives on one internal stream, and two HIP streams that fork and join
        inside one captured graph must not share it.  `dist.new_group` is collective — every rank reaches this on its first
        sharded evaluation."""
        if self._sibling is None:
            ranks = list(range(self.dist.get_world_size())) if self.group is None else self.dist.get_process_group_ranks(self.group)
            self._sibling = RowShard(group=self.dist.new_group(ranks), attn=self.attn)
            self._sibling._log_part = 1
        return self._sibling

    _log_part = 0

    def _log_partition(self) -> int:
        return self._log_part

    def reset_counters(self):
        super().reset_counters()
        if self._sibling is not None:
            self._sibling.reset_counters()

    def can_capture(self) -> bool:
        """May an evaluation that uses this shard be captured into a HIP graph?  RCCL collectives are stream operations and capture;
        the host-staged transport (gloo) copies through host memory and cannot."""
        return not self.staged

    # -- slabs: wire-side scratch, two alternating buffers per role (see FrameShard._slab for why recycling is safe) -----------------
    def _slab(self, tag: str, shape, like: torch.Tensor) -> torch.Tensor:
        key = (tag, tuple(shape), like.dtype, str(like.device))
        ent = self._slabs.get(key)
        if ent is None:
            ent = [[torch.empty(shape, dtype=like.dtype, device=like.device) for _ in range(2)], 0]
            self._slabs[key] = ent
        ent[1] ^= 1
        return ent[0][ent[1]]

    # -- 3x3 convolutions ---------------------------------------------------------------------------------------------------------
    def halo_exchange(self, x: torch.Tensor, below: bool = True):
        """x (n, h_local, w, C) -> (top, bottom): the last row of rank - 1 and (with `below`) the first row of rank + 1, each
        (n, w, C) contiguous, None where the frame ends (the conv kernel reads zeros there).  The slab itself is not touched."""
        n, h, w, c = x.shape
        up, down = self._neighbours(x[:, 0].contiguous() if below else None, x[:, h - 1].contiguous(), "halo_rows", from_next=below)
        return up, (down if below else None)

    def halo_stats_exchange(self, x: torch.Tensor, stats: torch.Tensor):
        """GroupNorm(32) [+ SiLU] followed by a 3x3 convolution, ONE exchange instead of two (round 6): the RAW boundary rows of `x`
        go to the two neighbours and this rank's partial (sum, sum of squares) per (frame, group) to every rank, all as point-to-point
        messages of one batch (an all-gather of the 17 KB of partials riding with the halo rows: the all-reduce it replaces is pure
        latency, ~15-20 us, 64 times per step).  Returns (top, bottom, total): the neighbours' raw rows ((n, w, C) or None at the ends
        of the frame) and the frame's statistics as the SUM of the partials in rank order — the same bits on every rank.  The caller
        normalises its slab and the two received rows with `total` (network.sgn_conv3)."""
        dist = self.dist
        n, h, w, c = x.shape
        dev = x.device
        ev = self._tick(x[:, 0], "halo_stats")
        first, last, st = self._out(x[:, 0].contiguous()), self._out(x[:, h - 1].contiguous()), self._out(stats)
        parts = [st if r == self.rank else torch.empty_like(st) for r in range(self.world)]
        ops_, prev_buf, next_buf = [], None, None
        for r in range(self.world):                       # same op order on every rank pair: (send to r, receive from r) by ascending r
            if r == self.rank:
                continue
            peer = self._global_rank(r)
            if r == self.rank - 1:
                prev_buf = torch.empty_like(last)
                ops_ += [dist.P2POp(dist.isend, first, peer, self.group), dist.P2POp(dist.irecv, prev_buf, peer, self.group)]
            elif r == self.rank + 1:
                next_buf = torch.empty_like(first)
                ops_ += [dist.P2POp(dist.isend, last, peer, self.group), dist.P2POp(dist.irecv, next_buf, peer, self.group)]
            ops_ += [dist.P2POp(dist.isend, st, peer, self.group), dist.P2POp(dist.irecv, parts[r], peer, self.group)]
        if ops_:
            for r_ in dist.batch_isend_irecv(ops_):
                r_.wait()
        self._tock(ev)
        nb = int(self.rank > 0) + int(self.rank < self.world - 1)
        self.bytes_sent += nb * first.numel() * first.element_size() + (self.world - 1) * st.numel() * st.element_size()
        total = parts[0].to(dev).clone()
        for r in range(1, self.world):
            total += parts[r].to(dev)
        return (None if prev_buf is None else prev_buf.to(dev)), (None if next_buf is None else next_buf.to(dev)), total

    def halo_rows(self, x: torch.Tensor, below: bool = True) -> torch.Tensor:
        """The same exchange as ONE extended tensor (n, 1 + h_local + below, w, C) with zero rows at the ends of the frame — what the
        conv kernels take with CcGemmDesc.vpad = 1.  Costs a copy of the slab; the network uses `halo_exchange` (vpad = 2), this form
        remains for the tests that pin the two against each other."""
        n, h, w, c = x.shape
        up, down = self.halo_exchange(x, below)
        out = x.new_zeros((n, h + 1 + int(below), w, c))
        out[:, 1:h + 1] = x
        if up is not None:
            out[:, 0] = up
        if below and down is not None:
            out[:, h + 1] = down
        return out

    def gn_stats(self, stats: torch.Tensor) -> torch.Tensor:
        """Local (sum, sum of squares) per (frame, group) -> the frame's, scaled by 1 / world: ccedit_groupnorm_spatial_apply divides
        by the LOCAL element count, and (sum / N) / (count / N) is the mean over the whole frame (exact for N a power of two)."""
        self.allreduce(stats)
        stats.mul_(1.0 / self.world)
        return stats

    # -- equal-split all-to-all / all-gather ------------------------------------------------------------------------------------
    def _a2a_equal(self, send: torch.Tensor, out: torch.Tensor, kind: str) -> torch.Tensor:
        ev = self._tick(send, kind)
        if self.staged and send.is_cuda:
            h_out = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_to_all_single(h_out, send.detach().cpu(), group=self.group)
            out.copy_(h_out)
        else:
            self.dist.all_to_all_single(out, send, group=self.group)
        self._tock(ev)
        self.bytes_sent += send.numel() * send.element_size() * (self.world - 1) // self.world
        return out

    def _row_perm(self, src: torch.Tensor, dst: torch.Tensor, blocks_cpu, key, max_rows: int) -> torch.Tensor:
        """dst rows = a block permutation of src rows (2-D tensors of one row width): HIP copy kernel on the device, index copy on
        the CPU (gloo tests of the plans)."""
        if src.is_cuda:
            from . import ops
            pl = self._plans.get(key)
            if pl is None:
                pl = self._plans[key] = torch.tensor(blocks_cpu, dtype=torch.int64, device=src.device)
            return ops.copy_row_blocks(src, dst, pl, max_rows)
        for s0, d0, nr in blocks_cpu:
            dst[d0:d0 + nr] = src[s0:s0 + nr]
        return dst

    def _col_blocks(self, src: torch.Tensor, dst: torch.Tensor, blocks_cpu, key, rows: int, row_elems: int) -> torch.Tensor:
        """For every block (src element offset, dst element offset): `rows` rows of `row_elems` elements, row pitches = the tensors'
        row strides — column slices <-> contiguous buffers."""
        if src.is_cuda:
            from . import ops
            pl = self._plans.get(key)
            if pl is None:
                es = src.element_size()
                pl = self._plans[key] = torch.tensor([(a * es, b * es) for a, b in blocks_cpu], dtype=torch.int64, device=src.device)
            return ops.copy_2d_blocks(src, dst, pl, rows, row_elems * src.element_size())
        sf, df = src.reshape(-1), dst.reshape(-1)
        ps, pd = src.stride(0), dst.stride(0)
        for a, b in blocks_cpu:
            sv = torch.as_strided(sf, (rows, row_elems), (ps, 1), a)
            torch.as_strided(df, (rows, row_elems), (pd, 1), b).copy_(sv)
        return dst

    # -- spatial self-attention, head-parallel ------------------------------------------------------------------------------------
    def heads_ok(self, heads: int) -> bool:
        return self.attn == "heads" and heads % self.world == 0

    def to_heads(self, parts, frames: int, p_local: int, cw: int):
        """parts: [(2-D tensor (frames * p_local, >= cols), first column)] — e.g. q, k, v as column blocks of one projection, each C
        wide.  Every rank's rows of the SAME cw columns (its heads: columns [r cw, (r + 1) cw) of each part) are brought together:
        returns [len(parts)] tensors (frames * world * p_local, cw), frame-major, the ranks' row blocks in rank order = whole frames
        of this rank's heads."""
        w_, np_ = self.world, len(parts)
        rows = frames * p_local
        like = parts[0][0]
        send = self._slab("hs", (w_ * np_ * rows, cw), like)
        for j, (t2d, c0) in enumerate(parts):        # block (dest r, part j): columns c0 + r cw .. of every local row
            assert t2d.shape[0] == rows and t2d.stride(1) == 1
            blocks = [(c0 + r * cw, ((r * np_ + j) * rows) * cw) for r in range(w_)]
            self._col_blocks(t2d, send, blocks, ("hpack", j, c0, rows, cw, t2d.stride(0), str(like.device)), rows, cw)
        recv = self._a2a_equal(send, self._slab("hr", (w_ * np_ * rows, cw), like), "to_heads")       # [src s][part][frame][p]
        full = self._slab("hf", (np_ * frames * w_ * p_local, cw), like)                               # [part][frame][s][p]
        blocks = [(((s * np_ + j) * frames + f) * p_local, ((j * frames + f) * w_ + s) * p_local, p_local)
                  for s in range(w_) for j in range(np_) for f in range(frames)]
        self._row_perm(recv, full, blocks, ("hunpack", np_, frames, p_local, cw, str(like.device)), p_local)
        n_ = frames * w_ * p_local
        return [full[j * n_:(j + 1) * n_] for j in range(np_)]

    def from_heads(self, o: torch.Tensor, frames: int, p_local: int, cw: int) -> torch.Tensor:
        """o (frames * world * p_local, cw): whole frames of this rank's heads -> (frames * p_local, world * cw): this rank's rows of
        all heads."""
        w_ = self.world
        rows = frames * p_local
        send = self._slab("os", (w_ * rows, cw), o)                                                     # [dest s][frame][p]
        blocks = [((f * w_ + s) * p_local, (s * frames + f) * p_local, p_local) for s in range(w_) for f in range(frames)]
        self._row_perm(o, send, blocks, ("opack", frames, p_local, cw, str(o.device)), p_local)
        recv = self._a2a_equal(send, self._slab("or", (w_ * rows, cw), o), "from_heads")              # [src r][rows]
        out = torch.empty((rows, w_ * cw), dtype=o.dtype, device=o.device)
        self._col_blocks(recv, out, [(r * rows * cw, r * cw) for r in range(w_)], ("ounpack", rows, cw, str(o.device)), rows, cw)
        return out

    # -- spatial self-attention, gathered K / V ----------------------------------------------------------------------------------
    def gather_rows(self, x: torch.Tensor) -> torch.Tensor:
        """x (frames, p_local, C): this rank's pixel rows of every frame -> (frames, world * p_local, C), every rank's in order.  One
        all_gather_into_tensor ([rank][frame][p]) and one row-block copy kernel into frame order."""
        frames, p_local, c = x.shape
        ev = self._tick(x, "gather_rows")
        send = x.contiguous()
        if self.staged and x.is_cuda:
            host = torch.empty((self.world * frames * p_local, c), dtype=x.dtype)
            self.dist.all_gather_into_tensor(host, send.detach().cpu().view(-1, c), group=self.group)
            recv = host.to(x.device)
        else:
            recv = self._slab("gr", (self.world * frames * p_local, c), x)
            self.dist.all_gather_into_tensor(recv, send.view(-1, c), group=self.group)
        self._tock(ev)
        self.bytes_sent += send.numel() * send.element_size()
        out = torch.empty((frames * self.world * p_local, c), dtype=x.dtype, device=x.device)
        blocks = [((s * frames + f) * p_local, (f * self.world + s) * p_local, p_local) for s in range(self.world) for f in range(frames)]
        self._row_perm(recv, out, blocks, ("gunpack", frames, p_local, c, str(x.device)), p_local)
        return out.view(frames, self.world * p_local, c)


class FrameShard(_ShardComm):
    """One clip's T keyframes split contiguously over the ranks of a process group (BASELINE.json config 4).

    Every rank holds frames [t0, t1) of EVERY clip of the batch, i.e. a (B, t_local, H, W, C) slab of each
    frames-outermost activation.  Spatial work is frame-local.  Temporal work (Conv1d k3 over T, GroupNorm over
    C/32 x T, temporal attention — every pixel independent) depends on `mode`:
      * "a2a" (default): `to_pixels` transposes the slab with one all-to-all into (all T frames x this rank's 1/world of
        the pixels), the UNSHARDED temporal kernels run on it, `to_frames` transposes back; `gather_pixels` collects the
        final prediction.  Nothing else is exchanged.
      * "halo" (round 1): `halo` (one boundary frame to / from each neighbour, point-to-point) for the convolutions,
        `allreduce` (per-(clip, pixel, group) sum / sum of squares, fp32) for the normalisations, `gather_frames`
        (all-gather of the K/V rows) for the attention.
    `broadcast` serves the TVI2V anchor frame in both modes.  Collectives go through torch.distributed: backend "nccl"
    (= RCCL on ROCm) moves device tensors directly; with "gloo" (CPU tests, or several ranks sharing one GPU) tensors are
    staged through host memory.  `bytes_sent`, `n_collectives` and (with `timing = []`) device events around every
    exchange are what `bench.py --shard-frames` reports.
    """

    def __init__(self, t_glob: int, rank: Optional[int] = None, world: Optional[int] = None, group=None,
                 mode: str = "a2a", heavy_last: bool = False):
        if mode not in ("a2a", "halo"):
            raise ValueError(f"FrameShard mode {mode!r}: 'a2a' or 'halo' (the row decomposition is parallel.RowShard)")
        self.mode = mode
        self._init_comm(rank, world, group)
        if self.world > t_glob:
            raise ValueError(f"cannot shard {t_glob} keyframes over {self.world} ranks")
        self.t_glob = t_glob
        self.bounds = frame_shards(t_glob, self.world)
        if heavy_last:                         # the ranks with one extra keyframe are the LAST ones (see cfg_pair)
            sizes = [b - a for a, b in self.bounds][::-1]
            self.bounds, s0 = [], 0
            for n in sizes:
                self.bounds.append((s0, s0 + n))
                s0 += n
        self.t0, self.t1 = self.bounds[self.rank]
        self.t_local = self.t1 - self.t0
        self.t_max = max(b - a for a, b in self.bounds)
        self._plans = {}
        self._slabs = {}

    def _log_partition(self) -> int:
        return int(self.bounds[0][1] - self.bounds[0][0] != self.t_max)

    # -- layout transposition (mode "a2a") ----------------------------------------------------------
    def _plan(self, b: int, hw: int, device):
        """Index plans of the frame <-> pixel transposition of a (b, t_local, hw, C) slab.

        wire order of the frame-side buffer: [dest rank r][clip][local frame][pixel of r's block]
        wire order of the pixel-side buffer: [source rank s][clip][frame of s][own pixel]
        (for b == 1 the latter IS (T, own pixels): no reordering on that side)."""
        key = (b, hw, str(device))
        pl = self._plans.get(key)
        if pl is not None:
            return pl
        if hw < self.world:
            raise ValueError(f"cannot split {hw} pixels over {self.world} ranks")
        pix = frame_shards(hw, self.world)
        p0, p1 = pix[self.rank]
        hw_me = p1 - p0
        rows = torch.arange(b * self.t_local * hw, dtype=torch.int64).view(b, self.t_local, hw)
        pack = torch.cat([rows[:, :, lo:hi].reshape(-1) for lo, hi in pix])            # wire position -> frame-layout row
        unpack = torch.empty_like(pack)
        unpack[pack] = torch.arange(pack.numel(), dtype=torch.int64)                  # frame-layout row -> wire position
        frame_rows = [b * self.t_local * (hi - lo) for lo, hi in pix]
        pixel_rows = [b * (hi - lo) * hw_me for lo, hi in self.bounds]
        std = torch.arange(b * self.t_glob * hw_me, dtype=torch.int64).view(b, self.t_glob, hw_me)
        wire_of_std = inv = None
        if b > 1:
            wire_of_std = torch.cat([std[:, lo:hi].reshape(-1) for lo, hi in self.bounds])   # wire position -> (b,T,p) row
            inv = torch.empty_like(wire_of_std)
            inv[wire_of_std] = torch.arange(wire_of_std.numel(), dtype=torch.int64)
            wire_of_std, inv = wire_of_std.to(device), inv.to(device)
        pl = dict(hw_me=hw_me, pack=pack.to(device), unpack=unpack.to(device), frame_rows=frame_rows,
                  pixel_rows=pixel_rows, to_wire=wire_of_std, from_wire=inv)
        if torch.device(device).type == "cuda":
            # The same two permutations as ROW BLOCKS for ccedit_copy_row_blocks (one HIP kernel per pack / unpack instead of an
            # ATen index_select over every row + a separate add): the wire order only permutes whole runs of rows.
            #   frame side: block (peer r, clip, local frame) = the pixels of r's block, contiguous on both sides
            #   pixel side (b > 1): block (peer s, clip) = s's keyframes of that clip x my pixels, contiguous on both sides
            fb, wpos = [], 0
            for lo, hi in pix:
                for bi in range(b):
                    for tl in range(self.t_local):
                        fb.append(((bi * self.t_local + tl) * hw + lo, wpos, hi - lo))
                        wpos += hi - lo
            pl["frame_blocks"] = torch.tensor(fb, dtype=torch.int64, device=device)                       # (frame row, wire row, rows)
            pl["frame_blocks_inv"] = torch.tensor([(w_, f_, n_) for f_, w_, n_ in fb], dtype=torch.int64, device=device)
            pl["frame_max"] = max(n_ for _, _, n_ in fb)
            if b > 1:
                pb, wpos = [], 0
                for lo, hi in self.bounds:
                    for bi in range(b):
                        pb.append(((bi * self.t_glob + lo) * hw_me, wpos, (hi - lo) * hw_me))
                        wpos += (hi - lo) * hw_me
                pl["pixel_blocks"] = torch.tensor(pb, dtype=torch.int64, device=device)                   # (std row, wire row, rows)
                pl["pixel_blocks_inv"] = torch.tensor([(w_, s_, n_) for s_, w_, n_ in pb], dtype=torch.int64, device=device)
                pl["pixel_max"] = max(n_ for _, _, n_ in pb)
        self._plans[key] = pl
        return pl

    def _slab(self, tag: str, rows: int, like: torch.Tensor) -> torch.Tensor:
        """Wire-side scratch (pack output, unpack input): two alternating slabs per role instead of an allocation per exchange.
        Safe to recycle: a slab is next written by a kernel on the stream that already waited for the collective that read /
        wrote it (all_to_all_single returns with that stream dependency in place); results handed to the caller are never slabs."""
        key = (tag, rows, like.shape[1], like.dtype, str(like.device))
        ent = self._slabs.get(key)
        if ent is None:
            ent = [[torch.empty((rows, like.shape[1]), dtype=like.dtype, device=like.device) for _ in range(2)], 0]
            self._slabs[key] = ent
        ent[1] ^= 1
        return ent[0][ent[1]]

    def _all_to_all(self, send: torch.Tensor, out_rows, in_rows, out: Optional[torch.Tensor] = None) -> torch.Tensor:
        if out is None:
            out = torch.empty((sum(out_rows), send.shape[1]), dtype=send.dtype, device=send.device)
        ev = self._tick(send)
        if self.staged and send.is_cuda:
            h_in = send.detach().cpu()
            h_out = torch.empty(out.shape, dtype=out.dtype)
            self.dist.all_to_all_single(h_out, h_in, list(out_rows), list(in_rows), group=self.group)
            out.copy_(h_out)
        else:
            self.dist.all_to_all_single(out, send, list(out_rows), list(in_rows), group=self.group)
        self._tock(ev)
        own = in_rows[self.rank]
        self.bytes_sent += (send.shape[0] - own) * send.shape[1] * send.element_size()
        return out

    def hw_local(self, hw: int) -> int:
        lo, hi = frame_shards(hw, self.world)[self.rank]
        return hi - lo

    def to_pixels(self, x2d: torch.Tensor, b: int, hw: int) -> torch.Tensor:
        """(b * t_local * hw, C) rows of my keyframes -> (b * t_glob * hw_local, C): ALL keyframes of my pixel block."""
        pl = self._plan(b, hw, x2d.device)
        assert x2d.shape[0] == b * self.t_local * hw, (x2d.shape, b, self.t_local, hw)
        if not x2d.is_cuda:                    # CPU tensors (gloo tests of the collective pattern): ATen
            send = x2d.index_select(0, pl["pack"])
            y = self._all_to_all(send, pl["pixel_rows"], pl["frame_rows"])
            return y if b == 1 else y.index_select(0, pl["from_wire"])
        from . import ops
        send = ops.copy_row_blocks(x2d.contiguous(), self._slab("send_f", x2d.shape[0], x2d), pl["frame_blocks"], pl["frame_max"])
        n_out = sum(pl["pixel_rows"])
        if b == 1:                             # the wire order IS (T, own pixels): the received buffer is the result
            return self._all_to_all(send, pl["pixel_rows"], pl["frame_rows"])
        w = self._all_to_all(send, pl["pixel_rows"], pl["frame_rows"], out=self._slab("recv_p", n_out, x2d))
        return ops.copy_row_blocks(w, torch.empty_like(w), pl["pixel_blocks_inv"], pl["pixel_max"])

    def to_frames(self, y2d: torch.Tensor, b: int, hw: int, add: Optional[torch.Tensor] = None) -> torch.Tensor:
        """Inverse of to_pixels; `add` (frame layout) is summed into the result (the ResBlock skip)."""
        pl = self._plan(b, hw, y2d.device)
        assert y2d.shape[0] == b * self.t_glob * pl["hw_me"], (y2d.shape, b, self.t_glob, pl["hw_me"])
        if not y2d.is_cuda:
            send = y2d.contiguous() if b == 1 else y2d.index_select(0, pl["to_wire"])
            w = self._all_to_all(send, pl["frame_rows"], pl["pixel_rows"])
            x = w.index_select(0, pl["unpack"])
            if add is not None:
                x.add_(add)
            return x
        from . import ops
        y2d = y2d.contiguous()
        send = y2d if b == 1 else ops.copy_row_blocks(y2d, self._slab("send_p", y2d.shape[0], y2d), pl["pixel_blocks"], pl["pixel_max"])
        n_out = sum(pl["frame_rows"])
        w = self._all_to_all(send, pl["frame_rows"], pl["pixel_rows"], out=self._slab("recv_f", n_out, y2d))
        # unpack + the ResBlock skip in ONE pass: x = w[wire order] + add, fp32 add, one rounding (== x.add_(add))
        return ops.copy_row_blocks(w, torch.empty_like(w), pl["frame_blocks_inv"], pl["frame_max"],
                                   add=None if add is None else add.contiguous())

    # -- helpers --------------------------------------------------------------------------------
    def halo(self, first: torch.Tensor, last: torch.Tensor):
        """Send my first local frame to rank-1 and my last to rank+1; return (frame before my first, frame after my
        last) — None at the clip ends.  first/last: contiguous tensors of identical shape."""
        return self._neighbours(first, last, "halo")

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

    @staticmethod
    def cfg_pair(t_glob: int, groups=(None, None), mode: str = "a2a"):
        """Shards for the two CFG halves of a step: the second one puts its longer shards on the LAST ranks, so the
        2 x t_glob frame instances are spread as evenly as whole frames allow (17 over 8: 5,4,4,4,4,4,4,5).  `groups`:
        one process group per half — the halves run on two streams, and on ONE communicator the exchanges of half 1 would
        queue behind all of half 0's (correct, but serialised: the advertised overlap needs two).  When the second group is not
        given and torch.distributed is initialised with more than one rank, it is created here over the same ranks
        (`dist.new_group` is collective: every rank calls cfg_pair at the same point, as bench.py and the tests do)."""
        g0, g1 = groups
        if g1 is None:
            import torch.distributed as dist
            if dist.is_available() and dist.is_initialized() and dist.get_world_size(g0) > 1:
                ranks = list(range(dist.get_world_size())) if g0 is None else dist.get_process_group_ranks(g0)
                g1 = dist.new_group(ranks)
        return (FrameShard(t_glob, group=g0, mode=mode), FrameShard(t_glob, group=g1, mode=mode, heavy_last=True))

    def gather_pixels(self, y2d: torch.Tensor, b: int, hw: int) -> torch.Tensor:
        """(b * t_glob * hw_local, C) pixel-layout rows -> (b * t_glob * hw, C): every rank's pixel block, in order."""
        pix = frame_shards(hw, self.world)
        hw_me, hw_max = pix[self.rank][1] - pix[self.rank][0], max(hi - lo for lo, hi in pix)
        c = y2d.shape[1]
        yl = y2d.reshape(b * self.t_glob, hw_me, c)
        if hw_me < hw_max:
            yl = torch.cat([yl, yl.new_zeros((b * self.t_glob, hw_max - hw_me, c))], dim=1)
        ev = self._tick(y2d, "gather_pixels")
        send = self._out(yl)
        bufs = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(bufs, send, group=self.group)
        self._tock(ev)
        self.bytes_sent += send.numel() * send.element_size()
        full = torch.cat([bufs[r][:, : hi - lo] for r, (lo, hi) in enumerate(pix)], dim=1).to(y2d.device)
        return full.reshape(b * self.t_glob * hw, c)

    def gather_frames(self, x: torch.Tensor, b: int) -> torch.Tensor:
        """x: (b * t_local, ...) local frames of every clip -> (b * t_glob, ...) with all ranks' frames in order."""
        rest = x.shape[1:]
        xl = x.reshape(b, self.t_local, *rest)
        if self.t_local < self.t_max:          # equal-size all-gather: pad short shards
            pad = torch.zeros((b, self.t_max - self.t_local, *rest), dtype=x.dtype, device=x.device)
            xl = torch.cat([xl, pad], dim=1)
        ev = self._tick(x, "gather_frames")
        send = self._out(xl)
        bufs = [torch.empty_like(send) for _ in range(self.world)]
        self.dist.all_gather(bufs, send, group=self.group)
        self._tock(ev)
        self.bytes_sent += send.numel() * send.element_size()
        parts = [bufs[r][:, : (hi - lo)] for r, (lo, hi) in enumerate(self.bounds)]
        full = torch.cat(parts, dim=1).to(x.device)
        return full.reshape(b * self.t_glob, *rest)
