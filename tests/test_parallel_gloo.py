"""N>1 path on CPU: world_size-2 gloo processes exercise the rank plumbing bench.py uses for its multi-GPU
(replica) mode — clip assignment, barrier-bracketed timing with MAX over ranks — and the frame-shard plan."""
import os
import socket

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, n_clips, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccedit_amd.parallel import max_over_ranks, shard_clips
    mine = shard_clips(n_clips, rank, world)
    # every rank "processes" its clips; the whole-job time is the slowest rank's
    dist.barrier()
    local_seconds = 0.25 * len(mine) + 0.01 * rank
    job_seconds = max_over_ranks(local_seconds)
    dist.barrier()
    owned = torch.zeros(n_clips, dtype=torch.int64)
    owned[mine] = 1
    dist.all_reduce(owned)                                    # each clip owned exactly once across ranks
    q.put((rank, mine, job_seconds, owned.tolist()))
    dist.destroy_process_group()


@pytest.mark.timeout(120)
def test_replica_mode_two_ranks_gloo():
    world, n_clips = 2, 5
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=90) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert res[0][1] == [0, 2, 4] and res[1][1] == [1, 3]
    assert res[0][3] == [1] * n_clips
    # rank 0 has 3 clips (0.75 s), rank 1 has 2 (0.51 s): both must report the max
    assert res[0][2] == pytest.approx(0.75) and res[1][2] == pytest.approx(0.75)


def test_frame_shard_plan():
    from ccedit_amd.parallel import frame_shards, sharding_efficiency
    assert frame_shards(34, 8) == [(0, 5), (5, 10), (10, 14), (14, 18), (18, 22), (22, 26), (26, 30), (30, 34)]
    assert frame_shards(34, 2) == [(0, 17), (17, 34)]
    cover = [i for a, b in frame_shards(17, 4) for i in range(a, b)]
    assert cover == list(range(17))
    # SURVEY.md §8e ceilings: 34 instances over 8 ranks -> 0.85; 17 frames over 8 -> 0.71; over 2 -> 0.94
    assert sharding_efficiency(34, 8) == pytest.approx(0.85)
    assert sharding_efficiency(17, 8) == pytest.approx(17 / 24)
    assert sharding_efficiency(17, 2) == pytest.approx(17 / 18)
    from ccedit_amd.parallel import cfg_pair_efficiency
    assert cfg_pair_efficiency(17, 8) == pytest.approx(34 / 40)         # 5,4,4,4,4,4,4,5
    assert cfg_pair_efficiency(17, 4) == pytest.approx(34 / 36)
    assert cfg_pair_efficiency(17, 2) == pytest.approx(1.0)


def _shard_worker(rank, world, port, t_glob, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccedit_amd.parallel import FrameShard
    sh = FrameShard(t_glob)
    b, hw, c = 2, 5, 4
    # global tensor: value encodes (clip, frame); every rank holds its own keyframes of every clip
    full = (torch.arange(b)[:, None] * 100 + torch.arange(t_glob)[None, :]).float()[:, :, None, None].expand(b, t_glob, hw, c)
    local = full[:, sh.t0:sh.t1].contiguous()
    prev, nxt = sh.halo(local[:, 0].contiguous(), local[:, -1].contiguous())
    ok = True
    if sh.t0 > 0:
        ok &= torch.equal(prev, full[:, sh.t0 - 1])
    else:
        ok &= prev is None
    if sh.t1 < t_glob:
        ok &= torch.equal(nxt, full[:, sh.t1])
    else:
        ok &= nxt is None
    g = sh.gather_frames(local.reshape(b * sh.t_local, hw, c), b)
    ok &= torch.equal(g.reshape(b, t_glob, hw, c), full)
    st = local.sum(dim=1)                      # partial statistic over the local frames
    sh.allreduce(st)
    ok &= torch.allclose(st, full.sum(dim=1))
    # TVI2V: the centre keyframe's rows travel from the rank that holds it to every rank
    centre = t_glob // 2
    owner = sh.owner_of(centre)
    ok &= sh.bounds[owner][0] <= centre < sh.bounds[owner][1]
    anchor = local[:, centre - sh.t0].contiguous() if sh.rank == owner else torch.empty(b, hw, c)
    sh.broadcast(anchor, owner)
    ok &= torch.equal(anchor, full[:, centre])
    # mode "a2a": frame layout <-> pixel layout (all T frames of this rank's pixel block), both CFG-pair partitions
    for b2 in (1, 2):
        for shp in FrameShard.cfg_pair(t_glob):
            hw2 = 11                                                   # uneven pixel blocks too
            fullp = (torch.arange(b2)[:, None, None] * 10000 + torch.arange(t_glob)[None, :, None] * 100
                     + torch.arange(hw2)[None, None, :]).float()[..., None].expand(b2, t_glob, hw2, c).contiguous()
            loc = fullp[:, shp.t0:shp.t1].reshape(-1, c).contiguous()
            pix = shp.to_pixels(loc, b2, hw2)
            from ccedit_amd.parallel import frame_shards as fs
            p0, p1 = fs(hw2, world)[rank]
            ok &= torch.equal(pix.view(b2, t_glob, p1 - p0, c), fullp[:, :, p0:p1])
            ok &= shp.hw_local(hw2) == p1 - p0
            skip = torch.full_like(loc, 0.5)
            back = shp.to_frames(pix, b2, hw2, add=skip)
            ok &= torch.equal(back, loc + 0.5)
            allp = shp.gather_pixels(pix, b2, hw2)
            ok &= torch.equal(allp.view(b2, t_glob, hw2, c), fullp)
            ok &= shp.n_collectives == 3 and shp.bytes_sent > 0
    pair = FrameShard.cfg_pair(t_glob)
    load = torch.tensor([pair[0].t_local + pair[1].t_local])
    loads = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(loads, load)
    q.put((rank, bool(ok), sh.t0, sh.t1, [int(v) for v in loads]))
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world,t_glob", [(2, 17), (3, 5), (4, 17)])
def test_frame_shard_primitives_gloo(world, t_glob):
    """halo exchange / statistics all-reduce / K-V all-gather / centre-frame broadcast of the frame-sharded mode, and the
    all-to-all layout transposition of mode "a2a" with its mirrored CFG-pair partitions; uneven shards included
    (17 keyframes over 4 ranks: 5,4,4,4 and 4,4,4,5)."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_shard_worker, args=(r, world, port, t_glob, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(r[1] for r in res), res
    assert res[0][2] == 0 and res[-1][3] == t_glob and all(res[i][3] == res[i + 1][2] for i in range(world - 1))
    from ccedit_amd.parallel import cfg_pair_efficiency
    loads = res[0][4]
    assert sum(loads) == 2 * t_glob
    assert (sum(loads) / world) / max(loads) == pytest.approx(cfg_pair_efficiency(t_glob, world))
    if (world, t_glob) == (4, 17):
        assert loads == [9, 8, 8, 9]


def _subgroup_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccedit_amd.parallel import FrameShard
    members = [1, 2]
    g = dist.new_group(members)                       # every rank creates the group; only its members use it
    ok = True
    if rank in members:
        t_glob, hw, c = 4, 6, 3
        full = torch.arange(t_glob * hw * c, dtype=torch.float32).view(1, t_glob, hw, c)
        for mode in ("halo", "a2a"):
            sh = FrameShard(t_glob, group=g, mode=mode)
            ok &= sh.rank == members.index(rank) and sh.world == 2
            local = full[:, sh.t0:sh.t1].contiguous()
            prev, nxt = sh.halo(local[:, 0].contiguous(), local[:, -1].contiguous())      # peers are GLOBAL ranks 1 and 2
            ok &= (prev is None) if sh.rank == 0 else torch.equal(prev, full[:, sh.t0 - 1])
            ok &= (nxt is None) if sh.rank == 1 else torch.equal(nxt, full[:, sh.t1])
            pix = sh.to_pixels(local.reshape(-1, c), 1, hw)
            ok &= torch.equal(sh.to_frames(pix, 1, hw), local.reshape(-1, c))
            anchor = local[:, 0].contiguous() if sh.rank == 1 else torch.empty(1, hw, c)
            sh.broadcast(anchor, 1)                                                        # group rank 1 = global rank 2
            ok &= torch.equal(anchor, full[:, 2])
    q.put((rank, bool(ok)))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
def test_frame_shard_inside_a_subgroup_gloo():
    """A FrameShard built on a process sub-group addresses its point-to-point peers and broadcast sources by GLOBAL rank
    (torch.distributed's P2POp / broadcast convention): group {1, 2} of a 3-rank world."""
    world = 3
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_subgroup_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(ok for _, ok in res), res


# ------------------------------------------------------------------------------------------
# parallel.RowShard (BASELINE.json config 4, the balanced decomposition): every rank holds 1 / world of the ROWS of all frames
# ------------------------------------------------------------------------------------------
def _row_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from ccedit_amd.parallel import RowShard
    rs = RowShard()
    RowShard.issue_log = []
    n, h, w, c = 34, 8 * world, 6, 4                       # T = 17 keyframes x 2 clips
    g = torch.Generator().manual_seed(5)
    full = torch.randn(n, h, w, c, generator=g)
    r0, r1 = rs.rows(h)
    mine = full[:, r0:r1].contiguous()
    ok = True
    # halo rows: the extended slab is the zero-padded frame's rows [r0 - 1, r1 + 1)
    pad = torch.nn.functional.pad(full, (0, 0, 0, 0, 1, 1))
    ok &= torch.equal(rs.halo_rows(mine), pad[:, r0:r1 + 2])
    ok &= torch.equal(rs.halo_rows(mine, below=False), pad[:, r0:r1 + 1])
    # GroupNorm sums: local (sum, sum of squares) -> the frame's, divided by the number of ranks
    st = torch.stack([mine.double().sum(dim=(1, 2, 3)), (mine.double() ** 2).sum(dim=(1, 2, 3))], dim=1)
    want = torch.stack([full.double().sum(dim=(1, 2, 3)), (full.double() ** 2).sum(dim=(1, 2, 3))], dim=1) / world
    ok &= torch.allclose(rs.gn_stats(st), want, rtol=1e-12)
    # K / V rows: every rank ends with all pixel rows of every frame, in row order
    ok &= torch.equal(rs.gather_rows(mine.view(n, -1, c)), full.view(n, -1, c))
    # halo rows as two separate tensors (what the conv kernel reads in place): the rows above / below, None at the frame's ends
    up, down = rs.halo_exchange(mine)
    ok &= (up is None) == (rank == 0) and (down is None) == (rank == world - 1)
    ok &= up is None or torch.equal(up, full[:, r0 - 1])
    ok &= down is None or torch.equal(down, full[:, r1])
    up2, down2 = rs.halo_exchange(mine, below=False)          # a stride-2 convolution: only the row above travels
    ok &= down2 is None and (up2 is None or torch.equal(up2, full[:, r0 - 1]))
    # head-parallel attention exchange: q | k | v as column blocks of one projection -> whole frames of this rank's heads, and back
    heads, d = 8, 4
    C = heads * d
    cw = C // world
    qkv_full = torch.randn(n, h * w, 3 * C, generator=g)
    p_local = (r1 - r0) * w
    qkv = qkv_full[:, r0 * w:r1 * w].reshape(n * p_local, 3 * C).contiguous()
    parts = rs.to_heads([(qkv, 0), (qkv, C), (qkv, 2 * C)], n, p_local, cw)
    for j, got in enumerate(parts):
        want = qkv_full[:, :, j * C + rank * cw: j * C + (rank + 1) * cw].reshape(n * h * w, cw)
        ok &= torch.equal(got, want)
    o_full = torch.randn(n, h * w, C, generator=g)             # every rank computes its heads' columns of the whole frames
    mine_o = o_full[:, :, rank * cw:(rank + 1) * cw].reshape(n * h * w, cw).contiguous()
    back = rs.from_heads(mine_o, n, p_local, cw)
    ok &= torch.equal(back, o_full[:, r0 * w:r1 * w].reshape(n * p_local, C))
    # GroupNorm statistics riding on the halo exchange (round 6): raw boundary rows to the neighbours + every rank's partial sums to
    # everybody in ONE batch; the total is the sum of the partials in rank order — identical bits on every rank
    st_loc = torch.stack([mine.double().sum(dim=(1, 2, 3)), (mine.double() ** 2).sum(dim=(1, 2, 3))], dim=1)
    up3, down3, total = rs.halo_stats_exchange(mine, st_loc.clone())
    ok &= (up3 is None) == (rank == 0) and (down3 is None) == (rank == world - 1)
    ok &= up3 is None or torch.equal(up3, full[:, r0 - 1])
    ok &= down3 is None or torch.equal(down3, full[:, r1])
    parts = [torch.stack([full[:, a:b].double().sum(dim=(1, 2, 3)), (full[:, a:b].double() ** 2).sum(dim=(1, 2, 3))], dim=1)
             for a, b in ((r * h // world, (r + 1) * h // world) for r in range(world))]
    want_total = parts[0].clone()
    for p_ in parts[1:]:
        want_total += p_
    ok &= torch.equal(total, want_total)
    # all_agree: the ranks take the same branch (graph or eager after a capture attempt, network._forward_graphed) — true only when
    # every rank says so; not a data-path exchange (neither counted nor logged)
    nc = rs.n_collectives
    ok &= rs.all_agree(True) is True and rs.all_agree(rank != world - 1) is False and rs.all_agree(rank == 0) is (world == 1)
    ok &= rs.n_collectives == nc
    q.put((rank, bool(ok), rs.n_collectives, rs.bytes_sent, [k for _, k, _ in RowShard.issue_log]))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.timeout(180)
@pytest.mark.parametrize("world", [2, 4, 8])
def test_row_shard_primitives_gloo(world):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_row_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=150) for _ in range(world))
    for p in procs:
        p.join(30)
        assert p.exitcode == 0
    assert all(r[1] for r in res), [r[:2] for r in res]
    assert all(r[2] == 9 and r[4] == res[0][4] for r in res)              # same collectives, same order, on every rank
    assert res[0][4] == ["halo_rows", "halo_rows", "allreduce", "gather_rows", "halo_rows", "halo_rows", "to_heads", "from_heads", "halo_stats"]
    # an interior rank sends two boundary rows per halo exchange, the end ranks one
    assert res[0][3] < res[world // 2][3] or world == 2


def test_row_shard_is_balanced_where_frame_sharding_is_not():
    from ccedit_amd.parallel import cfg_pair_efficiency, row_sharding_efficiency, sharding_efficiency
    for world in (2, 4, 8):
        assert row_sharding_efficiency(64, world) == 1.0                     # 64 / 32 / 16 / 8 latent rows divide by 2, 4, 8
    assert sharding_efficiency(17, 8) < cfg_pair_efficiency(17, 8) < 0.9 < row_sharding_efficiency(64, 8)
    with pytest.raises(ValueError):
        row_sharding_efficiency(60, 8)
