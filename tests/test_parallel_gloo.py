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
