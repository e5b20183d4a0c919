"""CPU, world_size 2 over gloo: the N>1 path of the framework -- env sharding, per-rank seeds and the one
episode-statistics all-reduce -- matches a single-process computation over all envs."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from omniisaacgymenvs_amd.utils import distributed as od


def _free_port() -> int:
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _global_buffers(n):
    g = torch.Generator().manual_seed(3)
    return (torch.randn(n, generator=g), (torch.rand(n, generator=g) < 0.3).long(),
            torch.randint(0, 1000, (n,), generator=g), torch.randint(0, 50, (n,), generator=g).float())


def _worker(rank, world, port, n, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    rew, reset, prog, succ = _global_buffers(n)
    start, count = od.shard_envs(n, rank, world)
    sl = slice(start, start + count)
    stats = od.local_stats(rew[sl], reset[sl], prog[sl], succ[sl])
    out = od.all_reduce_stats(stats)
    q.put((rank, start, count, od.rank_seed(42, rank), out))
    dist.barrier()
    dist.destroy_process_group()


def test_sharding_covers_every_env_once():
    for total, world in ((1 << 20, 8), (1000, 3), (5, 8), (0, 2)):
        blocks = [od.shard_envs(total, r, world) for r in range(world)]
        assert sum(c for _, c in blocks) == total
        pos = 0
        for start, count in blocks:
            assert start == pos
            pos += count
    assert od.rank_seed(42, 3) == 45 and od.rank_seed(-1, 3) == -1


def test_two_rank_statistics_match_single_process():
    n, world = 1001, 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    rew, reset, prog, succ = _global_buffers(n)
    want = od.all_reduce_stats(od.local_stats(rew, reset, prog, succ))  # no process group here: plain local result
    assert sorted(r[3] for r in results) == [42, 43]
    assert sorted((r[1], r[2]) for r in results) == [(0, 501), (501, 500)]
    for _, _, _, _, got in results:
        for k, v in want.items():
            assert abs(got[k] - v) <= 1e-9 * max(1.0, abs(v)), (k, got[k], v)
