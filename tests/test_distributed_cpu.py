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


def _step_buffers(step, n):
    """The (global) buffers of one step: seeded per step, so every rank and the single-process reference see the same."""
    g = torch.Generator().manual_seed(100 + step)
    return (torch.randn(n, generator=g), (torch.rand(n, generator=g) < 0.2).long(),
            torch.randint(0, 1000, (n,), generator=g), torch.randint(0, 50, (n,), generator=g).float())


def _reducer_worker(rank, world, port, n, steps, every, q):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    start, count = od.shard_envs(n, rank, world)
    sl = slice(start, start + count)
    red = od.StatsReducer("cpu", every=every)
    seen = []
    for s in range(steps):
        rew, reset, prog, succ = _step_buffers(s, n)
        red.step(rew[sl], reset[sl], prog[sl], succ[sl])
        seen.append(None if red.latest is None else red.latest.clone())
    final = red.finish()
    q.put((rank, [None if t is None else t.tolist() for t in seen], final, red.collectives))
    dist.barrier()
    dist.destroy_process_group()


def _run_reducer(every, steps=7, n=1001, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_reducer_worker, args=(r, world, port, n, steps, every, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    # single process: cumulative sums after each step
    cum, acc = [], torch.zeros(len(od.STAT_KEYS), dtype=torch.float64)
    for s in range(steps):
        acc = acc + od.local_stats(*_step_buffers(s, n))
        cum.append(acc.clone())
    return results, cum


def test_per_step_reducer_is_one_step_late_and_equals_a_single_process_run():
    """the collective runs every step, asynchronously, and what a step sees is the whole-job sums as of the
    PREVIOUS step -- equal to a single-process run over all envs."""
    steps = 7
    results, cum = _run_reducer(every=1, steps=steps)
    for rank, seen, final, collectives in results:
        assert collectives == steps
        assert seen[0] is None  # nothing to consume on the first call
        for s in range(1, steps):
            want = cum[s - 1]  # issued by call s-1 (sums through step s-1), consumed by call s
            got = torch.tensor(seen[s], dtype=torch.float64)
            assert torch.allclose(got, want, rtol=1e-12, atol=1e-9), (rank, s, got, want)
        t = cum[-1].tolist()
        assert abs(final["mean_reward"] - t[0] / t[4]) <= 1e-12 * max(1.0, abs(t[0] / t[4]))
        assert final["num_done"] == t[1] and final["env_steps"] == t[4] and final["num_envs"] == 1001 and final["steps"] == steps


def test_reducer_every_k_steps_drains_the_tail():
    steps, every = 7, 3
    results, cum = _run_reducer(every=every, steps=steps)
    for rank, seen, final, collectives in results:
        assert collectives == steps // every
        # calls 3 and 6 (1-based) issue; the result of call 3 (sums through step index 2) is visible from call 4 on
        assert seen[2] is None and torch.allclose(torch.tensor(seen[3], dtype=torch.float64), cum[2])
        assert torch.allclose(torch.tensor(seen[6], dtype=torch.float64), cum[5])
        t = cum[-1].tolist()  # finish() covers the steps behind the last collective with one blocking all-reduce
        assert final["num_done"] == t[1] and final["env_steps"] == t[4]


def test_reducer_without_a_process_group_is_local():
    red = od.StatsReducer("cpu")
    assert not red.active
    rew, reset, prog, succ = _step_buffers(0, 50)
    red.step(rew, reset, prog, succ)
    red.step(rew, reset, prog, succ)
    out = red.finish()
    assert red.latest is None and red.collectives == 0
    assert out["env_steps"] == 100 and out["num_envs"] == 50 and out["num_done"] == 2 * int(reset.sum())
