"""Multi-GPU plumbing of the path: one process per GPU, envs sharded, no data-path collective.

The reference runs one process per GPU, each with its own envs and ``seed + rank``
(/root/reference/omniisaacgymenvs/scripts/rlgames_train.py:114-120,171); the envs never exchange anything.
What is reduced across ranks is the episode-statistics vector the trainer logs (RLGPUAlgoObserver,
utils/rlgames/rlgames_utils.py:72-96): a handful of sums, all-reduced as ONE small tensor over RCCL
(``backend="nccl"`` on ROCm) or gloo (CPU tests).
"""
from __future__ import annotations

from typing import Dict, Tuple

import torch
import torch.distributed as dist

STAT_KEYS = ("reward_sum", "num_done", "episode_length_sum", "success_sum", "num_envs")


def shard_envs(total_envs: int, rank: int, world_size: int) -> Tuple[int, int]:
    """Contiguous block [start, start + count) of rank ``rank``; the first ``total % world`` ranks
    take one extra env.  Env ids inside a rank are local, as in the reference."""
    if not 0 <= rank < world_size:
        raise ValueError(f"rank {rank} outside world of {world_size}")
    base, extra = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < extra else 0)
    start = rank * base + min(rank, extra)
    return start, count


def rank_seed(seed: int, rank: int) -> int:
    """rlgames_train.py:171: ``cfg.seed + global_rank`` (``-1`` stays 'random')."""
    return seed if seed == -1 else seed + rank


def local_stats(rew_buf: torch.Tensor, reset_buf: torch.Tensor, progress_buf: torch.Tensor,
                successes: torch.Tensor = None) -> torch.Tensor:
    """This rank's statistics vector (float64, on the buffers' device), STAT_KEYS order."""
    done = reset_buf != 0
    out = torch.zeros(len(STAT_KEYS), dtype=torch.float64, device=rew_buf.device)
    out[0] = rew_buf.double().sum()
    out[1] = done.sum()
    out[2] = (progress_buf.double() * done).sum()
    if successes is not None:
        out[3] = successes.double().sum()
    out[4] = rew_buf.shape[0]
    return out


def all_reduce_stats(stats: torch.Tensor) -> Dict[str, float]:
    """Sum over ranks (one 40-byte collective; no-op without an initialised process group) -> whole-job means."""
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if stats.is_cuda and dist.get_backend() != "nccl":
            stats = stats.cpu()  # gloo rehearsal of a GPU run: reduce on the host
        dist.all_reduce(stats, op=dist.ReduceOp.SUM)
    s = stats.tolist()
    n, done = max(s[4], 1.0), max(s[1], 1.0)
    return {"mean_reward": s[0] / n, "done_fraction": s[1] / n, "mean_episode_length": s[2] / done,
            "mean_successes": s[3] / n, "num_envs": s[4], "num_done": s[1]}


class StatsReducer:
    """The path's one collective, where the trainer has it: EVERY step, off the critical path, consumed one step late.

    The reference's ranks never exchange env data (scripts/rlgames_train.py:114-120,171); what the trainer combines is
    the episode statistics RLGPUAlgoObserver logs (utils/rlgames/rlgames_utils.py:72-96).  Per call of ``step``:

      1. the collective issued by the PREVIOUS call is waited for (RCCL: the current stream waits for the communicator's
         stream, the host does not block; a whole step of kernels has run since, so the stream does not stall either) and
         its result becomes ``latest`` -- whole-job running sums as of one step ago;
      2. this rank's step is folded into ``acc`` (five running f64 sums, STAT_KEYS order) on the device by
         ``oige_episode_stats_accumulate`` from the buffers the post kernel has just written;
      3. a snapshot of ``acc`` goes out as ONE 40-byte ``all_reduce(SUM, async_op=True)``; RCCL runs it on its own stream
         next to the following step's kernels.

    ``every`` > 1 issues the collective on every k-th call only (the fold still runs each step).  CPU tensors (the gloo
    tests and ``bench.py --task _dry``) are folded with torch; device tensors always go through the HIP library.
    """

    def __init__(self, device, every: int = 1, force: bool = False):
        self.device = torch.device(device)
        self.every = max(1, int(every))
        # force: run the collective on a one-rank group too (rehearsal of the RCCL path on a single GPU)
        self.active = dist.is_available() and dist.is_initialized() and (dist.get_world_size() > 1 or force)
        self.backend = dist.get_backend() if self.active else "none"
        # where the collective's buffers live: on the device for RCCL, on the host for gloo
        self._on = self.device if (self.backend == "nccl" or self.device.type == "cpu") else torch.device("cpu")
        self.acc = torch.zeros(len(STAT_KEYS), dtype=torch.float64, device=self.device)
        self._send = [torch.zeros(len(STAT_KEYS), dtype=torch.float64, device=self._on) for _ in range(2)]
        self._slot = 0
        self._work = None
        self._latest = None
        self._calls = 0
        self.collectives = 0
        self._partials = None
        self._io = None
        self._io_key = None

    # -- this rank's step -> acc
    def _fold(self, rew_buf, reset_buf, progress_buf, successes) -> None:
        if self.device.type != "cuda":  # host tensors: gloo tests, dry runs
            self.acc += local_stats(rew_buf, reset_buf, progress_buf, successes)
            return
        from .. import _lib

        n = int(rew_buf.shape[0])
        key = (rew_buf.data_ptr(), reset_buf.data_ptr(), progress_buf.data_ptr(), None if successes is None else successes.data_ptr(), n)
        if key != self._io_key:  # the argument block is rebuilt only when a buffer was rebound
            lib = _lib.load()
            _lib.require_device_tensor(rew_buf, torch.float32, (n,), "rew_buf")
            _lib.require_device_tensor(progress_buf, torch.int64, (n,), "progress_buf")
            if reset_buf.dtype not in (torch.int64, torch.bool, torch.uint8) or tuple(reset_buf.shape) != (n,):
                raise _lib.OigeError(f"reset_buf: expected (N,) int64 / bool, got {reset_buf.dtype} {tuple(reset_buf.shape)}")
            if successes is not None:
                _lib.require_device_tensor(successes, torch.float32, (n,), "successes")
            if self._partials is None or self._partials.numel() < 4 * int(lib.oige_episode_stats_blocks(n)):
                self._partials = torch.zeros(4 * int(lib.oige_episode_stats_blocks(n)), dtype=torch.float64, device=self.device)
            io = _lib.EpisodeStatsIo()
            io.rew_buf, io.reset_buf, io.progress_buf = _lib.ptr(rew_buf), _lib.ptr(reset_buf), _lib.ptr(progress_buf)
            io.successes = _lib.ptr(successes)
            io.partials, io.acc = _lib.ptr(self._partials), _lib.ptr(self.acc)
            io.reset_elem_bytes = reset_buf.element_size()
            self._io, self._io_key, self._n = io, key, n
            self._fn = lib.oige_episode_stats_accumulate
        import ctypes

        _lib.check(self._fn(ctypes.byref(self._io), self._n, _lib.current_stream(self.device)), "oige_episode_stats_accumulate")

    def _consume(self) -> None:
        if self._work is not None:
            self._work.wait()
            self._latest = self._send[self._slot ^ 1]
            self._work = None

    def step(self, rew_buf, reset_buf, progress_buf, successes=None) -> None:
        self._consume()
        self._fold(rew_buf, reset_buf, progress_buf, successes)
        self._calls += 1
        if not self.active or self._calls % self.every:
            return
        out = self._send[self._slot]
        if out.device != self.acc.device:  # gloo rehearsal of a GPU run: through the host (blocking; not the product path)
            out.copy_(self.acc)
        else:
            out.copy_(self.acc, non_blocking=True)
        self._work = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)
        self._slot ^= 1
        self.collectives += 1

    @property
    def latest(self):
        """Whole-job running sums (STAT_KEYS order) as of the last CONSUMED collective: one step behind ``step``.
        None before the second call.  A device tensor under RCCL: reading it on the host synchronises."""
        return self._latest

    def finish(self) -> Dict[str, float]:
        """Drain: wait for the collective in flight (or, without a process group, take this rank's sums) -> whole-job means."""
        self._consume()
        if self.active and self._latest is not None and self._calls % self.every == 0:
            s = self._latest.tolist()
        elif self.active:  # the last steps were not covered by a collective: one blocking all-reduce of the final sums
            t = self.acc.to(self._on).clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            s = t.tolist()
        else:
            s = self.acc.tolist()
        n, done = max(s[4], 1.0), max(s[1], 1.0)
        return {"mean_reward": s[0] / n, "done_fraction": s[1] / n, "mean_episode_length": s[2] / done,
                "mean_successes": s[3] / n, "num_envs": s[4] / max(self._calls, 1), "num_done": s[1], "env_steps": s[4],
                "steps": float(self._calls)}
