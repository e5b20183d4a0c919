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
    """The path's one collective, where a trainer has it: inside the step loop, off the critical path, consumed late.

    The reference's ranks never exchange env data (scripts/rlgames_train.py:114-120,171); what the trainer combines is
    the episode statistics RLGPUAlgoObserver collects every step (utils/rlgames/rlgames_utils.py:52-56) and reduces once
    per epoch (:72-96; an epoch is ``horizon_length`` steps: 16 for Ant / ShadowHand, 32 Humanoid, 48 AnymalTerrain,
    cfg/train/*PPO.yaml).  Per call of ``step``:

      1. the collective issued by an EARLIER call is waited for (RCCL: the current stream waits for the communicator's
         stream, the host does not block; at least a whole step of kernels has run since) and its result becomes
         ``latest`` -- whole-job running sums, one collective behind;
      2. this rank's step is added to its device-side statistics slots: by the task's own post kernel when the task
         keeps them (``attach(task)``: LocomotionTask -- no launch, no traffic), else by one launch of
         ``oige_episode_stats_accumulate`` over the buffers the post kernel has just written;
      3. on every ``every``-th call the slots are folded (``oige_episode_stats_fold``) and a snapshot goes out as ONE
         40-byte ``all_reduce(SUM, async_op=True)``; RCCL runs it on its own stream next to the following steps' kernels.

    ``every=1`` is a collective per step.  Measured on MI355X (one-rank RCCL group, Ant, 1 M envs, 127 us steps): +35 us
    per step -- the cross-stream hand-over (an event record and a wait on each side) idles the GPU for ~25 us per
    collective, the 40 bytes themselves are nothing -- so the default cadence of ``bench.py`` is the trainer's own, one per
    epoch.  CPU tensors (the gloo tests and ``bench.py --task _dry``) are summed with torch; device tensors always go
    through the HIP library.
    """

    def __init__(self, device, every: int = 1, force: bool = False, consume: str = "wait"):
        self.device = torch.device(device)
        self.every = max(1, int(every))
        # "wait": the next call waits for the collective (RCCL: a stream-side wait; deterministic: ``latest`` is exactly one
        # collective behind).  "poll": the next calls only ASK whether it has finished (Work.is_completed(), an event query)
        # and the compute stream never waits for the communicator's stream -- on ROCm each cross-stream wait idles the GPU
        # for tens of microseconds (measured: 128 us per collective with "wait", Ant, 1 M envs; profiles/r03/collective_cost.txt)
        if consume not in ("wait", "poll"):
            raise ValueError(f"consume must be 'wait' or 'poll', not {consume!r}")
        self.consume = consume

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
        self._env_steps = 0
        self.collectives = 0
        self._slots = None      # device: (num_slots, 4) f64 running sums
        self._scratch = None
        self._fused = False     # the task's post kernel keeps the slots
        self._io = None
        self._io_key = None

    # -- the task keeps the slots itself
    def attach(self, task) -> bool:
        """Let ``task``'s post kernel keep this rank's statistics (tasks with ``enable_episode_stats``); False: not offered."""
        hook = getattr(task, "enable_episode_stats", None)
        if hook is None or self.device.type != "cuda":
            return False
        self._slots = hook()
        self._fused = True
        return True

    def warm(self, rounds: int = 2) -> None:
        """``rounds`` whole collectives through the very path ``step`` takes (fold, async all-reduce, wait), regardless of the
        cadence: the first ones load code objects, allocate the fold scratch and set the communicator's channels up (0.9 ms
        together on MI355X) -- a warm-up shorter than ``every`` steps would leave all of that to the first collective of the
        measured region."""
        if not self.active:
            return
        for _ in range(rounds):
            self._issue()
            self._consume()
        if self.device.type == "cuda":
            torch.cuda.current_stream(self.device).synchronize()

    def reset(self) -> None:
        """Forget everything counted so far (the collective in flight is drained first)."""
        self._consume()
        self.acc.zero_()
        if self._slots is not None:
            self._slots.zero_()
        self._calls = self._env_steps = self.collectives = 0
        self._latest = None

    # -- this rank's step -> slots (device) / acc (host tensors)
    def _accumulate(self, rew_buf, reset_buf, progress_buf, successes) -> None:
        self._env_steps += int(rew_buf.shape[0])
        if self.device.type != "cuda":  # host tensors: gloo tests, dry runs
            s = local_stats(rew_buf, reset_buf, progress_buf, successes)
            self.acc[:4] += s[:4]
            self.acc[4] = float(self._env_steps)
            return
        if self._fused:
            return  # the post kernel has already added this step to its tiles' slots
        from .. import _lib

        n = int(rew_buf.shape[0])
        key = (id(rew_buf), id(reset_buf), id(progress_buf), id(successes))
        if key != self._io_key:  # the argument block is rebuilt only when a buffer was rebound (the tensors are held below)
            lib = _lib.load()
            _lib.require_device_tensor(rew_buf, torch.float32, (n,), "rew_buf")
            _lib.require_device_tensor(progress_buf, torch.int64, (n,), "progress_buf")
            if reset_buf.dtype not in (torch.int64, torch.bool, torch.uint8) or tuple(reset_buf.shape) != (n,) or not reset_buf.is_cuda:
                raise _lib.OigeError(f"reset_buf: expected a device (N,) int64 / bool tensor, got {reset_buf.dtype} {tuple(reset_buf.shape)} on {reset_buf.device}")
            if successes is not None:
                _lib.require_device_tensor(successes, torch.float32, (n,), "successes")
            blocks = int(lib.oige_episode_stats_blocks(n))
            if self._slots is None or self._slots.shape[0] != blocks:
                self._slots = torch.zeros((blocks, 4), dtype=torch.float64, device=self.device)
            io = _lib.EpisodeStatsIo()
            io.rew_buf, io.reset_buf, io.progress_buf = _lib.ptr(rew_buf), _lib.ptr(reset_buf), _lib.ptr(progress_buf)
            io.successes, io.slots = _lib.ptr(successes), _lib.ptr(self._slots)
            io.reset_elem_bytes = reset_buf.element_size()
            self._io, self._io_key, self._n = io, key, n
            self._held = (rew_buf, reset_buf, progress_buf, successes)
            self._fn = lib.oige_episode_stats_accumulate
        import ctypes

        _lib.check(self._fn(ctypes.byref(self._io), self._n, _lib.current_stream(self.device)), "oige_episode_stats_accumulate")

    def _fold(self, out=None) -> None:
        """slots -> ``out`` (default: acc).  Device tensors only; host tensors keep acc current themselves."""
        if self.device.type != "cuda" or self._slots is None:
            if out is not None:
                out.copy_(self.acc)
            return
        from .. import _lib

        lib, k = _lib.load(), int(self._slots.shape[0])
        need = int(lib.oige_episode_stats_fold_scratch(k))
        if need and (self._scratch is None or self._scratch.numel() < need):
            self._scratch = torch.zeros(need, dtype=torch.float64, device=self.device)
        _lib.check(lib.oige_episode_stats_fold(_lib.ptr(self._slots), k, _lib.ptr(self.acc if out is None else out), float(self._env_steps),
                                               _lib.ptr(self._scratch) if need else None, _lib.current_stream(self.device)),
                   "oige_episode_stats_fold")

    def _consume(self, block: bool = True) -> None:
        if self._work is not None:
            if not self._work.is_completed():
                if not block:
                    return
                self._work.wait()  # RCCL: the current stream waits for the communicator's; gloo: the host does
            self._latest = self._send[self._slot ^ 1]
            self._work = None

    def step(self, rew_buf, reset_buf, progress_buf, successes=None) -> None:
        self._consume(block=self.consume == "wait")
        self._accumulate(rew_buf, reset_buf, progress_buf, successes)
        self._calls += 1
        if not self.active or self._calls % self.every:
            return
        self._issue()

    def _issue(self) -> None:
        self._consume()  # at most one collective in flight: the send buffer about to be reused is then free for certain
        out = self._send[self._slot]
        if out.device != self.acc.device:  # gloo rehearsal of a GPU run: through the host (blocking; not the product path)
            self._fold()
            out.copy_(self.acc)
        else:
            self._fold(out)  # the slots are summed straight into the buffer that goes out
        self._work = dist.all_reduce(out, op=dist.ReduceOp.SUM, async_op=True)
        self._slot ^= 1
        self.collectives += 1

    @property
    def latest(self):
        """Whole-job running sums (STAT_KEYS order) as of the last CONSUMED collective: one collective behind ``step``.
        None before the first one has been consumed.  A device tensor under RCCL: reading it on the host synchronises."""
        return self._latest

    def finish(self) -> Dict[str, float]:
        """Drain: wait for the collective in flight (or, without a process group, take this rank's sums) -> whole-job means."""
        self._consume()
        if self.active and self._latest is not None and self._calls % self.every == 0:
            s = self._latest.tolist()
        elif self.active:  # the last steps were not covered by a collective: one blocking all-reduce of the final sums
            self._fold()
            t = self.acc.to(self._on).clone()
            dist.all_reduce(t, op=dist.ReduceOp.SUM)
            s = t.tolist()
        else:
            self._fold()
            s = self.acc.tolist()
        n, done = max(s[4], 1.0), max(s[1], 1.0)
        return {"mean_reward": s[0] / n, "done_fraction": s[1] / n, "mean_episode_length": s[2] / done,
                "mean_successes": s[3] / n, "num_envs": s[4] / max(self._calls, 1), "num_done": s[1], "env_steps": s[4],
                "steps": float(self._calls)}
