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
