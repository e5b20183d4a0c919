"""ORACLE (test infrastructure only) -- observation / action noise of the reference's domain randomisation on CPU torch.

Restates /root/reference/omniisaacgymenvs/utils/domain_randomization/randomize.py:
  apply_observations_randomization :283-312 (apply_actions_randomization :314-342 is the same code on the
  actions buffers), _apply_uncorrelated_noise :344-371, _apply_correlated_noise :373-410.
The sampling is a callback so a test can feed either torch's RNG (exact replay of the reference's draws) or
the Philox samples the sync-free HIP path generates (``philox_samples`` below restates csrc/dr_noise.hip's
generator: Box-Muller on Philox4x32-10 uniforms).

Pinning: tests/test_oracle_pinned.py runs the reference's four methods from their source on a stub ``self``
(pure torch -> pinned bit-for-bit, container only).
"""
from __future__ import annotations

from typing import Callable, Dict, Optional

import numpy as np
import torch

from . import philox

Tensor = torch.Tensor


def sample(p: Dict, rows: int, width: int) -> Tensor:
    """:346-361 / :381-400 with torch's global RNG."""
    dist, dp = p["distribution"], p["distribution_parameters"]
    if dist in ("gaussian", "normal"):
        return torch.normal(mean=dp[0], std=dp[1], size=(rows, width))
    if dist == "uniform":
        return (dp[1] - dp[0]) * torch.rand((rows, width)) + dp[0]
    if dist in ("loguniform", "log_uniform"):
        return torch.exp((np.log(dp[1]) - np.log(dp[0])) * torch.rand((rows, width)) + np.log(dp[0]))
    raise ValueError(dist)


def apply_randomization(buffer: Tensor, reset_buf: Tensor, counter: Tensor, correlated: Tensor, params: Dict,
                        draw: Optional[Callable[[str, Tensor], Tensor]] = None) -> Tensor:
    """In place on ``buffer``, ``counter`` (int32) and ``correlated``.  ``draw(kind, ids) -> (len(ids), W)`` with kind
    "on_reset" / "on_interval"; default: torch's RNG as the reference."""
    w = buffer.shape[1]
    if draw is None:
        draw = lambda kind, ids: sample(params[kind], len(ids), w)
    env_ids = reset_buf.nonzero(as_tuple=False).squeeze(-1)
    counter[env_ids] = 0
    counter += 1
    if "on_reset" in params:
        if len(env_ids) > 0:
            correlated[env_ids] = draw("on_reset", env_ids)
        op = params["on_reset"]["operation"]
        if op == "additive":
            buffer += correlated
        elif op == "scaling":
            buffer *= correlated
    if "on_interval" in params:
        randomize_ids = (counter >= params["on_interval"]["frequency_interval"]).nonzero(as_tuple=False).squeeze(-1)
        counter[randomize_ids] = 0
        noise = draw("on_interval", randomize_ids)
        op = params["on_interval"]["operation"]
        if op == "additive":
            buffer[randomize_ids] += noise
        elif op == "scaling":
            buffer[randomize_ids] *= noise
    return buffer


def philox_samples(p: Dict, n: int, width: int, seed: int, step: int, stream_id: int, kind: str, row0: int = 0) -> Tensor:
    """(N,W) samples the sync-free kernel would use for EVERY element: element i takes word i % 4 of the block
    with counter (i // 4, (4 + 2*stream_id + kind) << 16, step) -- csrc/dr_noise.hip noise_samples().
    ``row0``: the n rows are rows row0 .. row0 + n of a larger buffer (``row0 * width`` must be a multiple of 4)."""
    total = n * width
    nblk = (total + 3) // 4
    assert (row0 * width) % 4 == 0
    gf = np.arange(nblk, dtype=np.uint64) + np.uint64(row0 * width // 4)
    tag = 4 + 2 * stream_id + (1 if kind == "on_interval" else 0)
    w = philox.philox4x32_10(gf & philox.MASK, (gf >> np.uint64(32)) | np.uint64(tag << 16), np.uint64(step & 0xFFFFFFFF),
                             np.uint64(step >> 32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([philox.u01(x) for x in w], axis=-1)  # (nblk, 4) fp32
    dist, dp = p["distribution"], p["distribution_parameters"]
    f32 = np.float32
    if dist in ("gaussian", "normal"):
        out = np.empty_like(u)
        for h in range(2):
            r = np.sqrt(f32(-2.0) * np.log(f32(1.0) - u[:, 2 * h]))
            th = f32(6.28318530717958647692) * u[:, 2 * h + 1]
            out[:, 2 * h] = (r * np.cos(th)) * f32(dp[1]) + f32(dp[0])
            out[:, 2 * h + 1] = (r * np.sin(th)) * f32(dp[1]) + f32(dp[0])
    elif dist == "uniform":
        out = f32(dp[1] - dp[0]) * u + f32(dp[0])
    else:
        out = np.exp(f32(np.log(dp[1]) - np.log(dp[0])) * u + f32(np.log(dp[0])))
    return torch.from_numpy(out.astype(np.float32).reshape(-1)[:total].reshape(n, width).copy())
