"""ctypes front end of csrc/dr_noise.hip: the per-step observation / action noise of
/root/reference/omniisaacgymenvs/utils/domain_randomization/randomize.py:283-410.

Two flows, same kernel:
  * ``sync_free=True``  one launch, samples from in-kernel Philox keyed by (seed, step, element, stream);
  * ``sync_free=False`` the reference's flow: reset ids and randomize ids are compacted on the device, their
    counts are read back (the reference's ``len()`` syncs too), the samples are drawn with the very torch calls of
    :346-361 / :381-400 in the same order, and the kernel scatters them by compaction rank.
No torch/CPU fallback: the buffers must live on a ROCm device.
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Any, Dict, Optional

import torch

from ... import _lib

_OPS = {"additive": _lib.NOISE_ADDITIVE, "scaling": _lib.NOISE_SCALING}
_DISTS = {"gaussian": _lib.NOISE_GAUSSIAN, "normal": _lib.NOISE_GAUSSIAN, "uniform": _lib.NOISE_UNIFORM,
          "loguniform": _lib.NOISE_LOGUNIFORM, "log_uniform": _lib.NOISE_LOGUNIFORM}


def _term(params: Optional[Dict[str, Any]], key: str) -> _lib.NoiseTerm:
    t = _lib.NoiseTerm()
    if params is None or key not in params:
        return t
    p = params[key]
    if p["operation"] not in _OPS:
        raise _lib.OigeError(f"The specified {p['operation']} operation type is not supported.")
    if p["distribution"] not in _DISTS:
        raise _lib.OigeError(f"The specified {p['distribution']} distribution is not supported.")
    t.enabled = 1
    t.operation = _OPS[p["operation"]]
    t.distribution = _DISTS[p["distribution"]]
    t.frequency_interval = int(p.get("frequency_interval", 0))
    a, b = p["distribution_parameters"][0], p["distribution_parameters"][1]
    if t.distribution == _lib.NOISE_GAUSSIAN:
        t.p0, t.p1 = a, b
    elif t.distribution == _lib.NOISE_UNIFORM:
        t.p0, t.p1 = a, b - a
    else:
        t.p0, t.p1 = math.log(a), math.log(b) - math.log(a)
    return t


def _draw(p: Dict[str, Any], rows: int, width: int, device) -> torch.Tensor:
    """The reference's own sampling calls (:346-361, :381-400)."""
    dist, dp = p["distribution"], p["distribution_parameters"]
    if dist in ("gaussian", "normal"):
        return torch.normal(mean=dp[0], std=dp[1], size=(rows, width), device=device)
    if dist == "uniform":
        return (dp[1] - dp[0]) * torch.rand((rows, width), device=device) + dp[0]
    return torch.exp((math.log(dp[1]) - math.log(dp[0])) * torch.rand((rows, width), device=device) + math.log(dp[0]))


class NoiseState:
    """Device scratch of one noised buffer (observations or actions)."""

    def __init__(self, n: int, device) -> None:
        self.reset_workspace = _lib.new_workspace(n, device)
        self.interval_workspace = _lib.new_workspace(n, device)


def build_args(buffer: torch.Tensor, reset_buf: Optional[torch.Tensor], counter: torch.Tensor, correlated: torch.Tensor,
               params: Dict[str, Any], seed: int, stream_id: int, step: int, clamped_out: Optional[torch.Tensor] = None,
               clip: float = math.inf, step_ptr: Optional[int] = None) -> _lib.NoiseArgs:
    """The argument block of the in-kernel-Philox flow; also what the in-hand kernels take to apply the noise themselves
    (``oige_hand_post_physics_dr`` / ``oige_hand_pre_physics_dr``: they hold the reset flags in registers, ``reset_buf`` is
    not read and may be None)."""
    n, w = buffer.shape
    rq = _lib.require_device_tensor
    rq(buffer, torch.float32, (n, w), "noise buffer")
    rq(counter, torch.int32, (n,), "counter buffer")
    rq(correlated, torch.float32, (n, w), "correlated noise buffer")
    a = _lib.NoiseArgs()
    a.buffer, a.counter, a.correlated_noise = buffer.data_ptr(), counter.data_ptr(), correlated.data_ptr()
    if reset_buf is not None:
        rq(reset_buf, torch.int64, (n,), "reset_buf")
        a.reset_buf = reset_buf.data_ptr()
    if clamped_out is not None:
        rq(clamped_out, torch.float32, (n, w), "clamped_out")
        a.clamped_out = clamped_out.data_ptr()
        a.clip = float(min(clip, 3.0e38))
    a.seed, a.step, a.stream_id, a.width = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step), int(stream_id), int(w)
    a.step_ptr = step_ptr  # graph capture: the Philox step counter lives on the device
    a.on_reset, a.on_interval = _term(params, "on_reset"), _term(params, "on_interval")
    return a


def apply(buffer: torch.Tensor, reset_buf: torch.Tensor, counter: torch.Tensor, correlated: torch.Tensor,
          params: Dict[str, Any], state: NoiseState, seed: int, stream_id: int, step: int, sync_free: bool = True,
          clamped_out: Optional[torch.Tensor] = None, clip: float = math.inf, step_ptr: Optional[int] = None) -> torch.Tensor:
    n, w = buffer.shape
    rq = _lib.require_device_tensor
    rq(buffer, torch.float32, (n, w), "noise buffer")
    if reset_buf.dtype != torch.int64:  # AnymalTerrain's bool reset_buf (anymal_terrain.py:465)
        reset_buf = reset_buf.to(torch.int64)
    rq(reset_buf, torch.int64, (n,), "reset_buf")
    rq(counter, torch.int32, (n,), "counter buffer")
    rq(correlated, torch.float32, (n, w), "correlated noise buffer")
    a = _lib.NoiseArgs()
    a.buffer, a.reset_buf, a.counter = buffer.data_ptr(), reset_buf.data_ptr(), counter.data_ptr()
    a.correlated_noise = correlated.data_ptr()
    if clamped_out is not None:
        rq(clamped_out, torch.float32, (n, w), "clamped_out")
        a.clamped_out = clamped_out.data_ptr()
        a.clip = float(min(clip, 3.0e38))
    a.seed, a.step, a.stream_id, a.width = int(seed) & 0xFFFFFFFFFFFFFFFF, int(step), int(stream_id), int(w)
    a.step_ptr = step_ptr  # graph capture: the Philox step counter lives on the device
    a.on_reset, a.on_interval = _term(params, "on_reset"), _term(params, "on_interval")
    lib, stream = _lib.load(), _lib.current_stream(buffer.device)
    keep = []
    if not sync_free:
        a.reset_workspace, a.interval_workspace = state.reset_workspace.data_ptr(), state.interval_workspace.data_ptr()
        _lib.check(lib.oige_dr_noise_count(C.byref(a), n, stream), "oige_dr_noise_count")
        k, m = int(state.reset_workspace[-1].item()), int(state.interval_workspace[-1].item())
        # same draws, same order as the reference: correlated first (only when there are reset ids), then uncorrelated
        rs = _draw(params["on_reset"], k, w, buffer.device).contiguous() if ("on_reset" in params and k > 0) else None
        it = _draw(params["on_interval"], m, w, buffer.device).contiguous() if "on_interval" in params else None
        one = torch.zeros(4, device=buffer.device)  # non-NULL placeholder keeps the kernel on the pre-drawn path
        a.reset_samples = (rs if rs is not None else one).data_ptr()
        a.interval_samples = (it if it is not None and m > 0 else one).data_ptr()
        keep += [rs, it, one]
    _lib.check(lib.oige_dr_noise_apply(C.byref(a), n, stream), "oige_dr_noise_apply")
    state._keep = keep
    return buffer
