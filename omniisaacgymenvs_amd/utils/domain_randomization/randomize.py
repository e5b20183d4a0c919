"""Observation / action noise of the reference's domain randomisation
(/root/reference/omniisaacgymenvs/utils/domain_randomization/randomize.py:215-410), the part of
``Randomizer`` that is task math.  Physics-parameter randomisation through
``omni.replicator.isaac`` (:151-213, :412-771) mutates PhysX state and is out of scope.

The arithmetic runs in liboige_hip.so (csrc/dr_noise.hip) -- see ``apply_*_randomization``.
"""
from __future__ import annotations

from typing import Any, Dict, Optional

import numpy as np
import torch

_NEEDED_RESET = ("operation", "distribution", "distribution_parameters")
_NEEDED_INTERVAL = ("frequency_interval", "operation", "distribution", "distribution_parameters")


class Randomizer:
    def __init__(self, main_config: Dict[str, Any], task_config: Dict[str, Any]):
        self._cfg = task_config
        self._config = main_config
        self.randomize = False
        dr_config = self._cfg.get("domain_randomization", None)
        self.distributions: Dict[str, Any] = dict()
        self.active_domain_randomizations: Dict[Any, Any] = dict()
        self._observations_dr_params: Optional[Dict[str, Any]] = None
        self._actions_dr_params: Optional[Dict[str, Any]] = None
        if dr_config is not None:
            randomize = dr_config.get("randomize", False)
            randomization_params = dr_config.get("randomization_params", None)
            if randomize and randomization_params is not None:
                self.randomize = True
                self.min_frequency = dr_config.get("min_frequency", 1)

    # ------------------------------------------------------------------ setup (:151-281)
    def set_up_domain_randomization(self, task) -> None:
        if not self.randomize:
            dr_config = self._cfg.get("domain_randomization", None)
            if dr_config is None:
                raise ValueError("No domain randomization parameters are specified in the task yaml config file")
            print("Domain randomization will not be applied.")
            return
        params = self._cfg["domain_randomization"]["randomization_params"]
        for opt in params.keys():
            if opt == "observations":
                self._set_up_observations_randomization(task)
            elif opt == "actions":
                self._set_up_actions_randomization(task)
            # "simulation" / "rigid_prim_views" / "articulation_views": PhysX-side, delegated

    def _validate(self, params: Optional[Dict[str, Any]], what: str) -> Dict[str, Any]:
        if params is None:
            raise ValueError(f"{what.capitalize()} randomization parameters are not provided.")
        if "on_reset" in params and not set(_NEEDED_RESET).issubset(params["on_reset"].keys()):
            raise ValueError(
                f"Please ensure the following {what} on_reset randomization parameters are provided: "
                "operation, distribution, distribution_parameters."
            )
        if "on_interval" in params and not set(_NEEDED_INTERVAL).issubset(params["on_interval"].keys()):
            raise ValueError(
                f"Please ensure the following {what} on_interval randomization parameters are provided: "
                "frequency_interval, operation, distribution, distribution_parameters."
            )
        for when in ("on_reset", "on_interval"):
            if when in params:
                self.active_domain_randomizations[(what, when)] = np.array(params[when]["distribution_parameters"])
        return params

    def _set_up_observations_randomization(self, task) -> None:
        task.randomize_observations = True
        self._observations_dr_params = self._validate(
            self._cfg["domain_randomization"]["randomization_params"]["observations"], "observations"
        )
        n, dev = self._cfg["env"]["numEnvs"], self._config["rl_device"]
        self._observations_counter_buffer = torch.zeros((n), dtype=torch.int, device=dev)
        self._observations_correlated_noise = torch.zeros((n, task.num_observations), device=dev)
        self._observations_step = 0

    def _set_up_actions_randomization(self, task) -> None:
        task.randomize_actions = True
        self._actions_dr_params = self._validate(
            self._cfg["domain_randomization"]["randomization_params"]["actions"], "actions"
        )
        n, dev = self._cfg["env"]["numEnvs"], self._config["rl_device"]
        self._actions_counter_buffer = torch.zeros((n), dtype=torch.int, device=dev)
        self._actions_correlated_noise = torch.zeros((n, task.num_actions), device=dev)
        self._actions_step = 0

    # ------------------------------------------------------------------ per-step noise (:283-410)
    # ``sync_free`` (default): samples from in-kernel Philox, no host sync.  ``sync_free = False``: the reference's
    # flow with torch's RNG (ids compacted on the device, counts read back, torch.normal / torch.rand draws).
    sync_free = True
    capturing = False   # set by VecEnvRLGames.capture_step: step counters are read from (and advanced on) the device

    def _advance(self, what: str, device) -> Optional[int]:
        """Advance the host step counter of ``what`` ("observations" / "actions"); while a step is being captured (or once one
        has been) a device twin is advanced on the stream too.  Returns the twin's address while capturing, else None."""
        name, dev_name = f"_{what}_step", f"_{what}_step_dev"
        setattr(self, name, getattr(self, name) + 1)
        twin = getattr(self, dev_name, None)
        if self.capturing and twin is None:
            twin = torch.full((1,), getattr(self, name) - 1, dtype=torch.int64, device=device)
            setattr(self, dev_name, twin)
        if twin is not None:
            twin += 1
        return twin.data_ptr() if (self.capturing and twin is not None) else None

    def apply_observations_randomization(self, observations, reset_buf, clamped_out=None, clip=float("inf")):
        from . import noise_hip

        if not hasattr(self, "_observations_noise_state"):
            self._observations_noise_state = noise_hip.NoiseState(observations.shape[0], observations.device)
        sp = self._advance("observations", observations.device)
        return noise_hip.apply(
            observations, reset_buf, self._observations_counter_buffer, self._observations_correlated_noise,
            self._observations_dr_params, self._observations_noise_state, seed=int(self._config.get("seed", 0)),
            stream_id=0, step=self._observations_step, sync_free=self.sync_free, clamped_out=clamped_out, clip=clip, step_ptr=sp,
        )

    # The in-hand kernels apply the noise in their own epilogue / prologue: these hand them the
    # argument block, advancing the same step counter the separate launch would, so both flows draw the same samples.
    def fused_observations_args(self, observations, clamped_out=None, clip=float("inf")):
        from . import noise_hip

        sp = self._advance("observations", observations.device)
        return noise_hip.build_args(
            observations, None, self._observations_counter_buffer, self._observations_correlated_noise,
            self._observations_dr_params, seed=int(self._config.get("seed", 0)), stream_id=0, step=self._observations_step,
            clamped_out=clamped_out, clip=clip, step_ptr=sp)

    def fused_actions_args(self, actions):
        from . import noise_hip

        sp = self._advance("actions", actions.device)
        return noise_hip.build_args(
            actions, None, self._actions_counter_buffer, self._actions_correlated_noise, self._actions_dr_params,
            seed=int(self._config.get("seed", 0)), stream_id=1, step=self._actions_step, step_ptr=sp)

    def apply_actions_randomization(self, actions, reset_buf, clamped_out=None, clip=float("inf")):
        from . import noise_hip

        if not hasattr(self, "_actions_noise_state"):
            self._actions_noise_state = noise_hip.NoiseState(actions.shape[0], actions.device)
        sp = self._advance("actions", actions.device)
        return noise_hip.apply(
            actions, reset_buf, self._actions_counter_buffer, self._actions_correlated_noise,
            self._actions_dr_params, self._actions_noise_state, seed=int(self._config.get("seed", 0)),
            stream_id=1, step=self._actions_step, sync_free=self.sync_free, clamped_out=clamped_out, clip=clip, step_ptr=sp,
        )
