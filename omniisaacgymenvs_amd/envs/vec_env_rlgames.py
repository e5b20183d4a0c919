"""VecEnvRLGames: the rl_games-facing wrapper, behaviour-compatible with
/root/reference/omniisaacgymenvs/envs/vec_env_rlgames.py:38-115.

``step(actions)``: action DR -> clamp -> ``task.pre_physics_step`` -> ``control_frequency_inv``
world steps -> ``task.post_physics_step`` -> observation DR -> states -> clamp / device hop ->
``({"obs", "states"}, rew, resets, extras)``.  ``reset()`` flags every env and steps once with
zero actions.  The recording branch (:55-70) needs a renderer and is reduced to its counters.

``fused_io`` (default True) removes the wrapper's extra full passes over actions / obs / states without
changing a value the policy sees:
  * ``torch.clamp(actions)`` (:77) is folded into the task's pre_physics kernel where it supports that;
  * with DR, the noise kernel writes the clamped copy in the same launch (:94-97 + :40 in one pass); and where the
    task supports it (``fused_dr``, the in-hand family) there is no noise launch at all: the pre_physics kernel noises
    the actions as it reads them, the post_physics kernel noises and clamps the observations as it writes them
    (same Philox samples, bit-identical values);
  * ``torch.clamp(x, -inf, inf)`` (clip key absent: Ant, Humanoid, AnymalTerrain's states) is skipped: the task's
    buffer itself is returned instead of a fresh copy of it.  A caller that keeps ``obs`` across ``step`` calls
    without copying it (rl_games copies it into its experience buffer first) should set ``fused_io = False``,
    which restores the reference's allocate-per-step behaviour.
"""
from __future__ import annotations

import gc
import math
from datetime import datetime

import torch

from ..tasks.base.rl_task import quiesce_gc
from .vec_env_base import VecEnvBase


class VecEnvRLGames(VecEnvBase):
    fused_io = True
    fused_dr = True  # with fused_io: tasks that can (the in-hand family) apply the DR noise inside their own kernels
    _obs_is_clamped = False

    def _clamp_obs(self, x):
        t = self._task
        if self.fused_io and math.isinf(t.clip_obs):
            return x  # clamp(x, -inf, inf) == x: no pass, no copy
        return torch.clamp(x, -t.clip_obs, t.clip_obs)

    def _process_data(self):
        t = self._task
        if not self._obs_is_clamped:
            self._obs = self._clamp_obs(self._obs)
        self._obs = self._obs.to(t.rl_device)
        self._rew = self._rew.to(t.rl_device)
        self._states = self._clamp_obs(self._states).to(t.rl_device)
        self._resets = self._resets.to(t.rl_device)
        self._extras = self._extras

    def _persistent(self, name, like):
        buf = getattr(self, name, None)
        if buf is None or buf.shape != like.shape or buf.device != like.device:
            buf = torch.empty_like(like)
            setattr(self, name, buf)
        return buf

    def set_task(self, task, backend="numpy", sim_params=None, init_sim=True, rendering_dt=1.0 / 60.0) -> None:
        super().set_task(task, backend, sim_params, init_sim, rendering_dt)
        self.num_states = self._task.num_states
        self.state_space = self._task.state_space

    def _advance_world(self, to_render: bool) -> None:
        t = self._task
        substeps = t.control_frequency_inv
        render_last = (self.sim_frame_count + substeps) % t.rendering_interval == 0
        for k in range(substeps):
            last = k == substeps - 1
            self._world.step(render=(to_render if (render_last and last) else False))
            self.sim_frame_count += 1

    def step(self, actions):
        t = self._task
        to_render = self._render
        if self._record:
            if not hasattr(self, "step_count"):
                self.step_count = 0
            self.step_count += 1

        fused = self.fused_io and actions.device == torch.device(t.device) and actions.dtype == torch.float32
        finite_act_clip = not math.isinf(t.clip_actions)
        t.clamp_actions_in_kernel = False
        in_kernel_dr = (fused and self.fused_dr and getattr(t, "supports_fused_dr", False) and t._dr_randomizer.sync_free
                        and actions.is_contiguous() and self._world.is_playing())
        action_noise = None
        if t.randomize_actions and in_kernel_dr and t.supports_action_clamp:
            action_noise = t._dr_randomizer.fused_actions_args(actions)
            t.clamp_actions_in_kernel = finite_act_clip
        elif t.randomize_actions and fused and finite_act_clip and actions.is_contiguous():
            clamped = self._persistent("_act_clamped", actions)  # noise + clamp in one launch
            t._dr_randomizer.apply_actions_randomization(actions=actions, reset_buf=t.reset_buf, clamped_out=clamped,
                                                         clip=t.clip_actions)
            actions = clamped
        else:
            if t.randomize_actions:
                actions = t._dr_randomizer.apply_actions_randomization(actions=actions, reset_buf=t.reset_buf)
            if fused and t.supports_action_clamp:
                t.clamp_actions_in_kernel = finite_act_clip  # the pre_physics kernel clamps (or there is nothing to clamp)
                actions = actions.to(t.device)
            else:
                actions = torch.clamp(actions, -t.clip_actions, t.clip_actions).to(t.device)

        if action_noise is not None:
            t.pre_physics_step(actions, action_noise=action_noise)
        else:
            t.pre_physics_step(actions)
        t.clamp_actions_in_kernel = False
        self._advance_world(to_render)

        self._obs_is_clamped = False
        if t.randomize_observations and in_kernel_dr and not math.isinf(t.clip_obs) and t.rl_device == t.device:
            clamped = self._persistent("_obs_clamped", t.obs_buf)
            noise = t._dr_randomizer.fused_observations_args(t.obs_buf, clamped_out=clamped, clip=t.clip_obs)
            self._obs, self._rew, self._resets, self._extras = t.post_physics_step(obs_noise=noise)
            self._obs, self._obs_is_clamped = clamped, True
            self._states = t.get_states()
            self._process_data()
            return {"obs": self._obs, "states": self._states}, self._rew, self._resets, self._extras
        self._obs, self._rew, self._resets, self._extras = t.post_physics_step()

        if t.randomize_observations:
            obs = self._obs.to(device=t.rl_device)
            if self.fused_io and not math.isinf(t.clip_obs) and obs.is_contiguous():
                clamped = self._persistent("_obs_clamped", obs)  # noise + clamp in one launch
                t._dr_randomizer.apply_observations_randomization(observations=obs, reset_buf=t.reset_buf,
                                                                  clamped_out=clamped, clip=t.clip_obs)
                self._obs, self._obs_is_clamped = clamped, True
            else:
                self._obs = t._dr_randomizer.apply_observations_randomization(observations=obs, reset_buf=t.reset_buf)

        self._states = t.get_states()
        self._process_data()

        obs_dict = {"obs": self._obs, "states": self._states}
        return obs_dict, self._rew, self._resets, self._extras

    # ------------------------------------------------------------------ HIP-graph step (launch-bound sizes)
    def capture_step(self, actions: torch.Tensor) -> None:
        """Capture the whole ``step(actions)`` -- action noise / clamp, ``pre_physics_step``, the (delegated) world step,
        ``post_physics_step``, observation noise / clamp -- into one HIP graph.  ``actions`` is the static input buffer (refill
        it in place before each ``replay_step()``).  Needs the fused wrapper flow (``fused_io``), a sync-free capturable task
        (``RLTask.capture_step``) and the sync-free randomizer.  The Philox step counters of the task and of the two noise
        streams are read from device scalars the graph advances (``step_ptr``), so replayed steps draw what eager steps would.
        One eager ``step`` runs first as warm-up; the second one is captured, not executed."""
        t = self._task
        if t.graph_capturable is not True:
            raise NotImplementedError(f"{type(t).__name__}: {t.graph_capturable}")
        if not (self.fused_io and getattr(t, "sync_free", False) and getattr(t, "fused", False)):
            raise RuntimeError("capture_step needs env.fused_io and the task's sync-free fused path")
        dr = getattr(t, "_dr_randomizer", None)
        if dr is not None and (t.randomize_actions or t.randomize_observations) and not dr.sync_free:
            raise RuntimeError("capture_step needs the sync-free randomizer (in-kernel Philox samples)")
        dev = t.obs_buf.device
        # the graph of an earlier capture of this wrapper is released NOW, before the window opens (its destructor calls
        # hipGraphExecDestroy, which HIP rejects while a capture is open)
        self._graph = self._graph_out = None
        t._step_dev = torch.full((1,), getattr(t, "_step_count", 0), dtype=torch.int64, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream(dev))
        gc_on = None
        try:
            t._capturing = True
            if dr is not None:
                dr.capturing = True
            gc_on = quiesce_gc()
            with torch.cuda.stream(side):
                t._step_dev += 1
                self.step(actions)  # warm-up
            torch.cuda.current_stream(dev).wait_stream(side)
            host = (getattr(t, "_step_count", 0), getattr(dr, "_observations_step", 0), getattr(dr, "_actions_step", 0))
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph, stream=side):
                t._step_dev += 1
                out = self.step(actions)
            t._step_count = host[0]  # the captured step has not run yet
            if dr is not None:
                if hasattr(dr, "_observations_step"):
                    dr._observations_step = host[1]
                if hasattr(dr, "_actions_step"):
                    dr._actions_step = host[2]
        finally:
            t._capturing = False
            if dr is not None:
                dr.capturing = False
            if gc_on:
                gc.enable()
        self._graph, self._graph_out, self._graph_actions = graph, out, actions

    def replay_step(self):
        """One captured ``step``: returns the same ``(obs_dict, rew, resets, extras)`` objects, refreshed in place."""
        t = self._task
        self._graph.replay()
        t._step_count = getattr(t, "_step_count", 0) + 1
        t._counts_key = None
        dr = getattr(t, "_dr_randomizer", None)
        if dr is not None:
            if hasattr(dr, "_observations_step") and t.randomize_observations:
                dr._observations_step += 1
            if hasattr(dr, "_actions_step") and t.randomize_actions:
                dr._actions_step += 1
        return self._graph_out

    def reset(self, seed=None, options=None):
        """Flag all envs, then one zero-action step to recompute observations and states."""
        now = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        print(f"[{now}] Running RL reset")
        self._task.reset()
        actions = torch.zeros((self.num_envs, self._task.num_actions), device=self._task.rl_device)
        obs_dict, _, _, _ = self.step(actions)
        return obs_dict
