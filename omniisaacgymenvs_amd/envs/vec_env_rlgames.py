"""VecEnvRLGames: the rl_games-facing wrapper, behaviour-compatible with
/root/reference/omniisaacgymenvs/envs/vec_env_rlgames.py:38-115.

``step(actions)``: action DR -> clamp -> ``task.pre_physics_step`` -> ``control_frequency_inv``
world steps -> ``task.post_physics_step`` -> observation DR -> states -> clamp / device hop ->
``({"obs", "states"}, rew, resets, extras)``.  ``reset()`` flags every env and steps once with
zero actions.  The recording branch (:55-70) needs a renderer and is reduced to its counters.
"""
from __future__ import annotations

from datetime import datetime

import torch

from .vec_env_base import VecEnvBase


class VecEnvRLGames(VecEnvBase):
    def _process_data(self):
        t = self._task
        self._obs = torch.clamp(self._obs, -t.clip_obs, t.clip_obs).to(t.rl_device)
        self._rew = self._rew.to(t.rl_device)
        self._states = torch.clamp(self._states, -t.clip_obs, t.clip_obs).to(t.rl_device)
        self._resets = self._resets.to(t.rl_device)
        self._extras = self._extras

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

        if t.randomize_actions:
            actions = t._dr_randomizer.apply_actions_randomization(actions=actions, reset_buf=t.reset_buf)

        actions = torch.clamp(actions, -t.clip_actions, t.clip_actions).to(t.device)

        t.pre_physics_step(actions)
        self._advance_world(to_render)
        self._obs, self._rew, self._resets, self._extras = t.post_physics_step()

        if t.randomize_observations:
            self._obs = t._dr_randomizer.apply_observations_randomization(
                observations=self._obs.to(device=t.rl_device), reset_buf=t.reset_buf
            )

        self._states = t.get_states()
        self._process_data()

        obs_dict = {"obs": self._obs, "states": self._states}
        return obs_dict, self._rew, self._resets, self._extras

    def reset(self, seed=None, options=None):
        """Flag all envs, then one zero-action step to recompute observations and states."""
        now = datetime.now().strftime("%Y-%m-%d %H:%M:%S")
        print(f"[{now}] Running RL reset")
        self._task.reset()
        actions = torch.zeros((self.num_envs, self._task.num_actions), device=self._task.rl_device)
        obs_dict, _, _, _ = self.step(actions)
        return obs_dict
