"""Stand-in for ``omni.isaac.gym.vec_env.VecEnvBase`` (third party, not in the reference tree):
exactly the surface the reference uses -- ``_task, _world, _render, _record, sim_frame_count,
num_envs, action_space, observation_space, set_task(...), world, render_enabled, close()``
(uses: envs/vec_env_rlgames.py:46-50,54-55,81-90; tasks/base/rl_task.py:210,215,291;
scripts/random_policy.py:104-125).  No Kit / SimulationApp is booted: the world is the
synthetic one of sim/synthetic.py.
"""
from __future__ import annotations

from typing import Optional

from ..sim.synthetic import SyntheticWorld


class VecEnvBase:
    def __init__(self, headless: bool = True, sim_device: int = 0, enable_livestream: bool = False,
                 enable_viewport: bool = False, launch_simulation_app: bool = False, experience: Optional[str] = None):
        self._render = (not headless) or enable_livestream or enable_viewport
        self._record = False
        self.sim_frame_count = 0
        self._world = None
        self._task = None
        self._sim_device = sim_device

    def set_task(self, task, backend="numpy", sim_params=None, init_sim=True, rendering_dt=1.0 / 60.0) -> None:
        self._world = SyntheticWorld()
        self._task = task
        self._num_envs = task.num_envs
        self.observation_space = task.observation_space
        self.action_space = task.action_space
        self.num_actions = task.num_actions
        self.num_observations = task.num_observations
        if init_sim:
            task.set_up_scene(None)
            self._world.reset()
            task.post_reset()

    @property
    def world(self):
        return self._world

    @property
    def render_enabled(self) -> bool:
        return self._render

    @property
    def num_envs(self) -> int:
        return self._num_envs

    @property
    def simulation_app(self):
        return None

    def render(self, mode="human") -> None:
        return None

    def close(self) -> None:
        return None

    def seed(self, seed=-1):
        return seed
