"""AntLocomotionTask -- mirror of /root/reference/omniisaacgymenvs/tasks/ant.py:44-109.

60 observations, 8 actions, gears 15, four foot force sensors, limit cost = count of scaled dof
positions > 0.99 (no abs, ant.py:107-109; computed inside the fused HIP kernel).
The USD asset / ArticulationView are replaced by the synthetic view of sim/synthetic.py.
"""
from __future__ import annotations

import torch

from ..sim.synthetic import SyntheticArticulationView
from .base.rl_task import RLTask
from .shared.locomotion import LocomotionTask

# Stand-ins for the DOF limits stored in the Ant USD asset (not in the reference tree):
# +-0.7 rad hips, [0.52, 1.75] / [-1.75, -0.52] ankles.
ANT_DOF_LIMITS = torch.tensor(
    [[-0.7, 0.7], [0.52, 1.75], [-0.7, 0.7], [-1.75, -0.52], [-0.7, 0.7], [-1.75, -0.52], [-0.7, 0.7], [0.52, 1.75]]
)
ANT_FORCE_LINKS = ["front_left_foot", "front_right_foot", "left_back_foot", "right_back_foot"]


class AntLocomotionTask(LocomotionTask):
    _limit_cost_kind = 0
    _num_sensors = 4

    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        LocomotionTask.__init__(self, name=name, env=env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._num_observations = 60
        self._num_actions = 8
        self._ant_positions = torch.tensor([0, 0, 0.5])
        LocomotionTask.update_config(self)

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._ants = SyntheticArticulationView(
            name="ant_view", count=self._num_envs, num_dof=8, device=self._device, num_bodies=4,
            dof_limits=ANT_DOF_LIMITS, body_names={n: i for i, n in enumerate(ANT_FORCE_LINKS)},
        )
        self._ants.positions[:] = self._env_pos + self._ant_positions.to(self._device)
        if scene is not None:
            scene.add(self._ants)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def get_robot(self):
        return self._ants

    def post_reset(self):
        self.joint_gears = torch.tensor([15, 15, 15, 15, 15, 15, 15, 15], dtype=torch.float32, device=self._device)
        dof_limits = self._ants.get_dof_limits()
        self.dof_limits_lower = dof_limits[0, :, 0].to(self._device)
        self.dof_limits_upper = dof_limits[0, :, 1].to(self._device)
        self.motor_effort_ratio = torch.ones_like(self.joint_gears, device=self._device)
        self._sensor_indices = torch.tensor(
            [self._ants._body_indices[j] for j in ANT_FORCE_LINKS], device=self._device, dtype=torch.long
        )
        LocomotionTask.post_reset(self)

    def get_dof_at_limit_cost(self):
        """(N,) int64: per env, how many scaled DOF positions exceed 0.99 -- no abs (ant.py:102-109)."""
        return self._limit_cost(torch.int64)
