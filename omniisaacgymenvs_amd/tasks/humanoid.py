"""HumanoidLocomotionTask -- mirror of /root/reference/omniisaacgymenvs/tasks/humanoid.py:44-141.

87 observations, 21 actions, the gear table of humanoid.py:91-116, motor_effort_ratio =
gears / max(gears) (:117-118), two foot force sensors, limit cost humanoid.py:135-141 (inside the
fused HIP kernel).
"""
from __future__ import annotations

import torch

from ..sim.synthetic import SyntheticArticulationView
from .base.rl_task import RLTask
from .shared.locomotion import LocomotionTask

HUMANOID_GEARS = [
    67.5, 67.5,              # lower_waist
    67.5, 67.5,              # right_upper_arm
    67.5, 67.5,              # left_upper_arm
    67.5,                    # pelvis
    45.0, 45.0,              # right / left lower_arm
    45.0, 135.0, 45.0,       # right_thigh x, y, z
    45.0, 135.0, 45.0,       # left_thigh x, y, z
    90.0, 90.0,              # knees
    22.5, 22.5, 22.5, 22.5,  # feet
]
HUMANOID_FORCE_LINKS = ["left_foot", "right_foot"]


def humanoid_stand_in_limits() -> torch.Tensor:
    """(21, 2) stand-in DOF limits (the real ones live in the Humanoid USD asset, not in the tree)."""
    g = torch.Generator().manual_seed(21)
    half = 0.6 + 1.2 * torch.rand(21, generator=g)
    mid = 0.5 * (torch.rand(21, generator=g) - 0.5)
    return torch.stack([mid - half, mid + half], dim=-1)


class HumanoidLocomotionTask(LocomotionTask):
    _limit_cost_kind = 1
    _num_sensors = 2

    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._num_observations = 87
        self._num_actions = 21
        self._humanoid_positions = torch.tensor([0, 0, 1.34])
        LocomotionTask.__init__(self, name=name, env=env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        LocomotionTask.update_config(self)

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._humanoids = SyntheticArticulationView(
            name="humanoid_view", count=self._num_envs, num_dof=21, device=self._device, num_bodies=2,
            dof_limits=humanoid_stand_in_limits(), body_names={n: i for i, n in enumerate(HUMANOID_FORCE_LINKS)},
        )
        self._humanoids.positions[:] = self._env_pos + self._humanoid_positions.to(self._device)
        if scene is not None:
            scene.add(self._humanoids)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def get_robot(self):
        return self._humanoids

    def post_reset(self):
        self.joint_gears = torch.tensor(HUMANOID_GEARS, device=self._device)
        self.max_motor_effort = torch.max(self.joint_gears)
        self.motor_effort_ratio = self.joint_gears / self.max_motor_effort
        dof_limits = self._humanoids.get_dof_limits()
        self.dof_limits_lower = dof_limits[0, :, 0].to(self._device)
        self.dof_limits_upper = dof_limits[0, :, 1].to(self._device)
        self._sensor_indices = torch.tensor(
            [self._humanoids._body_indices[j] for j in HUMANOID_FORCE_LINKS], device=self._device, dtype=torch.long
        )
        LocomotionTask.post_reset(self)

    def get_dof_at_limit_cost(self):
        """(N,) fp32: scaled excess over 0.98 of |scaled DOF position|, weighted by motor effort (humanoid.py:130-141)."""
        return self._limit_cost(torch.float32)
