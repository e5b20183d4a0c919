"""AllegroHandTask -- mirror of /root/reference/omniisaacgymenvs/tasks/allegro_hand.py:42-144.

16 DOFs, all actuated (views/allegro_hand_view.py:53), no fingertip bodies, no states.  The ``full`` (72) and
``full_no_vel`` (50) observation packers (:117-144) run inside ``allegro_post_kernel`` (csrc/hand.hip) together with
compute_hand_reward; resets and action scaling are the shared in-hand kernels instantiated for 16 DOFs.
"""
from __future__ import annotations

import torch

from ..sim.synthetic import SyntheticArticulationView
from .shared.in_hand_manipulation import InHandManipulationTask


def allegro_hand_stand_in_limits() -> torch.Tensor:
    """(16, 2) stand-in DOF limits (the real ones live in the AllegroHand USD asset)."""
    g = torch.Generator().manual_seed(16)
    lower = -(0.2 + 0.6 * torch.rand(16, generator=g))
    upper = 0.3 + 1.3 * torch.rand(16, generator=g)
    return torch.stack([lower, upper], dim=-1)


class SyntheticAllegroHandView(SyntheticArticulationView):
    """AllegroHandView surface used by the task: ``actuated_dof_indices`` (allegro_hand_view.py:47-53)."""

    def __init__(self, name: str, count: int, device) -> None:
        super().__init__(name=name, count=count, num_dof=16, device=device, dof_limits=allegro_hand_stand_in_limits())
        self._actuated_dof_indices = [i for i in range(self.num_dof)]

    @property
    def actuated_dof_indices(self):
        return self._actuated_dof_indices


class AllegroHandTask(InHandManipulationTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        InHandManipulationTask.__init__(self, name=name, env=env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self.object_type = self._task_cfg["env"]["objectType"]
        assert self.object_type in ["block"]
        self.obs_type = self._task_cfg["env"]["observationType"]
        if not (self.obs_type in ["full_no_vel", "full"]):
            raise Exception("Unknown type of observations!\nobservationType should be one of: [full_no_vel, full]")
        print("Obs type:", self.obs_type)
        self.num_obs_dict = {"full_no_vel": 50, "full": 72}
        self.object_scale = torch.tensor([1.0, 1.0, 1.0])
        self._num_observations = self.num_obs_dict[self.obs_type]
        self._num_actions = 16
        self._num_states = 0
        # what the shared C-ABI argument block also carries (unused by the Allegro kernels)
        self.asymmetric_obs = False
        self.num_fingertips = 0
        self.force_torque_obs_scale = 10.0
        InHandManipulationTask.update_config(self)

    def get_starting_positions(self):
        self.hand_start_translation = torch.tensor([0.0, 0.0, 0.5], device=self.device)
        self.hand_start_orientation = torch.tensor([0.257551, 0.283045, 0.683330, -0.621782], device=self.device)
        self.pose_dy, self.pose_dz = -0.2, 0.06

    def get_hand_view(self, scene=None):
        view = SyntheticAllegroHandView("allegro_hand_view", self._num_envs, self.device)
        view.positions[:] = self._env_pos + self.hand_start_translation
        return view
