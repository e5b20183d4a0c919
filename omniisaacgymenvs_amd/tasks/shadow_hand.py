"""ShadowHandTask -- mirror of /root/reference/omniisaacgymenvs/tasks/shadow_hand.py:42-306.

Observation packers (openai 42 / full 157 obs, asymmetric full_state states 187) run inside the
fused HIP kernel (csrc/hand.hip).  ``full_no_vel`` and the non-asymmetric ``full_state`` obs are
accepted by the constructor as in the reference, and fail at the first step as they do there
(shape error at shadow_hand.py:190 / undefined dof_force_tensor at :269).
"""
from __future__ import annotations

import torch

from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .shared.in_hand_manipulation import InHandManipulationTask

# the four coupled distal joints (FFJ0, MFJ0, RFJ0, LFJ0) are not actuated; the real DOF numbering comes
# from the USD joint order (views/shadow_hand_view.py:59-84); stand-in: DOFs 5, 9, 13, 18
SHADOW_HAND_ACTUATED = [i for i in range(24) if i not in (5, 9, 13, 18)]
FINGERTIPS = ["robot0_ffdistal", "robot0_mfdistal", "robot0_rfdistal", "robot0_lfdistal", "robot0_thdistal"]


def shadow_hand_stand_in_limits() -> torch.Tensor:
    """(24, 2) stand-in DOF limits (the real ones live in the ShadowHand USD asset)."""
    g = torch.Generator().manual_seed(24)
    lower = -(0.2 + 0.6 * torch.rand(24, generator=g))
    upper = 0.3 + 1.3 * torch.rand(24, generator=g)
    return torch.stack([lower, upper], dim=-1)


class SyntheticShadowHandView(SyntheticArticulationView):
    """ShadowHandView surface used by the task: ``_fingers``, ``actuated_dof_indices``, ``_sensor_indices``."""

    def __init__(self, name: str, count: int, device) -> None:
        super().__init__(name=name, count=count, num_dof=24, device=device, num_bodies=5,
                         dof_limits=shadow_hand_stand_in_limits(), body_names={n: i for i, n in enumerate(FINGERTIPS)})
        self._fingers = SyntheticRigidPrimView("finger_view", count * 5, device)
        self._actuated_dof_indices = list(SHADOW_HAND_ACTUATED)
        self._sensor_indices = torch.tensor([self._body_indices[j] for j in FINGERTIPS], device=device, dtype=torch.long)

    @property
    def actuated_dof_indices(self):
        return self._actuated_dof_indices


class ShadowHandTask(InHandManipulationTask):
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
        if not (self.obs_type in ["openai", "full_no_vel", "full", "full_state"]):
            raise Exception(
                "Unknown type of observations!\nobservationType should be one of: [openai, full_no_vel, full, full_state]"
            )
        print("Obs type:", self.obs_type)
        self.num_obs_dict = {"openai": 42, "full_no_vel": 77, "full": 157, "full_state": 187}
        self.asymmetric_obs = self._task_cfg["env"]["asymmetric_observations"]
        self.use_vel_obs = False
        self.fingertip_obs = True
        self.fingertips = ["robot0:ffdistal", "robot0:mfdistal", "robot0:rfdistal", "robot0:lfdistal", "robot0:thdistal"]
        self.num_fingertips = len(self.fingertips)
        self.object_scale = torch.tensor([1.0, 1.0, 1.0])
        self.force_torque_obs_scale = 10.0
        num_states = 0
        if self.asymmetric_obs:
            num_states = 187
        self._num_observations = self.num_obs_dict[self.obs_type]
        self._num_actions = 20
        self._num_states = num_states
        InHandManipulationTask.update_config(self)

    def get_starting_positions(self):
        self.hand_start_translation = torch.tensor([0.0, 0.0, 0.5], device=self.device)
        self.hand_start_orientation = torch.tensor([1.0, 0.0, 0.0, 0.0], device=self.device)
        self.pose_dy, self.pose_dz = -0.39, 0.10

    def get_hand_view(self, scene=None):
        view = SyntheticShadowHandView("shadow_hand_view", self._num_envs, self.device)
        view.positions[:] = self._env_pos + self.hand_start_translation
        return view
