"""ORACLE (test infrastructure only) -- Crazyflie task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/crazyflie.py, op for op:
  * get_observations   :140-164  (18 columns: target offset, the three body axes, linvel, angvel)
  * calculate_metrics  :337-378  (reward + the eight episode sums), is_done :380-395 (NOT sticky)
  * pre_physics_step   :166-258  (reset_idx, set_targets of progress % 500 == 0, first-order motor filter on the rotor speed,
                                  one shared 4-vector of thrust noise per step, thrust along the body z axis, rotor speeds)
  * set_targets        :300-311, reset_idx :313-335 (its four torch_rand_float(-0.0, 0.0, ...) draws are all zero but still
                                  consume the generator; extras["episode"] = means of the episode sums over the reset ids)
Pinning: tests/test_oracle_pinned_crazyflie.py (container only), bit for bit; ``quat_axis`` is a restated omni helper.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict

import numpy as np
import torch

from . import torch_utils as tu

Tensor = torch.Tensor
EPS = 1e-6
EPISODE_KEYS = ["rew_pos", "rew_orient", "rew_effort", "rew_spin", "raw_dist", "raw_orient", "raw_effort", "raw_spin"]


@dataclass
class CrazyflieSpec:
    """cfg/task/Crazyflie.yaml:5-13; crazyflie.py:57-90, :262-268."""

    max_episode_length: int = 700
    dt: float = 0.01
    gravity_z: float = -9.81
    motor_damp_time_up: float = 0.15
    motor_damp_time_down: float = 0.15
    mass: float = 0.028
    thrust_to_weight: float = 1.9
    prop_max_rot: float = 433.3

    @property
    def motor_tau_up(self) -> float:
        return 4 * self.dt / (self.motor_damp_time_up + EPS)

    @property
    def motor_tau_down(self) -> float:
        return 4 * self.dt / (self.motor_damp_time_down + EPS)

    @property
    def thrust_max(self) -> Tensor:
        motor_assymetry = np.array([1.0, 1.0, 1.0, 1.0])
        motor_assymetry = motor_assymetry * 4.0 / np.sum(motor_assymetry)
        grav_z = -1.0 * self.gravity_z
        return torch.tensor(grav_z * self.mass * self.thrust_to_weight * motor_assymetry / 4.0, dtype=torch.float32)


@dataclass
class CrazyflieBuffers:
    obs_buf: Tensor            # (N,18)
    rew_buf: Tensor
    reset_buf: Tensor          # int64
    progress_buf: Tensor       # int64
    target_positions: Tensor   # (N,3)
    actions: Tensor            # (N,4)
    thrusts: Tensor            # (N,4,3)
    thrust_cmds_damp: Tensor   # (N,4)
    thrust_rot_damp: Tensor    # (N,4)
    dof_pos: Tensor            # (N,4)
    dof_vel: Tensor            # (N,4)
    episode_sums: Dict[str, Tensor] = field(default_factory=dict)
    extras: Dict = field(default_factory=dict)

    def clone(self) -> "CrazyflieBuffers":
        d = {k: v.clone() for k, v in self.__dict__.items() if isinstance(v, torch.Tensor)}
        d["episode_sums"] = {k: v.clone() for k, v in self.episode_sums.items()}
        d["extras"] = {}
        return CrazyflieBuffers(**d)


def observations(b: CrazyflieBuffers, root_pos, root_rot, root_vel, env_pos) -> None:
    root_positions = root_pos - env_pos
    b.obs_buf[..., 0:3] = b.target_positions - root_positions
    b.obs_buf[..., 3:6] = tu.quat_axis(root_rot, 0)
    b.obs_buf[..., 6:9] = tu.quat_axis(root_rot, 1)
    b.obs_buf[..., 9:12] = tu.quat_axis(root_rot, 2)
    b.obs_buf[..., 12:15] = root_vel[:, :3]
    b.obs_buf[..., 15:18] = root_vel[:, 3:]


def metrics(b: CrazyflieBuffers, root_pos, root_rot, root_vel, env_pos):
    root_positions = root_pos - env_pos
    root_angvels = root_vel[:, 3:]
    target_dist = torch.sqrt(torch.square(b.target_positions - root_positions).sum(-1))
    pos_reward = 1.0 / (1.0 + target_dist)
    ups = tu.quat_axis(root_rot, 2)
    orient_z = ups[..., 2]
    up_reward = torch.clamp(ups[..., 2], min=0.0, max=1.0)
    effort = torch.square(b.actions).sum(-1)
    effort_reward = 0.05 * torch.exp(-0.5 * effort)
    spin = torch.square(root_angvels).sum(-1)
    spin_reward = 0.01 * torch.exp(-1.0 * spin)
    b.rew_buf[:] = pos_reward + pos_reward * (up_reward + spin_reward) - effort_reward
    b.episode_sums["rew_pos"] += pos_reward
    b.episode_sums["rew_orient"] += up_reward
    b.episode_sums["rew_effort"] += effort_reward
    b.episode_sums["rew_spin"] += spin_reward
    b.episode_sums["raw_dist"] += target_dist
    b.episode_sums["raw_orient"] += ups[..., 2]
    b.episode_sums["raw_effort"] += effort
    b.episode_sums["raw_spin"] += spin
    return target_dist, root_positions, orient_z


def is_done(spec: CrazyflieSpec, b: CrazyflieBuffers, target_dist, root_positions, orient_z) -> None:
    ones = torch.ones_like(b.reset_buf)
    die = torch.zeros_like(b.reset_buf)
    die = torch.where(target_dist > 5.0, ones, die)
    die = torch.where(root_positions[..., 2] < 0.5, ones, die)
    die = torch.where(root_positions[..., 2] > 5.0, ones, die)
    die = torch.where(orient_z < 0.0, ones, die)
    b.reset_buf[:] = torch.where(b.progress_buf >= spec.max_episode_length - 1, ones, die)


def post_physics(spec: CrazyflieSpec, b: CrazyflieBuffers, root_pos, root_rot, root_vel, env_pos) -> None:
    b.progress_buf[:] += 1  # rl_task.py:289
    observations(b, root_pos, root_rot, root_vel, env_pos)
    td, rp, oz = metrics(b, root_pos, root_rot, root_vel, env_pos)
    is_done(spec, b, td, rp, oz)


def reset_idx(spec: CrazyflieSpec, b: CrazyflieBuffers, env_ids: Tensor, initial_root_pos, initial_root_rot):
    """:313-335 (every draw is torch_rand_float(-0.0, 0.0, ...) == 0: the caller burns the generator)."""
    n = len(env_ids)
    b.dof_pos[env_ids, :] = 0.0
    b.dof_vel[env_ids, :] = 0
    root_pos = initial_root_pos.clone()
    root_pos[env_ids, 0] += 0.0
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    b.thrust_cmds_damp[env_ids] = 0
    b.thrust_rot_damp[env_ids] = 0
    b.extras["episode"] = {}
    for key in b.episode_sums.keys():
        b.extras["episode"][key] = torch.mean(b.episode_sums[key][env_ids]) / spec.max_episode_length
        b.episode_sums[key][env_ids] = 0.0
    return dict(dof_pos=b.dof_pos[env_ids].clone(), dof_vel=b.dof_vel[env_ids].clone(), root_pos=root_pos[env_ids],
                root_rot=initial_root_rot[env_ids].clone(), root_vel=torch.zeros(n, 6))


def set_targets(b: CrazyflieBuffers, env_ids: Tensor, env_pos, initial_ball_rot):
    n = len(env_ids)
    b.target_positions[env_ids, 0:2] = torch.zeros((n, 2))
    b.target_positions[env_ids, 2] = torch.ones(n) * 2.0
    ball_pos = b.target_positions[env_ids] + env_pos[env_ids]
    ball_pos[:, 2] += 0.0
    return dict(ball_pos=ball_pos[:, 0:3], ball_rot=initial_ball_rot[env_ids].clone())


def pre_physics(spec: CrazyflieSpec, b: CrazyflieBuffers, actions: Tensor, root_rot, env_pos, initial_root_pos, initial_root_rot,
                initial_ball_rot, thrust_noise4: Tensor):
    """:166-258.  ``thrust_noise4`` = the step's ``0.01 * torch.randn(4)`` (:197).
    Returns (reset ids, reset rows | None, target ids, target rows | None)."""
    n = actions.shape[0]
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = trows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(spec, b, reset_env_ids, initial_root_pos, initial_root_rot)
    set_target_ids = (b.progress_buf % 500 == 0).nonzero(as_tuple=False).squeeze(-1)
    if len(set_target_ids) > 0:
        trows = set_targets(b, set_target_ids, env_pos, initial_ball_rot)
    actions = actions.clone()
    b.actions = actions
    thrust_cmds = torch.clamp(actions, min=-1.0, max=1.0)
    thrust_cmds = (thrust_cmds + 1.0) / 2.0
    motor_tau = spec.motor_tau_up * torch.ones((n, 4), dtype=torch.float32)
    motor_tau[thrust_cmds < b.thrust_cmds_damp] = spec.motor_tau_down
    motor_tau[motor_tau > 1.0] = 1.0
    thrust_rot = thrust_cmds**0.5
    b.thrust_rot_damp = motor_tau * (thrust_rot - b.thrust_rot_damp) + b.thrust_rot_damp
    b.thrust_cmds_damp = b.thrust_rot_damp**2
    thrust_noise = thrust_cmds * thrust_noise4
    b.thrust_cmds_damp = torch.clamp(b.thrust_cmds_damp + thrust_noise, min=0.0, max=1.0)
    thrusts = spec.thrust_max * b.thrust_cmds_damp
    rot_x, rot_y, rot_z = tu.quat_axis(root_rot, 0), tu.quat_axis(root_rot, 1), tu.quat_axis(root_rot, 2)
    rot_matrix = torch.cat((rot_x, rot_y, rot_z), 1).reshape(-1, 3, 3)
    force_xy = torch.zeros(n, 8).reshape(-1, 4, 2)
    thrusts = torch.cat((force_xy, thrusts.reshape(-1, 4, 1)), 2)
    for i in range(4):
        b.thrusts[:, i] = torch.squeeze(torch.matmul(rot_matrix, thrusts[:, i][:, :, None]), -1)
    b.thrusts[reset_env_ids] = 0
    prop_rot = b.thrust_cmds_damp * spec.prop_max_rot
    b.dof_vel[:, 0] = prop_rot[:, 0]
    b.dof_vel[:, 1] = -1.0 * prop_rot[:, 1]
    b.dof_vel[:, 2] = prop_rot[:, 2]
    b.dof_vel[:, 3] = -1.0 * prop_rot[:, 3]
    return reset_env_ids, rows, set_target_ids, trows
