"""ORACLE (test infrastructure only) -- Quadcopter task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/quadcopter.py, op for op:
  * get_observations   :124-141  (21 columns: target offset / 3, quat, linvel / 2, angvel / pi, 8 DOF positions)
  * calculate_metrics  :229-251  (three inverse-quadratic terms, clip at 0)
  * is_done            :253-261  (NOT sticky; target_dist > 3, z < 0.3, progress >= max - 1)
  * pre_physics_step   :143-175  (reset_idx of the flagged envs, integrate DOF targets and thrusts, clamp, z forces,
                                  reset envs: thrusts = forces = 0, targets = their re-drawn DOF positions)
  * reset_idx          :203-227  (draw order: (n,8) DOF positions, then x, y, z offsets (n,1) each)
Inputs are const: the reference mutates the view's dof_pos / dof_vel buffers in place (:206-207); here the re-drawn rows
are returned instead.

Pinning: tests/test_oracle_pinned_quadcopter.py runs the reference's method source on a stub ``self`` (container only) and
requires equality bit for bit; ``quat_axis`` underneath is a restated omni helper (UNPINNED, oracle/torch_utils.py).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

from . import torch_utils as tu

Tensor = torch.Tensor


@dataclass
class QuadcopterSpec:
    """cfg/task/Quadcopter.yaml:7-17; quadcopter.py:46-55."""

    max_episode_length: int = 500
    dt: float = 0.01
    max_thrust: float = 2.0
    num_dof: int = 8
    lower: Optional[Tensor] = None  # (8,) DOF limits (USD asset; stand-in +-0.8)
    upper: Optional[Tensor] = None

    def __post_init__(self):
        if self.lower is None:
            self.lower = -0.8 * torch.ones(8)
            self.upper = 0.8 * torch.ones(8)


@dataclass
class QuadcopterBuffers:
    obs_buf: Tensor                # (N,21)
    rew_buf: Tensor
    reset_buf: Tensor              # int64
    progress_buf: Tensor           # int64
    target_positions: Tensor       # (N,3)
    dof_position_targets: Tensor   # (N,8)
    thrusts: Tensor                # (N,4)
    forces: Tensor                 # (N,4,3)

    def clone(self) -> "QuadcopterBuffers":
        return QuadcopterBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def observations(b: QuadcopterBuffers, root_pos, root_rot, root_vel, dof_pos, env_pos) -> None:
    root_positions = root_pos - env_pos
    b.obs_buf[..., 0:3] = (b.target_positions - root_positions) / 3
    b.obs_buf[..., 3:7] = root_rot
    b.obs_buf[..., 7:10] = root_vel[:, :3] / 2
    b.obs_buf[..., 10:13] = root_vel[:, 3:] / math.pi
    b.obs_buf[..., 13:21] = dof_pos


def metrics(b: QuadcopterBuffers, root_pos, root_rot, root_vel, env_pos):
    """Returns (target_dist, root_positions), which is_done reads (:237-238)."""
    root_positions = root_pos - env_pos
    root_angvels = root_vel[:, 3:]
    target_dist = torch.sqrt(torch.square(b.target_positions - root_positions).sum(-1))
    pos_reward = 1.0 / (1.0 + 3 * target_dist * target_dist)
    ups = tu.quat_axis(root_rot, 2)
    tiltage = torch.abs(1 - ups[..., 2])
    up_reward = 1.0 / (1.0 + 10 * tiltage * tiltage)
    spinnage = torch.abs(root_angvels[..., 2])
    spinnage_reward = 1.0 / (1.0 + 0.001 * spinnage * spinnage)
    rew = pos_reward + pos_reward * (up_reward + spinnage_reward + spinnage * spinnage * (-1 / 400))
    b.rew_buf[:] = torch.clip(rew, 0.0, None)
    return target_dist, root_positions


def is_done(spec: QuadcopterSpec, b: QuadcopterBuffers, target_dist, root_positions) -> None:
    ones = torch.ones_like(b.reset_buf)
    die = torch.zeros_like(b.reset_buf)
    die = torch.where(target_dist > 3.0, ones, die)
    die = torch.where(root_positions[..., 2] < 0.3, ones, die)
    b.reset_buf[:] = torch.where(b.progress_buf >= spec.max_episode_length - 1, ones, die)


def post_physics(spec: QuadcopterSpec, b: QuadcopterBuffers, root_pos, root_rot, root_vel, dof_pos, env_pos) -> None:
    """rl_task.py:289-295 order."""
    b.progress_buf[:] += 1
    observations(b, root_pos, root_rot, root_vel, dof_pos, env_pos)
    td, rp = metrics(b, root_pos, root_rot, root_vel, env_pos)
    is_done(spec, b, td, rp)


@dataclass
class QuadcopterResetRows:
    dof_pos: Tensor    # (n,8)
    dof_vel: Tensor    # (n,8) zeros
    root_pos: Tensor   # (n,3)
    root_rot: Tensor   # (n,4)
    root_vel: Tensor   # (n,6) zeros


def reset_idx(b: QuadcopterBuffers, env_ids: Tensor, initial_root_pos, initial_root_rot, u_dof, u_x, u_y, u_z) -> QuadcopterResetRows:
    """:203-227 with the four U[0,1) draws supplied: (n,8), (n,1), (n,1), (n,1)."""
    n = len(env_ids)
    U = lambda lo, hi, u: (hi - lo) * u + lo  # torch_rand_float
    dof_pos = U(-0.2, 0.2, u_dof)
    root_pos = initial_root_pos.clone()
    root_pos[env_ids, 0] += U(-1.5, 1.5, u_x).view(-1)
    root_pos[env_ids, 1] += U(-1.5, 1.5, u_y).view(-1)
    root_pos[env_ids, 2] += U(-0.2, 1.5, u_z).view(-1)
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    return QuadcopterResetRows(dof_pos, torch.zeros(n, 8), root_pos[env_ids], initial_root_rot[env_ids].clone(), torch.zeros(n, 6))


def pre_physics(spec: QuadcopterSpec, b: QuadcopterBuffers, actions: Tensor, initial_root_pos, initial_root_rot, draw):
    """:143-175.  ``draw(ids) -> (u_dof (n,8), u_x, u_y, u_z (n,1))`` in the reference's order.  Returns (env_ids, rows | None)."""
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(b, reset_env_ids, initial_root_pos, initial_root_rot, *draw(reset_env_ids))
    actions = actions.clone()
    dof_action_speed_scale = 8 * math.pi
    b.dof_position_targets += spec.dt * dof_action_speed_scale * actions[:, 0:8]
    b.dof_position_targets[:] = tu.tensor_clamp(b.dof_position_targets, spec.lower, spec.upper)
    thrust_action_speed_scale = 100
    b.thrusts += spec.dt * thrust_action_speed_scale * actions[:, 8:12]
    lim = spec.max_thrust * torch.ones(4)
    b.thrusts[:] = tu.tensor_clamp(b.thrusts, -lim, lim)
    for i in range(4):
        b.forces[:, i, 2] = b.thrusts[:, i]
    b.thrusts[reset_env_ids] = 0.0
    b.forces[reset_env_ids] = 0.0
    if rows is not None:
        b.dof_position_targets[reset_env_ids] = rows.dof_pos  # = self.dof_pos[reset_env_ids], mutated in place by reset_idx
    return reset_env_ids, rows
