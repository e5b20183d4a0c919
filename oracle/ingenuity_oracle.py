"""ORACLE (test infrastructure only) -- Ingenuity task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/ingenuity.py, op for op:
  * get_observations   :128-147  (13 columns: target offset / 3, quat, linvel / 2, angvel / pi)
  * calculate_metrics  :244-268, is_done :270-281 (NOT sticky; target_dist > 20, z < 0.5, progress >= max - 1)
  * pre_physics_step   :149-190  (reset_idx of the flagged envs; set_targets of the envs with progress % 500 == 0 -- which
                                  includes the envs just reset; two thrust vectors from six actions; rotor spin velocities)
  * set_targets        :213-223, reset_idx :225-249
Draw order: reset_idx: dof_pos[:, 1], dof_pos[:, 3], x, y, z ((n,1) each); set_targets: (m,2), then (m,).
Pinning: tests/test_oracle_pinned_ingenuity.py (container only), bit for bit; ``quat_axis`` is a restated omni helper.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import torch

from . import torch_utils as tu

Tensor = torch.Tensor


@dataclass
class IngenuitySpec:
    """cfg/task/Ingenuity.yaml:5-13; ingenuity.py:56-57."""

    max_episode_length: int = 2000
    dt: float = 0.01
    thrust_limit: float = 2000
    thrust_lateral_component: float = 0.2


@dataclass
class IngenuityBuffers:
    obs_buf: Tensor            # (N,13)
    rew_buf: Tensor
    reset_buf: Tensor          # int64
    progress_buf: Tensor       # int64
    target_positions: Tensor   # (N,3)
    thrusts: Tensor            # (N,2,3)
    dof_pos: Tensor            # (N,4) the task's own copy (:201)
    dof_vel: Tensor            # (N,4)

    def clone(self) -> "IngenuityBuffers":
        return IngenuityBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def observations(b: IngenuityBuffers, root_pos, root_rot, root_vel, env_pos) -> None:
    root_positions = root_pos - env_pos
    b.obs_buf[..., 0:3] = (b.target_positions - root_positions) / 3
    b.obs_buf[..., 3:7] = root_rot
    b.obs_buf[..., 7:10] = root_vel[:, :3] / 2
    b.obs_buf[..., 10:13] = root_vel[:, 3:] / math.pi


def metrics(b: IngenuityBuffers, root_pos, root_rot, root_vel, env_pos):
    root_positions = root_pos - env_pos
    root_angvels = root_vel[:, 3:]
    target_dist = torch.sqrt(torch.square(b.target_positions - root_positions).sum(-1))
    pos_reward = 1.0 / (1.0 + 2.5 * target_dist * target_dist)
    ups = tu.quat_axis(root_rot, 2)
    tiltage = torch.abs(1 - ups[..., 2])
    up_reward = 1.0 / (1.0 + 30 * tiltage * tiltage)
    spinnage = torch.abs(root_angvels[..., 2])
    spinnage_reward = 1.0 / (1.0 + 10 * spinnage * spinnage)
    b.rew_buf[:] = pos_reward + pos_reward * (up_reward + spinnage_reward)
    return target_dist, root_positions


def is_done(spec: IngenuitySpec, b: IngenuityBuffers, target_dist, root_positions) -> None:
    ones = torch.ones_like(b.reset_buf)
    die = torch.zeros_like(b.reset_buf)
    die = torch.where(target_dist > 20.0, ones, die)
    die = torch.where(root_positions[..., 2] < 0.5, ones, die)
    b.reset_buf[:] = torch.where(b.progress_buf >= spec.max_episode_length - 1, ones, die)


def post_physics(spec: IngenuitySpec, b: IngenuityBuffers, root_pos, root_rot, root_vel, env_pos) -> None:
    b.progress_buf[:] += 1  # rl_task.py:289
    observations(b, root_pos, root_rot, root_vel, env_pos)
    td, rp = metrics(b, root_pos, root_rot, root_vel, env_pos)
    is_done(spec, b, td, rp)


def reset_idx(b: IngenuityBuffers, env_ids: Tensor, initial_root_pos, initial_root_rot, u: Tensor):
    """:225-249 with the five (n,1) U[0,1) draws as the columns of ``u`` (n,5)."""
    n = len(env_ids)
    U = lambda lo, hi, x: (hi - lo) * x + lo
    b.dof_pos[env_ids, 1] = U(-0.2, 0.2, u[:, 0:1]).squeeze(-1)
    b.dof_pos[env_ids, 3] = U(-0.2, 0.2, u[:, 1:2]).squeeze(-1)
    b.dof_vel[env_ids, :] = 0
    root_pos = initial_root_pos.clone()
    root_pos[env_ids, 0] += U(-0.5, 0.5, u[:, 2:3]).view(-1)
    root_pos[env_ids, 1] += U(-0.5, 0.5, u[:, 3:4]).view(-1)
    root_pos[env_ids, 2] += U(-0.5, 0.5, u[:, 4:5]).view(-1)
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    return dict(dof_pos=b.dof_pos[env_ids].clone(), dof_vel=b.dof_vel[env_ids].clone(), root_pos=root_pos[env_ids],
                root_rot=initial_root_rot[env_ids].clone(), root_vel=torch.zeros(n, 6))


def set_targets(b: IngenuityBuffers, env_ids: Tensor, env_pos, initial_ball_rot, u_xy: Tensor, u_z: Tensor):
    """:213-223 with the draws supplied: (m,2) and (m,) U[0,1)."""
    b.target_positions[env_ids, 0:2] = u_xy * 2 - 1
    b.target_positions[env_ids, 2] = u_z + 1
    ball_pos = b.target_positions[env_ids] + env_pos[env_ids]
    ball_pos[:, 2] += 0.4
    return dict(ball_pos=ball_pos[:, 0:3], ball_rot=initial_ball_rot[env_ids].clone())


def pre_physics(spec: IngenuitySpec, b: IngenuityBuffers, actions: Tensor, env_pos, initial_root_pos, initial_root_rot,
                initial_ball_rot, draw_reset, draw_targets):
    """:149-190.  ``draw_reset(ids) -> (n,5)``, ``draw_targets(ids) -> ((m,2), (m,))``.
    Returns (reset ids, reset rows | None, target ids, target rows | None)."""
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = trows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(b, reset_env_ids, initial_root_pos, initial_root_rot, draw_reset(reset_env_ids))
    set_target_ids = (b.progress_buf % 500 == 0).nonzero(as_tuple=False).squeeze(-1)
    if len(set_target_ids) > 0:
        trows = set_targets(b, set_target_ids, env_pos, initial_ball_rot, *draw_targets(set_target_ids))
    actions = actions.clone()
    tl, lat = spec.thrust_limit, spec.thrust_lateral_component
    vertical_thrust_prop_0 = torch.clamp(actions[:, 2] * tl, -tl, tl)
    vertical_thrust_prop_1 = torch.clamp(actions[:, 5] * tl, -tl, tl)
    lateral_fraction_prop_0 = torch.clamp(actions[:, 0:2] * lat, -lat, lat)
    lateral_fraction_prop_1 = torch.clamp(actions[:, 3:5] * lat, -lat, lat)
    b.thrusts[:, 0, 2] = spec.dt * vertical_thrust_prop_0
    b.thrusts[:, 0, 0:2] = b.thrusts[:, 0, 2, None] * lateral_fraction_prop_0
    b.thrusts[:, 1, 2] = spec.dt * vertical_thrust_prop_1
    b.thrusts[:, 1, 0:2] = b.thrusts[:, 1, 2, None] * lateral_fraction_prop_1
    b.thrusts[reset_env_ids] = 0
    b.dof_vel[:, 1] = 50
    b.dof_vel[:, 3] = -50
    return reset_env_ids, rows, set_target_ids, trows
