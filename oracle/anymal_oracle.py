"""ORACLE (test infrastructure only) -- Anymal (flat ground) per-step task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/anymal.py:
  get_observations :143-178, pre_physics_step :180-194, reset_idx :196-231, calculate_metrics :274-309,
  is_done :311-314 (NOT sticky: ``reset_buf[:] = time_out | fallen_over``), in RLTask.post_physics_step's order
  (tasks/base/rl_task.py:289-295), and AnymalView.is_base_below_threshold
  (robots/articulations/views/anymal_view.py:76-80).
Quirks kept: projected_gravity uses quat_rotate, not its inverse (:154); ``last_actions`` / ``last_dof_vel`` are
refreshed inside calculate_metrics (:304-305); fallen envs get reward -1 after the clip (:308).

Pinning: tests/test_oracle_pinned_anymal_flat.py runs the reference methods from their own source on a stub ``self``
(bit for bit, container only); quat_rotate / quat_rotate_inverse are the restated omni helpers (UNPINNED layer).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

import torch

from . import torch_utils as tu

Tensor = torch.Tensor

# Anymal.dof_names: taken to be the YAML's listing order (cfg/task/Anymal.yaml:33-48), as for AnymalTerrain
DEFAULT_JOINT_ANGLES = [0.03, 0.03, -0.03, -0.03, 0.4, -0.4, 0.4, -0.4, -0.8, 0.8, -0.8, 0.8]


@dataclass
class AnymalFlatSpec:
    """cfg/task/Anymal.yaml:7-66 through update_config (:51-101)."""

    lin_vel_scale: float = 2.0
    ang_vel_scale: float = 0.25
    dof_pos_scale: float = 1.0
    dof_vel_scale: float = 0.05
    action_scale: float = 13.5
    dt: float = 1 / 60  # :90
    max_episode_length_s: float = 50
    command_x_range: tuple = (-2.0, 2.0)
    command_y_range: tuple = (-1.0, 1.0)
    command_yaw_range: tuple = (-1.0, 1.0)
    raw_rew_scales: Dict[str, float] = field(default_factory=lambda: dict(
        lin_vel_xy=1.0, ang_vel_z=0.5, lin_vel_z=-0.03, joint_acc=-0.0003, action_rate=-0.006, cosmetic=-0.06))
    base_threshold: float = 0.51  # :307

    @property
    def max_episode_length(self) -> int:
        return int(self.max_episode_length_s / self.dt + 0.5)  # :92

    @property
    def rew_scales(self) -> Dict[str, float]:
        out = {}
        for k, v in self.raw_rew_scales.items():  # :96-97
            v *= self.dt
            out[k] = v
        return out


@dataclass
class AnymalFlatBuffers:
    obs_buf: Tensor          # (N,48)
    rew_buf: Tensor
    reset_buf: Tensor        # int64
    progress_buf: Tensor     # int64
    commands: Tensor         # (N,3)
    actions: Tensor          # (N,12)
    last_actions: Tensor
    last_dof_vel: Tensor
    current_targets: Tensor  # (N,12)
    default_dof_pos: Tensor  # (N,12)

    def clone(self) -> "AnymalFlatBuffers":
        return AnymalFlatBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def observations(spec: AnymalFlatSpec, b: AnymalFlatBuffers, rot: Tensor, vel: Tensor, dof_pos: Tensor, dof_vel: Tensor) -> Tensor:
    n = rot.shape[0]
    gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat((n, 1))
    base_lin_vel = tu.quat_rotate_inverse(rot, vel[:, 0:3]) * spec.lin_vel_scale
    base_ang_vel = tu.quat_rotate_inverse(rot, vel[:, 3:6]) * spec.ang_vel_scale
    projected_gravity = tu.quat_rotate(rot, gravity_vec)
    dof_pos_scaled = (dof_pos - b.default_dof_pos) * spec.dof_pos_scale
    commands_scaled = b.commands * torch.tensor([spec.lin_vel_scale, spec.lin_vel_scale, spec.ang_vel_scale])
    return torch.cat((base_lin_vel, base_ang_vel, projected_gravity, commands_scaled, dof_pos_scaled,
                      dof_vel * spec.dof_vel_scale, b.actions), dim=-1)


def metrics(spec: AnymalFlatSpec, b: AnymalFlatBuffers, pos: Tensor, rot: Tensor, vel: Tensor, dof_pos: Tensor,
            dof_vel: Tensor) -> Tuple[Tensor, Tensor]:
    """calculate_metrics (:274-309): returns (reward, fallen_over) and refreshes last_actions / last_dof_vel."""
    rs = spec.rew_scales
    base_lin_vel = tu.quat_rotate_inverse(rot, vel[:, 0:3])
    base_ang_vel = tu.quat_rotate_inverse(rot, vel[:, 3:6])
    lin_vel_error = torch.sum(torch.square(b.commands[:, :2] - base_lin_vel[:, :2]), dim=1)
    ang_vel_error = torch.square(b.commands[:, 2] - base_ang_vel[:, 2])
    rew_lin_vel_xy = torch.exp(-lin_vel_error / 0.25) * rs["lin_vel_xy"]
    rew_ang_vel_z = torch.exp(-ang_vel_error / 0.25) * rs["ang_vel_z"]
    rew_lin_vel_z = torch.square(base_lin_vel[:, 2]) * rs["lin_vel_z"]
    rew_joint_acc = torch.sum(torch.square(b.last_dof_vel - dof_vel), dim=1) * rs["joint_acc"]
    rew_action_rate = torch.sum(torch.square(b.last_actions - b.actions), dim=1) * rs["action_rate"]
    rew_cosmetic = torch.sum(torch.abs(dof_pos[:, 0:4] - b.default_dof_pos[:, 0:4]), dim=1) * rs["cosmetic"]
    total_reward = rew_lin_vel_xy + rew_ang_vel_z + rew_joint_acc + rew_action_rate + rew_cosmetic + rew_lin_vel_z
    total_reward = torch.clip(total_reward, 0.0, None)
    b.last_actions[:] = b.actions[:]
    b.last_dof_vel[:] = dof_vel[:]
    fallen_over = (pos[:, 2] - 0.0) < spec.base_threshold  # anymal_view.py:76-80 with ground_heights=0.0
    total_reward[torch.nonzero(fallen_over)] = -1
    return total_reward, fallen_over


def post_physics(spec: AnymalFlatSpec, b: AnymalFlatBuffers, pos, rot, vel, dof_pos, dof_vel) -> Tensor:
    """RLTask.post_physics_step (rl_task.py:289-295) for this task; returns fallen_over."""
    b.progress_buf[:] += 1
    b.obs_buf[:] = observations(spec, b, rot, vel, dof_pos, dof_vel)
    rew, fallen_over = metrics(spec, b, pos, rot, vel, dof_pos, dof_vel)
    b.rew_buf[:] = rew
    time_out = b.progress_buf >= spec.max_episode_length - 1
    b.reset_buf[:] = time_out | fallen_over
    return fallen_over


@dataclass
class AnymalFlatResetRows:
    env_ids: Tensor
    dof_pos: Tensor
    dof_vel: Tensor
    root_pos: Tensor
    root_rot: Tensor
    root_vel: Tensor


def reset_idx(spec: AnymalFlatSpec, b: AnymalFlatBuffers, env_ids: Tensor, initial_root_pos: Tensor, initial_root_rot: Tensor,
              u_vel: Tensor, u_x: Tensor, u_y: Tensor, u_yaw: Tensor) -> AnymalFlatResetRows:
    """:196-231 with the four torch_rand_float draws given as U[0,1) tensors ((n,12), (n,1), (n,1), (n,1))."""
    U = tu.uniform_from_unit
    n = len(env_ids)
    velocities = U(-0.1, 0.1, u_vel)
    dof_pos = b.default_dof_pos[env_ids]
    b.current_targets[env_ids] = dof_pos[:]
    b.commands[env_ids, 0] = U(spec.command_x_range[0], spec.command_x_range[1], u_x).squeeze()
    b.commands[env_ids, 1] = U(spec.command_y_range[0], spec.command_y_range[1], u_y).squeeze()
    b.commands[env_ids, 2] = U(spec.command_yaw_range[0], spec.command_yaw_range[1], u_yaw).squeeze()
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    b.last_actions[env_ids] = 0.0
    b.last_dof_vel[env_ids] = 0.0
    return AnymalFlatResetRows(env_ids, dof_pos.clone(), velocities, initial_root_pos[env_ids].clone(),
                               initial_root_rot[env_ids].clone(), torch.zeros((n, 6)))


def pre_physics(spec: AnymalFlatSpec, b: AnymalFlatBuffers, actions: Tensor, lower: Tensor, upper: Tensor,
                initial_root_pos: Tensor, initial_root_rot: Tensor, draw=None) -> Optional[AnymalFlatResetRows]:
    """:180-194.  ``draw(ids) -> (u_vel, u_x, u_y, u_yaw)``."""
    rows = None
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    if len(reset_env_ids) > 0:
        rows = reset_idx(spec, b, reset_env_ids, initial_root_pos, initial_root_rot, *draw(reset_env_ids))
    b.actions[:] = actions.clone()
    current_targets = b.current_targets + spec.action_scale * b.actions * spec.dt
    b.current_targets[:] = tu.tensor_clamp(current_targets, lower, upper)
    return rows
