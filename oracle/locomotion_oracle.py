"""ORACLE (test infrastructure only) -- Ant / Humanoid per-step task math on CPU torch.

Restates, op for op and in the reference's evaluation order:
  * jit get_observations      /root/reference/omniisaacgymenvs/tasks/shared/locomotion.py:225-284
  * jit normalize_angle       locomotion.py:220-221
  * jit calculate_metrics     locomotion.py:296-345
  * jit is_done               locomotion.py:288-292
  * Ant limit cost            tasks/ant.py:107-109        (int64 count, no abs)
  * Humanoid limit cost       tasks/humanoid.py:135-141
  * pre_physics_step          locomotion.py:117-131
  * reset_idx                 locomotion.py:133-158
  * post_physics_step order   tasks/base/rl_task.py:278-298 (progress += 1 first)

Pinning: calculate_metrics / is_done / both limit costs / normalize_angle are checked
bit-for-bit against the reference's own function source executed under local torch
(tests/test_oracle_pinned.py, container only) and against tests/golden/*.npz.
get_observations depends on the un-vendored omni helpers (oracle/torch_utils.py):
PARITY UNPINNED at the helper layer, composition pinned by executing the reference's
function source on top of those helpers.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import torch

from . import torch_utils as tu

Tensor = torch.Tensor


@dataclass
class LocomotionSpec:
    """Scalars a LocomotionTask reads from YAML (locomotion.py:50-65) plus per-robot tables."""

    num_dof: int
    num_sensors: int
    limit_cost_kind: str  # "ant" | "humanoid"
    joint_gears: Tensor
    motor_effort_ratio: Tensor
    dof_limits_lower: Tensor
    dof_limits_upper: Tensor
    episode_length: float = 1000.0
    dof_vel_scale: float = 0.2
    angular_velocity_scale: float = 1.0
    contact_force_scale: float = 0.1
    power_scale: float = 0.5
    heading_weight: float = 0.5
    up_weight: float = 0.1
    actions_cost_scale: float = 0.005
    energy_cost_scale: float = 0.05
    joints_at_limit_cost_scale: float = 0.1
    death_cost: float = -2.0
    termination_height: float = 0.31
    alive_reward_scale: float = 0.5
    dt: float = 1.0 / 60.0  # locomotion.py:178
    target: tuple = (1000.0, 0.0, 0.0)  # locomotion.py:176
    start_rotation: tuple = (1.0, 0.0, 0.0, 0.0)  # locomotion.py:168
    basis0: tuple = (1.0, 0.0, 0.0)  # heading, locomotion.py:170,173
    basis1: tuple = (0.0, 0.0, 1.0)  # up, locomotion.py:169,174

    @property
    def num_obs(self) -> int:
        return 12 + 3 * self.num_dof + 6 * self.num_sensors


def normalize_angle(x: Tensor) -> Tensor:
    return torch.atan2(torch.sin(x), torch.cos(x))


def per_env_constants(spec: LocomotionSpec, n: int) -> Dict[str, Tensor]:
    """The (N,k) constant tensors post_reset builds (locomotion.py:168-177)."""
    start = torch.tensor(spec.start_rotation, dtype=torch.float32)
    return {
        "targets": torch.tensor(spec.target, dtype=torch.float32).repeat((n, 1)),
        "inv_start_rot": tu.quat_conjugate(start).repeat((n, 1)),
        "basis_vec0": torch.tensor(spec.basis0, dtype=torch.float32).repeat((n, 1)),
        "basis_vec1": torch.tensor(spec.basis1, dtype=torch.float32).repeat((n, 1)),
    }


def observations(spec: LocomotionSpec, pos, rot, vel6, dof_pos, dof_vel, sensors, actions, potentials):
    """locomotion.py:75-115 + jit get_observations :225-284.

    Inputs are treated as const (the reference's in-place ``to_target[:, 2] = 0`` acts on
    a temporary).  Returns obs, potentials, prev_potentials, up_vec, heading_vec.
    """
    n = pos.shape[0]
    c = per_env_constants(spec, n)
    velocity, ang_velocity = vel6[:, 0:3], vel6[:, 3:6]

    to_target = c["targets"] - pos
    to_target[:, 2] = 0.0
    prev_potentials = potentials.clone()
    new_potentials = -torch.norm(to_target, p=2, dim=-1) / spec.dt

    torso_quat, up_proj, heading_proj, up_vec, heading_vec = tu.compute_heading_and_up(
        rot, c["inv_start_rot"], to_target, c["basis_vec0"], c["basis_vec1"], 2
    )
    vel_loc, angvel_loc, roll, _pitch, yaw, angle_to_target = tu.compute_rot(
        torso_quat, velocity, ang_velocity, c["targets"], pos
    )
    dof_pos_scaled = tu.unscale(dof_pos, spec.dof_limits_lower, spec.dof_limits_upper)

    obs = torch.cat(
        (
            pos[:, 2].view(-1, 1),
            vel_loc,
            angvel_loc * spec.angular_velocity_scale,
            normalize_angle(yaw).unsqueeze(-1),
            normalize_angle(roll).unsqueeze(-1),
            normalize_angle(angle_to_target).unsqueeze(-1),
            up_proj.unsqueeze(-1),
            heading_proj.unsqueeze(-1),
            dof_pos_scaled,
            dof_vel * spec.dof_vel_scale,
            sensors.reshape(n, -1) * spec.contact_force_scale,
            actions,
        ),
        dim=-1,
    )
    return obs, new_potentials, prev_potentials, up_vec, heading_vec


def dof_at_limit_cost(spec: LocomotionSpec, obs: Tensor) -> Tensor:
    d = spec.num_dof
    if spec.limit_cost_kind == "ant":
        # tasks/ant.py:109 -- count of obs > 0.99 (no abs), int64
        return torch.sum(obs[:, 12 : 12 + d] > 0.99, dim=-1)
    # tasks/humanoid.py:137-140
    scaled_cost = spec.joints_at_limit_cost_scale * (torch.abs(obs[:, 12 : 12 + d]) - 0.98) / 0.02
    return torch.sum((torch.abs(obs[:, 12 : 12 + d]) > 0.98) * scaled_cost * spec.motor_effort_ratio.unsqueeze(0), dim=-1)


def metrics(spec: LocomotionSpec, obs, actions, potentials, prev_potentials) -> Tensor:
    """jit calculate_metrics, locomotion.py:296-345."""
    d = spec.num_dof
    hw = torch.ones_like(obs[:, 11]) * spec.heading_weight
    heading_reward = torch.where(obs[:, 11] > 0.8, hw, spec.heading_weight * obs[:, 11] / 0.8)
    up_reward = torch.zeros_like(heading_reward)
    up_reward = torch.where(obs[:, 10] > 0.93, up_reward + spec.up_weight, up_reward)
    actions_cost = torch.sum(actions**2, dim=-1)
    electricity_cost = torch.sum(
        torch.abs(actions * obs[:, 12 + d : 12 + d * 2]) * spec.motor_effort_ratio.unsqueeze(0), dim=-1
    )
    alive_reward = torch.ones_like(potentials) * spec.alive_reward_scale
    progress_reward = potentials - prev_potentials
    total = (
        progress_reward
        + alive_reward
        + up_reward
        + heading_reward
        - spec.actions_cost_scale * actions_cost
        - spec.energy_cost_scale * electricity_cost
        - dof_at_limit_cost(spec, obs)
    )
    return torch.where(obs[:, 0] < spec.termination_height, torch.ones_like(total) * spec.death_cost, total)


def done(spec: LocomotionSpec, obs, reset_buf, progress_buf) -> Tensor:
    """jit is_done, locomotion.py:288-292 (sticky; max length compared as float)."""
    reset = torch.where(obs[:, 0] < spec.termination_height, torch.ones_like(reset_buf), reset_buf)
    return torch.where(progress_buf >= float(spec.episode_length) - 1, torch.ones_like(reset_buf), reset)


@dataclass
class LocomotionBuffers:
    """Persistent state, SURVEY.md appendix B (rl_task.py:162-166, locomotion.py:179-184)."""

    obs_buf: Tensor
    rew_buf: Tensor
    reset_buf: Tensor
    progress_buf: Tensor
    potentials: Tensor
    prev_potentials: Tensor
    up_vec: Tensor
    heading_vec: Tensor
    actions: Tensor

    @staticmethod
    def fresh(spec: LocomotionSpec, n: int) -> "LocomotionBuffers":
        pot = torch.tensor([-1000.0 / spec.dt], dtype=torch.float32).repeat(n)
        return LocomotionBuffers(
            obs_buf=torch.zeros((n, spec.num_obs), dtype=torch.float32),
            rew_buf=torch.zeros(n, dtype=torch.float32),
            reset_buf=torch.ones(n, dtype=torch.long),
            progress_buf=torch.zeros(n, dtype=torch.long),
            potentials=pot,
            prev_potentials=pot.clone(),
            up_vec=torch.tensor(spec.basis1, dtype=torch.float32).repeat((n, 1)),
            heading_vec=torch.tensor(spec.basis0, dtype=torch.float32).repeat((n, 1)),
            actions=torch.zeros((n, spec.num_dof), dtype=torch.float32),
        )

    def clone(self) -> "LocomotionBuffers":
        return LocomotionBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def post_physics(spec: LocomotionSpec, b: LocomotionBuffers, pos, rot, vel6, dof_pos, dof_vel, sensors) -> None:
    """RLTask.post_physics_step order (rl_task.py:289-296), mutating ``b`` in place."""
    b.progress_buf[:] += 1
    (b.obs_buf[:], b.potentials[:], b.prev_potentials[:], b.up_vec[:], b.heading_vec[:]) = observations(
        spec, pos, rot, vel6, dof_pos, dof_vel, sensors, b.actions, b.potentials
    )
    b.rew_buf[:] = metrics(spec, b.obs_buf, b.actions, b.potentials, b.prev_potentials)
    b.reset_buf[:] = done(spec, b.obs_buf, b.reset_buf, b.progress_buf)


@dataclass
class ResetWrites:
    """What reset_idx hands to the view setters (locomotion.py:145-149)."""

    env_ids: Tensor
    dof_pos: Tensor
    dof_vel: Tensor
    root_pos: Tensor
    root_rot: Tensor
    root_vel: Tensor


def reset_idx(
    spec: LocomotionSpec,
    b: LocomotionBuffers,
    env_ids: Tensor,
    initial_root_pos,
    initial_root_rot,
    initial_dof_pos,
    u_pos: Tensor,
    u_vel: Tensor,
) -> ResetWrites:
    """locomotion.py:133-158 with the two torch.rand draws supplied as U[0,1) tensors of
    shape (num_resets, D): ``u_pos`` is the first draw (:137), ``u_vel`` the second (:139);
    row k belongs to env_ids[k]."""
    n = len(env_ids)
    dof_pos = tu.uniform_from_unit(-0.2, 0.2, u_pos)
    dof_pos[:] = tu.tensor_clamp(initial_dof_pos[env_ids] + dof_pos, spec.dof_limits_lower, spec.dof_limits_upper)
    dof_vel = tu.uniform_from_unit(-0.1, 0.1, u_vel)
    root_pos, root_rot = initial_root_pos[env_ids], initial_root_rot[env_ids]
    root_vel = torch.zeros((n, 6))
    targets = torch.tensor(spec.target, dtype=torch.float32).repeat((initial_root_pos.shape[0], 1))
    to_target = targets[env_ids] - initial_root_pos[env_ids]
    to_target[:, 2] = 0.0
    b.prev_potentials[env_ids] = -torch.norm(to_target, p=2, dim=-1) / spec.dt
    b.potentials[env_ids] = b.prev_potentials[env_ids].clone()
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    return ResetWrites(env_ids, dof_pos, dof_vel, root_pos, root_rot, root_vel)


def pre_physics(
    spec: LocomotionSpec,
    b: LocomotionBuffers,
    actions: Tensor,
    initial_root_pos,
    initial_root_rot,
    initial_dof_pos,
    draw=None,
):
    """locomotion.py:117-131.  ``draw(ids) -> (u_pos, u_vel)`` supplies the two (n, D)
    U[0,1) draws for the compacted, ascending ``ids`` (row k <-> ids[k]); returns
    (forces, ResetWrites | None)."""
    ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    writes = None
    if len(ids) > 0:
        u_pos, u_vel = draw(ids)
        assert u_pos.shape == (len(ids), spec.num_dof) and u_vel.shape == u_pos.shape
        writes = reset_idx(spec, b, ids, initial_root_pos, initial_root_rot, initial_dof_pos, u_pos, u_vel)
    b.actions = actions.clone()
    forces = b.actions * spec.joint_gears * spec.power_scale
    return forces, writes


# ------------------------------------------------------------------ robot tables
def ant_spec() -> LocomotionSpec:
    """cfg/task/Ant.yaml:11-32 + tasks/ant.py:56-57,89-93.  The DOF limits live in the Ant USD
    asset (not in the reference tree); the values below are stand-ins."""
    lo = torch.tensor([-0.7, 0.52, -0.7, -1.75, -0.7, -1.75, -0.7, 0.52], dtype=torch.float32)
    hi = torch.tensor([0.7, 1.75, 0.7, -0.52, 0.7, -0.52, 0.7, 1.75], dtype=torch.float32)
    gears = torch.tensor([15.0] * 8, dtype=torch.float32)
    return LocomotionSpec(
        num_dof=8, num_sensors=4, limit_cost_kind="ant", joint_gears=gears,
        motor_effort_ratio=torch.ones_like(gears), dof_limits_lower=lo, dof_limits_upper=hi,
        episode_length=1000, dof_vel_scale=0.2, angular_velocity_scale=1.0, contact_force_scale=0.1,
        power_scale=0.5, heading_weight=0.5, up_weight=0.1, actions_cost_scale=0.005,
        energy_cost_scale=0.05, joints_at_limit_cost_scale=0.1, death_cost=-2.0,
        termination_height=0.31, alive_reward_scale=0.5,
    )


HUMANOID_GEARS = [67.5] * 7 + [45.0, 45.0, 45.0, 135.0, 45.0, 45.0, 135.0, 45.0, 90.0, 90.0, 22.5, 22.5, 22.5, 22.5]


def humanoid_spec() -> LocomotionSpec:
    """cfg/task/Humanoid.yaml:11-32 + tasks/humanoid.py:48-49,91-118 (gear table, ratio =
    gears / max(gears)).  DOF limits are stand-ins (USD asset not in tree)."""
    gears = torch.tensor(HUMANOID_GEARS, dtype=torch.float32)
    g = torch.Generator().manual_seed(21)
    half = 0.6 + 1.2 * torch.rand(21, generator=g)
    mid = 0.5 * (torch.rand(21, generator=g) - 0.5)
    return LocomotionSpec(
        num_dof=21, num_sensors=2, limit_cost_kind="humanoid", joint_gears=gears,
        motor_effort_ratio=gears / torch.max(gears), dof_limits_lower=(mid - half), dof_limits_upper=(mid + half),
        episode_length=1000, dof_vel_scale=0.1, angular_velocity_scale=0.25, contact_force_scale=0.01,
        power_scale=1.0, heading_weight=0.5, up_weight=0.1, actions_cost_scale=0.01,
        energy_cost_scale=0.05, joints_at_limit_cost_scale=0.25, death_cost=-1.0,
        termination_height=0.8, alive_reward_scale=2.0,
    )
