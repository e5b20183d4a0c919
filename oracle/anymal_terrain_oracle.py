"""ORACLE (test infrastructure only) -- AnymalTerrain per-step task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/anymal_terrain.py:
  pre_physics_step PD torques :391-407 (one sub-step), post_physics_step :409-444 in its order
  (progress, frame quantities :421-427, check_termination :452-466, calculate_metrics :468-533,
  the in-step reset_idx :318-367 with update_terrain_level :369-380, get_observations +
  get_heights :535-604, uniform obs noise :438-439, last_* copies :441-442), jit quat_apply_yaw
  :608-612 and wrap_to_pi :615-619, the scan grid init_height_points :173-187 and the noise
  scale vector :157-171.

Reference quirks kept on purpose: reset_buf becomes a BOOL tensor (:465) and
reset_idx leaves it 1 (:358); the curriculum compares against the norm of the WHOLE (n,2) command
matrix of the reset envs (:376); ``~timeout_buf`` is a bitwise NOT on int64 (:521); reset rows of
the view buffers (dof_pos, dof_vel, base pose) are overwritten before get_observations reads them,
while base_lin_vel / base_ang_vel / projected_gravity keep their pre-reset values; last_dof_vel ends
the step holding the NEW velocities of reset envs (:355 is overwritten by :442).
The view buffers are treated as const here: the mutated copies are local.

Pinning: wrap_to_pi, check_termination and calculate_metrics run from the reference's source on a
stub ``self`` (pure torch -> pinned); quat_apply_yaw / the frame quantities need the restated omni
helpers (UNPINNED layer).  tests/test_oracle_pinned_anymal.py, container only.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Dict, Optional

import numpy as np
import torch

from . import torch_utils as tu

Tensor = torch.Tensor

EPISODE_KEYS = ["lin_vel_xy", "lin_vel_z", "ang_vel_z", "ang_vel_xy", "orient", "torques", "joint_acc", "base_height",
                "air_time", "collision", "stumble", "action_rate", "hip"]  # anymal_terrain.py:68-82 (dict order)

# Anymal.dof_names order is taken to be the YAML's listing order (cfg/task/AnymalTerrain.yaml:48-62);
# the real order comes from the USD articulation (anymal_terrain.py:258-262).
DEFAULT_JOINT_ANGLES = [0.03, 0.03, -0.03, -0.03, 0.4, -0.4, 0.4, -0.4, -0.8, 0.8, -0.8, 0.8]


@dataclass
class AnymalSpec:
    """cfg/task/AnymalTerrain.yaml:5-103 through update_config (:85-155)."""

    lin_vel_scale: float = 2.0
    ang_vel_scale: float = 0.25
    dof_pos_scale: float = 1.0
    dof_vel_scale: float = 0.05
    height_meas_scale: float = 5.0
    action_scale: float = 0.5
    Kp: float = 80.0
    Kd: float = 2.0
    decimation: int = 4
    sim_dt: float = 0.005
    max_episode_length_s: float = 20.0
    push_interval_s: float = 15.0
    curriculum: bool = True
    command_x_range: tuple = (-1.0, 1.0)
    command_y_range: tuple = (-1.0, 1.0)
    command_yaw_range: tuple = (-3.14, 3.14)
    base_init_state: tuple = (0.0, 0.0, 0.62, 1.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
    add_noise: bool = True
    noise_level: float = 1.0
    dof_pos_noise: float = 0.01
    dof_vel_noise: float = 1.5
    lin_vel_noise: float = 0.1
    ang_vel_noise: float = 0.2
    gravity_noise: float = 0.05
    height_noise: float = 0.06
    # terrain (tasks/utils/anymal_terrain_generator.py:38-60)
    horizontal_scale: float = 0.1
    vertical_scale: float = 0.005
    border_size: float = 20
    env_length: float = 8.0
    env_rows: int = 10
    env_cols: int = 20
    raw_rew_scales: Dict[str, float] = field(default_factory=lambda: dict(
        termination=0.0, lin_vel_xy=1.0, lin_vel_z=-4.0, ang_vel_z=0.5, ang_vel_xy=-0.05, orient=-0.0, torque=-0.00002,
        joint_acc=-0.0005, base_height=-0.0, action_rate=-0.01, hip=-0.0, fallen_over=-1.0))
    default_joint_angles: tuple = tuple(DEFAULT_JOINT_ANGLES)

    @property
    def dt(self) -> float:
        return self.decimation * self.sim_dt  # :130

    @property
    def max_episode_length(self) -> int:
        return int(self.max_episode_length_s / self.dt + 0.5)  # :132

    @property
    def rew_scales(self) -> Dict[str, float]:
        out = {}
        for k, v in self.raw_rew_scales.items():  # :140-141  (``*= self.dt``)
            v *= self.dt
            out[k] = v
        return out


def noise_scale_vec(spec: AnymalSpec) -> Tensor:
    """_get_noise_scale_vec :157-171."""
    v = torch.zeros(188)
    nl = spec.noise_level
    v[:3] = spec.lin_vel_noise * nl * spec.lin_vel_scale
    v[3:6] = spec.ang_vel_noise * nl * spec.ang_vel_scale
    v[6:9] = spec.gravity_noise * nl
    v[9:12] = 0.0
    v[12:24] = spec.dof_pos_noise * nl * spec.dof_pos_scale
    v[24:36] = spec.dof_vel_noise * nl * spec.dof_vel_scale
    v[36:176] = spec.height_noise * nl * spec.height_meas_scale
    v[176:188] = 0.0
    return v


def init_height_points(n: int) -> Tensor:
    """:173-187 -- 14 x 10 grid, x-major."""
    y = 0.1 * torch.tensor([-5, -4, -3, -2, -1, 1, 2, 3, 4, 5])
    x = 0.1 * torch.tensor([-8, -7, -6, -5, -4, -3, -2, 2, 3, 4, 5, 6, 7, 8])
    grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
    points = torch.zeros(n, grid_x.numel(), 3)
    points[:, :, 0] = grid_x.flatten()
    points[:, :, 1] = grid_y.flatten()
    return points


def quat_apply_yaw(quat: Tensor, vec: Tensor) -> Tensor:
    quat_yaw = quat.clone().view(-1, 4)
    quat_yaw[:, 1:3] = 0.0
    quat_yaw = tu.normalize(quat_yaw)
    return tu.quat_apply(quat_yaw, vec)


def wrap_to_pi(angles: Tensor) -> Tensor:
    """:615-619 AS TORCHSCRIPT RUNS IT: the in-place ``angles %= 2*np.pi`` of a jit-scripted function
    compiles to ``aten::fmod_`` (C fmod, sign of the dividend), not Python's floor-mod -- found by
    executing the reference source (``wrap_to_pi.code``).  Negative angles therefore stay negative:
    wrap_to_pi(-4.0) = -4.0, not 2.2832."""
    angles = torch.fmod(angles, 2 * np.pi)
    angles = angles - 2 * np.pi * (angles > np.pi)
    return angles


def pd_torques(spec: AnymalSpec, actions, default_dof_pos, dof_pos, dof_vel) -> Tensor:
    """:398-403."""
    return torch.clip(spec.Kp * (spec.action_scale * actions + default_dof_pos - dof_pos) - spec.Kd * dof_vel, -80.0, 80.0)


def get_heights(spec: AnymalSpec, base_quat, base_pos, height_points, height_samples) -> Tensor:
    """:582-604 (env_ids=None branch)."""
    n, npts = base_quat.shape[0], height_points.shape[1]
    points = quat_apply_yaw(base_quat.repeat(1, npts), height_points) + (base_pos[:, 0:3]).unsqueeze(1)
    points += spec.border_size
    points = (points / spec.horizontal_scale).long()
    px = points[:, :, 0].view(-1)
    py = points[:, :, 1].view(-1)
    px = torch.clip(px, 0, height_samples.shape[0] - 2)
    py = torch.clip(py, 0, height_samples.shape[1] - 2)
    heights1 = height_samples[px, py]
    heights2 = height_samples[px + 1, py + 1]
    heights = torch.min(heights1, heights2)
    return heights.view(n, -1) * spec.vertical_scale


@dataclass
class AnymalState:
    """View buffers at post_physics time (refresh_*_state_tensors :382-389 + contact forces)."""

    base_pos: Tensor         # (N,3)
    base_quat: Tensor        # (N,4)
    base_velocities: Tensor  # (N,6)
    dof_pos: Tensor          # (N,12)
    dof_vel: Tensor          # (N,12)
    knee_contact: Tensor     # (4N,3)
    base_contact: Tensor     # (N,3)


@dataclass
class AnymalBuffers:
    """Persistent task state."""

    obs_buf: Tensor
    rew_buf: Tensor
    reset_buf: Tensor
    timeout_buf: Tensor
    progress_buf: Tensor
    commands: Tensor
    actions: Tensor
    last_actions: Tensor
    last_dof_vel: Tensor
    torques: Tensor
    feet_air_time: Tensor
    episode_sums: Dict[str, Tensor]
    terrain_levels: Tensor
    terrain_types: Tensor
    env_origins: Tensor
    default_dof_pos: Tensor
    extras: Dict[str, object] = field(default_factory=dict)

    def clone(self) -> "AnymalBuffers":
        d = {}
        for k, v in self.__dict__.items():
            if isinstance(v, dict):
                d[k] = {kk: (vv.clone() if isinstance(vv, torch.Tensor) else vv) for kk, vv in v.items()}
            else:
                d[k] = v.clone()
        return AnymalBuffers(**d)


def check_termination(spec: AnymalSpec, b: AnymalBuffers, s: AnymalState):
    n = s.base_pos.shape[0]
    b.timeout_buf = torch.where(b.progress_buf >= spec.max_episode_length - 1, torch.ones_like(b.timeout_buf),
                                torch.zeros_like(b.timeout_buf))
    knee_contact = torch.norm(s.knee_contact.view(n, 4, 3), dim=-1) > 1.0
    has_fallen = (torch.norm(s.base_contact, dim=1) > 1.0) | (torch.sum(knee_contact, dim=-1) > 1.0)
    b.reset_buf = has_fallen.clone()
    b.reset_buf = torch.where(b.timeout_buf.bool(), torch.ones_like(b.reset_buf), b.reset_buf)
    return has_fallen


def calculate_metrics(spec: AnymalSpec, b: AnymalBuffers, base_lin_vel, base_ang_vel, projected_gravity, base_pos,
                      dof_pos, dof_vel, has_fallen) -> None:
    rs = spec.rew_scales
    lin_vel_error = torch.sum(torch.square(b.commands[:, :2] - base_lin_vel[:, :2]), dim=1)
    ang_vel_error = torch.square(b.commands[:, 2] - base_ang_vel[:, 2])
    rew_lin_vel_xy = torch.exp(-lin_vel_error / 0.25) * rs["lin_vel_xy"]
    rew_ang_vel_z = torch.exp(-ang_vel_error / 0.25) * rs["ang_vel_z"]
    rew_lin_vel_z = torch.square(base_lin_vel[:, 2]) * rs["lin_vel_z"]
    rew_ang_vel_xy = torch.sum(torch.square(base_ang_vel[:, :2]), dim=1) * rs["ang_vel_xy"]
    rew_orient = torch.sum(torch.square(projected_gravity[:, :2]), dim=1) * rs["orient"]
    rew_base_height = torch.square(base_pos[:, 2] - 0.52) * rs["base_height"]
    rew_torque = torch.sum(torch.square(b.torques), dim=1) * rs["torque"]
    rew_joint_acc = torch.sum(torch.square(b.last_dof_vel - dof_vel), dim=1) * rs["joint_acc"]
    rew_fallen_over = has_fallen * rs["fallen_over"]
    rew_action_rate = torch.sum(torch.square(b.last_actions - b.actions), dim=1) * rs["action_rate"]
    rew_hip = torch.sum(torch.abs(dof_pos[:, 0:4] - b.default_dof_pos[:, 0:4]), dim=1) * rs["hip"]
    b.rew_buf = (rew_lin_vel_xy + rew_ang_vel_z + rew_lin_vel_z + rew_ang_vel_xy + rew_orient + rew_base_height
                 + rew_torque + rew_joint_acc + rew_action_rate + rew_hip + rew_fallen_over)
    b.rew_buf = torch.clip(b.rew_buf, min=0.0, max=None)
    b.rew_buf += rs["termination"] * b.reset_buf * ~b.timeout_buf
    es = b.episode_sums
    es["lin_vel_xy"] += rew_lin_vel_xy
    es["ang_vel_z"] += rew_ang_vel_z
    es["lin_vel_z"] += rew_lin_vel_z
    es["ang_vel_xy"] += rew_ang_vel_xy
    es["orient"] += rew_orient
    es["torques"] += rew_torque
    es["joint_acc"] += rew_joint_acc
    es["action_rate"] += rew_action_rate
    es["base_height"] += rew_base_height
    es["hip"] += rew_hip


@dataclass
class AnymalResetDraws:
    """The six torch_rand_float draws of reset_idx, as U[0,1) tensors, in call order (:321-349)."""

    dof_offset: Tensor  # (n,12)
    dof_vel: Tensor     # (n,12)
    xy: Tensor          # (n,2)
    cmd_x: Tensor       # (n,1)
    cmd_y: Tensor       # (n,1)
    cmd_yaw: Tensor     # (n,1)


def reset_idx(spec: AnymalSpec, b: AnymalBuffers, view: Dict[str, Tensor], env_ids: Tensor, d: AnymalResetDraws,
              terrain_origins: Tensor, init_done: bool = True) -> None:
    """:318-367 mutating ``view`` (local copies of the view buffers) and ``b``."""
    U = tu.uniform_from_unit
    n = len(env_ids)
    positions_offset = U(0.5, 1.5, d.dof_offset)
    velocities = U(-0.1, 0.1, d.dof_vel)
    view["dof_pos"][env_ids] = b.default_dof_pos[env_ids] * positions_offset
    view["dof_vel"][env_ids] = velocities
    # update_terrain_level :369-380
    if init_done and spec.curriculum:
        root_pos = view["base_pos"]
        distance = torch.norm(root_pos[env_ids, :2] - b.env_origins[env_ids, :2], dim=1)
        b.terrain_levels[env_ids] -= 1 * (distance < torch.norm(b.commands[env_ids, :2]) * spec.max_episode_length_s * 0.25)
        b.terrain_levels[env_ids] += 1 * (distance > spec.env_length / 2)
        b.terrain_levels[env_ids] = torch.clip(b.terrain_levels[env_ids], 0) % spec.env_rows
        b.env_origins[env_ids] = terrain_origins[b.terrain_levels[env_ids], b.terrain_types[env_ids]]
    init = torch.tensor(spec.base_init_state, dtype=torch.float)
    view["base_pos"][env_ids] = init[0:3]
    view["base_pos"][env_ids, 0:3] += b.env_origins[env_ids]
    view["base_pos"][env_ids, 0:2] += U(-0.5, 0.5, d.xy)
    view["base_quat"][env_ids] = init[3:7]
    view["base_velocities"][env_ids] = init[7:]
    b.commands[env_ids, 0] = U(spec.command_x_range[0], spec.command_x_range[1], d.cmd_x).squeeze()
    b.commands[env_ids, 1] = U(spec.command_y_range[0], spec.command_y_range[1], d.cmd_y).squeeze()
    b.commands[env_ids, 3] = U(spec.command_yaw_range[0], spec.command_yaw_range[1], d.cmd_yaw).squeeze()
    b.commands[env_ids] *= (torch.norm(b.commands[env_ids, :2], dim=1) > 0.25).unsqueeze(1)
    b.last_actions[env_ids] = 0.0
    b.last_dof_vel[env_ids] = 0.0
    b.feet_air_time[env_ids] = 0.0
    b.progress_buf[env_ids] = 0
    b.reset_buf[env_ids] = 1
    b.extras["episode"] = {}
    for key in b.episode_sums.keys():
        b.extras["episode"]["rew_" + key] = torch.mean(b.episode_sums[key][env_ids]) / spec.max_episode_length_s
        b.episode_sums[key][env_ids] = 0.0
    b.extras["episode"]["terrain_level"] = torch.mean(b.terrain_levels.float())


def post_physics(spec: AnymalSpec, b: AnymalBuffers, s: AnymalState, height_samples: Tensor, terrain_origins: Tensor,
                 draw=None, noise_u: Optional[Tensor] = None):
    """:409-444.  ``draw(ids) -> AnymalResetDraws``; ``noise_u`` (N,188) U[0,1) replaces torch.rand_like.
    Returns (env_ids, view) where ``view`` holds the post-reset copies of the view buffers."""
    n = s.base_pos.shape[0]
    b.progress_buf[:] += 1
    view = dict(base_pos=s.base_pos.clone(), base_quat=s.base_quat.clone(), base_velocities=s.base_velocities.clone(),
                dof_pos=s.dof_pos.clone(), dof_vel=s.dof_vel.clone())
    gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat((n, 1))
    forward_vec = torch.tensor([1.0, 0.0, 0.0]).repeat((n, 1))
    base_lin_vel = tu.quat_rotate_inverse(view["base_quat"], view["base_velocities"][:, 0:3])
    base_ang_vel = tu.quat_rotate_inverse(view["base_quat"], view["base_velocities"][:, 3:6])
    projected_gravity = tu.quat_rotate_inverse(view["base_quat"], gravity_vec)
    forward = tu.quat_apply(view["base_quat"], forward_vec)
    heading = torch.atan2(forward[:, 1], forward[:, 0])
    b.commands[:, 2] = torch.clip(0.5 * wrap_to_pi(b.commands[:, 3] - heading), -1.0, 1.0)
    has_fallen = check_termination(spec, b, s)
    calculate_metrics(spec, b, base_lin_vel, base_ang_vel, projected_gravity, view["base_pos"], view["dof_pos"],
                      view["dof_vel"], has_fallen)
    env_ids = b.reset_buf.nonzero(as_tuple=False).flatten()
    if len(env_ids) > 0:
        reset_idx(spec, b, view, env_ids, draw(env_ids), terrain_origins)
    measured = get_heights(spec, view["base_quat"], view["base_pos"], init_height_points(n), height_samples)
    heights = torch.clip(view["base_pos"][:, 2].unsqueeze(1) - 0.5 - measured, -1, 1.0) * spec.height_meas_scale
    commands_scale = torch.tensor([spec.lin_vel_scale, spec.lin_vel_scale, spec.ang_vel_scale])
    b.obs_buf = torch.cat((base_lin_vel * spec.lin_vel_scale, base_ang_vel * spec.ang_vel_scale, projected_gravity,
                           b.commands[:, :3] * commands_scale, view["dof_pos"] * spec.dof_pos_scale,
                           view["dof_vel"] * spec.dof_vel_scale, heights, b.actions), dim=-1)
    if spec.add_noise:
        u = noise_u if noise_u is not None else torch.rand_like(b.obs_buf)
        b.obs_buf += (2 * u - 1) * noise_scale_vec(spec)
    b.last_actions[:] = b.actions[:]
    b.last_dof_vel[:] = view["dof_vel"][:]
    return env_ids, view
