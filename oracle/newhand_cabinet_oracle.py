"""ORACLE (test infrastructure only) -- the three five-finger cabinet tasks on CPU torch:
TomoDJnewhandCabinet (tasks/tomodj_newhand_cabinet.py, 26 DOFs, 26 actions), TomoDJnewhandTendonCabinet
(tasks/tomodj_newhandTendon_cabinet.py, 26 DOFs of which 22 actuated) and TomoSPCnewhandCabinet
(tasks/tomospc_newhand_cabinet.py, 25 DOFs).

Restates, op for op (line numbers of tomodj_newhand_cabinet.py):
  * get_observations      five drawer keypoints and five fingertip keypoints through ``tf_combine(rot, pos, identity, offset)[1]``;
                          obs = scaled dof_pos (D), dof_vel * 0.1 (D), sum over the five keypoint differences (3), drawer joint
                          position and velocity
  * compute_tomodj_reward keypoint distances, hand/drawer axis alignment (TomoSPC: zeroed below 0.6, doubled from 0.8),
                          around-handle test on the x offset of all five fingertips, action penalty, three drawer bonuses;
                          Tendon: -100 where ``terminate_buf == 1`` -- the flags of the PREVIOUS step's is_done
  * is_done               sticky; Tendon adds ``terminate_buf`` (any scaled joint position outside +-1.05)
  * pre_physics_step / reset_idx   as FrankaCabinet; Tendon integrates only the actuated joints' targets
``tf_combine / tf_vector / tensor_clamp`` are restated omni helpers (parity unpinned); the rest is pinned bit for bit to the
three reference files' own method source by tests/test_oracle_pinned_newhand_cabinet.py (container only).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import torch_utils as tu

Tensor = torch.Tensor


def _offsets(width: float, z0: float, z1: float) -> List[List[float]]:
    """drawer_local_keypoint_offsets as post_reset builds them (torch.linspace in fp32)."""
    o = torch.zeros((5, 3))
    o[1:, 1] = (torch.linspace(0.0, 1.0, 4) - 0.5) * width - 0.01
    o[0, 1] = o[1, 1]
    for i in range(5):
        o[i, 0] = 0.305
        o[i, 2] = z0 if i == 0 else z1
    return o.tolist()


@dataclass
class NewhandSpec:
    num_dofs: int = 26
    num_actions: int = 26
    actuated_dof_indices: Optional[List[int]] = None   # None: all joints, in order
    max_episode_length: int = 500
    action_scale: float = 7.5
    dof_vel_scale: float = 0.1
    dist_reward_scale: float = 10.0
    rot_reward_scale: float = 3.0
    around_handle_reward_scale: float = 2.0
    open_reward_scale: float = 12.5
    action_penalty_scale: float = 0.01
    dt: float = 1 / 60.0
    num_props: int = 4
    rot_reward_thresholds: bool = False
    terminate_on_limits: bool = False
    around_lower: float = 0.2732
    around_upper: float = 0.3149
    drawer_local_keypoint_offsets: List[List[float]] = field(default_factory=lambda: _offsets(0.1, 0.0165, 0.0135))
    local_thumb_pose_pos: List[float] = field(default_factory=lambda: [0.038, 0.008, 0.0])
    local_another_finger_pose_pos: List[float] = field(default_factory=lambda: [0.024, -0.002, 0.0])
    left_hand_forward_axis: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.0])
    drawer_inward_axis: List[float] = field(default_factory=lambda: [-1.0, 0.0, 0.0])
    left_hand_up_axis: List[float] = field(default_factory=lambda: [0.0, 0.0, 1.0])
    drawer_up_axis: List[float] = field(default_factory=lambda: [0.0, 0.0, 1.0])
    default_dof_pos: List[float] = field(default_factory=lambda: [0.52, 0.52, 0.26, -0.26, -0.25, 0.17, -0.52] + [0.0] * 19)


def tomodj_newhand_spec(**kw) -> NewhandSpec:
    return NewhandSpec(**kw)


def tomodj_newhand_tendon_spec(actuated: List[int], **kw) -> NewhandSpec:
    return NewhandSpec(**{**dict(num_actions=len(actuated), actuated_dof_indices=list(actuated), terminate_on_limits=True), **kw})


@dataclass
class NewhandBuffers:
    obs_buf: Tensor
    rew_buf: Tensor
    reset_buf: Tensor
    progress_buf: Tensor
    actions: Tensor             # (N,A)
    dof_targets: Tensor         # (N,D)
    keypoints_finger: Tensor    # (N,5,3)
    keypoints_drawer: Tensor    # (N,5,3)
    terminate_buf: Tensor       # (N,) int64 (Tendon)
    dof_pos_scaled: Optional[Tensor] = None

    def clone(self) -> "NewhandBuffers":
        return NewhandBuffers(**{k: (v.clone() if isinstance(v, Tensor) else v) for k, v in self.__dict__.items()})


def _rep(v, n) -> Tensor:
    return torch.tensor(v, dtype=torch.float32).repeat((n, 1))


def observations(spec: NewhandSpec, b: NewhandBuffers, st, lower: Tensor, upper: Tensor) -> None:
    """``st``: hand_rot, finger_pos[5], finger_rot[5] (thumb first), drawer_pos/rot, dof_pos/vel (N,D), cabinet_dof_pos/vel (N,4)."""
    n = st["drawer_pos"].shape[0]
    ident = _rep([1.0, 0.0, 0.0, 0.0], n)
    for idx, off in enumerate(spec.drawer_local_keypoint_offsets):
        b.keypoints_drawer[:, idx] = tu.tf_combine(st["drawer_rot"], st["drawer_pos"], ident, torch.tensor(off).repeat(n, 1))[1]
    b.keypoints_finger[:, 0] = tu.tf_combine(st["finger_rot"][0], st["finger_pos"][0], ident, _rep(spec.local_thumb_pose_pos, n))[1]
    for i in range(4):
        b.keypoints_finger[:, i + 1] = tu.tf_combine(st["finger_rot"][i + 1], st["finger_pos"][i + 1], ident,
                                                     _rep(spec.local_another_finger_pose_pos, n))[1]
    b.dof_pos_scaled = 2.0 * (st["dof_pos"] - lower) / (upper - lower) - 1.0
    to_target = torch.sum(b.keypoints_drawer - b.keypoints_finger, dim=1)
    b.obs_buf = torch.cat((b.dof_pos_scaled, st["dof_vel"] * spec.dof_vel_scale, to_target, st["cabinet_dof_pos"][:, 3].unsqueeze(-1),
                           st["cabinet_dof_vel"][:, 3].unsqueeze(-1)), dim=-1)


def metrics(spec: NewhandSpec, b: NewhandBuffers, st) -> None:
    n = b.rew_buf.shape[0]
    cab3 = st["cabinet_dof_pos"][:, 3]
    d_keypoints = torch.norm(b.keypoints_drawer - b.keypoints_finger, p=2, dim=-1)
    d = torch.sum(d_keypoints[:, 1:], dim=-1)
    dist_reward = -(d + d_keypoints[:, 0]) / 2
    axis1 = tu.tf_vector(st["hand_rot"], _rep(spec.left_hand_forward_axis, n))
    axis2 = tu.tf_vector(st["drawer_rot"], _rep(spec.drawer_inward_axis, n))
    axis3 = tu.tf_vector(st["hand_rot"], _rep(spec.left_hand_up_axis, n))
    axis4 = tu.tf_vector(st["drawer_rot"], _rep(spec.drawer_up_axis, n))
    dot1 = torch.bmm(axis1.view(n, 1, 3), axis2.view(n, 3, 1)).squeeze(-1).squeeze(-1)
    dot2 = torch.bmm(axis3.view(n, 1, 3), axis4.view(n, 3, 1)).squeeze(-1).squeeze(-1)
    rot_reward = 0.5 * (torch.sign(dot1) * dot1**2 + torch.sign(dot2) * dot2**2)
    if spec.rot_reward_thresholds:
        rot_reward = torch.where(rot_reward[:] < 0.6, 0, rot_reward)
        rot_reward = torch.where(rot_reward[:] >= 0.8, 2 * rot_reward, rot_reward)
    around = torch.zeros_like(dist_reward)
    inside = torch.ones_like(dist_reward, dtype=torch.bool)
    for i in range(5):
        dx = b.keypoints_finger[:, i, 0] - st["drawer_pos"][:, 0]
        inside = inside & (dx > spec.around_lower) & (dx < spec.around_upper)
    around = torch.where(inside, around + 0.5, around)
    action_penalty = torch.sum(b.actions**2, dim=-1)
    open_reward = cab3 * around + cab3
    rewards = (spec.dist_reward_scale * dist_reward + spec.rot_reward_scale * rot_reward + spec.around_handle_reward_scale * around
               + spec.open_reward_scale * open_reward - spec.action_penalty_scale * action_penalty)
    rewards = torch.where(cab3 > 0.01, rewards + 0.5, rewards)
    rewards = torch.where(cab3 > 0.1, rewards + around, rewards)
    rewards = torch.where(cab3 > 0.19, rewards + (2.0 * around), rewards)
    if spec.terminate_on_limits:
        rewards = torch.where(b.terminate_buf == 1, -100, rewards)
    b.rew_buf[:] = rewards


def is_done(spec: NewhandSpec, b: NewhandBuffers, st) -> None:
    b.reset_buf = torch.where(st["cabinet_dof_pos"][:, 3] > 0.19, torch.ones_like(b.reset_buf), b.reset_buf)
    b.reset_buf = torch.where(b.progress_buf >= spec.max_episode_length - 1, torch.ones_like(b.reset_buf), b.reset_buf)
    if spec.terminate_on_limits:
        b.terminate_buf = torch.zeros_like(b.reset_buf)
        b.terminate_buf = torch.where(torch.any(b.dof_pos_scaled < -1.05, dim=1), torch.ones_like(b.terminate_buf), b.terminate_buf)
        b.terminate_buf = torch.where(torch.any(b.dof_pos_scaled > 1.05, dim=1), torch.ones_like(b.terminate_buf), b.terminate_buf)
        b.reset_buf = torch.where(b.terminate_buf == 1, torch.ones_like(b.reset_buf), b.reset_buf)


def post_physics(spec: NewhandSpec, b: NewhandBuffers, st, lower, upper) -> None:
    b.progress_buf[:] += 1  # rl_task.py:289
    observations(spec, b, st, lower, upper)
    metrics(spec, b, st)
    is_done(spec, b, st)


def reset_idx(spec: NewhandSpec, b: NewhandBuffers, env_ids: Tensor, lower, upper, u: Tensor, default_prop_pos=None, default_prop_rot=None):
    n, D, P = len(env_ids), spec.num_dofs, spec.num_props
    pos = tu.tensor_clamp(torch.tensor(spec.default_dof_pos, dtype=torch.float32).unsqueeze(0) + 0.25 * (u - 0.5), lower, upper)
    b.dof_targets[env_ids, :] = pos
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    rows = dict(dof_pos=pos.clone(), dof_vel=torch.zeros(n, D), cabinet_dof_pos=torch.zeros(n, 4), cabinet_dof_vel=torch.zeros(n, 4))
    if P > 0:
        prop_ids = (env_ids.unsqueeze(-1) * P + torch.arange(P)).flatten()
        rows.update(prop_ids=prop_ids, prop_pos=default_prop_pos[prop_ids], prop_rot=default_prop_rot[prop_ids])
    return rows


def pre_physics(spec: NewhandSpec, b: NewhandBuffers, actions: Tensor, lower, upper, speed_scales, draw, **props):
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(spec, b, reset_env_ids, lower, upper, draw(reset_env_ids), **props)
    b.actions = actions.clone()
    if spec.actuated_dof_indices is None:
        targets = b.dof_targets + speed_scales * spec.dt * b.actions * spec.action_scale
        b.dof_targets[:] = tu.tensor_clamp(targets, lower, upper)
    else:
        act = torch.tensor(spec.actuated_dof_indices)
        targets = b.dof_targets[:, act] + speed_scales[act] * spec.dt * b.actions * spec.action_scale
        b.dof_targets[:, act] = tu.tensor_clamp(targets, lower[act], upper[act])
    return reset_env_ids, rows


def tomospc_newhand_spec(**kw) -> NewhandSpec:
    return NewhandSpec(**{**dict(num_dofs=25, num_actions=25, rot_reward_thresholds=True, around_upper=0.3154,
                                 drawer_local_keypoint_offsets=_offsets(0.075, 0.015, 0.005),
                                 left_hand_forward_axis=[0.0, 1.0, 0.0], left_hand_up_axis=[1.0, 0.0, 0.0],
                                 default_dof_pos=[-0.52, 0.26, 0.79, 0.35, 1.05, 0.0] + [0.0] * 19), **kw})
