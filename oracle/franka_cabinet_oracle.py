"""ORACLE (test infrastructure only) -- FrankaCabinet / TomoDJCabinet / TomoSPCCabinet task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/franka_cabinet.py, op for op:
  * get_observations        :283-358  (grasp transforms :475-494; 23 columns: scaled dof_pos 9, dof_vel * 0.1 9, to_target 3,
                                       drawer joint position and velocity = cabinet DOF 3)
  * compute_franka_reward   :496-601  (8 terms + three drawer-opening bonuses)
  * is_done                 :468-473  (sticky: ``where(cond, 1, reset_buf)`` twice)
  * pre_physics_step        :360-373, reset_idx :375-414
The reference reads BOTH finger poses from ``_lfingers`` (:336-337), so ``around_handle_reward`` and ``finger_dist_reward``
are identically zero there; this restatement takes the two positions separately and the caller passes what the reference
passes.  tasks/tomodj_cabinet.py:373-712 (9 DOFs, finger joints 7, 8) and tasks/tomospc_cabinet.py (8 DOFs, finger joints
6, 7) share the skeleton with ``reward_variant="tomo"`` (dist_reward = -d, rot_reward zeroed below 0.6 and doubled from 0.8,
finger-height tests <= / >=, bonus thresholds 0.1 / 0.19, done at 0.19): ``tomodj_spec()`` / ``tomospc_spec()``.
``tf_combine / tf_vector / tensor_clamp`` are restated omni helpers (parity unpinned); everything else is pinned
bit for bit to the reference's own method source by tests/test_oracle_pinned_franka_cabinet.py (container only).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List

import torch

from . import torch_utils as tu

Tensor = torch.Tensor
FRANKA_DEFAULT_DOF_POS = [1.157, -1.066, -0.155, -2.239, -1.841, 1.003, 0.469, 0.035, 0.035]  # :277-279


@dataclass
class FrankaCabinetSpec:
    """cfg/task/FrankaCabinet.yaml:8-32; franka_cabinet.py:52-53."""

    max_episode_length: int = 500
    action_scale: float = 7.5
    dof_vel_scale: float = 0.1
    dist_reward_scale: float = 2.0
    rot_reward_scale: float = 0.5
    around_handle_reward_scale: float = 10.0
    open_reward_scale: float = 7.5
    finger_dist_reward_scale: float = 100.0
    action_penalty_scale: float = 0.01
    finger_close_reward_scale: float = 10.0
    dt: float = 1 / 60.0
    num_props: int = 4
    num_dofs: int = 9
    reward_variant: str = "franka"     # "franka" | "tomo"
    open_bonus_threshold: float = 0.2  # franka_cabinet.py:592 / tomodj_cabinet.py:675
    open_done_threshold: float = 0.39  # :593, :470 / tomodj_cabinet.py:676, :569
    default_dof_pos: List[float] = field(default_factory=lambda: list(FRANKA_DEFAULT_DOF_POS))
    franka_local_grasp_pos: List[float] = field(default_factory=lambda: [0.0, 0.04, 0.1034])   # stand-in for the USD-derived pose
    franka_local_grasp_rot: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.0, 0.0])
    drawer_local_grasp_pos: List[float] = field(default_factory=lambda: [0.3, 0.01, 0.0])      # :260
    drawer_local_grasp_rot: List[float] = field(default_factory=lambda: [1.0, 0.0, 0.0, 0.0])
    gripper_forward_axis: List[float] = field(default_factory=lambda: [0.0, 0.0, 1.0])         # :264-275
    drawer_inward_axis: List[float] = field(default_factory=lambda: [-1.0, 0.0, 0.0])
    gripper_up_axis: List[float] = field(default_factory=lambda: [0.0, 1.0, 0.0])
    drawer_up_axis: List[float] = field(default_factory=lambda: [0.0, 0.0, 1.0])


def tomodj_spec(**kw) -> FrankaCabinetSpec:
    """cfg/task/TomoDJCabinet.yaml; tomodj_cabinet.py:112-114."""
    return FrankaCabinetSpec(**{**dict(dist_reward_scale=5.0, rot_reward_scale=1.5, around_handle_reward_scale=2.0, open_reward_scale=15.0,
                                       finger_dist_reward_scale=15.0, finger_close_reward_scale=10.0, reward_variant="tomo",
                                       open_bonus_threshold=0.1, open_done_threshold=0.19,
                                       default_dof_pos=[0.52, 0.52, 0.26, -0.26, -0.25, 0.17, -0.52, 0.035, 0.035]), **kw})


def tomospc_spec(**kw) -> FrankaCabinetSpec:
    """cfg/task/TomoSPCCabinet.yaml; tomospc_cabinet.py:292-294."""
    return FrankaCabinetSpec(**{**dict(dist_reward_scale=5.0, rot_reward_scale=1.25, around_handle_reward_scale=2.0, open_reward_scale=20.0,
                                       finger_dist_reward_scale=13.0, finger_close_reward_scale=9.0, reward_variant="tomo",
                                       open_bonus_threshold=0.1, open_done_threshold=0.19, num_dofs=8,
                                       default_dof_pos=[-0.52, 0.26, 0.79, 0.35, 1.05, 0, 0.035, 0.035]), **kw})


@dataclass
class FrankaCabinetBuffers:
    obs_buf: Tensor             # (N,23)
    rew_buf: Tensor
    reset_buf: Tensor           # int64
    progress_buf: Tensor        # int64
    actions: Tensor             # (N,9)
    franka_dof_targets: Tensor  # (N,9)
    franka_grasp_rot: Tensor    # (N,4) world
    franka_grasp_pos: Tensor    # (N,3)
    drawer_grasp_rot: Tensor
    drawer_grasp_pos: Tensor

    def clone(self) -> "FrankaCabinetBuffers":
        return FrankaCabinetBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def _rep(v, n) -> Tensor:
    return torch.tensor(v, dtype=torch.float32).repeat((n, 1))


def observations(spec: FrankaCabinetSpec, b: FrankaCabinetBuffers, st, lower: Tensor, upper: Tensor) -> None:
    """:283-358.  ``st``: hand_pos/rot, drawer_pos/rot, franka_dof_pos/vel (N,9), cabinet_dof_pos/vel (N,4)."""
    n = st["hand_pos"].shape[0]
    b.franka_grasp_rot, b.franka_grasp_pos = tu.tf_combine(st["hand_rot"], st["hand_pos"], _rep(spec.franka_local_grasp_rot, n),
                                                           _rep(spec.franka_local_grasp_pos, n))
    b.drawer_grasp_rot, b.drawer_grasp_pos = tu.tf_combine(st["drawer_rot"], st["drawer_pos"], _rep(spec.drawer_local_grasp_rot, n),
                                                           _rep(spec.drawer_local_grasp_pos, n))
    dof_pos_scaled = 2.0 * (st["franka_dof_pos"] - lower) / (upper - lower) - 1.0
    to_target = b.drawer_grasp_pos - b.franka_grasp_pos
    b.obs_buf = torch.cat((dof_pos_scaled, st["franka_dof_vel"] * spec.dof_vel_scale, to_target,
                           st["cabinet_dof_pos"][:, 3].unsqueeze(-1), st["cabinet_dof_vel"][:, 3].unsqueeze(-1)), dim=-1)


def metrics(spec: FrankaCabinetSpec, b: FrankaCabinetBuffers, st) -> None:
    """:496-601 with the argument binding of :439-466."""
    n = b.rew_buf.shape[0]
    cab3 = st["cabinet_dof_pos"][:, 3]
    lf, rf, dg = st["lfinger_pos"], st["rfinger_pos"], b.drawer_grasp_pos
    tomo = spec.reward_variant == "tomo"
    D = spec.num_dofs
    d = torch.norm(b.franka_grasp_pos - dg, p=2, dim=-1)
    if tomo:
        dist_reward = -d
    else:
        dist_reward = 1.0 / (1.0 + d**2)
        dist_reward *= dist_reward
        dist_reward = torch.where(d <= 0.02, dist_reward * 2, dist_reward)
    axis1 = tu.tf_vector(b.franka_grasp_rot, _rep(spec.gripper_forward_axis, n))
    axis2 = tu.tf_vector(b.drawer_grasp_rot, _rep(spec.drawer_inward_axis, n))
    axis3 = tu.tf_vector(b.franka_grasp_rot, _rep(spec.gripper_up_axis, n))
    axis4 = tu.tf_vector(b.drawer_grasp_rot, _rep(spec.drawer_up_axis, n))
    dot1 = torch.bmm(axis1.view(n, 1, 3), axis2.view(n, 3, 1)).squeeze(-1).squeeze(-1)
    dot2 = torch.bmm(axis3.view(n, 1, 3), axis4.view(n, 3, 1)).squeeze(-1).squeeze(-1)
    rot_reward = 0.5 * (torch.sign(dot1) * dot1**2 + torch.sign(dot2) * dot2**2)
    if tomo:
        rot_reward = torch.where(rot_reward[:] < 0.6, 0, rot_reward)
        rot_reward = torch.where(rot_reward[:] >= 0.8, 2 * rot_reward, rot_reward)
    zero = torch.zeros_like(rot_reward)
    c1 = (lf[:, 2] <= dg[:, 2]) if tomo else (lf[:, 2] > dg[:, 2])
    c2 = (rf[:, 2] >= dg[:, 2]) if tomo else (rf[:, 2] < dg[:, 2])
    around = torch.where(c1, torch.where(c2, zero + 0.5, zero), zero)
    lfinger_dist = torch.abs(lf[:, 2] - dg[:, 2])
    rfinger_dist = torch.abs(rf[:, 2] - dg[:, 2])
    finger_dist_reward = torch.where(c1, torch.where(c2, (0.04 - lfinger_dist) + (0.04 - rfinger_dist), zero), zero)
    jp = st["franka_dof_pos"]
    finger_close_reward = torch.where(d <= 0.03, (0.04 - jp[:, D - 2]) + (0.04 - jp[:, D - 1]), zero)
    action_penalty = torch.sum(b.actions**2, dim=-1)
    open_reward = cab3 * around + cab3
    rewards = (spec.dist_reward_scale * dist_reward + spec.rot_reward_scale * rot_reward + spec.around_handle_reward_scale * around
               + spec.open_reward_scale * open_reward + spec.finger_dist_reward_scale * finger_dist_reward
               - spec.action_penalty_scale * action_penalty + finger_close_reward * spec.finger_close_reward_scale)
    rewards = torch.where(cab3 > 0.01, rewards + 0.5, rewards)
    rewards = torch.where(cab3 > spec.open_bonus_threshold, rewards + around, rewards)
    rewards = torch.where(cab3 > spec.open_done_threshold, rewards + (2.0 * around), rewards)
    b.rew_buf[:] = rewards


def is_done(spec: FrankaCabinetSpec, b: FrankaCabinetBuffers, st) -> None:
    b.reset_buf = torch.where(st["cabinet_dof_pos"][:, 3] > spec.open_done_threshold, torch.ones_like(b.reset_buf), b.reset_buf)
    b.reset_buf = torch.where(b.progress_buf >= spec.max_episode_length - 1, torch.ones_like(b.reset_buf), b.reset_buf)


def post_physics(spec: FrankaCabinetSpec, b: FrankaCabinetBuffers, st, lower, upper) -> None:
    b.progress_buf[:] += 1  # rl_task.py:289
    observations(spec, b, st, lower, upper)
    metrics(spec, b, st)
    is_done(spec, b, st)


def reset_idx(b: FrankaCabinetBuffers, env_ids: Tensor, lower, upper, u: Tensor, num_props: int = 0, default_prop_pos=None,
              default_prop_rot=None, default_dof_pos=FRANKA_DEFAULT_DOF_POS):
    """:375-414 with the (n,D) U[0,1) draw supplied.  Returns the rows handed to the view setters."""
    n, D = len(env_ids), len(default_dof_pos)
    pos = tu.tensor_clamp(torch.tensor(default_dof_pos, dtype=torch.float32).unsqueeze(0) + 0.25 * (u - 0.5), lower, upper)
    b.franka_dof_targets[env_ids, :] = pos
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    rows = dict(dof_pos=pos.clone(), dof_vel=torch.zeros(n, D), dof_targets=pos.clone(), cabinet_dof_pos=torch.zeros(n, 4),
                cabinet_dof_vel=torch.zeros(n, 4))
    if num_props > 0:
        prop_ids = (env_ids.unsqueeze(-1) * num_props + torch.arange(num_props)).flatten()
        rows.update(prop_ids=prop_ids, prop_pos=default_prop_pos[prop_ids], prop_rot=default_prop_rot[prop_ids])
    return rows


def pre_physics(spec: FrankaCabinetSpec, b: FrankaCabinetBuffers, actions: Tensor, lower, upper, speed_scales, draw, **props):
    """:360-373.  ``draw(ids) -> (n,9)`` U[0,1).  Returns (reset ids, rows | None)."""
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(b, reset_env_ids, lower, upper, draw(reset_env_ids), default_dof_pos=spec.default_dof_pos, **props)
    b.actions = actions.clone()
    targets = b.franka_dof_targets + speed_scales * spec.dt * b.actions * spec.action_scale
    b.franka_dof_targets[:] = tu.tensor_clamp(targets, lower, upper)
    return reset_env_ids, rows
