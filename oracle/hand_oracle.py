"""ORACLE (test infrastructure only) -- ShadowHand / in-hand manipulation task math on CPU torch.

Restates, op for op:
  * jit compute_hand_reward   /root/reference/omniisaacgymenvs/tasks/shared/in_hand_manipulation.py:457-531
  * jit randomize_rotation    in_hand_manipulation.py:450-453
  * pre_physics_step          in_hand_manipulation.py:314-372 (absolute + relative control)
  * reset_target_pose         in_hand_manipulation.py:377-395
  * reset_idx                 in_hand_manipulation.py:397-441
  * get_object_goal_observations :256-261 and the ShadowHand packers tasks/shadow_hand.py:120-262
    (openai 42, full 157, full_no_vel 77 incl. its overlapping-slice quirk at :190-191, and the
    asymmetric full_state states_buf 187; the non-asymmetric full_state obs :263-306 is dead code
    in the reference -- undefined dof_force_tensor -- and is not restated).
  * the AllegroHand packers tasks/allegro_hand.py:101-144 (``hand="allegro"``: 16 DOFs, all
    actuated, no fingertips, no states; full 72 / full_no_vel 50).
Inputs are const: the reference's in-place ``object_pos -= env_pos`` on view buffers is restated
as a subtraction into a temporary.

Pinning: compute_hand_reward / randomize_rotation are executed from the reference's source on
the restated omni helpers (helper layer UNPINNED, composition pinned); the packers and
reset/pre_physics arithmetic are executed from the reference's method source on a stub ``self``
(tests/test_oracle_pinned_hand.py, container only), ShadowHand and AllegroHand alike.
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np
import torch

from . import torch_utils as tu

Tensor = torch.Tensor

# sorted DOF indices of the 20 actuated joints among the 24 hand DOFs.  The real list comes from the
# USD joint order (views/shadow_hand_view.py:59-84: looked up by name, then sorted); the four
# un-actuated coupled distal joints (FFJ0, MFJ0, RFJ0, LFJ0) are taken to be DOFs 5, 9, 13, 18.
ACTUATED_DOF_INDICES = [i for i in range(24) if i not in (5, 9, 13, 18)]


@dataclass
class HandSpec:
    """cfg/task/ShadowHand.yaml:7-49 defaults; ShadowHandOpenAI_FF.yaml:36-51 via openai_ff()."""

    hand: str = "shadow"  # "shadow" (tasks/shadow_hand.py) | "allegro" (tasks/allegro_hand.py)
    num_hand_dofs: int = 24
    num_actions: int = 20
    num_fingertips: int = 5
    obs_type: str = "full"
    asymmetric_obs: bool = False
    dist_reward_scale: float = -10.0
    rot_reward_scale: float = 1.0
    rot_eps: float = 0.1
    action_penalty_scale: float = -0.0002
    success_tolerance: float = 0.1
    reach_goal_bonus: float = 250.0
    fall_dist: float = 0.24
    fall_penalty: float = 0.0
    vel_obs_scale: float = 0.2
    force_torque_obs_scale: float = 10.0  # shadow_hand.py:84
    reset_position_noise: float = 0.01
    reset_dof_pos_noise: float = 0.2
    reset_dof_vel_noise: float = 0.0
    hand_dof_speed_scale: float = 20.0
    use_relative_control: bool = False
    act_moving_average: float = 1.0
    max_episode_length: int = 600
    max_consecutive_successes: int = 0
    av_factor: float = 0.1
    dt: float = 1.0 / 60  # in_hand_manipulation.py:92
    goal_displacement: tuple = (-0.2, -0.06, 0.12)  # :111
    actuated_dof_indices: List[int] = field(default_factory=lambda: list(ACTUATED_DOF_INDICES))
    lower: Optional[Tensor] = None
    upper: Optional[Tensor] = None

    def __post_init__(self):
        if self.lower is None:
            nd = self.num_hand_dofs
            g = torch.Generator().manual_seed(nd)
            self.lower = -(0.2 + 0.6 * torch.rand(nd, generator=g))
            self.upper = 0.3 + 1.3 * torch.rand(nd, generator=g)

    @property
    def num_obs(self) -> int:
        if self.hand == "allegro":
            return {"full_no_vel": 50, "full": 72}[self.obs_type]  # allegro_hand.py:62-65
        return {"openai": 42, "full_no_vel": 77, "full": 157, "full_state": 187}[self.obs_type]

    @property
    def num_states(self) -> int:
        return 187 if self.asymmetric_obs else 0


def openai_ff_spec() -> HandSpec:
    """ShadowHandOpenAI_FF.yaml: obs=openai, asymmetric states, mcs=50, resetTime 8 ->
    max_episode_length = round(8 / (3 * 1/60)) = 160 (in_hand_manipulation.py:92-97)."""
    return HandSpec(obs_type="openai", asymmetric_obs=True, fall_penalty=-50.0, success_tolerance=0.4,
                    max_consecutive_successes=50, av_factor=0.1, act_moving_average=0.3,
                    max_episode_length=int(round(8.0 / (3 * (1.0 / 60)))))


def allegro_spec(obs_type: str = "full") -> HandSpec:
    """cfg/task/AllegroHand.yaml:7-42 (same reward scalars as ShadowHand.yaml); 16 DOFs, all actuated
    (views/allegro_hand_view.py:53), no fingertip bodies, no states (allegro_hand.py:69-71)."""
    return HandSpec(hand="allegro", num_hand_dofs=16, num_actions=16, num_fingertips=0, obs_type=obs_type,
                    actuated_dof_indices=list(range(16)))


def randomize_rotation(rand0, rand1, x_unit_tensor, y_unit_tensor):
    return tu.quat_mul(
        tu.quat_from_angle_axis(rand0 * np.pi, x_unit_tensor), tu.quat_from_angle_axis(rand1 * np.pi, y_unit_tensor)
    )


def compute_hand_reward(spec: HandSpec, reset_buf, reset_goal_buf, progress_buf, successes, consecutive_successes,
                        object_pos, object_rot, target_pos, target_rot, actions):
    """in_hand_manipulation.py:482-531.  ``av_factor`` reaches the jit function as a 0-dim fp32
    tensor converted to float (:59, :293), hence the fp32 round trip below."""
    max_episode_length = float(spec.max_episode_length)
    av_factor = float(torch.tensor(spec.av_factor, dtype=torch.float))
    goal_dist = torch.norm(object_pos - target_pos, p=2, dim=-1)
    quat_diff = tu.quat_mul(object_rot, tu.quat_conjugate(target_rot))
    rot_dist = 2.0 * torch.asin(torch.clamp(torch.norm(quat_diff[:, 1:4], p=2, dim=-1), max=1.0))
    dist_rew = goal_dist * spec.dist_reward_scale
    rot_rew = 1.0 / (torch.abs(rot_dist) + spec.rot_eps) * spec.rot_reward_scale
    action_penalty = torch.sum(actions**2, dim=-1)
    reward = dist_rew + rot_rew + action_penalty * spec.action_penalty_scale
    goal_resets = torch.where(torch.abs(rot_dist) <= spec.success_tolerance, torch.ones_like(reset_goal_buf), reset_goal_buf)
    successes = successes + goal_resets
    reward = torch.where(goal_resets == 1, reward + spec.reach_goal_bonus, reward)
    reward = torch.where(goal_dist >= spec.fall_dist, reward + spec.fall_penalty, reward)
    resets = torch.where(goal_dist >= spec.fall_dist, torch.ones_like(reset_buf), reset_buf)
    if spec.max_consecutive_successes > 0:
        progress_buf = torch.where(torch.abs(rot_dist) <= spec.success_tolerance, torch.zeros_like(progress_buf), progress_buf)
        resets = torch.where(successes >= spec.max_consecutive_successes, torch.ones_like(resets), resets)
    resets = torch.where(progress_buf >= max_episode_length - 1, torch.ones_like(resets), resets)
    if spec.max_consecutive_successes > 0:
        reward = torch.where(progress_buf >= max_episode_length - 1, reward + 0.5 * spec.fall_penalty, reward)
    num_resets = torch.sum(resets)
    finished_cons_successes = torch.sum(successes * resets.float())
    cons_successes = torch.where(
        num_resets > 0,
        av_factor * finished_cons_successes / num_resets + (1.0 - av_factor) * consecutive_successes,
        consecutive_successes,
    )
    return reward, resets, goal_resets, progress_buf, successes, cons_successes


@dataclass
class HandState:
    """What the views hand over each step (shadow_hand.py:121-135), world frame, const."""

    object_pos: Tensor          # (N,3)
    object_rot: Tensor          # (N,4)
    object_velocities: Tensor   # (N,6)
    fingertip_pos: Optional[Tensor]       # (5N,3); None for AllegroHand
    fingertip_rot: Optional[Tensor]       # (5N,4)
    fingertip_velocities: Optional[Tensor]  # (5N,6)
    hand_dof_pos: Tensor        # (N,24|16)
    hand_dof_vel: Tensor        # (N,24|16)
    vec_sensor: Optional[Tensor]          # (N,5,6)
    env_pos: Tensor             # (N,3)


@dataclass
class HandBuffers:
    obs_buf: Tensor
    states_buf: Tensor
    rew_buf: Tensor
    reset_buf: Tensor
    reset_goal_buf: Tensor
    progress_buf: Tensor
    successes: Tensor
    consecutive_successes: Tensor
    goal_pos: Tensor
    goal_rot: Tensor
    actions: Tensor
    prev_targets: Tensor
    cur_targets: Tensor

    @staticmethod
    def fresh(spec: HandSpec, n: int) -> "HandBuffers":
        z = lambda *s: torch.zeros(*s, dtype=torch.float32)
        rot = z(n, 4)
        rot[:, 0] = 1
        return HandBuffers(z(n, spec.num_obs), z(n, spec.num_states), z(n), torch.ones(n, dtype=torch.long),
                           torch.ones(n, dtype=torch.long), torch.zeros(n, dtype=torch.long), z(n), z(1), z(n, 3), rot,
                           z(n, spec.num_actions), z(n, spec.num_hand_dofs), z(n, spec.num_hand_dofs))

    def clone(self) -> "HandBuffers":
        return HandBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def observations(spec: HandSpec, s: HandState, b: HandBuffers) -> None:
    """shadow_hand.py:120-152 writing b.obs_buf / b.states_buf in place (other columns untouched)."""
    n, nf, nd = s.object_pos.shape[0], spec.num_fingertips, spec.num_hand_dofs
    object_pos = s.object_pos - s.env_pos
    object_rot = s.object_rot
    object_linvel, object_angvel = s.object_velocities[:, 0:3], s.object_velocities[:, 3:6]
    rel = tu.quat_mul(object_rot, tu.quat_conjugate(b.goal_rot))
    o = b.obs_buf
    if spec.hand == "allegro":  # allegro_hand.py:101-144
        o[:, 0:nd] = tu.unscale(s.hand_dof_pos, spec.lower, spec.upper)
        if spec.obs_type == "full_no_vel":  # :118-128
            o[:, 16:19] = object_pos
            o[:, 19:23] = object_rot
            o[:, 23:26] = b.goal_pos
            o[:, 26:30] = b.goal_rot
            o[:, 30:34] = rel
            o[:, 34:50] = b.actions
        elif spec.obs_type == "full":  # :130-144
            o[:, nd:2 * nd] = spec.vel_obs_scale * s.hand_dof_vel
            o[:, 32:35] = object_pos
            o[:, 35:39] = object_rot
            o[:, 39:42] = object_linvel
            o[:, 42:45] = spec.vel_obs_scale * object_angvel
            o[:, 45:48] = b.goal_pos
            o[:, 48:52] = b.goal_rot
            o[:, 52:56] = rel
            o[:, 56:72] = b.actions
        else:
            raise RuntimeError("AllegroHand observationType must be full or full_no_vel (allegro_hand.py:59)")
        return
    fingertip_pos = s.fingertip_pos - s.env_pos.repeat((1, nf)).reshape(n * nf, 3)
    if spec.obs_type == "openai":  # :162-165
        o[:, 0:15] = fingertip_pos.reshape(n, 15)
        o[:, 15:18] = object_pos
        o[:, 18:22] = rel
        o[:, 22:42] = b.actions
    elif spec.obs_type == "full_no_vel":  # :186-196 (24:37 is a 13-wide slice fed 3 columns: broadcast error)
        raise RuntimeError("full_no_vel: the reference raises a shape error at shadow_hand.py:190")
    elif spec.obs_type == "full":  # :198-217
        o[:, 0:nd] = tu.unscale(s.hand_dof_pos, spec.lower, spec.upper)
        o[:, nd:2 * nd] = spec.vel_obs_scale * s.hand_dof_vel
        o[:, 48:51] = object_pos
        o[:, 51:55] = object_rot
        o[:, 55:58] = object_linvel
        o[:, 58:61] = spec.vel_obs_scale * object_angvel
        o[:, 61:64] = b.goal_pos
        o[:, 64:68] = b.goal_rot
        o[:, 68:72] = rel
        o[:, 72:87] = fingertip_pos.reshape(n, 3 * nf)
        o[:, 87:107] = s.fingertip_rot.reshape(n, 4 * nf)
        o[:, 107:137] = s.fingertip_velocities.reshape(n, 6 * nf)
        o[:, 137:157] = b.actions
    else:
        raise RuntimeError("full_state obs (non-asymmetric) is dead code in the reference (shadow_hand.py:263-306)")
    if spec.asymmetric_obs:  # compute_full_state(True), :220-262
        t = b.states_buf
        t[:, 0:nd] = tu.unscale(s.hand_dof_pos, spec.lower, spec.upper)
        t[:, nd:2 * nd] = spec.vel_obs_scale * s.hand_dof_vel
        t[:, 48:51] = object_pos
        t[:, 51:55] = object_rot
        t[:, 55:58] = object_linvel
        t[:, 58:61] = spec.vel_obs_scale * object_angvel
        t[:, 61:64] = b.goal_pos
        t[:, 64:68] = b.goal_rot
        t[:, 68:72] = rel
        t[:, 72:87] = fingertip_pos.reshape(n, 3 * nf)
        t[:, 87:107] = s.fingertip_rot.reshape(n, 4 * nf)
        t[:, 107:137] = s.fingertip_velocities.reshape(n, 6 * nf)
        t[:, 137:167] = spec.force_torque_obs_scale * s.vec_sensor.reshape(n, -1)
        t[:, 167:187] = b.actions


def post_physics(spec: HandSpec, s: HandState, b: HandBuffers) -> None:
    """rl_task.py:289-295 order: progress += 1, observations, metrics (is_done is a no-op, :374)."""
    b.progress_buf[:] += 1
    observations(spec, s, b)
    object_pos = s.object_pos - s.env_pos
    (b.rew_buf[:], b.reset_buf[:], b.reset_goal_buf[:], b.progress_buf[:], b.successes[:],
     b.consecutive_successes[:]) = compute_hand_reward(
        spec, b.reset_buf, b.reset_goal_buf, b.progress_buf, b.successes, b.consecutive_successes,
        object_pos, s.object_rot, b.goal_pos, b.goal_rot, b.actions)


@dataclass
class HandInit:
    """post_reset constants (in_hand_manipulation.py:233-247)."""

    object_init_pos: Tensor  # (N,3) env-local
    goal_init_pos: Tensor    # (N,3)
    env_pos: Tensor          # (N,3)
    default_pos: Tensor      # (24,)
    default_vel: Tensor      # (24,)


def reset_target_pose(spec: HandSpec, b: HandBuffers, init: HandInit, env_ids: Tensor, rand_floats: Tensor):
    """:377-395 with the (n,4) U(-1,1) draw supplied.  Returns (goal world pos, goal rot) rows."""
    n_all = b.goal_pos.shape[0]
    x_unit = torch.tensor([1, 0, 0], dtype=torch.float).repeat((n_all, 1))
    y_unit = torch.tensor([0, 1, 0], dtype=torch.float).repeat((n_all, 1))
    new_rot = randomize_rotation(rand_floats[:, 0], rand_floats[:, 1], x_unit[env_ids], y_unit[env_ids])
    b.goal_pos[env_ids] = init.goal_init_pos[env_ids, 0:3]
    b.goal_rot[env_ids] = new_rot
    goal_pos, goal_rot = b.goal_pos.clone(), b.goal_rot.clone()
    goal_pos[env_ids] = b.goal_pos[env_ids] + torch.tensor(spec.goal_displacement) + init.env_pos[env_ids]
    b.reset_goal_buf[env_ids] = 0
    return goal_pos[env_ids], goal_rot[env_ids]


def reset_idx(spec: HandSpec, b: HandBuffers, init: HandInit, env_ids: Tensor, rand_floats: Tensor, goal_rand: Tensor):
    """:397-441.  ``rand_floats`` (n, 2*24+5) U(-1,1) is drawn BEFORE the nested reset_target_pose
    draws ``goal_rand`` (n,4).  Returns the rows handed to the setters."""
    nd = spec.num_hand_dofs
    n_all = b.goal_pos.shape[0]
    goal_world = reset_target_pose(spec, b, init, env_ids, goal_rand)
    x_unit = torch.tensor([1, 0, 0], dtype=torch.float).repeat((n_all, 1))
    y_unit = torch.tensor([0, 1, 0], dtype=torch.float).repeat((n_all, 1))
    new_object_pos = init.object_init_pos[env_ids] + spec.reset_position_noise * rand_floats[:, 0:3] + init.env_pos[env_ids]
    new_object_rot = randomize_rotation(rand_floats[:, 3], rand_floats[:, 4], x_unit[env_ids], y_unit[env_ids])
    delta_max = spec.upper - init.default_pos
    delta_min = spec.lower - init.default_pos
    rand_delta = delta_min + (delta_max - delta_min) * 0.5 * (rand_floats[:, 5:5 + nd] + 1.0)
    pos = init.default_pos + spec.reset_dof_pos_noise * rand_delta
    dof_vel = init.default_vel + spec.reset_dof_vel_noise * rand_floats[:, 5 + nd:5 + nd * 2]
    b.prev_targets[env_ids, :nd] = pos
    b.cur_targets[env_ids, :nd] = pos
    b.progress_buf[env_ids] = 0
    b.reset_buf[env_ids] = 0
    b.successes[env_ids] = 0
    return dict(goal_world_pos=goal_world[0], goal_rot=goal_world[1], object_pos=new_object_pos, object_rot=new_object_rot,
                object_vel=torch.zeros(len(env_ids), 6), dof_pos=pos, dof_vel=dof_vel)


def apply_actions(spec: HandSpec, b: HandBuffers, actions: Tensor) -> Tensor:
    """:331-362.  Returns the (N,20) targets passed to set_joint_position_targets."""
    act = torch.tensor(spec.actuated_dof_indices, dtype=torch.long)
    b.actions = actions.clone()
    lo, hi = spec.lower[act], spec.upper[act]
    if spec.use_relative_control:
        targets = b.prev_targets[:, act] + spec.hand_dof_speed_scale * spec.dt * b.actions
        b.cur_targets[:, act] = tu.tensor_clamp(targets, lo, hi)
    else:
        b.cur_targets[:, act] = tu.scale(b.actions, lo, hi)
        b.cur_targets[:, act] = (spec.act_moving_average * b.cur_targets[:, act]
                                 + (1.0 - spec.act_moving_average) * b.prev_targets[:, act])
        b.cur_targets[:, act] = tu.tensor_clamp(b.cur_targets[:, act], lo, hi)
    b.prev_targets[:, act] = b.cur_targets[:, act]
    return b.cur_targets[:, act]


def pre_physics(spec: HandSpec, b: HandBuffers, init: HandInit, actions: Tensor, draw_goal=None, draw_reset=None):
    """:314-362.  ``draw_goal(ids) -> (n,4)`` feeds reset_target_pose(goal_env_ids);
    ``draw_reset(ids) -> ((n,53), (n,4))`` feeds reset_idx (its own draw, then the nested goal draw).
    Returns (targets (N,20), goal_ids, goal_rows | None, env_ids, reset_rows | None)."""
    env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    goal_env_ids = b.reset_goal_buf.nonzero(as_tuple=False).squeeze(-1)
    goal_rows = reset_rows = None
    if len(goal_env_ids) > 0:
        goal_rows = reset_target_pose(spec, b, init, goal_env_ids, draw_goal(goal_env_ids))
    if len(env_ids) > 0:
        rf, gr = draw_reset(env_ids)
        reset_rows = reset_idx(spec, b, init, env_ids, rf, gr)
    targets = apply_actions(spec, b, actions)
    return targets, goal_env_ids, goal_rows, env_ids, reset_rows
