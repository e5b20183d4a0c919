"""ORACLE (test infrastructure only) -- BallBalance task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/ball_balance.py, op for op:
  * get_observations   :159-185  (24 columns: 3 actuated DOF pos / vel, ball pos / linvel, 4 x 3 force-sensor channels / 20)
  * calculate_metrics  :286-299
  * is_done            :301-308  (STICKY: starts from reset_buf; timeout, ball below 1.5 radii)
  * pre_physics_step   :187-206  (reset_idx of the flagged envs, integrate + clamp the position targets, zero those of the
                                  reset envs)
  * reset_idx          :208-260  (draw order: dists, dirs, hspeeds, vspeeds (lower == upper == 5: still drawn), height; (n,1) each)
Pinning: tests/test_oracle_pinned_ball_balance.py runs the reference's method source on a stub ``self`` (container only);
``torch_random_dir_2`` underneath is a restated omni helper (UNPINNED, oracle/torch_utils.py).
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import List, Optional

import torch

from . import torch_utils as tu

Tensor = torch.Tensor


@dataclass
class BallBalanceSpec:
    """cfg/task/BallBalance.yaml:7-13; ball_balance.py:58-71."""

    max_episode_length: int = 600
    dt: float = 0.01
    action_speed_scale: float = 20
    ball_radius: float = 0.1
    num_dof: int = 6  # three upper_leg + three lower_leg joints (USD asset; stand-in order below)
    actuated_dof_indices: List[int] = field(default_factory=lambda: [1, 3, 5])
    lower: Optional[Tensor] = None
    upper: Optional[Tensor] = None

    def __post_init__(self):
        if self.lower is None:
            self.lower = -1.2 * torch.ones(self.num_dof)
            self.upper = 1.2 * torch.ones(self.num_dof)


@dataclass
class BallBalanceBuffers:
    obs_buf: Tensor               # (N,24)
    rew_buf: Tensor
    reset_buf: Tensor             # int64
    progress_buf: Tensor          # int64
    dof_position_targets: Tensor  # (N,num_dof)

    def clone(self) -> "BallBalanceBuffers":
        return BallBalanceBuffers(**{k: v.clone() for k, v in self.__dict__.items()})


def observations(spec: BallBalanceSpec, b: BallBalanceBuffers, ball_pos, ball_vel, dof_pos, dof_vel, sensors, env_pos):
    """Returns (ball_positions, ball_linvels), which the metrics read (:182-183)."""
    act = torch.tensor(spec.actuated_dof_indices, dtype=torch.long)
    ball_positions = ball_pos[:, 0:3] - env_pos
    ball_linvels = ball_vel[:, 0:3]
    b.obs_buf[..., 0:3] = dof_pos[..., act]
    b.obs_buf[..., 3:6] = dof_vel[..., act]
    b.obs_buf[..., 6:9] = ball_positions
    b.obs_buf[..., 9:12] = ball_linvels
    b.obs_buf[..., 12:15] = sensors[..., 0] / 20.0
    b.obs_buf[..., 15:18] = sensors[..., 3] / 20.0
    b.obs_buf[..., 18:21] = sensors[..., 4] / 20.0
    b.obs_buf[..., 21:24] = sensors[..., 5] / 20.0
    return ball_positions, ball_linvels


def metrics(b: BallBalanceBuffers, ball_positions, ball_linvels) -> None:
    ball_dist = torch.sqrt(
        ball_positions[..., 0] * ball_positions[..., 0]
        + (ball_positions[..., 2] - 0.7) * (ball_positions[..., 2] - 0.7)
        + (ball_positions[..., 1]) * ball_positions[..., 1]
    )
    ball_speed = torch.sqrt(
        ball_linvels[..., 0] * ball_linvels[..., 0] + ball_linvels[..., 1] * ball_linvels[..., 1]
        + ball_linvels[..., 2] * ball_linvels[..., 2]
    )
    pos_reward = 1.0 / (1.0 + ball_dist)
    speed_reward = 1.0 / (1.0 + ball_speed)
    b.rew_buf[:] = pos_reward * speed_reward


def is_done(spec: BallBalanceSpec, b: BallBalanceBuffers, ball_positions) -> None:
    reset = torch.where(b.progress_buf >= spec.max_episode_length - 1, torch.ones_like(b.reset_buf), b.reset_buf)
    reset = torch.where(ball_positions[..., 2] < spec.ball_radius * 1.5, torch.ones_like(b.reset_buf), reset)
    b.reset_buf[:] = reset


def post_physics(spec, b, ball_pos, ball_vel, dof_pos, dof_vel, sensors, env_pos) -> None:
    b.progress_buf[:] += 1  # rl_task.py:289
    bp, bl = observations(spec, b, ball_pos, ball_vel, dof_pos, dof_vel, sensors, env_pos)
    metrics(b, bp, bl)
    is_done(spec, b, bp)


@dataclass
class BallBalanceInit:
    """post_reset constants (:266-272)."""

    ball_pos: Tensor   # (N,3)
    ball_rot: Tensor   # (N,4)
    ball_vel: Tensor   # (N,6)
    bot_pos: Tensor    # (N,3)
    bot_rot: Tensor    # (N,4)
    bot_vel: Tensor    # (N,6)
    dof_pos: Tensor    # (N,num_dof)


def reset_idx(b: BallBalanceBuffers, init: BallBalanceInit, env_ids: Tensor, u: Tensor):
    """:208-260 with the five (n,1) U[0,1) draws supplied as the columns of ``u`` (n,5): dists, dirs, hspeeds, vspeeds, height."""
    U = lambda lo, hi, x: (hi - lo) * x + lo  # torch_rand_float
    min_d, max_d, min_height, max_height = 0.001, 0.4, 1.0, 2.0
    dists = U(min_d, max_d, u[:, 0:1])
    dirs = tu.random_dir_2_from_unit(u[:, 1:2])
    hpos = dists * dirs
    speedscales = (dists - min_d) / (max_d - min_d)
    hspeeds = U(0, 2, u[:, 2:3])
    hvels = -speedscales * hspeeds * dirs
    vspeeds = -U(5.0, 5.0, u[:, 3:4]).squeeze()
    ball_pos, ball_rot = init.ball_pos.clone(), init.ball_rot.clone()
    ball_pos[env_ids, 0:2] += hpos[..., 0:2]
    ball_pos[env_ids, 2] += U(min_height, max_height, u[:, 4:5]).squeeze()
    ball_rot[env_ids, 0] = 1
    ball_rot[env_ids, 1:] = 0
    ball_velocities = init.ball_vel.clone()
    ball_velocities[env_ids, 0:2] = hvels[..., 0:2]
    ball_velocities[env_ids, 2] = vspeeds
    ball_velocities[env_ids, 3:6] = 0
    b.reset_buf[env_ids] = 0
    b.progress_buf[env_ids] = 0
    return dict(ball_pos=ball_pos[env_ids], ball_rot=ball_rot[env_ids], ball_vel=ball_velocities[env_ids],
                bot_pos=init.bot_pos[env_ids].clone(), bot_rot=init.bot_rot[env_ids].clone(), bot_vel=init.bot_vel[env_ids].clone(),
                dof_pos=init.dof_pos[env_ids].clone())


def pre_physics(spec: BallBalanceSpec, b: BallBalanceBuffers, init: BallBalanceInit, actions: Tensor, draw):
    """:187-206.  ``draw(ids) -> (n,5)`` U[0,1).  Returns (env_ids, rows | None)."""
    reset_env_ids = b.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    rows = None
    if len(reset_env_ids) > 0:
        rows = reset_idx(b, init, reset_env_ids, draw(reset_env_ids))
    act = torch.tensor(spec.actuated_dof_indices, dtype=torch.long)
    b.dof_position_targets[..., act] += spec.dt * spec.action_speed_scale * actions
    b.dof_position_targets[:] = tu.tensor_clamp(b.dof_position_targets, spec.lower, spec.upper)
    b.dof_position_targets[reset_env_ids] = 0
    return reset_env_ids, rows
