"""ORACLE (test infrastructure only) -- Cartpole per-step task math on CPU torch.

Restates /root/reference/omniisaacgymenvs/tasks/cartpole.py:
  get_observations :91-106, pre_physics_step :108-122, reset_idx :124-144,
  calculate_metrics :153-158, is_done :160-164 (NOT sticky, ``>= 500``), with
  RLTask.post_physics_step's order (tasks/base/rl_task.py:289-295).
Pinned bit-for-bit against the reference's own method source executed on a stub ``self``
(tests/test_oracle_pinned.py, container only) and against tests/golden/cartpole_*.npz.
"""
from __future__ import annotations

import math
from dataclasses import dataclass

import numpy as np
import torch

Tensor = torch.Tensor


@dataclass
class CartpoleSpec:
    reset_dist: float = 3.0        # cfg/task/Cartpole.yaml:10
    max_push_effort: float = 400.0  # Cartpole.yaml:11
    max_episode_length: int = 500   # cartpole.py:44
    cart_dof_idx: int = 0
    pole_dof_idx: int = 1
    num_dof: int = 2


def observations(spec: CartpoleSpec, dof_pos: Tensor, dof_vel: Tensor) -> Tensor:
    obs = torch.zeros((dof_pos.shape[0], 4), dtype=torch.float32)
    obs[:, 0] = dof_pos[:, spec.cart_dof_idx]
    obs[:, 1] = dof_vel[:, spec.cart_dof_idx]
    obs[:, 2] = dof_pos[:, spec.pole_dof_idx]
    obs[:, 3] = dof_vel[:, spec.pole_dof_idx]
    return obs


def metrics(spec: CartpoleSpec, obs: Tensor) -> Tensor:
    cart_pos, cart_vel, pole_pos, pole_vel = obs[:, 0], obs[:, 1], obs[:, 2], obs[:, 3]
    reward = 1.0 - pole_pos * pole_pos - 0.01 * torch.abs(cart_vel) - 0.005 * torch.abs(pole_vel)
    reward = torch.where(torch.abs(cart_pos) > spec.reset_dist, torch.ones_like(reward) * -2.0, reward)
    reward = torch.where(torch.abs(pole_pos) > np.pi / 2, torch.ones_like(reward) * -2.0, reward)
    return reward


def done(spec: CartpoleSpec, obs: Tensor, progress_buf: Tensor) -> Tensor:
    cart_pos, pole_pos = obs[:, 0], obs[:, 2]
    resets = torch.where(torch.abs(cart_pos) > spec.reset_dist, 1, 0)
    resets = torch.where(torch.abs(pole_pos) > math.pi / 2, 1, resets)
    resets = torch.where(progress_buf >= spec.max_episode_length, 1, resets)
    return resets


def post_physics(spec: CartpoleSpec, dof_pos, dof_vel, obs_buf, rew_buf, reset_buf, progress_buf) -> None:
    progress_buf[:] += 1
    obs_buf[:] = observations(spec, dof_pos, dof_vel)
    rew_buf[:] = metrics(spec, obs_buf)
    reset_buf[:] = done(spec, obs_buf, progress_buf)


def reset_idx(spec: CartpoleSpec, env_ids: Tensor, u: Tensor, reset_buf: Tensor, progress_buf: Tensor):
    """``u`` is (4, n): the four torch.rand(n) draws of :129,:130,:134,:135 in that order."""
    n = len(env_ids)
    dof_pos = torch.zeros((n, spec.num_dof))
    dof_pos[:, spec.cart_dof_idx] = 1.0 * (1.0 - 2.0 * u[0])
    dof_pos[:, spec.pole_dof_idx] = 0.125 * math.pi * (1.0 - 2.0 * u[1])
    dof_vel = torch.zeros((n, spec.num_dof))
    dof_vel[:, spec.cart_dof_idx] = 0.5 * (1.0 - 2.0 * u[2])
    dof_vel[:, spec.pole_dof_idx] = 0.25 * math.pi * (1.0 - 2.0 * u[3])
    reset_buf[env_ids] = 0
    progress_buf[env_ids] = 0
    return dof_pos, dof_vel


def pre_physics(spec: CartpoleSpec, actions: Tensor, reset_buf: Tensor, progress_buf: Tensor, draw=None):
    """:108-122.  ``draw(ids) -> u (4, n)``.  Returns (forces, ids, (dof_pos, dof_vel) | None)."""
    ids = reset_buf.nonzero(as_tuple=False).squeeze(-1)
    writes = None
    if len(ids) > 0:
        writes = reset_idx(spec, ids, draw(ids), reset_buf, progress_buf)
    forces = torch.zeros((actions.shape[0], spec.num_dof), dtype=torch.float32)
    forces[:, spec.cart_dof_idx] = spec.max_push_effort * actions[:, 0]
    return forces, ids, writes
