"""Synthetic state tensors for the BASELINE.json configs (recipes: SURVEY.md section 8d).

They stand in for what PhysX would hand the task through the view getters.  Generated with a
seeded ``torch.Generator`` on the requested device; the same recipe on CPU feeds the oracle.
"""
from __future__ import annotations

from typing import Dict

import torch

Tensor = torch.Tensor


def _gen(device, seed: int) -> torch.Generator:
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    return g


def locomotion_state(n: int, lo: Tensor, hi: Tensor, num_sensors: int, device="cpu", seed: int = 1234,
                     z_range=(0.2, 0.9)) -> Dict[str, Tensor]:
    """C2 (Ant: z in U(0.2,0.9), ~15 % below 0.31) / C3 (Humanoid: z in U(0.5,1.5), term 0.8)."""
    g = _gen(device, seed)
    d = lo.shape[0]
    lo, hi = lo.to(device), hi.to(device)
    f = dict(device=device, dtype=torch.float32, generator=g)
    pos = torch.stack(
        [
            torch.rand(n, **f) * 100.0 - 50.0,
            torch.rand(n, **f) * 100.0 - 50.0,
            torch.rand(n, **f) * (z_range[1] - z_range[0]) + z_range[0],
        ],
        dim=-1,
    ).contiguous()
    rot = torch.randn(n, 4, **f)
    rot = (rot / rot.norm(dim=-1, keepdim=True)).contiguous()
    vel = torch.randn(n, 6, **f)
    dof_pos = (lo - 0.05) + torch.rand(n, d, **f) * ((hi - lo) + 0.1)
    dof_vel = torch.randn(n, d, **f) * 3.0
    sensors = torch.randn(n, num_sensors, 6, **f) * 20.0
    actions = torch.rand(n, d, **f) * 2.0 - 1.0
    to_target = torch.tensor([1000.0, 0.0, 0.0], device=device) - pos
    potentials = -(to_target[:, :2].norm(dim=-1)) * 60.0 + torch.randn(n, **f)
    progress = torch.randint(0, 1000, (n,), device=device, generator=g, dtype=torch.int64)
    reset = torch.zeros(n, dtype=torch.int64, device=device)
    return dict(pos=pos, rot=rot, vel=vel, dof_pos=dof_pos, dof_vel=dof_vel, sensors=sensors, actions=actions,
                potentials=potentials, progress=progress, reset=reset)


def cartpole_state(n: int, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """C1: cart in U(-3.5,3.5), pole in U(-2,2) (~20 % trip each limit), vel ~ N(0,1), progress < 510."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    dof_pos = torch.stack([torch.rand(n, **f) * 7.0 - 3.5, torch.rand(n, **f) * 4.0 - 2.0], dim=-1).contiguous()
    dof_vel = torch.randn(n, 2, **f)
    progress = torch.randint(0, 510, (n,), device=device, generator=g, dtype=torch.int64)
    actions = torch.rand(n, 1, **f) * 2.0 - 1.0
    return dict(dof_pos=dof_pos, dof_vel=dof_vel, progress=progress, actions=actions)


def load_locomotion_state(task, st: Dict[str, Tensor]) -> None:
    """Copy a synthetic state into the task's view buffers and RL buffers (device of the task)."""
    v = task.get_robot()
    dev = task.device
    v.positions.copy_(st["pos"].to(dev))
    v.orientations.copy_(st["rot"].to(dev))
    v.velocities.copy_(st["vel"].to(dev))
    v.joint_positions.copy_(st["dof_pos"].to(dev))
    v.joint_velocities.copy_(st["dof_vel"].to(dev))
    v.measured_joint_forces.copy_(st["sensors"].to(dev))
    task.actions.copy_(st["actions"].to(dev))
    task.potentials.copy_(st["potentials"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))
