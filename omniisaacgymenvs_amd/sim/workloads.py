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


def hand_state(n: int, env_pos: Tensor, goal_pos_local: Tensor, lower: Tensor, upper: Tensor, device="cpu",
               seed: int = 1234, num_actions: int = 20, fingertips: int = 5) -> Dict[str, Tensor]:
    """C4: object near the goal (N(0,0.1) -> ~2 % beyond fallDistance 0.24), unit
    quaternions with 5 % of the objects exactly at the goal rotation (successes), fingertip state
    N(0,1), dof_pos in U(lo,hi), sensors N(0,1), actions U(-1,1), successes randint(0,50),
    progress randint(0,160).  AllegroHand: ``num_actions=16, fingertips=0`` (no fingertip / sensor arrays)."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos, goal_pos_local = env_pos.to(device), goal_pos_local.to(device)
    lower, upper = lower.to(device), upper.to(device)

    def quat(m):
        q = torch.randn(m, 4, **f)
        return (q / q.norm(dim=-1, keepdim=True)).contiguous()

    goal_rot = quat(n)
    object_rot = quat(n)
    same = torch.rand(n, **f) < 0.05
    object_rot = torch.where(same.unsqueeze(-1), goal_rot, object_rot).contiguous()
    object_pos = (goal_pos_local + torch.randn(n, 3, **f) * 0.1 + env_pos).contiguous()
    nf = fingertips
    return dict(
        object_pos=object_pos, object_rot=object_rot, object_vel=torch.randn(n, 6, **f),
        goal_rot=goal_rot,
        ft_pos=(torch.randn(n * nf, 3, **f) + env_pos.repeat_interleave(nf, dim=0)).contiguous(),
        ft_rot=quat(n * nf), ft_vel=torch.randn(n * nf, 6, **f),
        dof_pos=(lower + torch.rand(n, lower.shape[0], **f) * (upper - lower)).contiguous(),
        dof_vel=torch.randn(n, lower.shape[0], **f),
        sensors=torch.randn(n, nf, 6, **f),
        actions=torch.rand(n, num_actions, **f) * 2.0 - 1.0,
        successes=torch.randint(0, 50, (n,), device=device, generator=g).float(),
        progress=torch.randint(0, 160, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.02).long(),
        reset_goal=(torch.rand(n, **f) < 0.02).long(),
    )


def load_hand_state(task, st: Dict[str, Tensor]) -> None:
    dev = task.device
    task._objects.positions.copy_(st["object_pos"].to(dev))
    task._objects.orientations.copy_(st["object_rot"].to(dev))
    task._objects.velocities.copy_(st["object_vel"].to(dev))
    task.goal_rot.copy_(st["goal_rot"].to(dev))
    if task.num_fingertips > 0:
        fingers = task._hands._fingers
        fingers.positions.copy_(st["ft_pos"].to(dev))
        fingers.orientations.copy_(st["ft_rot"].to(dev))
        fingers.velocities.copy_(st["ft_vel"].to(dev))
        task._hands.measured_joint_forces.copy_(st["sensors"].to(dev))
    task._hands.joint_positions.copy_(st["dof_pos"].to(dev))
    task._hands.joint_velocities.copy_(st["dof_vel"].to(dev))
    task.actions.copy_(st["actions"].to(dev))
    task.successes.copy_(st["successes"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))
    task.reset_goal_buf.copy_(st["reset_goal"].to(dev))


ANYMAL_EPISODE_KEYS = 13


def anymal_state(n: int, terrain_origins: Tensor, default_dof_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """C5: robots spread over the terrain patches they are assigned to (so the curriculum's
    distance test goes both ways), z in U(0.3,0.9), contact forces |N(0,0.35)| per knee component / |N(0,0.4)| per base component
    (~4 % of knees and ~10 % of bases exceed the 1 N threshold -> ~11 % of the envs fall per step),
    commands per AnymalTerrain.yaml:33-37, progress up to the episode length."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    terrain_origins = terrain_origins.to(device)
    rows, cols = terrain_origins.shape[0], terrain_origins.shape[1]
    levels = torch.randint(0, rows, (n,), device=device, generator=g, dtype=torch.int64)
    types = torch.randint(0, cols, (n,), device=device, generator=g, dtype=torch.int64)
    env_origins = terrain_origins[levels, types].clone()
    base_pos = env_origins + torch.cat([torch.rand(n, 2, **f) * 10.0 - 5.0, torch.rand(n, 1, **f) * 0.6 + 0.3], dim=-1)
    quat = torch.randn(n, 4, **f)
    quat = (quat / quat.norm(dim=-1, keepdim=True)).contiguous()
    commands = torch.cat([torch.rand(n, 2, **f) * 2.0 - 1.0, torch.zeros(n, 1, device=device),
                          torch.rand(n, 1, **f) * 6.28 - 3.14], dim=-1).contiguous()
    return dict(
        base_pos=base_pos.contiguous(), base_quat=quat, base_vel=torch.randn(n, 6, **f),
        dof_pos=(default_dof_pos.to(device)[:1] + torch.randn(n, 12, **f) * 0.3).contiguous(),
        dof_vel=torch.randn(n, 12, **f) * 2.0,
        knee_contact=torch.randn(4 * n, 3, **f).abs() * 0.35, base_contact=torch.randn(n, 3, **f).abs() * 0.4,
        commands=commands, actions=torch.rand(n, 12, **f) * 2.0 - 1.0, last_actions=torch.rand(n, 12, **f) * 2.0 - 1.0,
        last_dof_vel=torch.randn(n, 12, **f) * 2.0, torques=torch.randn(n, 12, **f) * 30.0,
        feet_air_time=torch.rand(n, 4, **f), episode_sums=torch.rand(ANYMAL_EPISODE_KEYS, n, **f),
        terrain_levels=levels, terrain_types=types, env_origins=env_origins.contiguous(),
        progress=torch.randint(0, 1001, (n,), device=device, generator=g, dtype=torch.int64),
    )


def load_anymal_state(task, st: Dict[str, Tensor]) -> None:
    dev = task.device
    v = task._anymals
    v.positions.copy_(st["base_pos"].to(dev))
    v.orientations.copy_(st["base_quat"].to(dev))
    v.velocities.copy_(st["base_vel"].to(dev))
    v.joint_positions.copy_(st["dof_pos"].to(dev))
    v.joint_velocities.copy_(st["dof_vel"].to(dev))
    v._knees.net_contact_forces.copy_(st["knee_contact"].to(dev))
    v._base.net_contact_forces.copy_(st["base_contact"].to(dev))
    task.commands.copy_(st["commands"].to(dev))
    task.actions.copy_(st["actions"].to(dev))
    task.last_actions.copy_(st["last_actions"].to(dev))
    task.last_dof_vel.copy_(st["last_dof_vel"].to(dev))
    task.torques.copy_(st["torques"].to(dev))
    task.feet_air_time.copy_(st["feet_air_time"].to(dev))
    task._episode_sums.copy_(st["episode_sums"].to(dev))
    task.terrain_levels.copy_(st["terrain_levels"].to(dev))
    task.terrain_types.copy_(st["terrain_types"].to(dev))
    task.env_origins.copy_(st["env_origins"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))


def anymal_flat_state(n: int, default_dof_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """Anymal on flat ground: base height U(0.35, 0.75) (~40 % below the 0.51 fall threshold is too
    much for a steady state, so z is drawn from U(0.49, 0.75): ~8 % fall per step), progress up to the 3000-step episode."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    quat = torch.randn(n, 4, **f)
    quat = (quat / quat.norm(dim=-1, keepdim=True)).contiguous()
    pos = torch.cat([torch.randn(n, 2, **f) * 3.0, torch.rand(n, 1, **f) * 0.26 + 0.49], dim=-1).contiguous()
    return dict(
        pos=pos, rot=quat, vel=torch.randn(n, 6, **f),
        dof_pos=(default_dof_pos.to(device)[:1] + torch.randn(n, 12, **f) * 0.3).contiguous(), dof_vel=torch.randn(n, 12, **f) * 2.0,
        commands=(torch.rand(n, 3, **f) * 2.0 - 1.0), actions=torch.rand(n, 12, **f) * 2.0 - 1.0,
        last_actions=torch.rand(n, 12, **f) * 2.0 - 1.0, last_dof_vel=torch.randn(n, 12, **f) * 2.0,
        current_targets=(default_dof_pos.to(device)[:1] + torch.randn(n, 12, **f) * 0.2).contiguous(),
        progress=torch.randint(0, 3001, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.1).long(),
    )


def load_anymal_flat_state(task, st: Dict[str, Tensor]) -> None:
    dev, v = task.device, task._anymals
    v.positions.copy_(st["pos"].to(dev))
    v.orientations.copy_(st["rot"].to(dev))
    v.velocities.copy_(st["vel"].to(dev))
    v.joint_positions.copy_(st["dof_pos"].to(dev))
    v.joint_velocities.copy_(st["dof_vel"].to(dev))
    task.commands.copy_(st["commands"].to(dev))
    task.actions.copy_(st["actions"].to(dev))
    task.last_actions.copy_(st["last_actions"].to(dev))
    task.last_dof_vel.copy_(st["last_dof_vel"].to(dev))
    task.current_targets.copy_(st["current_targets"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def quadcopter_state(n: int, env_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """Quadcopter: root within ~1.2 m of the target at (0, 0, 1) (a few per cent beyond the 3 m / below
    the 0.3 m reset thresholds), unit quaternions near upright, DOF positions in +-0.8, progress up to the 500-step episode."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos = env_pos.to(device)
    q = torch.randn(n, 4, **f) * 0.3
    q[:, 0] += 1.0
    q = (q / q.norm(dim=-1, keepdim=True)).contiguous()
    off = torch.cat([torch.randn(n, 2, **f) * 1.2, torch.rand(n, 1, **f) * 1.9 + 0.2], -1)
    return dict(
        root_pos=(env_pos + off).contiguous(), root_rot=q, root_vel=torch.randn(n, 6, **f) * 2.0,
        dof_pos=torch.rand(n, 8, **f) * 1.6 - 0.8, actions=torch.rand(n, 12, **f) * 2.0 - 1.0,
        targets=torch.rand(n, 8, **f) * 1.6 - 0.8, thrusts=torch.rand(n, 4, **f) * 4.0 - 2.0,
        progress=torch.randint(0, 500, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.05).long(),
    )


def load_quadcopter_state(task, st: Dict[str, Tensor]) -> None:
    dev, v = task.device, task._copters
    v.positions.copy_(st["root_pos"].to(dev))
    v.orientations.copy_(st["root_rot"].to(dev))
    v.velocities.copy_(st["root_vel"].to(dev))
    v.joint_positions.copy_(st["dof_pos"].to(dev))
    task.dof_position_targets.copy_(st["targets"].to(dev))
    task.thrusts.copy_(st["thrusts"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def ball_balance_state(n: int, env_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """BallBalance: ball within ~0.3 m of the tray centre, z in (0.05, 1.25) (~8 % below the 0.15 reset
    height), 6 DOFs in +-1, force sensors N(0, 30), progress up to the 600-step episode."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos = env_pos.to(device)
    off = torch.cat([torch.randn(n, 2, **f) * 0.3, torch.rand(n, 1, **f) * 1.2 + 0.05], -1)
    return dict(
        ball_pos=(env_pos + off).contiguous(), ball_vel=torch.randn(n, 6, **f) * 1.5, dof_pos=torch.rand(n, 6, **f) * 2 - 1,
        dof_vel=torch.randn(n, 6, **f), sensors=torch.randn(n, 3, 6, **f) * 30, actions=torch.rand(n, 3, **f) * 2.0 - 1.0,
        targets=torch.rand(n, 6, **f) * 2.4 - 1.2,
        progress=torch.randint(0, 600, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.05).long(),
    )


def load_ball_balance_state(task, st: Dict[str, Tensor]) -> None:
    dev = task.device
    task._balls.positions.copy_(st["ball_pos"].to(dev))
    task._balls.velocities.copy_(st["ball_vel"].to(dev))
    task._balance_bots.joint_positions.copy_(st["dof_pos"].to(dev))
    task._balance_bots.joint_velocities.copy_(st["dof_vel"].to(dev))
    task._balance_bots.measured_joint_forces.copy_(st["sensors"].to(dev))
    task.dof_position_targets.copy_(st["targets"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def ingenuity_state(n: int, env_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """Ingenuity: targets in [-1,1]^2 x [1,2], root within ~1.5 m of its target (a few per cent below the
    0.5 m reset height), progress up to the 2000-step episode (0.2 % of the envs at a multiple of 500: new targets)."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos = env_pos.to(device)
    tp = torch.cat([torch.rand(n, 2, **f) * 2 - 1, torch.rand(n, 1, **f) + 1], -1).contiguous()
    q = torch.randn(n, 4, **f) * 0.3
    q[:, 0] += 1.0
    q = (q / q.norm(dim=-1, keepdim=True)).contiguous()
    return dict(
        root_pos=(env_pos + tp + torch.randn(n, 3, **f) * 0.7).contiguous(), root_rot=q, root_vel=torch.randn(n, 6, **f),
        targets=tp, actions=torch.rand(n, 6, **f) * 2.0 - 1.0, dof_pos=torch.rand(n, 4, **f) - 0.5, dof_vel=torch.randn(n, 4, **f),
        progress=torch.randint(0, 2000, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.05).long(),
    )


def load_ingenuity_state(task, st: Dict[str, Tensor]) -> None:
    dev, v = task.device, task._copters
    v.positions.copy_(st["root_pos"].to(dev))
    v.orientations.copy_(st["root_rot"].to(dev))
    v.velocities.copy_(st["root_vel"].to(dev))
    task.target_positions.copy_(st["targets"].to(dev))
    task.dof_pos.copy_(st["dof_pos"].to(dev))
    task.dof_vel.copy_(st["dof_vel"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def crazyflie_state(n: int, env_pos: Tensor, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """Crazyflie: target (0, 0, 2), root within ~1 m of it, near-upright quaternions (a few per cent
    upside down or out of the [0.5, 5] m band), filter states in [0, 1], progress up to the 700-step episode."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos = env_pos.to(device)
    q = torch.randn(n, 4, **f) * 0.45
    q[:, 0] += 1.0
    q = (q / q.norm(dim=-1, keepdim=True)).contiguous()
    tp = torch.zeros(n, 3, device=device)
    tp[:, 2] = 2.0
    return dict(
        root_pos=(env_pos + tp + torch.randn(n, 3, **f) * 0.8).contiguous(), root_rot=q, root_vel=torch.randn(n, 6, **f) * 0.7,
        targets=tp, actions=torch.rand(n, 4, **f) * 2.0 - 1.0, prev_actions=torch.rand(n, 4, **f) * 2.0 - 1.0,
        cmds_damp=torch.rand(n, 4, **f), rot_damp=torch.rand(n, 4, **f), episode_sums=torch.rand(8, n, **f) * 10,
        progress=torch.randint(0, 700, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.05).long(),
    )


def load_crazyflie_state(task, st: Dict[str, Tensor]) -> None:
    dev, v = task.device, task._copters
    v.positions.copy_(st["root_pos"].to(dev))
    v.orientations.copy_(st["root_rot"].to(dev))
    v.velocities.copy_(st["root_vel"].to(dev))
    task.target_positions.copy_(st["targets"].to(dev))
    task.actions.copy_(st["prev_actions"].to(dev))
    task.thrust_cmds_damp.copy_(st["cmds_damp"].to(dev))
    task.thrust_rot_damp.copy_(st["rot_damp"].to(dev))
    task._episode_sums.copy_(st["episode_sums"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def _rotate(q: Tensor, v: Tensor) -> Tensor:
    xyz = q[:, 1:]
    t = torch.linalg.cross(xyz, v) * 2
    return v + q[:, 0:1] * t + torch.linalg.cross(xyz, t)


def franka_cabinet_state(n: int, env_pos: Tensor, local_grasp_pos: Tensor, device="cpu", seed: int = 1234, num_dofs: int = 9,
                         drawer_scale: float = 0.45) -> Dict[str, Tensor]:
    """FrankaCabinet: the hand within ~0.2 m of the drawer handle, a quarter of the envs within a few
    centimetres (the d <= 0.02 / 0.03 branches), drawer joint U(0, 0.45) (13 % open beyond 0.39 -> reset), fingers around the
    handle height, progress up to the 500-step episode."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos = env_pos.to(device)
    unit = lambda: torch.nn.functional.normalize(torch.randn(n, 4, **f), dim=-1).contiguous()
    hand_rot, drawer_rot = unit(), unit()
    drawer_rot = torch.where(torch.rand(n, 1, **f) < 0.5, torch.tensor([1.0, 0.0, 0.0, 0.0], device=device), drawer_rot).contiguous()
    drawer_pos = (env_pos + torch.tensor([0.35, 0.0, 0.4], device=device) + torch.randn(n, 3, **f) * 0.05).contiguous()
    drawer_grasp = drawer_pos + _rotate(drawer_rot, torch.tensor([0.3, 0.01, 0.0], device=device).repeat(n, 1))
    spread = torch.where(torch.rand(n, 1, **f) < 0.25, torch.tensor(0.012, device=device), torch.tensor(0.12, device=device))
    franka_grasp = drawer_grasp + torch.randn(n, 3, **f) * spread
    hand_pos = (franka_grasp - _rotate(hand_rot, local_grasp_pos.to(device).reshape(1, 3).repeat(n, 1))).contiguous()
    D = num_dofs
    lo = torch.tensor([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973, 0.0, 0.0], device=device)[9 - D:]
    hi = torch.tensor([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973, 0.04, 0.04], device=device)[9 - D:]
    lf = franka_grasp + torch.randn(n, 3, **f) * 0.03
    rf = franka_grasp + torch.randn(n, 3, **f) * 0.03
    return dict(
        hand_pos=hand_pos, hand_rot=hand_rot, drawer_pos=drawer_pos, drawer_rot=drawer_rot,
        franka_dof_pos=(lo + (hi - lo) * torch.rand(n, D, **f)).contiguous(), franka_dof_vel=torch.randn(n, D, **f),
        cabinet_dof_pos=torch.rand(n, 4, **f) * drawer_scale, cabinet_dof_vel=torch.randn(n, 4, **f) * 0.3,
        lfinger_pos=lf.contiguous(), rfinger_pos=rf.contiguous(),
        actions=torch.rand(n, D, **f) * 2.0 - 1.0, prev_actions=torch.rand(n, D, **f) * 2.0 - 1.0,
        targets=(lo + (hi - lo) * torch.rand(n, D, **f)).contiguous(),
        progress=torch.randint(0, 498, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.03).long(),
    )


def load_franka_cabinet_state(task, st: Dict[str, Tensor]) -> None:
    dev, fr, cab = task.device, task._frankas, task._cabinets
    fr._hands.positions.copy_(st["hand_pos"].to(dev))
    fr._hands.orientations.copy_(st["hand_rot"].to(dev))
    cab._drawers.positions.copy_(st["drawer_pos"].to(dev))
    cab._drawers.orientations.copy_(st["drawer_rot"].to(dev))
    fr.joint_positions.copy_(st["franka_dof_pos"].to(dev))
    fr.joint_velocities.copy_(st["franka_dof_vel"].to(dev))
    cab.joint_positions.copy_(st["cabinet_dof_pos"].to(dev))
    cab.joint_velocities.copy_(st["cabinet_dof_vel"].to(dev))
    fr._lfingers.positions.copy_(st["lfinger_pos"].to(dev))
    fr._rfingers.positions.copy_(st["rfinger_pos"].to(dev))
    task.actions.copy_(st["prev_actions"].to(dev))
    task.franka_dof_targets.copy_(st["targets"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))


def newhand_cabinet_state(n: int, env_pos: Tensor, lower: Tensor, upper: Tensor, num_actions: int, device="cpu", seed: int = 1234) -> Dict[str, Tensor]:
    """Five-finger cabinet tasks: fingertips around the drawer handle (half of the envs within millimetres,
    so that the around-handle test on all five x offsets fires), joint positions up to 3 % beyond their limits (Tendon's
    terminate_buf), drawer joint U(0, 0.22) (done beyond 0.19)."""
    g = _gen(device, seed)
    f = dict(device=device, dtype=torch.float32, generator=g)
    env_pos, lo, hi = env_pos.to(device), lower.to(device), upper.to(device)
    D = lo.shape[0]
    ident = torch.tensor([1.0, 0.0, 0.0, 0.0], device=device)
    unit = lambda: torch.nn.functional.normalize(torch.randn(n, 4, **f), dim=-1)
    drawer_pos = (env_pos + torch.tensor([0.35, 0.0, 0.4], device=device) + torch.randn(n, 3, **f) * 0.05).contiguous()
    drawer_rot = torch.where(torch.rand(n, 1, **f) < 0.6, ident, unit()).contiguous()
    near = drawer_pos + torch.tensor([0.29, 0.0, 0.01], device=device)
    spread = torch.where(torch.rand(n, 1, **f) < 0.5, torch.tensor(0.006, device=device), torch.tensor(0.08, device=device))
    return dict(
        hand_rot=unit().contiguous(), drawer_pos=drawer_pos, drawer_rot=drawer_rot,
        finger_pos=[(near + torch.randn(n, 3, **f) * spread).contiguous() for _ in range(5)],
        finger_rot=[torch.where(torch.rand(n, 1, **f) < 0.5, ident, unit()).contiguous() for _ in range(5)],
        dof_pos=(lo + (hi - lo) * (torch.rand(n, D, **f) * 1.06 - 0.03)).contiguous(), dof_vel=torch.randn(n, D, **f),
        cabinet_dof_pos=torch.rand(n, 4, **f) * 0.22, cabinet_dof_vel=torch.randn(n, 4, **f) * 0.3,
        actions=torch.rand(n, num_actions, **f) * 2.0 - 1.0, prev_actions=torch.rand(n, num_actions, **f) * 2.0 - 1.0,
        targets=(lo + (hi - lo) * torch.rand(n, D, **f)).contiguous(),
        progress=torch.randint(0, 498, (n,), device=device, generator=g, dtype=torch.int64),
        reset=(torch.rand(n, **f) < 0.03).long(), terminate=(torch.rand(n, **f) < 0.05).long(),
    )


def load_newhand_cabinet_state(task, st: Dict[str, Tensor]) -> None:
    from ..tasks.newhand_cabinet import FINGER_VIEWS

    dev, rb, cab = task.device, task._robots, task._cabinets
    rb._hands.orientations.copy_(st["hand_rot"].to(dev))
    for v, p, q in zip(FINGER_VIEWS, st["finger_pos"], st["finger_rot"]):
        getattr(rb, v).positions.copy_(p.to(dev))
        getattr(rb, v).orientations.copy_(q.to(dev))
    cab._drawers.positions.copy_(st["drawer_pos"].to(dev))
    cab._drawers.orientations.copy_(st["drawer_rot"].to(dev))
    rb.joint_positions.copy_(st["dof_pos"].to(dev))
    rb.joint_velocities.copy_(st["dof_vel"].to(dev))
    cab.joint_positions.copy_(st["cabinet_dof_pos"].to(dev))
    cab.joint_velocities.copy_(st["cabinet_dof_vel"].to(dev))
    task.actions.copy_(st["prev_actions"].to(dev))
    task.robot_dof_targets.copy_(st["targets"].to(dev))
    task.progress_buf.copy_(st["progress"].to(dev))
    task.reset_buf.copy_(st["reset"].to(dev))
    task.terminate_buf.copy_(st["terminate"].to(dev))
