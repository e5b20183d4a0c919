"""Container only (needs /root/reference): the AnymalTerrain oracle equals the reference's own
method source.  The reference class's post_physics_step / check_termination / calculate_metrics /
reset_idx / update_terrain_level / get_observations / get_heights are compiled verbatim from
tasks/anymal_terrain.py, attached to a stub object (fake views, fake world), and executed under CPU
torch on top of the restated omni helpers; torch's RNG is replayed from the same seed."""
import types

import numpy as np
import pytest
import torch

from oracle import anymal_terrain_oracle as ao
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/anymal_terrain.py"
METHODS = ["post_physics_step", "check_termination", "calculate_metrics", "reset_idx", "update_terrain_level",
           "get_observations", "get_heights", "refresh_dof_state_tensors", "refresh_body_state_tensors", "pre_physics_step"]


def _torch_rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _View:
    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.calls = []

    def get_world_poses(self, clone=True):
        return self.pos, self.rot

    def get_velocities(self, clone=True):
        return self.vel

    def get_joint_positions(self, clone=True):
        return self.q

    def get_joint_velocities(self, clone=True):
        return self.qd

    def get_net_contact_forces(self, clone=True):
        return self.contact

    def set_world_poses(self, positions=None, orientations=None, indices=None):
        self.calls.append(("poses", positions.clone(), orientations.clone(), indices.clone()))

    def set_velocities(self, velocities=None, indices=None):
        self.calls.append(("vel", velocities.clone(), indices.clone()))

    def set_joint_positions(self, positions=None, indices=None):
        self.calls.append(("q", positions.clone(), indices.clone()))

    def set_joint_velocities(self, velocities=None, indices=None):
        self.calls.append(("qd", velocities.clone(), indices.clone()))

    def set_joint_efforts(self, t):
        self.calls.append(("efforts", t.clone()))


def _make_inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    spec = ao.AnymalSpec()
    rows, cols = 1200, 2000
    hs = torch.randint(-60, 200, (rows, cols), generator=g, dtype=torch.int16)
    origins = torch.zeros(10, 20, 3)
    for i in range(10):
        for j in range(20):
            origins[i, j] = torch.tensor([(i + 0.5) * 8.0, (j + 0.5) * 8.0, 0.01 * ((i * 7 + j) % 9)])
    levels = torch.randint(0, 10, (n,), generator=g)
    types = torch.randint(0, 20, (n,), generator=g)
    env_origins = origins[levels, types].clone()
    base_pos = env_origins + torch.cat([torch.rand(n, 2, generator=g) * 10 - 5, torch.rand(n, 1, generator=g) * 0.6 + 0.3], -1)
    if n > 4:  # scan points that clip at the map border and negative coordinates (trunc toward zero)
        base_pos[0] = torch.tensor([-20.3, -19.95, 0.5])
        base_pos[1] = torch.tensor([99.9, 180.2, 0.5])
        base_pos[2] = torch.tensor([-25.0, 300.0, 0.5])
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    s = ao.AnymalState(
        base_pos=base_pos, base_quat=q, base_velocities=torch.randn(n, 6, generator=g),
        dof_pos=torch.tensor(ao.DEFAULT_JOINT_ANGLES) + torch.randn(n, 12, generator=g) * 0.3,
        dof_vel=torch.randn(n, 12, generator=g) * 2, knee_contact=torch.randn(4 * n, 3, generator=g).abs() * 0.7,
        base_contact=torch.randn(n, 3, generator=g).abs() * 0.5)
    commands = torch.cat([torch.rand(n, 2, generator=g) * 2 - 1, torch.zeros(n, 1), torch.rand(n, 1, generator=g) * 6.28 - 3.14], -1)
    b = ao.AnymalBuffers(
        obs_buf=torch.zeros(n, 188), rew_buf=torch.zeros(n), reset_buf=torch.ones(n, dtype=torch.long),
        timeout_buf=torch.zeros(n, dtype=torch.long), progress_buf=torch.randint(0, 1001, (n,), generator=g),
        commands=commands, actions=torch.rand(n, 12, generator=g) * 2 - 1, last_actions=torch.rand(n, 12, generator=g) * 2 - 1,
        last_dof_vel=torch.randn(n, 12, generator=g) * 2, torques=torch.randn(n, 12, generator=g) * 30,
        feet_air_time=torch.rand(n, 4, generator=g), episode_sums={k: torch.rand(n, generator=g) for k in ao.EPISODE_KEYS},
        terrain_levels=levels, terrain_types=types, env_origins=env_origins,
        default_dof_pos=torch.tensor(ao.DEFAULT_JOINT_ANGLES).repeat(n, 1))
    if n > 8:
        b.progress_buf[3:7] = torch.tensor([997, 998, 999, 1000])
    return spec, s, b, hs, origins


def _reference_stub(spec, s, b, hs, origins):
    ns = {"quat_rotate_inverse": tu.quat_rotate_inverse, "quat_apply": tu.quat_apply, "normalize": tu.normalize,
          "torch_rand_float": _torch_rand_float, "SimulationContext": types.SimpleNamespace(step=lambda w, render=False: None)}
    ns["quat_apply_yaw"] = ref_exec.load_function(F, "quat_apply_yaw", helpers=True)
    ns["wrap_to_pi"] = ref_exec.load_function(F, "wrap_to_pi")
    n = s.base_pos.shape[0]
    cls = type("RefAnymalTerrain", (), {m: ref_exec.load_function(F, m, cls="AnymalTerrainTask", extra=ns) for m in METHODS})
    o = cls()
    knees = _View(contact=s.knee_contact.clone(), pos=torch.zeros(4 * n, 3), rot=torch.zeros(4 * n, 4))
    base = _View(contact=s.base_contact.clone())
    o._anymals = _View(pos=s.base_pos.clone(), rot=s.base_quat.clone(), vel=s.base_velocities.clone(), q=s.dof_pos.clone(),
                       qd=s.dof_vel.clone(), _knees=knees, _base=base)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o.get_states = lambda: None
    o.push_robots = lambda: None
    o._num_envs = o.num_envs = n
    o.device = "cpu"
    o.num_dof = 12
    o.common_step_counter = 0
    o.push_interval = int(spec.push_interval_s / spec.dt + 0.5)
    o.max_episode_length = spec.max_episode_length
    o.max_episode_length_s = spec.max_episode_length_s
    o.rew_scales = spec.rew_scales
    o.lin_vel_scale, o.ang_vel_scale = spec.lin_vel_scale, spec.ang_vel_scale
    o.dof_pos_scale, o.dof_vel_scale, o.height_meas_scale = spec.dof_pos_scale, spec.dof_vel_scale, spec.height_meas_scale
    o.commands_scale = torch.tensor([spec.lin_vel_scale, spec.lin_vel_scale, spec.ang_vel_scale])
    o.gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat((n, 1))
    o.forward_vec = torch.tensor([1.0, 0.0, 0.0]).repeat((n, 1))
    o.add_noise = spec.add_noise
    o.noise_scale_vec = ao.noise_scale_vec(spec)
    o.init_done, o.curriculum = True, spec.curriculum
    o.terrain = types.SimpleNamespace(border_size=spec.border_size, horizontal_scale=spec.horizontal_scale,
                                      vertical_scale=spec.vertical_scale, env_length=spec.env_length, env_rows=spec.env_rows)
    o.height_samples, o.terrain_origins = hs, origins
    o.height_points = ao.init_height_points(n)
    o.num_height_points = 140
    o.base_init_state = torch.tensor(spec.base_init_state, dtype=torch.float)
    o.command_x_range, o.command_y_range, o.command_yaw_range = spec.command_x_range, spec.command_y_range, spec.command_yaw_range
    o.Kp, o.Kd, o.action_scale, o.decimation = spec.Kp, spec.Kd, spec.action_scale, 1
    c = b.clone()
    for k in ("obs_buf", "rew_buf", "reset_buf", "timeout_buf", "progress_buf", "commands", "actions", "last_actions",
              "last_dof_vel", "torques", "feet_air_time", "episode_sums", "terrain_levels", "env_origins", "default_dof_pos"):
        setattr(o, k, getattr(c, k))
    o.terrain_types = c.terrain_types
    o.extras = {}
    return o


@pytest.mark.parametrize("n,seed", [(40, 0), (600, 1)])
def test_post_physics_step_equals_reference_source(n, seed):
    spec, s, b, hs, origins = _make_inputs(n, seed)
    ref = _reference_stub(spec, s, b, hs, origins)
    torch.manual_seed(1000 + seed)
    ref.post_physics_step()

    torch.manual_seed(1000 + seed)

    def draw(ids):
        k = len(ids)
        return ao.AnymalResetDraws(torch.rand(k, 12), torch.rand(k, 12), torch.rand(k, 2), torch.rand(k, 1), torch.rand(k, 1),
                                   torch.rand(k, 1))

    class _LazyNoise:  # rand_like must be drawn AFTER the reset draws, as in the reference
        pass

    env_ids, view = ao.post_physics(spec, b, s, hs, origins, draw=draw, noise_u=None)
    assert len(env_ids) > 0
    for k in ("obs_buf", "rew_buf", "reset_buf", "timeout_buf", "progress_buf", "commands", "last_actions", "last_dof_vel",
              "feet_air_time", "terrain_levels", "env_origins"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert b.reset_buf.dtype == torch.bool
    for k in ao.EPISODE_KEYS:
        assert torch.equal(b.episode_sums[k], ref.episode_sums[k]), k
        assert torch.equal(b.extras["episode"]["rew_" + k], ref.extras["episode"]["rew_" + k]), k
    assert torch.equal(b.extras["episode"]["terrain_level"], ref.extras["episode"]["terrain_level"])
    # rows handed to the view setters
    calls = {c[0]: c for c in ref._anymals.calls}
    assert torch.equal(calls["poses"][1], view["base_pos"][env_ids]) and torch.equal(calls["poses"][2], view["base_quat"][env_ids])
    assert torch.equal(calls["q"][1], view["dof_pos"][env_ids]) and torch.equal(calls["qd"][1], view["dof_vel"][env_ids])
    assert torch.equal(calls["vel"][1], view["base_velocities"][env_ids])


def test_pd_torques_and_wrap_to_pi_equal_reference_source():
    spec, s, b, hs, origins = _make_inputs(100, 5)
    ref = _reference_stub(spec, s, b, hs, origins)
    ref.dof_pos, ref.dof_vel = s.dof_pos.clone(), s.dof_vel.clone()
    ref.pre_physics_step(b.actions.clone())
    assert torch.equal(ref.torques, ao.pd_torques(spec, b.actions, b.default_dof_pos, s.dof_pos, s.dof_vel))
    x = torch.linspace(-30, 30, 2001)
    assert torch.equal(ref_exec.load_function(F, "wrap_to_pi")(x.clone()), ao.wrap_to_pi(x.clone()))
