"""Container only (needs /root/reference): the BallBalance oracle equals the reference's own method source
(tasks/ball_balance.py) executed on a stub ``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import types

import pytest
import torch

from oracle import ball_balance_oracle as bo
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/ball_balance.py"
METHODS = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done"]


def _rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _Prims:
    def __init__(self, pos, rot, vel):
        self.pos, self.rot, self.vel, self.calls = pos, rot, vel, {}

    def get_world_poses(self, clone=True):
        return (self.pos.clone(), self.rot.clone()) if clone else (self.pos, self.rot)

    def get_velocities(self, clone=True):
        return self.vel

    def set_world_poses(self, p, r, indices=None): self.calls["pose"] = (p.clone(), r.clone(), indices.clone())
    def set_velocities(self, v, indices=None): self.calls["vel"] = v.clone()


class _Bots(_Prims):
    num_dof = 6

    def __init__(self, pos, rot, vel, q, qd, sensors):
        super().__init__(pos, rot, vel)
        self.q, self.qd, self.sensors = q, qd, sensors

    def get_joint_positions(self, clone=True): return self.q
    def get_joint_velocities(self, clone=True): return self.qd
    def get_measured_joint_forces(self, joint_indices=None): return self.sensors
    def set_joint_positions(self, v, indices=None): self.calls["q"] = v.clone()
    def set_joint_position_targets(self, v): self.calls["targets"] = v.clone()


def inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    spec = bo.BallBalanceSpec()
    env_pos = torch.randn(n, 3, generator=g) * 5
    env_pos[:, 2] = 0
    st = dict(ball_pos=env_pos + torch.cat([torch.randn(n, 2, generator=g) * 0.3, torch.rand(n, 1, generator=g) * 1.2 + 0.05], -1),
              ball_vel=torch.randn(n, 6, generator=g) * 1.5, dof_pos=torch.rand(n, 6, generator=g) * 2 - 1,
              dof_vel=torch.randn(n, 6, generator=g), sensors=torch.randn(n, 3, 6, generator=g) * 30, env_pos=env_pos)
    b = bo.BallBalanceBuffers(obs_buf=torch.zeros(n, 24), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.25).long(),
                              progress_buf=torch.randint(0, 601, (n,), generator=g),
                              dof_position_targets=torch.rand(n, 6, generator=g) * 2.4 - 1.2)
    if n > 8:
        b.progress_buf[:4] = torch.tensor([597, 598, 599, 600])
        b.reset_buf[:4] = 0
        st["ball_pos"][4:7, 2] = torch.tensor([0.14999, 0.15, 0.15001])
        b.reset_buf[4:7] = 0
    q = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    init = bo.BallBalanceInit(ball_pos=env_pos + torch.tensor([0.0, 0.0, 1.0]), ball_rot=q.clone(), ball_vel=torch.randn(n, 6, generator=g),
                              bot_pos=env_pos + torch.tensor([0.0, 0.0, 0.56]), bot_rot=q.clone(), bot_vel=torch.zeros(n, 6),
                              dof_pos=torch.rand(n, 6, generator=g) * 0.2)
    return spec, st, b, init


def stub(spec, st, b, init):
    ns = {"tensor_clamp": tu.tensor_clamp, "torch_rand_float": _rand_float, "torch_random_dir_2": tu.torch_random_dir_2}
    cls = type("RefBallBalance", (), {m: ref_exec.load_function(F, m, cls="BallBalanceTask", extra=ns) for m in METHODS})
    o = cls()
    n = st["ball_pos"].shape[0]
    q0 = torch.zeros(n, 4)
    q0[:, 0] = 1
    o._balls = _Prims(st["ball_pos"].clone(), q0, st["ball_vel"].clone())
    o._balance_bots = _Bots(init.bot_pos.clone(), init.bot_rot.clone(), init.bot_vel.clone(), st["dof_pos"].clone(),
                            st["dof_vel"].clone(), st["sensors"].clone())
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device = o.device = "cpu"
    o._num_envs = o.num_envs = n
    o._env_pos = st["env_pos"].clone()
    o._dt, o._action_speed_scale, o._max_episode_length, o._ball_radius = spec.dt, spec.action_speed_scale, spec.max_episode_length, spec.ball_radius
    o.actuated_dof_indices = torch.tensor(spec.actuated_dof_indices)
    o._sensor_indices = torch.tensor([0, 1, 2])
    o.bbot_dof_lower_limits, o.bbot_dof_upper_limits = spec.lower.clone(), spec.upper.clone()
    o.initial_ball_pos, o.initial_ball_rot, o.initial_ball_velocities = init.ball_pos.clone(), init.ball_rot.clone(), init.ball_vel.clone()
    o.initial_bot_pos, o.initial_bot_rot, o.initial_bot_velocities = init.bot_pos.clone(), init.bot_rot.clone(), init.bot_vel.clone()
    o.initial_dof_positions = init.dof_pos.clone()
    for k, v in b.clone().__dict__.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(n, seed):
    spec, st, b, init = inputs(n, seed)
    ref = stub(spec, st, b, init)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    bo.post_physics(spec, b, st["ball_pos"], st["ball_vel"], st["dof_pos"], st["dof_vel"], st["sensors"], st["env_pos"])
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert list(b.reset_buf[:7]) == [0, 1, 1, 1, 1, 0, 0]  # progress 598 no, 599.. yes; z 0.14999 yes, 0.15 / 0.15001 no


@pytest.mark.parametrize("n,seed", [(33, 2), (700, 3)])
def test_pre_physics_equals_reference_source(n, seed):
    spec, st, b, init = inputs(n, seed)
    ref = stub(spec, st, b, init)
    actions = torch.rand(n, 3, generator=torch.Generator().manual_seed(9)) * 2 - 1
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    draw = lambda ids: torch.cat([torch.rand(len(ids), 1) for _ in range(5)], dim=-1)
    ids, rows = bo.pre_physics(spec, b, init, actions, draw)
    assert len(ids) > 0
    for k in ("reset_buf", "progress_buf", "dof_position_targets"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
    bc, tc = ref._balls.calls, ref._balance_bots.calls
    assert torch.equal(rows["ball_pos"], bc["pose"][0]) and torch.equal(rows["ball_rot"], bc["pose"][1])
    assert torch.equal(rows["ball_vel"], bc["vel"]) and torch.equal(bc["pose"][2], ids.to(torch.int32))
    assert torch.equal(rows["bot_pos"], tc["pose"][0]) and torch.equal(rows["bot_rot"], tc["pose"][1])
    assert torch.equal(rows["bot_vel"], tc["vel"]) and torch.equal(rows["dof_pos"], tc["q"])
    assert torch.equal(b.dof_position_targets, tc["targets"])
