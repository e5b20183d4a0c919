"""Container only (needs /root/reference): the Quadcopter oracle equals the reference's own method source
(tasks/quadcopter.py) executed on a stub ``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import math
import types

import pytest
import torch

from oracle import quadcopter_oracle as qo
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/quadcopter.py"
METHODS = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done"]


def _rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _View:
    name = "quadcopter_view"
    num_dof = 8

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.calls = {}
        self.rotors = types.SimpleNamespace(apply_forces=lambda f, is_global=False: self.calls.__setitem__("forces", f.clone()))

    def get_world_poses(self, clone=True):
        return (self.pos.clone(), self.rot.clone()) if clone else (self.pos, self.rot)

    def get_velocities(self, clone=True):
        return self.vel

    def get_joint_positions(self, clone=True):
        return self.q

    def set_joint_positions(self, v, indices=None): self.calls["q"] = v.clone()
    def set_joint_velocities(self, v, indices=None): self.calls["qd"] = v.clone()
    def set_world_poses(self, p=None, r=None, indices=None, positions=None): self.calls["pose"] = (p.clone(), r.clone())
    def set_velocities(self, v, indices=None): self.calls["vel"] = v.clone()
    def set_joint_position_targets(self, v, indices=None): self.calls["targets"] = v.clone()


def inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    spec = qo.QuadcopterSpec()
    env_pos = torch.randn(n, 3, generator=g) * 4
    env_pos[:, 2] = 0
    q = torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    st = dict(root_pos=env_pos + torch.cat([torch.randn(n, 2, generator=g) * 1.2, torch.rand(n, 1, generator=g) * 2.0 + 0.1], -1),
              root_rot=q, root_vel=torch.randn(n, 6, generator=g) * 2, dof_pos=torch.rand(n, 8, generator=g) * 1.6 - 0.8,
              env_pos=env_pos)
    tp = torch.zeros(n, 3)
    tp[:, 2] = 1.0
    b = qo.QuadcopterBuffers(
        obs_buf=torch.zeros(n, 21), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.3).long(),
        progress_buf=torch.randint(0, 501, (n,), generator=g), target_positions=tp,
        dof_position_targets=torch.rand(n, 8, generator=g) * 1.6 - 0.8, thrusts=torch.rand(n, 4, generator=g) * 4 - 2,
        forces=torch.randn(n, 4, 3, generator=g))
    if n > 8:
        b.progress_buf[:4] = torch.tensor([497, 498, 499, 500])
        st["root_pos"][4:7, 2] = torch.tensor([0.29999, 0.3, 0.30001])
        st["root_vel"][7:9, 5] = torch.tensor([40.0, -45.0])  # spinnage^2 / 400 beats the other terms: reward clips at 0
    return spec, st, b


def stub(spec, st, b):
    ns = {"quat_axis": tu.quat_axis, "tensor_clamp": tu.tensor_clamp, "torch_rand_float": _rand_float, "math": math}
    cls = type("RefQuadcopter", (), {m: ref_exec.load_function(F, m, cls="QuadcopterTask", extra=ns) for m in METHODS})
    o = cls()
    n = st["root_pos"].shape[0]
    o._copters = _View(pos=st["root_pos"].clone(), rot=st["root_rot"].clone(), vel=st["root_vel"].clone(), q=st["dof_pos"].clone())
    o._balls = types.SimpleNamespace(set_world_poses=lambda positions=None: None)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device = "cpu"
    o._num_envs = n
    o._env_pos = st["env_pos"].clone()
    o._max_episode_length = spec.max_episode_length
    o.dt = spec.dt
    o.thrust_lower_limits = -spec.max_thrust * torch.ones(4)
    o.thrust_upper_limits = spec.max_thrust * torch.ones(4)
    o.dof_lower_limits, o.dof_upper_limits = spec.lower.clone(), spec.upper.clone()
    for k, v in b.clone().__dict__.items():
        setattr(o, k, v)
    o.root_pos, o.root_rot = o._copters.pos, o._copters.rot
    o.root_velocities, o.dof_pos, o.dof_vel = o._copters.vel, o._copters.q, torch.randn(n, 8)
    o.initial_root_pos, o.initial_root_rot = st["root_pos"].clone() + 0.5, st["root_rot"].clone()
    return o


@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    qo.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], st["dof_pos"], st["env_pos"])
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert 0 < int(b.reset_buf.sum()) < n and (b.rew_buf == 0).any() and (b.rew_buf > 0).any()


@pytest.mark.parametrize("n,seed", [(33, 2), (700, 3)])
def test_pre_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    actions = torch.rand(n, 12, generator=torch.Generator().manual_seed(9)) * 2 - 1
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    draw = lambda ids: (torch.rand(len(ids), 8), torch.rand(len(ids), 1), torch.rand(len(ids), 1), torch.rand(len(ids), 1))
    ids, rows = qo.pre_physics(spec, b, actions, ref.initial_root_pos, ref.initial_root_rot, draw)
    assert len(ids) > 0
    for k in ("reset_buf", "progress_buf", "dof_position_targets", "thrusts", "forces"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
    c = ref._copters.calls
    assert torch.equal(rows.dof_pos, c["q"]) and torch.equal(rows.dof_vel, c["qd"]) and torch.equal(rows.root_vel, c["vel"])
    assert torch.equal(rows.root_pos, c["pose"][0]) and torch.equal(rows.root_rot, c["pose"][1])
    assert torch.equal(b.dof_position_targets, c["targets"]) and torch.equal(b.forces, c["forces"])
