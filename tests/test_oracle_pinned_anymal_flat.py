"""Container only (needs /root/reference): the Anymal (flat) oracle equals the reference's own method source
(tasks/anymal.py) executed on a stub ``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import types

import pytest
import torch

from oracle import anymal_oracle as an
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/anymal.py"
METHODS = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done"]


def _rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _View:
    name = "anymalview"
    count = 0
    num_dof = 12

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.calls = {}

    def get_world_poses(self, clone=True):
        return (self.pos.clone(), self.rot.clone()) if clone else (self.pos, self.rot)

    def get_velocities(self, clone=True):
        return self.vel

    def get_joint_positions(self, clone=True):
        return self.q

    def get_joint_velocities(self, clone=True):
        return self.qd

    def is_base_below_threshold(self, threshold, ground_heights):  # anymal_view.py:76-80
        base_pos, _ = self.get_world_poses()
        base_heights = base_pos[:, 2]
        base_heights -= ground_heights
        return base_heights[:] < threshold

    def set_joint_positions(self, v, indices=None): self.calls["q"] = v.clone()
    def set_joint_velocities(self, v, indices=None): self.calls["qd"] = v.clone()
    def set_world_poses(self, p, r, indices=None): self.calls["pose"] = (p.clone(), r.clone())
    def set_velocities(self, v, indices=None): self.calls["vel"] = v.clone()
    def set_joint_position_targets(self, v, indices=None): self.calls["targets"] = v.clone()


def _inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g)
    q = q / q.norm(dim=-1, keepdim=True)
    pos = torch.cat([torch.randn(n, 2, generator=g), torch.rand(n, 1, generator=g) * 0.4 + 0.35], -1)
    default = torch.tensor(an.DEFAULT_JOINT_ANGLES).repeat(n, 1)
    st = dict(pos=pos, rot=q, vel=torch.randn(n, 6, generator=g), dof_pos=default + torch.randn(n, 12, generator=g) * 0.3,
              dof_vel=torch.randn(n, 12, generator=g) * 2)
    b = an.AnymalFlatBuffers(
        obs_buf=torch.zeros(n, 48), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.3).long(),
        progress_buf=torch.randint(0, 3001, (n,), generator=g), commands=torch.rand(n, 3, generator=g) * 2 - 1,
        actions=torch.rand(n, 12, generator=g) * 2 - 1, last_actions=torch.rand(n, 12, generator=g) * 2 - 1,
        last_dof_vel=torch.randn(n, 12, generator=g) * 2, current_targets=default + torch.randn(n, 12, generator=g) * 0.2,
        default_dof_pos=default)
    if n > 8:
        b.progress_buf[:4] = torch.tensor([2997, 2998, 2999, 3000])
        st["pos"][4:7, 2] = torch.tensor([0.50999, 0.51, 0.51001])
    lower, upper = default[0] - 0.5, default[0] + 0.5
    return an.AnymalFlatSpec(), st, b, lower, upper


def _stub(spec, st, b, lower, upper):
    ns = {"quat_rotate": tu.quat_rotate, "quat_rotate_inverse": tu.quat_rotate_inverse, "tensor_clamp": tu.tensor_clamp,
          "torch_rand_float": _rand_float}
    cls = type("RefAnymal", (), {m: ref_exec.load_function(F, m, cls="AnymalTask", extra=ns) for m in METHODS})
    o = cls()
    n = st["pos"].shape[0]
    o._anymals = _View(pos=st["pos"].clone(), rot=st["rot"].clone(), vel=st["vel"].clone(), q=st["dof_pos"].clone(),
                       qd=st["dof_vel"].clone(), count=n)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device = o.device = "cpu"
    o._num_envs = n
    for k in ("lin_vel_scale", "ang_vel_scale", "dof_pos_scale", "dof_vel_scale", "action_scale", "dt", "max_episode_length",
              "command_x_range", "command_y_range", "command_yaw_range", "rew_scales"):
        setattr(o, k, getattr(spec, k))
    o.gravity_vec = torch.tensor([0.0, 0.0, -1.0]).repeat((n, 1))
    c = b.clone()
    for k, v in c.__dict__.items():
        setattr(o, k, v)
    o.commands_x, o.commands_y, o.commands_yaw = o.commands[..., 0], o.commands[..., 1], o.commands[..., 2]
    o.anymal_dof_lower_limits, o.anymal_dof_upper_limits = lower, upper
    o.initial_root_pos, o.initial_root_rot = st["pos"].clone() + 1.0, st["rot"].clone()
    return o


@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(n, seed):
    spec, st, b, lower, upper = _inputs(n, seed)
    ref = _stub(spec, st, b, lower, upper)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    an.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"])
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "last_actions", "last_dof_vel"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert int(b.reset_buf.sum()) > 0 and (b.rew_buf == -1).any()


@pytest.mark.parametrize("n,seed", [(33, 2), (700, 3)])
def test_pre_physics_equals_reference_source(n, seed):
    spec, st, b, lower, upper = _inputs(n, seed)
    ref = _stub(spec, st, b, lower, upper)
    actions = torch.rand(n, 12, generator=torch.Generator().manual_seed(9)) * 2 - 1
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)

    def draw(ids):
        k = len(ids)
        return torch.rand(k, 12), torch.rand(k, 1), torch.rand(k, 1), torch.rand(k, 1)

    rows = an.pre_physics(spec, b, actions, lower, upper, ref.initial_root_pos, ref.initial_root_rot, draw=draw)
    for k in ("reset_buf", "progress_buf", "commands", "actions", "last_actions", "last_dof_vel", "current_targets"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
    c = ref._anymals.calls
    assert torch.equal(rows.dof_pos, c["q"]) and torch.equal(rows.dof_vel, c["qd"]) and torch.equal(rows.root_vel, c["vel"])
    assert torch.equal(rows.root_pos, c["pose"][0]) and torch.equal(rows.root_rot, c["pose"][1])
    assert torch.equal(b.current_targets, c["targets"])
