"""Container only (needs /root/reference): the Crazyflie oracle equals the reference's own method source
(tasks/crazyflie.py) executed on a stub ``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import types

import numpy as np
import pytest
import torch

from oracle import crazyflie_oracle as co
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/crazyflie.py"
METHODS = ["get_observations", "pre_physics_step", "reset_idx", "set_targets", "calculate_metrics", "is_done"]


def _rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _View:
    name = "crazyflie_view"
    num_dof = 4

    def __init__(self, **kw):
        self.__dict__.update(kw)
        self.calls = {}
        rec = lambda i: types.SimpleNamespace(apply_forces=lambda f, indices=None: self.calls.__setitem__(f"thrust{i}", f.clone()))
        self.physics_rotors = [rec(i) for i in range(4)]

    def get_world_poses(self, clone=True):
        return (self.pos.clone(), self.rot.clone()) if clone else (self.pos, self.rot)

    def get_velocities(self, clone=True):
        return self.vel

    def set_joint_positions(self, v, indices=None): self.calls["q"] = v.clone()
    def set_joint_velocities(self, v, indices=None): self.calls["qd_all" if indices is None else "qd"] = v.clone()
    def set_world_poses(self, p, r, indices=None): self.calls["pose"] = (p.clone(), r.clone())
    def set_velocities(self, v, indices=None): self.calls["vel"] = v.clone()


def inputs(n, seed):
    g = torch.Generator().manual_seed(seed)
    spec = co.CrazyflieSpec()
    env_pos = torch.randn(n, 3, generator=g) * 6
    env_pos[:, 2] = 0
    q = torch.randn(n, 4, generator=g) * 0.6
    q[:, 0] += 1.0
    q = torch.nn.functional.normalize(q, dim=-1)
    tp = torch.zeros(n, 3)
    tp[:, 2] = 2.0
    st = dict(root_pos=env_pos + tp + torch.randn(n, 3, generator=g) * 1.2, root_rot=q, root_vel=torch.randn(n, 6, generator=g) * 0.7,
              env_pos=env_pos)
    b = co.CrazyflieBuffers(
        obs_buf=torch.zeros(n, 18), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.25).long(),
        progress_buf=torch.randint(0, 701, (n,), generator=g), target_positions=tp, actions=torch.rand(n, 4, generator=g) * 2 - 1,
        thrusts=torch.randn(n, 4, 3, generator=g), thrust_cmds_damp=torch.rand(n, 4, generator=g), thrust_rot_damp=torch.rand(n, 4, generator=g),
        dof_pos=torch.rand(n, 4, generator=g), dof_vel=torch.randn(n, 4, generator=g),
        episode_sums={k: torch.rand(n, generator=g) * 10 for k in co.EPISODE_KEYS})
    if n > 10:
        b.progress_buf[:4] = torch.tensor([697, 698, 699, 700])
        b.progress_buf[7:10] = torch.tensor([500, 0, 499])
        b.reset_buf[7:10] = 0
        st["root_pos"][4:7, 2] = torch.tensor([0.49999, 0.5, 5.00001])
        st["root_rot"][10] = torch.tensor([0.0, 1.0, 0.0, 0.0])  # upside down: orient_z < 0
    return spec, st, b


def stub(spec, st, b):
    ns = {"quat_axis": tu.quat_axis, "torch_rand_float": _rand_float, "np": np, "EPS": co.EPS}
    cls = type("RefCrazyflie", (), {m: ref_exec.load_function(F, m, cls="CrazyflieTask", extra=ns) for m in METHODS})
    o = cls()
    n = st["root_pos"].shape[0]
    o._copters = _View(pos=st["root_pos"].clone(), rot=st["root_rot"].clone(), vel=st["root_vel"].clone())
    balls = {}
    o._balls = types.SimpleNamespace(set_world_poses=lambda p, r, indices=None: balls.update(pose=(p.clone(), r.clone(), indices.clone())))
    o._ball_calls = balls
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device = "cpu"
    o._num_envs = n
    o._env_pos = st["env_pos"].clone()
    o._max_episode_length, o.dt = spec.max_episode_length, spec.dt
    o.motor_tau_up, o.motor_tau_down = spec.motor_tau_up, spec.motor_tau_down
    o.thrust_max, o.prop_max_rot = spec.thrust_max, spec.prop_max_rot
    o.all_indices = torch.arange(n, dtype=torch.int32)
    c = b.clone()
    for k, v in c.__dict__.items():
        setattr(o, k, v)
    o.root_pos, o.root_rot, o.root_velocities = o._copters.pos, o._copters.rot, o._copters.vel
    o.initial_root_pos, o.initial_root_rot = st["root_pos"].clone() + 0.25, st["root_rot"].clone()
    o.initial_ball_rot = torch.nn.functional.normalize(torch.randn(n, 4, generator=torch.Generator().manual_seed(3)), dim=-1)
    return o


@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    co.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], st["env_pos"])
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    for k in co.EPISODE_KEYS:
        assert torch.equal(b.episode_sums[k], ref.episode_sums[k]), k
    assert int(b.reset_buf[10]) == 1 and int(b.reset_buf[2:4].sum()) == 2 and 0 < int(b.reset_buf.sum()) < n


@pytest.mark.parametrize("n,seed", [(33, 2), (700, 3)])
def test_pre_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    actions = torch.rand(n, 4, generator=torch.Generator().manual_seed(9)) * 3 - 1.5
    k = int((b.reset_buf != 0).sum())
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    torch.rand(k, 4), torch.rand(k, 1), torch.rand(k, 1), torch.rand(k, 1)  # reset_idx's four zero-range draws
    noise4 = 0.01 * torch.randn(4, dtype=torch.float32)
    ids, rows, tids, trows = co.pre_physics(spec, b, actions, st["root_rot"], st["env_pos"], ref.initial_root_pos, ref.initial_root_rot,
                                            ref.initial_ball_rot, noise4)
    assert len(ids) > 0 and len(tids) > len(ids)
    for name in ("reset_buf", "progress_buf", "target_positions", "actions", "thrusts", "thrust_cmds_damp", "thrust_rot_damp", "dof_pos",
                 "dof_vel"):
        assert torch.equal(getattr(b, name), getattr(ref, name)), name
    for name in co.EPISODE_KEYS:
        assert torch.equal(b.episode_sums[name], ref.episode_sums[name]), name
        assert torch.equal(b.extras["episode"][name], ref.extras["episode"][name]), name
    c = ref._copters.calls
    assert torch.equal(rows["dof_pos"], c["q"]) and torch.equal(rows["dof_vel"], c["qd"]) and torch.equal(rows["root_vel"], c["vel"])
    assert torch.equal(rows["root_pos"], c["pose"][0]) and torch.equal(rows["root_rot"], c["pose"][1])
    assert torch.equal(b.dof_vel, c["qd_all"])
    for i in range(4):
        assert torch.equal(b.thrusts[:, i], c[f"thrust{i}"])
    bp = ref._ball_calls["pose"]
    assert torch.equal(trows["ball_pos"], bp[0]) and torch.equal(trows["ball_rot"], bp[1]) and torch.equal(bp[2], tids)
