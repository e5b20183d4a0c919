"""Container only (needs /root/reference): the five-finger cabinet oracle equals the reference's own method source
(tasks/tomodj_newhand_cabinet.py, tasks/tomodj_newhandTendon_cabinet.py, tasks/tomospc_newhand_cabinet.py) executed on a stub
``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import types

import pytest
import torch

from oracle import newhand_cabinet_oracle as no
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
TENDON_ACTUATED = list(range(7)) + [7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20, 22, 23, 24]
VARIANTS = {
    "tomodj_newhand": dict(file="tasks/tomodj_newhand_cabinet.py", cls="TomoDJnewhandCabinetTask", robot="tomodj", spec=no.tomodj_newhand_spec),
    "tomodj_tendon": dict(file="tasks/tomodj_newhandTendon_cabinet.py", cls="TomoDJnewhandTendonCabinetTask", robot="tomodj",
                          spec=lambda: no.tomodj_newhand_tendon_spec(TENDON_ACTUATED)),
    "tomospc_newhand": dict(file="tasks/tomospc_newhand_cabinet.py", cls="TomoSPCnewhandCabinetTask", robot="tomospc", spec=no.tomospc_newhand_spec),
}
FINGER_VIEWS = ["_thumbs", "_forefingers", "_middle_fingers", "_ring_fingers", "_little_fingers"]


class _Rigid:
    def __init__(self, pos, rot):
        self.pos, self.rot, self.calls = pos, rot, {}

    def get_world_poses(self, clone=True):
        return self.pos, self.rot

    def set_world_poses(self, p, r, indices=None):
        self.calls["pose"] = (p.clone(), r.clone(), indices.clone())


class _Art:
    def __init__(self, name, q, qd):
        self.name, self.q, self.qd, self.calls = name, q, qd, {}
        self.num_dof, self.count = q.shape[1], q.shape[0]

    def get_joint_positions(self, clone=True): return self.q
    def get_joint_velocities(self, clone=True): return self.qd
    def set_joint_positions(self, v, indices=None): self.calls["q"] = (v.clone(), indices.clone())
    def set_joint_velocities(self, v, indices=None): self.calls["qd"] = (v.clone(), indices.clone())

    def set_joint_position_targets(self, v, indices=None, joint_indices=None):
        self.calls.setdefault("targets", []).append((v.clone(), indices.clone(), joint_indices))


def limits(D):
    g = torch.Generator().manual_seed(D)
    lo = -(torch.rand(D, generator=g) + 0.3)
    hi = torch.rand(D, generator=g) + 0.3
    return lo, hi


def inputs(variant, n, seed):
    spec = VARIANTS[variant]["spec"]()
    D, A = spec.num_dofs, spec.num_actions
    g = torch.Generator().manual_seed(seed)
    q = lambda: torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    lo, hi = limits(D)
    drawer_pos = torch.randn(n, 3, generator=g) * 3
    drawer_rot = torch.where(torch.rand(n, 1, generator=g) < 0.6, torch.tensor([1.0, 0.0, 0.0, 0.0]), q())
    near = drawer_pos + torch.tensor([0.29, 0.0, 0.01])
    spread = torch.where(torch.rand(n, 1, generator=g) < 0.5, torch.tensor(0.006), torch.tensor(0.08))
    st = dict(hand_rot=q(), drawer_pos=drawer_pos, drawer_rot=drawer_rot,
              finger_pos=[near + torch.randn(n, 3, generator=g) * spread for _ in range(5)],
              finger_rot=[torch.where(torch.rand(n, 1, generator=g) < 0.5, torch.tensor([1.0, 0.0, 0.0, 0.0]), q()) for _ in range(5)],
              dof_pos=lo + (hi - lo) * (torch.rand(n, D, generator=g) * 1.06 - 0.03), dof_vel=torch.randn(n, D, generator=g),
              cabinet_dof_pos=torch.rand(n, 4, generator=g) * 0.22, cabinet_dof_vel=torch.randn(n, 4, generator=g))
    b = no.NewhandBuffers(obs_buf=torch.zeros(n, 2 * D + 5), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.2).long(),
                          progress_buf=torch.randint(0, 498, (n,), generator=g), actions=torch.rand(n, A, generator=g) * 2 - 1,
                          dof_targets=lo + (hi - lo) * torch.rand(n, D, generator=g), keypoints_finger=torch.zeros(n, 5, 3),
                          keypoints_drawer=torch.zeros(n, 5, 3), terminate_buf=(torch.rand(n, generator=g) < 0.1).long())
    if n > 12:
        b.progress_buf[:3] = torch.tensor([496, 497, 498])
        st["cabinet_dof_pos"][3:9, 3] = torch.tensor([0.0099, 0.0101, 0.1, 0.1001, 0.19, 0.1901])
    return spec, st, b, lo, hi


def stub(variant, spec, st, b, lo, hi):
    info = VARIANTS[variant]
    r = info["robot"]
    ns = {"tf_combine": tu.tf_combine, "tf_vector": tu.tf_vector, "tensor_clamp": tu.tensor_clamp}
    methods = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done", f"compute_{r}_reward"]
    cls = type("RefNewhand", (), {m: ref_exec.load_function(info["file"], m, cls=info["cls"], extra=ns) for m in methods})
    o = cls()
    n, D = st["drawer_pos"].shape[0], spec.num_dofs
    robots = _Art(f"{r}_view", st["dof_pos"].clone(), st["dof_vel"].clone())
    robots._hands = _Rigid(torch.zeros(n, 3), st["hand_rot"].clone())
    for name, p, q in zip(FINGER_VIEWS, st["finger_pos"], st["finger_rot"]):
        setattr(robots, name, _Rigid(p.clone(), q.clone()))
    setattr(o, f"_{r}s", robots)
    o._cabinets = _Art("cabinet_view", st["cabinet_dof_pos"].clone(), st["cabinet_dof_vel"].clone())
    o._cabinets._drawers = _Rigid(st["drawer_pos"].clone(), st["drawer_rot"].clone())
    o._props = _Rigid(None, None)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device, o.device, o._num_envs, o.num_envs = "cpu", "cpu", n, n
    o._max_episode_length, o.dt, o.action_scale, o.dof_vel_scale = spec.max_episode_length, spec.dt, spec.action_scale, spec.dof_vel_scale
    for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale", "action_penalty_scale", "num_props"):
        setattr(o, k, getattr(spec, k))
    o.finger_dist_reward_scale, o.finger_close_reward_scale, o.distX_offset = 15.0, 10.0, 0.04
    rep = lambda v: torch.tensor(v, dtype=torch.float32).repeat((n, 1))
    for k in ("left_hand_forward_axis", "drawer_inward_axis", "left_hand_up_axis", "drawer_up_axis"):
        setattr(o, k, rep(getattr(spec, k)))
    setattr(o, f"{r}_local_thumb_pose_pos", rep(spec.local_thumb_pose_pos))
    setattr(o, f"{r}_local_another_finger_pose_pos", rep(spec.local_another_finger_pose_pos))
    o.drawer_local_keypoint_offsets = torch.tensor(spec.drawer_local_keypoint_offsets)
    o.identity_quat = rep([1.0, 0.0, 0.0, 0.0])
    setattr(o, f"{r}_default_dof_pos", torch.tensor(spec.default_dof_pos))
    setattr(o, f"{r}_dof_lower_limits", lo.clone())
    setattr(o, f"{r}_dof_upper_limits", hi.clone())
    speed = torch.ones(D)
    speed[D - 19:] = 5
    setattr(o, f"{r}_dof_speed_scales", speed)
    setattr(o, f"num_{r}_dofs", D)
    setattr(o, f"{r}_dof_pos", robots.q)
    setattr(o, f"{r}_dof_targets", b.dof_targets.clone())
    if spec.actuated_dof_indices is not None:
        o.actuated_dof_indices = list(spec.actuated_dof_indices)
    g = torch.Generator().manual_seed(77)
    o.default_prop_pos, o.default_prop_rot = torch.randn(n * spec.num_props, 3, generator=g), torch.randn(n * spec.num_props, 4, generator=g)
    o.prop_indices = torch.arange(n * spec.num_props).view(n, spec.num_props)
    c = b.clone()
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "actions", "keypoints_finger", "keypoints_drawer", "terminate_buf"):
        setattr(o, k, getattr(c, k))
    return o, speed


@pytest.mark.parametrize("variant", list(VARIANTS))
@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(variant, n, seed):
    spec, st, b, lo, hi = inputs(variant, n, seed)
    ref, _ = stub(variant, spec, st, b, lo, hi)
    term_in = b.terminate_buf.clone()
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    no.post_physics(spec, b, st, lo, hi)
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "keypoints_finger", "keypoints_drawer"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
    assert 0 < int(b.reset_buf.sum()) < n
    if spec.terminate_on_limits:
        assert torch.equal(b.terminate_buf, ref.terminate_buf)
        assert bool((b.rew_buf[term_in == 1] == -100).all()) and int((term_in == 1).sum()) > 0   # the PREVIOUS step's flags
        assert 0 < int(b.terminate_buf.sum()) < n
    if n > 100:
        dx = torch.stack([b.keypoints_finger[:, i, 0] - st["drawer_pos"][:, 0] for i in range(5)], -1)
        assert int(((dx > spec.around_lower) & (dx < spec.around_upper)).all(-1).sum()) > 3   # the around-handle branch is live


@pytest.mark.parametrize("variant", list(VARIANTS))
def test_pre_physics_equals_reference_source(variant):
    n = 400
    spec, st, b, lo, hi = inputs(variant, n, 4)
    ref, speed = stub(variant, spec, st, b, lo, hi)
    D, A, r = spec.num_dofs, spec.num_actions, VARIANTS[variant]["robot"]
    actions = torch.rand(n, A, generator=torch.Generator().manual_seed(9)) * 3 - 1.5
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    ids, rows = no.pre_physics(spec, b, actions, lo, hi, speed, lambda i: torch.rand((len(i), D)), default_prop_pos=ref.default_prop_pos,
                               default_prop_rot=ref.default_prop_rot)
    assert len(ids) > 0
    assert torch.equal(b.reset_buf, ref.reset_buf) and torch.equal(b.progress_buf, ref.progress_buf) and torch.equal(b.actions, ref.actions)
    assert torch.equal(b.dof_targets, getattr(ref, f"{r}_dof_targets"))
    fc, cc = getattr(ref, f"_{r}s").calls, ref._cabinets.calls
    assert torch.equal(rows["dof_pos"], fc["q"][0]) and torch.equal(rows["dof_vel"], fc["qd"][0])
    assert torch.equal(rows["cabinet_dof_pos"], cc["q"][0]) and torch.equal(rows["cabinet_dof_vel"], cc["qd"][0])
    pp = ref._props.calls["pose"]
    assert torch.equal(rows["prop_pos"], pp[0]) and torch.equal(rows["prop_ids"].to(torch.int32), pp[2])
    last = fc["targets"][-1]
    if spec.actuated_dof_indices is None:
        assert torch.equal(last[0], b.dof_targets)
    else:
        assert torch.equal(last[0], b.dof_targets[:, spec.actuated_dof_indices]) and list(last[2]) == spec.actuated_dof_indices
