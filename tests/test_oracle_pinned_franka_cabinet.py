"""Container only (needs /root/reference): the FrankaCabinet oracle equals the reference's own method source
(tasks/franka_cabinet.py) executed on a stub ``self`` under CPU torch, bit for bit, torch's RNG replayed from the same seed."""
import types

import pytest
import torch

from oracle import franka_cabinet_oracle as fo
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
F = "tasks/franka_cabinet.py"
METHODS = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done", "compute_grasp_transforms",
           "compute_franka_reward"]
LOWER = torch.tensor([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973, 0.0, 0.0])
UPPER = torch.tensor([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973, 0.04, 0.04])


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
    def set_joint_position_targets(self, v, indices=None): self.calls.setdefault("targets", []).append((v.clone(), indices.clone()))


def inputs(n, seed, distinct_fingers=False):
    g = torch.Generator().manual_seed(seed)
    q = lambda: torch.nn.functional.normalize(torch.randn(n, 4, generator=g), dim=-1)
    spec = fo.FrankaCabinetSpec()
    drawer_pos = torch.randn(n, 3, generator=g) * 3
    hand_pos = drawer_pos + torch.tensor([0.3, 0.01, 0.0]) + torch.randn(n, 3, generator=g) * 0.15
    st = dict(hand_pos=hand_pos, hand_rot=q(), drawer_pos=drawer_pos, drawer_rot=q(),
              franka_dof_pos=LOWER + (UPPER - LOWER) * torch.rand(n, 9, generator=g), franka_dof_vel=torch.randn(n, 9, generator=g),
              cabinet_dof_pos=torch.rand(n, 4, generator=g) * 0.45, cabinet_dof_vel=torch.randn(n, 4, generator=g),
              lfinger_pos=hand_pos + torch.randn(n, 3, generator=g) * 0.05)
    st["rfinger_pos"] = hand_pos + torch.randn(n, 3, generator=g) * 0.05 if distinct_fingers else st["lfinger_pos"]
    b = fo.FrankaCabinetBuffers(obs_buf=torch.zeros(n, 23), rew_buf=torch.zeros(n), reset_buf=(torch.rand(n, generator=g) < 0.2).long(),
                                progress_buf=torch.randint(0, 500, (n,), generator=g), actions=torch.rand(n, 9, generator=g) * 2 - 1,
                                franka_dof_targets=LOWER + (UPPER - LOWER) * torch.rand(n, 9, generator=g),
                                franka_grasp_rot=torch.zeros(n, 4), franka_grasp_pos=torch.zeros(n, 3), drawer_grasp_rot=torch.zeros(n, 4),
                                drawer_grasp_pos=torch.zeros(n, 3))
    if n > 12:
        b.progress_buf[:3] = torch.tensor([497, 498, 499])
        st["cabinet_dof_pos"][3:9, 3] = torch.tensor([0.0099, 0.0101, 0.2, 0.2001, 0.39, 0.3901])
        st["hand_pos"][9] = st["drawer_pos"][9]     # close: the d <= 0.02 / 0.03 branches depend on the local grasp offsets
    return spec, st, b


def stub(spec, st, b, distinct_fingers=False):
    ns = {"tf_combine": tu.tf_combine, "tf_vector": tu.tf_vector, "tensor_clamp": tu.tensor_clamp}
    cls = type("RefFranka", (), {m: ref_exec.load_function(F, m, cls="FrankaCabinetTask", extra=ns) for m in METHODS})
    o = cls()
    n = st["hand_pos"].shape[0]
    o._frankas = _Art("franka_view", st["franka_dof_pos"].clone(), st["franka_dof_vel"].clone())
    o._frankas._hands = _Rigid(st["hand_pos"].clone(), st["hand_rot"].clone())
    o._frankas._lfingers = _Rigid(st["lfinger_pos"].clone(), st["hand_rot"].clone())
    o._frankas._rfingers = _Rigid(st["rfinger_pos"].clone(), st["hand_rot"].clone())
    o._cabinets = _Art("cabinet_view", st["cabinet_dof_pos"].clone(), st["cabinet_dof_vel"].clone())
    o._cabinets._drawers = _Rigid(st["drawer_pos"].clone(), st["drawer_rot"].clone())
    o._props = _Rigid(None, None)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device, o._num_envs, o.num_envs = "cpu", n, n
    o._max_episode_length, o.dt, o.action_scale, o.dof_vel_scale = spec.max_episode_length, spec.dt, spec.action_scale, spec.dof_vel_scale
    for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale", "finger_dist_reward_scale",
              "action_penalty_scale", "finger_close_reward_scale", "num_props"):
        setattr(o, k, getattr(spec, k))
    o.distX_offset = 0.04
    rep = lambda v: torch.tensor(v, dtype=torch.float32).repeat((n, 1))
    for k in ("franka_local_grasp_pos", "franka_local_grasp_rot", "drawer_local_grasp_pos", "drawer_local_grasp_rot", "gripper_forward_axis",
              "drawer_inward_axis", "gripper_up_axis", "drawer_up_axis"):
        setattr(o, k, rep(getattr(spec, k)))
    o.franka_default_dof_pos = torch.tensor(fo.FRANKA_DEFAULT_DOF_POS)
    o.franka_dof_lower_limits, o.franka_dof_upper_limits = LOWER.clone(), UPPER.clone()
    o.franka_dof_speed_scales = torch.ones(9)
    o.franka_dof_speed_scales[[7, 8]] = 0.1
    o.num_franka_dofs = 9
    o.franka_dof_pos = o._frankas.q
    g = torch.Generator().manual_seed(77)
    o.default_prop_pos, o.default_prop_rot = torch.randn(n * spec.num_props, 3, generator=g), torch.randn(n * spec.num_props, 4, generator=g)
    o.prop_indices = torch.arange(n * spec.num_props).view(n, spec.num_props)
    c = b.clone()
    for k, v in c.__dict__.items():
        setattr(o, k, v)
    return o


@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_post_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    fo.post_physics(spec, b, st, LOWER, UPPER)
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "franka_grasp_rot", "franka_grasp_pos", "drawer_grasp_rot", "drawer_grasp_pos"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert 0 < int(b.reset_buf.sum()) < n
    # the reference reads the right finger from the left-finger view (:336-337): both bonuses are identically zero there
    assert torch.equal(ref.franka_rfinger_pos, ref.franka_lfinger_pos)


def test_reward_with_distinct_fingers_equals_reference_function():
    """compute_franka_reward itself (fed two different finger positions) -- the branches the task's own binding never reaches."""
    n = 500
    spec, st, b = inputs(n, 5, distinct_fingers=True)
    ref = stub(spec, st, b)
    fo.observations(spec, b, st, LOWER, UPPER)
    fo.metrics(spec, b, st)
    want = ref.compute_franka_reward(b.reset_buf, b.progress_buf, b.actions, st["cabinet_dof_pos"], b.franka_grasp_pos, b.drawer_grasp_pos,
                                     b.franka_grasp_rot, b.drawer_grasp_rot, st["lfinger_pos"], st["rfinger_pos"], ref.gripper_forward_axis,
                                     ref.drawer_inward_axis, ref.gripper_up_axis, ref.drawer_up_axis, n, spec.dist_reward_scale,
                                     spec.rot_reward_scale, spec.around_handle_reward_scale, spec.open_reward_scale,
                                     spec.finger_dist_reward_scale, spec.action_penalty_scale, 0.04, spec.max_episode_length,
                                     st["franka_dof_pos"], spec.finger_close_reward_scale)
    assert torch.equal(b.rew_buf, want)
    lf, rf, dg = st["lfinger_pos"][:, 2], st["rfinger_pos"][:, 2], b.drawer_grasp_pos[:, 2]
    assert int(((lf > dg) & (rf < dg)).sum()) > 20


@pytest.mark.parametrize("n,seed", [(33, 2), (700, 3)])
def test_pre_physics_equals_reference_source(n, seed):
    spec, st, b = inputs(n, seed)
    ref = stub(spec, st, b)
    actions = torch.rand(n, 9, generator=torch.Generator().manual_seed(9)) * 3 - 1.5
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    speed = torch.ones(9)
    speed[[7, 8]] = 0.1
    ids, rows = fo.pre_physics(spec, b, actions, LOWER, UPPER, speed, lambda i: torch.rand((len(i), 9)), num_props=spec.num_props,
                               default_prop_pos=ref.default_prop_pos, default_prop_rot=ref.default_prop_rot)
    assert len(ids) > 0
    for name in ("reset_buf", "progress_buf", "actions", "franka_dof_targets"):
        assert torch.equal(getattr(b, name), getattr(ref, name)), name
    fc, cc = ref._frankas.calls, ref._cabinets.calls
    assert torch.equal(rows["dof_pos"], fc["q"][0]) and torch.equal(rows["dof_vel"], fc["qd"][0])
    assert torch.equal(fc["q"][1], ids.to(torch.int32))
    assert torch.equal(rows["dof_targets"], fc["targets"][0][0]) and torch.equal(b.franka_dof_targets, fc["targets"][1][0])
    assert torch.equal(rows["cabinet_dof_pos"], cc["q"][0]) and torch.equal(rows["cabinet_dof_vel"], cc["qd"][0])
    pp = ref._props.calls["pose"]
    assert torch.equal(rows["prop_pos"], pp[0]) and torch.equal(rows["prop_rot"], pp[1]) and torch.equal(rows["prop_ids"].to(torch.int32), pp[2])
    # reset_idx writes the new positions into the view's own buffer through self.franka_dof_pos (:390)
    assert torch.equal(ref._frankas.q[ids], rows["dof_pos"])


# ---------------------------------------------------------------- TomoDJCabinet / TomoSPCCabinet (same skeleton, own reward)
class _Op:
    def __init__(self, kind): self.kind = kind
    def GetOpType(self): return self.kind
    def Set(self, v): pass


class _UsdGeom:
    """What tomodj_cabinet.py:397-422 touches per env to move two debug cubes: harmless stand-ins."""
    class XformOp:
        TypeTranslate, TypeOrient = "translate", "orient"

    @staticmethod
    def Xformable(prim):
        return types.SimpleNamespace(GetOrderedXformOps=lambda: [_Op("translate"), _Op("orient")])


_Gf = types.SimpleNamespace(Vec3f=lambda *a: a, Quatd=lambda *a: a)
TOMO = {
    "tomodj": dict(file="tasks/tomodj_cabinet.py", cls="TomoDJCabinetTask", reward="compute_tomodj_reward", spec=fo.tomodj_spec, module_level_grasp=True),
    "tomospc": dict(file="tasks/tomospc_cabinet.py", cls="TomoSPCCabinetTask", reward="compute_tomospc_reward", spec=fo.tomospc_spec, module_level_grasp=False),
}


def tomo_inputs(robot, n, seed):
    spec = TOMO[robot]["spec"]()
    D = spec.num_dofs
    _, st, b = inputs(n, seed, distinct_fingers=True)
    g = torch.Generator().manual_seed(seed + 100)
    lo, hi = LOWER[9 - D:].clone(), UPPER[9 - D:].clone()
    st["franka_dof_pos"] = lo + (hi - lo) * torch.rand(n, D, generator=g)
    st["franka_dof_vel"] = torch.randn(n, D, generator=g)
    st["cabinet_dof_pos"][:, 3] *= 0.5   # done / bonus thresholds are 0.19 / 0.1 here
    if n > 12:
        st["cabinet_dof_pos"][3:9, 3] = torch.tensor([0.0099, 0.0101, 0.1, 0.1001, 0.19, 0.1901])
    b.actions = torch.rand(n, D, generator=g) * 2 - 1
    b.franka_dof_targets = lo + (hi - lo) * torch.rand(n, D, generator=g)
    b.obs_buf = torch.zeros(n, 2 * D + 5)
    return spec, st, b, lo, hi


def tomo_stub(robot, spec, st, b, lo, hi):
    info, r = TOMO[robot], robot
    ns = {"tf_combine": tu.tf_combine, "tf_vector": tu.tf_vector, "tensor_clamp": tu.tensor_clamp, "UsdGeom": _UsdGeom, "Gf": _Gf}
    if info["module_level_grasp"]:
        ns["compute_grasp_transforms"] = ref_exec.load_function(info["file"], "compute_grasp_transforms", extra=ns)
    methods = ["get_observations", "pre_physics_step", "reset_idx", "calculate_metrics", "is_done", info["reward"]]
    if not info["module_level_grasp"]:
        methods.append("compute_grasp_transforms")
    cls = type("RefTomo", (), {m: ref_exec.load_function(info["file"], m, cls=info["cls"], extra=ns) for m in methods})
    o = cls()
    n, D = st["hand_pos"].shape[0], spec.num_dofs
    o.ensure_tensor = lambda v: v
    robots = _Art(f"{r}_view", st["franka_dof_pos"].clone(), st["franka_dof_vel"].clone())
    robots._hands = robots.hands = _Rigid(st["hand_pos"].clone(), st["hand_rot"].clone())
    robots._lfingers, robots._rfingers = _Rigid(st["lfinger_pos"].clone(), st["hand_rot"].clone()), _Rigid(st["rfinger_pos"].clone(), st["hand_rot"].clone())
    robots.fingertip_views = {"left_fingers_view": robots._lfingers, "right_fingers_view": robots._rfingers}
    setattr(o, f"_{r}s", robots)
    o._cabinets = _Art("cabinet_view", st["cabinet_dof_pos"].clone(), st["cabinet_dof_vel"].clone())
    o._cabinets._drawers = o._cabinets.drawers = _Rigid(st["drawer_pos"].clone(), st["drawer_rot"].clone())
    o._props = _Rigid(None, None)
    o._stage = types.SimpleNamespace(GetPrimAtPath=lambda p: None)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._device, o._num_envs, o.num_envs, o._env_pos = "cpu", n, n, torch.zeros(n, 3)
    o._max_episode_length, o.dt, o.action_scale, o.dof_vel_scale = spec.max_episode_length, spec.dt, spec.action_scale, spec.dof_vel_scale
    for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale", "finger_dist_reward_scale",
              "action_penalty_scale", "finger_close_reward_scale", "num_props"):
        setattr(o, k, getattr(spec, k))
    o.distX_offset, o.num_cabinet_dofs = 0.04, 4
    rep = lambda v: torch.tensor(v, dtype=torch.float32).repeat((n, 1))
    for k in ("drawer_local_grasp_pos", "drawer_local_grasp_rot", "gripper_forward_axis", "drawer_inward_axis", "gripper_up_axis", "drawer_up_axis"):
        setattr(o, k, rep(getattr(spec, k)))
    setattr(o, f"{r}_local_grasp_pos", rep(spec.franka_local_grasp_pos))
    setattr(o, f"{r}_local_grasp_rot", rep(spec.franka_local_grasp_rot))
    setattr(o, f"{r}_default_dof_pos", torch.tensor(spec.default_dof_pos))
    setattr(o, f"{r}_dof_lower_limits", lo.clone())
    setattr(o, f"{r}_dof_upper_limits", hi.clone())
    speed = torch.ones(D)
    speed[[D - 2, D - 1]] = 0.1
    setattr(o, f"{r}_dof_speed_scales", speed)
    setattr(o, f"num_{r}_dofs", D)
    setattr(o, f"{r}_dof_pos", robots.q)
    g = torch.Generator().manual_seed(77)
    o.default_prop_pos, o.default_prop_rot = torch.randn(n * spec.num_props, 3, generator=g), torch.randn(n * spec.num_props, 4, generator=g)
    o.prop_indices = torch.arange(n * spec.num_props).view(n, spec.num_props)
    c = b.clone()
    for k, v in c.__dict__.items():
        setattr(o, k.replace("franka_", f"{r}_"), v)
    return o, speed


@pytest.mark.parametrize("robot", ["tomodj", "tomospc"])
@pytest.mark.parametrize("n,seed", [(33, 0), (700, 1)])
def test_tomo_post_physics_equals_reference_source(robot, n, seed):
    spec, st, b, lo, hi = tomo_inputs(robot, n, seed)
    ref, _ = tomo_stub(robot, spec, st, b, lo, hi)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    fo.post_physics(spec, b, st, lo, hi)
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "franka_grasp_rot", "franka_grasp_pos", "drawer_grasp_rot", "drawer_grasp_pos"):
        assert torch.equal(getattr(b, k), getattr(ref, k.replace("franka_", f"{robot}_"))), k
    assert 0 < int(b.reset_buf.sum()) < n
    lf, rf, dg = st["lfinger_pos"][:, 2], st["rfinger_pos"][:, 2], b.drawer_grasp_pos[:, 2]
    if n > 100:
        assert int(((lf <= dg) & (rf >= dg)).sum()) > 5   # the around-handle / finger-distance terms are live in these tasks


@pytest.mark.parametrize("robot", ["tomodj", "tomospc"])
def test_tomo_pre_physics_equals_reference_source(robot):
    n = 400
    spec, st, b, lo, hi = tomo_inputs(robot, n, 4)
    ref, speed = tomo_stub(robot, spec, st, b, lo, hi)
    D = spec.num_dofs
    actions = torch.rand(n, D, generator=torch.Generator().manual_seed(9)) * 3 - 1.5
    torch.manual_seed(41)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(41)
    ids, rows = fo.pre_physics(spec, b, actions, lo, hi, speed, lambda i: torch.rand((len(i), D)), num_props=spec.num_props,
                               default_prop_pos=ref.default_prop_pos, default_prop_rot=ref.default_prop_rot)
    assert len(ids) > 0
    assert torch.equal(b.reset_buf, ref.reset_buf) and torch.equal(b.progress_buf, ref.progress_buf) and torch.equal(b.actions, ref.actions)
    assert torch.equal(b.franka_dof_targets, getattr(ref, f"{robot}_dof_targets"))
    fc, cc = getattr(ref, f"_{robot}s").calls, ref._cabinets.calls
    assert torch.equal(rows["dof_pos"], fc["q"][0]) and torch.equal(rows["dof_vel"], fc["qd"][0])
    assert torch.equal(rows["cabinet_dof_pos"], cc["q"][0]) and torch.equal(rows["cabinet_dof_vel"], cc["qd"][0])
    pp = ref._props.calls["pose"]
    assert torch.equal(rows["prop_pos"], pp[0]) and torch.equal(rows["prop_ids"].to(torch.int32), pp[2])
