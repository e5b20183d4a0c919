"""Container only (needs /root/reference): the in-hand manipulation oracle (ShadowHand and AllegroHand) equals the
reference's own source executed under CPU torch, bit for bit:

  * the jit functions compute_hand_reward / randomize_rotation (in_hand_manipulation.py:449-531), compiled by
    TorchScript from the reference text on the restated omni helpers (helper layer unpinned, composition pinned);
  * the methods get_object_goal_observations / calculate_metrics / pre_physics_step / reset_target_pose / reset_idx
    (in_hand_manipulation.py:256-441) and the packers of shadow_hand.py:120-262 / allegro_hand.py:101-144, run on a
    stub ``self`` with torch's RNG replayed from the same seed.
"""
import types

import pytest
import torch

from oracle import hand_oracle as ho
from oracle import ref_exec
from oracle import torch_utils as tu

pytestmark = pytest.mark.skipif(not ref_exec.available(), reason="/root/reference not present (GPU box)")
SHARED = "tasks/shared/in_hand_manipulation.py"
SHARED_METHODS = ["get_object_goal_observations", "calculate_metrics", "pre_physics_step", "reset_target_pose", "reset_idx"]
PACKERS = {
    "shadow": ("tasks/shadow_hand.py", "ShadowHandTask",
               ["get_observations", "compute_fingertip_observations", "compute_full_observations", "compute_full_state"]),
    "allegro": ("tasks/allegro_hand.py", "AllegroHandTask", ["get_observations", "compute_full_observations"]),
}


def _rand_float(lower, upper, shape, device):
    return (upper - lower) * torch.rand(*shape, device=device) + lower


class _Prims:
    def __init__(self, pos, rot, vel):
        self.pos, self.rot, self.vel = pos, rot, vel
        self.calls = {}

    def get_world_poses(self, clone=True):
        return (self.pos.clone(), self.rot.clone()) if clone else (self.pos, self.rot)

    def get_velocities(self, clone=True):
        return self.vel

    def set_world_poses(self, p, r, indices=None):
        self.calls.setdefault("poses", []).append((p.clone(), r.clone(), indices.clone()))

    def set_velocities(self, v, indices=None):
        self.calls["vel"] = (v.clone(), indices.clone())


class _Hands:
    name = "hand_view"

    def __init__(self, q, qd, fingers, sensor):
        self.q, self.qd, self._fingers, self.sensor = q, qd, fingers, sensor
        self._sensor_indices = None
        self.calls = {}

    def get_joint_positions(self, clone=True):
        return self.q

    def get_joint_velocities(self, clone=True):
        return self.qd

    def get_measured_joint_forces(self, joint_indices=None):
        return self.sensor

    def set_joint_position_targets(self, v, indices=None, joint_indices=None):
        self.calls["targets" if joint_indices is not None else "reset_targets"] = v.clone()

    def set_joint_positions(self, v, indices=None):
        self.calls["q"] = v.clone()

    def set_joint_velocities(self, v, indices=None):
        self.calls["qd"] = v.clone()


def _inputs(spec: ho.HandSpec, n: int, seed: int):
    g = torch.Generator().manual_seed(seed)
    nf, nd, na = spec.num_fingertips, spec.num_hand_dofs, spec.num_actions
    unit = lambda m: torch.nn.functional.normalize(torch.randn(m, 4, generator=g), dim=-1)
    env_pos = torch.randn(n, 3, generator=g) * 3
    goal_pos = torch.randn(n, 3, generator=g) * 0.1
    s = ho.HandState(
        object_pos=env_pos + goal_pos + torch.randn(n, 3, generator=g) * 0.12, object_rot=unit(n),
        object_velocities=torch.randn(n, 6, generator=g),
        fingertip_pos=torch.randn(n * nf, 3, generator=g) if nf else None, fingertip_rot=unit(n * nf) if nf else None,
        fingertip_velocities=torch.randn(n * nf, 6, generator=g) if nf else None,
        hand_dof_pos=spec.lower + (spec.upper - spec.lower) * torch.rand(n, nd, generator=g),
        hand_dof_vel=torch.randn(n, nd, generator=g), vec_sensor=torch.randn(n, nf, 6, generator=g) if nf else None,
        env_pos=env_pos)
    b = ho.HandBuffers.fresh(spec, n)
    b.goal_pos, b.goal_rot = goal_pos, unit(n)
    k = max(n // 10, 1)
    b.goal_rot[:k] = s.object_rot[:k]  # successes
    b.reset_buf = (torch.rand(n, generator=g) < 0.2).long()
    b.reset_goal_buf = (torch.rand(n, generator=g) < 0.2).long()
    b.progress_buf = torch.randint(0, spec.max_episode_length + 1, (n,), generator=g)
    b.successes = torch.randint(0, 52, (n,), generator=g).float()
    b.consecutive_successes = torch.tensor([1.25])
    b.actions = torch.rand(n, na, generator=g) * 2 - 1
    b.prev_targets = spec.lower + (spec.upper - spec.lower) * torch.rand(n, nd, generator=g)
    b.cur_targets = spec.lower + (spec.upper - spec.lower) * torch.rand(n, nd, generator=g)
    init = ho.HandInit(object_init_pos=torch.randn(n, 3, generator=g) * 0.1, goal_init_pos=torch.randn(n, 3, generator=g) * 0.1,
                       env_pos=env_pos, default_pos=torch.zeros(nd), default_vel=torch.zeros(nd))
    return s, b, init


def _stub(spec: ho.HandSpec, s: ho.HandState, b: ho.HandBuffers, init: ho.HandInit):
    jit = {name: ref_exec.load_function(SHARED, name, helpers=True) for name in ("compute_hand_reward", "randomize_rotation")}
    ns = {h: getattr(tu, h) for h in ("quat_mul", "quat_conjugate", "unscale", "scale", "tensor_clamp")}
    ns.update(jit, torch_rand_float=_rand_float)
    methods = {m: ref_exec.load_function(SHARED, m, cls="InHandManipulationTask", extra=ns) for m in SHARED_METHODS}
    rel, cls_name, names = PACKERS[spec.hand]
    methods.update({m: ref_exec.load_function(rel, m, cls=cls_name, extra=ns) for m in names})
    o = type("RefHand", (), methods)()
    n, nf = s.object_pos.shape[0], spec.num_fingertips
    fingers = _Prims(s.fingertip_pos.clone(), s.fingertip_rot.clone(), s.fingertip_velocities.clone()) if nf else None
    o._hands = _Hands(s.hand_dof_pos.clone(), s.hand_dof_vel.clone(), fingers, s.vec_sensor.clone() if nf else None)
    o._objects = _Prims(s.object_pos.clone(), s.object_rot.clone(), s.object_velocities.clone())
    o._goals = _Prims(None, None, None)
    o.world = types.SimpleNamespace(is_playing=lambda: True)
    o._dr_randomizer = types.SimpleNamespace(randomize=False)
    o._device = o.device = "cpu"
    o._num_envs = o.num_envs = n
    o._env_pos = s.env_pos.clone()
    for k in ("obs_type", "num_hand_dofs", "num_actions", "num_fingertips", "dist_reward_scale", "rot_reward_scale", "rot_eps",
              "action_penalty_scale", "success_tolerance", "reach_goal_bonus", "fall_dist", "fall_penalty", "vel_obs_scale",
              "force_torque_obs_scale", "reset_position_noise", "reset_dof_pos_noise", "reset_dof_vel_noise",
              "hand_dof_speed_scale", "use_relative_control", "act_moving_average", "max_episode_length",
              "max_consecutive_successes", "dt", "actuated_dof_indices"):
        setattr(o, k, getattr(spec, k))
    o.asymmetric_obs = spec.asymmetric_obs
    o.av_factor = torch.tensor(spec.av_factor, dtype=torch.float)  # in_hand_manipulation.py:59
    o.print_success_stat = False
    o.extras = {}
    o.randomization_buf = torch.zeros(n, dtype=torch.long)
    o.hand_dof_lower_limits, o.hand_dof_upper_limits = spec.lower.clone(), spec.upper.clone()
    o.hand_dof_default_pos, o.hand_dof_default_vel = init.default_pos.clone(), init.default_vel.clone()
    o.object_init_pos, o.goal_init_pos = init.object_init_pos.clone(), init.goal_init_pos.clone()
    o.object_init_velocities = torch.zeros(n, 6)
    o.goal_displacement_tensor = torch.tensor(spec.goal_displacement)
    o.x_unit_tensor = torch.tensor([1, 0, 0], dtype=torch.float).repeat((n, 1))
    o.y_unit_tensor = torch.tensor([0, 1, 0], dtype=torch.float).repeat((n, 1))
    o.hand_dof_targets = torch.zeros(n, spec.num_hand_dofs)
    for k, v in b.clone().__dict__.items():
        setattr(o, k, v)
    return o


SPECS = {
    "shadow_openai_ff": ho.openai_ff_spec,
    "shadow_full": lambda: ho.HandSpec(obs_type="full"),
    "shadow_full_asym_rel": lambda: ho.HandSpec(obs_type="full", asymmetric_obs=True, use_relative_control=True),
    "allegro_full": lambda: ho.allegro_spec("full"),
    "allegro_full_no_vel": lambda: ho.allegro_spec("full_no_vel"),
    "allegro_mcs_ema": lambda: ho.HandSpec(hand="allegro", num_hand_dofs=16, num_actions=16, num_fingertips=0, obs_type="full",
                                           actuated_dof_indices=list(range(16)), max_consecutive_successes=50,
                                           fall_penalty=-50.0, act_moving_average=0.3, reset_dof_vel_noise=0.1),
}


@pytest.mark.parametrize("kind", list(SPECS))
@pytest.mark.parametrize("n,seed", [(37, 0), (600, 1)])
def test_post_physics_equals_reference_source(kind, n, seed):
    spec = SPECS[kind]()
    s, b, init = _inputs(spec, n, seed)
    ref = _stub(spec, s, b, init)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    ho.post_physics(spec, s, b)
    names = ["obs_buf", "rew_buf", "reset_buf", "reset_goal_buf", "progress_buf", "successes", "consecutive_successes"]
    if spec.asymmetric_obs:
        names.append("states_buf")
    for k in names:
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
        assert getattr(b, k).dtype == getattr(ref, k).dtype, k
    assert int(b.reset_buf.sum()) > 0 and int(b.reset_goal_buf.sum()) > 0
    assert torch.equal(ref.extras["consecutive_successes"], b.consecutive_successes.mean())


@pytest.mark.parametrize("kind", list(SPECS))
@pytest.mark.parametrize("n,seed", [(37, 2), (600, 3)])
def test_pre_physics_equals_reference_source(kind, n, seed):
    spec = SPECS[kind]()
    s, b, init = _inputs(spec, n, seed)
    ref = _stub(spec, s, b, init)
    actions = torch.rand(n, spec.num_actions, generator=torch.Generator().manual_seed(5)) * 2 - 1
    torch.manual_seed(77)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(77)
    nd = spec.num_hand_dofs
    draw_goal = lambda ids: 2.0 * torch.rand(len(ids), 4) + (-1.0)

    def draw_reset(ids):
        rf = 2.0 * torch.rand(len(ids), 2 * nd + 5) + (-1.0)
        return rf, 2.0 * torch.rand(len(ids), 4) + (-1.0)

    targets, goal_ids, goal_rows, env_ids, rows = ho.pre_physics(spec, b, init, actions, draw_goal=draw_goal, draw_reset=draw_reset)
    for k in ("reset_buf", "reset_goal_buf", "progress_buf", "successes", "goal_pos", "goal_rot", "actions", "prev_targets",
              "cur_targets"):
        assert torch.equal(getattr(b, k), getattr(ref, k)), k
    assert len(goal_ids) > 0 and len(env_ids) > 0
    h, ob, gl = ref._hands.calls, ref._objects.calls, ref._goals.calls
    assert torch.equal(targets, h["targets"])
    assert torch.equal(rows["dof_pos"], h["q"]) and torch.equal(rows["dof_vel"], h["qd"]) and torch.equal(rows["dof_pos"], h["reset_targets"])
    assert torch.equal(rows["object_pos"], ob["poses"][0][0]) and torch.equal(rows["object_rot"], ob["poses"][0][1])
    assert torch.equal(rows["object_vel"], ob["vel"][0]) and torch.equal(ob["vel"][1], env_ids.to(torch.int32))
    # goal-only ids first, then the nested call of reset_idx (:324-329, :401)
    assert torch.equal(goal_rows[0], gl["poses"][0][0]) and torch.equal(goal_rows[1], gl["poses"][0][1])
    assert torch.equal(gl["poses"][0][2], goal_ids.to(torch.int32))
    assert torch.equal(rows["goal_world_pos"], gl["poses"][1][0]) and torch.equal(rows["goal_rot"], gl["poses"][1][1])
