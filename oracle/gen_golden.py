"""Generate tests/golden/*.npz -- golden input/output vectors for the hot path.

    python -m oracle.gen_golden            (container only: needs /root/reference)

Expected outputs come from the REFERENCE'S OWN function source, read from /root/reference at
run time and executed under local CPU torch by oracle/ref_exec.py:
  * pure-torch functions run verbatim: calculate_metrics, is_done, normalize_angle
    (tasks/shared/locomotion.py:219-345), both get_dof_at_limit_cost (tasks/ant.py:106-109,
    tasks/humanoid.py:134-141), Cartpole calculate_metrics / is_done / reset_idx arithmetic
    (tasks/cartpole.py:124-164) -> these outputs are PINNED by the reference;
  * get_observations (locomotion.py:224-284) runs verbatim ON TOP OF the restated omni helpers
    (oracle/torch_utils.py) -> composition pinned, helper layer UNPINNED.
  * the in-hand family (ShadowHand, AllegroHand), Anymal (flat) and AnymalTerrain: the reference's METHODS run on a stub
    ``self`` (the stubs of tests/test_oracle_pinned_*.py), torch's RNG replayed from a recorded seed; every input, every
    output and the draws themselves are stored (the AnymalTerrain heightfield is regenerated from its seed: 4.8 MB).
Only data is written (inputs + expected outputs); no reference text is stored.
"""
from __future__ import annotations

import math
import os

import numpy as np
import torch

from . import cartpole_oracle as co
from . import locomotion_oracle as lo
from . import ref_exec

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
LOCO = "tasks/shared/locomotion.py"


def _nextafter(x: float, toward: float) -> float:
    return float(np.nextafter(np.float32(x), np.float32(toward)))


def locomotion_inputs(spec: lo.LocomotionSpec, n: int, seed: int):
    """Random state (SURVEY.md 8d recipe) with hand-built edge rows in front."""
    g = torch.Generator().manual_seed(seed)
    d, s = spec.num_dof, spec.num_sensors
    lo_, hi_ = spec.dof_limits_lower, spec.dof_limits_upper
    th = spec.termination_height
    zr = (0.2, 0.9) if spec.limit_cost_kind == "ant" else (0.5, 1.5)
    pos = torch.stack([torch.rand(n, generator=g) * 100 - 50, torch.rand(n, generator=g) * 100 - 50,
                       torch.rand(n, generator=g) * (zr[1] - zr[0]) + zr[0]], -1)
    rot = torch.randn(n, 4, generator=g)
    rot = rot / rot.norm(dim=-1, keepdim=True)
    vel = torch.randn(n, 6, generator=g)
    dof_pos = (lo_ - 0.05) + torch.rand(n, d, generator=g) * ((hi_ - lo_) + 0.1)
    dof_vel = torch.randn(n, d, generator=g) * 3
    sensors = torch.randn(n, s, 6, generator=g) * 20
    actions = torch.rand(n, d, generator=g) * 2 - 1
    potentials = -(torch.tensor([1000.0, 0, 0]) - pos)[:, :2].norm(dim=-1) * 60 + torch.randn(n, generator=g)
    progress = torch.randint(0, 1000, (n,), generator=g)
    reset = (torch.rand(n, generator=g) < 0.1).long()

    # ---- edge rows (SURVEY.md 8c "golden vectors to commit")
    r2 = math.sqrt(0.5)
    edge_rot = [
        (1.0, 0.0, 0.0, 0.0),          # identity
        (0.0, 0.0, 0.0, 1.0),          # 180 deg about z: yaw = pi (mod-2pi seam, normalize_angle at +-pi)
        (0.0, 1.0, 0.0, 0.0),          # 180 deg about x
        (r2, 0.0, r2, 0.0),            # pitch +90: gimbal branch |sinp| >= 1
        (r2, 0.0, -r2, 0.0),           # pitch -90
        (r2, 0.0, 0.0, r2),            # yaw +90 (x -> y)
        (r2, 0.0, 0.0, -r2),           # yaw -90
        (-1.0, 0.0, 0.0, 0.0),         # -identity (same rotation)
        (0.9999999, 0.0, 0.0, 4e-4),   # yaw ~ 0+
        (0.9999999, 0.0, 0.0, -4e-4),  # yaw ~ 0-  (euler % 2pi jumps to ~2pi)
        (4e-4, 0.0, 0.0, 0.9999999),   # yaw ~ pi-
        (-4e-4, 0.0, 0.0, 0.9999999),  # yaw ~ -pi+
    ]
    k = min(len(edge_rot), n)
    rot[:k] = torch.tensor(edge_rot[:k])
    z_edge = [th, _nextafter(th, -1.0), _nextafter(th, 10.0)]
    for i, z in enumerate(z_edge):
        if i < n:
            pos[i, 2] = z
    if n > 12:
        pos[12] = torch.tensor([1000.0, 0.0, 0.6])      # to_target == 0 -> normalize eps clamp
        pos[13] = torch.tensor([1000.0, 1e-12, 0.6])    # to_target tiny
        pos[14] = torch.tensor([999.0, 0.0, 1000.0])    # walk_target_angle uses z
    ep = int(spec.episode_length)
    for i, p in enumerate([ep - 3, ep - 2, ep - 1, ep]):  # progress is incremented before is_done
        if 15 + i < n:
            progress[15 + i] = p
            pos[15 + i, 2] = zr[1]
            reset[15 + i] = 0
    if n > 24:
        # scaled dof positions exactly around the limit-cost thresholds (+-0.98, +-0.99, -0.995)
        for row, t in zip(range(19, 25), [0.98, 0.99, -0.98, -0.99, -0.995, 1.0]):
            dof_pos[row] = 0.5 * (t * (hi_ - lo_) + hi_ + lo_)
        actions[19] = 0.0
        actions[20] = 1.0
        actions[21] = -1.0
    return dict(pos=pos, rot=rot, vel=vel, dof_pos=dof_pos, dof_vel=dof_vel, sensors=sensors, actions=actions,
                potentials=potentials, progress=progress, reset=reset)


def locomotion_golden(kind: str, n: int, seed: int):
    spec = lo.ant_spec() if kind == "ant" else lo.humanoid_spec()
    st = locomotion_inputs(spec, n, seed)
    c = lo.per_env_constants(spec, n)
    ref_obs = ref_exec.load_function(LOCO, "get_observations", helpers=True, extra={"__deps__": [(LOCO, "normalize_angle")]})
    ref_met = ref_exec.load_function(LOCO, "calculate_metrics")
    ref_done = ref_exec.load_function(LOCO, "is_done")
    progress_after = st["progress"] + 1  # rl_task.py:289
    obs, pot, prev_pot, up_vec, heading_vec = ref_obs(
        st["pos"], st["rot"], st["vel"][:, 0:3], st["vel"][:, 3:6], st["dof_pos"], st["dof_vel"], c["targets"],
        st["potentials"], spec.dt, c["inv_start_rot"], c["basis_vec0"], c["basis_vec1"], spec.dof_limits_lower,
        spec.dof_limits_upper, spec.dof_vel_scale, st["sensors"], n, spec.contact_force_scale, st["actions"],
        spec.angular_velocity_scale)
    if kind == "ant":
        lim = ref_exec.load_function("tasks/ant.py", "get_dof_at_limit_cost")(obs, spec.num_dof)
    else:
        lim = ref_exec.load_function("tasks/humanoid.py", "get_dof_at_limit_cost")(
            obs, spec.motor_effort_ratio, spec.joints_at_limit_cost_scale)
    rew = ref_met(obs, st["actions"], spec.up_weight, spec.heading_weight, pot, prev_pot, spec.actions_cost_scale,
                  spec.energy_cost_scale, spec.termination_height, spec.death_cost, spec.num_dof, lim,
                  spec.alive_reward_scale, spec.motor_effort_ratio)
    reset_out = ref_done(obs, spec.termination_height, st["reset"], progress_after, float(spec.episode_length))
    data = {f"in_{k}": v.numpy() for k, v in st.items()}
    data.update(
        lo=spec.dof_limits_lower.numpy(), hi=spec.dof_limits_upper.numpy(), gears=spec.joint_gears.numpy(),
        motor_effort_ratio=spec.motor_effort_ratio.numpy(),
        out_obs=obs.numpy(), out_potentials=pot.numpy(), out_prev_potentials=prev_pot.numpy(),
        out_up_vec=up_vec.numpy(), out_heading_vec=heading_vec.numpy(), out_rew=rew.numpy(),
        out_reset=reset_out.numpy(), out_progress=progress_after.numpy(), out_limit_cost=lim.numpy(),
    )
    return data


def cartpole_golden(n: int, seed: int):
    spec = co.CartpoleSpec()
    g = torch.Generator().manual_seed(seed)
    dof_pos = torch.stack([torch.rand(n, generator=g) * 7 - 3.5, torch.rand(n, generator=g) * 4 - 2], -1)
    dof_vel = torch.randn(n, 2, generator=g)
    progress = torch.randint(0, 510, (n,), generator=g)
    dof_pos[0] = torch.tensor([3.0, 0.0])
    dof_pos[1] = torch.tensor([_nextafter(3.0, 10.0), 0.0])
    dof_pos[2] = torch.tensor([0.0, float(np.float32(math.pi / 2))])
    dof_pos[3] = torch.tensor([0.0, -_nextafter(float(np.float32(math.pi / 2)), 10.0)])
    progress[4:8] = torch.tensor([497, 498, 499, 500])
    progress_after = progress + 1
    C = "tasks/cartpole.py"
    obs = co.observations(spec, dof_pos, dof_vel)  # pure slice copies (:95-103)
    stub = ref_exec.Stub(
        cart_pos=obs[:, 0], cart_vel=obs[:, 1], pole_pos=obs[:, 2], pole_vel=obs[:, 3], _reset_dist=spec.reset_dist,
        _max_episode_length=spec.max_episode_length, rew_buf=torch.zeros(n), reset_buf=torch.ones(n, dtype=torch.long),
        progress_buf=progress_after.clone())
    ref_exec.load_function(C, "calculate_metrics", cls="CartpoleTask")(stub)
    ref_exec.load_function(C, "is_done", cls="CartpoleTask")(stub)

    # reset_idx arithmetic (:124-135) with the four torch.rand(n) draws replayed from a fixed seed
    ids = stub.reset_buf.nonzero(as_tuple=False).squeeze(-1)
    torch.manual_seed(seed + 100)
    u = torch.stack([torch.rand(len(ids)) for _ in range(4)])
    torch.manual_seed(seed + 100)
    captured = {}

    class _View:
        num_dof = 2

        def set_joint_positions(self, v, indices=None):
            captured["dof_pos"] = v.clone()

        def set_joint_velocities(self, v, indices=None):
            captured["dof_vel"] = v.clone()

    rstub = ref_exec.Stub(_cartpoles=_View(), _device="cpu", _cart_dof_idx=0, _pole_dof_idx=1,
                          reset_buf=stub.reset_buf.clone(), progress_buf=stub.progress_buf.clone())
    ref_exec.load_function(C, "reset_idx", cls="CartpoleTask")(rstub, ids)
    return dict(
        in_dof_pos=dof_pos.numpy(), in_dof_vel=dof_vel.numpy(), in_progress=progress.numpy(),
        out_obs=obs.numpy(), out_rew=stub.rew_buf.numpy(), out_reset=stub.reset_buf.numpy(),
        out_progress=progress_after.numpy(), reset_ids=ids.numpy(), reset_u=u.numpy(),
        reset_dof_pos=captured["dof_pos"].numpy(), reset_dof_vel=captured["dof_vel"].numpy(),
        reset_reset_buf=rstub.reset_buf.numpy(), reset_progress_buf=rstub.progress_buf.numpy(),
    )


def _tests_on_path():
    import sys

    tests = os.path.join(os.path.dirname(OUT))
    if tests not in sys.path:
        sys.path.insert(0, tests)


def _flat(prefix: str, obj) -> dict:
    """dataclass / dict of tensors -> {prefix_field: ndarray} (None fields skipped, dict fields one level deep)."""
    out = {}
    items = obj.items() if isinstance(obj, dict) else obj.__dict__.items()
    for k, v in items:
        if isinstance(v, torch.Tensor):
            out[f"{prefix}_{k}"] = v.detach().clone().numpy()
        elif isinstance(v, dict):
            for kk, vv in v.items():
                if isinstance(vv, torch.Tensor):
                    out[f"{prefix}_{k}__{kk}"] = vv.detach().clone().numpy()
    return out


def hand_golden(kind: str, n: int, seed: int):
    """ShadowHand / AllegroHand: post_physics_step (progress += 1, get_observations, calculate_metrics) and
    pre_physics_step, each from the same inputs, run from the reference's method source."""
    _tests_on_path()
    import test_oracle_pinned_hand as ph

    spec = ph.SPECS[kind]()
    s, b, init = ph._inputs(spec, n, seed)
    data = {}
    data.update(_flat("s", s))
    data.update(_flat("b", b))
    data.update(_flat("init", init))
    ref = ph._stub(spec, s, b, init)
    ref.progress_buf[:] += 1  # rl_task.py:289
    ref.get_observations()
    ref.calculate_metrics()
    for k in ("obs_buf", "states_buf", "rew_buf", "reset_buf", "reset_goal_buf", "progress_buf", "successes", "consecutive_successes"):
        data[f"post_{k}"] = getattr(ref, k).detach().clone().numpy()
    # pre_physics_step from the same inputs, torch's generator seeded; the draws are recorded in the reference's order
    ref = ph._stub(spec, s, b, init)
    actions = torch.rand(n, spec.num_actions, generator=torch.Generator().manual_seed(seed + 50)) * 2 - 1
    ng, ne = int((b.reset_goal_buf != 0).sum()), int((b.reset_buf != 0).sum())
    torch.manual_seed(seed + 100)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(seed + 100)
    u_goal = 2.0 * torch.rand(ng, 4) + (-1.0)
    u_reset = 2.0 * torch.rand(ne, 2 * spec.num_hand_dofs + 5) + (-1.0)
    u_reset_goal = 2.0 * torch.rand(ne, 4) + (-1.0)
    data.update(pre_actions=actions.numpy(), pre_u_goal=u_goal.numpy(), pre_u_reset=u_reset.numpy(), pre_u_reset_goal=u_reset_goal.numpy())
    for k in ("reset_buf", "reset_goal_buf", "progress_buf", "successes", "goal_pos", "goal_rot", "actions", "prev_targets", "cur_targets"):
        data[f"pre_{k}"] = getattr(ref, k).detach().clone().numpy()
    h, ob, gl = ref._hands.calls, ref._objects.calls, ref._goals.calls
    data.update(pre_targets=h["targets"].numpy(), pre_dof_pos=h["q"].numpy(), pre_dof_vel=h["qd"].numpy(),
                pre_object_pos=ob["poses"][0][0].numpy(), pre_object_rot=ob["poses"][0][1].numpy(),
                pre_goal_world_pos=gl["poses"][0][0].numpy(), pre_goal_rot_rows=gl["poses"][0][1].numpy(),
                pre_r_goal_world_pos=gl["poses"][1][0].numpy(), pre_r_goal_rot=gl["poses"][1][1].numpy())
    return data


def anymal_flat_golden(n: int, seed: int):
    _tests_on_path()
    import test_oracle_pinned_anymal_flat as pf

    spec, st, b, lower, upper = pf._inputs(n, seed)
    data = {f"st_{k}": v.numpy() for k, v in st.items()}
    data.update(_flat("b", b))
    data.update(lower=lower.numpy(), upper=upper.numpy())
    ref = pf._stub(spec, st, b, lower, upper)
    ref.progress_buf[:] += 1
    ref.get_observations()
    ref.calculate_metrics()
    ref.is_done()
    for k in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "last_actions", "last_dof_vel"):
        data[f"post_{k}"] = getattr(ref, k).detach().clone().numpy()
    ref = pf._stub(spec, st, b, lower, upper)
    actions = torch.rand(n, 12, generator=torch.Generator().manual_seed(seed + 50)) * 2 - 1
    k = int((b.reset_buf != 0).sum())
    torch.manual_seed(seed + 100)
    ref.pre_physics_step(actions.clone())
    torch.manual_seed(seed + 100)
    draws = [torch.rand(k, 12), torch.rand(k, 1), torch.rand(k, 1), torch.rand(k, 1)]
    data.update(pre_actions=actions.numpy(), pre_initial_root_pos=ref.initial_root_pos.numpy(),
                pre_initial_root_rot=ref.initial_root_rot.numpy(), **{f"pre_u{i}": d.numpy() for i, d in enumerate(draws)})
    for kk in ("reset_buf", "progress_buf", "commands", "actions", "last_actions", "last_dof_vel", "current_targets"):
        data[f"pre_{kk}"] = getattr(ref, kk).detach().clone().numpy()
    c = ref._anymals.calls
    data.update(pre_dof_pos=c["q"].numpy(), pre_dof_vel=c["qd"].numpy(), pre_root_vel=c["vel"].numpy(),
                pre_root_pos=c["pose"][0].numpy(), pre_root_rot=c["pose"][1].numpy())
    return data


def anymal_terrain_golden(n: int, seed: int):
    """AnymalTerrain post_physics_step from the reference's method source.  The (1200, 2000) int16 heightfield is not
    stored: it is the first draw of torch.Generator().manual_seed(seed) (``hs_sum`` guards the regeneration)."""
    _tests_on_path()
    import test_oracle_pinned_anymal as pa
    from . import anymal_terrain_oracle as ao

    spec, s, b, hs, origins = pa._make_inputs(n, seed)
    data = {}
    data.update(_flat("s", s))
    data.update(_flat("b", b))
    data.update(origins=origins.numpy(), hs_sum=np.array(int(hs.long().sum())), hs_seed=np.array(seed))
    ref = pa._reference_stub(spec, s, b, hs, origins)
    torch.manual_seed(1000 + seed)
    ref.post_physics_step()
    k = int(ref.extras["episode"]["terrain_level"].numel() and len([c for c in ref._anymals.calls if c[0] == "q"][0][1]))
    torch.manual_seed(1000 + seed)
    draws = [torch.rand(k, 12), torch.rand(k, 12), torch.rand(k, 2), torch.rand(k, 1), torch.rand(k, 1), torch.rand(k, 1)]
    noise_u = torch.rand(n, 188)  # rand_like(obs_buf), drawn after the reset draws (:438)
    data.update(noise_u=noise_u.numpy(), **{f"draw{i}": d.numpy() for i, d in enumerate(draws)})
    for kk in ("obs_buf", "rew_buf", "reset_buf", "timeout_buf", "progress_buf", "commands", "last_actions", "last_dof_vel",
               "feet_air_time", "terrain_levels", "env_origins"):
        data[f"post_{kk}"] = getattr(ref, kk).detach().clone().numpy()
    for kk in ao.EPISODE_KEYS:
        data[f"post_episode_sums__{kk}"] = ref.episode_sums[kk].numpy()
        data[f"post_extras__{kk}"] = ref.extras["episode"]["rew_" + kk].numpy()
    data["post_extras__terrain_level"] = ref.extras["episode"]["terrain_level"].numpy()
    calls = {c[0]: c for c in ref._anymals.calls}
    data.update(reset_base_pos=calls["poses"][1].numpy(), reset_base_quat=calls["poses"][2].numpy(),
                reset_dof_pos=calls["q"][1].numpy(), reset_dof_vel=calls["qd"][1].numpy(), reset_base_vel=calls["vel"][1].numpy())
    return data


def main() -> None:
    assert ref_exec.available(), "needs /root/reference (run in the build container)"
    os.makedirs(OUT, exist_ok=True)
    for kind in ("shadow_openai_ff", "shadow_full", "allegro_full", "allegro_mcs_ema"):
        for n, seed in ((64, 0), (257, 1)):
            path = os.path.join(OUT, f"hand_{kind}_n{n}_s{seed}.npz")
            np.savez_compressed(path, **hand_golden(kind, n, seed))
            print("wrote", path, os.path.getsize(path), "bytes")
    for n, seed in ((64, 0), (257, 1)):
        path = os.path.join(OUT, f"anymalflat_n{n}_s{seed}.npz")
        np.savez_compressed(path, **anymal_flat_golden(n, seed))
        print("wrote", path, os.path.getsize(path), "bytes")
    for n, seed in ((40, 0), (257, 1)):
        path = os.path.join(OUT, f"anymalterrain_n{n}_s{seed}.npz")
        np.savez_compressed(path, **anymal_terrain_golden(n, seed))
        print("wrote", path, os.path.getsize(path), "bytes")
    for kind in ("ant", "humanoid"):
        for n, seed in ((64, 0), (257, 1)):
            path = os.path.join(OUT, f"{kind}_n{n}_s{seed}.npz")
            np.savez_compressed(path, **locomotion_golden(kind, n, seed))
            print("wrote", path, os.path.getsize(path), "bytes")
    for n, seed in ((64, 0), (257, 1)):
        path = os.path.join(OUT, f"cartpole_n{n}_s{seed}.npz")
        np.savez_compressed(path, **cartpole_golden(n, seed))
        print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
