"""-m gpu parity tests: HIP AnymalTerrain path (through the C ABI) vs the CPU oracle.
Bar (north_star): reset / timeout / has_fallen flags, reset ids, progress, terrain levels bit-exact;
fp32 observations, rewards, commands, episode sums, extras within 1e-5 rel / 1e-6 abs."""
import numpy as np
import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import anymal_terrain_oracle as ao
from oracle import philox

pytestmark = pytest.mark.gpu


def _spec(task) -> ao.AnymalSpec:
    return ao.AnymalSpec(curriculum=bool(task.curriculum), add_noise=bool(task.add_noise))


def _loaded(n, seed=1234, **over):
    env, task = make_task("AnymalTerrain", n, **over)
    st = workloads.anymal_state(n, task.terrain_origins.cpu(), task.default_dof_pos.cpu(), device="cpu", seed=seed)
    if n > 8:  # scan points that clip at the map border / negative coordinates; progress straddling the timeout
        st["base_pos"][0] = torch.tensor([-20.3, -19.95, 0.5])
        st["base_pos"][1] = torch.tensor([99.9, 180.2, 0.5])
        st["base_pos"][2] = torch.tensor([-25.0, 300.0, 0.5])
        st["progress"][3:7] = torch.tensor([997, 998, 999, 1000])
    workloads.load_anymal_state(task, st)
    return env, task, st


def _oracle_inputs(task, st):
    s = ao.AnymalState(st["base_pos"].clone(), st["base_quat"].clone(), st["base_vel"].clone(), st["dof_pos"].clone(),
                       st["dof_vel"].clone(), st["knee_contact"].clone(), st["base_contact"].clone())
    n = st["base_pos"].shape[0]
    b = ao.AnymalBuffers(
        obs_buf=torch.zeros(n, 188), rew_buf=torch.zeros(n), reset_buf=torch.ones(n, dtype=torch.long),
        timeout_buf=torch.zeros(n, dtype=torch.long), progress_buf=st["progress"].clone(), commands=st["commands"].clone(),
        actions=st["actions"].clone(), last_actions=st["last_actions"].clone(), last_dof_vel=st["last_dof_vel"].clone(),
        torques=st["torques"].clone(), feet_air_time=st["feet_air_time"].clone(),
        episode_sums={k: st["episode_sums"][i].clone() for i, k in enumerate(ao.EPISODE_KEYS)},
        terrain_levels=st["terrain_levels"].clone(), terrain_types=st["terrain_types"].clone(),
        env_origins=st["env_origins"].clone(), default_dof_pos=task.default_dof_pos.cpu().clone())
    return s, b


def _philox_draws(seed, step):
    def draw(ids):
        u = torch.from_numpy(philox.uniforms(ids.numpy(), 29, seed, step, stream=1))
        return ao.AnymalResetDraws(u[:, 0:12], u[:, 12:24], u[:, 24:26], u[:, 26:27], u[:, 27:28], u[:, 28:29])
    return draw


def _philox_noise(n, seed, step):
    """U[0,1) the kernel draws for obs element i: word i % 4 of block (i // 4) on stream 3."""
    gf = np.arange(n * 188 // 4, dtype=np.uint64)
    w = philox.philox4x32_10(gf & philox.MASK, (gf >> np.uint64(32)) | np.uint64(3 << 16), np.uint64(step & 0xFFFFFFFF),
                             np.uint64(step >> 32), seed & 0xFFFFFFFF, (seed >> 32) & 0xFFFFFFFF)
    u = np.stack([philox.u01(x) for x in w], axis=-1).reshape(n, 188)
    return torch.from_numpy(u)


def _compare(task, b, env_ids, view, n):
    o = task._reset_out
    k = int(o["count"].item())
    assert k == len(env_ids)
    assert_equal(o["env_ids"][:k], env_ids, "env_ids == reset_buf.nonzero()")
    assert task.reset_buf.dtype == torch.bool and b.reset_buf.dtype == torch.bool
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.timeout_buf, b.timeout_buf, "timeout_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_equal(task.terrain_levels, b.terrain_levels, "terrain_levels")
    assert_equal(task.env_origins, b.env_origins, "env_origins")
    assert_equal(task.feet_air_time, b.feet_air_time, "feet_air_time")
    assert_equal(task.last_actions, b.last_actions, "last_actions")
    assert_close(task.last_dof_vel, b.last_dof_vel, "last_dof_vel")
    assert_close(task.commands, b.commands, "commands")
    assert_close(task.rew_buf, b.rew_buf, "rew_buf")
    assert_close(task.obs_buf, b.obs_buf, "obs_buf")
    for i, key in enumerate(ao.EPISODE_KEYS):
        assert_close(task.episode_sums[key], b.episode_sums[key], "episode_sums." + key)
    if k > 0:
        for key in ao.EPISODE_KEYS:
            assert_close(task.extras["episode"]["rew_" + key].reshape(1), b.extras["episode"]["rew_" + key].reshape(1),
                         "extras.rew_" + key)
        assert_close(task.extras["episode"]["terrain_level"].reshape(1), b.extras["episode"]["terrain_level"].reshape(1),
                     "extras.terrain_level")
        assert_close(o["base_pos"][:k], view["base_pos"][env_ids], "reset rows base_pos")
        assert_equal(o["base_quat"][:k], view["base_quat"][env_ids], "reset rows base_quat")
        assert_equal(o["base_vel"][:k], view["base_velocities"][env_ids], "reset rows base_vel")
        assert_close(o["dof_pos"][:k], view["dof_pos"][env_ids], "reset rows dof_pos")
        assert_close(o["dof_vel"][:k], view["dof_vel"][env_ids], "reset rows dof_vel")


@pytest.mark.parametrize("n", [1, 40, 64, 257, 4096])
@pytest.mark.parametrize("noise", ["philox", "given", "off"])
def test_post_physics_matches_oracle(n, noise):
    env, task, st = _loaded(n)
    task._anymals.apply_resets = False  # keep the view buffers as loaded; the rows are compared instead
    task.add_noise = noise != "off"
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    seed, step = task._rand_seed, task._step_count + 1
    given = torch.rand(n, 188, generator=torch.Generator().manual_seed(9)) if noise == "given" else None
    noise_u = given if noise == "given" else (_philox_noise(n, seed, step) if noise == "philox" else None)
    env_ids, view = ao.post_physics(spec, b, s, task.height_samples.cpu(), task.terrain_origins.cpu(),
                                    draw=_philox_draws(seed, step), noise_u=noise_u)
    task.post_physics_step(noise_u=given.to(task.device) if given is not None else None)
    torch.cuda.synchronize()
    knees = (torch.norm(st["knee_contact"].view(n, 4, 3), dim=-1) > 1.0).sum(-1) > 1.0
    assert_equal(task.has_fallen, (torch.norm(st["base_contact"], dim=1) > 1.0) | knees, "has_fallen")
    _compare(task, b, env_ids, view, n)


@pytest.mark.parametrize("noise", ["philox", "off"])
def test_height_scan_keeps_nan_and_infinite_base_heights(noise):
    """heights = clip(z - 0.5 - measured, -1, 1) * scale (:537-539): torch.clip keeps a NaN and maps +-inf to +-1.  The scan kernel
    takes the clip as one median instruction, which would return a bound for a NaN, and repairs NaN rows on a path of its own,
    chosen per wave: envs with such a base height sit in the first, a middle and the last (ragged) workgroup here."""
    n = 300
    env, task = make_task("AnymalTerrain", n)
    st = workloads.anymal_state(n, task.terrain_origins.cpu(), task.default_dof_pos.cpu(), device="cpu", seed=77)
    nan, inf = float("nan"), float("inf")
    for i, z in ((0, nan), (5, inf), (6, -inf), (63, nan), (64, nan), (130, inf), (131, nan), (288, nan), (299, -inf)):
        st["base_pos"][i, 2] = z
    workloads.load_anymal_state(task, st)
    task._anymals.apply_resets = False
    task.add_noise = noise != "off"
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    seed, step = task._rand_seed, task._step_count + 1
    ao.post_physics(spec, b, s, task.height_samples.cpu(), task.terrain_origins.cpu(), draw=_philox_draws(seed, step),
                    noise_u=_philox_noise(n, seed, step) if noise == "philox" else None)
    task.post_physics_step()
    torch.cuda.synchronize()
    got, want = task.obs_buf.cpu(), b.obs_buf
    # (an env that resets in this step is measured at its reset pose, :421-437: its heights are numbers again)
    stay = ~b.reset_buf
    nan_rows = [i for i in (0, 63, 64, 131, 288) if stay[i]]
    inf_rows = [i for i in (5, 6, 130, 299) if stay[i]]
    assert len(nan_rows) >= 3 and len(inf_rows) >= 2, "the seed keeps enough of the marked envs out of the reset"
    assert want[nan_rows][:, 36:176].isnan().all(), "the oracle keeps NaN heights (scan columns 36:176)"
    assert not want[inf_rows][:, 36:176].isnan().any()
    if noise == "off":
        assert (want[inf_rows][:, 36:176].abs() == spec.height_meas_scale).all()
    assert_equal(got.isnan(), want.isnan(), "NaN pattern of obs_buf")
    assert_close(torch.nan_to_num(got, nan=0.0), torch.nan_to_num(want, nan=0.0), "obs_buf where it is a number")
    assert_equal(task.rew_buf.cpu().isnan(), b.rew_buf.isnan(), "NaN pattern of rew_buf")
    assert_close(torch.nan_to_num(task.rew_buf.cpu(), nan=0.0), torch.nan_to_num(b.rew_buf, nan=0.0), "rew_buf where it is a number")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")


def test_second_step_uses_the_reset_rows_and_new_draws():
    """Two steps in a row with the setters applied: the second step starts from the first one's reset rows."""
    n = 300
    env, task, st = _loaded(n)
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    hs, origins = task.height_samples.cpu(), task.terrain_origins.cpu()
    seed = task._rand_seed
    for it in range(2):
        step = task._step_count + 1
        env_ids, view = ao.post_physics(spec, b, s, hs, origins, draw=_philox_draws(seed, step),
                                        noise_u=_philox_noise(n, seed, step))
        task.post_physics_step()
        torch.cuda.synchronize()
        _compare(task, b, env_ids, view, n)
        # the oracle's view copies are what PhysX would now hold
        s = ao.AnymalState(view["base_pos"], view["base_quat"], view["base_velocities"], view["dof_pos"], view["dof_vel"],
                           s.knee_contact, s.base_contact)
        assert_close(task._anymals.joint_positions, view["dof_pos"], "view dof_pos after setters")
        assert_close(task._anymals.positions, view["base_pos"], "view base_pos after setters")


def test_reset_idx_uses_torch_draws_like_the_reference():
    n = 500
    env, task, st = _loaded(n)
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    b.reset_buf = torch.zeros(n, dtype=torch.bool)
    task.reset_buf = torch.zeros(n, dtype=torch.bool, device=task.device)
    ids = torch.tensor([0, 1, 63, 64, 255, 256, 499])
    torch.manual_seed(77)
    task.reset_idx(ids.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(77)
    k, dev = len(ids), task.device
    d = ao.AnymalResetDraws(*(torch.rand(k, w, device=dev).cpu() for w in (12, 12, 2, 1, 1, 1)))
    view = dict(base_pos=s.base_pos.clone(), base_quat=s.base_quat.clone(), base_velocities=s.base_velocities.clone(),
                dof_pos=s.dof_pos.clone(), dof_vel=s.dof_vel.clone())
    ao.reset_idx(spec, b, view, ids, d, task.terrain_origins.cpu())
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_equal(task.terrain_levels, b.terrain_levels, "terrain_levels")
    assert_equal(task.env_origins, b.env_origins, "env_origins")
    assert_close(task.commands, b.commands, "commands")
    assert_equal(task.last_actions, b.last_actions, "last_actions")
    assert_equal(task.last_dof_vel, b.last_dof_vel, "last_dof_vel")
    assert_equal(task.feet_air_time, b.feet_air_time, "feet_air_time")
    for key in ao.EPISODE_KEYS:
        assert_equal(task.episode_sums[key], b.episode_sums[key], "episode_sums." + key)
        assert_close(task.extras["episode"]["rew_" + key].reshape(1), b.extras["episode"]["rew_" + key].reshape(1), "extras." + key)
    assert_close(task._anymals.joint_positions, view["dof_pos"], "view dof_pos")
    assert_close(task._anymals.joint_velocities, view["dof_vel"], "view dof_vel")
    assert_close(task._anymals.positions, view["base_pos"], "view base_pos")
    assert_equal(task._anymals.orientations, view["base_quat"], "view base_quat")
    assert_equal(task._anymals.velocities, view["base_velocities"], "view base_velocities")


def test_post_reset_state_matches_reference_semantics():
    env, task = make_task("AnymalTerrain", 200)
    assert task.init_done and int(task.progress_buf.sum().item()) == 0
    assert_equal(task.env_origins, task.terrain_origins[task.terrain_levels, task.terrain_types], "env_origins (init: no curriculum move)")
    dp = task._anymals.joint_positions / task.default_dof_pos
    assert bool(((dp >= 0.5) & (dp <= 1.5)).all())
    assert bool((task._anymals.orientations == torch.tensor([1.0, 0, 0, 0], device=task.device)).all())
    off = (task._anymals.positions - task.env_origins)[:, :2]
    assert bool((off.abs() <= 0.5).all()) and bool(((task._anymals.positions - task.env_origins)[:, 2] - 0.62).abs().max() < 1e-6)


@pytest.mark.parametrize("n", [1, 64, 1000])
def test_pd_torques_match_oracle(n):
    env, task, st = _loaded(n)
    spec = _spec(task)
    task.decimation = 1
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    want = ao.pd_torques(spec, st["actions"], task.default_dof_pos.cpu(), st["dof_pos"], st["dof_vel"])
    assert_close(task.torques, want, "torques")
    assert_equal(task._anymals.joint_efforts, task.torques.cpu(), "efforts handed to the view")


def test_full_size_properties():
    """C5 per-GPU size (2M envs / 8 GPUs = 262144): pass-through columns, reset ids == nonzero(), flags' closed forms."""
    n = 1 << 18
    env, task, st = _loaded(n)
    task._anymals.apply_resets = False
    task.post_physics_step()
    torch.cuda.synchronize()
    obs = task.obs_buf
    assert_equal(obs[:, 176:188], st["actions"], "obs actions")
    assert bool((obs[:, 36:176].abs() <= 5.0 + 0.06 * 5.0 + 1e-6).all())
    assert_equal(task.last_actions, st["actions"], "last_actions")
    timeout = st["progress"] + 1 >= 999
    assert_equal(task.timeout_buf, timeout.long(), "timeout_buf")
    knees = (st["knee_contact"].view(n, 4, 3).norm(dim=-1) > 1.0).sum(-1) > 1
    fallen = (st["base_contact"].norm(dim=-1) > 1.0) | knees
    assert_equal(task.reset_buf, fallen | timeout, "reset_buf closed form")
    ids = task.reset_buf.nonzero().flatten()
    o = task._reset_out
    assert int(o["count"].item()) == ids.numel()
    assert_equal(o["env_ids"][: ids.numel()], ids, "env_ids == nonzero()")
    assert int(task.progress_buf[ids].sum().item()) == 0
    assert_equal(task.progress_buf[~task.reset_buf], (st["progress"] + 1)[~task.reset_buf.cpu()], "progress of live envs")
    assert bool((task.rew_buf >= 0).all())


def test_full_size_matches_the_oracle():
    """C5's per-GPU size (2 M envs / 8 GPUs = 262144) against the ORACLE, not only closed forms -- the whole step
    (flags, ids, curriculum with its whole-matrix command norm over all ~29 000 reset envs, the 140-point height scan with its
    Philox noise, reward, episode sums, extras means, reset rows).  The oracle takes a few seconds at this size, so every
    row is compared, not a slice: the curriculum norm and the extras means couple all the envs of a step."""
    n = 1 << 18
    env, task, st = _loaded(n)
    task._anymals.apply_resets = False
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    seed, step = task._rand_seed, task._step_count + 1
    env_ids, view = ao.post_physics(spec, b, s, task.height_samples.cpu(), task.terrain_origins.cpu(),
                                    draw=_philox_draws(seed, step), noise_u=_philox_noise(n, seed, step))
    task.post_physics_step()
    torch.cuda.synchronize()
    assert 0.05 * n < len(env_ids) < 0.3 * n
    _compare(task, b, env_ids, view, n)
    lv = task.terrain_levels.cpu()[env_ids]
    assert len(torch.unique(lv)) > 3  # the curriculum really moved envs in both directions


def _task_tensors(task):
    out = {k: v for k, v in vars(task).items() if isinstance(v, torch.Tensor) and k not in ("_step_dev", "base_lin_vel", "base_ang_vel",
                                                                                           "projected_gravity", "_reset_mask")}
    out.update({f"view.{k}": v for k, v in vars(task._anymals).items() if isinstance(v, torch.Tensor)})
    out.update({f"extras.{k}": v for k, v in task.extras.get("episode", {}).items()})
    return out


def test_split_stage_methods_match_fused():
    """check_termination / calculate_metrics / get_observations (and the inline preamble, compute_body_frame)
    are callable on their own, as in the reference (anymal_terrain.py:409-466, 468-552); their sequence -- with reset_idx(env_ids)
    in the middle and the reference's two elementwise closing lines -- leaves every tensor of the task and of its views
    BIT-identical to the fused step, over three consecutive steps with resets in each."""
    from omniisaacgymenvs_amd import _lib

    n = 700
    (_, fused, st), (_, split, _) = _loaded(n), _loaded(n)
    fused.materialize_frame = True
    noise = torch.rand(n, 188, generator=torch.Generator().manual_seed(4)).to(fused.device)
    seed = fused._rand_seed
    total = 0
    for it in range(3):
        fused.post_physics_step(noise_u=noise)
        t = split
        t.refresh_dof_state_tensors()
        t.refresh_body_state_tensors()
        t._advance_step_counter()
        step = t._step_count
        t._stages(_lib.ANYMAL_STAGE_PROGRESS)          # :410
        t.compute_body_frame()                          # :421-427
        t.check_termination()                           # :429
        t.calculate_metrics()                           # :431
        ids = t.reset_buf.nonzero(as_tuple=False).flatten()
        total += len(ids)
        if len(ids) > 0:                                # :433-435, with the draws the fused kernel takes from Philox stream 1
            d = _philox_draws(seed, step)(ids.cpu())
            t.reset_idx(ids, draws=(d.dof_offset, d.dof_vel, d.xy, torch.cat([d.cmd_x, d.cmd_y, d.cmd_yaw], dim=-1)))
        t.get_observations()                            # :437
        t.obs_buf += (2 * noise - 1) * t.noise_scale_vec  # :438-439
        t.last_actions[:] = t.actions[:]                # :441
        t.last_dof_vel[:] = t.dof_vel[:]                # :442
        torch.cuda.synchronize()
        A, B = _task_tensors(fused), _task_tensors(split)
        assert set(A) == set(B)
        for k in A:
            assert_equal(B[k], A[k], f"step {it}: {k}")
        for k in ("base_lin_vel", "base_ang_vel", "projected_gravity"):
            assert_equal(getattr(split, k), getattr(fused, k), f"step {it}: {k}")
    assert total > 50


def test_unfused_flow_and_update_terrain_level():
    """task.fused = False runs the reference's own sequence through the stage methods (torch draws); update_terrain_level(env_ids)
    on its own equals the oracle's (:369-380)."""
    n = 300
    env, task, st = _loaded(n)
    task.fused = False
    task.post_physics_step()
    torch.cuda.synchronize()
    assert task.reset_buf.dtype == torch.bool and bool(torch.isfinite(task.obs_buf).all())
    assert int(task.progress_buf[task.reset_buf].sum()) == 0
    assert_equal(task.progress_buf[~task.reset_buf], (st["progress"] + 1)[~task.reset_buf.cpu()], "progress of live envs")
    # update_terrain_level on its own
    env, task, st = _loaded(n)
    ids = torch.arange(0, n, 3)
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    want_levels = b.terrain_levels.clone()
    # anymal_terrain.py:374-379 (the oracle holds these lines inside its reset_idx)
    distance = torch.norm(s.base_pos[ids, :2] - b.env_origins[ids, :2], dim=1)
    b.terrain_levels[ids] -= 1 * (distance < torch.norm(b.commands[ids, :2]) * spec.max_episode_length_s * 0.25)
    b.terrain_levels[ids] += 1 * (distance > spec.env_length / 2)
    b.terrain_levels[ids] = torch.clip(b.terrain_levels[ids], 0) % spec.env_rows
    b.env_origins[ids] = task.terrain_origins.cpu()[b.terrain_levels[ids], b.terrain_types[ids]]
    task.update_terrain_level(ids.to(task.device))
    torch.cuda.synchronize()
    assert_equal(task.terrain_levels, b.terrain_levels, "terrain_levels")
    assert_equal(task.env_origins, b.env_origins, "env_origins")
    assert not torch.equal(b.terrain_levels, want_levels)


def test_device_side_push_robots_over_several_steps():
    """push_robots (:419-420, :446-450) inside the post kernel: on steps where common_step_counter % push_interval == 0 the
    base xy velocities of EVERY env become U(-1,1) (Philox stream 7) before anything reads them, so that step's
    observations and reward already see the push; on the other steps nothing is touched and the setter's count is 0."""
    n = 300
    env, task, st = _loaded(n)
    assert task.device_push and task.graph_capturable is True
    task.push_interval = 3
    task.common_step_counter = 1  # so that the 2nd and the 5th step push
    spec = _spec(task)
    s, b = _oracle_inputs(task, st)
    hs, origins = task.height_samples.cpu(), task.terrain_origins.cpu()
    seed = task._rand_seed
    pushes = []
    for it in range(5):
        step = task._step_count + 1
        pushing = (task.common_step_counter + 1) % 3 == 0
        pushes.append(pushing)
        if pushing:
            u = torch.from_numpy(philox.uniforms(np.arange(n), 2, seed, step, stream=7))
            s.base_velocities[:, 0:2] = 2.0 * u + (-1.0)
        env_ids, view = ao.post_physics(spec, b, s, hs, origins, draw=_philox_draws(seed, step), noise_u=_philox_noise(n, seed, step))
        task.post_physics_step()
        torch.cuda.synchronize()
        assert int(task._push_count) == (n if pushing else 0)
        _compare(task, b, env_ids, view, n)
        assert_close(task._anymals.velocities, view["base_velocities"], "view velocities after push + reset setters", rtol=0, atol=0)
        s = ao.AnymalState(view["base_pos"], view["base_quat"], view["base_velocities"], view["dof_pos"], view["dof_vel"],
                           s.knee_contact, s.base_contact)
    assert pushes == [False, True, False, False, True]


def test_host_side_push_robots_is_still_there():
    """device_push = False: the reference's host branch with torch's own draw (and no graph capture)."""
    n = 64
    env, task, st = _loaded(n)
    task.device_push = False
    assert task.graph_capturable is not True
    task.push_interval = 2
    torch.manual_seed(3)
    want = 2.0 * torch.rand(n, 2, device=task.device) + (-1.0)
    torch.manual_seed(3)
    before = task._anymals.velocities.clone()
    task.post_physics_step()  # common_step_counter 1: no push
    task._anymals.velocities.copy_(before)
    task.post_physics_step()  # 2: push
    torch.cuda.synchronize()
    assert task.common_step_counter == 2
    resets = task.reset_buf.cpu()
    keep = ~resets  # envs that reset this step got their base velocity rows from reset_idx afterwards
    assert_close(task._anymals.velocities[:, 0:2].cpu()[keep], want.cpu()[keep], "pushed xy velocities", rtol=0, atol=0)


def test_height_min_table_is_the_reference_minimum_and_changes_nothing():
    """oige_anymal_height_min tabulates min(h[px, py], h[px + 1, py + 1]) (:598-600) once per map; the step with the table
    (one gather per scan point) is bit-identical to the step with the reference's two gathers."""
    n = 700
    outs = []
    for use in (True, False):
        env, task, st = _loaded(n)
        task.use_height_min = use
        task.post_physics_step()
        torch.cuda.synchronize()
        outs.append((task.obs_buf.clone(), task.rew_buf.clone(), task.reset_buf.clone()))
        if use:
            h = task.height_samples
            want = h.clone()
            want[:-1, :-1] = torch.minimum(h[:-1, :-1], h[1:, 1:])
            assert_equal(task._height_min, want, "height_min table")
            assert task._height_min_table() is task._height_min          # cached
            task.height_samples = h.clone()                              # a rebound map gets a new table
            assert task._height_min_table() is not None and task._height_min_key[0] is task.height_samples
    for a, b, what in zip(outs[0], outs[1], ("obs_buf", "rew_buf", "reset_buf")):
        assert_equal(a, b, what)


@pytest.mark.parametrize("n", [257, 5000, 262144])
def test_step_without_the_scan_launch_equals_the_step_with_it(n):
    """With oige_anymal_buffers.tile_counts_u8 the head kernel's waves derive their offset in the id list and the command norm
    themselves (no scan + norm launch); NULL keeps the launch.  Same ids, rows, terrain levels and observations, two steps."""
    outs = []
    for with_u8 in (True, False):
        env, task, st = _loaded(n)
        task._anymals.apply_resets = False
        if not with_u8:
            inner = task._buffers

            def no_u8(inner=inner):
                b = type(inner()).from_buffer_copy(inner())  # a copy: the mirror keeps (and may re-check) its own block
                b.tile_counts_u8 = None
                return b

            task._buffers = no_u8
        got = []
        for _ in range(2):
            task.post_physics_step()
            torch.cuda.synchronize()
            o = task._reset_out
            k = int(o["count"].item())
            assert k > 0
            got += [o["env_ids"][:k].clone(), o["base_pos"][:k].clone(), o["dof_pos"][:k].clone(), task.terrain_levels.clone(),
                    task.env_origins.clone(), task.commands.clone(), task.obs_buf.clone(), task.rew_buf.clone(),
                    task.reset_buf.clone(), task.progress_buf.clone(), task.extras["episode"]["terrain_level"].clone()]
        outs.append(got)
    for x, y in zip(*outs):
        assert torch.equal(x, y)
