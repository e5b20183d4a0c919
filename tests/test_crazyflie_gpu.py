"""-m gpu parity tests: HIP Crazyflie path (through the C ABI) vs the CPU oracle (pinned to tasks/crazyflie.py by
tests/test_oracle_pinned_crazyflie.py).  Bar: reset flags, both id lists, progress bit-exact; fp32 observations, rewards,
episode sums, extras, thrusts, filter states, rotor speeds within 1e-5 rel / 1e-6 abs."""
import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import crazyflie_oracle as co

pytestmark = pytest.mark.gpu


def _loaded(n, seed=1234):
    env, task = make_task("Crazyflie", n)
    st = workloads.crazyflie_state(n, task._env_pos.cpu(), device="cpu", seed=seed)
    if n > 12:
        st["progress"][:4] = torch.tensor([696, 697, 698, 699])
        st["progress"][7:10] = torch.tensor([500, 0, 499])
        st["reset"][7:10] = 0
        st["root_pos"][4:7, 2] = torch.tensor([0.4999, 0.5001, 5.001])
        st["root_rot"][10] = torch.tensor([0.0, 1.0, 0.0, 0.0])
    workloads.load_crazyflie_state(task, st)
    task.thrusts.copy_(torch.randn(n, 4, 3, generator=torch.Generator().manual_seed(seed)).to(task.device))
    return env, task, st


def _oracle_buffers(task) -> co.CrazyflieBuffers:
    c = lambda t: t.detach().cpu().clone()
    return co.CrazyflieBuffers(c(task.obs_buf), c(task.rew_buf), c(task.reset_buf), c(task.progress_buf), c(task.target_positions),
                               c(task.actions), c(task.thrusts), c(task.thrust_cmds_damp), c(task.thrust_rot_damp), c(task.dof_pos),
                               c(task.dof_vel), {k: c(task.episode_sums[k]) for k in co.EPISODE_KEYS})


def _spec(task) -> co.CrazyflieSpec:
    return co.CrazyflieSpec(max_episode_length=task._max_episode_length, dt=task.dt)


@pytest.mark.parametrize("n", [1, 40, 64, 257, 4096])
@pytest.mark.parametrize("fused", [True, False])
def test_post_physics_matches_oracle(n, fused):
    env, task, st = _loaded(n)
    task.fused = fused
    spec, b = _spec(task), _oracle_buffers(task)
    env_pos = task._env_pos.cpu()
    co.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], env_pos)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    rp = st["root_pos"] - env_pos
    td = torch.sqrt(torch.square(b.target_positions - rp).sum(-1))
    ok = ((td - 5.0).abs() > 1e-5) & ((rp[:, 2] - 0.5).abs() > 1e-6) & ((rp[:, 2] - 5.0).abs() > 1e-5) & (b.obs_buf[:, 11].abs() > 1e-6)
    assert_equal(reset[ok], b.reset_buf[ok], "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(obs, b.obs_buf, "obs_buf")
    assert_close(rew, b.rew_buf, "rew_buf")
    for k in co.EPISODE_KEYS:
        assert_close(task.episode_sums[k], b.episode_sums[k], f"episode_sums[{k}]")
    if n > 12:
        assert int(reset[10]) == 1 and int(reset[4]) == 1 and int(reset[6]) == 1 and [int(x) for x in reset[2:4]] == [1, 1]


@pytest.mark.parametrize("n", [1, 64, 300, 5000])
@pytest.mark.parametrize("clamp", [False, True])
def test_sync_free_pre_physics_matches_oracle(n, clamp):
    env, task, st = _loaded(n)
    for v in (task._copters, task._balls):
        v.apply_resets = False
    if n >= 64:
        task.reset_buf[: n // 3] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    actions = torch.rand(n, 4, generator=torch.Generator().manual_seed(5)) * 3 - 1.5
    task.clamp_actions_in_kernel = clamp
    want_actions = torch.clamp(actions, -task.clip_actions, task.clip_actions) if clamp else actions
    torch.manual_seed(99)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(99)
    noise4 = (0.01 * torch.randn(4, dtype=torch.float32, device=task.device)).cpu()
    ids, rows, tids, trows = co.pre_physics(spec, b, want_actions, st["root_rot"], task._env_pos.cpu(), task.initial_root_pos.cpu(),
                                            task.initial_root_rot.cpu(), task.initial_ball_rot.cpu(), noise4)
    o = task._reset_out
    k, m = int(o["count"].item()), int(o["target_count"].item())
    assert k == len(ids) and m == len(tids)
    assert_equal(o["env_ids"][:k], ids, "env ids == reset_buf.nonzero()")
    assert_equal(o["target_ids"][:m], tids, "target ids == (progress % 500 == 0).nonzero()")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_equal(task.actions, b.actions, "actions")
    assert_close(task.thrust_rot_damp, b.thrust_rot_damp, "thrust_rot_damp")
    assert_close(task.thrust_cmds_damp, b.thrust_cmds_damp, "thrust_cmds_damp")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.dof_vel, b.dof_vel, "dof_vel (rotor speeds)")
    assert_close(task.dof_pos, b.dof_pos, "dof_pos")
    assert_close(task.target_positions, b.target_positions, "target_positions")
    assert_close(task._copters.physics_rotors[3].applied_forces, b.thrusts[:, 3], "forces handed to rotor 3")
    for key in co.EPISODE_KEYS:
        assert_close(task.episode_sums[key], b.episode_sums[key], f"episode_sums[{key}]")
        if rows is not None:
            assert_close(task.extras["episode"][key].reshape(1), b.extras["episode"][key].reshape(1), f"extras[{key}]")
    if rows is not None:
        for name in ("dof_pos", "dof_vel", "root_pos", "root_rot", "root_vel"):
            assert_close(o[name][:k], rows[name], f"{name} rows")
    if trows is not None:
        assert_close(o["ball_pos"][:m], trows["ball_pos"], "ball_pos rows")
        assert_equal(o["ball_rot"][:m], trows["ball_rot"], "ball_rot rows")


def test_reference_flow_matches_oracle_and_keeps_torch_rng_in_step():
    n = 500
    env, task, st = _loaded(n)
    task.sync_free = False
    task.reset_buf[::7] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    k = int((b.reset_buf != 0).sum())
    actions = torch.rand(n, 4, generator=torch.Generator().manual_seed(6)) * 2 - 1
    torch.manual_seed(77)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(77)
    dev = task.device
    torch.rand(k, 4, device=dev), torch.rand(k, 1, device=dev), torch.rand(k, 1, device=dev), torch.rand(k, 1, device=dev)
    noise4 = (0.01 * torch.randn(4, dtype=torch.float32, device=dev)).cpu()
    ids, rows, tids, trows = co.pre_physics(spec, b, actions, st["root_rot"], task._env_pos.cpu(), task.initial_root_pos.cpu(),
                                            task.initial_root_rot.cpu(), task.initial_ball_rot.cpu(), noise4)
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.thrust_cmds_damp, b.thrust_cmds_damp, "thrust_cmds_damp")
    assert_close(task.dof_vel, b.dof_vel, "dof_vel")
    for key in co.EPISODE_KEYS:
        assert_close(task.episode_sums[key], b.episode_sums[key], f"episode_sums[{key}]")
        assert_close(task.extras["episode"][key].reshape(1), b.extras["episode"][key].reshape(1), f"extras[{key}]")


def test_full_size_properties():
    n = 1 << 20
    env, task, st = _loaded(n)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(obs[:, 12:18], st["root_vel"], "obs velocities")
    want = reset.nonzero().squeeze(-1).clone()
    want_t = ((task.progress_buf % 500 == 0) | (reset != 0)).nonzero().squeeze(-1).clone()
    sums = task._episode_sums.clone()
    for v in (task._copters, task._balls):
        v.apply_resets = False
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    assert int(o["count"].item()) == want.numel() and want.numel() > 0
    assert int(o["target_count"].item()) == want_t.numel()
    assert_equal(o["env_ids"][: want.numel()], want, "env ids == nonzero()")
    assert int(task.reset_buf.sum().item()) == 0
    assert float(task.thrusts[want].abs().max()) == 0.0 and float(task._episode_sums[:, want].abs().max()) == 0.0
    mean0 = sums[0, want].double().mean().item() / task._max_episode_length
    assert abs(task.extras["episode"]["rew_pos"].item() - mean0) <= 1e-5 * abs(mean0) + 1e-7
