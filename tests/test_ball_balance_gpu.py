"""-m gpu parity tests: HIP BallBalance path (through the C ABI) vs the CPU oracle (pinned to tasks/ball_balance.py by
tests/test_oracle_pinned_ball_balance.py).  Bar: reset flags, reset ids, progress bit-exact; fp32 observations, rewards,
targets, reset rows within 1e-5 rel / 1e-6 abs."""
import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import ball_balance_oracle as bo
from oracle import philox

pytestmark = pytest.mark.gpu


def _loaded(n, seed=1234):
    env, task = make_task("BallBalance", n)
    st = workloads.ball_balance_state(n, task._env_pos.cpu(), device="cpu", seed=seed)
    if n > 8:
        st["progress"][:4] = torch.tensor([596, 597, 598, 599])
        st["reset"][:7] = 0
        st["ball_pos"][4:7, 2] = torch.tensor([0.1499, 0.1501, 0.16])
    workloads.load_ball_balance_state(task, st)
    return env, task, st


def _oracle_buffers(task) -> bo.BallBalanceBuffers:
    c = lambda t: t.detach().cpu().clone()
    return bo.BallBalanceBuffers(c(task.obs_buf), c(task.rew_buf), c(task.reset_buf), c(task.progress_buf), c(task.dof_position_targets))


def _spec(task) -> bo.BallBalanceSpec:
    return bo.BallBalanceSpec(max_episode_length=task._max_episode_length, dt=task._dt, action_speed_scale=task._action_speed_scale,
                              ball_radius=task._ball_radius, num_dof=task._balance_bots.num_dof,
                              actuated_dof_indices=[int(i) for i in task.actuated_dof_indices],
                              lower=task.bbot_dof_lower_limits.cpu(), upper=task.bbot_dof_upper_limits.cpu())


def _init(task) -> bo.BallBalanceInit:
    c = lambda t: t.detach().cpu().clone()
    return bo.BallBalanceInit(c(task.initial_ball_pos), c(task.initial_ball_rot), c(task.initial_ball_velocities), c(task.initial_bot_pos),
                              c(task.initial_bot_rot), c(task.initial_bot_velocities), c(task.initial_dof_positions))


@pytest.mark.parametrize("n", [1, 40, 64, 257, 4096])
@pytest.mark.parametrize("fused", [True, False])
def test_post_physics_matches_oracle(n, fused):
    env, task, st = _loaded(n)
    task.fused = fused
    spec, b = _spec(task), _oracle_buffers(task)
    bo.post_physics(spec, b, st["ball_pos"], st["ball_vel"], st["dof_pos"], st["dof_vel"], st["sensors"], task._env_pos.cpu())
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(reset, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(obs, b.obs_buf, "obs_buf")
    assert_close(rew, b.rew_buf, "rew_buf")
    if n > 8:
        assert [int(x) for x in reset[:7]] == [0, 0, 1, 1, 1, 0, 0]
    tiles = (n + 63) // 64
    pad = lambda t: torch.nn.functional.pad(t, (0, tiles * 64 - n)).view(tiles, 64).ne(0).sum(-1).int()
    assert_equal(task._reset_workspace[:tiles], pad(task.reset_buf.cpu()), "reset tile counts")


@pytest.mark.parametrize("n", [1, 64, 300, 5000])
@pytest.mark.parametrize("clamp", [False, True])
def test_sync_free_pre_physics_matches_oracle(n, clamp):
    env, task, st = _loaded(n)
    for v in (task._balls, task._balance_bots):
        v.apply_resets = False
    if n >= 64:
        task.reset_buf[: n // 3] = 1
    spec, b, init = _spec(task), _oracle_buffers(task), _init(task)
    seed, step = task._rand_seed, task._step_count + 1
    draw = lambda ids: torch.from_numpy(philox.uniforms(ids.numpy(), 8, seed, step))[:, :5].contiguous()
    actions = torch.rand(n, 3, generator=torch.Generator().manual_seed(5)) * 3 - 1.5
    task.clamp_actions_in_kernel = clamp
    want_actions = torch.clamp(actions, -task.clip_actions, task.clip_actions) if clamp else actions
    ids, rows = bo.pre_physics(spec, b, init, want_actions, draw)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    k = int(o["count"].item())
    assert k == len(ids)
    assert_equal(o["env_ids"][:k], ids, "env ids == reset_buf.nonzero()")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(task.dof_position_targets, b.dof_position_targets, "dof_position_targets")
    assert_close(task._balance_bots.joint_position_targets, b.dof_position_targets, "targets handed to the view")
    if rows is not None:
        for name in ("ball_pos", "ball_rot", "ball_vel", "bot_pos", "bot_rot", "bot_vel", "dof_pos"):
            assert_close(o[name][:k], rows[name], f"{name} rows")


def test_reference_flow_uses_torch_draws():
    n = 500
    env, task, st = _loaded(n)
    task.sync_free = False
    task.reset_buf[::7] = 1
    spec, b, init = _spec(task), _oracle_buffers(task), _init(task)
    actions = torch.rand(n, 3, generator=torch.Generator().manual_seed(6)) * 2 - 1
    torch.manual_seed(77)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(77)
    draw = lambda ids: torch.cat([torch.rand(len(ids), 1, device=task.device) for _ in range(5)], dim=-1).cpu()
    ids, rows = bo.pre_physics(spec, b, init, actions, draw)
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_close(task.dof_position_targets, b.dof_position_targets, "dof_position_targets")
    dev_ids = ids.to(task.device)
    assert_close(task._balls.positions[dev_ids], rows["ball_pos"], "ball positions of the reset envs")
    assert_close(task._balls.velocities[dev_ids], rows["ball_vel"], "ball velocities of the reset envs")


def test_full_size_properties():
    n = 1 << 20
    env, task, st = _loaded(n)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(obs[:, 9:12], st["ball_vel"][:, 0:3], "obs ball linvel")
    assert_equal(obs[:, 0:3], st["dof_pos"][:, [1, 3, 5]], "obs actuated dof pos")
    assert float(rew.min()) > 0.0 and float(rew.max()) <= 1.0
    want = reset.nonzero().squeeze(-1).clone()
    for v in (task._balls, task._balance_bots):
        v.apply_resets = False
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    assert int(o["count"].item()) == want.numel() and want.numel() > 0
    assert_equal(o["env_ids"][: want.numel()], want, "env ids == nonzero()")
    assert int(task.reset_buf.sum().item()) == 0
    assert float(task.dof_position_targets[want].abs().max()) == 0.0
