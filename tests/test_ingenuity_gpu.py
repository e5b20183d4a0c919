"""-m gpu parity tests: HIP Ingenuity path (through the C ABI) vs the CPU oracle (pinned to tasks/ingenuity.py by
tests/test_oracle_pinned_ingenuity.py).  Bar: reset flags, both id lists, progress bit-exact; fp32 observations,
rewards, thrusts, targets, reset rows within 1e-5 rel / 1e-6 abs."""
import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import ingenuity_oracle as io
from oracle import philox

pytestmark = pytest.mark.gpu


def _loaded(n, seed=1234):
    env, task = make_task("Ingenuity", n)
    st = workloads.ingenuity_state(n, task._env_pos.cpu(), device="cpu", seed=seed)
    if n > 12:
        st["progress"][:4] = torch.tensor([1996, 1997, 1998, 1999])
        st["progress"][7:10] = torch.tensor([500, 1000, 499])
        st["reset"][7:10] = 0
        st["root_pos"][4:7, 2] = torch.tensor([0.4999, 0.5001, 0.51])
        st["root_pos"][10] = task._env_pos.cpu()[10] + torch.tensor([25.0, 0.0, 1.0])
    workloads.load_ingenuity_state(task, st)
    task.thrusts.copy_(torch.randn(n, 2, 3, generator=torch.Generator().manual_seed(seed)).to(task.device))
    return env, task, st


def _oracle_buffers(task) -> io.IngenuityBuffers:
    c = lambda t: t.detach().cpu().clone()
    return io.IngenuityBuffers(c(task.obs_buf), c(task.rew_buf), c(task.reset_buf), c(task.progress_buf), c(task.target_positions),
                               c(task.thrusts), c(task.dof_pos), c(task.dof_vel))


def _spec(task) -> io.IngenuitySpec:
    return io.IngenuitySpec(max_episode_length=task._max_episode_length, dt=task.dt, thrust_limit=task.thrust_limit,
                            thrust_lateral_component=task.thrust_lateral_component)


@pytest.mark.parametrize("n", [1, 40, 64, 257, 4096])
@pytest.mark.parametrize("fused", [True, False])
def test_post_physics_matches_oracle(n, fused):
    env, task, st = _loaded(n)
    task.fused = fused
    spec, b = _spec(task), _oracle_buffers(task)
    env_pos = task._env_pos.cpu()
    io.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], env_pos)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    rp = st["root_pos"] - env_pos
    td = torch.sqrt(torch.square(b.target_positions - rp).sum(-1))
    ok = ((td - 20.0).abs() > 1e-5) & ((rp[:, 2] - 0.5).abs() > 1e-6)
    assert_equal(reset[ok], b.reset_buf[ok], "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(obs, b.obs_buf, "obs_buf")
    assert_close(rew, b.rew_buf, "rew_buf")
    if n > 12:
        assert int(reset[10]) == 1 and int(reset[4]) == 1 and [int(x) for x in reset[2:4]] == [1, 1]


@pytest.mark.parametrize("n", [1, 64, 300, 5000])
@pytest.mark.parametrize("clamp", [False, True])
def test_sync_free_pre_physics_matches_oracle(n, clamp):
    env, task, st = _loaded(n)
    for v in (task._copters, task._balls):
        v.apply_resets = False
    if n >= 64:
        task.reset_buf[: n // 3] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    seed, step = task._rand_seed, task._step_count + 1
    uni = lambda ids: torch.from_numpy(philox.uniforms(ids.numpy(), 12, seed, step))
    draw_reset = lambda ids: uni(ids)[:, 0:5].contiguous()
    draw_targets = lambda ids: (uni(ids)[:, 8:10].contiguous(), uni(ids)[:, 10].contiguous())
    actions = torch.rand(n, 6, generator=torch.Generator().manual_seed(5)) * 3 - 1.5
    task.clamp_actions_in_kernel = clamp
    want_actions = torch.clamp(actions, -task.clip_actions, task.clip_actions) if clamp else actions
    ids, rows, tids, trows = io.pre_physics(spec, b, want_actions, task._env_pos.cpu(), task.initial_root_pos.cpu(),
                                            task.initial_root_rot.cpu(), task.initial_ball_rot.cpu(), draw_reset, draw_targets)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    k, m = int(o["count"].item()), int(o["target_count"].item())
    assert k == len(ids) and m == len(tids)
    assert_equal(o["env_ids"][:k], ids, "env ids == reset_buf.nonzero()")
    assert_equal(o["target_ids"][:m], tids, "target ids == (progress % 500 == 0).nonzero()")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.target_positions, b.target_positions, "target_positions")
    assert_close(task.dof_pos, b.dof_pos, "dof_pos")
    assert_equal(task.dof_vel, b.dof_vel, "dof_vel")
    assert_close(task._copters.physics_rotors[1].applied_forces, b.thrusts[:, 1], "forces handed to rotor 1")
    if rows is not None:
        for name in ("dof_pos", "dof_vel", "root_pos", "root_rot", "root_vel"):
            assert_close(o[name][:k], rows[name], f"{name} rows")
    if trows is not None:
        assert_close(o["ball_pos"][:m], trows["ball_pos"], "ball_pos rows")
        assert_equal(o["ball_rot"][:m], trows["ball_rot"], "ball_rot rows")


def test_reference_flow_uses_torch_draws():
    n = 500
    env, task, st = _loaded(n)
    task.sync_free = False
    task.reset_buf[::7] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    actions = torch.rand(n, 6, generator=torch.Generator().manual_seed(6)) * 2 - 1
    torch.manual_seed(77)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(77)
    dev = task.device
    draw_reset = lambda ids: torch.cat([torch.rand(len(ids), 1, device=dev) for _ in range(5)], dim=-1).cpu()
    draw_targets = lambda ids: (torch.rand((len(ids), 2), device=dev).cpu(), torch.rand(len(ids), device=dev).cpu())
    ids, rows, tids, trows = io.pre_physics(spec, b, actions, task._env_pos.cpu(), task.initial_root_pos.cpu(),
                                            task.initial_root_rot.cpu(), task.initial_ball_rot.cpu(), draw_reset, draw_targets)
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.target_positions, b.target_positions, "target_positions")
    assert_close(task.dof_pos, b.dof_pos, "dof_pos")
    assert_equal(task.dof_vel, b.dof_vel, "dof_vel")
    assert_close(task._copters.positions[ids.to(dev)], rows["root_pos"], "view root pos of the reset envs")
    assert_close(task._balls.positions[tids.to(dev)], trows["ball_pos"], "ball positions of the re-targeted envs")


def test_full_size_properties():
    n = 1 << 20
    env, task, st = _loaded(n)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(obs[:, 3:7], st["root_rot"], "obs quaternion")
    assert float(rew.min()) >= 0.0 and float(rew.max()) <= 3.0
    want = reset.nonzero().squeeze(-1).clone()
    want_t = ((task.progress_buf % 500 == 0) | (reset != 0)).nonzero().squeeze(-1).clone()
    for v in (task._copters, task._balls):
        v.apply_resets = False
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    assert int(o["count"].item()) == want.numel() and want.numel() > 0
    assert int(o["target_count"].item()) == want_t.numel() and want_t.numel() > want.numel()
    assert_equal(o["env_ids"][: want.numel()], want, "env ids == nonzero()")
    assert_equal(o["target_ids"][: want_t.numel()], want_t, "target ids")
    assert int(task.reset_buf.sum().item()) == 0
    assert float(task.thrusts[want].abs().max()) == 0.0
