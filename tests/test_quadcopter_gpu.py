"""-m gpu parity tests: HIP Quadcopter path (through the C ABI) vs the CPU oracle (pinned to tasks/quadcopter.py by
tests/test_oracle_pinned_quadcopter.py).  Bar: reset flags, reset ids, progress bit-exact; fp32 observations, rewards,
targets, thrusts, forces within 1e-5 rel / 1e-6 abs."""
import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from oracle import philox
from oracle import quadcopter_oracle as qo

pytestmark = pytest.mark.gpu


def _loaded(n, seed=1234):
    env, task = make_task("Quadcopter", n)
    st = workloads.quadcopter_state(n, task._env_pos.cpu(), device="cpu", seed=seed)
    if n > 8:
        st["progress"][:4] = torch.tensor([496, 497, 498, 499])
        st["root_pos"][4:7, 2] = torch.tensor([0.2999, 0.3001, 0.31])
        st["root_vel"][7:9, 5] = torch.tensor([40.0, -45.0])  # the reward clips at 0
    workloads.load_quadcopter_state(task, st)
    task.forces.copy_(torch.randn(n, 4, 3, generator=torch.Generator().manual_seed(seed)).to(task.device))
    return env, task, st


def _oracle_buffers(task) -> qo.QuadcopterBuffers:
    c = lambda t: t.detach().cpu().clone()
    return qo.QuadcopterBuffers(c(task.obs_buf), c(task.rew_buf), c(task.reset_buf), c(task.progress_buf), c(task.target_positions),
                                c(task.dof_position_targets), c(task.thrusts), c(task.forces))


def _spec(task) -> qo.QuadcopterSpec:
    return qo.QuadcopterSpec(max_episode_length=task._max_episode_length, dt=task.dt, lower=task.dof_lower_limits.cpu(),
                             upper=task.dof_upper_limits.cpu())


@pytest.mark.parametrize("n", [1, 40, 64, 257, 4096])
@pytest.mark.parametrize("fused", [True, False])
def test_post_physics_matches_oracle(n, fused):
    env, task, st = _loaded(n)
    task.fused = fused
    spec, b = _spec(task), _oracle_buffers(task)
    env_pos = task._env_pos.cpu()
    qo.post_physics(spec, b, st["root_pos"], st["root_rot"], st["root_vel"], st["dof_pos"], env_pos)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    rp = st["root_pos"] - env_pos
    td = torch.sqrt(torch.square(b.target_positions - rp).sum(-1))
    ok = ((td - 3.0).abs() > 1e-6) & ((rp[:, 2] - 0.3).abs() > 1e-6)
    assert_equal(reset[ok], b.reset_buf[ok], "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(obs, b.obs_buf, "obs_buf")
    assert_close(rew, b.rew_buf, "rew_buf")
    if n > 8:
        assert float(rew[7]) == 0.0 and int(reset[4]) == 1
    if bool(ok.all()):
        tiles = (n + 63) // 64
        pad = lambda t: torch.nn.functional.pad(t, (0, tiles * 64 - n)).view(tiles, 64).ne(0).sum(-1).int()
        assert_equal(task._reset_workspace[:tiles], pad(task.reset_buf.cpu()), "reset tile counts")


@pytest.mark.parametrize("n", [1, 64, 300, 5000])
@pytest.mark.parametrize("clamp", [False, True])
def test_sync_free_pre_physics_matches_oracle(n, clamp):
    env, task, st = _loaded(n)
    task._copters.apply_resets = False
    if n >= 64:
        task.reset_buf[:n // 3] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    seed, step = task._rand_seed, task._step_count + 1

    def draw(ids):
        u = torch.from_numpy(philox.uniforms(ids.numpy(), 12, seed, step))
        return u[:, :8].contiguous(), u[:, 8:9], u[:, 9:10], u[:, 10:11]

    actions = torch.rand(n, 12, generator=torch.Generator().manual_seed(5)) * 3 - 1.5
    task.clamp_actions_in_kernel = clamp
    want_actions = torch.clamp(actions, -task.clip_actions, task.clip_actions) if clamp else actions
    ids, rows = qo.pre_physics(spec, b, want_actions, task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), draw)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    k = int(o["count"].item())
    assert k == len(ids)
    assert_equal(o["env_ids"][:k], ids, "env ids == reset_buf.nonzero()")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_close(task.dof_position_targets, b.dof_position_targets, "dof_position_targets")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.forces, b.forces, "forces")
    assert_close(task._copters.joint_position_targets, b.dof_position_targets, "targets handed to the view")
    assert_close(task._copters.rotors.applied_forces, b.forces, "forces handed to the rotors")
    if rows is not None:
        assert_close(o["dof_pos"][:k], rows.dof_pos, "dof_pos rows")
        assert_equal(o["dof_vel"][:k], rows.dof_vel, "dof_vel rows")
        assert_close(o["root_pos"][:k], rows.root_pos, "root_pos rows")
        assert_equal(o["root_rot"][:k], rows.root_rot, "root_rot rows")
        assert_equal(o["root_vel"][:k], rows.root_vel, "root_vel rows")


@pytest.mark.parametrize("sync_free", [False])
def test_reference_flow_uses_torch_draws(sync_free):
    n = 500
    env, task, st = _loaded(n)
    task.sync_free = sync_free
    task.reset_buf[::7] = 1
    spec, b = _spec(task), _oracle_buffers(task)
    actions = torch.rand(n, 12, generator=torch.Generator().manual_seed(6)) * 2 - 1
    torch.manual_seed(77)
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(77)

    def draw(ids):
        k = len(ids)
        dev = task.device
        return (torch.rand(k, 8, device=dev).cpu(), torch.rand(k, 1, device=dev).cpu(), torch.rand(k, 1, device=dev).cpu(),
                torch.rand(k, 1, device=dev).cpu())

    ids, rows = qo.pre_physics(spec, b, actions, task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), draw)
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_close(task.dof_position_targets, b.dof_position_targets, "dof_position_targets")
    assert_close(task.thrusts, b.thrusts, "thrusts")
    assert_close(task.forces, b.forces, "forces")
    dev_ids = ids.to(task.device)
    assert_close(task._copters.joint_positions[dev_ids], rows.dof_pos, "view dof_pos of the reset envs")
    assert_close(task._copters.positions[dev_ids], rows.root_pos, "view root pos of the reset envs")


def test_full_size_properties():
    n = 1 << 20
    env, task, st = _loaded(n)
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(obs[:, 13:21], st["dof_pos"], "obs dof columns")
    assert_equal(obs[:, 3:7], st["root_rot"], "obs quaternion")
    assert float(rew.min()) >= 0.0 and float(rew.max()) <= 3.0
    want = reset.nonzero().squeeze(-1).clone()
    task._copters.apply_resets = False
    task.pre_physics_step(st["actions"].to(task.device))
    torch.cuda.synchronize()
    o = task._reset_out
    assert int(o["count"].item()) == want.numel() and want.numel() > 0
    assert_equal(o["env_ids"][: want.numel()], want, "env ids == nonzero()")
    assert int(task.reset_buf.sum().item()) == 0
    assert float(task.thrusts[want].abs().max()) == 0.0
