"""-m gpu parity tests: HIP locomotion path (through the C ABI) vs the CPU oracle on identical
seeded inputs.  Bar (BASELINE.json north_star): done flags, reset ids and progress counters
bit-exact; fp32 observations and rewards within 1e-5 relative / 1e-6 absolute."""
import numpy as np
import pytest
import torch

from helpers import ATOL, RTOL, assert_close, assert_equal, make_task, oracle_buffers, oracle_spec
from omniisaacgymenvs_amd.sim import workloads
from oracle import locomotion_oracle as lo
from oracle import philox

pytestmark = pytest.mark.gpu

Z_RANGE = {"Ant": (0.2, 0.9), "Humanoid": (0.5, 1.5)}


def _loaded_task(name, n, seed=1234, reset_frac=0.1):
    env, task = make_task(name, n)
    st = workloads.locomotion_state(n, task.dof_limits_lower.cpu(), task.dof_limits_upper.cpu(), task._num_sensors,
                                    device="cpu", seed=seed, z_range=Z_RANGE[name])
    g = torch.Generator().manual_seed(seed + 1)
    st["reset"] = (torch.rand(n, generator=g) < reset_frac).long()
    # progress values that straddle the episode-length threshold
    st["progress"][: min(n, 6)] = torch.tensor([996, 997, 998, 999, 1000, 0])[: min(n, 6)]
    workloads.load_locomotion_state(task, st)
    return env, task, st


def _assert_no_threshold_flip(obs_gpu, obs_expected, what):
    """Rewards jump by up_weight where up_proj crosses 0.93 and change slope where heading_proj crosses 0.8
    (locomotion.py:316-319).  The kernel evaluates both in the reference's operation order (census:
    profiles/r02/threshold_flips.json, 0 flips, max |up_proj error| 0.0), so NO row is excluded from the reward
    comparison any more: the decisions must agree on every row, rows within 1e-6 of a threshold included."""
    o, e = obs_gpu.detach().cpu(), obs_expected.detach().cpu()
    near = int(((e[:, 10] - 0.93).abs() <= 1e-6).sum() + ((e[:, 11] - 0.8).abs() <= 1e-6).sum())
    flips = int(((o[:, 10] > 0.93) != (e[:, 10] > 0.93)).sum() + ((o[:, 11] > 0.8) != (e[:, 11] > 0.8)).sum())
    assert flips == 0, f"{what}: {flips} threshold decision(s) differ from the oracle's ({near} row(s) within 1e-6 of a threshold)"


@pytest.mark.parametrize("name", ["Ant", "Humanoid"])
@pytest.mark.parametrize("n", [1, 64, 257, 4096])
def test_fused_post_physics_matches_oracle(name, n):
    env, task, st = _loaded_task(name, n)
    spec, b = oracle_spec(task), oracle_buffers(task)
    lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_equal(reset, b.reset_buf, "reset_buf")
    assert_close(obs, b.obs_buf, "obs_buf")
    assert_close(task.potentials, b.potentials, "potentials", rtol=0, atol=0)
    assert_equal(task.prev_potentials, b.prev_potentials, "prev_potentials")
    assert_close(task.up_vec, b.up_vec, "up_vec")
    assert_close(task.heading_vec, b.heading_vec, "heading_vec")
    _assert_no_threshold_flip(obs, b.obs_buf, "post_physics")
    assert_close(rew, b.rew_buf, "rew_buf")  # every row
    # the DONE stage's per-tile counts describe reset_buf
    tiles = (n + 63) // 64
    want = torch.nn.functional.pad(b.reset_buf, (0, tiles * 64 - n)).view(tiles, 64).ne(0).sum(-1).int()
    assert_equal(task._reset_workspace[:tiles], want, "tile counts")


@pytest.mark.parametrize("name", ["Ant", "Humanoid"])
def test_split_stage_methods_match_fused(name):
    n = 1000
    env, task, st = _loaded_task(name, n)
    env2, task2, _ = _loaded_task(name, n)
    task.post_physics_step()
    task2.fused = False
    task2.post_physics_step()  # RLTask order: progress, get_observations, calculate_metrics, is_done
    torch.cuda.synchronize()
    for attr in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "potentials", "prev_potentials"):
        assert_equal(getattr(task2, attr), getattr(task, attr), attr)


@pytest.mark.parametrize("name", ["Ant", "Humanoid"])
@pytest.mark.parametrize("n,reset_frac", [(64, 0.0), (257, 0.3), (1000, 1.0), (5000, 0.05)])
def test_sync_free_pre_physics_matches_oracle(name, n, reset_frac):
    env, task, st = _loaded_task(name, n, reset_frac=reset_frac)
    d = task._robots.num_dof
    spec, b = oracle_spec(task), oracle_buffers(task)
    g = torch.Generator().manual_seed(7)
    actions = (torch.rand(n, d, generator=g) * 2 - 1)
    seed, step = task._rand_seed, task._step_count + 1

    def draw(ids):
        u = torch.from_numpy(philox.uniforms(ids.numpy(), 2 * d, seed, step))
        return u[:, :d].contiguous(), u[:, d:].contiguous()

    init = (task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), task.initial_dof_pos.cpu())
    forces, writes = lo.pre_physics(spec, b, actions, *init, draw=draw)
    task._robots.apply_resets = False  # inspect the compacted outputs directly
    task.pre_physics_step(actions.to(task.device))
    torch.cuda.synchronize()
    k = int(task._reset_out["count"].item())
    assert k == (0 if writes is None else len(writes.env_ids))
    assert_equal(task._forces, forces, "forces")
    assert_equal(task.actions, b.actions, "actions clone")
    assert_equal(task.reset_buf, b.reset_buf, "reset_buf")
    assert_equal(task.progress_buf, b.progress_buf, "progress_buf")
    assert_equal(task.potentials, b.potentials, "potentials")
    assert_equal(task.prev_potentials, b.prev_potentials, "prev_potentials")
    if writes is not None:
        o = task._reset_out
        assert_equal(o["env_ids"][:k], writes.env_ids, "reset_env_ids")
        assert_equal(o["dof_pos"][:k], writes.dof_pos, "dof_pos")
        assert_equal(o["dof_vel"][:k], writes.dof_vel, "dof_vel")
        assert_equal(o["root_pos"][:k], writes.root_pos, "root_pos")
        assert_equal(o["root_rot"][:k], writes.root_rot, "root_rot")
        assert_equal(o["root_vel"][:k], writes.root_vel, "root_vel")


@pytest.mark.parametrize("name", ["Ant", "Humanoid"])
def test_reset_idx_uses_torch_draws_like_the_reference(name):
    n = 777
    env, task, st = _loaded_task(name, n)
    d = task._robots.num_dof
    ids = torch.tensor([0, 5, 6, 64, 65, 300, 776], dtype=torch.int64)
    torch.manual_seed(123)
    task.reset_idx(ids.to(task.device))
    torch.cuda.synchronize()
    torch.manual_seed(123)
    u_pos = torch.rand(len(ids), d, device=task.device).cpu()
    u_vel = torch.rand(len(ids), d, device=task.device).cpu()
    spec, b = oracle_spec(task), oracle_buffers(task)
    b.reset_buf[ids] = 1  # oracle_buffers snapshots the already-reset task; re-arm for the check below
    w = lo.reset_idx(spec, b, ids, task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), task.initial_dof_pos.cpu(),
                     u_pos, u_vel)
    v = task._robots
    assert_equal(v.joint_positions[ids.to(task.device)], w.dof_pos, "view dof_pos")
    assert_equal(v.joint_velocities[ids.to(task.device)], w.dof_vel, "view dof_vel")
    assert_equal(v.velocities[ids.to(task.device)], w.root_vel, "view root vel")
    assert_equal(task.reset_buf[ids.to(task.device)], torch.zeros(len(ids), dtype=torch.int64), "reset_buf[ids]")
    assert_equal(task.progress_buf[ids.to(task.device)], torch.zeros(len(ids), dtype=torch.int64), "progress_buf[ids]")
    assert_equal(task.potentials[ids.to(task.device)], b.potentials[ids], "potentials[ids]")


@pytest.mark.parametrize("name", ["Ant"])
def test_reference_flow_with_host_count(name):
    """sync_free=False reproduces the reference's pre_physics flow (host len(), torch draws)."""
    n = 300
    env, task, st = _loaded_task(name, n, reset_frac=0.2)
    task.sync_free = False
    want_ids = task.reset_buf.nonzero().squeeze(-1).clone()
    torch.manual_seed(5)
    task.pre_physics_step(torch.zeros(n, 8, device=task.device))
    torch.cuda.synchronize()
    assert int(task.reset_buf.sum().item()) == 0
    assert_equal(task.progress_buf[want_ids], torch.zeros(len(want_ids), dtype=torch.int64), "progress reset")
    lo_, hi_ = task.dof_limits_lower, task.dof_limits_upper
    q = task._robots.joint_positions[want_ids]
    assert bool(((q >= lo_) & (q <= hi_)).all())


def test_multi_step_episode_through_vec_env():
    """VecEnvRLGames.reset/step drive the fused path for several steps; compare the whole
    trajectory of buffers against the oracle stepping the same synthetic state."""
    n = 512
    env, task, st = _loaded_task("Ant", n, reset_frac=0.0)
    spec, b = oracle_spec(task), oracle_buffers(task)
    init = (task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), task.initial_dof_pos.cpu())
    g = torch.Generator().manual_seed(11)
    task._robots.apply_resets = False  # physics is delegated: the synthetic state stays put
    for it in range(4):
        actions = torch.rand(n, 8, generator=g) * 3 - 1.5  # beyond +-1: exercises the wrapper clamp
        seed, step = task._rand_seed, task._step_count + 1

        def draw(ids):
            u = torch.from_numpy(philox.uniforms(ids.numpy(), 16, seed, step))
            return u[:, :8].contiguous(), u[:, 8:].contiguous()

        lo.pre_physics(spec, b, torch.clamp(actions, -1.0, 1.0), *init, draw=draw)
        lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])
        obs_dict, rew, resets, extras = env.step(actions.to(task.device))
        torch.cuda.synchronize()
        assert_equal(resets, b.reset_buf, f"step {it} resets")
        assert_equal(task.progress_buf, b.progress_buf, f"step {it} progress")
        assert_close(obs_dict["obs"], b.obs_buf, f"step {it} obs")
        _assert_no_threshold_flip(obs_dict["obs"], b.obs_buf, f"step {it}")
        assert_close(rew, b.rew_buf, f"step {it} rew")  # every row


@pytest.mark.parametrize("name,n", [("Ant", 1 << 20), ("Humanoid", 1 << 19)])
def test_full_size_properties(name, n):
    """BASELINE.json sizes: properties that need no CPU oracle pass over a million rows --
    reset ids == nonzero() (ascending, bit-exact), idempotent observations, pass-through
    columns equal their inputs, and a sampled slice checked against the oracle."""
    env, task, st = _loaded_task(name, n, reset_frac=0.03)
    d, s = task._robots.num_dof, task._num_sensors
    dev = task.device
    task.post_physics_step()
    obs1 = task.obs_buf.clone()
    # pass-through columns
    assert_equal(obs1[:, 0], st["pos"][:, 2], "obs[:,0] == pos.z")
    assert_equal(obs1[:, 12 + 2 * d + 6 * s:], st["actions"], "obs actions columns")
    assert_equal(obs1[:, 12 + d:12 + 2 * d], st["dof_vel"] * task.dof_vel_scale, "obs dof_vel columns")
    # idempotence of the observation stage (pure function of the state)
    task.get_observations()
    assert_equal(task.obs_buf, obs1, "get_observations idempotent")
    # done flags vs a direct evaluation of the two conditions
    want_reset = ((st["pos"][:, 2] < task.termination_height) | (task.progress_buf.cpu() >= 999) | (st["reset"] != 0)).long()
    assert_equal(task.reset_buf, want_reset, "reset_buf")
    want_ids = task.reset_buf.nonzero().squeeze(-1).clone()
    task._robots.apply_resets = False
    task.pre_physics_step(st["actions"].to(dev))
    torch.cuda.synchronize()
    k = int(task._reset_out["count"].item())
    assert k == want_ids.numel()
    assert_equal(task._reset_out["env_ids"][:k], want_ids, "reset_env_ids == nonzero()")
    assert int(task.reset_buf.sum().item()) == 0
    # sampled slice against the oracle
    sl = slice(n // 2 - 500, n // 2 + 500)
    spec = oracle_spec(task)
    obs_o, pot_o, _, _, _ = lo.observations(spec, st["pos"][sl], st["rot"][sl], st["vel"][sl], st["dof_pos"][sl],
                                            st["dof_vel"][sl], st["sensors"][sl], st["actions"][sl], st["potentials"][sl])
    assert_close(obs1[sl], obs_o, "obs slice")


@pytest.mark.parametrize("name", ["Ant", "Humanoid"])
@pytest.mark.parametrize("n", [1, 257, 4096])
def test_get_dof_at_limit_cost_matches_the_reference_method(name, n):
    """The public method (ant.py:102-109 -> int64 count, no abs; humanoid.py:130-141 -> fp32) on an obs_buf with the
    threshold rows SURVEY.md 8c asks for: +-0.98, +-0.99, -0.995, one ULP either side."""
    env, task, st = _loaded_task(name, n)
    task.post_physics_step()
    d = task._robots.num_dof
    edge = torch.tensor([0.98, -0.98, 0.99, -0.99, -0.995, 0.995, float(np.nextafter(np.float32(0.99), np.float32(2))),
                         float(np.nextafter(np.float32(0.99), np.float32(0))), float(np.nextafter(np.float32(0.98), np.float32(2))),
                         float(np.nextafter(np.float32(0.98), np.float32(0))), 1.5, -1.5, 0.0])
    rows = min(n, len(edge))
    task.obs_buf[:rows, 12:12 + d] = edge[:rows, None].to(task.obs_buf.device)
    got = task.get_dof_at_limit_cost()
    torch.cuda.synchronize()
    want = lo.dof_at_limit_cost(oracle_spec(task), task.obs_buf.cpu())
    assert got.shape == (n,) and got.dtype == want.dtype == (torch.int64 if name == "Ant" else torch.float32)
    if name == "Ant":
        assert_equal(got, want, "dof_at_limit_cost")  # a count: bit-exact
        if n >= 6:
            assert got[:6].tolist() == [0, 0, 0, 0, 0, d]  # no abs: -0.995 does not count
    else:
        assert_close(got, want, "dof_at_limit_cost")
