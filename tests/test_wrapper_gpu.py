"""-m gpu: VecEnvRLGames.step end to end (envs/vec_env_rlgames.py:52-104 of the reference) on the fused path --
action noise -> clamp -> pre_physics_step -> post_physics_step -> observation noise -> clamp -- against the oracle
running the reference's sequence on CPU.  fused_io on and off must hand the policy the same values."""
import copy

import pytest
import torch

from helpers import assert_close, assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads
from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS
from oracle import dr_noise_oracle as dro
from oracle import hand_oracle as ho
from oracle import locomotion_oracle as lo
from oracle import philox
from test_hand_gpu import _oracle_buffers, _oracle_state, _spec

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("fused_io,fused_dr", [(True, True), (True, False), (False, False)])
def test_shadow_hand_step_with_dr_noise_matches_reference_sequence(fused_io, fused_dr):
    n = 600
    dr = copy.deepcopy(TASK_DEFAULTS["ShadowHandOpenAI_FF"]["domain_randomization"])
    dr["randomize"] = True
    env, task = make_task("ShadowHandOpenAI_FF", n, domain_randomization=dr)
    env.fused_io, env.fused_dr = fused_io, fused_dr
    st = workloads.hand_state(n, task._env_pos.cpu(), task.goal_pos.cpu(), task.hand_dof_lower_limits.cpu(),
                              task.hand_dof_upper_limits.cpu(), device="cpu", seed=7)
    workloads.load_hand_state(task, st)
    for v in (task._hands, task._objects, task._goals):
        v.apply_resets = False
    params = dr["randomization_params"]
    r = task._dr_randomizer
    spec, s, b = _spec(task), _oracle_state(task, st), _oracle_buffers(task)
    init = ho.HandInit(task.object_init_pos.cpu(), task.goal_init_pos.cpu(), task._env_pos.cpu(),
                       task.hand_dof_default_pos.cpu(), task.hand_dof_default_vel.cpu())
    a_cnt, a_cor = torch.zeros(n, dtype=torch.int), torch.zeros(n, 20)
    o_cnt, o_cor = torch.zeros(n, dtype=torch.int), torch.zeros(n, 42)
    g = torch.Generator().manual_seed(3)
    dr_seed = int(task._cfg["seed"])
    for it in range(1, 4):
        actions = torch.rand(n, 20, generator=g) * 3 - 1.5  # beyond +-1: exercises the clamp
        # --- reference sequence on the oracle
        ta = {k: dro.philox_samples(params["actions"][k], n, 20, dr_seed, it, 1, k) for k in params["actions"]}
        noisy = dro.apply_randomization(actions.clone(), b.reset_buf, a_cnt, a_cor, params["actions"], draw=lambda kind, ids: ta[kind][ids])
        clamped = torch.clamp(noisy, -task.clip_actions, task.clip_actions)
        seed, step = task._rand_seed, task._step_count + 1
        um = lambda ids, k, stream: torch.from_numpy(2.0 * philox.uniforms(ids.numpy(), k, seed, step, stream) + (-1.0)).float()
        ho.pre_physics(spec, b, init, clamped, draw_goal=lambda ids: um(ids, 4, 0), draw_reset=lambda ids: (um(ids, 53, 1), um(ids, 4, 2)))
        ho.post_physics(spec, s, b)
        to = {k: dro.philox_samples(params["observations"][k], n, 42, dr_seed, it, 0, k) for k in params["observations"]}
        dro.apply_randomization(b.obs_buf, b.reset_buf, o_cnt, o_cor, params["observations"], draw=lambda kind, ids: to[kind][ids])
        want_obs = torch.clamp(b.obs_buf, -task.clip_obs, task.clip_obs)
        want_states = torch.clamp(b.states_buf, -task.clip_obs, task.clip_obs)
        # --- product
        obs_dict, rew, resets, extras = env.step(actions.to(task.device))
        torch.cuda.synchronize()
        assert_close(obs_dict["obs"], want_obs, f"step {it} obs")
        assert_close(obs_dict["states"], want_states, f"step {it} states")
        assert_close(task.obs_buf, b.obs_buf, f"step {it} task.obs_buf (noised in place, unclamped)")
        assert_equal(resets, b.reset_buf, f"step {it} resets")
        assert_equal(task.progress_buf, b.progress_buf, f"step {it} progress")
        assert_close(task.actions, b.actions, f"step {it} task.actions (clamped clone)")
        assert_close(task.cur_targets, b.cur_targets, f"step {it} cur_targets")
        assert float(obs_dict["obs"].abs().max()) <= task.clip_obs


@pytest.mark.parametrize("fused_io", [True, False])
def test_ant_step_clamps_actions_in_kernel(fused_io):
    from helpers import oracle_buffers, oracle_spec

    n = 300
    env, task = make_task("Ant", n)
    env.fused_io = fused_io
    st = workloads.locomotion_state(n, task.dof_limits_lower.cpu(), task.dof_limits_upper.cpu(), 4, device="cpu", seed=5)
    workloads.load_locomotion_state(task, st)
    task._robots.apply_resets = False
    spec, b = oracle_spec(task), oracle_buffers(task)
    init = (task.initial_root_pos.cpu(), task.initial_root_rot.cpu(), task.initial_dof_pos.cpu())
    actions = torch.rand(n, 8, generator=torch.Generator().manual_seed(1)) * 4 - 2
    seed, step = task._rand_seed, task._step_count + 1

    def draw(ids):
        u = torch.from_numpy(philox.uniforms(ids.numpy(), 16, seed, step))
        return u[:, :8].contiguous(), u[:, 8:].contiguous()

    lo.pre_physics(spec, b, torch.clamp(actions, -1.0, 1.0), *init, draw=draw)
    lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])
    dev_actions = actions.to(task.device)
    obs_dict, rew, resets, _ = env.step(dev_actions)
    torch.cuda.synchronize()
    assert_equal(dev_actions, actions, "the policy's action tensor is not modified")
    assert_equal(task.actions, torch.clamp(actions, -1.0, 1.0), "task.actions = clamped clone")
    assert_close(task._forces, torch.clamp(actions, -1.0, 1.0) * task.joint_gears.cpu() * task.power_scale, "forces")
    assert_close(obs_dict["obs"], b.obs_buf, "obs")
    assert_equal(resets, b.reset_buf, "resets")
    if fused_io:
        assert obs_dict["obs"].data_ptr() == task.obs_buf.data_ptr()  # clip_obs = inf: no copy
    else:
        assert obs_dict["obs"].data_ptr() != task.obs_buf.data_ptr()


@pytest.mark.parametrize("cfg,n", [("ShadowHandOpenAI_FF", 600), ("ShadowHand", 257), ("ShadowHandOpenAI_FF", 4096)])
def test_in_kernel_dr_noise_is_bit_identical_to_the_separate_launch(cfg, n):
    """fused_dr (noise applied inside hand_pre / hand_post) against the oige_dr_noise_apply launches: same Philox keys,
    same arithmetic -> identical bits in everything the step leaves behind."""
    dr = copy.deepcopy(TASK_DEFAULTS[cfg]["domain_randomization"])
    dr["randomize"] = True
    runs = []
    for fused_dr in (True, False):
        torch.manual_seed(123)  # post_reset draws its initial DOF state from torch's generator
        env, task = make_task(cfg, n, domain_randomization=dr, seed=11)
        env.fused_dr = fused_dr
        st = workloads.hand_state(n, task._env_pos.cpu(), task.goal_pos.cpu(), task.hand_dof_lower_limits.cpu(),
                                  task.hand_dof_upper_limits.cpu(), device="cpu", seed=7)
        workloads.load_hand_state(task, st)
        for v in (task._hands, task._objects, task._goals):
            v.apply_resets = False
        g = torch.Generator().manual_seed(5)
        out = []
        for it in range(3):
            actions = (torch.rand(n, 20, generator=g) * 3 - 1.5).to(task.device)
            obs_dict, rew, resets, _ = env.step(actions)
            r = task._dr_randomizer
            out.append([t.clone() for t in (obs_dict["obs"], obs_dict["states"], rew, resets, task.obs_buf, task.actions, actions,
                                            task.cur_targets, r._observations_correlated_noise, r._actions_correlated_noise,
                                            r._observations_counter_buffer, r._actions_counter_buffer)])
        torch.cuda.synchronize()
        runs.append(out)
    for it, (a, b) in enumerate(zip(*runs)):
        for k, (x, y) in enumerate(zip(a, b)):
            assert torch.equal(x, y), f"step {it} item {k}"
    assert float(runs[0][-1][8].abs().max()) > 0  # the correlated noise was refreshed at least once
