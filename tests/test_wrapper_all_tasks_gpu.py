"""-m gpu: every task mirror runs through the unchanged ``VecEnvRLGames.reset() / step()`` contract
(envs/vec_env_rlgames.py:57-104): shapes, dtypes, the observation clamp, and agreement with the task-level calls."""
import math

import pytest
import torch

from helpers import assert_equal, make_task

pytestmark = pytest.mark.gpu
TASKS = ["Cartpole", "Ant", "Humanoid", "ShadowHand", "ShadowHandOpenAI_FF", "AllegroHand", "Anymal", "AnymalTerrain", "Quadcopter",
         "BallBalance", "Ingenuity", "Crazyflie", "FrankaCabinet", "TomoDJCabinet", "TomoSPCCabinet", "TomoDJnewhandCabinet",
         "TomoDJnewhandTendonCabinet", "TomoSPCnewhandCabinet"]


@pytest.mark.parametrize("fused_io", [True, False])
@pytest.mark.parametrize("name", TASKS)
def test_step_contract(name, fused_io):
    n = 300
    torch.manual_seed(1)
    env, task = make_task(name, n)
    env.fused_io = fused_io
    obs = env.reset()
    assert set(obs) >= {"obs"} and obs["obs"].shape == (n, task.num_observations)
    g = torch.Generator().manual_seed(2)
    for _ in range(3):
        actions = (torch.rand(n, task.num_actions, generator=g) * 4 - 2).to(task.rl_device)   # beyond clip_actions on purpose
        obs, rew, resets, extras = env.step(actions)
        torch.cuda.synchronize()
        o = obs["obs"]
        assert o.shape == (n, task.num_observations) and o.dtype == torch.float32 and bool(torch.isfinite(o).all())
        assert rew.shape == (n,) and rew.dtype == torch.float32 and bool(torch.isfinite(rew).all())
        assert resets.shape == (n,) and resets.dtype in (torch.int64, torch.bool)
        assert isinstance(extras, dict)
        if not math.isinf(task.clip_obs):
            assert float(o.abs().max()) <= task.clip_obs
        if task.num_states > 0:
            assert obs["states"].shape == (n, task.num_states)
    assert int(task.progress_buf.max()) >= 1


@pytest.mark.parametrize("name", ["Quadcopter", "Crazyflie", "FrankaCabinet", "TomoDJnewhandTendonCabinet"])
def test_fused_and_reference_wrapper_flows_agree(name):
    """fused_io folds the action clamp into the pre kernel and skips the no-op copies; the reference flow clamps with torch."""
    n = 500
    outs = []
    for fused_io in (True, False):
        torch.manual_seed(3)
        env, task = make_task(name, n)
        env.fused_io = fused_io
        env.reset()
        g = torch.Generator().manual_seed(4)
        for _ in range(3):
            torch.manual_seed(11)   # Crazyflie draws its thrust noise with torch
            actions = (torch.rand(n, task.num_actions, generator=g) * 4 - 2).to(task.rl_device)
            obs, rew, resets, _ = env.step(actions)
        torch.cuda.synchronize()
        outs.append((obs["obs"].clone(), rew.clone(), resets.clone(), task.progress_buf.clone()))
    for a, b, what in zip(outs[0], outs[1], ("obs", "rew", "resets", "progress")):
        assert_equal(a, b, what)
