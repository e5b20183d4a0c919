"""-m gpu: the launcher mirror (scripts/random_policy.py of the reference) runs from hydra-style overrides, eager and as a
replayed HIP graph, and both leave the same task state behind for the same seed."""
import pytest
import torch

from omniisaacgymenvs_amd.scripts import random_policy as rp

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("task", ["Ant", "ShadowHandOpenAI_FF", "FrankaCabinet"])
def test_eager_and_graph_runs(task, capsys):
    stats = []
    for graph, drop in ((False, False), (True, False), (True, True)):
        cfg = rp.compose(None, [f"task={task}", "num_envs=512", "seed=7"])
        assert cfg["task_name"] == task and cfg["task"]["env"]["numEnvs"] == 512
        out = rp.run(cfg, frames=20, graph=graph, report_every=0.0, drop_resets=drop)
        assert out["frames"] == 20 and out["env_steps_per_s"] > 0
        stats.append(out)
    text = capsys.readouterr().out
    assert "Setting seed: 7" in text and "FPS * num_envs:" in text


def test_anymal_terrain_runs_as_a_graph():
    """r01 refused the capture (push_robots was a host-side branch every push_interval steps); with the push inside the flags
    kernel the launcher replays AnymalTerrain like the other tasks."""
    cfg = rp.compose(None, ["task=AnymalTerrain", "num_envs=256", "seed=7"])
    out = rp.run(cfg, frames=6, graph=True, report_every=0.0)
    assert out["frames"] == 6 and out["env_steps_per_s"] > 0


def test_main_parses_the_command_line(capsys):
    assert rp.main(["task=Cartpole", "num_envs=128", "--frames", "5"]) == 0
    assert "5 frames of 128 envs" in capsys.readouterr().out


def test_rlgpu_env_adapter():
    """rlgames_train.py:68-70's registration, then the five methods rl_games calls."""
    from omniisaacgymenvs_amd.utils.rlgames.rlgames_utils import RLGPUEnv, env_configurations
    from helpers import make_task

    env, task = make_task("ShadowHandOpenAI_FF", 128)
    env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": lambda **kwargs: env})
    vec = RLGPUEnv("rlgpu", 128)
    info = vec.get_env_info()
    assert set(info) == {"action_space", "observation_space", "state_space"}
    assert info["action_space"].shape == (20,) and info["observation_space"].shape == (42,) and info["state_space"].shape == (187,)
    assert vec.get_number_of_agents() == 1
    obs = vec.reset()
    obs, rew, done, extras = vec.step(torch.zeros(128, 20, device=task.rl_device))
    assert obs["obs"].shape == (128, 42) and obs["states"].shape == (128, 187) and rew.shape == done.shape == (128,)
    env2, _ = make_task("Ant", 64)
    env_configurations.register("rlgpu", {"vecenv_type": "RLGPU", "env_creator": lambda **kwargs: env2})
    assert set(RLGPUEnv("rlgpu", 64).get_env_info()) == {"action_space", "observation_space"}
