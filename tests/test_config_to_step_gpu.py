"""-m gpu: the reference's launch sequence end to end on files -- a cfg/ tree in the reference's format (config.yaml with
its defaults list, task files with hydra interpolations) composed by utils/hydra_cfg.load_config, handed to
utils/task_util.initialize_task, stepped through VecEnvRLGames -- gives bit for bit what the in-memory tables give.
The tree is written here from TASK_DEFAULTS (the reference's own files do not travel to the GPU box; they are compared with
these tables on the CPU side, tests/test_hydra_cfg_cpu.py)."""
import copy

import pytest
import torch
import yaml

from helpers import assert_equal, make_task
from omniisaacgymenvs_amd.envs.vec_env_rlgames import VecEnvRLGames
from omniisaacgymenvs_amd.utils import hydra_cfg
from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS
from omniisaacgymenvs_amd.utils.task_util import initialize_task

pytestmark = pytest.mark.gpu

ROOT = {
    "task_name": "${task.name}", "num_envs": "", "seed": 42, "physics_engine": "physx", "pipeline": "gpu", "sim_device": "gpu",
    "device_id": 0, "rl_device": "cuda:0", "multi_gpu": False, "test": False, "headless": True, "enable_livestream": False,
    "enable_recording": False, "recording_interval": 2000, "recording_length": 100, "warp": False,
    "defaults": ["_self_", {"task": "Cartpole"}, {"train": "${task}PPO"}, {"override hydra/job_logging": "disabled"}],
    "hydra": {"output_subdir": None},
}


def _write_tree(root, name):
    (root / "task").mkdir()
    (root / "config.yaml").write_text(yaml.safe_dump(ROOT))
    task = copy.deepcopy(TASK_DEFAULTS[name])
    task["env"]["numEnvs"] = "${resolve_default:%d,${...num_envs}}" % task["env"]["numEnvs"]
    task["physics_engine"] = "${..physics_engine}"
    task["sim"]["use_gpu_pipeline"] = '${eq:${...pipeline},"gpu"}'
    (root / "task" / f"{name}.yaml").write_text(yaml.safe_dump(task))


@pytest.mark.parametrize("name", ["Ant", "ShadowHandOpenAI_FF", "AnymalTerrain", "Crazyflie", "TomoDJnewhandTendonCabinet"])
def test_files_to_step_matches_tables(tmp_path, name):
    n = 256
    _write_tree(tmp_path, name)
    outs = []
    for from_files in (True, False):
        torch.manual_seed(5)
        if from_files:
            cfg = hydra_cfg.load_config(str(tmp_path), [f"task={name}", f"num_envs={n}", "seed=42"])
            assert cfg["sim_device"] == "gpu" and cfg["task"]["env"]["numEnvs"] == n and cfg["task_name"] == TASK_DEFAULTS[name]["name"]
            cfg["task_name"] = name  # the registry key (rlgames_train.py passes task=<key>; ShadowHandOpenAI_FF's own name is ShadowHand)
            env = VecEnvRLGames(headless=cfg["headless"], sim_device=cfg["device_id"])
            task = initialize_task(cfg, env)
            assert str(task.device) == "cuda:0"
        else:
            env, task = make_task(name, n, seed=42)
        env.reset()
        g = torch.Generator().manual_seed(6)
        for _ in range(3):
            torch.manual_seed(12)
            actions = (torch.rand(n, task.num_actions, generator=g) * 2 - 1).to(task.rl_device)
            obs, rew, resets, _ = env.step(actions)
        torch.cuda.synchronize()
        outs.append((obs["obs"].clone(), rew.clone(), resets.clone(), task.progress_buf.clone()))
    for a, b, what in zip(outs[0], outs[1], ("obs", "rew", "resets", "progress")):
        assert_equal(a, b, what)
