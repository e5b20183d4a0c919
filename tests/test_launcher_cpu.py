"""CPU: the launcher's config composition (scripts/random_policy.py mirror) and set_seed."""
import os
import random

import numpy as np
import pytest
import torch

from omniisaacgymenvs_amd.scripts import random_policy as rp
from omniisaacgymenvs_amd.utils.seed import set_seed

REF_CFG = "/root/reference/omniisaacgymenvs/cfg"


def test_compose_from_tables():
    cfg = rp.compose(None, ["task=Humanoid", "num_envs=96", "task.env.episodeLength=50", "seed=-1"])
    assert cfg["task_name"] == "Humanoid" and cfg["seed"] == -1
    assert cfg["task"]["env"]["numEnvs"] == 96 and cfg["task"]["env"]["episodeLength"] == 50
    assert rp.compose(None, [])["task_name"] == "Cartpole"  # config.yaml:80 default task


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="/root/reference not present (GPU box)")
def test_compose_from_the_reference_files():
    cfg = rp.compose(REF_CFG, ["task=ShadowHandOpenAI_FF", "num_envs=64", "headless=True"])
    assert cfg["task_name"] == "ShadowHandOpenAI_FF" and cfg["task"]["name"] == "ShadowHand"
    assert cfg["task"]["env"]["numEnvs"] == 64 and cfg["task"]["domain_randomization"]["randomize"] is True


def test_set_seed(capsys):
    assert set_seed(123) == 123
    a = (random.random(), float(np.random.rand()), float(torch.rand(1)))
    set_seed(123)
    assert a == (random.random(), float(np.random.rand()), float(torch.rand(1)))
    assert set_seed(-1, torch_deterministic=True) == 42
    torch.use_deterministic_algorithms(False)
    s = set_seed(-1)
    assert 0 <= s < 10000
    assert "Setting seed: 123" in capsys.readouterr().out
