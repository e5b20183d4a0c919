"""The reader of the reference's configuration files (utils/hydra_cfg.py): composition, overrides, interpolation and the
four resolvers of utils/hydra_cfg/hydra_utils.py:35-44 on a small tree written here, then -- when /root/reference is
present -- every task file in scope composed from the reference's own cfg/ and compared with the tables this package
carries (TASK_DEFAULTS)."""
import json
import os

import pytest
import yaml

from omniisaacgymenvs_amd.utils import hydra_cfg as hc
from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS, SimConfig

REF_CFG = "/root/reference/omniisaacgymenvs/cfg"

ROOT_YAML = """
task_name: ${task.name}
num_envs: ''
seed: 42
physics_engine: 'physx'
pipeline: 'gpu'
sim_device: 'gpu'
device_id: 0
rl_device: 'cuda:0'
max_iterations: ''
checkpoint: ''
headless: False
test: False
warp: False
wandb_name: ${train.params.config.name}
defaults:
  - _self_
  - task: Toy
  - train: ${task}PPO
  - override hydra/job_logging: disabled
hydra:
  output_subdir: null
"""
TOY_YAML = """
name: Toy
physics_engine: ${..physics_engine}
env:
  numEnvs: ${resolve_default:512,${...num_envs}}
  episodeLength: 100
  label: "run-${..name}-${...seed}"
  gains: [1.0, '${..episodeLength}']
sim:
  dt: 0.01
  use_gpu_pipeline: ${eq:${...pipeline},"gpu"}
  physx:
    use_gpu: ${eq:${....sim_device},"gpu"}
    has_phys: ${contains:"PHYS",${....physics_engine}}
"""
CHILD_YAML = """
defaults:
  - Toy
  - _self_
env:
  episodeLength: 7
"""
TRAIN_YAML = """
params:
  seed: ${...seed}
  load_checkpoint: ${if:${...checkpoint},True,False}
  config:
    name: ${resolve_default:Toy,${....experiment_name}}
    max_epochs: ${resolve_default:50,${....max_iterations}}
    num_actors: ${....task.env.numEnvs}
"""


@pytest.fixture()
def cfg_dir(tmp_path):
    (tmp_path / "task").mkdir()
    (tmp_path / "train").mkdir()
    (tmp_path / "config.yaml").write_text(ROOT_YAML + "experiment_name: ''\n")
    (tmp_path / "task" / "Toy.yaml").write_text(TOY_YAML)
    (tmp_path / "task" / "ToyChild.yaml").write_text(CHILD_YAML)
    (tmp_path / "train" / "ToyPPO.yaml").write_text(TRAIN_YAML)
    return str(tmp_path)


def test_defaults_relative_keys_and_resolvers(cfg_dir):
    c = hc.load_config(cfg_dir)
    assert "defaults" not in c and "hydra" not in c
    assert c["task_name"] == "Toy" and c["task"]["physics_engine"] == "physx"
    env = c["task"]["env"]
    assert env["numEnvs"] == 512 and isinstance(env["numEnvs"], int)  # '' -> the default, typed
    assert env["label"] == "run-Toy-42"  # mixed text: string
    assert env["gains"] == [1.0, 100]  # a lone interpolation keeps its type, lists are walked
    assert c["task"]["sim"]["use_gpu_pipeline"] is True and c["task"]["sim"]["physx"] == {"use_gpu": True, "has_phys": True}
    p = c["train"]["params"]
    assert p["seed"] == 42 and p["load_checkpoint"] is False
    assert p["config"] == {"name": "Toy", "max_epochs": 50, "num_actors": 512}
    assert c["wandb_name"] == "Toy"
    assert "${" not in json.dumps(c)


def test_overrides_command_line_and_dict(cfg_dir):
    c = hc.load_config(cfg_dir, ["task=ToyChild", "train=ToyPPO", "num_envs=64", "pipeline=cpu", "task.env.extra=3", "checkpoint=runs/x.pth",
                                 "max_iterations=9", "headless=True"])
    assert c["task"]["env"]["numEnvs"] == 64 and c["task"]["env"]["episodeLength"] == 7  # child file over its parent
    assert c["task"]["env"]["gains"] == [1.0, 7] and c["task"]["env"]["extra"] == 3
    assert c["task"]["sim"]["use_gpu_pipeline"] is False and c["headless"] is True
    assert c["train"]["params"]["load_checkpoint"] is True and c["train"]["params"]["config"]["max_epochs"] == 9
    assert c["train"]["params"]["config"]["num_actors"] == 64
    d = hc.load_task_config(cfg_dir, "Toy", num_envs=128, **{"task.env.episodeLength": 5})
    assert "train" not in d and "wandb_name" not in d
    assert d["task"]["env"]["numEnvs"] == 128 and d["task"]["env"]["gains"] == [1.0, 5]


def test_errors_name_the_node(cfg_dir):
    with pytest.raises(hc.ConfigError, match="no task config named 'Nope'"):
        hc.load_config(cfg_dir, {"task": "Nope"})
    with pytest.raises(hc.ConfigError, match="task.env.bad.*not found"):
        hc.load_config(cfg_dir, {"task.env.bad": "${..missing}"})
    with pytest.raises(hc.ConfigError, match="unknown resolver 'oc.env'"):
        hc.load_config(cfg_dir, {"task.env.bad": "${oc.env:HOME}"})
    with pytest.raises(hc.ConfigError, match="climbs above the root"):
        hc.load_config(cfg_dir, {"seed": "${...x}"})
    with pytest.raises(hc.ConfigError, match="circular"):
        hc.load_config(cfg_dir, {"task.env.a": "${.b}", "task.env.b": "${.a}"})
    with pytest.raises(hc.ConfigError, match="expected key=value"):
        hc.load_config(cfg_dir, ["headless"])


def test_sim_config_takes_the_composed_dict(cfg_dir):
    c = hc.load_task_config(cfg_dir, "Toy", device_id=3)
    sc = SimConfig(c)
    assert sc.task_config is c["task"]
    assert sc.config["sim_device"] == "cuda:3"  # 'gpu' + device_id, as sim_config.py:152-163 rewrites it
    assert SimConfig(hc.load_task_config(cfg_dir, "Toy", pipeline="cpu")).config["sim_device"] == "cpu"
    assert sc.get_physics_params()["dt"] == 0.01


def _flat(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _flat(v, prefix + k + ".")
        else:
            yield prefix + k, v


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("name", sorted(TASK_DEFAULTS))
def test_reference_task_files_compose_and_match_the_tables(name):
    c = hc.load_config(REF_CFG, {"task": name})
    assert "${" not in json.dumps(c)
    assert c["task_name"] == c["task"]["name"] == TASK_DEFAULTS[name]["name"]
    assert c["train"]["params"]["config"]["num_actors"] == c["task"]["env"]["numEnvs"]
    assert c["task"]["sim"]["use_gpu_pipeline"] is True
    with open(os.path.join(REF_CFG, "task", name + ".yaml")) as f:
        raw = yaml.safe_load(f)
    if "defaults" not in raw:  # numEnvs: ${resolve_default:N,${...num_envs}} -> N
        want = int(raw["env"]["numEnvs"].split(":")[1].split(",")[0])
        assert c["task"]["env"]["numEnvs"] == want
    assert hc.load_config(REF_CFG, [f"task={name}", "num_envs=64"])["task"]["env"]["numEnvs"] == 64
    # every key this package's tables carry is there under the same path, with the same value
    ours = dict(_flat({k: v for k, v in TASK_DEFAULTS[name].items() if k in ("env", "domain_randomization")}))
    theirs = dict(_flat({k: v for k, v in c["task"].items() if k in ("env", "domain_randomization")}))
    missing = [k for k in ours if k not in theirs]
    assert not missing, missing
    for k, v in ours.items():
        if k == "env.numEnvs":
            continue
        if isinstance(v, (int, float)) and not isinstance(v, bool):
            assert float(v) == float(theirs[k]), (k, v, theirs[k])
        else:
            assert v == theirs[k], (k, v, theirs[k])
    assert float(c["task"]["sim"]["dt"]) == float(TASK_DEFAULTS[name]["sim"]["dt"])
