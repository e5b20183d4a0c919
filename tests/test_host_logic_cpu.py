"""CPU: host-side logic that needs no GPU -- configuration tables against the reference's YAML values, the synthetic
terrain's output contract, the Randomizer's parameter validation (randomize.py:215-281) and the noise-term folding."""
import math
import os

import numpy as np
import pytest
import yaml

from omniisaacgymenvs_amd.sim.terrain import SyntheticTerrain
from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS, task_config
from omniisaacgymenvs_amd.utils.domain_randomization import noise_hip
from omniisaacgymenvs_amd.utils.domain_randomization.randomize import Randomizer

REF_CFG = "/root/reference/omniisaacgymenvs/cfg/task"


def _flat(d, prefix=""):
    for k, v in d.items():
        if isinstance(v, dict):
            yield from _flat(v, prefix + k + ".")
        else:
            yield prefix + k, v


@pytest.mark.skipif(not os.path.isdir(REF_CFG), reason="/root/reference not present (GPU box)")
@pytest.mark.parametrize("name", ["Cartpole", "Ant", "Humanoid", "ShadowHand", "ShadowHandOpenAI_FF", "AnymalTerrain", "Anymal", "AllegroHand", "Quadcopter", "BallBalance", "Ingenuity", "Crazyflie", "FrankaCabinet", "TomoDJCabinet", "TomoSPCCabinet", "TomoDJnewhandCabinet",
                                  "TomoDJnewhandTendonCabinet", "TomoSPCnewhandCabinet"])
def test_task_defaults_carry_the_reference_yaml_values(name):
    """Every scalar env.* key we carry equals the reference YAML's value (hydra interpolations skipped)."""
    with open(os.path.join(REF_CFG, name + ".yaml")) as f:
        ref = yaml.safe_load(f)
    ours = dict(_flat(TASK_DEFAULTS[name]["env"]))
    theirs = dict(_flat(ref["env"]))
    checked = 0
    for k, v in ours.items():
        if k not in theirs or (isinstance(theirs[k], str) and "${" in theirs[k]):
            continue
        checked += 1
        if isinstance(v, float) or isinstance(theirs[k], float):
            assert float(v) == float(theirs[k]), (name, k, v, theirs[k])
        else:
            assert v == theirs[k], (name, k, v, theirs[k])
    assert checked >= 4


def test_synthetic_terrain_keeps_the_reference_output_contract():
    t = SyntheticTerrain(TASK_DEFAULTS["AnymalTerrain"]["env"]["terrain"], num_robots=64)
    assert t.heightsamples.dtype == np.int16 and t.heightsamples.shape == (1200, 2000) == (t.tot_rows, t.tot_cols)
    assert t.env_origins.shape == (10, 20, 3)
    assert (t.horizontal_scale, t.vertical_scale, t.border_size) == (0.1, 0.005, 20)
    assert t.env_origins[3, 5, 0] == 3.5 * 8.0 and t.env_origins[3, 5, 1] == 5.5 * 8.0
    b = t.border
    assert not t.heightsamples[:b].any() and not t.heightsamples[:, :b].any()  # flat border
    assert t.heightsamples[b:-b, b:-b].any()


def test_task_config_overrides_and_top_level_sections():
    cfg = task_config("ShadowHand", num_envs=77, observationType="openai", domain_randomization={"randomize": True})
    assert cfg["env"]["numEnvs"] == 77 and cfg["env"]["observationType"] == "openai"
    assert cfg["domain_randomization"] == {"randomize": True} and "domain_randomization" not in cfg["env"]
    assert TASK_DEFAULTS["ShadowHand"]["env"]["observationType"] == "full"  # defaults untouched


def test_randomizer_validation_matches_the_reference_errors():
    class _T:
        num_observations, num_actions = 42, 20
        randomize_observations = randomize_actions = False

    def make(obs):
        cfg = {"env": {"numEnvs": 8}, "domain_randomization": {"randomize": True, "randomization_params": {"observations": obs}}}
        return Randomizer({"rl_device": "cpu", "seed": 1}, cfg)

    with pytest.raises(ValueError, match="on_reset randomization parameters"):
        make({"on_reset": {"operation": "additive"}}).set_up_domain_randomization(_T())
    with pytest.raises(ValueError, match="on_interval randomization parameters"):
        make({"on_interval": {"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0, 1]}}
             ).set_up_domain_randomization(_T())
    with pytest.raises(ValueError, match="not provided"):
        make(None).set_up_domain_randomization(_T())
    t = _T()
    r = make({"on_reset": {"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0, 0.1]}})
    r.set_up_domain_randomization(t)
    assert t.randomize_observations and r._observations_counter_buffer.dtype.is_floating_point is False
    assert tuple(r._observations_correlated_noise.shape) == (8, 42)
    off = Randomizer({"rl_device": "cpu"}, {"env": {"numEnvs": 8}})
    with pytest.raises(ValueError, match="No domain randomization parameters"):
        off.set_up_domain_randomization(_T())


def test_noise_terms_fold_their_parameters_like_python():
    p = {"on_reset": {"operation": "scaling", "distribution": "uniform", "distribution_parameters": [0.9, 1.1]},
         "on_interval": {"frequency_interval": 3, "operation": "additive", "distribution": "loguniform",
                         "distribution_parameters": [0.5, 2.0]}}
    a, b = noise_hip._term(p, "on_reset"), noise_hip._term(p, "on_interval")
    assert (a.enabled, a.operation, a.distribution) == (1, 1, 1)
    assert a.p0 == np.float32(0.9) and a.p1 == np.float32(1.1 - 0.9)
    assert (b.enabled, b.operation, b.distribution, b.frequency_interval) == (1, 0, 2, 3)
    assert b.p0 == np.float32(math.log(0.5)) and b.p1 == np.float32(math.log(2.0) - math.log(0.5))
    assert noise_hip._term(p, "absent").enabled == 0
    with pytest.raises(Exception, match="not supported"):
        noise_hip._term({"on_reset": {"operation": "xor", "distribution": "uniform", "distribution_parameters": [0, 1]}}, "on_reset")


def test_counts_valid_key_follows_the_tensor_not_its_address():
    """OIGE_PRE_COUNTS_VALID keyed on (data_ptr, _version) can match a REBOUND reset_buf that landed on a recycled
    address.  The key holds the tensor objects: a rebind, an in-place torch write or reset() invalidates it."""
    import torch

    from omniisaacgymenvs_amd.tasks.base.rl_task import RLTask

    t = RLTask.__new__(RLTask)
    t.reset_buf = torch.zeros(8, dtype=torch.int64)
    goal = torch.zeros(8, dtype=torch.int64)
    assert not t._counts_valid(t.reset_buf)          # nothing marked yet
    t._mark_counts(t.reset_buf, goal)
    assert t._counts_valid(t.reset_buf, goal)
    assert not t._counts_valid(t.reset_buf)          # a different set of flag buffers
    t.reset_buf[3] = 1                               # torch-side write: version moves
    assert not t._counts_valid(t.reset_buf, goal)
    t._mark_counts(t.reset_buf, goal)
    old_ptr = t.reset_buf.data_ptr()
    view = t.reset_buf.view(-1)                      # same storage, same address, same version -- another object
    assert view.data_ptr() == old_ptr and not t._counts_valid(view, goal)
    t._capturing = True                              # never baked into a graph
    assert not t._counts_valid(t.reset_buf, goal)
    t._capturing = False
    assert t._counts_valid(t.reset_buf, goal)
    t.reset()                                        # rl_task.py:274-276 rebinds reset_buf
    assert t._counts_key is None and not t._counts_valid(t.reset_buf, goal)
