"""The task registry mirrors the reference's (utils/task_util.py:32-132): same keys for every task in scope, a clear
error for the ones that are not, and initialize_task's call sequence on a recording env."""
import pytest

from omniisaacgymenvs_amd.utils import task_util
from omniisaacgymenvs_amd.utils.config import TASK_DEFAULTS, SimConfig, global_config, task_config

# task_util.py:70-100 of the reference, every key
REFERENCE_KEYS = {
    "AllegroHand", "Ant", "Anymal", "AnymalTerrain", "BallBalance", "Cartpole", "CartpoleCamera",
    "FactoryTaskMoldPick_TomoDJ", "FactoryTaskMoldPick_TomoDJ_newhand", "FactoryTaskNutBoltPick",
    "FactoryTaskNutBoltPick_TomoSPC", "FactoryTaskNutBoltPick_TomoDJ", "FactoryTaskNutBoltPlace",
    "FactoryTaskNutBoltScrew", "FrankaCabinet", "TomoSPCCabinet", "TomoDJCabinet", "TomoSPCnewhandCabinet",
    "TomoDJnewhandCabinet", "TomoDJnewhandTendonCabinet", "FrankaDeformable", "Humanoid", "Ingenuity", "Quadcopter",
    "Crazyflie", "ShadowHand", "ShadowHandOpenAI_FF", "ShadowHandOpenAI_LSTM", "TargetFollowing",
}


def test_registry_partitions_the_reference_keys():
    task_map, warp_map = task_util.import_tasks()
    assert set(task_map) | set(task_util.OUT_OF_SCOPE) == REFERENCE_KEYS
    assert not set(task_map) & set(task_util.OUT_OF_SCOPE)
    assert set(task_map) == set(TASK_DEFAULTS)  # every registered task has its YAML defaults and vice versa
    assert len(warp_map) == 0


def test_class_names_match_the_reference():
    task_map, _ = task_util.import_tasks()
    want = {"Ant": "AntLocomotionTask", "Humanoid": "HumanoidLocomotionTask", "ShadowHandOpenAI_LSTM": "ShadowHandTask",
            "TomoDJnewhandTendonCabinet": "TomoDJnewhandTendonCabinetTask", "Crazyflie": "CrazyflieTask"}
    for key, cls_name in want.items():
        assert task_map[key].__name__ == cls_name
    for key, cls in task_map.items():
        if key.startswith("ShadowHand"):
            continue
        assert cls.__name__.startswith(key), (key, cls.__name__)


def test_out_of_scope_and_unknown_names_say_why():
    task_map, warp_map = task_util.import_tasks()
    with pytest.raises(KeyError, match="outside the hot-path scope"):
        task_map["FactoryTaskNutBoltPick"]
    with pytest.raises(KeyError, match="no such task"):
        task_map["Antt"]
    with pytest.raises(KeyError, match="warp=False"):
        warp_map["Ant"]


def test_sim_config_single_dict_form():
    cfg = global_config(task_name="Ant", task=task_config("Ant", num_envs=64))
    sc = SimConfig(cfg)
    assert sc.task_config is cfg["task"] and sc.task_config["env"]["numEnvs"] == 64
    phys = sc.get_physics_params()
    assert phys["dt"] == cfg["task"]["sim"]["dt"] and phys["rendering_dt"] == phys["dt"]


def test_initialize_task_call_sequence(monkeypatch):
    calls = {}

    class FakeTask:
        def __init__(self, name, sim_config, env):
            calls["ctor"] = (name, sim_config.task_config["env"]["numEnvs"], env)

    class FakeEnv:
        def set_task(self, task, sim_params, backend, init_sim, rendering_dt):
            calls["set_task"] = (task, backend, init_sim, rendering_dt, sim_params["dt"])

    monkeypatch.setattr(task_util, "import_tasks", lambda: ({"Ant": FakeTask}, {}))
    env = FakeEnv()
    cfg = global_config(task_name="Ant", task=task_config("Ant", num_envs=128))
    task = task_util.initialize_task(cfg, env, init_sim=False)
    assert isinstance(task, FakeTask)
    assert calls["ctor"] == ("Ant", 128, env)
    dt = cfg["task"]["sim"]["dt"]
    assert calls["set_task"] == (task, "torch", False, dt, dt)
