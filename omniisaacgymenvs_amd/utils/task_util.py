"""Task registry: the reference's ``import_tasks`` / ``initialize_task`` (utils/task_util.py:32-132) over the mirror
classes of this package, so a launcher written against the reference resolves ``task_name`` the same way.

Tasks outside the hot-path scope (Factory*, CartpoleCamera, FrankaDeformable, TargetFollowing: SURVEY.md 8, DESIGN.md
"out of scope") and the Warp variants have no mirror; asking for one raises a KeyError that says so instead of the
bare key.
"""
from __future__ import annotations

from typing import Any, Dict, Tuple

# names the reference registers (task_util.py:70-105) that this package deliberately does not build
OUT_OF_SCOPE = (
    "CartpoleCamera", "FactoryTaskMoldPick_TomoDJ", "FactoryTaskMoldPick_TomoDJ_newhand", "FactoryTaskNutBoltPick",
    "FactoryTaskNutBoltPick_TomoSPC", "FactoryTaskNutBoltPick_TomoDJ", "FactoryTaskNutBoltPlace",
    "FactoryTaskNutBoltScrew", "FrankaDeformable", "TargetFollowing",
)


class _TaskMap(dict):
    def __init__(self, items, what: str):
        super().__init__(items)
        self._what = what

    def __missing__(self, key):
        if key in OUT_OF_SCOPE:
            raise KeyError(f"{key}: outside the hot-path scope of omniisaacgymenvs_amd (see DESIGN.md, out of scope)")
        raise KeyError(f"{key}: no such {self._what}; known: {sorted(self)}")


def import_tasks() -> Tuple[Dict[str, Any], Dict[str, Any]]:
    """(task_map, task_map_warp) keyed like task_util.py:70-105."""
    from omniisaacgymenvs_amd.tasks.allegro_hand import AllegroHandTask
    from omniisaacgymenvs_amd.tasks.ant import AntLocomotionTask
    from omniisaacgymenvs_amd.tasks.anymal import AnymalTask
    from omniisaacgymenvs_amd.tasks.anymal_terrain import AnymalTerrainTask
    from omniisaacgymenvs_amd.tasks.ball_balance import BallBalanceTask
    from omniisaacgymenvs_amd.tasks.cartpole import CartpoleTask
    from omniisaacgymenvs_amd.tasks.crazyflie import CrazyflieTask
    from omniisaacgymenvs_amd.tasks.franka_cabinet import FrankaCabinetTask
    from omniisaacgymenvs_amd.tasks.humanoid import HumanoidLocomotionTask
    from omniisaacgymenvs_amd.tasks.ingenuity import IngenuityTask
    from omniisaacgymenvs_amd.tasks.quadcopter import QuadcopterTask
    from omniisaacgymenvs_amd.tasks.shadow_hand import ShadowHandTask
    from omniisaacgymenvs_amd.tasks.tomodj_cabinet import TomoDJCabinetTask
    from omniisaacgymenvs_amd.tasks.tomodj_newhand_cabinet import TomoDJnewhandCabinetTask
    from omniisaacgymenvs_amd.tasks.tomodj_newhandTendon_cabinet import TomoDJnewhandTendonCabinetTask
    from omniisaacgymenvs_amd.tasks.tomospc_cabinet import TomoSPCCabinetTask
    from omniisaacgymenvs_amd.tasks.tomospc_newhand_cabinet import TomoSPCnewhandCabinetTask

    task_map = _TaskMap({
        "AllegroHand": AllegroHandTask,
        "Ant": AntLocomotionTask,
        "Anymal": AnymalTask,
        "AnymalTerrain": AnymalTerrainTask,
        "BallBalance": BallBalanceTask,
        "Cartpole": CartpoleTask,
        "Crazyflie": CrazyflieTask,
        "FrankaCabinet": FrankaCabinetTask,
        "Humanoid": HumanoidLocomotionTask,
        "Ingenuity": IngenuityTask,
        "Quadcopter": QuadcopterTask,
        "ShadowHand": ShadowHandTask,
        "ShadowHandOpenAI_FF": ShadowHandTask,
        "ShadowHandOpenAI_LSTM": ShadowHandTask,
        "TomoDJCabinet": TomoDJCabinetTask,
        "TomoDJnewhandCabinet": TomoDJnewhandCabinetTask,
        "TomoDJnewhandTendonCabinet": TomoDJnewhandTendonCabinetTask,
        "TomoSPCCabinet": TomoSPCCabinetTask,
        "TomoSPCnewhandCabinet": TomoSPCnewhandCabinetTask,
    }, "task")
    # the Warp variants are the same math through another tracing compiler; the HIP kernels replace both
    task_map_warp = _TaskMap({}, "Warp task (the HIP path replaces the Warp variants: set warp=False)")
    return task_map, task_map_warp


def initialize_task(config: Dict[str, Any], env, init_sim: bool = True):
    """Build ``config["task_name"]`` and hand it to ``env`` (task_util.py:109-132): ``config`` is the resolved
    global config with the task YAML under ``config["task"]``."""
    from omniisaacgymenvs_amd.utils.config import SimConfig

    sim_config = SimConfig(config)
    task_map, task_map_warp = import_tasks()
    cfg = sim_config.config
    warp = bool(cfg.get("warp", False))
    if warp:
        task_map = task_map_warp
    task = task_map[cfg["task_name"]](name=cfg["task_name"], sim_config=sim_config, env=env)

    physics = sim_config.get_physics_params()
    env.set_task(task=task, sim_params=physics, backend="warp" if warp else "torch", init_sim=init_sim,
                 rendering_dt=physics["rendering_dt"])
    return task


def make_task(name: str, n: int, device: str = "cuda:0", seed: int = 42, **env_overrides):
    """``(env, task)`` of ``n`` envs through the reference's own front door: table config -> ``VecEnvRLGames`` ->
    ``initialize_task(config, env)`` (task_util.py:109-132).  Used by bench.py, smoke() and the tests; oracle-free."""
    from omniisaacgymenvs_amd.envs.vec_env_rlgames import VecEnvRLGames
    from omniisaacgymenvs_amd.utils.config import global_config, task_config

    cfg = global_config(sim_device=device, rl_device=device, seed=seed, task_name=name,
                        task=task_config(name, num_envs=n, **env_overrides))
    env = VecEnvRLGames(headless=True)
    task = initialize_task(cfg, env)
    return env, task
