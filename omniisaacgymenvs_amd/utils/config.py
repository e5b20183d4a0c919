"""Task / global configuration without hydra (hydra & omegaconf are not part of this path).

The reference reads its scalars from ``cfg/task/<Task>.yaml`` through
``sim_config.task_config["env"][...]`` (e.g. tasks/shared/locomotion.py:50-65).  The
dictionaries below carry the same keys and the YAML's values (file:line cited), so the task
classes read them exactly the way the reference does.  ``SimConfig`` is the small slice of
utils/config_utils/sim_config.py the task constructors touch: ``.config`` and ``.task_config``.
"""
from __future__ import annotations

import copy
from typing import Any, Dict, Optional


def global_config(**overrides: Any) -> Dict[str, Any]:
    """cfg/config.yaml globals the hot path reads (rl_task.py:72-77,104; vec_env_rlgames.py:58-68)."""
    cfg: Dict[str, Any] = {
        "seed": 42,
        "test": False,
        "sim_device": "cuda:0",
        "rl_device": "cuda:0",
        "headless": True,
        "enable_livestream": False,
        "enable_recording": False,
        "recording_interval": 2000,
        "recording_length": 100,
        "multi_gpu": False,
        "warp": False,
        "task_name": "Cartpole",
    }
    cfg.update(overrides)
    return cfg


_SIM = {"dt": 0.0083, "add_ground_plane": True, "add_distant_light": False, "enable_cameras": False}

TASK_DEFAULTS: Dict[str, Dict[str, Any]] = {
    # cfg/task/Cartpole.yaml:7-14
    "Cartpole": {
        "name": "Cartpole",
        "env": {
            "numEnvs": 512, "envSpacing": 4.0, "resetDist": 3.0, "maxEffort": 400.0,
            "clipObservations": 5.0, "clipActions": 1.0, "controlFrequencyInv": 2,
        },
        "sim": dict(_SIM),
    },
    # cfg/task/Ant.yaml:8-32
    "Ant": {
        "name": "Ant",
        "env": {
            "numEnvs": 4096, "envSpacing": 5, "episodeLength": 1000, "enableDebugVis": False,
            "clipActions": 1.0, "powerScale": 0.5, "controlFrequencyInv": 2,
            "headingWeight": 0.5, "upWeight": 0.1,
            "actionsCost": 0.005, "energyCost": 0.05, "dofVelocityScale": 0.2, "angularVelocityScale": 1.0,
            "contactForceScale": 0.1, "jointsAtLimitCost": 0.1, "deathCost": -2.0, "terminationHeight": 0.31,
            "alive_reward_scale": 0.5,
        },
        "sim": dict(_SIM),
    },
    # cfg/task/Humanoid.yaml:8-32
    "Humanoid": {
        "name": "Humanoid",
        "env": {
            "numEnvs": 4096, "envSpacing": 5, "episodeLength": 1000, "enableDebugVis": False,
            "clipActions": 1.0, "powerScale": 1.0, "controlFrequencyInv": 2,
            "headingWeight": 0.5, "upWeight": 0.1,
            "actionsCost": 0.01, "energyCost": 0.05, "dofVelocityScale": 0.1, "angularVelocityScale": 0.25,
            "contactForceScale": 0.01, "jointsAtLimitCost": 0.25, "deathCost": -1.0, "terminationHeight": 0.8,
            "alive_reward_scale": 2.0,
        },
        "sim": dict(_SIM),
    },
}

# cfg/task/ShadowHand.yaml:7-49, :139-158 (domain_randomization: obs/action noise part only)
_SHADOW_HAND_ENV = {
    "numEnvs": 8192, "envSpacing": 0.75, "episodeLength": 600,
    "clipObservations": 5.0, "clipActions": 1.0,
    "useRelativeControl": False, "dofSpeedScale": 20.0, "actionsMovingAverage": 1.0, "controlFrequencyInv": 2,
    "startPositionNoise": 0.01, "startRotationNoise": 0.0,
    "resetPositionNoise": 0.01, "resetRotationNoise": 0.0,
    "resetDofPosRandomInterval": 0.2, "resetDofVelRandomInterval": 0.0,
    "forceScale": 0.0, "forceProbRange": [0.001, 0.1], "forceDecay": 0.99, "forceDecayInterval": 0.08,
    "distRewardScale": -10.0, "rotRewardScale": 1.0, "rotEps": 0.1, "actionPenaltyScale": -0.0002,
    "reachGoalBonus": 250, "fallDistance": 0.24, "fallPenalty": 0.0, "velObsScale": 0.2,
    "objectType": "block", "observationType": "full", "asymmetric_observations": False,
    "successTolerance": 0.1, "printNumSuccesses": False, "maxConsecutiveSuccesses": 0,
}
_SHADOW_HAND_DR = {
    "randomize": False,
    "min_frequency": 720,
    "randomization_params": {
        "observations": {
            "on_reset": {"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0, 0.0001]},
            "on_interval": {"frequency_interval": 1, "operation": "additive", "distribution": "gaussian",
                            "distribution_parameters": [0, 0.002]},
        },
        "actions": {
            "on_reset": {"operation": "additive", "distribution": "gaussian", "distribution_parameters": [0, 0.015]},
            "on_interval": {"frequency_interval": 1, "operation": "additive", "distribution": "gaussian",
                            "distribution_parameters": [0.0, 0.05]},
        },
    },
}
TASK_DEFAULTS["ShadowHand"] = {"name": "ShadowHand", "env": dict(_SHADOW_HAND_ENV), "sim": dict(_SIM),
                              "domain_randomization": copy.deepcopy(_SHADOW_HAND_DR)}
# cfg/task/ShadowHandOpenAI_FF.yaml:7-51 (resetTime 8 s, 20 Hz control, asymmetric states, mcs 50)
_OPENAI_FF_ENV = dict(_SHADOW_HAND_ENV)
_OPENAI_FF_ENV.update({
    "episodeLength": 160, "resetTime": 8, "actionsMovingAverage": 0.3, "controlFrequencyInv": 3, "forceScale": 1.0,
    "fallPenalty": -50.0, "observationType": "openai", "asymmetric_observations": True, "successTolerance": 0.4,
    "maxConsecutiveSuccesses": 50, "averFactor": 0.1,
})
TASK_DEFAULTS["ShadowHandOpenAI_FF"] = {"name": "ShadowHand", "env": _OPENAI_FF_ENV, "sim": dict(_SIM, dt=0.016667),  # :55
                                       "domain_randomization": copy.deepcopy(_SHADOW_HAND_DR)}
TASK_DEFAULTS["ShadowHandOpenAI_FF"]["domain_randomization"]["randomize"] = True  # ShadowHandOpenAI_FF.yaml:138 (ShadowHand.yaml: False)

# cfg/task/ShadowHandOpenAI_LSTM.yaml:3-8: the FF task YAML unchanged (the LSTM lives in the train YAML, out of scope)
TASK_DEFAULTS["ShadowHandOpenAI_LSTM"] = copy.deepcopy(TASK_DEFAULTS["ShadowHandOpenAI_FF"])

# cfg/task/AllegroHand.yaml:7-42 (no asymmetric observations, no force perturbation keys, no domain_randomization section)
_ALLEGRO_HAND_ENV = {k: v for k, v in _SHADOW_HAND_ENV.items()
                     if k not in ("forceScale", "forceProbRange", "forceDecay", "forceDecayInterval", "asymmetric_observations")}
_ALLEGRO_HAND_ENV.update({"controlFrequencyInv": 4})
TASK_DEFAULTS["AllegroHand"] = {"name": "AllegroHand", "env": _ALLEGRO_HAND_ENV, "sim": dict(_SIM, dt=0.0083)}

# cfg/task/Quadcopter.yaml:7-17
TASK_DEFAULTS["Quadcopter"] = {
    "name": "Quadcopter",
    "env": {"numEnvs": 4096, "envSpacing": 1.25, "maxEpisodeLength": 500, "enableDebugVis": False,
            "clipObservations": 5.0, "clipActions": 1.0},
    "sim": dict(_SIM, dt=0.01),
}

# cfg/task/BallBalance.yaml:7-13
TASK_DEFAULTS["BallBalance"] = {
    "name": "BallBalance",
    "env": {"numEnvs": 4096, "envSpacing": 2.0, "maxEpisodeLength": 600, "actionSpeedScale": 20,
            "clipObservations": 5.0, "clipActions": 1.0},
    "sim": dict(_SIM, dt=0.01),
}

# cfg/task/Ingenuity.yaml:5-13
TASK_DEFAULTS["Ingenuity"] = {
    "name": "Ingenuity",
    "env": {"numEnvs": 4096, "envSpacing": 2.5, "maxEpisodeLength": 2000, "enableDebugVis": False,
            "clipObservations": 5.0, "clipActions": 1.0},
    "sim": dict(_SIM, dt=0.01, gravity=[0.0, 0.0, -3.721]),
}

# cfg/task/Crazyflie.yaml:5-13
TASK_DEFAULTS["Crazyflie"] = {
    "name": "Crazyflie",
    "env": {"numEnvs": 4096, "envSpacing": 2.5, "maxEpisodeLength": 700, "enableDebugVis": False,
            "clipObservations": 5.0, "clipActions": 1.0},
    "sim": dict(_SIM, dt=0.01, gravity=[0.0, 0.0, -9.81]),
}

# cfg/task/FrankaCabinet.yaml:7-32
TASK_DEFAULTS["FrankaCabinet"] = {
    "name": "FrankaCabinet",
    "env": {"numEnvs": 4096, "envSpacing": 3.0, "episodeLength": 500, "enableDebugVis": False, "clipObservations": 5.0,
            "clipActions": 1.0, "controlFrequencyInv": 2, "startPositionNoise": 0.0, "startRotationNoise": 0.0, "numProps": 4,
            "aggregateMode": 3, "actionScale": 7.5, "dofVelocityScale": 0.1, "actionPenaltyScale": 0.01,
            "aroundHandleRewardScale": 10.0, "distRewardScale": 2.0, "fingerCloseRewardScale": 10.0,
            "fingerDistRewardScale": 100.0, "openRewardScale": 7.5, "rotRewardScale": 0.5},
    "sim": dict(_SIM, dt=0.0083, gravity=[0.0, 0.0, -9.81]),
}

# cfg/task/TomoDJCabinet.yaml, cfg/task/TomoSPCCabinet.yaml: FrankaCabinet's keys with their own reward scales
TASK_DEFAULTS["TomoDJCabinet"] = {
    "name": "TomoDJCabinet",
    "env": dict(TASK_DEFAULTS["FrankaCabinet"]["env"], aroundHandleRewardScale=2, distRewardScale=5, fingerCloseRewardScale=10,
                fingerDistRewardScale=15, openRewardScale=15, rotRewardScale=1.5),
    "sim": dict(_SIM, dt=0.0083, gravity=[0.0, 0.0, -9.81]),
}
TASK_DEFAULTS["TomoSPCCabinet"] = {
    "name": "TomoSPCCabinet",
    "env": dict(TASK_DEFAULTS["FrankaCabinet"]["env"], aroundHandleRewardScale=2, distRewardScale=5, fingerCloseRewardScale=9,
                fingerDistRewardScale=13, openRewardScale=20, rotRewardScale=1.25),
    "sim": dict(_SIM, dt=0.0083, gravity=[0.0, 0.0, -9.81]),
}

# cfg/task/TomoDJnewhandCabinet.yaml, TomoDJnewhandTendonCabinet.yaml, TomoSPCnewhandCabinet.yaml (identical scalars)
for _name in ("TomoDJnewhandCabinet", "TomoDJnewhandTendonCabinet", "TomoSPCnewhandCabinet"):
    TASK_DEFAULTS[_name] = {
        "name": _name,
        "env": dict(TASK_DEFAULTS["FrankaCabinet"]["env"], aroundHandleRewardScale=2, distRewardScale=10, fingerCloseRewardScale=10,
                    fingerDistRewardScale=15, openRewardScale=12.5, rotRewardScale=3),
        "sim": dict(_SIM, dt=0.0083, gravity=[0.0, 0.0, -9.81]),
    }

# cfg/task/AnymalTerrain.yaml:5-103
TASK_DEFAULTS["AnymalTerrain"] = {
    "name": "AnymalTerrain",
    "env": {
        "numEnvs": 2048, "numObservations": 188, "numActions": 12, "envSpacing": 3.0,
        "terrain": {
            "staticFriction": 1.0, "dynamicFriction": 1.0, "restitution": 0.0, "curriculum": True, "maxInitMapLevel": 0,
            "mapLength": 8.0, "mapWidth": 8.0, "numLevels": 10, "numTerrains": 20,
            "terrainProportions": [0.1, 0.1, 0.35, 0.25, 0.2], "slopeTreshold": 0.5,
        },
        "baseInitState": {"pos": [0.0, 0.0, 0.62], "rot": [1.0, 0.0, 0.0, 0.0], "vLinear": [0.0, 0.0, 0.0],
                          "vAngular": [0.0, 0.0, 0.0]},
        "randomCommandVelocityRanges": {"linear_x": [-1.0, 1.0], "linear_y": [-1.0, 1.0], "yaw": [-3.14, 3.14]},
        "control": {"stiffness": 80.0, "damping": 2.0, "actionScale": 0.5, "decimation": 4},
        "defaultJointAngles": {
            "LF_HAA": 0.03, "LH_HAA": 0.03, "RF_HAA": -0.03, "RH_HAA": -0.03,
            "LF_HFE": 0.4, "LH_HFE": -0.4, "RF_HFE": 0.4, "RH_HFE": -0.4,
            "LF_KFE": -0.8, "LH_KFE": 0.8, "RF_KFE": -0.8, "RH_KFE": 0.8,
        },
        "learn": {
            "terminalReward": 0.0, "linearVelocityXYRewardScale": 1.0, "linearVelocityZRewardScale": -4.0,
            "angularVelocityXYRewardScale": -0.05, "angularVelocityZRewardScale": 0.5, "orientationRewardScale": -0.0,
            "torqueRewardScale": -0.00002, "jointAccRewardScale": -0.0005, "baseHeightRewardScale": -0.0,
            "actionRateRewardScale": -0.01, "fallenOverRewardScale": -1.0, "hipRewardScale": -0.0,
            "linearVelocityScale": 2.0, "angularVelocityScale": 0.25, "dofPositionScale": 1.0, "dofVelocityScale": 0.05,
            "heightMeasurementScale": 5.0,
            "addNoise": True, "noiseLevel": 1.0, "dofPositionNoise": 0.01, "dofVelocityNoise": 1.5,
            "linearVelocityNoise": 0.1, "angularVelocityNoise": 0.2, "gravityNoise": 0.05, "heightMeasurementNoise": 0.06,
            "pushInterval_s": 15, "episodeLength_s": 20,
        },
    },
    "sim": {"dt": 0.005, "add_ground_plane": False, "add_distant_light": False, "enable_cameras": False,
            "default_physics_material": {"static_friction": 1.0, "dynamic_friction": 1.0, "restitution": 0.0}},
}

# cfg/task/Anymal.yaml:7-66
TASK_DEFAULTS["Anymal"] = {
    "name": "Anymal",
    "env": {
        "numEnvs": 4096, "envSpacing": 4.0, "clipObservations": 5.0, "clipActions": 1.0, "controlFrequencyInv": 2,
        "baseInitState": {"pos": [0.0, 0.0, 0.62], "rot": [0.0, 0.0, 0.0, 1.0], "vLinear": [0.0, 0.0, 0.0],
                          "vAngular": [0.0, 0.0, 0.0]},
        "randomCommandVelocityRanges": {"linear_x": [-2.0, 2.0], "linear_y": [-1.0, 1.0], "yaw": [-1.0, 1.0]},
        "control": {"stiffness": 85.0, "damping": 2.0, "actionScale": 13.5},
        "defaultJointAngles": {
            "LF_HAA": 0.03, "LH_HAA": 0.03, "RF_HAA": -0.03, "RH_HAA": -0.03,
            "LF_HFE": 0.4, "LH_HFE": -0.4, "RF_HFE": 0.4, "RH_HFE": -0.4,
            "LF_KFE": -0.8, "LH_KFE": 0.8, "RF_KFE": -0.8, "RH_KFE": 0.8,
        },
        "learn": {
            "linearVelocityXYRewardScale": 1.0, "angularVelocityZRewardScale": 0.5, "linearVelocityZRewardScale": -0.03,
            "jointAccRewardScale": -0.0003, "actionRateRewardScale": -0.006, "cosmeticRewardScale": -0.06,
            "linearVelocityScale": 2.0, "angularVelocityScale": 0.25, "dofPositionScale": 1.0, "dofVelocityScale": 0.05,
            "episodeLength_s": 50,
        },
    },
    "sim": {"dt": 0.01, "add_ground_plane": True, "add_distant_light": False, "enable_cameras": False},
}


def task_config(name: str, num_envs: Optional[int] = None, **env_overrides: Any) -> Dict[str, Any]:
    cfg = copy.deepcopy(TASK_DEFAULTS[name])
    if num_envs is not None:
        cfg["env"]["numEnvs"] = int(num_envs)
    for top in ("domain_randomization", "sim"):  # top-level sections of the task YAML; the rest is env.*
        if top in env_overrides:
            cfg[top] = copy.deepcopy(env_overrides.pop(top))
    cfg["env"].update(env_overrides)
    return cfg


class SimConfig:
    """``sim_config.config`` / ``sim_config.task_config`` as the task constructors use them
    (tasks/ant.py:52-55).  PhysX parameter parsing (the rest of the reference class) is out of scope."""

    def __init__(self, config: Optional[Dict[str, Any]] = None, task_config_: Optional[Dict[str, Any]] = None):
        self._config = config if config is not None else global_config()
        if task_config_ is None:  # the reference's form: SimConfig(config) with the task YAML under config["task"]
            task_config_ = self._config.get("task", {})
        self._task_config = task_config_
        self._config.setdefault("task", self._task_config)
        self._sanitize_device()

    def _sanitize_device(self) -> None:
        """cfg/config.yaml:22-26 spells the device 'gpu' / 'cpu' + device_id; the reference rewrites it from the task's
        pipeline flag before any task reads it (sim_config.py:152-163).  Torch spellings pass through."""
        dev = self._config.get("sim_device")
        if dev in ("gpu", "cpu"):
            gpu_pipeline = self._task_config.get("sim", {}).get("use_gpu_pipeline", dev == "gpu")
            self._config["sim_device"] = f"cuda:{self._config.get('device_id', 0)}" if gpu_pipeline else "cpu"

    @property
    def config(self) -> Dict[str, Any]:
        return self._config

    @property
    def task_config(self) -> Dict[str, Any]:
        return self._task_config

    def get_physics_params(self) -> Dict[str, Any]:
        """The ``sim`` section of the task YAML plus ``rendering_dt`` (sim_config.py: get_physics_params, as
        task_util.py:124-131 reads it); nothing here drives a solver, the synthetic world ignores it."""
        sim = dict(self._task_config.get("sim", {}))
        sim.setdefault("rendering_dt", sim.get("dt", 1.0 / 60.0))
        return sim

    # no-ops kept so reference-style task code can call them
    def apply_articulation_settings(self, *a, **k) -> None:
        return None

    def parse_actor_config(self, actor_name: str) -> Dict[str, Any]:
        return {}
