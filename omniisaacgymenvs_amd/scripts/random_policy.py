"""The reference's smallest launcher (scripts/random_policy.py:48-130): compose the config, build the task through the
registry, drive ``VecEnvRLGames.step`` with uniform random actions and print frames per second.

    python -m omniisaacgymenvs_amd.scripts.random_policy task=Ant num_envs=4096 [--cfg-dir <reference>/cfg]
                                                         [--frames 2000] [--graph]

Arguments of the form ``key=value`` are hydra-style overrides.  With ``--cfg-dir`` the reference's own YAML files are
read (utils/hydra_cfg.py); without it the tables of utils/config.py stand in for them.  ``--graph`` replays the wrapper's
whole step as one HIP graph (``VecEnvRLGames.capture_step``); ``--drop-resets`` leaves the synthetic views' setters out
(at small sizes their torch scatters, stand-ins for PhysX's, cost more than the task math).  The physics step between the two task calls is the
synthetic world's (state stays as loaded), so the number printed is the rate of the task math and the wrapper alone.
"""
from __future__ import annotations

import argparse
import sys
import time
from typing import Any, Dict, List, Optional

import torch

from omniisaacgymenvs_amd.envs.vec_env_rlgames import VecEnvRLGames
from omniisaacgymenvs_amd.utils import hydra_cfg
from omniisaacgymenvs_amd.utils.config import global_config, task_config
from omniisaacgymenvs_amd.utils.seed import set_seed
from omniisaacgymenvs_amd.utils.task_util import initialize_task


def compose(cfg_dir: Optional[str], overrides: List[str]) -> Dict[str, Any]:
    ov = hydra_cfg._parse_overrides(overrides)
    if cfg_dir:
        key = ov.get("task", None)
        cfg = hydra_cfg.load_config(cfg_dir, dict(ov))
        if isinstance(key, str):
            cfg["task_name"] = key  # the registry key (task=ShadowHandOpenAI_FF names itself ShadowHand)
        return cfg
    name = ov.pop("task", "Cartpole")
    num_envs = ov.pop("num_envs", None)
    cfg = global_config(task_name=name, task=task_config(name, num_envs=num_envs or None), headless=True)
    for k, v in ov.items():
        hydra_cfg._set_dotted(cfg, k, v)
    return cfg


def run(cfg: Dict[str, Any], frames: int, graph: bool = False, report_every: float = 1.0, drop_resets: bool = False
        ) -> Dict[str, float]:
    env = VecEnvRLGames(headless=cfg.get("headless", True), sim_device=cfg.get("device_id", 0),
                        enable_livestream=cfg.get("enable_livestream", False))
    cfg["seed"] = set_seed(cfg.get("seed", 42), torch_deterministic=cfg.get("torch_deterministic", False))
    task = initialize_task(cfg, env)
    env.reset()
    high, low = float(env.action_space.high[0]), float(env.action_space.low[0])
    shape = (env.num_envs, env.action_space.shape[0])
    actions = torch.empty(shape, device=task.rl_device)
    if drop_resets:  # the setters stand in for PhysX's; dropping their input leaves the task math and the wrapper alone
        task.set_reset_scatter(False)
    if graph:
        actions.uniform_(low, high)
        if not drop_resets:
            task.set_reset_scatter("device")
        env.capture_step(actions)
    num_frames, total, prev = 0, 0, time.time()
    t0 = time.perf_counter()
    while total < frames and env.world.is_playing():
        actions.uniform_(low, high)
        if graph:
            env.replay_step()
        else:
            env.step(actions)
        num_frames += 1
        total += 1
        if time.time() - prev >= report_every:
            print("FPS:", num_frames, "FPS * num_envs:", env.num_envs * num_frames)
            num_frames, prev = 0, time.time()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    out = {"frames": total, "seconds": dt, "fps": total / dt, "env_steps_per_s": total * env.num_envs / dt}
    print(f"{total} frames of {env.num_envs} envs in {dt:.3f} s: {out['fps']:.0f} FPS, {out['env_steps_per_s']:.3e} env-steps/s")
    env.close()
    return out


def main(argv: Optional[List[str]] = None) -> int:
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("--cfg-dir", default=None, help="a cfg/ tree in the reference's format (config.yaml, task/, train/)")
    ap.add_argument("--frames", type=int, default=2000)
    ap.add_argument("--graph", action="store_true")
    ap.add_argument("--drop-resets", action="store_true",
                    help="do not scatter reset rows into the synthetic views (their setters emulate PhysX's, outside the path)")
    ap.add_argument("overrides", nargs="*", help="hydra-style key=value items, e.g. task=Ant num_envs=4096")
    args = ap.parse_args(argv)
    cfg = compose(args.cfg_dir, args.overrides)
    run(cfg, args.frames, graph=args.graph, drop_resets=args.drop_resets)
    return 0


if __name__ == "__main__":
    sys.exit(main())
