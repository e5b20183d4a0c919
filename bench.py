#!/usr/bin/env python3
"""bench.py -- fused obs+reward+reset env-steps/s (BASELINE.json metric).

    python bench.py [--gpus N] [--steps K] [--warmup W] [--envs E] [--task Ant|Humanoid]

One "step" = one pre_physics_step (reset-id compaction, reset of the flagged envs, action
scaling) + one post_physics_step (progress, obs, reward, done) over a batch of synthetic state
resident in HBM, through the RLTask API -> ctypes -> liboige_hip.so.  Physics is delegated
(north_star) and not part of the step.  N=1 workload: BASELINE.json configs[1], "Ant
obs(60)+reward+reset fused kernel, 1M envs synthetic root/DOF state on one MI355X".
For N>1 every rank owns its own 1M envs (weak scaling; envs are independent, so there is no
data-path collective); ranks all-reduce a 32-byte episode-statistics vector over RCCL.

Prints ONE JSON line (rank 0).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import torch  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec peak (MI355X_MICROARCH.md: 8.0 TB/s spec, ~6.3 achievable)


def algorithmic_bytes(d: int, s: int, with_vecs: bool) -> dict:
    """bytes that must cross HBM once per env-step."""
    obs = 12 + 3 * d + 6 * s
    post_r = 12 + 16 + 24 + 4 * d + 4 * d + 24 * s + 4 * d + 4 + 8 + 8
    post_w = 4 * obs + 4 + 4 + 4 + 8 + 8 + (24 if with_vecs else 0)
    pre = 4 * d + 4 * d + 4 * d  # actions in, forces out, actions clone out (locomotion.py:125)
    return {"post": post_r + post_w, "pre": pre, "obs": obs}


def cpu_baseline(task_name: str, n: int, budget_s: float = 12.0):
    """The oracle (CPU torch restatement of the reference) timed on the host cores: bounded sample."""
    from omniisaacgymenvs_amd.sim import workloads
    from oracle import locomotion_oracle as lo

    spec = lo.ant_spec() if task_name == "Ant" else lo.humanoid_spec()
    z = (0.2, 0.9) if task_name == "Ant" else (0.5, 1.5)
    st = workloads.locomotion_state(n, spec.dof_limits_lower, spec.dof_limits_upper, spec.num_sensors, "cpu", 1234, z)
    b = lo.LocomotionBuffers.fresh(spec, n)
    b.reset_buf[:] = 0
    b.progress_buf[:] = st["progress"]
    b.potentials[:] = st["potentials"]
    init_pos, init_rot = st["pos"].clone(), st["rot"].clone()
    init_dof = torch.zeros(n, spec.num_dof)
    g = torch.Generator().manual_seed(0)

    def draw(ids):
        return (torch.rand(len(ids), spec.num_dof, generator=g), torch.rand(len(ids), spec.num_dof, generator=g))

    def step():
        lo.pre_physics(spec, b, st["actions"], init_pos, init_rot, init_dof, draw=draw)
        lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])

    step()  # warm-up
    t0 = time.perf_counter()
    k = 0
    while True:
        step()
        k += 1
        if time.perf_counter() - t0 > budget_s or k >= 20:
            break
    dt = (time.perf_counter() - t0) / k
    return {
        "value": n / dt, "unit": "env-steps/s", "cores": torch.get_num_threads(), "kind": "port",
        "sample": f"{task_name} oracle (CPU torch restatement) pre+post, {n} envs, {k} steps, {dt*1e3:.1f} ms/step",
    }


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--envs", type=int, default=1 << 20, help="envs per GPU")
    ap.add_argument("--task", default="Ant", choices=["Ant", "Humanoid"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-envs", type=int, default=1 << 18)
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{local_rank}"))
    else:
        dist = None
        torch.cuda.set_device(0)
    dev = f"cuda:{local_rank if world > 1 else 0}"

    from helpers import make_task
    from omniisaacgymenvs_amd.sim import workloads

    n = args.envs
    env, task = make_task(args.task, n, device=dev, seed=1234 + rank)
    d, s = task._robots.num_dof, task._num_sensors
    z = (0.2, 0.9) if args.task == "Ant" else (0.5, 1.5)
    st = workloads.locomotion_state(n, task.dof_limits_lower, task.dof_limits_upper, s, device=dev, seed=1234 + rank, z_range=z)
    workloads.load_locomotion_state(task, st)
    task._robots.apply_resets = False  # PhysX-side scatter is delegated; the kernels still compute every reset row
    actions = st["actions"]
    stats = torch.zeros(8, dtype=torch.float32, device=dev)

    def step():
        task.pre_physics_step(actions)
        task.post_physics_step()

    ev_pairs = []

    def timed_step():
        task.pre_physics_step(actions)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        task.post_physics_step()
        e1.record()
        ev_pairs.append((e0, e1))

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        timed_step()
    if dist is not None:
        # episode statistics: [sum reward, n done, sum episode length of done envs, ...] -- 32 bytes
        stats[0] = task.rew_buf.sum()
        stats[1] = task.reset_buf.sum()
        stats[2] = (task.progress_buf * task.reset_buf).sum()
        dist.all_reduce(stats)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    post_ms = sum(a.elapsed_time(b) for a, b in ev_pairs) / len(ev_pairs)
    ab = algorithmic_bytes(d, s, task.materialize_vecs)
    achieved = ab["post"] * n / (post_ms * 1e-3) / 1e9
    resets_per_step = int(task._reset_out["count"].item())

    if rank == 0:
        out = {
            "metric": "fused obs+reward+reset env-steps/s",
            "value": world * n * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f32",
            "data": "synthetic",
            "config": {
                "workload": f"{args.task} obs({ab['obs']})+reward+reset fused, {n} envs per GPU, synthetic root/DOF state",
                "envs_per_gpu": n, "num_dof": d, "num_sensors": s, "resets_per_step": resets_per_step,
                "step": "pre_physics_step + post_physics_step via RLTask API (physics delegated)",
            },
            "roofline": {
                "bound": "hbm", "kernel": "locomotion_post_kernel", "achieved": achieved, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                "algorithmic_bytes_per_env": ab["post"], "avg_kernel_ms": post_ms,
            },
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(args.task, min(n, args.cpu_envs))
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
