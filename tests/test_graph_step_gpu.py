"""-m gpu: LocomotionTask.capture_step / replay_step (one HIP graph per step, for launch-bound sizes) is bit-identical to the
eager sync-free step -- same kernels, same Philox keys (the step counter moves to a device scalar) -- and cheaper on the host."""
import json
import os
import time

import pytest
import torch

from helpers import assert_equal, make_task
from omniisaacgymenvs_amd.sim import workloads

pytestmark = pytest.mark.gpu


def _task(name, n, mode):
    torch.manual_seed(7)
    env, task = make_task(name, n)
    z = (0.2, 0.9) if name == "Ant" else (0.5, 1.5)
    st = workloads.locomotion_state(n, task.dof_limits_lower, task.dof_limits_upper, task._num_sensors, device=task.device, seed=11, z_range=z)
    workloads.load_locomotion_state(task, st)
    task._robots.apply_resets = mode
    return task, st


def _snapshot(task):
    r = task._robots
    return {k: v.clone() for k, v in dict(obs=task.obs_buf, rew=task.rew_buf, reset=task.reset_buf, progress=task.progress_buf,
                                          potentials=task.potentials, prev=task.prev_potentials, actions=task.actions, forces=task._forces,
                                          q=r.joint_positions, qd=r.joint_velocities, pos=r.positions, rot=r.orientations,
                                          vel=r.velocities, ids=task._reset_out["env_ids"], count=task._reset_out["count"]).items()}


@pytest.mark.parametrize("name,n", [("Ant", 4096), ("Ant", 300), ("Humanoid", 1000)])
def test_graph_replay_equals_eager_steps(name, n):
    eager, st = _task(name, n, True)        # host-applied reset rows (reads count)
    graph, _ = _task(name, n, "device")     # device-applied reset rows (no read-back: legal under capture)
    g = torch.Generator().manual_seed(3)
    acts = [(torch.rand(n, eager._robots.num_dof, generator=g) * 2 - 1).to(eager.device) for _ in range(6)]
    static = acts[0].clone()
    graph.capture_step(static)              # runs step 0 eagerly (warm-up), captures the step
    eager.pre_physics_step(acts[0])
    eager.post_physics_step()
    total_resets = 0
    for a in acts[1:]:
        static.copy_(a)
        obs, rew, reset, _ = graph.replay_step()
        eager.pre_physics_step(a)
        eager.post_physics_step()
        torch.cuda.synchronize()
        A, B = _snapshot(eager), _snapshot(graph)
        k = int(A["count"])
        total_resets += k
        assert int(B["count"]) == k
        for key in A:
            if key == "ids":
                assert_equal(B[key][:k], A[key][:k], key)
            else:
                assert_equal(B[key], A[key], key)
        assert obs.data_ptr() == graph.obs_buf.data_ptr()
    assert total_resets > 0 and graph._step_count == eager._step_count
    # an eager step in between keeps the device counter in step with the host's
    graph.pre_physics_step(acts[1]); graph.post_physics_step()
    eager.pre_physics_step(acts[1]); eager.post_physics_step()
    static.copy_(acts[2])
    graph.replay_step()
    eager.pre_physics_step(acts[2]); eager.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(graph.obs_buf, eager.obs_buf, "obs after a mixed eager/replay sequence")
    assert_equal(graph._robots.joint_positions, eager._robots.joint_positions, "joint positions after a mixed sequence")


def test_capture_needs_the_sync_free_path():
    task, st = _task("Ant", 64, "device")
    task.sync_free = False
    with pytest.raises(RuntimeError, match="sync-free"):
        task.capture_step(st["actions"])


def test_replay_cost_at_the_reference_size():
    """Measurement: wall time per step at 4096 envs (the reference's Ant default), eager mirror vs graph replay."""
    n, steps = 4096, 2000
    out = {}
    # views: False = the reset rows are handed over and dropped (the task math alone, as bench.py measures it);
    # "device" = the synthetic view scatters them with ~25 small torch kernels per step (its cost, in both columns)
    for name, mode in (("Ant", False), ("Humanoid", False), ("Ant", "device")):
        task, st = _task(name, n, mode)
        a = st["actions"].clone()
        for _ in range(20):
            task.pre_physics_step(a); task.post_physics_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            task.pre_physics_step(a); task.post_physics_step()
        torch.cuda.synchronize()
        eager_us = (time.perf_counter() - t0) / steps * 1e6
        task.capture_step(a)
        for _ in range(20):
            task.replay_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            task.replay_step()
        torch.cuda.synchronize()
        out[f"{name}, view scatter {mode}"] = {"envs": n, "eager_us_per_step": eager_us, "graph_replay_us_per_step": (time.perf_counter() - t0) / steps * 1e6}
    print(json.dumps(out))
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "graph_step_timing.json"), "w") as f:
            json.dump(out, f, indent=1)
    for rec in out.values():
        assert rec["graph_replay_us_per_step"] < rec["eager_us_per_step"]
