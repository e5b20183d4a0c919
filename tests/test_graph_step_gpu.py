"""-m gpu: RLTask.capture_step / replay_step (one HIP graph per step, for launch-bound sizes) is bit-identical to the eager
sync-free step -- same kernels, same Philox keys (the step counter moves to a device scalar, reset_io.step_ptr) -- and cheaper
on the host.  Every tensor the task and its views hold is compared after every step."""
import json
import os
import sys
import time

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402  (the adapters build a task with a loaded synthetic state)
from helpers import assert_equal  # noqa: E402

pytestmark = pytest.mark.gpu
CASES = [("Ant", 4096), ("Ant", 300), ("Humanoid", 1000), ("ShadowHand", 1000), ("AllegroHand", 700), ("Anymal", 1000), ("Quadcopter", 900),
         ("BallBalance", 900), ("Ingenuity", 900), ("Crazyflie", 900), ("FrankaCabinet", 900), ("TomoSPCCabinet", 500),
         ("TomoDJnewhandTendonCabinet", 500), ("AnymalTerrain", 700)]


def _views(task):
    out = {}
    for k, v in vars(task).items():
        if hasattr(v, "apply_resets"):
            out[k] = v
            for kk, vv in vars(v).items():
                if hasattr(vv, "apply_resets"):
                    out[f"{k}.{kk}"] = vv
    return out


def _build(name, n, mode):
    torch.manual_seed(7)
    b = bench.ADAPTERS[name](name, n, "cuda:0", 11)
    for v in _views(b.task).values():
        v.apply_resets = mode
    return b


def _tensors(task):
    out = {}
    for k, v in vars(task).items():
        if isinstance(v, torch.Tensor) and k not in ("_step_dev", "_graph_actions", "_noise4"):
            out[k] = v
        elif isinstance(v, dict) and k == "_reset_out":
            out.update({f"_reset_out.{kk}": vv for kk, vv in v.items() if "count" in kk})
    for name, view in _views(task).items():
        for k, v in vars(view).items():
            if isinstance(v, torch.Tensor):
                out[f"{name}.{k}"] = v
    return out


@pytest.mark.parametrize("name,n", [("Ant", 1000), ("ShadowHand", 500), ("FrankaCabinet", 500)])
def test_split_capture_equals_eager_steps(name, n):
    """Two graphs (pre / post), for a physics step in between: same results as the eager calls."""
    eager, graph = _build(name, n, True), _build(name, n, "device")
    g = torch.Generator().manual_seed(4)
    acts = [(torch.rand(n, eager.task.num_actions, generator=g) * 2 - 1).to("cuda:0") for _ in range(4)]
    static = acts[0].clone()
    te, tg = eager.task, graph.task
    tg.capture_step(static, split=True)
    te.pre_physics_step(acts[0])
    te.post_physics_step()
    for a in acts[1:]:
        static.copy_(a)
        tg.replay_pre()
        te.pre_physics_step(a)
        torch.cuda.synchronize()
        assert_equal(tg.reset_buf, te.reset_buf, "reset_buf after pre")
        tg.replay_post()
        te.post_physics_step()
        torch.cuda.synchronize()
        A_, B_ = _tensors(te), _tensors(tg)
        for key in A_:
            assert_equal(B_[key], A_[key], key)


@pytest.mark.parametrize("name,n", CASES)
def test_graph_replay_equals_eager_steps(name, n):
    eager, graph = _build(name, n, True), _build(name, n, "device")   # host-applied vs device-applied reset rows
    g = torch.Generator().manual_seed(3)
    A = eager.task.num_actions
    acts = [(torch.rand(n, A, generator=g) * 2 - 1).to("cuda:0") for _ in range(6)]
    static = acts[0].clone()
    te, tg = eager.task, graph.task
    if name == "AnymalTerrain":          # push_robots on every other step: the device-side branch must replay, not freeze
        te.push_interval = tg.push_interval = 2
    torch.manual_seed(99)
    tg.capture_step(static)              # runs step 0 eagerly (warm-up), captures the step
    torch.manual_seed(99)
    te.pre_physics_step(acts[0])
    te.post_physics_step()
    total_resets = 0
    for a in acts[1:]:
        static.copy_(a)
        torch.manual_seed(5)             # Crazyflie draws its thrust noise with torch: same generator state on both sides
        obs, rew, reset, _ = tg.replay_step()
        torch.manual_seed(5)
        te.pre_physics_step(a)
        te.post_physics_step()
        torch.cuda.synchronize()
        total_resets += int(te._reset_out["count"])
        A_, B_ = _tensors(te), _tensors(tg)
        assert set(A_) == set(B_)
        for key in A_:
            assert_equal(B_[key], A_[key], key)
        assert obs.data_ptr() == tg.obs_buf.data_ptr()
    assert total_resets > 0
    if hasattr(te, "_step_count"):   # tasks with in-kernel Philox draws (Crazyflie has none)
        assert tg._step_count == te._step_count
    # an eager step in between keeps the device counter in step with the host's
    torch.manual_seed(6); tg.pre_physics_step(acts[1]); tg.post_physics_step()
    torch.manual_seed(6); te.pre_physics_step(acts[1]); te.post_physics_step()
    static.copy_(acts[2])
    torch.manual_seed(8); tg.replay_step()
    torch.manual_seed(8); te.pre_physics_step(acts[2]); te.post_physics_step()
    torch.cuda.synchronize()
    assert_equal(tg.obs_buf, te.obs_buf, "obs after a mixed eager/replay sequence")
    assert_equal(tg.reset_buf, te.reset_buf, "reset_buf after a mixed eager/replay sequence")


def test_capture_needs_the_sync_free_path_and_a_capturable_task():
    b = _build("Ant", 64, "device")
    b.task.sync_free = False
    with pytest.raises(RuntimeError, match="sync-free"):
        b.task.capture_step(b.actions)
    t = _build("AnymalTerrain", 64, False)
    t.task.device_push = False           # the reference's host-side push_robots branch cannot be frozen into a graph
    with pytest.raises(NotImplementedError, match="host-side branch"):
        t.task.capture_step(t.actions)


def test_replay_cost_at_the_reference_size():
    """Measurement: wall time per step at 4096 envs (the reference's default size), eager mirror vs graph replay; the reset rows
    are handed over and dropped (the task math alone, as bench.py measures it)."""
    n, steps = 4096, 2000
    out = {}
    for name in ("Ant", "Humanoid", "ShadowHand", "AllegroHand", "Anymal", "Quadcopter", "BallBalance", "Ingenuity", "Crazyflie", "FrankaCabinet",
                 "TomoDJnewhandCabinet"):
        b = _build(name, n, False)
        task, a = b.task, b.actions.clone()
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
        out[name] = {"envs": n, "eager_us_per_step": round(eager_us, 2), "graph_replay_us_per_step": round((time.perf_counter() - t0) / steps * 1e6, 2)}
    for name in ("ShadowHand", "Ant"):   # the wrapper's whole step (ShadowHand: DR noise on observations and actions + clamps)
        b = _build(name, n, False)
        env, a = b.env, b.actions.clone()
        for _ in range(20):
            env.step(a)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.step(a)
        torch.cuda.synchronize()
        eager_us = (time.perf_counter() - t0) / steps * 1e6
        env.capture_step(a)
        for _ in range(20):
            env.replay_step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            env.replay_step()
        torch.cuda.synchronize()
        out[f"{name} VecEnvRLGames.step"] = {"envs": n, "eager_us_per_step": round(eager_us, 2),
                                             "graph_replay_us_per_step": round((time.perf_counter() - t0) / steps * 1e6, 2)}
    print(json.dumps(out))
    d = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(d):
        with open(os.path.join(d, "graph_step_timing.json"), "w") as f:
            json.dump(out, f, indent=1)
    for rec in out.values():   # a measurement, not a gate: only a gross regression fails (the boxes are shared, timings jitter,
        assert rec["graph_replay_us_per_step"] < 5.0 * rec["eager_us_per_step"]   # and the eager step is ~20 us of host work now)


@pytest.mark.parametrize("cfg,n", [("ShadowHand", 700), ("AllegroHand", 300), ("Ant", 1000), ("FrankaCabinet", 500)])
def test_wrapper_capture_equals_eager_env_steps(cfg, n):
    """VecEnvRLGames.capture_step: action noise / clamp + task step + observation noise / clamp in one graph, bit-identical to
    eager env.step() calls (ShadowHand carries the reference's DR noise on both observations and actions)."""
    def build(mode):
        b = _build(cfg, n, mode)   # the bench adapter: a task with a loaded synthetic state (resets every step)
        return b.env, b.task

    (ee, te), (eg, tg) = build(True), build("device")
    g = torch.Generator().manual_seed(5)
    acts = [(torch.rand(n, te.num_actions, generator=g) * 3 - 1.5).to("cuda:0") for _ in range(6)]
    static = acts[0].clone()
    eg.capture_step(static)
    ee.step(acts[0].clone())
    resets_seen = 0
    for a in acts[1:]:
        static.copy_(a)
        og, rg, dg, _ = eg.replay_step()
        oe, re_, de, _ = ee.step(a.clone())
        torch.cuda.synchronize()
        assert_equal(og["obs"], oe["obs"], "obs")
        assert_equal(rg, re_, "rew")
        assert_equal(dg, de, "resets")
        if tg.num_states > 0:
            assert_equal(og["states"], oe["states"], "states")
        A_, B_ = _tensors(te), _tensors(tg)
        for key in A_:
            assert_equal(B_[key], A_[key], key)
        resets_seen += int(de.sum())
    if cfg in ("ShadowHand", "Ant"):   # their synthetic states keep producing resets after the first step
        assert resets_seen > 0
    dr_e, dr_g = getattr(te, "_dr_randomizer", None), getattr(tg, "_dr_randomizer", None)
    if dr_e is not None and te.randomize_observations:
        assert dr_g._observations_step == dr_e._observations_step and dr_g._actions_step == dr_e._actions_step
        assert_equal(dr_g._observations_correlated_noise, dr_e._observations_correlated_noise, "correlated observation noise")
    # an eager env.step between replays keeps every device counter in step
    ee.step(acts[1].clone()); eg.step(acts[1].clone())
    static.copy_(acts[2])
    og, _, _, _ = eg.replay_step()
    oe, _, _, _ = ee.step(acts[2].clone())
    torch.cuda.synchronize()
    assert_equal(og["obs"], oe["obs"], "obs after a mixed eager/replay sequence")


def test_capture_survives_dropped_graphs_and_restores_gc():
    """a CUDAGraph destroyed while another capture is open ends the process (hipGraphExecDestroy is rejected
    inside a capture window; SIGABRT from a destructor).  Two ways a graph can die there: (a) the cyclic collector frees an
    older graph that sat in a reference cycle -- capture_step collects and switches the collector off around the window
    (quiesce_gc); (b) a second capture_step on the SAME task rebinds task._graph -- the old graph is now released before the
    window opens.  Both are provoked here; the process must survive, replay must equal eager steps, gc must be back on."""
    import gc

    n = 256
    a = _build("Ant", n, "device")
    acts = (torch.rand(n, a.task.num_actions, generator=torch.Generator().manual_seed(1)) * 2 - 1).to("cuda:0")
    a.task.capture_step(acts.clone())
    a.task.replay_step()
    torch.cuda.synchronize()

    class Holder:  # (a): the captured graph of task A in a reference cycle, then dropped
        pass

    h = Holder()
    h.me, h.graph = h, a.task._graph
    a.task._graph = None
    del h
    gc.disable()  # garbage stays where it is until capture_step itself decides what to do with it
    try:
        b = _build("Humanoid", n, "device")
        acts_b = (torch.rand(n, b.task.num_actions, generator=torch.Generator().manual_seed(2)) * 2 - 1).to("cuda:0")
        b.task.capture_step(acts_b.clone())
        assert not gc.isenabled()  # quiesce_gc left the collector the way it found it
    finally:
        gc.enable()
    b.task.replay_step()
    old = b.task._graph
    b.task.capture_step(acts_b.clone())  # (b): re-capture on the same task
    assert b.task._graph is not old and gc.isenabled()
    del old
    # replay == eager after all of that
    e = _build("Humanoid", n, True)
    static = b.task._graph_actions
    e.task.pre_physics_step(acts_b); e.task.post_physics_step()      # warm-up step of the 1st capture
    e.task.pre_physics_step(acts_b); e.task.post_physics_step()      # the replay above
    e.task.pre_physics_step(acts_b); e.task.post_physics_step()      # warm-up step of the 2nd capture
    static.copy_(acts_b)
    b.task.replay_step()
    e.task.pre_physics_step(acts_b); e.task.post_physics_step()
    torch.cuda.synchronize()
    A_, B_ = _tensors(e.task), _tensors(b.task)
    for key in A_:
        assert_equal(B_[key], A_[key], key)

    # an exception inside the window leaves neither the capturing flag nor the collector in the wrong state
    c = _build("Ant", n, "device")

    def boom(_actions):
        raise RuntimeError("boom")

    c.task.pre_physics_step = boom
    with pytest.raises(RuntimeError, match="boom"):
        c.task.capture_step(acts.clone())
    assert c.task._capturing is False and gc.isenabled()
