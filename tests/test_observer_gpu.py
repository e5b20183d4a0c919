"""-m gpu parity tests: the device-side RLGPUAlgoObserver (through the C ABI) vs the CPU oracle (pinned to
utils/rlgames/rlgames_utils.py by tests/test_oracle_pinned_observer.py).  Bar: same tags in the same order, same steps,
values within 1e-5 rel / 1e-6 abs (f64 running sums against torch.mean of the concatenation)."""
import types

import pytest
import torch

from observer_cases import Writer, infos_stream
from omniisaacgymenvs_amd.utils.rlgames.rlgames_utils import AverageMeter, RLGPUAlgoObserver
from oracle import observer_oracle as oo

pytestmark = pytest.mark.gpu


def _observer(writer):
    obs = RLGPUAlgoObserver()
    obs.after_init(types.SimpleNamespace(device="cuda:0", ppo_device="cuda:0", games_to_track=100, writer=writer))
    return obs


def _same(a, b):
    assert [(t, s) for t, _, s in a] == [(t, s) for t, _, s in b]
    for (t, x, _), (_, y, _) in zip(a, b):
        assert abs(x - y) <= 1e-5 * abs(y) + 1e-6 or (x != x and y != y), (t, x, y)


@pytest.mark.parametrize("steps,seed", [(1, 0), (40, 1), (1000, 2)])
@pytest.mark.parametrize("host_number", [False, True])
def test_observer_matches_oracle(steps, seed, host_number):
    wd, wo = Writer(), Writer()
    dev, orc = _observer(wd), oo.ObserverOracle(wo)
    for epoch in range(3):
        for a, b in zip(infos_stream(steps, seed + epoch, "cuda:0", host_number), infos_stream(steps, seed + epoch, "cpu", host_number)):
            dev.process_infos(a, None)
            orc.process_infos(b, None)
        assert bool(dev.ep_infos) == bool(orc.ep_infos)
        dev.after_print_stats(100 * epoch, epoch, 1.5 * epoch)
        orc.after_print_stats(100 * epoch, epoch, 1.5 * epoch)
        assert not dev.ep_infos
    _same(wd.calls, wo.calls)
    assert sum(t.startswith("Episode/") for t, _, _ in wd.calls) == 3 * (5 + int(host_number))


def test_epoch_without_episode_infos_logs_only_direct_scalars_and_scores():
    w = Writer()
    obs = _observer(w)
    obs.process_infos({"consecutive_successes": torch.tensor(3.0, device="cuda:0")}, None)
    obs.mean_scores.update(torch.tensor([[1.0], [3.0]], device="cuda:0"))
    obs.after_print_stats(10, 1, 2.0)
    assert [t for t, _, _ in w.calls] == ["consecutive_successes/frame", "consecutive_successes/iter", "consecutive_successes/time",
                                          "scores/mean", "scores/iter", "scores/time"]
    assert w.calls[3][1] == 2.0
    obs.after_clear_stats()
    assert obs.mean_scores.current_size == 0


def test_with_a_task_extras_buffer():
    """The Crazyflie mirror binds extras['episode'] to views of one (8,) device buffer: the observer reads them by pointer."""
    from helpers import make_task
    from omniisaacgymenvs_amd.sim import workloads

    n = 4096
    env, task = make_task("Crazyflie", n)
    st = workloads.crazyflie_state(n, task._env_pos.cpu(), device="cpu")
    workloads.load_crazyflie_state(task, st)
    w = Writer()
    obs = _observer(w)
    want = {k: [] for k in task.episode_sums}
    for step in range(5):
        task.post_physics_step()
        task.pre_physics_step(st["actions"].to(task.device))
        obs.process_infos(task.extras, None)
        for k in want:
            want[k].append(float(task.extras["episode"][k]))
    obs.after_print_stats(0, 0, 0.0)
    got = {t[len("Episode/"):]: v for t, v, _ in w.calls}
    for k, vals in want.items():
        m = sum(vals) / len(vals)
        assert abs(got[k] - m) <= 1e-5 * abs(m) + 1e-6, k


def test_average_meter_window():
    m = AverageMeter(1, 4)
    m.update(torch.tensor([[1.0], [2.0]]))
    m.update(torch.tensor([[4.0], [6.0], [8.0]]))
    assert m.current_size == 4 and abs(float(m.get_mean()) - (1.5 * 1 + 6.0 * 3) / 4) < 1e-6


def test_epoch_cost_against_the_reference_pattern(tmp_path):
    """Measurement, not parity: one epoch of 1000 steps x 10 keys.  The reference pattern (the oracle run with device tensors:
    a Python list, then 10 x 1000 torch.cat launches and allocations) against one accumulate launch per step + one means
    launch.  The numbers go to gpurun_out/observer_timing.json when that directory exists."""
    import json
    import os
    import time

    steps, keys = 1000, 10
    buf = torch.randn(steps, keys, device="cuda:0")
    names = [f"rew_{i}" for i in range(keys)]
    stream = [{"episode": {k: buf[t, i] for i, k in enumerate(names)}} for t in range(steps)]

    def run(obs):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for infos in stream:
            obs.process_infos(infos, None)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        obs.after_print_stats(0, 0, 0.0)
        torch.cuda.synchronize()
        return (t1 - t0) * 1e3, (time.perf_counter() - t1) * 1e3

    wd, wo = Writer(), Writer()
    dev, orc = _observer(wd), oo.ObserverOracle(wo, device="cuda:0")
    run(dev), run(orc)  # warm-up (code objects, allocator)
    wd.calls.clear(), wo.calls.clear()
    d_steps, d_print = run(dev)
    o_steps, o_print = run(orc)
    _same(wd.calls, wo.calls)
    rec = {"steps": steps, "keys": keys, "device_observer_ms": {"process_infos_total": d_steps, "after_print_stats": d_print},
           "reference_pattern_ms": {"process_infos_total": o_steps, "after_print_stats": o_print}}
    print(json.dumps(rec))
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    if os.path.isdir(out):
        with open(os.path.join(out, "observer_timing.json"), "w") as f:
            json.dump(rec, f, indent=1)
    assert d_print < 10.0 * o_print   # a measurement, not a gate (shared boxes, timings jitter)
