"""CPU: `python bench.py --gpus 2` is its own launcher -- the parent starts two ranks under torch.distributed.run before
any GPU call, the ranks rendezvous over gloo, all-reduce the episode statistics, take the max of their timings and rank 0
prints ONE JSON line with n_gpus = 2.  `--task _dry` replaces the HIP step by CPU tensor work, everything else is the code
path a real `--gpus N` run takes (README.md:292-314 and scripts/rlgames_train.py:114-120,171 of the reference)."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _run(extra, env=None):
    e = os.environ.copy()
    e.update(env or {})
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--task", "_dry", "--steps", "4", "--warmup", "1"] + extra,
                       capture_output=True, text=True, timeout=600, env=e)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout  # ONE JSON line, from rank 0 only
    return json.loads(lines[0]), p.stderr


def test_gpus_2_launches_two_ranks():
    out, err = _run(["--gpus", "2", "--envs", "1000", "--stats-every", "1"])
    assert "torch.distributed.run" in err  # the parent was the launcher
    assert out["n_gpus"] == 2 and out["ranks_seen"] == 2 and out["collective_backend"] == "gloo"
    st = out["config"]["episode_stats"]
    assert st["num_envs"] == 2000 and st["env_steps"] == 2000 * 4 and st["steps"] == 4  # both ranks' envs reached the all-reduce, every step
    # the collective sits INSIDE the timed step: one per step, and its host cost is on the line
    assert out["collective"]["calls_in_timed_region"] == 4 and out["collective"]["every"] == 1 and out["collective"]["bytes"] == 40
    assert out["collective_ms_per_step"] > 0
    assert out["rank_ms_per_step"]["min"] <= out["rank_ms_per_step"]["max"] == out["ms_per_step"]
    assert abs(out["value"] - 2 * 1000 * 4 / (out["ms_per_step"] * 4e-3)) < 1e-6 * out["value"]
    assert out["scaling"] == "weak" and out["data"] == "dry-run"


def test_strong_scaling_splits_the_env_count():
    out, _ = _run(["--gpus", "2", "--envs", "1024", "--scaling", "strong"])
    assert out["n_gpus"] == 2 and out["config"]["envs_per_gpu"] == 512 and out["config"]["episode_stats"]["num_envs"] == 1024


def test_single_rank_needs_no_launcher():
    out, err = _run(["--envs", "256"])
    assert "torch.distributed.run" not in err
    assert out["n_gpus"] == 1 and out["ranks_seen"] == 1 and out["collective_backend"] == "none"
    assert "collective" not in out  # nothing to exchange at N = 1: no fold, no collective


def test_default_cadence_is_one_collective_per_epoch():
    out, _ = _run(["--gpus", "2", "--envs", "512"])  # 4 timed steps < the 16-step epoch: one collective, on the last step
    assert out["collective"]["every"] == 4 and out["collective"]["calls_in_timed_region"] == 1
    assert out["config"]["episode_stats"]["env_steps"] == 2 * 512 * 4


def test_stats_every_k_steps():
    out, _ = _run(["--gpus", "2", "--envs", "512", "--stats-every", "2"])
    assert out["collective"]["calls_in_timed_region"] == 2 and out["collective"]["every"] == 2
    assert out["config"]["episode_stats"]["env_steps"] == 2 * 512 * 4


def test_bench_does_not_import_the_oracle_or_tests_at_module_scope():
    src = open(os.path.join(ROOT, "bench.py")).read()
    head = src.split("def cpu_baseline(")[0]
    for ln in head.splitlines():
        if ln.startswith(("import ", "from ")):
            assert "oracle" not in ln and "helpers" not in ln, ln
    assert '"tests"' not in src
    p = subprocess.run([sys.executable, "-c", "import sys; sys.argv=['bench.py']; import bench; "
                        "assert not [m for m in sys.modules if m == 'oracle' or m.startswith('oracle.')], 'oracle imported'"],
                       cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stderr[-1500:]
