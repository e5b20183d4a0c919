"""Container only (needs /root/reference): writes tests/golden/task_<Task>_n<N>_s<seed>.npz for the SURVEY.md 8f rank-3 / rank-4
tasks by running the REFERENCE's own method source (``progress_buf += 1; get_observations(); calculate_metrics(); is_done()``)
on the stubs of tests/test_oracle_pinned_*.py.  Test infrastructure, never imported by the product.
Usage: python -m oracle.gen_golden_tasks"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))

import golden_tasks as gt  # noqa: E402


def _reference(name: str, n: int, seed: int):
    """-> (st, b before the step, reference object after the step)."""
    if name in ("Quadcopter", "Ingenuity", "Crazyflie"):
        import importlib

        m = importlib.import_module(f"test_oracle_pinned_{name.lower()}")
        spec, st, b = m.inputs(n, seed)
        return st, b, m.stub(spec, st, b)
    if name == "BallBalance":
        import test_oracle_pinned_ball_balance as m

        spec, st, b, init = m.inputs(n, seed)
        return st, b, m.stub(spec, st, b, init)
    if name == "FrankaCabinet":
        import test_oracle_pinned_franka_cabinet as m

        spec, st, b = m.inputs(n, seed)
        return st, b, m.stub(spec, st, b)
    if name in ("TomoDJCabinet", "TomoSPCCabinet"):
        import test_oracle_pinned_franka_cabinet as m

        robot = "tomodj" if name == "TomoDJCabinet" else "tomospc"
        spec, st, b, lo, hi = m.tomo_inputs(robot, n, seed)
        ref, _ = m.tomo_stub(robot, spec, st, b, lo, hi)
        for k in gt.GRASP:  # the reference's attribute names carry the robot's name
            setattr(ref, "_alias_" + k, k.replace("franka_", f"{robot}_"))
        return st, b, ref
    import test_oracle_pinned_newhand_cabinet as m

    variant = {"TomoDJnewhandCabinet": "tomodj_newhand", "TomoDJnewhandTendonCabinet": "tomodj_tendon", "TomoSPCnewhandCabinet": "tomospc_newhand"}[name]
    spec, st, b, lo, hi = m.inputs(variant, n, seed)
    ref, _ = m.stub(variant, spec, st, b, lo, hi)
    return st, b, ref


def main() -> None:
    os.makedirs(gt.GOLDEN, exist_ok=True)
    for name, fam in gt.FAMILIES.items():
        for n, seed in ((64, 0), (257, 1)):
            st, b, ref = _reference(name, n, seed)
            rec = {}
            rec.update(gt.flatten("st", st))
            rec.update(gt.flatten("b", b))
            ref.progress_buf[:] += 1  # rl_task.py:289
            ref.get_observations()
            ref.calculate_metrics()
            ref.is_done()
            out = {}
            for k in gt.COMMON_OUT + fam.extra_out:
                out[k] = getattr(ref, getattr(ref, "_alias_" + k, k))
            rec.update(gt.flatten("out", out))
            path = os.path.join(gt.GOLDEN, f"task_{name}_n{n}_s{seed}.npz")
            np.savez_compressed(path, **rec)
            print(f"wrote {path} ({len(rec)} arrays)")


if __name__ == "__main__":
    main()
