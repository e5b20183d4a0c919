"""Golden vectors of the SURVEY.md 8f rank-3 / rank-4 tasks: one registry shared by the generator (oracle/gen_golden_tasks.py,
container only: it runs the reference's own method source through the pinned-test stubs), the CPU test (oracle == golden) and
the GPU test (HIP == golden).  A file holds the inputs (``st.*``: what the views return; ``b.*``: the task buffers before the
step) and what the reference's ``progress += 1; get_observations(); calculate_metrics(); is_done()`` left (``out.*``)."""
from __future__ import annotations

import dataclasses
import glob
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
COMMON_OUT = ("obs_buf", "rew_buf", "reset_buf", "progress_buf")


# ------------------------------------------------------------------ (de)serialisation
def flatten(prefix: str, obj) -> dict:
    items = dataclasses.asdict(obj).items() if dataclasses.is_dataclass(obj) else obj.items()
    out = {}
    for k, v in items:
        if isinstance(v, torch.Tensor):
            out[f"{prefix}.{k}"] = v.detach().cpu().numpy()
        elif isinstance(v, (list, tuple)) and v and isinstance(v[0], torch.Tensor):
            out.update({f"{prefix}.{k}.{i}": t.detach().cpu().numpy() for i, t in enumerate(v)})
        elif isinstance(v, dict):
            out.update({f"{prefix}.{k}.{kk}": t.detach().cpu().numpy() for kk, t in v.items()})
    return out


def unflatten(g: dict, prefix: str) -> dict:
    out = {}
    for key, v in g.items():
        if not key.startswith(prefix + "."):
            continue
        parts = key[len(prefix) + 1:].split(".")
        if len(parts) == 1:
            out[parts[0]] = v
        elif parts[1].isdigit():
            out.setdefault(parts[0], {})[int(parts[1])] = v
        else:
            out.setdefault(parts[0], {})[parts[1]] = v
    for k, v in out.items():
        if isinstance(v, dict) and all(isinstance(i, int) for i in v):
            out[k] = [v[i] for i in range(len(v))]
    return out


def files(name: str):
    return sorted(glob.glob(os.path.join(GOLDEN, f"task_{name}_n*.npz")))


def load(path: str) -> dict:
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


# ------------------------------------------------------------------ the task families
def _buffers(cls, fields: dict):
    names = {f.name for f in dataclasses.fields(cls)}
    return cls(**{k: v for k, v in fields.items() if k in names})


class Family:
    """name: task name of the mirror; pinned: (module, inputs args -> (spec, st, b, ...), stub args); oracle: post_physics call."""

    def __init__(self, name, oracle_mod, buffers_cls, spec_fn, post_args, extra_out=(), limits=None):
        self.name, self.oracle_mod, self.buffers_cls, self.spec_fn = name, oracle_mod, buffers_cls, spec_fn
        self.post_args, self.extra_out, self.limits = post_args, tuple(extra_out), limits

    def spec(self):
        return self.spec_fn()

    def oracle_post(self, st: dict, bfields: dict):
        import importlib

        om = importlib.import_module(f"oracle.{self.oracle_mod}")
        b = _buffers(getattr(om, self.buffers_cls), {k: (v.clone() if isinstance(v, torch.Tensor) else {kk: t.clone() for kk, t in v.items()})
                                                     for k, v in bfields.items()})
        spec = self.spec()
        if self.post_args == "st":
            lo, hi = self.limits(spec)
            om.post_physics(spec, b, st, lo, hi)
        else:
            om.post_physics(spec, b, *[st[k] for k in self.post_args])
        return b

    def outputs(self, obj) -> dict:
        out = {k: getattr(obj, k) for k in COMMON_OUT}
        for k in self.extra_out:
            v = getattr(obj, k)
            out[k] = v
        return out


def _cab_limits(spec):
    lo = torch.tensor([-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973, 0.0, 0.0])
    hi = torch.tensor([2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973, 0.04, 0.04])
    return lo[9 - spec.num_dofs:], hi[9 - spec.num_dofs:]


def _nh_limits(spec):
    g = torch.Generator().manual_seed(spec.num_dofs)
    return -(torch.rand(spec.num_dofs, generator=g) + 0.3), torch.rand(spec.num_dofs, generator=g) + 0.3


TENDON_ACTUATED = list(range(7)) + [7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20, 22, 23, 24]


def _lazy(mod, fn, *a):
    def f():
        import importlib

        return getattr(importlib.import_module(f"oracle.{mod}"), fn)(*a)
    return f


GRASP = ("franka_grasp_rot", "franka_grasp_pos", "drawer_grasp_rot", "drawer_grasp_pos")
FAMILIES = {
    "Quadcopter": Family("Quadcopter", "quadcopter_oracle", "QuadcopterBuffers", _lazy("quadcopter_oracle", "QuadcopterSpec"),
                         ("root_pos", "root_rot", "root_vel", "dof_pos", "env_pos")),
    "BallBalance": Family("BallBalance", "ball_balance_oracle", "BallBalanceBuffers", _lazy("ball_balance_oracle", "BallBalanceSpec"),
                          ("ball_pos", "ball_vel", "dof_pos", "dof_vel", "sensors", "env_pos")),
    "Ingenuity": Family("Ingenuity", "ingenuity_oracle", "IngenuityBuffers", _lazy("ingenuity_oracle", "IngenuitySpec"),
                        ("root_pos", "root_rot", "root_vel", "env_pos")),
    "Crazyflie": Family("Crazyflie", "crazyflie_oracle", "CrazyflieBuffers", _lazy("crazyflie_oracle", "CrazyflieSpec"),
                        ("root_pos", "root_rot", "root_vel", "env_pos"), extra_out=("episode_sums",)),
    "FrankaCabinet": Family("FrankaCabinet", "franka_cabinet_oracle", "FrankaCabinetBuffers", _lazy("franka_cabinet_oracle", "FrankaCabinetSpec"),
                            "st", extra_out=GRASP, limits=_cab_limits),
    "TomoDJCabinet": Family("TomoDJCabinet", "franka_cabinet_oracle", "FrankaCabinetBuffers", _lazy("franka_cabinet_oracle", "tomodj_spec"),
                            "st", extra_out=GRASP, limits=_cab_limits),
    "TomoSPCCabinet": Family("TomoSPCCabinet", "franka_cabinet_oracle", "FrankaCabinetBuffers", _lazy("franka_cabinet_oracle", "tomospc_spec"),
                             "st", extra_out=GRASP, limits=_cab_limits),
    "TomoDJnewhandCabinet": Family("TomoDJnewhandCabinet", "newhand_cabinet_oracle", "NewhandBuffers",
                                   _lazy("newhand_cabinet_oracle", "tomodj_newhand_spec"), "st",
                                   extra_out=("keypoints_finger", "keypoints_drawer"), limits=_nh_limits),
    "TomoDJnewhandTendonCabinet": Family("TomoDJnewhandTendonCabinet", "newhand_cabinet_oracle", "NewhandBuffers",
                                         _lazy("newhand_cabinet_oracle", "tomodj_newhand_tendon_spec", TENDON_ACTUATED), "st",
                                         extra_out=("keypoints_finger", "keypoints_drawer", "terminate_buf"), limits=_nh_limits),
    "TomoSPCnewhandCabinet": Family("TomoSPCnewhandCabinet", "newhand_cabinet_oracle", "NewhandBuffers",
                                    _lazy("newhand_cabinet_oracle", "tomospc_newhand_spec"), "st",
                                    extra_out=("keypoints_finger", "keypoints_drawer"), limits=_nh_limits),
}
