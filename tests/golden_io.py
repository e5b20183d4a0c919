"""Load tests/golden/*.npz (written by oracle/gen_golden.py from the reference's own function source)."""
from __future__ import annotations

import glob
import os

import numpy as np
import torch

from oracle import locomotion_oracle as lo

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def files(prefix: str):
    return sorted(glob.glob(os.path.join(GOLDEN, f"{prefix}_n*.npz")))


def load(path: str):
    with np.load(path, allow_pickle=False) as z:
        return {k: torch.from_numpy(z[k]) for k in z.files}


def spec_for(kind: str, g) -> lo.LocomotionSpec:
    spec = lo.ant_spec() if kind == "ant" else lo.humanoid_spec()
    spec.dof_limits_lower, spec.dof_limits_upper = g["lo"], g["hi"]
    spec.joint_gears, spec.motor_effort_ratio = g["gears"], g["motor_effort_ratio"]
    return spec


def unflat(g, prefix: str, cls=None):
    """Inverse of oracle/gen_golden._flat: {prefix_field: tensor} -> cls(**fields) (missing fields None) or a dict."""
    import dataclasses

    p = prefix + "_"
    d = {}
    for k, v in g.items():
        if not k.startswith(p):
            continue
        name = k[len(p):]
        if "__" in name:
            a, b = name.split("__", 1)
            d.setdefault(a, {})[b] = v.clone()
        else:
            d[name] = v.clone()
    if cls is None:
        return d
    for f in dataclasses.fields(cls):
        if f.default is dataclasses.MISSING and f.default_factory is dataclasses.MISSING:
            d.setdefault(f.name, None)
    return cls(**d)
