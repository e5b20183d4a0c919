"""Build liboige_hip.so in-tree with hipcc for gfx950 (cross-compiles without a GPU).

    python -m omniisaacgymenvs_amd.build [--force]

-ffp-contract=off: the kernels spell out every fma the reference's torch ops perform
(oige_math.hpp) so fp32 rounding follows the reference; the compiler must not add more.
"""
from __future__ import annotations

import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "liboige_hip.so")
SOURCES = ["oige_core.hip", "locomotion.hip", "cartpole.hip", "hand.hip", "anymal_terrain.hip", "dr_noise.hip", "anymal.hip", "quadcopter.hip", "ball_balance.hip", "ingenuity.hip", "crazyflie.hip", "franka_cabinet.hip", "newhand_cabinet.hip", "observer.hip"]


def dependencies():
    """Every file the library is built from: all of csrc/ (*.hip, *.hpp) and include/*.h -- globbed, so a new header
    cannot be forgotten in the stale check."""
    import glob

    inc = os.path.join(HERE, "..", "include")
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")) + glob.glob(os.path.join(CSRC, "*.hpp"))
                  + glob.glob(os.path.join(inc, "*.h")))


FLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-ffp-contract=off",
    "-fPIC",
    "-shared",
    "-Wall",
    "-Wno-unused-function",
]


def _hipcc() -> str:
    for cand in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if cand and (os.path.isabs(cand) and os.path.exists(cand) or not os.path.isabs(cand)):
            return cand
    return "hipcc"


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = dependencies() + [os.path.abspath(__file__)]
    return any(os.path.getmtime(d) > t for d in deps)


def build_variant(name: str, defines, verbose: bool = True) -> str:
    """Experiment builds: omniisaacgymenvs_amd/variants/liboige_<name>.so with extra -D flags (select with OIGE_LIB)."""
    out_dir = os.path.join(HERE, "variants")
    os.makedirs(out_dir, exist_ok=True)
    out = os.path.join(out_dir, f"liboige_{name}.so")
    cmd = [_hipcc()] + FLAGS + [f"-D{d}" for d in defines] + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", out]
    if verbose:
        print("[oige build]", name, defines, flush=True)
    subprocess.check_call(cmd)
    return out


def build(force: bool = False, verbose: bool = True) -> str:
    if not force and not is_stale():
        return LIB
    cmd = [_hipcc()] + FLAGS + [os.path.join(CSRC, s) for s in SOURCES] + ["-o", LIB]
    if verbose:
        print("[oige build]", " ".join(cmd), flush=True)
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv)
    print(LIB)
