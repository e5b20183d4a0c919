"""Shared helpers for the parity tests: build a product task + the matching oracle spec."""
from __future__ import annotations

import torch

from omniisaacgymenvs_amd.utils.task_util import make_task  # noqa: F401  (re-exported for the tests)
from oracle import locomotion_oracle as lo

# north_star tolerance for fp32 observations and rewards
RTOL, ATOL = 1e-5, 1e-6


def oracle_spec(task) -> lo.LocomotionSpec:
    """Oracle spec carrying exactly the task's scalars and tables (moved to CPU)."""
    base = lo.ant_spec() if task._limit_cost_kind == 0 else lo.humanoid_spec()
    base.dof_limits_lower = task.dof_limits_lower.cpu()
    base.dof_limits_upper = task.dof_limits_upper.cpu()
    base.joint_gears = task.joint_gears.cpu()
    base.motor_effort_ratio = task.motor_effort_ratio.cpu()
    return base


def oracle_buffers(task) -> lo.LocomotionBuffers:
    return lo.LocomotionBuffers(
        obs_buf=task.obs_buf.cpu().clone(), rew_buf=task.rew_buf.cpu().clone(), reset_buf=task.reset_buf.cpu().clone(),
        progress_buf=task.progress_buf.cpu().clone(), potentials=task.potentials.cpu().clone(),
        prev_potentials=task.prev_potentials.cpu().clone(), up_vec=task.up_vec.cpu().clone(),
        heading_vec=task.heading_vec.cpu().clone(), actions=task.actions.cpu().clone(),
    )


def assert_close(actual: torch.Tensor, expected: torch.Tensor, what: str, rtol=RTOL, atol=ATOL, mask=None):
    a, e = actual.detach().cpu(), expected.detach().cpu()
    assert a.shape == e.shape and a.dtype == e.dtype, f"{what}: {a.shape}/{a.dtype} vs {e.shape}/{e.dtype}"
    err = (a - e).abs()
    tol = atol + rtol * e.abs()
    bad = err > tol
    if mask is not None:
        bad = bad & mask
    if bad.any():
        idx = bad.nonzero()[0].tolist()
        raise AssertionError(
            f"{what}: {int(bad.sum())} of {bad.numel()} elements beyond rtol={rtol} atol={atol}; first at {idx}: "
            f"got {a[tuple(idx)].item()!r} want {e[tuple(idx)].item()!r}"
        )


def assert_equal(actual: torch.Tensor, expected: torch.Tensor, what: str):
    a, e = actual.detach().cpu(), expected.detach().cpu()
    assert a.dtype == e.dtype, f"{what}: dtype {a.dtype} vs {e.dtype}"
    assert torch.equal(a, e), f"{what}: {(a != e).sum().item()} mismatches (bit-exact required)"
