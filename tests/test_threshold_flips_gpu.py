"""-m gpu: census of the threshold flips SURVEY.md 7 ("hard parts") and appendix A.3 warn about.

Three compares feed bit-exact outputs from computed fp32 values: `abs(rot_dist) <= success_tolerance` and
`goal_dist >= fall_dist` (compute_hand_reward, in_hand_manipulation.py:497-507) and `up_proj > 0.93` (locomotion.py:319,
reward only).  HIP's asinf / sinf / cosf and torch-CPU's differ by an ulp here and there, so an env that sits within a few
ulps of a threshold can land on the other side.  These tests walk the inputs across each threshold in steps of about one
ulp of the compared quantity, count the rows on which the GPU and the oracle decide differently, assert that every such row
lies within 1e-6 of the threshold (the mask the parity tests use) and that their number stays small, and leave the counts in
gpurun_out/threshold_flips.json (quoted in DESIGN.md section 4)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from helpers import make_task, oracle_buffers, oracle_spec
from omniisaacgymenvs_amd.sim import workloads
from oracle import hand_oracle as ho
from oracle import locomotion_oracle as lo
from oracle import torch_utils as tu
from test_hand_gpu import _loaded as hand_loaded, _oracle_buffers as hand_buffers, _oracle_state as hand_state, _spec as hand_spec

pytestmark = pytest.mark.gpu
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out", "threshold_flips.json")
HALF = 2048  # rows on either side of the threshold


def _record(key, value):
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    data = json.load(open(OUT)) if os.path.exists(OUT) else {}
    data[key] = value
    json.dump(data, open(OUT, "w"), indent=1, sort_keys=True)


def _walk(center: float, n: int) -> torch.Tensor:
    """n fp32 values, consecutive representable numbers centred on fp32(center)."""
    c = np.float32(center).view(np.int32)
    return torch.from_numpy((c + np.arange(-n // 2, n - n // 2, dtype=np.int32)).view(np.float32).copy())


@pytest.mark.parametrize("cfg_name", ["ShadowHand", "ShadowHandOpenAI_FF"])
def test_rot_dist_against_success_tolerance(cfg_name):
    n = 2 * HALF
    env, task, st = hand_loaded(cfg_name, n)
    tol = task.success_tolerance
    theta = _walk(tol, n)  # rotation about x by theta against an identity goal: rot_dist = 2 asin(sin(theta / 2)) ~ theta
    st["object_rot"] = torch.stack([torch.cos(theta / 2), torch.sin(theta / 2), torch.zeros(n), torch.zeros(n)], dim=-1).contiguous()
    st["goal_rot"] = torch.tensor([1.0, 0.0, 0.0, 0.0]).repeat(n, 1)
    st["object_pos"] = (task.goal_pos.cpu() + task._env_pos.cpu()).contiguous()  # goal_dist = 0: only the rotation decides
    st["progress"][:] = 5
    st["successes"][:] = 0
    st["reset"][:] = 0
    st["reset_goal"][:] = 0
    workloads.load_hand_state(task, st)
    spec, s, b = hand_spec(task), hand_state(task, st), hand_buffers(task)
    qd = tu.quat_mul(s.object_rot, tu.quat_conjugate(b.goal_rot))
    rd = 2.0 * torch.asin(torch.clamp(torch.norm(qd[:, 1:4], p=2, dim=-1), max=1.0))
    ho.post_physics(spec, s, b)
    task.post_physics_step()
    torch.cuda.synchronize()
    got = task.reset_goal_buf.cpu()
    flips = got != b.reset_goal_buf
    assert 0 < int(b.reset_goal_buf.sum()) < n  # the walk does straddle the threshold
    dist = (rd.abs() - tol).abs()
    assert bool((dist[flips] <= 1e-6).all()), "a flip outside the +-1e-6 mask of the parity tests"
    ulp = float(np.spacing(np.float32(tol)))
    _record(f"{cfg_name}.rot_dist<=success_tolerance({tol})", {
        "rows": n, "rows_within_1e-6": int((dist <= 1e-6).sum()), "flips": int(flips.sum()),
        "farthest_flip_ulps": float((dist[flips].max() / ulp) if flips.any() else 0.0), "ulp": ulp})
    assert int(flips.sum()) <= 8, int(flips.sum())
    # the reward follows the flag: off the flipped rows it is within tolerance
    keep = ~flips
    torch.testing.assert_close(task.rew_buf.cpu()[keep], b.rew_buf[keep], rtol=1e-5, atol=1e-6)


def test_goal_dist_against_fall_dist():
    n = 2 * HALF
    env, task, st = hand_loaded("ShadowHandOpenAI_FF", n)
    fall = task.fall_dist
    d = _walk(fall, n)  # displacement along x only: the norm is exact, so the compare sees the walked value itself
    off = torch.zeros(n, 3)
    off[:, 0] = d
    st["object_pos"] = (task.goal_pos.cpu() + off + task._env_pos.cpu()).contiguous()
    st["progress"][:] = 5
    st["successes"][:] = 0
    st["reset"][:] = 0
    workloads.load_hand_state(task, st)
    spec, s, b = hand_spec(task), hand_state(task, st), hand_buffers(task)
    ho.post_physics(spec, s, b)
    task.post_physics_step()
    torch.cuda.synchronize()
    flips = task.reset_buf.cpu() != b.reset_buf
    assert 0 < int(b.reset_buf.sum()) < n
    _record(f"ShadowHandOpenAI_FF.goal_dist>=fall_dist({fall})", {"rows": n, "flips": int(flips.sum())})
    assert int(flips.sum()) == 0  # subtraction, squares of one non-zero component and sqrt are correctly rounded on both sides


def test_up_proj_against_093():
    n = 2 * HALF
    env, task = make_task("Ant", n)
    st = workloads.locomotion_state(n, task.dof_limits_lower.cpu(), task.dof_limits_upper.cpu(), 4, device="cpu")
    tilt = _walk(math.acos(0.93), n)  # rotation about x by `tilt`: up_vec.z = cos(tilt) walks across 0.93
    st["rot"] = torch.stack([torch.cos(tilt / 2), torch.sin(tilt / 2), torch.zeros(n), torch.zeros(n)], dim=-1).contiguous()
    st["pos"][:, 2] = 0.6  # alive: the reward is not overridden by the death cost
    workloads.load_locomotion_state(task, st)
    spec, b = oracle_spec(task), oracle_buffers(task)
    lo.post_physics(spec, b, st["pos"], st["rot"], st["vel"], st["dof_pos"], st["dof_vel"], st["sensors"])
    obs, rew, reset, _ = task.post_physics_step()
    torch.cuda.synchronize()
    up_o, up_g = b.obs_buf[:, 10], obs[:, 10].cpu()
    flips = (up_o > 0.93) != (up_g > 0.93)
    assert 0 < int((up_o > 0.93).sum()) < n
    dist = (up_o - 0.93).abs()
    assert bool((dist[flips] <= 1e-6).all())
    _record("Ant.up_proj>0.93", {"rows": n, "rows_within_1e-6": int((dist <= 1e-6).sum()), "flips": int(flips.sum()),
                                 "max_abs_up_proj_error": float((up_o - up_g).abs().max())})
    assert torch.equal(reset.cpu(), b.reset_buf)  # done flags do not depend on it
    keep = ~flips
    torch.testing.assert_close(rew.cpu()[keep], b.rew_buf[keep], rtol=1e-5, atol=1e-6)
    # a flipped row differs by exactly the up reward (locomotion.py:319)
    if flips.any():
        torch.testing.assert_close((rew.cpu()[flips] - b.rew_buf[flips]).abs(), torch.full((int(flips.sum()),), task.up_weight), rtol=1e-4, atol=1e-5)
