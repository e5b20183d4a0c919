"""Known-answer tests for the UNPINNED helper layer (oracle/torch_utils.py restates
omni.isaac.core.utils.torch, which is not in the reference tree): algebraic invariants and the
facts the reference's own call sites fix."""
import math

import pytest

import torch

from oracle import torch_utils as tu


def _rand_quat(n, seed=0):
    g = torch.Generator().manual_seed(seed)
    q = torch.randn(n, 4, generator=g)
    return q / q.norm(dim=-1, keepdim=True)


def test_quat_mul_is_hamilton_wxyz():
    # factory_control.py:216-221: quat_mul(q, conj(q))[:, 0] is the squared norm, rest zero
    g = torch.Generator().manual_seed(1)
    q = torch.randn(100, 4, generator=g)
    p = tu.quat_mul(q, tu.quat_conjugate(q))
    assert torch.allclose(p[:, 0], (q * q).sum(-1), rtol=1e-5, atol=1e-6)
    assert torch.allclose(p[:, 1:], torch.zeros(100, 3), atol=1e-5)
    i, j, k = torch.tensor([[0., 1, 0, 0]]), torch.tensor([[0., 0, 1, 0]]), torch.tensor([[0., 0, 0, 1]])
    assert torch.allclose(tu.quat_mul(i, j), k) and torch.allclose(tu.quat_mul(j, k), i)
    a, b = _rand_quat(64, 2), _rand_quat(64, 3)
    assert torch.allclose(tu.quat_mul(a, b).norm(dim=-1), torch.ones(64), atol=1e-5)


def test_rotate_conventions():
    r2 = math.sqrt(0.5)
    qz90 = torch.tensor([[r2, 0.0, 0.0, r2]])
    x = torch.tensor([[1.0, 0.0, 0.0]])
    for f in (tu.quat_rotate, tu.quat_apply):
        assert torch.allclose(f(qz90, x), torch.tensor([[0.0, 1.0, 0.0]]), atol=1e-6)  # 90 deg about z: x -> y
    q, v = _rand_quat(200, 4), torch.randn(200, 3, generator=torch.Generator().manual_seed(5))
    assert torch.allclose(tu.quat_rotate_inverse(q, tu.quat_rotate(q, v)), v, atol=1e-5)
    assert torch.allclose(tu.quat_apply(q, v), tu.quat_rotate(q, v), atol=1e-5)
    assert torch.allclose(tu.quat_rotate(q, v).norm(dim=-1), v.norm(dim=-1), rtol=1e-5)


def test_scale_unscale_clamp():
    lo_, hi_ = torch.tensor([-0.7, 0.52]), torch.tensor([0.7, 1.75])
    x = torch.tensor([[0.1, 1.0], [-0.7, 1.75]])
    assert torch.allclose(tu.scale(tu.unscale(x, lo_, hi_), lo_, hi_), x, atol=1e-6)
    assert torch.allclose(tu.unscale(torch.stack([lo_, hi_]), lo_, hi_), torch.tensor([[-1.0, -1.0], [1.0, 1.0]]), atol=2e-7)
    assert torch.equal(tu.tensor_clamp(torch.tensor([[-9.0, 9.0]]), lo_, hi_), torch.tensor([[-0.7, 1.75]]))


def test_euler_and_axis_angle():
    ang = torch.tensor([0.3, -1.2, 2.5])
    z = torch.tensor([[0.0, 0.0, 1.0]]).repeat(3, 1)
    q = tu.quat_from_angle_axis(ang, z)
    roll, pitch, yaw = tu.get_euler_xyz(q)
    assert torch.allclose(yaw, ang % (2 * math.pi), atol=1e-5)  # % 2pi, warp/shared/locomotion.py:275
    assert torch.allclose(roll, torch.zeros(3), atol=1e-6) and torch.allclose(pitch, torch.zeros(3), atol=1e-6)
    r2 = math.sqrt(0.5)
    _, pitch, _ = tu.get_euler_xyz(torch.tensor([[r2, 0.0, r2, 0.0], [r2, 0.0, -r2, 0.0]]))
    assert torch.allclose(pitch, torch.tensor([math.pi / 2, (-math.pi / 2) % (2 * math.pi)]), atol=2e-3)
    assert torch.allclose(q.norm(dim=-1), torch.ones(3), atol=1e-6)


def test_normalize_eps():
    assert torch.equal(tu.normalize(torch.zeros(2, 3)), torch.zeros(2, 3))  # x / max(|x|, 1e-9)
    v = torch.tensor([[3.0, 4.0, 0.0]])
    assert torch.allclose(tu.normalize(v), torch.tensor([[0.6, 0.8, 0.0]]))


def test_fma_division_by_constant_equals_ieee_division(tmp_path):
    """csrc/anymal_terrain.hip div_by_const: q0 = x * RN(1/h), two fma residual corrections == RN(x / h) bit for bit."""
    import os
    import shutil
    import subprocess

    if shutil.which("gcc") is None:
        pytest.skip("gcc not available")
    src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "div_check.c")
    exe = str(tmp_path / "div_check")
    subprocess.run(["gcc", "-O2", "-ffp-contract=off", "-o", exe, src, "-lm"], check=True)
    out = subprocess.run([exe, "2000000"], capture_output=True, text=True)
    assert out.returncode == 0, out.stdout
    assert "mismatches 0" in out.stdout


def test_normalize_angle_as_a_reduction_matches_atan2_of_sin_cos():
    """csrc/oige_math.hpp evaluates the jit normalize_angle (locomotion.py:220-221: atan2(sin x, cos x)) as the reduction of x
    to (-pi, pi] in double -- five instructions instead of three library calls.  This replays that arithmetic with numpy
    against torch's own atan2(sin, cos): within 1 ulp everywhere the locomotion kernels can reach ([-3 pi - 1, 2 pi + 1]:
    yaw, roll in [0, 2 pi), angle_to_target = atan2(..) - yaw), and on the same side of the +-pi seam for every float within 40
    ulps of a multiple of pi.  (The GPU side of the same statement: tests/test_locomotion_gpu.py compares obs columns 7:10.)"""
    import math

    import numpy as np

    g = torch.Generator().manual_seed(0)
    x = (torch.rand(2_000_000, generator=g) * (5 * math.pi + 2) - (3 * math.pi + 1)).float()
    seams = []
    for m in range(-4, 5):
        v = np.float32(m * math.pi)
        for _ in range(40):
            v = np.nextafter(v, np.float32(-100))
        for _ in range(80):
            seams.append(float(v))
            v = np.nextafter(v, np.float32(100))
    x = torch.cat([x, torch.tensor(seams, dtype=torch.float32)])
    ref = torch.atan2(torch.sin(x), torch.cos(x))
    xd = x.double().numpy()
    k = np.rint(xd * 0.15915494309189535)
    mine = torch.from_numpy((xd - k * 6.283185307179586).astype(np.float32))  # (numpy has no fma: the product is exact enough in double)
    d = (ref - mine).abs()
    assert float(d.max()) <= 2.4e-7, float(d.max())
    assert int(((ref > 0) != (mine > 0)).sum()) == 0  # the seam: same sign on every float around an odd multiple of pi
    assert bool((d <= 1e-6 + 1e-5 * ref.abs()).all())
