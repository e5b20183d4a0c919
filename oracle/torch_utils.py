"""ORACLE (test infrastructure, never shipped on the product path).

CPU restatement of the third-party helper math the reference's hot path calls:
``omni.isaac.core.utils.torch.{maths,rotations}`` from Isaac Sim 4.0.0 (pinned by
``/root/reference/docker/dockerfile:1`` and ``README.md:33``; NOT vendored in the
reference tree and NOT installed in this image).

PARITY UNPINNED at this layer: the reference holds no test, fixture or golden vector
for these helpers.  The definitions below restate the published
Isaac Gym / Isaac Sim ``torch_utils`` algorithms and are constrained by the reference's
own call sites and by its in-tree Warp restatement, cited per function.  Quaternions
are wxyz (scalar first): ``docs/examples/transfering_policies_from_isaac_gym.md:26-34``.

Everything is plain fp32 torch on CPU, one torch op per arithmetic step, in the
evaluation order of the published helpers, so rounding is the reference's rounding.
Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s cpu_baseline leg may
import this module.
"""
from __future__ import annotations

import math
from typing import Tuple

import torch

Tensor = torch.Tensor
TWO_PI = 2.0 * math.pi


# ----------------------------------------------------------------------------- maths
def tensor_clamp(t: Tensor, lower: Tensor, upper: Tensor) -> Tensor:
    """max(min(t, upper), lower), broadcasting (D,) limits.

    Call site: tasks/shared/locomotion.py:138, tasks/shared/in_hand_manipulation.py:337-361.
    """
    return torch.max(torch.min(t, upper), lower)


def scale(x: Tensor, lower: Tensor, upper: Tensor) -> Tensor:
    """[-1,1] -> [lower, upper]; inverse of :func:`unscale`.

    Call site: tasks/shared/in_hand_manipulation.py:343-347.
    """
    return 0.5 * (x + 1.0) * (upper - lower) + lower


def unscale(x: Tensor, lower: Tensor, upper: Tensor) -> Tensor:
    """[lower, upper] -> [-1,1].

    Call site: tasks/shared/locomotion.py:263; formula confirmed by the in-tree Warp
    restatement tasks/warp/shared/locomotion.py:289-290.
    """
    return (2.0 * x - upper - lower) / (upper - lower)


def torch_rand_float(lower: float, upper: float, shape, generator=None) -> Tensor:
    """(upper-lower)*U[0,1)+lower.

    Call site: tasks/shared/locomotion.py:137,139; spelled out inline at
    utils/domain_randomization/randomize.py:353-355.
    """
    return (upper - lower) * torch.rand(*shape, generator=generator) + lower


def uniform_from_unit(lower: float, upper: float, u: Tensor) -> Tensor:
    """Same affine map as :func:`torch_rand_float`, applied to pre-drawn U[0,1) samples."""
    return (upper - lower) * u + lower


def normalize(x: Tensor, eps: float = 1e-9) -> Tensor:
    """x / max(||x||_2, eps) along the last dim.

    Call sites: tasks/anymal_terrain.py:611 (on quats), compute_heading_and_up (on to_target).
    """
    return x / x.norm(p=2, dim=-1).clamp(min=eps, max=None).unsqueeze(-1)


# ------------------------------------------------------------------------- rotations
def quat_mul(a: Tensor, b: Tensor) -> Tensor:
    """Hamilton product, wxyz, in Isaac's factored 8-multiply form.

    Convention evidence: tasks/factory/factory_control.py:216-221 uses
    quat_mul(q, conj(q))[:, 0] as the squared norm; in_hand_manipulation.py:487 takes
    [:, 1:4] as the vector part.
    """
    shape = a.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 4)
    w1, x1, y1, z1 = a[:, 0], a[:, 1], a[:, 2], a[:, 3]
    w2, x2, y2, z2 = b[:, 0], b[:, 1], b[:, 2], b[:, 3]
    ww = (z1 + x1) * (x2 + y2)
    yy = (w1 - y1) * (w2 + z2)
    zz = (w1 + y1) * (w2 - z2)
    xx = ww + yy + zz
    qq = 0.5 * (xx + (z1 - x1) * (x2 - y2))
    w = qq - ww + (z1 - y1) * (y2 - z2)
    x = qq - xx + (x1 + w1) * (x2 + w2)
    y = qq - yy + (w1 - x1) * (y2 + z2)
    z = qq - zz + (z1 + y1) * (w2 - x2)
    return torch.stack([w, x, y, z], dim=-1).view(shape)


def quat_conjugate(a: Tensor) -> Tensor:
    """[w, -x, -y, -z].  Call site: tasks/shared/locomotion.py:171."""
    shape = a.shape
    a = a.reshape(-1, 4)
    return torch.cat((a[:, 0:1], -a[:, 1:]), dim=-1).view(shape)


def quat_apply(a: Tensor, b: Tensor) -> Tensor:
    """Rotate b by quaternion a via t = 2 (a_vec x b); b + w t + a_vec x t.

    Call site: tasks/anymal_terrain.py:425 (forward vector), :612 (quat_apply_yaw).
    """
    shape = b.shape
    a = a.reshape(-1, 4)
    b = b.reshape(-1, 3)
    xyz = a[:, 1:]
    t = xyz.cross(b, dim=-1) * 2
    return (b + a[:, 0:1] * t + xyz.cross(t, dim=-1)).view(shape)


def _dot3(u: Tensor, v: Tensor) -> Tensor:
    """Row-wise dot product the way the helpers spell it: a (n,1,3)x(n,3,1) bmm."""
    n = u.shape[0]
    return torch.bmm(u.view(n, 1, 3), v.view(n, 3, 1)).squeeze(-1)


def quat_rotate(q: Tensor, v: Tensor) -> Tensor:
    """v(2w^2-1) + 2w(q_vec x v) + 2 q_vec (q_vec . v).

    Call site: get_basis_vector inside compute_heading_and_up (locomotion.py:255).
    """
    q_w = q[:, 0]
    q_vec = q[:, 1:]
    a = v * (2.0 * q_w**2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * _dot3(q_vec, v) * 2.0
    return a + b + c


def random_dir_2_from_unit(u: Tensor) -> Tensor:
    """torch_random_dir_2(shape, device) with its U[0,1) draw supplied: angle = torch_rand_float(-pi, pi, shape).squeeze(-1);
    stack([cos(angle), sin(angle)], -1).  Call site: tasks/ball_balance.py:222.  Restated from the public IsaacGym
    torch_utils lineage: UNPINNED."""
    angle = ((math.pi - (-math.pi)) * u + (-math.pi)).squeeze(-1)
    return torch.stack([torch.cos(angle), torch.sin(angle)], dim=-1)


def torch_random_dir_2(shape, device="cpu") -> Tensor:
    return random_dir_2_from_unit(torch.rand(*shape, device=device))


def quat_axis(q: Tensor, axis: int = 0) -> Tensor:
    """The body axis `axis` in the world frame: quat_rotate(q, e_axis).

    Call sites: tasks/quadcopter.py:241, ingenuity.py:257, crazyflie.py (uprightness terms).  Restated from the public
    IsaacGym torch_utils lineage (basis_vec = zeros(n, 3); basis_vec[:, axis] = 1; quat_rotate(q, basis_vec)): UNPINNED.
    """
    basis_vec = torch.zeros(q.shape[0], 3, device=q.device)
    basis_vec[:, axis] = 1
    return quat_rotate(q, basis_vec)


def quat_rotate_inverse(q: Tensor, v: Tensor) -> Tensor:
    """Same three terms as :func:`quat_rotate` with the cross term subtracted.

    Call sites: compute_rot (locomotion.py:259), tasks/anymal_terrain.py:422-424.
    """
    q_w = q[:, 0]
    q_vec = q[:, 1:]
    a = v * (2.0 * q_w**2 - 1.0).unsqueeze(-1)
    b = torch.cross(q_vec, v, dim=-1) * q_w.unsqueeze(-1) * 2.0
    c = q_vec * _dot3(q_vec, v) * 2.0
    return a - b + c


def quat_from_angle_axis(angle: Tensor, axis: Tensor) -> Tensor:
    """normalize([cos(a/2), normalize(axis) sin(a/2)]).

    Call site: tasks/shared/in_hand_manipulation.py:451-453.
    """
    theta = (angle / 2).unsqueeze(-1)
    xyz = normalize(axis) * theta.sin()
    w = theta.cos()
    return normalize(torch.cat([w, xyz], dim=-1))


def get_basis_vector(q: Tensor, v: Tensor) -> Tensor:
    return quat_rotate(q, v)


def copysign(a: float, b: Tensor) -> Tensor:
    """|a| * sign(b) as fp32 (so sign(0) = 0 gives 0, as in the published helper)."""
    a_t = torch.tensor(a, dtype=torch.float).repeat(b.shape[0])
    return torch.abs(a_t) * torch.sign(b)


def get_euler_xyz(q: Tensor) -> Tuple[Tensor, Tensor, Tensor]:
    """Roll/pitch/yaw of a wxyz quaternion, each wrapped with ``% 2*pi``.

    Formulas (incl. the gimbal branch and the modulo) confirmed by the in-tree Warp
    restatement tasks/warp/shared/locomotion.py:253-276.
    """
    qw, qx, qy, qz = 0, 1, 2, 3
    sinr_cosp = 2.0 * (q[:, qw] * q[:, qx] + q[:, qy] * q[:, qz])
    cosr_cosp = q[:, qw] * q[:, qw] - q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] + q[:, qz] * q[:, qz]
    roll = torch.atan2(sinr_cosp, cosr_cosp)

    sinp = 2.0 * (q[:, qw] * q[:, qy] - q[:, qz] * q[:, qx])
    pitch = torch.where(torch.abs(sinp) >= 1, copysign(math.pi / 2.0, sinp), torch.asin(sinp))

    siny_cosp = 2.0 * (q[:, qw] * q[:, qz] + q[:, qx] * q[:, qy])
    cosy_cosp = q[:, qw] * q[:, qw] + q[:, qx] * q[:, qx] - q[:, qy] * q[:, qy] - q[:, qz] * q[:, qz]
    yaw = torch.atan2(siny_cosp, cosy_cosp)

    two_pi = 2.0 * math.pi
    return roll % two_pi, pitch % two_pi, yaw % two_pi


def compute_heading_and_up(
    torso_rotation: Tensor, inv_start_rot: Tensor, to_target: Tensor, vec0: Tensor, vec1: Tensor, up_idx: int
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Arity/order fixed by the call site tasks/shared/locomotion.py:255-257; the
    projections by tasks/warp/shared/locomotion.py:335-339."""
    num_envs = torso_rotation.shape[0]
    target_dirs = normalize(to_target)
    torso_quat = quat_mul(torso_rotation, inv_start_rot)
    up_vec = get_basis_vector(torso_quat, vec1).view(num_envs, 3)
    heading_vec = get_basis_vector(torso_quat, vec0).view(num_envs, 3)
    up_proj = up_vec[:, up_idx]
    heading_proj = torch.bmm(heading_vec.view(num_envs, 1, 3), target_dirs.view(num_envs, 3, 1)).view(num_envs)
    return torso_quat, up_proj, heading_proj, up_vec, heading_vec


def compute_rot(
    torso_quat: Tensor, velocity: Tensor, ang_velocity: Tensor, targets: Tensor, torso_positions: Tensor
) -> Tuple[Tensor, Tensor, Tensor, Tensor, Tensor, Tensor]:
    """Arity fixed by tasks/shared/locomotion.py:259-261.  walk_target_angle uses the
    z (not y) component, as tasks/warp/shared/locomotion.py:347-348 shows."""
    vel_loc = quat_rotate_inverse(torso_quat, velocity)
    angvel_loc = quat_rotate_inverse(torso_quat, ang_velocity)
    roll, pitch, yaw = get_euler_xyz(torso_quat)
    walk_target_angle = torch.atan2(targets[:, 2] - torso_positions[:, 2], targets[:, 0] - torso_positions[:, 0])
    angle_to_target = walk_target_angle - yaw
    return vel_loc, angvel_loc, roll, pitch, yaw, angle_to_target


HELPER_NAMES = (
    "tensor_clamp scale unscale torch_rand_float normalize quat_mul quat_conjugate quat_apply "
    "quat_rotate quat_rotate_inverse quat_axis quat_from_angle_axis get_basis_vector copysign get_euler_xyz "
    "compute_heading_and_up compute_rot"
).split()


def tf_combine(q1: Tensor, t1: Tensor, q2: Tensor, t2: Tensor) -> Tuple[Tensor, Tensor]:
    """Compose two rigid transforms: (q1 * q2, q1 applied to t2, plus t1).  Restated omni helper (parity unpinned);
    call sites: tasks/franka_cabinet.py:253, :487-492."""
    return quat_mul(q1, q2), quat_apply(q1, t2) + t1


def tf_vector(q: Tensor, v: Tensor) -> Tensor:
    """Rotate a vector by a quaternion.  Call sites: tasks/franka_cabinet.py:532-535."""
    return quat_apply(q, v)


def tf_inverse(q: Tensor, t: Tensor) -> Tuple[Tensor, Tensor]:
    """Inverse of a rigid transform.  Call site: tasks/franka_cabinet.py:250 (set-up only)."""
    q_inv = quat_conjugate(q)
    return q_inv, -quat_apply(q_inv, t)
