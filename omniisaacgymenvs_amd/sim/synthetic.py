"""Stand-in for the delegated layer L0 (PhysX tensor API): synthetic state behind the getters
the task math reads and the setters it writes.  There is no Isaac Sim on ROCm; physics stays
out of scope (BASELINE.json north_star), so ``SyntheticWorld.step`` integrates nothing.

The view mirrors the slice of ``omni.isaac.core`` ArticulationView the reference's hot path
uses (tasks/shared/locomotion.py:76-84,128-149; tasks/cartpole.py:92-93,122,139-140):
getters with ``clone=False`` hand out the internal buffers, setters scatter rows at ``indices``.

Extension for the sync-free path: every setter also accepts ``count=<device int64 scalar>``
with full-capacity ``values``/``indices`` buffers (rows past ``count`` are undefined).  PhysX
would consume (values, indices, count) on the device; this stand-in either applies them with a
host read of ``count`` (``apply_resets=True``: functional tests), drops them
(``apply_resets=False``: benchmarking the task math alone), or applies them on the device without
reading ``count`` (``apply_resets="device"``: a full-capacity masked scatter, legal inside HIP graph
capture -- what ``LocomotionTask.capture_step`` needs).
"""
from __future__ import annotations

from typing import Dict, Optional

import torch

Tensor = torch.Tensor


class SyntheticWorld:
    """``env.world`` as the tasks and VecEnvRLGames use it (vec_env_rlgames.py:81-90, locomotion.py:118)."""

    def __init__(self) -> None:
        self._playing = True
        self.steps = 0

    def is_playing(self) -> bool:
        return self._playing

    def play(self) -> None:
        self._playing = True

    def stop(self) -> None:
        self._playing = False

    def step(self, render: bool = False) -> None:
        self.steps += 1

    def reset(self) -> None:
        self.steps = 0


class SyntheticArticulationView:
    def __init__(
        self,
        name: str,
        count: int,
        num_dof: int,
        device,
        num_bodies: int = 1,
        dof_limits: Optional[Tensor] = None,
        dof_names: Optional[Dict[str, int]] = None,
        body_names: Optional[Dict[str, int]] = None,
        apply_resets: bool = True,
    ) -> None:
        self.name = name
        self.count = count
        self.num_dof = num_dof
        self.num_bodies = num_bodies
        self._device = torch.device(device)
        self.apply_resets = apply_resets
        self.sensor_bodies_only = True  # measured_joint_forces holds only the force-sensor bodies, in task order
        f = dict(dtype=torch.float32, device=self._device)
        self.positions = torch.zeros((count, 3), **f)
        self.orientations = torch.zeros((count, 4), **f)
        self.orientations[:, 0] = 1.0
        self.velocities = torch.zeros((count, 6), **f)
        self.joint_positions = torch.zeros((count, num_dof), **f)
        self.joint_velocities = torch.zeros((count, num_dof), **f)
        self.joint_efforts = torch.zeros((count, num_dof), **f)
        self.joint_position_targets = torch.zeros((count, num_dof), **f)
        self.measured_joint_forces = torch.zeros((count, num_bodies, 6), **f)
        if dof_limits is None:
            dof_limits = torch.stack([-torch.ones(num_dof), torch.ones(num_dof)], dim=-1)
        self._dof_limits = dof_limits.to(**f).unsqueeze(0)  # (1, D, 2); the tasks read row 0 (ant.py:90-92)
        self._dof_names = dof_names or {}
        self._body_indices = body_names or {}

    # ------------------------------------------------------------------ getters
    def get_world_poses(self, clone: bool = True):
        if clone:
            return self.positions.clone(), self.orientations.clone()
        return self.positions, self.orientations

    def get_velocities(self, clone: bool = True) -> Tensor:
        return self.velocities.clone() if clone else self.velocities

    def get_joint_positions(self, clone: bool = True) -> Tensor:
        return self.joint_positions.clone() if clone else self.joint_positions

    def get_joint_velocities(self, clone: bool = True) -> Tensor:
        return self.joint_velocities.clone() if clone else self.joint_velocities

    def get_measured_joint_forces(self, joint_indices: Optional[Tensor] = None, clone: bool = False) -> Tensor:
        """(N, len(joint_indices), 6).  The synthetic view stores exactly the sensor bodies the task
        registered (``sensor_bodies_only``), so the common call returns the buffer itself."""
        if joint_indices is None or self.sensor_bodies_only:
            return self.measured_joint_forces
        return self.measured_joint_forces[:, joint_indices, :]

    def get_dof_limits(self) -> Tensor:
        return self._dof_limits

    def get_dof_index(self, name: str) -> int:
        return self._dof_names[name]

    # ------------------------------------------------------------------ setters
    def _rows(self, indices, count):
        if count is None:
            return None if indices is None else indices.long()
        if not self.apply_resets:
            return False
        n = int(count.item())  # the host read PhysX would not need; test mode only
        return indices[:n].long()

    def _is_identity(self, indices, rows: int) -> bool:
        """``indices`` is arange(rows)?  Checked once per index tensor (one host read), remembered by identity + version."""
        memo = self.__dict__.setdefault("_identity_memo", {})
        hit = memo.get(id(indices))
        if hit is not None and hit[0] is indices and hit[1] == indices._version:
            return hit[2]
        ok = indices.shape[0] == rows and bool(torch.equal(indices.long(), torch.arange(rows, device=indices.device)))
        memo[id(indices)] = (indices, indices._version, ok)
        return ok

    def _scatter(self, buf: Tensor, values: Tensor, indices, count) -> None:
        if (values.data_ptr() == buf.data_ptr() and values.shape == buf.shape
                and (indices is None or self._is_identity(indices, buf.shape[0]))):
            # the caller edited the view's own buffer in place (clone=False getter, e.g. push_robots) and hands it back row for
            # row: nothing to move, whatever `count` says (a row that "does not land" already holds its own value).  Any other
            # index list is a real scatter
            return
        if values.data_ptr() == buf.data_ptr():
            values = values.clone()  # a scatter of the buffer onto itself: read everything before the first write
        if count is not None and self.apply_resets == "device":
            # rows k >= count must not land: point them at row 0's target and give them row 0's value (duplicates then write
            # identical data), or -- when count == 0 -- at a row that is rewritten with its own current value
            cap = indices.shape[0]
            valid = torch.arange(cap, device=buf.device) < count
            idx0 = indices[:1].long().clamp(0, buf.shape[0] - 1)          # (1,): indexing with a 0-d tensor would read it back
            idx = torch.where(valid, indices.long(), idx0)
            v0 = torch.where(count.reshape(-1)[:1] > 0, values[0], buf.index_select(0, idx0)[0])
            vals = torch.where(valid.reshape(-1, *([1] * (values.dim() - 1))), values, v0)
            buf.index_put_((idx,), vals)
            return
        rows = self._rows(indices, count)
        if rows is False:
            return
        if rows is None:
            buf.copy_(values)
        else:
            buf[rows] = values[: rows.shape[0]]

    def set_joint_positions(self, positions: Tensor, indices=None, count=None) -> None:
        self._scatter(self.joint_positions, positions, indices, count)

    def set_joint_velocities(self, velocities: Tensor, indices=None, count=None) -> None:
        if (self.apply_resets is False and count is None and (indices is None or indices.shape[0] == self.count)
                and velocities.shape == self.joint_velocities.shape):
            # benchmark mode, all envs (ingenuity.py:240, crazyflie.py:258: the rotor spin of every env, every step): keep a
            # reference, as set_joint_position_targets does -- the N*D copy is the PhysX stand-in's work, not the task's
            # (a 6 us launch and its gap per step at 1 M envs, a tenth of Ingenuity's step)
            self.joint_velocities = velocities
        else:
            self._scatter(self.joint_velocities, velocities, indices, count)

    def set_joint_efforts(self, efforts: Tensor, indices=None, count=None) -> None:
        # the common case (all envs, arange indices) keeps a reference instead of copying N*D floats
        if indices is None or (count is None and indices.shape[0] == self.count):
            self.joint_efforts = efforts
        else:
            self._scatter(self.joint_efforts, efforts, indices, count)

    def set_joint_position_targets(self, positions: Tensor, indices=None, joint_indices=None, count=None) -> None:
        if joint_indices is not None:
            # all envs, a subset of joints (in_hand_manipulation.py:360-362): keep the (N, len(joint_indices)) block
            self.joint_position_targets_subset = (positions, joint_indices)
        elif self.apply_resets is False and (indices is None or (count is None and indices.shape[0] == self.count)):
            # benchmark mode (reset hand-overs dropped), all envs: keep a reference, like set_joint_efforts.  PhysX consumes
            # the caller's tensor; an N*D copy or index_put here bills the stand-in's work to the task step (50 us per step
            # on flat Anymal at 1 M envs).  Not in the functional modes: a later reset scatter would write through the alias.
            self.joint_position_targets = positions
        elif indices is None:
            self.joint_position_targets.copy_(positions)
        else:
            self._scatter(self.joint_position_targets, positions, indices, count)

    def set_world_poses(self, positions: Optional[Tensor] = None, orientations: Optional[Tensor] = None, indices=None,
                        count=None) -> None:
        if positions is not None:
            self._scatter(self.positions, positions, indices, count)
        if orientations is not None:
            self._scatter(self.orientations, orientations, indices, count)

    def set_velocities(self, velocities: Tensor, indices=None, count=None) -> None:
        self._scatter(self.velocities, velocities, indices, count)


class SyntheticRigidPrimView:
    """RigidPrimView slice the hand task uses (in_hand_manipulation.py:257-259,394,413-414;
    shadow_hand.py:123,127): world poses and velocities of ``count`` rigid bodies."""

    def __init__(self, name: str, count: int, device, apply_resets: bool = True) -> None:
        self.name = name
        self.count = count
        self._device = torch.device(device)
        self.apply_resets = apply_resets
        f = dict(dtype=torch.float32, device=self._device)
        self.positions = torch.zeros((count, 3), **f)
        self.orientations = torch.zeros((count, 4), **f)
        self.orientations[:, 0] = 1.0
        self.velocities = torch.zeros((count, 6), **f)

    def get_world_poses(self, clone: bool = True):
        if clone:
            return self.positions.clone(), self.orientations.clone()
        return self.positions, self.orientations

    def get_velocities(self, clone: bool = True) -> Tensor:
        return self.velocities.clone() if clone else self.velocities

    _rows = SyntheticArticulationView._rows
    _scatter = SyntheticArticulationView._scatter

    def set_world_poses(self, positions: Optional[Tensor] = None, orientations: Optional[Tensor] = None, indices=None,
                        count=None) -> None:
        if positions is not None:
            self._scatter(self.positions, positions, indices, count)
        if orientations is not None:
            self._scatter(self.orientations, orientations, indices, count)

    def set_velocities(self, velocities: Tensor, indices=None, count=None) -> None:
        self._scatter(self.velocities, velocities, indices, count)


class SyntheticContactView(SyntheticRigidPrimView):
    """RigidPrimView with ``track_contact_forces=True`` as AnymalView's ``_knees`` / ``_base`` are used
    (tasks/anymal_terrain.py:389,459,462): world poses plus net contact forces (count, 3)."""

    def __init__(self, name: str, count: int, device, apply_resets: bool = True) -> None:
        super().__init__(name, count, device, apply_resets)
        self.net_contact_forces = torch.zeros((count, 3), dtype=torch.float32, device=self._device)

    def get_net_contact_forces(self, clone: bool = True) -> Tensor:
        return self.net_contact_forces.clone() if clone else self.net_contact_forces
