"""QuadcopterTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/quadcopter.py:42-261, whose arithmetic
runs in liboige_hip.so (csrc/quadcopter.hip).  Same constructor, attributes and methods: ``get_observations /
calculate_metrics / is_done / pre_physics_step / reset_idx / post_reset`` plus the fused ``post_physics_step``.
No torch/CPU fallback.

``reset_idx(env_ids)`` draws with torch in the reference's order ((n,8) DOF positions, then x, y, z (n,1) each,
:206-212); the sync-free ``pre_physics_step`` uses in-kernel Philox keyed by (seed, step, env id).  The reference writes
the re-drawn DOF state into the view's buffers in place (:206-207); here the rows go through the setters.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .base.rl_task import RLTask


class SyntheticQuadcopterView(SyntheticArticulationView):
    """QuadcopterView surface used by the task: 8 DOFs, ``rotors`` with ``apply_forces`` (views/quadcopter_view.py)."""

    def __init__(self, name: str, count: int, device) -> None:
        limits = torch.stack([-0.8 * torch.ones(8), 0.8 * torch.ones(8)], dim=-1)  # stand-in: the real ones live in the USD
        super().__init__(name=name, count=count, num_dof=8, device=device, dof_limits=limits)
        self.rotors = SyntheticRigidPrimView("rotors_view", count * 4, device)
        self.rotors.applied_forces = None
        self.rotors.apply_forces = self._apply_forces

    def _apply_forces(self, forces, indices=None, is_global=False):
        self.rotors.applied_forces = forces


class QuadcopterTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._num_observations = 21
        self._num_actions = 12
        self._copter_position = torch.tensor([0, 0, 1.0])
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
        self._step_count = 0
        RLTask.__init__(self, name=name, env=env)
        max_thrust = 2.0
        self.thrust_lower_limits = -max_thrust * torch.ones(4, device=self._device, dtype=torch.float32)
        self.thrust_upper_limits = max_thrust * torch.ones(4, device=self._device, dtype=torch.float32)
        self.all_indices = torch.arange(self._num_envs, dtype=torch.int32, device=self._device)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._num_envs = self._task_cfg["env"]["numEnvs"]
        self._env_spacing = self._task_cfg["env"]["envSpacing"]
        self._max_episode_length = self._task_cfg["env"]["maxEpisodeLength"]
        self.dt = self._task_cfg["sim"]["dt"]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._copters = SyntheticQuadcopterView("quadcopter_view", self._num_envs, self.device)
        self._copters.positions[:] = self._env_pos + self._copter_position.to(self.device)
        self._balls = SyntheticRigidPrimView("targets_view", self._num_envs, self.device)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def post_reset(self):
        n, dev = self._num_envs, self._device
        f = dict(dtype=torch.float32, device=dev)
        self.dof_position_targets = torch.zeros((n, self._copters.num_dof), **f)
        self.thrusts = torch.zeros((n, 4), **f)
        self.forces = torch.zeros((n, self._copters.rotors.count // max(n, 1), 3), **f)
        self.target_positions = torch.zeros((n, 3), device=dev)
        self.target_positions[:, 2] = 1.0
        self.root_pos, self.root_rot = self._copters.get_world_poses(clone=False)
        self.root_velocities = self._copters.get_velocities(clone=False)
        self.dof_pos = self._copters.get_joint_positions(clone=False)
        self.dof_vel = self._copters.get_joint_velocities(clone=False)
        self.initial_root_pos, self.initial_root_rot = self.root_pos.clone(), self.root_rot.clone()
        dof_limits = self._copters.get_dof_limits()
        self.dof_lower_limits = dof_limits[0][:, 0].to(device=dev)
        self.dof_upper_limits = dof_limits[0][:, 1].to(device=dev)

        # device-side scratch of the HIP path
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "dof_pos": torch.zeros((n, 8), **f), "dof_vel": torch.zeros((n, 8), **f),
            "root_pos": torch.zeros((n, 3), **f), "root_rot": torch.zeros((n, 4), **f), "root_vel": torch.zeros((n, 6), **f),
        }
        self._counts_key = None
        self._host_consts = {"lower": self.dof_lower_limits.tolist(), "upper": self.dof_upper_limits.tolist(),
                             "max_thrust": float(self.thrust_upper_limits[0])}

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    def _params(self) -> _lib.QuadcopterParams:
        return self._block("params", (self._host_consts,), (self.dt, self.clip_actions, self._max_episode_length), self._build_params)

    def _build_params(self) -> _lib.QuadcopterParams:
        p = _lib.QuadcopterParams()
        p.dof_rate = self.dt * (8 * math.pi)     # self.dt * dof_action_speed_scale (:153-154), a Python double product
        p.thrust_rate = self.dt * 100            # :159-160
        p.max_thrust = self._host_consts["max_thrust"]  # read back once in post_reset: a per-step float(tensor) is a device sync
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        p.max_episode_length = int(self._max_episode_length)
        for i in range(8):
            p.dof_lower[i], p.dof_upper[i] = self._host_consts["lower"][i], self._host_consts["upper"][i]
        return p

    def _buffers(self) -> _lib.QuadcopterBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.target_positions, self.dof_position_targets,
              self.thrusts, self.forces, self._reset_workspace, self._tile_counts_u8)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.QuadcopterBuffers:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 21), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.target_positions, torch.float32, (n, 3), "target_positions")
        rq(self.dof_position_targets, torch.float32, (n, 8), "dof_position_targets")
        rq(self.thrusts, torch.float32, (n, 4), "thrusts")
        rq(self.forces, torch.float32, (n, 4, 3), "forces")
        b = _lib.QuadcopterBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.target_positions, b.dof_position_targets = self.target_positions.data_ptr(), self.dof_position_targets.data_ptr()
        b.thrusts, b.forces, b.reset_workspace = self.thrusts.data_ptr(), self.forces.data_ptr(), self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        return b

    def _state(self) -> _lib.QuadcopterState:
        self.root_pos, self.root_rot = self._copters.get_world_poses(clone=False)
        self.root_velocities = self._copters.get_velocities(clone=False)
        self.dof_pos = self._copters.get_joint_positions(clone=False)
        ts = (self.root_pos, self.root_rot, self.root_velocities, self.dof_pos, self._env_pos)
        return self._block("state", ts, (), self._build_state)

    def _build_state(self) -> _lib.QuadcopterState:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.root_pos, torch.float32, (n, 3), "root_pos")
        rq(self.root_rot, torch.float32, (n, 4), "root_rot")
        rq(self.root_velocities, torch.float32, (n, 6), "root_velocities")
        rq(self.dof_pos, torch.float32, (n, 8), "dof_pos")
        rq(self._env_pos, torch.float32, (n, 3), "_env_pos")
        s = _lib.QuadcopterState()
        s.root_pos, s.root_rot, s.root_velocities = self.root_pos.data_ptr(), self.root_rot.data_ptr(), self.root_velocities.data_ptr()
        s.dof_pos, s.env_pos = self.dof_pos.data_ptr(), self._env_pos.data_ptr()
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_quadcopter_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                      _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_quadcopter_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        return {self._copters.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        self._post(_lib.STAGE_METRICS)

    def is_done(self) -> None:
        self._post(_lib.STAGE_DONE)

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        self._post(_lib.STAGE_ALL if self._env.world.is_playing() else _lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _reset_io(self, u_dof=None, u_xyz=None) -> _lib.QuadcopterResetIO:
        ts = (self.initial_root_pos, self.initial_root_rot, u_dof, u_xyz, *self._reset_out.values(),
              self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u_dof, u_xyz))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u_dof, u_xyz) -> _lib.QuadcopterResetIO:
        r = _lib.QuadcopterResetIO()
        r.initial_root_pos, r.initial_root_rot = self.initial_root_pos.data_ptr(), self.initial_root_rot.data_ptr()
        if u_dof is not None:
            r.u_dof, r.u_xyz = u_dof.data_ptr(), u_xyz.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        r.root_pos_out, r.root_rot_out, r.root_vel_out = o["root_pos"].data_ptr(), o["root_rot"].data_ptr(), o["root_vel"].data_ptr()
        return r

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, 12), "actions")
        return actions

    def _apply_reset_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:
            kw, sl = dict(indices=o["env_ids"], count=o["count"]), slice(None)
        else:
            kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
        self._copters.set_joint_positions(o["dof_pos"][sl], **kw)
        self._copters.set_joint_velocities(o["dof_vel"][sl], **kw)
        self._copters.set_world_poses(o["root_pos"][sl], o["root_rot"][sl], **kw)
        self._copters.set_velocities(o["root_vel"][sl], **kw)

    def pre_physics_step(self, actions) -> None:
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        self._advance_step_counter()
        flags = 0
        if not self.sync_free:
            o = self._reset_out
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            num = int(o["count"].item())
            ids = o["env_ids"][:num].clone()
            if num > 0:
                self.reset_idx(ids)
        elif self._counts_valid(self.reset_buf):
            flags |= _lib.PRE_COUNTS_VALID
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_quadcopter_pre_physics(C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), n, flags,
                                                     _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_quadcopter_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        else:
            # the reference's flow: the flags were cleared by reset_idx before the kernel ran, so clear the actions of the
            # reset envs here as :168-171 does
            if num > 0:
                self.thrusts[ids] = 0.0
                self.forces[ids] = 0.0
                self.dof_position_targets[ids] = self._copters.joint_positions[ids]
        self._copters.set_joint_position_targets(self.dof_position_targets)
        self._copters.rotors.apply_forces(self.forces, is_global=False)

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        # same four draws, same order as the reference (:206, :210-212)
        u_dof = torch.rand(num_resets, 8, device=dev)
        u_xyz = torch.cat([torch.rand(num_resets, 1, device=dev), torch.rand(num_resets, 1, device=dev),
                           torch.rand(num_resets, 1, device=dev)], dim=-1).contiguous()
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io(u_dof, u_xyz)
        rc = _lib.load().oige_quadcopter_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets,
                                                   _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_quadcopter_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num_resets)
        self._balls.set_world_poses(positions=self.target_positions[:, 0:3] + self._env_pos)
