"""LocomotionTask -- host-side mirror of the reference class
(/root/reference/omniisaacgymenvs/tasks/shared/locomotion.py:42-211) whose arithmetic runs in
liboige_hip.so (csrc/locomotion.hip) through the C ABI of include/oige_hip.h.

Same method names, argument meaning and buffer contract as the reference:
``get_observations / pre_physics_step / reset_idx / post_reset / calculate_metrics / is_done``,
plus a fused ``post_physics_step`` (one launch: progress, obs, reward, done in the
rl_task.py:289-295 order).  There is no torch/CPU fallback: without the HIP library or a ROCm
device every method raises.

Two reset paths:
  * ``reset_idx(env_ids)`` draws with torch exactly like the reference (two ``torch.rand``
    of shape (n, D), :137 and :139) and hands the draws to the kernel -> same numbers as the
    reference for the same torch RNG state;
  * ``pre_physics_step`` (default ``sync_free=True``) never reads ``len(ids)`` on the host: ids
    are compacted on the device and the draws come from an in-kernel Philox4x32-10 keyed by
    (seed, step, env id).  ``sync_free=False`` reproduces the reference's flow (host count,
    torch draws).
"""
from __future__ import annotations

import ctypes as C
from abc import abstractmethod

import torch

from ... import _lib
from ..base.rl_task import RLTask


class LocomotionTask(RLTask):
    # overridden by AntLocomotionTask / HumanoidLocomotionTask
    _limit_cost_kind = 0
    _num_sensors = 0

    def __init__(self, name, env, offset=None) -> None:
        LocomotionTask.update_config(self)
        self.fused = True        # post_physics_step as one kernel
        self.sync_free = True    # pre_physics_step without the host read of len(reset_env_ids)
        self.materialize_vecs = True  # write up_vec / heading_vec back (locomotion.py:90-91)
        self._step_count = 0
        self.supports_action_clamp = True
        RLTask.__init__(self, name, env)

    def update_config(self):
        e = self._task_cfg["env"]
        self._num_envs = e["numEnvs"]
        self._env_spacing = e["envSpacing"]
        self._max_episode_length = e["episodeLength"]
        self.dof_vel_scale = e["dofVelocityScale"]
        self.angular_velocity_scale = e["angularVelocityScale"]
        self.contact_force_scale = e["contactForceScale"]
        self.power_scale = e["powerScale"]
        self.heading_weight = e["headingWeight"]
        self.up_weight = e["upWeight"]
        self.actions_cost_scale = e["actionsCost"]
        self.energy_cost_scale = e["energyCost"]
        self.joints_at_limit_cost_scale = e["jointsAtLimitCost"]
        self.death_cost = e["deathCost"]
        self.termination_height = e["terminationHeight"]
        self.alive_reward_scale = e["alive_reward_scale"]

    @abstractmethod
    def set_up_scene(self, scene) -> None:
        pass

    @abstractmethod
    def get_robot(self):
        pass

    # ------------------------------------------------------------------ C-ABI argument blocks
    _PARAM_SCALARS = ("dt", "dof_vel_scale", "contact_force_scale", "angular_velocity_scale", "heading_weight", "up_weight",
                      "actions_cost_scale", "energy_cost_scale", "joints_at_limit_cost_scale", "death_cost", "termination_height",
                      "alive_reward_scale", "_max_episode_length", "power_scale", "clip_actions")

    def _params(self) -> _lib.LocomotionParams:
        """Rebuilt when a public scalar (e.g. ``task.heading_weight``) or the host copy of the constant tables changed."""
        d = self.__dict__
        return self._block("params", (self._host_consts,), [d[k] for k in self._PARAM_SCALARS], self._build_params)

    def _build_params(self) -> _lib.LocomotionParams:
        p = _lib.LocomotionParams()
        d = self._robots.num_dof
        p.num_dof, p.num_sensors, p.limit_cost_kind = d, self._num_sensors, self._limit_cost_kind
        p.dt = self.dt
        p.dof_vel_scale = self.dof_vel_scale
        p.contact_force_scale = self.contact_force_scale
        p.angular_velocity_scale = self.angular_velocity_scale
        p.heading_weight = self.heading_weight
        p.up_weight = self.up_weight
        p.actions_cost_scale = self.actions_cost_scale
        p.energy_cost_scale = self.energy_cost_scale
        p.joints_at_limit_cost_scale = self.joints_at_limit_cost_scale
        p.death_cost = self.death_cost
        p.termination_height = self.termination_height
        p.alive_reward_scale = self.alive_reward_scale
        p.max_episode_length = float(self._max_episode_length)
        p.power_scale = self.power_scale
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        c = self._host_consts
        p.target[:] = c["target"]
        p.inv_start_rot[:] = c["inv_start_rot"]
        p.basis_vec0[:] = c["basis_vec0"]
        p.basis_vec1[:] = c["basis_vec1"]
        for name in ("dof_limits_lower", "dof_limits_upper", "joint_gears", "motor_effort_ratio"):
            arr = getattr(p, name)
            vals = c[name]
            for j in range(d):
                arr[j] = vals[j]
        return p

    def _refresh_host_consts(self) -> None:
        """Row 0 of the per-env-constant tensors + the (D,) tables, read to the host once (a NEW dict: the params block's key)."""
        self._host_consts = {
            "target": self.targets[0].tolist(),
            "inv_start_rot": self.inv_start_rot[0].tolist(),
            "basis_vec0": self.basis_vec0[0].tolist(),
            "basis_vec1": self.basis_vec1[0].tolist(),
            "dof_limits_lower": self.dof_limits_lower.tolist(),
            "dof_limits_upper": self.dof_limits_upper.tolist(),
            "joint_gears": self.joint_gears.tolist(),
            "motor_effort_ratio": self.motor_effort_ratio.tolist(),
        }

    def _buffers(self) -> _lib.LocomotionBuffers:
        vecs = self.materialize_vecs
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.potentials, self.prev_potentials, self.actions,
              self.up_vec if vecs else None, self.heading_vec if vecs else None, self._reset_workspace, self._tile_counts_u8,
              self._stats_slots)
        return self._block("buffers", ts, (), self._build_buffers)

    _stats_slots = None

    def enable_episode_stats(self) -> torch.Tensor:
        """Multi-GPU runs (utils/distributed.py: StatsReducer): the fused post kernel keeps this rank's episode statistics --
        per 64-env tile {sum reward, done envs, sum of their episode lengths, 0} as running f64 sums -- in the returned
        (tiles, 4) tensor: no launch and no traffic per step."""
        if self._stats_slots is None:
            tiles = max((self._num_envs + 63) // 64, 1)
            self._stats_slots = torch.zeros((tiles, 4), dtype=torch.float64, device=self.obs_buf.device)
        return self._stats_slots

    def _build_buffers(self) -> _lib.LocomotionBuffers:
        n, d = self._num_envs, self._robots.num_dof
        rq = _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, self._num_observations), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.potentials, torch.float32, (n,), "potentials")
        rq(self.prev_potentials, torch.float32, (n,), "prev_potentials")
        rq(self.actions, torch.float32, (n, d), "actions")
        b = _lib.LocomotionBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.potentials, b.prev_potentials = self.potentials.data_ptr(), self.prev_potentials.data_ptr()
        if self.materialize_vecs:
            rq(self.up_vec, torch.float32, (n, 3), "up_vec")
            rq(self.heading_vec, torch.float32, (n, 3), "heading_vec")
            b.up_vec, b.heading_vec = self.up_vec.data_ptr(), self.heading_vec.data_ptr()
        b.actions = self.actions.data_ptr()
        b.reset_workspace = self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        if self._stats_slots is not None:
            b.stats_slots = self._stats_slots.data_ptr()
        return b

    def _state(self) -> _lib.LocomotionState:
        torso_position, torso_rotation = self._robots.get_world_poses(clone=False)
        velocities = self._robots.get_velocities(clone=False)
        dof_pos = self._robots.get_joint_positions(clone=False)
        dof_vel = self._robots.get_joint_velocities(clone=False)
        # force sensors attached to the feet
        sensor_force_torques = self._robots.get_measured_joint_forces(joint_indices=self._sensor_indices)
        live = (torso_position, torso_rotation, velocities, dof_pos, dof_vel, sensor_force_torques)
        return self._block("state", live, (), lambda: self._build_state(*live))

    def _build_state(self, torso_position, torso_rotation, velocities, dof_pos, dof_vel, sensor_force_torques) -> _lib.LocomotionState:
        n, d = self._num_envs, self._robots.num_dof
        rq = _lib.require_device_tensor
        rq(torso_position, torch.float32, (n, 3), "torso_position")
        rq(torso_rotation, torch.float32, (n, 4), "torso_rotation")
        rq(velocities, torch.float32, (n, 6), "velocities")
        rq(dof_pos, torch.float32, (n, d), "dof_pos")
        rq(dof_vel, torch.float32, (n, d), "dof_vel")
        rq(sensor_force_torques, torch.float32, (n, self._num_sensors, 6), "sensor_force_torques")
        s = _lib.LocomotionState()
        s.torso_position, s.torso_rotation = torso_position.data_ptr(), torso_rotation.data_ptr()
        s.velocities = velocities.data_ptr()
        s.dof_pos, s.dof_vel = dof_pos.data_ptr(), dof_vel.data_ptr()
        s.sensor_force_torques = sensor_force_torques.data_ptr()
        return s

    def _post(self, stages: int) -> None:
        state = self._state() if stages & _lib.STAGE_OBS else None
        p, b = self._params(), self._buffers()
        rc = _lib.load().oige_locomotion_post_physics(
            C.byref(p), C.byref(state) if state is not None else None, C.byref(b), self._num_envs, stages,
            _lib.current_stream(self.obs_buf.device),
        )
        _lib.check(rc, "oige_locomotion_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        return {self._robots.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        self._post(_lib.STAGE_METRICS)

    def is_done(self) -> None:
        self._post(_lib.STAGE_DONE)

    def _limit_cost(self, dtype) -> torch.Tensor:
        """``get_dof_at_limit_cost()`` of the subclass (ant.py:102-109 / humanoid.py:130-141) from ``obs_buf``: a fresh
        ``(N,)`` tensor of the reference's dtype (int64 count for Ant, fp32 for Humanoid)."""
        out = torch.empty(self._num_envs, dtype=dtype, device=self.obs_buf.device)
        _lib.require_device_tensor(self.obs_buf, torch.float32, (self._num_envs, self._num_observations), "obs_buf")
        p = self._params()
        _lib.check(_lib.load().oige_locomotion_dof_at_limit_cost(C.byref(p), self.obs_buf.data_ptr(), out.data_ptr(), self._num_envs,
                                                                 _lib.current_stream(self.obs_buf.device)),
                   "oige_locomotion_dof_at_limit_cost")
        return out

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        if self._env.world.is_playing():
            self._post(_lib.STAGE_ALL)
            self.get_states()
            self.get_extras()
        else:
            self._post(_lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _reset_io(self, u_pos=None, u_vel=None) -> _lib.LocomotionResetIO:
        o = self._reset_out
        ts = (self.initial_root_pos, self.initial_root_rot, self.initial_dof_pos, u_pos, u_vel, o["env_ids"], o["count"], o["dof_pos"],
              o["dof_vel"], o["root_pos"], o["root_rot"], o["root_vel"], self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u_pos, u_vel))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u_pos, u_vel) -> _lib.LocomotionResetIO:
        r = _lib.LocomotionResetIO()
        r.initial_root_pos = self.initial_root_pos.data_ptr()
        r.initial_root_rot = self.initial_root_rot.data_ptr()
        r.initial_dof_pos = self.initial_dof_pos.data_ptr()
        if u_pos is not None:
            r.u_pos, r.u_vel = u_pos.data_ptr(), u_vel.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()  # graph capture: the Philox step counter lives on the device
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        r.root_pos_out, r.root_rot_out, r.root_vel_out = (
            o["root_pos"].data_ptr(), o["root_rot"].data_ptr(), o["root_vel"].data_ptr())
        return r

    def _apply_reset_writes(self, n=None) -> None:
        """Hand the compacted rows to the view setters (:145-149).  ``n`` None -> device count."""
        o = self._reset_out
        if n is None:  # whole-capacity buffers + the device-side count: nothing to slice
            ids, cnt, r = o["env_ids"], o["count"], self._robots
            r.set_joint_positions(o["dof_pos"], indices=ids, count=cnt)
            r.set_joint_velocities(o["dof_vel"], indices=ids, count=cnt)
            r.set_world_poses(o["root_pos"], o["root_rot"], indices=ids, count=cnt)
            r.set_velocities(o["root_vel"], indices=ids, count=cnt)
            return
        kw = dict(indices=o["env_ids"][:n])
        sl = slice(0, n)
        self._robots.set_joint_positions(o["dof_pos"][sl], **kw)
        self._robots.set_joint_velocities(o["dof_vel"][sl], **kw)
        self._robots.set_world_poses(o["root_pos"][sl], o["root_rot"][sl], **kw)
        self._robots.set_velocities(o["root_vel"][sl], **kw)

    def pre_physics_step(self, actions) -> None:
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        self._advance_step_counter()
        flags = 0
        if not self.sync_free:
            # the reference's flow: host-visible id list, torch-drawn noise (:121-123)
            o = self._reset_out
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            num = int(o["count"].item())
            if num > 0:
                self.reset_idx(o["env_ids"][:num].clone())
        elif self._counts_valid(self.reset_buf):
            flags |= _lib.PRE_COUNTS_VALID  # the DONE stage's tile counts still describe reset_buf (never baked into a graph)
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS  # the wrapper's clamp (vec_env_rlgames.py:77), folded into the kernel
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_locomotion_pre_physics(
            C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), self.actions.data_ptr(), self._forces.data_ptr(),
            n, flags, _lib.current_stream(self.obs_buf.device),
        )
        _lib.check(rc, "oige_locomotion_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        # applies joint torques
        self._robots.set_joint_efforts(self._forces, indices=self._all_indices)

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, self._robots.num_dof), "actions")
        return actions

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        d = self._robots.num_dof
        env_ids = env_ids.to(device=self._device, dtype=torch.int64).contiguous()
        # randomize DOF positions and velocities: same two draws, same order as the reference
        u_pos = torch.rand(num_resets, d, device=self._device)
        u_vel = torch.rand(num_resets, d, device=self._device)
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io(u_pos, u_vel)
        rc = _lib.load().oige_locomotion_reset_idx(
            C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets, _lib.current_stream(self.obs_buf.device)
        )
        _lib.check(rc, "oige_locomotion_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num_resets)

    def post_reset(self):
        self._robots = self.get_robot()
        self.initial_root_pos, self.initial_root_rot = self._robots.get_world_poses()
        self.initial_dof_pos = self._robots.get_joint_positions()
        dev, n, d = self._device, self.num_envs, self._robots.num_dof

        # initialize some data used later on (:168-184)
        self.start_rotation = torch.tensor([1, 0, 0, 0], device=dev, dtype=torch.float32)
        self.up_vec = torch.tensor([0, 0, 1], dtype=torch.float32, device=dev).repeat((n, 1))
        self.heading_vec = torch.tensor([1, 0, 0], dtype=torch.float32, device=dev).repeat((n, 1))
        inv = self.start_rotation.clone()
        inv[1:] = -inv[1:]  # quat_conjugate
        self.inv_start_rot = inv.repeat((n, 1))
        self.basis_vec0 = self.heading_vec.clone()
        self.basis_vec1 = self.up_vec.clone()
        self.targets = torch.tensor([1000, 0, 0], dtype=torch.float32, device=dev).repeat((n, 1))
        self.target_dirs = torch.tensor([1, 0, 0], dtype=torch.float32, device=dev).repeat((n, 1))
        self.dt = 1.0 / 60.0
        self.potentials = torch.tensor([-1000.0 / self.dt], dtype=torch.float32, device=dev).repeat(n)
        self.prev_potentials = self.potentials.clone()
        self.actions = torch.zeros((n, self.num_actions), device=dev)

        # device-side scratch of the HIP path (never resized)
        f = dict(dtype=torch.float32, device=dev)
        self._forces = torch.zeros((n, d), **f)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, dtype=torch.int64, device=dev),
            "count": torch.zeros(1, dtype=torch.int64, device=dev),
            "dof_pos": torch.zeros((n, d), **f),
            "dof_vel": torch.zeros((n, d), **f),
            "root_pos": torch.zeros((n, 3), **f),
            "root_rot": torch.zeros((n, 4), **f),
            "root_vel": torch.zeros((n, 6), **f),
        }
        self._all_indices = torch.arange(self._robots.count, dtype=torch.int32, device=dev)
        self._counts_key = None
        self._refresh_host_consts()

        # randomize all envs
        indices = torch.arange(self._robots.count, dtype=torch.int64, device=dev)
        self.reset_idx(indices)
