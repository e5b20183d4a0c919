"""AnymalTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/anymal.py:38-314 (flat-ground Anymal
locomotion), whose arithmetic runs in liboige_hip.so (csrc/anymal.hip).  Same constructor, attributes and methods:
``get_observations / calculate_metrics / is_done / pre_physics_step / reset_idx / post_reset`` plus the fused
``post_physics_step``.  No torch/CPU fallback.

``reset_idx(env_ids)`` draws with torch in the reference's order (velocities (n,12), then x, y, yaw (n,1) each, :199,
:217-225); the sync-free ``pre_physics_step`` uses in-kernel Philox keyed by (seed, step, env id).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from .anymal_terrain import ANYMAL_DOF_NAMES, SyntheticAnymalView
from .base.rl_task import RLTask


class AnymalTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._num_observations = 48
        self._num_actions = 12
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
        self._step_count = 0
        RLTask.__init__(self, name, env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        env, learn = self._task_cfg["env"], self._task_cfg["env"]["learn"]
        self.lin_vel_scale = learn["linearVelocityScale"]
        self.ang_vel_scale = learn["angularVelocityScale"]
        self.dof_pos_scale = learn["dofPositionScale"]
        self.dof_vel_scale = learn["dofVelocityScale"]
        self.action_scale = env["control"]["actionScale"]
        self.rew_scales = {
            "lin_vel_xy": learn["linearVelocityXYRewardScale"], "ang_vel_z": learn["angularVelocityZRewardScale"],
            "lin_vel_z": learn["linearVelocityZRewardScale"], "joint_acc": learn["jointAccRewardScale"],
            "action_rate": learn["actionRateRewardScale"], "cosmetic": learn["cosmeticRewardScale"],
        }
        rng = env["randomCommandVelocityRanges"]
        self.command_x_range, self.command_y_range, self.command_yaw_range = rng["linear_x"], rng["linear_y"], rng["yaw"]
        init = env["baseInitState"]
        self.base_init_state = init["pos"] + init["rot"] + init["vLinear"] + init["vAngular"]
        self.named_default_joint_angles = env["defaultJointAngles"]
        self.dt = 1 / 60
        self.max_episode_length_s = learn["episodeLength_s"]
        self.max_episode_length = int(self.max_episode_length_s / self.dt + 0.5)
        self.Kp = env["control"]["stiffness"]
        self.Kd = env["control"]["damping"]
        for key in self.rew_scales.keys():
            self.rew_scales[key] *= self.dt
        self._num_envs = env["numEnvs"]
        self._anymal_translation = torch.tensor([0.0, 0.0, 0.62])
        self._env_spacing = env["envSpacing"]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._anymals = SyntheticAnymalView("anymalview", self._num_envs, self.device)
        self._anymals.positions[:] = self._env_pos + self._anymal_translation.to(self.device)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def post_reset(self):
        n, dev = self.num_envs, self.device
        f = dict(dtype=torch.float, device=dev)
        self.default_dof_pos = torch.zeros((n, 12), **f)
        dof_names = self._anymals.dof_names
        for i in range(self.num_actions):
            self.default_dof_pos[:, i] = self.named_default_joint_angles[dof_names[i]]
        self.initial_root_pos, self.initial_root_rot = self._anymals.get_world_poses()
        self.current_targets = self.default_dof_pos.clone()
        dof_limits = self._anymals.get_dof_limits()
        self.anymal_dof_lower_limits = dof_limits[0, :, 0].to(device=dev)
        self.anymal_dof_upper_limits = dof_limits[0, :, 1].to(device=dev)
        self.commands = torch.zeros(n, 3, **f)
        self.commands_y = self.commands.view(n, 3)[..., 1]
        self.commands_x = self.commands.view(n, 3)[..., 0]
        self.commands_yaw = self.commands.view(n, 3)[..., 2]
        self.extras = {}
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0], device=dev).repeat((n, 1))
        self.actions = torch.zeros(n, self.num_actions, **f)
        self.last_dof_vel = torch.zeros((n, 12), **f)
        self.last_actions = torch.zeros(n, self.num_actions, **f)
        self.time_out_buf = torch.zeros_like(self.reset_buf)

        # device-side scratch of the HIP path
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "dof_pos": torch.zeros((n, 12), **f), "dof_vel": torch.zeros((n, 12), **f),
            "root_pos": torch.zeros((n, 3), **f), "root_rot": torch.zeros((n, 4), **f), "root_vel": torch.zeros((n, 6), **f),
        }
        self._all_indices = torch.arange(n, dtype=torch.int32, device=dev)
        self._counts_key = None
        self._host_consts = {
            "default": self.default_dof_pos[0].tolist() if n > 0 else [0.0] * 12,
            "lower": self.anymal_dof_lower_limits.tolist(), "upper": self.anymal_dof_upper_limits.tolist(),
        }
        indices = torch.arange(self._anymals.count, dtype=torch.int64, device=dev)
        self.reset_idx(indices)

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    def _params(self) -> _lib.AnymalFlatParams:
        sc = (self.lin_vel_scale, self.ang_vel_scale, self.dof_pos_scale, self.dof_vel_scale, self.action_scale, self.dt,
              self.clip_actions, self.max_episode_length, *self.rew_scales.values(), *self.command_x_range, *self.command_y_range,
              *self.command_yaw_range)
        return self._block("params", (self._host_consts,), sc, self._build_params)

    def _build_params(self) -> _lib.AnymalFlatParams:
        p = _lib.AnymalFlatParams()
        p.lin_vel_scale, p.ang_vel_scale = self.lin_vel_scale, self.ang_vel_scale
        p.dof_pos_scale, p.dof_vel_scale = self.dof_pos_scale, self.dof_vel_scale
        p.action_scale, p.dt, p.base_threshold = self.action_scale, self.dt, 0.51
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        p.max_episode_length = self.max_episode_length
        for i, k in enumerate(("lin_vel_xy", "ang_vel_z", "lin_vel_z", "joint_acc", "action_rate", "cosmetic")):
            p.rew_scales[i] = self.rew_scales[k]
        for i, r in enumerate((self.command_x_range, self.command_y_range, self.command_yaw_range)):
            p.cmd_lo[i], p.cmd_span[i] = r[0], r[1] - r[0]
        h = self._host_consts
        for i in range(12):
            p.default_dof_pos[i], p.dof_lower[i], p.dof_upper[i] = h["default"][i], h["lower"][i], h["upper"][i]
        return p

    def _buffers(self) -> _lib.AnymalFlatBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.commands, self.actions, self.last_actions,
              self.last_dof_vel, self.current_targets, self._reset_workspace, self._tile_counts_u8)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.AnymalFlatBuffers:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 48), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.commands, torch.float32, (n, 3), "commands")
        for name in ("actions", "last_actions", "last_dof_vel", "current_targets"):
            rq(getattr(self, name), torch.float32, (n, 12), name)
        b = _lib.AnymalFlatBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.commands, b.actions = self.commands.data_ptr(), self.actions.data_ptr()
        b.last_actions, b.last_dof_vel = self.last_actions.data_ptr(), self.last_dof_vel.data_ptr()
        b.current_targets, b.reset_workspace = self.current_targets.data_ptr(), self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        return b

    def _state(self) -> _lib.AnymalFlatState:
        torso_position, torso_rotation = self._anymals.get_world_poses(clone=False)
        root_velocities = self._anymals.get_velocities(clone=False)
        dof_pos = self._anymals.get_joint_positions(clone=False)
        dof_vel = self._anymals.get_joint_velocities(clone=False)
        live = (torso_position, torso_rotation, root_velocities, dof_pos, dof_vel)
        return self._block("state", live, (), lambda: self._build_state(*live))

    def _build_state(self, torso_position, torso_rotation, root_velocities, dof_pos, dof_vel) -> _lib.AnymalFlatState:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(torso_position, torch.float32, (n, 3), "torso_position")
        rq(torso_rotation, torch.float32, (n, 4), "torso_rotation")
        rq(root_velocities, torch.float32, (n, 6), "root_velocities")
        rq(dof_pos, torch.float32, (n, 12), "dof_pos")
        rq(dof_vel, torch.float32, (n, 12), "dof_vel")
        s = _lib.AnymalFlatState()
        s.base_pos, s.base_rot, s.velocities = torso_position.data_ptr(), torso_rotation.data_ptr(), root_velocities.data_ptr()
        s.dof_pos, s.dof_vel = dof_pos.data_ptr(), dof_vel.data_ptr()
        self._live_state = (torso_position, torso_rotation, root_velocities, dof_pos, dof_vel)
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_anymal_flat_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                       _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_flat_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        return {self._anymals.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        self._post(_lib.STAGE_METRICS)

    def is_done(self) -> None:
        self._post(_lib.STAGE_DONE)

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        if self._env.world.is_playing():
            self._post(_lib.STAGE_ALL)
        else:
            self._post(_lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _reset_io(self, u_vel=None, u_cmd=None) -> _lib.AnymalFlatResetIO:
        ts = (self.initial_root_pos, self.initial_root_rot, u_vel, u_cmd, *self._reset_out.values(),
              self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u_vel, u_cmd))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u_vel, u_cmd) -> _lib.AnymalFlatResetIO:
        r = _lib.AnymalFlatResetIO()
        r.initial_root_pos, r.initial_root_rot = self.initial_root_pos.data_ptr(), self.initial_root_rot.data_ptr()
        if u_vel is not None:
            r.u_vel, r.u_cmd = u_vel.data_ptr(), u_cmd.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        r.root_pos_out, r.root_rot_out, r.root_vel_out = o["root_pos"].data_ptr(), o["root_rot"].data_ptr(), o["root_vel"].data_ptr()
        return r

    def _apply_reset_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:
            kw, sl = dict(indices=o["env_ids"], count=o["count"]), slice(None)
        else:
            kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
        self._anymals.set_joint_positions(o["dof_pos"][sl], **kw)
        self._anymals.set_joint_velocities(o["dof_vel"][sl], **kw)
        self._anymals.set_world_poses(o["root_pos"][sl], o["root_rot"][sl], **kw)
        self._anymals.set_velocities(o["root_vel"][sl], **kw)

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, 12), "actions")
        return actions

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
            if num > 0:
                self.reset_idx(o["env_ids"][:num].clone())
        elif self._counts_valid(self.reset_buf):
            flags |= _lib.PRE_COUNTS_VALID
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_anymal_flat_pre_physics(C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r),
                                                      self.actions.data_ptr(), n, flags, _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_flat_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        self._anymals.set_joint_position_targets(self.current_targets, self._all_indices)

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        # same four draws, same order as the reference (:199, :217-225)
        u_vel = torch.rand(num_resets, 12, device=dev)
        u_cmd = torch.cat([torch.rand(num_resets, 1, device=dev), torch.rand(num_resets, 1, device=dev),
                           torch.rand(num_resets, 1, device=dev)], dim=-1).contiguous()
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io(u_vel, u_cmd)
        rc = _lib.load().oige_anymal_flat_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets,
                                                    _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_flat_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num_resets)
