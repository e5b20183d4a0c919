"""CrazyflieTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/crazyflie.py:43-395, whose arithmetic runs
in liboige_hip.so (csrc/crazyflie.hip).  Same constructor, attributes and methods: ``get_observations /
calculate_metrics / is_done / pre_physics_step / set_targets / reset_idx / post_reset`` plus the fused
``post_physics_step``.  No torch/CPU fallback.

The step's thrust noise is ``0.01 * torch.randn(4)`` drawn here with torch, exactly as the reference does (:197), and handed
to the kernel as a device pointer (no sync).  ``reset_idx`` burns the reference's four zero-range ``torch_rand_float``
draws (:316-323) so that torch's generator stays in step with a reference run.
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .base.rl_task import RLTask

EPS = 1e-6
EPISODE_KEYS = ["rew_pos", "rew_orient", "rew_effort", "rew_spin", "raw_dist", "raw_orient", "raw_effort", "raw_spin"]


class SyntheticCrazyflieView(SyntheticArticulationView):
    """CrazyflieView surface used by the task: 4 DOFs, ``physics_rotors[0..3].apply_forces`` (views/crazyflie_view.py)."""

    def __init__(self, name: str, count: int, device) -> None:
        super().__init__(name=name, count=count, num_dof=4, device=device)
        self.physics_rotors = []
        for i in range(4):
            r = SyntheticRigidPrimView(f"m{i + 1}_prop_view", count, device)
            r.applied_forces = None
            r.apply_forces = (lambda rr: (lambda forces, indices=None: setattr(rr, "applied_forces", forces)))(r)
            self.physics_rotors.append(r)


class CrazyflieTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._num_observations = 18
        self._num_actions = 4
        self._crazyflie_position = torch.tensor([0, 0, 1.0])
        self._ball_position = torch.tensor([0, 0, 1.0])
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
        RLTask.__init__(self, name=name, env=env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._num_envs = self._task_cfg["env"]["numEnvs"]
        self._env_spacing = self._task_cfg["env"]["envSpacing"]
        self._max_episode_length = self._task_cfg["env"]["maxEpisodeLength"]
        self.dt = self._task_cfg["sim"]["dt"]
        self.arm_length = 0.05
        self.motor_damp_time_up = 0.15
        self.motor_damp_time_down = 0.15
        self.motor_tau_up = 4 * self.dt / (self.motor_damp_time_up + EPS)
        self.motor_tau_down = 4 * self.dt / (self.motor_damp_time_down + EPS)
        self.mass = 0.028
        self.thrust_to_weight = 1.9
        self.motor_assymetry = np.array([1.0, 1.0, 1.0, 1.0])
        self.motor_assymetry = self.motor_assymetry * 4.0 / np.sum(self.motor_assymetry)
        self.grav_z = -1.0 * self._task_cfg["sim"]["gravity"][2]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._copters = SyntheticCrazyflieView("crazyflie_view", self._num_envs, self.device)
        self._copters.positions[:] = self._env_pos + self._crazyflie_position.to(self.device)
        self._balls = SyntheticRigidPrimView("ball_view", self._num_envs, self.device)
        self._balls.positions[:] = self._env_pos + self._ball_position.to(self.device)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def post_reset(self):
        n, dev = self._num_envs, self._device
        f = dict(dtype=torch.float32, device=dev)
        thrust_max = self.grav_z * self.mass * self.thrust_to_weight * self.motor_assymetry / 4.0
        self.thrusts = torch.zeros((n, 4, 3), **f)
        self.thrust_cmds_damp = torch.zeros((n, 4), **f)
        self.thrust_rot_damp = torch.zeros((n, 4), **f)
        self.thrust_max = torch.tensor(thrust_max, device=dev, dtype=torch.float32)
        self._thrust_max_host = [float(x) for x in self.thrust_max.tolist()]
        self.motor_linearity = 1.0
        self.prop_max_rot = 433.3
        self.target_positions = torch.zeros((n, 3), **f)
        self.target_positions[:, 2] = 1
        self.actions = torch.zeros((n, 4), **f)
        self.all_indices = torch.arange(n, dtype=torch.int32, device=dev)
        self.extras = {}
        self._episode_sums = torch.zeros((len(EPISODE_KEYS), n), **f)   # one block; the dict holds its rows
        self.episode_sums = {k: self._episode_sums[i] for i, k in enumerate(EPISODE_KEYS)}
        self.root_pos, self.root_rot = self._copters.get_world_poses()
        self.root_velocities = self._copters.get_velocities()
        self.dof_pos = self._copters.get_joint_positions()
        self.dof_vel = self._copters.get_joint_velocities()
        self.initial_ball_pos, self.initial_ball_rot = self._balls.get_world_poses(clone=False)
        self.initial_root_pos, self.initial_root_rot = self.root_pos.clone(), self.root_rot.clone()

        # device-side scratch of the HIP path
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._target_workspace = _lib.new_workspace(n, dev)
        self._reduce_workspace = torch.zeros(8 * max((n + 255) // 256, 1), **f)
        self._extras_buf = torch.zeros(8, **f)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "dof_pos": torch.zeros((n, 4), **f), "dof_vel": torch.zeros((n, 4), **f),
            "root_pos": torch.zeros((n, 3), **f), "root_rot": torch.zeros((n, 4), **f), "root_vel": torch.zeros((n, 6), **f),
            "target_ids": torch.zeros(n, **i64), "target_count": torch.zeros(1, **i64),
            "ball_pos": torch.zeros((n, 3), **f), "ball_rot": torch.zeros((n, 4), **f),
        }
        self._noise4 = torch.zeros(4, **f)
        self.set_targets(self.all_indices)

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    def _params(self) -> _lib.CrazyflieParams:
        sc = (self.motor_tau_up, self.motor_tau_down, self.prop_max_rot, self.clip_actions, self._max_episode_length)
        return self._block("params", (self._thrust_max_host,), sc, self._build_params)

    def _build_params(self) -> _lib.CrazyflieParams:
        p = _lib.CrazyflieParams()
        p.motor_tau_up, p.motor_tau_down, p.prop_max_rot = self.motor_tau_up, self.motor_tau_down, self.prop_max_rot
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        for i in range(4):
            p.thrust_max[i] = self._thrust_max_host[i]
        p.max_episode_length = int(self._max_episode_length)
        return p

    def _buffers(self) -> _lib.CrazyflieBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.target_positions, self.thrusts, self.actions,
              self.thrust_cmds_damp, self.thrust_rot_damp, self.dof_pos, self.dof_vel, self._episode_sums, self._extras_buf,
              self._reset_workspace, self._target_workspace, self._reduce_workspace)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.CrazyflieBuffers:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 18), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.target_positions, torch.float32, (n, 3), "target_positions")
        rq(self.thrusts, torch.float32, (n, 4, 3), "thrusts")
        for name in ("actions", "thrust_cmds_damp", "thrust_rot_damp", "dof_pos", "dof_vel"):
            rq(getattr(self, name), torch.float32, (n, 4), name)
        b = _lib.CrazyflieBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.target_positions, b.actions, b.thrusts = self.target_positions.data_ptr(), self.actions.data_ptr(), self.thrusts.data_ptr()
        b.thrust_cmds_damp, b.thrust_rot_damp = self.thrust_cmds_damp.data_ptr(), self.thrust_rot_damp.data_ptr()
        b.dof_pos, b.dof_vel = self.dof_pos.data_ptr(), self.dof_vel.data_ptr()
        b.episode_sums, b.extras = self._episode_sums.data_ptr(), self._extras_buf.data_ptr()
        b.reset_workspace, b.target_workspace = self._reset_workspace.data_ptr(), self._target_workspace.data_ptr()
        b.reduce_workspace = self._reduce_workspace.data_ptr()
        return b

    def _state(self) -> _lib.CrazyflieState:
        self.root_pos, self.root_rot = self._copters.get_world_poses(clone=False)
        self.root_velocities = self._copters.get_velocities(clone=False)
        return self._block("state", (self.root_pos, self.root_rot, self.root_velocities, self._env_pos), (), self._build_state)

    def _build_state(self) -> _lib.CrazyflieState:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.root_pos, torch.float32, (n, 3), "root_pos")
        rq(self.root_rot, torch.float32, (n, 4), "root_rot")
        rq(self.root_velocities, torch.float32, (n, 6), "root_velocities")
        s = _lib.CrazyflieState()
        s.root_pos, s.root_rot = self.root_pos.data_ptr(), self.root_rot.data_ptr()
        s.root_velocities, s.env_pos = self.root_velocities.data_ptr(), self._env_pos.data_ptr()
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_crazyflie_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                     _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_crazyflie_post_physics")
        if stages & _lib.STAGE_DONE:  # the DONE stage leaves the tile counts of both lists pre_physics_step compacts
            self._mark_counts(self.reset_buf, self.progress_buf)
        elif stages & _lib.STAGE_PROGRESS:  # progress moved under the target list's counts (a raw-pointer write: no version bump)
            self._counts_key = None

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

    def _reset_io(self) -> _lib.CrazyflieResetIO:
        ts = (self.initial_root_pos, self.initial_root_rot, self.initial_ball_rot, self._env_pos, self._noise4,
              *self._reset_out.values())
        return self._block("reset_io", ts, (), self._build_reset_io)

    def _build_reset_io(self) -> _lib.CrazyflieResetIO:
        r = _lib.CrazyflieResetIO()
        r.initial_root_pos, r.initial_root_rot = self.initial_root_pos.data_ptr(), self.initial_root_rot.data_ptr()
        r.initial_ball_rot, r.env_pos = self.initial_ball_rot.data_ptr(), self._env_pos.data_ptr()
        r.thrust_noise4 = self._noise4.data_ptr()
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        r.root_pos_out, r.root_rot_out, r.root_vel_out = o["root_pos"].data_ptr(), o["root_rot"].data_ptr(), o["root_vel"].data_ptr()
        r.target_ids, r.target_count = o["target_ids"].data_ptr(), o["target_count"].data_ptr()
        r.ball_pos_out, r.ball_rot_out = o["ball_pos"].data_ptr(), o["ball_rot"].data_ptr()
        return r

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

    def _apply_target_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:
            self._balls.set_world_poses(o["ball_pos"], o["ball_rot"], indices=o["target_ids"], count=o["target_count"])
        else:
            self._balls.set_world_poses(o["ball_pos"][:n], o["ball_rot"][:n], indices=o["target_ids"][:n])

    def _bind_extras(self) -> None:
        # 0-dim views of the device-side means: made once per buffer, handed out in a new dict every step
        self.extras["episode"] = dict(self._block("extras_views", (self._extras_buf,), (),
                                                  lambda: {k: self._extras_buf[i] for i, k in enumerate(EPISODE_KEYS)}))

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, 4), "actions")
        return actions

    def pre_physics_step(self, actions) -> None:
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        flags, ids = 0, None
        if not self.sync_free:
            o = self._reset_out
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            num = int(o["count"].item())
            if num > 0:
                ids = o["env_ids"][:num].clone()
                self.reset_idx(ids)
            set_target_ids = (self.progress_buf % 500 == 0).nonzero(as_tuple=False).squeeze(-1)
            if len(set_target_ids) > 0:
                self.set_targets(set_target_ids)
        elif self._counts_valid(self.reset_buf, self.progress_buf):
            flags |= _lib.PRE_COUNTS_VALID
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS
        self._noise4.normal_(0.0, 0.01)  # 0.01 * torch.randn(4) (:197): the same four draws, scaled inside the one launch
        self.root_pos, self.root_rot = self._copters.get_world_poses(clone=False)
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_crazyflie_pre_physics(C.byref(p), actions.data_ptr(), self.root_rot.data_ptr(), C.byref(b), C.byref(r),
                                                    n, flags, _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_crazyflie_pre_physics")
        self._counts_key = None  # the kernel cleared the flags of the envs it reset (raw-pointer writes: no version bump)
        if self.sync_free:
            self._bind_extras()  # holds the last reset's means; untouched on steps without a reset, as in the reference
            self._apply_reset_writes(None)
            self._apply_target_writes(None)
        elif ids is not None:
            self.thrusts[ids] = 0  # :244 (the flags were already cleared by reset_idx)
        self._copters.set_joint_velocities(self.dof_vel)
        for i in range(4):
            self._copters.physics_rotors[i].apply_forces(self.thrusts[:, i], indices=self.all_indices)

    def set_targets(self, env_ids):
        num_sets = len(env_ids)
        if num_sets == 0:
            return
        env_ids = env_ids.to(device=self._device, dtype=torch.int64).contiguous()
        o = self._reset_out
        o["target_ids"][:num_sets] = env_ids
        o["target_count"].fill_(num_sets)
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_crazyflie_set_targets(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_sets,
                                                    _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_crazyflie_set_targets")
        self._apply_target_writes(num_sets)

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        # the reference's four zero-range torch_rand_float draws (:316-323): values are 0, the generator still advances
        torch.rand(num_resets, 4, device=dev), torch.rand(num_resets, 1, device=dev), torch.rand(num_resets, 1, device=dev)
        torch.rand(num_resets, 1, device=dev)
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_crazyflie_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets,
                                                  _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_crazyflie_reset_idx")
        self._apply_reset_writes(num_resets)
        self.extras["episode"] = {}  # :331-335
        for key in self.episode_sums.keys():
            self.extras["episode"][key] = torch.mean(self.episode_sums[key][env_ids]) / self._max_episode_length
            self.episode_sums[key][env_ids] = 0.0
