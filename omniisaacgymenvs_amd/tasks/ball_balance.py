"""BallBalanceTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/ball_balance.py:44-308, whose arithmetic
runs in liboige_hip.so (csrc/ball_balance.hip).  Same constructor, attributes and methods: ``get_observations /
calculate_metrics / is_done / pre_physics_step / reset_idx / post_reset`` plus the fused ``post_physics_step``.
No torch/CPU fallback.

``reset_idx(env_ids)`` draws with torch in the reference's order (dists, dirs, hspeeds, vspeeds, height: five (n,1) draws,
:220-232); the sync-free ``pre_physics_step`` uses in-kernel Philox keyed by (seed, step, env id).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .base.rl_task import RLTask

BBOT_DOF_NAMES = ["upper_leg0", "lower_leg0", "upper_leg1", "lower_leg1", "upper_leg2", "lower_leg2"]  # stand-in order (USD)
BBOT_FORCE_LINKS = ["upper_leg0", "upper_leg1", "upper_leg2"]


class SyntheticBalanceBotView(SyntheticArticulationView):
    def __init__(self, name: str, count: int, device) -> None:
        limits = torch.stack([-1.2 * torch.ones(6), 1.2 * torch.ones(6)], dim=-1)  # stand-in: the real ones live in the USD
        super().__init__(name=name, count=count, num_dof=6, device=device, num_bodies=3, dof_limits=limits,
                         dof_names={n: i for i, n in enumerate(BBOT_DOF_NAMES)},
                         body_names={n: i for i, n in enumerate(BBOT_FORCE_LINKS)})
        self._dof_indices = self._dof_names


class BallBalanceTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._num_observations = 12 + 12
        self._num_actions = 3
        self.anchored = False
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
        self._step_count = 0
        RLTask.__init__(self, name, env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._num_envs = self._task_cfg["env"]["numEnvs"]
        self._env_spacing = self._task_cfg["env"]["envSpacing"]
        self._dt = self._task_cfg["sim"]["dt"]
        self._table_position = torch.tensor([0, 0, 0.56])
        self._ball_position = torch.tensor([0.0, 0.0, 1.0])
        self._ball_radius = 0.1
        self._action_speed_scale = self._task_cfg["env"]["actionSpeedScale"]
        self._max_episode_length = self._task_cfg["env"]["maxEpisodeLength"]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._balance_bots = SyntheticBalanceBotView("balance_bot_view", self._num_envs, self.device)
        self._balance_bots.positions[:] = self._env_pos + self._table_position.to(self.device)
        self._balls = SyntheticRigidPrimView("ball_view", self._num_envs, self.device)
        self._balls.positions[:] = self._env_pos + self._ball_position.to(self.device)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def post_reset(self):
        n, dev = self.num_envs, self._device
        f = dict(dtype=torch.float32, device=dev)
        dof_limits = self._balance_bots.get_dof_limits()
        self.bbot_dof_lower_limits, self.bbot_dof_upper_limits = torch.t(dof_limits[0].to(device=dev))
        self.initial_dof_positions = self._balance_bots.get_joint_positions()
        self.initial_bot_pos, self.initial_bot_rot = self._balance_bots.get_world_poses()
        self.initial_bot_velocities = self._balance_bots.get_velocities()
        self.initial_ball_pos, self.initial_ball_rot = self._balls.get_world_poses()
        self.initial_ball_velocities = self._balls.get_velocities()
        self.dof_position_targets = torch.zeros((n, self._balance_bots.num_dof), **f)
        actuated_joints = ["lower_leg0", "lower_leg1", "lower_leg2"]
        self.actuated_dof_indices = torch.tensor([self._balance_bots._dof_indices[j] for j in actuated_joints], device=dev,
                                                 dtype=torch.long)
        self._sensor_indices = torch.tensor([self._balance_bots._body_indices[j] for j in BBOT_FORCE_LINKS], device=dev,
                                            dtype=torch.long)

        # device-side scratch of the HIP path
        d = self._balance_bots.num_dof
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "ball_pos": torch.zeros((n, 3), **f), "ball_rot": torch.zeros((n, 4), **f), "ball_vel": torch.zeros((n, 6), **f),
            "bot_pos": torch.zeros((n, 3), **f), "bot_rot": torch.zeros((n, 4), **f), "bot_vel": torch.zeros((n, 6), **f),
            "dof_pos": torch.zeros((n, d), **f),
        }
        self._counts_key = None
        self._host_consts = {"lower": self.bbot_dof_lower_limits.tolist(), "upper": self.bbot_dof_upper_limits.tolist(),
                             "actuated": [int(i) for i in self.actuated_dof_indices]}

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    def _params(self) -> _lib.BallBalanceParams:
        sc = (self._dt, self._action_speed_scale, self._ball_radius, self.clip_actions, self._max_episode_length,
              self._balance_bots.num_dof)
        return self._block("params", (self._host_consts,), sc, self._build_params)

    def _build_params(self) -> _lib.BallBalanceParams:
        p = _lib.BallBalanceParams()
        p.rate = self._dt * self._action_speed_scale  # a Python double product, rounded once (:196-198)
        p.ball_low = self._ball_radius * 1.5           # :305
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        p.max_episode_length = int(self._max_episode_length)
        p.num_dof = self._balance_bots.num_dof
        h = self._host_consts
        for i in range(3):
            p.actuated_dof_indices[i] = h["actuated"][i]
        for i in range(p.num_dof):
            p.dof_lower[i], p.dof_upper[i] = h["lower"][i], h["upper"][i]
        return p

    def _buffers(self) -> _lib.BallBalanceBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.dof_position_targets, self._reset_workspace,
              self._tile_counts_u8)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.BallBalanceBuffers:
        n, rq = self._num_envs, _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 24), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.dof_position_targets, torch.float32, (n, self._balance_bots.num_dof), "dof_position_targets")
        b = _lib.BallBalanceBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.dof_position_targets, b.reset_workspace = self.dof_position_targets.data_ptr(), self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        return b

    def _state(self) -> _lib.BallBalanceState:
        ball_positions, _ = self._balls.get_world_poses(clone=False)
        ball_velocities = self._balls.get_velocities(clone=False)
        dof_pos = self._balance_bots.get_joint_positions(clone=False)
        dof_vel = self._balance_bots.get_joint_velocities(clone=False)
        sensors = self._balance_bots.get_measured_joint_forces(joint_indices=self._sensor_indices)
        live = (ball_positions, ball_velocities, dof_pos, dof_vel, sensors)
        return self._block("state", (*live, self._env_pos), (), lambda: self._build_state(*live))

    def _build_state(self, ball_positions, ball_velocities, dof_pos, dof_vel, sensors) -> _lib.BallBalanceState:
        n, d, rq = self._num_envs, self._balance_bots.num_dof, _lib.require_device_tensor
        rq(ball_positions, torch.float32, (n, 3), "ball_positions")
        rq(ball_velocities, torch.float32, (n, 6), "ball_velocities")
        rq(dof_pos, torch.float32, (n, d), "dof_pos")
        rq(dof_vel, torch.float32, (n, d), "dof_vel")
        rq(sensors, torch.float32, (n, 3, 6), "sensor_force_torques")
        s = _lib.BallBalanceState()
        s.ball_pos, s.ball_velocities = ball_positions.data_ptr(), ball_velocities.data_ptr()
        s.dof_pos, s.dof_vel, s.sensors, s.env_pos = dof_pos.data_ptr(), dof_vel.data_ptr(), sensors.data_ptr(), self._env_pos.data_ptr()
        self._live_state = (ball_positions, ball_velocities, dof_pos, dof_vel, sensors)
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_ball_balance_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                        _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_ball_balance_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        return {"ball_balance": {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        self._post(_lib.STAGE_METRICS)

    def is_done(self) -> None:
        self._post(_lib.STAGE_DONE)

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        self._post(_lib.STAGE_ALL if self._env.world.is_playing() else _lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _reset_io(self, u=None) -> _lib.BallBalanceResetIO:
        ts = (self.initial_ball_pos, self.initial_bot_pos, self.initial_bot_rot, self.initial_bot_velocities,
              self.initial_dof_positions, u, *self._reset_out.values(), self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u) -> _lib.BallBalanceResetIO:
        r = _lib.BallBalanceResetIO()
        r.initial_ball_pos = self.initial_ball_pos.data_ptr()
        r.initial_bot_pos, r.initial_bot_rot = self.initial_bot_pos.data_ptr(), self.initial_bot_rot.data_ptr()
        r.initial_bot_velocities, r.initial_dof_positions = self.initial_bot_velocities.data_ptr(), self.initial_dof_positions.data_ptr()
        if u is not None:
            r.u = u.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.ball_pos_out, r.ball_rot_out, r.ball_vel_out = o["ball_pos"].data_ptr(), o["ball_rot"].data_ptr(), o["ball_vel"].data_ptr()
        r.bot_pos_out, r.bot_rot_out, r.bot_vel_out = o["bot_pos"].data_ptr(), o["bot_rot"].data_ptr(), o["bot_vel"].data_ptr()
        r.dof_pos_out = o["dof_pos"].data_ptr()
        return r

    def _apply_reset_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:
            kw, sl = dict(indices=o["env_ids"], count=o["count"]), slice(None)
        else:
            kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
        self._balls.set_world_poses(o["ball_pos"][sl], o["ball_rot"][sl], **kw)
        self._balls.set_velocities(o["ball_vel"][sl], **kw)
        self._balance_bots.set_world_poses(o["bot_pos"][sl], o["bot_rot"][sl], **kw)
        self._balance_bots.set_velocities(o["bot_vel"][sl], **kw)
        self._balance_bots.set_joint_positions(o["dof_pos"][sl], **kw)

    def _checked_actions(self, actions):
        actions = actions.to(self.device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, 3), "actions")
        return actions

    def pre_physics_step(self, actions) -> None:
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        self._advance_step_counter()
        flags, ids = 0, None
        if not self.sync_free:
            o = self._reset_out
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            num = int(o["count"].item())
            if num > 0:
                ids = o["env_ids"][:num].clone()
                self.reset_idx(ids)
        elif self._counts_valid(self.reset_buf):
            flags |= _lib.PRE_COUNTS_VALID
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_ball_balance_pre_physics(C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), n, flags,
                                                       _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_ball_balance_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        elif ids is not None:
            self.dof_position_targets[ids] = 0  # :204 (the flags were already cleared by reset_idx)
        self._balance_bots.set_joint_position_targets(self.dof_position_targets)

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        u = torch.cat([torch.rand(num_resets, 1, device=dev) for _ in range(5)], dim=-1).contiguous()  # :220-232 order
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io(u)
        rc = _lib.load().oige_ball_balance_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets,
                                                     _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_ball_balance_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num_resets)
