"""CartpoleTask -- mirror of /root/reference/omniisaacgymenvs/tasks/cartpole.py:40-164 whose
arithmetic runs in liboige_hip.so (csrc/cartpole.hip).  No torch/CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView
from .base.rl_task import RLTask


class CartpoleTask(RLTask):
    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self._max_episode_length = 500
        self._num_observations = 4
        self._num_actions = 1
        self.fused = True
        self.sync_free = True
        self._step_count = 0
        RLTask.__init__(self, name, env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        self._num_envs = self._task_cfg["env"]["numEnvs"]
        self._env_spacing = self._task_cfg["env"]["envSpacing"]
        self._cartpole_positions = torch.tensor([0.0, 0.0, 2.0])
        self._reset_dist = self._task_cfg["env"]["resetDist"]
        self._max_push_effort = self._task_cfg["env"]["maxEffort"]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        self._cartpoles = SyntheticArticulationView(
            name="cartpole_view", count=self._num_envs, num_dof=2, device=self._device,
            dof_names={"cartJoint": 0, "poleJoint": 1},
        )
        self._cartpoles.positions[:] = self._env_pos + self._cartpole_positions.to(self._device)
        if scene is not None:
            scene.add(self._cartpoles)

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block)
    def _params(self) -> _lib.CartpoleParams:
        sc = (self._cart_dof_idx, self._pole_dof_idx, self._cartpoles.num_dof, self._reset_dist, self._max_push_effort,
              self._max_episode_length)
        return self._block("params", (), sc, self._build_params)

    def _build_params(self) -> _lib.CartpoleParams:
        p = _lib.CartpoleParams()
        p.cart_dof_idx, p.pole_dof_idx, p.num_dof = self._cart_dof_idx, self._pole_dof_idx, self._cartpoles.num_dof
        p.reset_dist, p.max_push_effort = self._reset_dist, self._max_push_effort
        p.max_episode_length = float(self._max_episode_length)
        return p

    def _buffers(self) -> _lib.CartpoleBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self._reset_workspace)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.CartpoleBuffers:
        n = self._num_envs
        rq = _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 4), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        b = _lib.CartpoleBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.reset_workspace = self._reset_workspace.data_ptr()
        return b

    def _checked_state(self, dof_pos, dof_vel):
        n, d = self._num_envs, self._cartpoles.num_dof
        _lib.require_device_tensor(dof_pos, torch.float32, (n, d), "dof_pos")
        _lib.require_device_tensor(dof_vel, torch.float32, (n, d), "dof_vel")
        return dof_pos.data_ptr(), dof_vel.data_ptr()

    def _post(self, stages: int) -> None:
        n = self._num_envs
        dof_pos = self._cartpoles.get_joint_positions(clone=False)
        dof_vel = self._cartpoles.get_joint_velocities(clone=False)
        pos_ptr, vel_ptr = self._block("state", (dof_pos, dof_vel), (), lambda: self._checked_state(dof_pos, dof_vel))
        p, b = self._params(), self._buffers()
        rc = _lib.load().oige_cartpole_post_physics(
            C.byref(p), pos_ptr, vel_ptr, C.byref(b), n, stages,
            _lib.current_stream(self.obs_buf.device),
        )
        _lib.check(rc, "oige_cartpole_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        # views of the observation columns, as the reference caches them (:95-98)
        self.cart_pos, self.cart_vel = self.obs_buf[:, 0], self.obs_buf[:, 1]
        self.pole_pos, self.pole_vel = self.obs_buf[:, 2], self.obs_buf[:, 3]
        return {self._cartpoles.name: {"obs_buf": self.obs_buf}}

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

    def _reset_io(self, u=None) -> _lib.CartpoleResetIO:
        o = self._reset_out
        ts = (u, o["env_ids"], o["count"], o["dof_pos"], o["dof_vel"], self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u) -> _lib.CartpoleResetIO:
        r = _lib.CartpoleResetIO()
        if u is not None:
            r.u = u.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        return r

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, 1), "actions")
        return actions

    def pre_physics_step(self, actions) -> None:
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        self._advance_step_counter()
        flags = 0
        o = self._reset_out
        if not self.sync_free:
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            num = int(o["count"].item())
            if num > 0:
                self.reset_idx(o["env_ids"][:num].clone())
        elif self._counts_valid(self.reset_buf):
            flags |= _lib.PRE_COUNTS_VALID
        p, b, r = self._params(), self._buffers(), self._reset_io()
        rc = _lib.load().oige_cartpole_pre_physics(
            C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), self._forces.data_ptr(), n, flags,
            _lib.current_stream(self.obs_buf.device),
        )
        _lib.check(rc, "oige_cartpole_pre_physics")
        self._counts_key = None
        if self.sync_free:
            kw = dict(indices=o["env_ids"], count=o["count"])
            self._cartpoles.set_joint_positions(o["dof_pos"], **kw)
            self._cartpoles.set_joint_velocities(o["dof_vel"], **kw)
        self._cartpoles.set_joint_efforts(self._forces, indices=self._all_indices)

    def reset_idx(self, env_ids):
        num_resets = len(env_ids)
        if num_resets == 0:
            return
        env_ids = env_ids.to(device=self._device, dtype=torch.int64).contiguous()
        # four torch.rand(num_resets) draws in the reference's order (:129,:130,:134,:135)
        u = torch.stack([torch.rand(num_resets, device=self._device) for _ in range(4)], dim=0).contiguous()
        o = self._reset_out
        o["env_ids"][:num_resets] = env_ids
        o["count"].fill_(num_resets)
        p, b, r = self._params(), self._buffers(), self._reset_io(u)
        rc = _lib.load().oige_cartpole_reset_idx(
            C.byref(p), C.byref(b), C.byref(r), self._num_envs, num_resets, _lib.current_stream(self.obs_buf.device)
        )
        _lib.check(rc, "oige_cartpole_reset_idx")
        self._counts_key = None
        # apply resets
        indices = o["env_ids"][:num_resets].to(dtype=torch.int32)
        self._cartpoles.set_joint_positions(o["dof_pos"][:num_resets], indices=indices)
        self._cartpoles.set_joint_velocities(o["dof_vel"][:num_resets], indices=indices)

    def post_reset(self):
        self._cart_dof_idx = self._cartpoles.get_dof_index("cartJoint")
        self._pole_dof_idx = self._cartpoles.get_dof_index("poleJoint")
        dev, n, d = self._device, self._num_envs, self._cartpoles.num_dof
        f = dict(dtype=torch.float32, device=dev)
        self._forces = torch.zeros((n, d), **f)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, dtype=torch.int64, device=dev),
            "count": torch.zeros(1, dtype=torch.int64, device=dev),
            "dof_pos": torch.zeros((n, d), **f),
            "dof_vel": torch.zeros((n, d), **f),
        }
        self._all_indices = torch.arange(self._cartpoles.count, dtype=torch.int32, device=dev)
        self._counts_key = None
        # randomize all envs
        indices = torch.arange(self._cartpoles.count, dtype=torch.int64, device=dev)
        self.reset_idx(indices)
