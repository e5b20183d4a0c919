"""AnymalTerrainTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/anymal_terrain.py:51-604
whose arithmetic runs in liboige_hip.so (csrc/anymal_terrain.hip).  Same constructor, attribute and
method names: ``pre_physics_step`` (PD torques per decimation sub-step, :391-407), ``post_physics_step``
(:409-444: progress, frame quantities, check_termination, calculate_metrics, the in-step ``reset_idx``,
``get_observations`` with the 140-point height scan, observation noise, last_* copies -- ONE C-ABI call),
``reset_idx(env_ids)`` (:318-367), ``push_robots`` (:446-450), ``post_reset`` (:264-316).  No torch/CPU fallback.

Differences a user can see, all forced by fusing (documented in DESIGN.md):
  * ``reset_buf`` is a BOOL tensor after the first step exactly as in the reference (:465), but it is the same
    storage every step instead of a fresh tensor;
  * ``obs_buf`` is written in place instead of being rebound to a new ``torch.cat`` result (:540);
  * the view buffers are read-only for the kernel: the reset rows (:334-339) reach the view through its setters
    as (rows, env_ids, count) with the count on the device -- no host sync in ``post_physics_step``;
  * the in-step reset and the observation noise draw from in-kernel Philox keyed by (seed, step, env id);
    ``reset_idx(env_ids)`` called directly draws with torch in the reference's order (:321,:322,:330,:341-349).
"""
from __future__ import annotations

import ctypes as C

import numpy as np
import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticContactView
from ..sim.terrain import SyntheticTerrain
from .base.rl_task import RLTask

# Anymal.dof_names: taken to be the YAML's listing order (cfg/task/AnymalTerrain.yaml:48-62); the real order comes
# from the USD articulation (anymal_terrain.py:258-262)
ANYMAL_DOF_NAMES = ["LF_HAA", "LH_HAA", "RF_HAA", "RH_HAA", "LF_HFE", "LH_HFE", "RF_HFE", "RH_HFE",
                    "LF_KFE", "LH_KFE", "RF_KFE", "RH_KFE"]
EPISODE_KEYS = ["lin_vel_xy", "lin_vel_z", "ang_vel_z", "ang_vel_xy", "orient", "torques", "joint_acc", "base_height",
                "air_time", "collision", "stumble", "action_rate", "hip"]  # dict order of :68-82 = episode_sums rows


class SyntheticAnymalView(SyntheticArticulationView):
    """AnymalView surface the task uses (views/anymal_view.py:46-59): ``_knees`` and ``_base`` contact views."""

    def __init__(self, name: str, count: int, device) -> None:
        super().__init__(name=name, count=count, num_dof=12, device=device,
                         dof_names={n: i for i, n in enumerate(ANYMAL_DOF_NAMES)})
        self._knees = SyntheticContactView("knees_view", count * 4, device)
        self._base = SyntheticContactView("base_view", count, device)
        self.dof_names = list(ANYMAL_DOF_NAMES)

    def is_base_below_threshold(self, threshold, ground_heights):  # views/anymal_view.py:76-80
        base_pos, _ = self.get_world_poses()
        base_heights = base_pos[:, 2]
        base_heights -= ground_heights
        return base_heights[:] < threshold


class AnymalTerrainTask(RLTask):
    # push_robots (:419-420, :446-450) runs inside oige_anymal_post_physics: the step counter test and the U(-1,1) draws
    # (Philox stream 7) are on the device, ``set_velocities`` gets (values, indices, count) with a device count that is N on
    # a push step and 0 otherwise -- no step differs from another on the host, so the step can be captured into a HIP
    # graph.  ``device_push = False`` restores the reference's host branch with torch's draw (and refuses capture).
    device_push = True
    _common_offset = 0
    sync_free = True   # the in-step reset has always been device-side (count on the device)
    fused = True

    @property
    def common_step_counter(self) -> int:
        """:273, :418: post_physics_step calls so far.  Kept as ``_step_count + offset`` so that replayed graph steps
        (``replay_step`` advances ``_step_count``) count too and the kernel's ``push_offset`` stays a constant."""
        return self._step_count + self._common_offset

    @common_step_counter.setter
    def common_step_counter(self, value: int) -> None:
        self._common_offset = int(value) - self._step_count

    @property
    def graph_capturable(self):
        return True if self.device_push else ("device_push is off: push_robots is a host-side branch that differs from "
                                               "step to step, a captured graph would freeze one side of it")

    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.height_samples = None
        self.custom_origins = False
        self.init_done = False
        self._env_spacing = 0.0
        self._num_observations = 188
        self._num_actions = 12
        self.update_config(sim_config)
        RLTask.__init__(self, name, env)
        self.height_points = self.init_height_points()
        self.measured_heights = None
        n, dev = self.num_envs, self.device
        self.default_dof_pos = torch.zeros((n, 12), dtype=torch.float, device=dev)
        # the 13 episode sums live in one (13, N) block; the dict holds its rows (:66-82)
        self._episode_sums = torch.zeros((len(EPISODE_KEYS), n), dtype=torch.float, device=dev)
        self.episode_sums = {k: self._episode_sums[i] for i, k in enumerate(EPISODE_KEYS)}
        self._step_count = 0

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        env, learn = self._task_cfg["env"], self._task_cfg["env"]["learn"]
        self.lin_vel_scale = learn["linearVelocityScale"]
        self.ang_vel_scale = learn["angularVelocityScale"]
        self.dof_pos_scale = learn["dofPositionScale"]
        self.dof_vel_scale = learn["dofVelocityScale"]
        self.height_meas_scale = learn["heightMeasurementScale"]
        self.action_scale = env["control"]["actionScale"]
        self.rew_scales = {
            "termination": learn["terminalReward"], "lin_vel_xy": learn["linearVelocityXYRewardScale"],
            "lin_vel_z": learn["linearVelocityZRewardScale"], "ang_vel_z": learn["angularVelocityZRewardScale"],
            "ang_vel_xy": learn["angularVelocityXYRewardScale"], "orient": learn["orientationRewardScale"],
            "torque": learn["torqueRewardScale"], "joint_acc": learn["jointAccRewardScale"],
            "base_height": learn["baseHeightRewardScale"], "action_rate": learn["actionRateRewardScale"],
            "hip": learn["hipRewardScale"], "fallen_over": learn["fallenOverRewardScale"],
        }
        rng = env["randomCommandVelocityRanges"]
        self.command_x_range, self.command_y_range, self.command_yaw_range = rng["linear_x"], rng["linear_y"], rng["yaw"]
        init = env["baseInitState"]
        self.base_init_state = init["pos"] + init["rot"] + init["vLinear"] + init["vAngular"]
        self.named_default_joint_angles = env["defaultJointAngles"]
        self.decimation = env["control"]["decimation"]
        self.dt = self.decimation * self._task_cfg["sim"]["dt"]
        self.max_episode_length_s = learn["episodeLength_s"]
        self.max_episode_length = int(self.max_episode_length_s / self.dt + 0.5)
        self.push_interval = int(learn["pushInterval_s"] / self.dt + 0.5)
        self.Kp = env["control"]["stiffness"]
        self.Kd = env["control"]["damping"]
        self.curriculum = env["terrain"]["curriculum"]
        self.base_threshold = 0.2
        self.knee_threshold = 0.1
        for key in self.rew_scales.keys():
            self.rew_scales[key] *= self.dt
        self._num_envs = env["numEnvs"]
        mat = self._task_cfg["sim"].setdefault("default_physics_material", {})
        mat["static_friction"] = env["terrain"]["staticFriction"]
        mat["dynamic_friction"] = env["terrain"]["dynamicFriction"]
        mat["restitution"] = env["terrain"]["restitution"]
        self._task_cfg["sim"]["add_ground_plane"] = False

    def _get_noise_scale_vec(self, cfg):
        learn = self._task_cfg["env"]["learn"]
        noise_vec = torch.zeros_like(self.obs_buf[0])
        self.add_noise = learn["addNoise"]
        noise_level = learn["noiseLevel"]
        self._noise_scales = [
            learn["linearVelocityNoise"] * noise_level * self.lin_vel_scale,
            learn["angularVelocityNoise"] * noise_level * self.ang_vel_scale,
            learn["gravityNoise"] * noise_level,
            learn["dofPositionNoise"] * noise_level * self.dof_pos_scale,
            learn["dofVelocityNoise"] * noise_level * self.dof_vel_scale,
            learn["heightMeasurementNoise"] * noise_level * self.height_meas_scale,
        ]
        noise_vec[:3], noise_vec[3:6], noise_vec[6:9] = self._noise_scales[0:3]
        noise_vec[12:24], noise_vec[24:36], noise_vec[36:176] = self._noise_scales[3:6]
        return noise_vec

    def init_height_points(self):
        y = 0.1 * torch.tensor([-5, -4, -3, -2, -1, 1, 2, 3, 4, 5], device=self.device)
        x = 0.1 * torch.tensor([-8, -7, -6, -5, -4, -3, -2, 2, 3, 4, 5, 6, 7, 8], device=self.device)
        grid_x, grid_y = torch.meshgrid(x, y, indexing="ij")
        self.num_height_points = grid_x.numel()
        # the reference materialises (N,140,3); the kernel rebuilds the grid from its indices, so one row suffices
        points = torch.zeros(1, self.num_height_points, 3, device=self.device)
        points[:, :, 0] = grid_x.flatten()
        points[:, :, 1] = grid_y.flatten()
        return points

    # ------------------------------------------------------------------ scene
    def _create_trimesh(self, create_mesh=True):
        self.terrain = SyntheticTerrain(self._task_cfg["env"]["terrain"], num_robots=self.num_envs)
        self.height_samples = torch.tensor(self.terrain.heightsamples).view(
            self.terrain.tot_rows, self.terrain.tot_cols).to(self.device).contiguous()

    def get_terrain(self, create_mesh=True):
        tcfg = self._task_cfg["env"]["terrain"]
        self.env_origins = torch.zeros((self.num_envs, 3), device=self.device)
        if not self.curriculum:
            tcfg["maxInitMapLevel"] = tcfg["numLevels"] - 1
        self.terrain_levels = torch.randint(0, tcfg["maxInitMapLevel"] + 1, (self.num_envs,), device=self.device)
        self.terrain_types = torch.randint(0, tcfg["numTerrains"], (self.num_envs,), device=self.device)
        self._create_trimesh(create_mesh=create_mesh)
        self.terrain_origins = torch.from_numpy(self.terrain.env_origins).to(self.device).to(torch.float).contiguous()

    def get_anymal(self):
        self.dof_names = list(ANYMAL_DOF_NAMES)
        for i in range(self.num_actions):
            self.default_dof_pos[:, i] = self.named_default_joint_angles[self.dof_names[i]]

    def set_up_scene(self, scene=None) -> None:
        self.get_terrain()
        self.get_anymal()
        RLTask.set_up_scene(self, scene)
        self._anymals = SyntheticAnymalView("anymal_view", self._num_envs, self.device)

    def initialize_views(self, scene=None):
        self.get_terrain(create_mesh=False)
        RLTask.initialize_views(self, scene)
        self._anymals = SyntheticAnymalView("anymal_view", self._num_envs, self.device)

    def post_reset(self):
        n, dev = self.num_envs, self.device
        f = dict(dtype=torch.float, device=dev)
        self._base_init_host = [float(v) for v in self.base_init_state]
        self.base_init_state = torch.tensor(self.base_init_state, **f)
        self.timeout_buf = torch.zeros(n, device=dev, dtype=torch.long)
        self.up_axis_idx = 2
        self.common_step_counter = 0
        self.extras = {}
        self.noise_scale_vec = self._get_noise_scale_vec(self._task_cfg)
        self.commands = torch.zeros(n, 4, **f)  # x vel, y vel, yaw vel, heading
        self.commands_scale = torch.tensor([self.lin_vel_scale, self.lin_vel_scale, self.ang_vel_scale], device=dev)
        self.gravity_vec = torch.tensor([0.0, 0.0, -1.0], **f).repeat((n, 1))
        self.forward_vec = torch.tensor([1.0, 0.0, 0.0], **f).repeat((n, 1))
        self.torques = torch.zeros(n, self.num_actions, **f)
        self.actions = torch.zeros(n, self.num_actions, **f)
        self.last_actions = torch.zeros(n, self.num_actions, **f)
        self.feet_air_time = torch.zeros(n, 4, **f)
        self.last_dof_vel = torch.zeros((n, 12), **f)
        self.env_origins[:] = self.terrain_origins[self.terrain_levels, self.terrain_types]  # :302-303 as one gather
        self.num_dof = self._anymals.num_dof
        self.refresh_dof_state_tensors()
        self.refresh_body_state_tensors()

        # device-side scratch of the HIP path
        tiles = max((n + 63) // 64, 1)
        self.has_fallen = torch.zeros(n, dtype=torch.bool, device=dev)
        self._reset_bool = torch.zeros(n, dtype=torch.bool, device=dev)
        self._reset_mask = torch.zeros(n, dtype=torch.uint8, device=dev)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        self._reduce_workspace = torch.zeros((16 + 512) * tiles + 4, **f)  # partial sums + pose records (oige_hip.h)
        self._extras_dev = torch.zeros(len(EPISODE_KEYS) + 1, **f)
        i64 = dict(dtype=torch.int64, device=dev)
        self._push_count = torch.zeros(1, **i64)
        self._all_indices = torch.arange(n, **i64)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "base_pos": torch.zeros((n, 3), **f), "base_quat": torch.zeros((n, 4), **f), "base_vel": torch.zeros((n, 6), **f),
            "dof_pos": torch.zeros((n, 12), **f), "dof_vel": torch.zeros((n, 12), **f),
        }
        self._default_dof_host = [float(v) for v in self.default_dof_pos[0].tolist()] if n > 0 else [0.0] * 12

        indices = torch.arange(self._num_envs, dtype=torch.int64, device=self._device)
        self.reset_idx(indices)
        self.init_done = True

    # ------------------------------------------------------------------ C-ABI argument blocks
    _PARAM_SCALARS = ("lin_vel_scale", "ang_vel_scale", "dof_pos_scale", "dof_vel_scale", "height_meas_scale", "action_scale", "Kp", "Kd",
                      "max_episode_length_s", "max_episode_length", "curriculum", "init_done", "add_noise")

    def _params(self) -> _lib.AnymalParams:
        """Cached (RLTask._block): rebuilt when a public scalar, the reward scales, the command ranges, the noise scales, the
        terrain or the height map changed."""
        d = self.__dict__
        scal = [d.get(k) for k in self._PARAM_SCALARS]
        scal += [tuple(self.rew_scales.values()), tuple(self.command_x_range), tuple(self.command_y_range), tuple(self.command_yaw_range),
                 tuple(d.get("_noise_scales", ())), tuple(d.get("_base_init_host", ())), tuple(self._default_dof_host)]
        return self._block("params", (self.terrain, self.height_samples), scal, self._build_params)

    def _build_params(self) -> _lib.AnymalParams:
        p = _lib.AnymalParams()
        p.lin_vel_scale, p.ang_vel_scale = self.lin_vel_scale, self.ang_vel_scale
        p.dof_pos_scale, p.dof_vel_scale = self.dof_pos_scale, self.dof_vel_scale
        p.height_meas_scale, p.action_scale = self.height_meas_scale, self.action_scale
        p.Kp, p.Kd = self.Kp, self.Kd
        p.max_episode_length_s = self.max_episode_length_s
        p.max_episode_length = self.max_episode_length
        p.curriculum, p.init_done = int(bool(self.curriculum)), int(bool(self.init_done))
        p.add_noise = int(bool(getattr(self, "add_noise", False)))
        t = self.terrain
        p.env_rows, p.env_cols = t.env_rows, t.env_cols
        p.hs_rows, p.hs_cols = self.height_samples.shape
        p.horizontal_scale, p.vertical_scale, p.border_size = t.horizontal_scale, t.vertical_scale, t.border_size
        p.half_env_length = t.env_length / 2
        rs = self.rew_scales
        for i, k in enumerate(("termination", "lin_vel_xy", "lin_vel_z", "ang_vel_z", "ang_vel_xy", "orient", "torque",
                               "joint_acc", "base_height", "action_rate", "hip", "fallen_over")):
            p.rew_scales[i] = rs[k]
        for i, r in enumerate((self.command_x_range, self.command_y_range, self.command_yaw_range)):
            p.cmd_lo[i], p.cmd_span[i] = r[0], r[1] - r[0]
        for i, v in enumerate(getattr(self, "_base_init_host", [0.0] * 13)):
            p.base_init_state[i] = v
        for i, v in enumerate(self._default_dof_host):
            p.default_dof_pos[i] = v
        for i, v in enumerate(getattr(self, "_noise_scales", [0.0] * 6)):
            p.noise_scales[i] = v
        return p

    def _state(self) -> _lib.AnymalState:
        knee = self._anymals._knees.get_net_contact_forces(clone=False)
        base = self._anymals._base.get_net_contact_forces(clone=False)
        hmin = self._height_min_table() if self.use_height_min else None
        live = (self.base_pos, self.base_quat, self.base_velocities, self.dof_pos, self.dof_vel, knee, base, self.height_samples,
                self.terrain_origins, hmin)
        return self._block("state", live, (), lambda: self._build_state(knee, base, hmin))

    def _build_state(self, knee, base, hmin) -> _lib.AnymalState:
        n = self._num_envs
        rq = _lib.require_device_tensor
        rq(self.base_pos, torch.float32, (n, 3), "base_pos")
        rq(self.base_quat, torch.float32, (n, 4), "base_quat")
        rq(self.base_velocities, torch.float32, (n, 6), "base_velocities")
        rq(self.dof_pos, torch.float32, (n, 12), "dof_pos")
        rq(self.dof_vel, torch.float32, (n, 12), "dof_vel")
        rq(knee, torch.float32, (4 * n, 3), "knee contact forces")
        rq(base, torch.float32, (n, 3), "base contact forces")
        rq(self.height_samples, torch.int16, (self.terrain.tot_rows, self.terrain.tot_cols), "height_samples")
        rq(self.terrain_origins, torch.float32, (self.terrain.env_rows, self.terrain.env_cols, 3), "terrain_origins")
        s = _lib.AnymalState()
        s.base_pos, s.base_quat, s.base_velocities = self.base_pos.data_ptr(), self.base_quat.data_ptr(), self.base_velocities.data_ptr()
        s.dof_pos, s.dof_vel = self.dof_pos.data_ptr(), self.dof_vel.data_ptr()
        s.knee_contact, s.base_contact = knee.data_ptr(), base.data_ptr()
        s.height_samples, s.terrain_origins = self.height_samples.data_ptr(), self.terrain_origins.data_ptr()
        if hmin is not None:
            s.height_min = hmin.data_ptr()
        return s

    # get_heights keeps min(h[px, py], h[px + 1, py + 1]) (:598-600): tabulated once per map, one gather per scan point
    use_height_min = True

    def _height_min_table(self) -> torch.Tensor:
        """(hs_rows, hs_cols) int16 from ``oige_anymal_height_min``; rebuilt when ``height_samples`` is rebound or edited."""
        hs = self.height_samples
        key = (hs, hs._version)
        cached = getattr(self, "_height_min_key", None)
        if cached is None or cached[0] is not key[0] or cached[1] != key[1]:
            out = torch.empty_like(hs)
            _lib.check(_lib.load().oige_anymal_height_min(hs.data_ptr(), hs.shape[0], hs.shape[1], out.data_ptr(),
                                                          _lib.current_stream(hs.device)), "oige_anymal_height_min")
            self._height_min, self._height_min_key = out, key
        return self._height_min

    def _buffers(self) -> _lib.AnymalBuffers:
        # `actions` and `torques` are NEW tensors every step, as in the reference (actions.clone() :390, torch.clip(...) :403):
        # as parts of the key they made every step a rebuild (17 us of host time); they are rewritten behind the cache instead
        ts = (self.obs_buf, self.rew_buf, self._reset_bool, self.has_fallen, self.timeout_buf, self.progress_buf, self.commands,
              self.last_actions, self.last_dof_vel, self.feet_air_time, self._episode_sums,
              self.terrain_levels, self.terrain_types, self.env_origins, self._extras_dev, self._reset_workspace,
              self._reduce_workspace, self._tile_counts_u8)
        b = self._block("buffers", ts, (), self._build_buffers, volatile=("actions", "torques"))
        a, t = self.actions, self.torques
        if a.shape != t.shape or a.dtype != torch.float32 or t.dtype != torch.float32 or a.device != self.obs_buf.device \
                or t.device != a.device or tuple(a.shape) != (self._num_envs, 12) or not (a.is_contiguous() and t.is_contiguous()):
            _lib.require_device_tensor(a, torch.float32, (self._num_envs, 12), "actions")
            _lib.require_device_tensor(t, torch.float32, (self._num_envs, 12), "torques")
            raise _lib.OigeError("AnymalTerrainTask: actions / torques must be contiguous (N, 12) fp32 tensors on the task's device")
        b.actions, b.torques = a.data_ptr(), t.data_ptr()
        return b

    def _build_buffers(self) -> _lib.AnymalBuffers:
        n = self._num_envs
        rq = _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 188), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.timeout_buf, torch.int64, (n,), "timeout_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.commands, torch.float32, (n, 4), "commands")
        for name in ("actions", "last_actions", "last_dof_vel", "torques"):
            rq(getattr(self, name), torch.float32, (n, 12), name)
        rq(self.feet_air_time, torch.float32, (n, 4), "feet_air_time")
        rq(self.terrain_levels, torch.int64, (n,), "terrain_levels")
        rq(self.terrain_types, torch.int64, (n,), "terrain_types")
        rq(self.env_origins, torch.float32, (n, 3), "env_origins")
        b = _lib.AnymalBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.has_fallen = self._reset_bool.data_ptr(), self.has_fallen.data_ptr()
        b.timeout_buf, b.progress_buf = self.timeout_buf.data_ptr(), self.progress_buf.data_ptr()
        b.commands, b.actions = self.commands.data_ptr(), self.actions.data_ptr()
        b.last_actions, b.last_dof_vel = self.last_actions.data_ptr(), self.last_dof_vel.data_ptr()
        b.torques, b.feet_air_time = self.torques.data_ptr(), self.feet_air_time.data_ptr()
        b.episode_sums = self._episode_sums.data_ptr()
        b.terrain_levels, b.terrain_types = self.terrain_levels.data_ptr(), self.terrain_types.data_ptr()
        b.env_origins, b.extras = self.env_origins.data_ptr(), self._extras_dev.data_ptr()
        b.reset_workspace, b.reduce_workspace = self._reset_workspace.data_ptr(), self._reduce_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        return b

    def _reset_io(self, draws=None, noise_u=None, mask=None, push=False) -> _lib.AnymalResetIO:
        pushing = bool(push and self.device_push and self.push_interval > 0)
        ts = (*(draws or ()), noise_u, mask, self.base_velocities if pushing else None, self._push_count, *self._reset_out.values(),
              self._step_dev if self._capturing else None)
        scal = (self._rand_seed, pushing, int(self.push_interval), int(self._common_offset))
        r = self._block("reset_io", ts, scal, lambda: self._build_reset_io(draws, noise_u, mask, push))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, draws, noise_u, mask, push) -> _lib.AnymalResetIO:
        r = _lib.AnymalResetIO()
        if draws is not None:
            r.u_dof_offset, r.u_dof_vel, r.u_xy, r.u_cmd = (d.data_ptr() for d in draws)
        if noise_u is not None:
            _lib.require_device_tensor(noise_u, torch.float32, (self._num_envs, 188), "noise_u")
            r.noise_u = noise_u.data_ptr()
        if mask is not None:
            r.reset_mask = mask.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()  # graph capture: the step counter lives on the device
        if push and self.device_push and self.push_interval > 0:
            _lib.require_device_tensor(self.base_velocities, torch.float32, (self._num_envs, 6), "base_velocities")
            r.push_velocities, r.push_count = self.base_velocities.data_ptr(), self._push_count.data_ptr()
            r.push_interval = int(self.push_interval)
            r.push_offset = int(self._common_offset)  # common_step_counter - _step_count
        o = self._reset_out
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.base_pos_out, r.base_quat_out, r.base_vel_out = o["base_pos"].data_ptr(), o["base_quat"].data_ptr(), o["base_vel"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        return r

    def _apply_reset_writes(self, n=None) -> None:
        """The four view setters of :334-339 with the kernel's compacted rows."""
        o = self._reset_out
        if n is None:
            kw, sl = dict(indices=o["env_ids"], count=o["count"]), slice(None)
        else:
            kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
        self._anymals.set_world_poses(positions=o["base_pos"][sl], orientations=o["base_quat"][sl], **kw)
        self._anymals.set_velocities(velocities=o["base_vel"][sl], **kw)
        self._anymals.set_joint_positions(positions=o["dof_pos"][sl], **kw)
        self._anymals.set_joint_velocities(velocities=o["dof_vel"][sl], **kw)

    def _bind_extras(self) -> None:
        # 0-dim views of the device-side means; the views are made once per buffer (fourteen slices were 10 us of every step),
        # the dict is a new one every step, as in the reference (:361-367)
        self.extras["episode"] = dict(self._block("extras_views", (self._extras_dev,), (), self._build_extras_views))

    def _build_extras_views(self) -> dict:
        ep = {"rew_" + k: self._extras_dev[i] for i, k in enumerate(EPISODE_KEYS)}
        ep["terrain_level"] = self._extras_dev[len(EPISODE_KEYS)]
        return ep

    # ------------------------------------------------------------------ RLTask methods
    def reset_idx(self, env_ids, draws=None):
        """:318-367.  ``draws``: optional (u_dof_offset (n,12), u_dof_vel (n,12), u_xy (n,2), u_cmd (n,3)) U[0,1)."""
        n = len(env_ids)
        if n == 0:
            return
        dev = self.device
        env_ids = env_ids.to(device=dev, dtype=torch.int64)
        if draws is None:  # the reference's six torch_rand_float calls, in order
            draws = (torch.rand(n, 12, device=dev), torch.rand(n, 12, device=dev), torch.rand(n, 2, device=dev),
                     torch.cat([torch.rand(n, 1, device=dev), torch.rand(n, 1, device=dev), torch.rand(n, 1, device=dev)], dim=-1))
        draws = tuple(d.to(device=dev, dtype=torch.float32).contiguous() for d in draws)
        self._reset_mask.zero_()
        self._reset_mask[env_ids] = 1
        p, s, b, r = self._params(), self._state(), self._buffers(), self._reset_io(draws=draws, mask=self._reset_mask)
        rc = _lib.load().oige_anymal_reset_idx(C.byref(p), C.byref(s), C.byref(b), C.byref(r), self._num_envs,
                                               _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_reset_idx")
        self.reset_buf[env_ids] = 1  # :358, on whatever dtype reset_buf currently has
        self._apply_reset_writes(n)
        self._bind_extras()

    def update_terrain_level(self, env_ids):
        """:369-380 on its own (inside a step it is part of the reset: oige_anymal_post_physics / oige_anymal_reset_idx)."""
        if not self.init_done or not self.curriculum:  # :370-372
            return
        env_ids = env_ids.to(device=self.device, dtype=torch.int64).contiguous()
        root_pos, _ = self._anymals.get_world_poses(clone=False)
        n = self._num_envs
        rq = _lib.require_device_tensor
        rq(root_pos, torch.float32, (n, 3), "root_pos")
        rq(self.commands, torch.float32, (n, 4), "commands")
        rq(self.terrain_levels, torch.int64, (n,), "terrain_levels")
        rq(self.terrain_types, torch.int64, (n,), "terrain_types")
        rq(self.env_origins, torch.float32, (n, 3), "env_origins")
        p = self._params()
        rc = _lib.load().oige_anymal_update_terrain_level(
            C.byref(p), root_pos.data_ptr(), self.terrain_origins.data_ptr(), self.commands.data_ptr(), self.terrain_levels.data_ptr(),
            self.terrain_types.data_ptr(), self.env_origins.data_ptr(), env_ids.data_ptr(), int(env_ids.numel()), n,
            _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_update_terrain_level")

    def refresh_dof_state_tensors(self):
        self.dof_pos = self._anymals.get_joint_positions(clone=False)
        self.dof_vel = self._anymals.get_joint_velocities(clone=False)

    def refresh_body_state_tensors(self):
        self.base_pos, self.base_quat = self._anymals.get_world_poses(clone=False)
        self.base_velocities = self._anymals.get_velocities(clone=False)
        self.knee_pos, self.knee_quat = self._anymals._knees.get_world_poses(clone=False)

    def pre_physics_step(self, actions):
        if not self.world.is_playing():
            return
        n = self._num_envs
        actions = actions.to(self.device)
        _lib.require_device_tensor(actions, torch.float32, (n, 12), "actions")
        self.actions = torch.empty_like(actions)  # actions.clone() (:390): written by the first sub-step's launch
        cloned = False
        p = self._params()
        lib, stream = _lib.load(), _lib.current_stream(self.obs_buf.device)
        for i in range(self.decimation):
            if self.world.is_playing():
                torques = torch.empty_like(self.torques)
                if not cloned:
                    rc = lib.oige_anymal_torques_clone(C.byref(p), actions.data_ptr(), self.actions.data_ptr(), self.dof_pos.data_ptr(),
                                                       self.dof_vel.data_ptr(), torques.data_ptr(), n, stream)
                    cloned = True
                else:
                    rc = lib.oige_anymal_torques(C.byref(p), self.actions.data_ptr(), self.dof_pos.data_ptr(),
                                                 self.dof_vel.data_ptr(), torques.data_ptr(), n, stream)
                _lib.check(rc, "oige_anymal_torques")
                self._anymals.set_joint_efforts(torques)
                self.torques = torques
                self.world.step(render=False)  # SimulationContext.step(self.world, render=False)
                self.refresh_dof_state_tensors()
        if not cloned:  # decimation 0 or the world stopped inside the loop: no launch took the copy
            self.actions.copy_(actions)

    # the reference keeps the body-frame quantities of :422-424 as attributes; the fused step leaves them in registers unless
    # asked to write them out (12 extra bytes x 3 per env and step)
    materialize_frame = False

    def _frame(self) -> _lib.AnymalFrame:
        if getattr(self, "base_lin_vel", None) is None:
            f = dict(dtype=torch.float, device=self.device)
            self.base_lin_vel = torch.zeros((self._num_envs, 3), **f)
            self.base_ang_vel = torch.zeros((self._num_envs, 3), **f)
            self.projected_gravity = torch.zeros((self._num_envs, 3), **f)
        ts = (self.base_lin_vel, self.base_ang_vel, self.projected_gravity)
        return self._block("frame", ts, (), self._build_frame)

    def _build_frame(self) -> _lib.AnymalFrame:
        x = _lib.AnymalFrame()
        for name in ("base_lin_vel", "base_ang_vel", "projected_gravity"):
            t = getattr(self, name)
            _lib.require_device_tensor(t, torch.float32, (self._num_envs, 3), name)
            setattr(x, name, t.data_ptr())
        return x

    def _stages(self, stages: int, noise_u=None, push: bool = False):
        """One ``oige_anymal_post_physics_stages`` call; returns the reset_io block it used."""
        p, s, b, r = self._params(), self._state(), self._buffers(), self._reset_io(noise_u=noise_u, push=push)
        rc = _lib.load().oige_anymal_post_physics_stages(C.byref(p), C.byref(s), C.byref(b), C.byref(r), C.byref(self._frame()),
                                                         self._num_envs, stages, _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_anymal_post_physics_stages")
        return r

    def post_physics_step(self, noise_u=None):
        if not self.world.is_playing():
            self.progress_buf[:] += 1
            return self.obs_buf, self.rew_buf, self.reset_buf, self.extras
        self.refresh_dof_state_tensors()
        self.refresh_body_state_tensors()
        self._advance_step_counter()  # common_step_counter += 1 (:418): it is _step_count + a constant, see the property
        if not self.device_push and self.common_step_counter % self.push_interval == 0:
            self.push_robots()
        if not self.fused:
            return self._post_physics_step_by_stages(noise_u)
        if self.materialize_frame:
            r = self._stages(_lib.ANYMAL_STAGE_ALL, noise_u=noise_u, push=True)
        else:
            p, s, b, r = self._params(), self._state(), self._buffers(), self._reset_io(noise_u=noise_u, push=True)
            rc = _lib.load().oige_anymal_post_physics(C.byref(p), C.byref(s), C.byref(b), C.byref(r), self._num_envs,
                                                      _lib.current_stream(self.obs_buf.device))
            _lib.check(rc, "oige_anymal_post_physics")
        if r.push_velocities:  # :450 with a device-side count: N on a push step, 0 otherwise
            self._anymals.set_velocities(self.base_velocities, indices=self._all_indices, count=self._push_count)
        self.reset_buf = self._reset_bool  # bool, as the reference rebinds it (:465)
        self._apply_reset_writes(None)
        self._bind_extras()
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _post_physics_step_by_stages(self, noise_u=None):
        """``task.fused = False``: the reference's own sequence (:410-442), one method per stage, host-visible id list, torch
        draws for the reset and the noise -- what a subclass that overrides a stage gets."""
        self._stages(_lib.ANYMAL_STAGE_PROGRESS)                                  # :410
        self.compute_body_frame()                                                 # :421-427
        self.check_termination()                                                  # :429
        self.calculate_metrics()                                                  # :431
        env_ids = self.reset_buf.nonzero(as_tuple=False).flatten()                # :433
        if len(env_ids) > 0:
            self.reset_idx(env_ids)                                               # :435
        self.get_observations()                                                   # :437
        if self.add_noise:                                                        # :438-439
            u = noise_u if noise_u is not None else torch.rand_like(self.obs_buf)
            self.obs_buf += (2 * u - 1) * self.noise_scale_vec
        self.last_actions[:] = self.actions[:]                                    # :441
        self.last_dof_vel[:] = self.dof_vel[:]                                    # :442
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def push_robots(self):
        self.base_velocities[:, 0:2] = 2.0 * torch.rand(self.num_envs, 2, device=self.device) + (-1.0)  # lin vel x/y
        self._anymals.set_velocities(self.base_velocities)

    # ---- the stages of post_physics_step as methods of their own, as in the reference (each ONE C-ABI call on the same
    # kernels as the fused step, which stays bit-identical to their sequence: tests/test_anymal_gpu.py)
    def compute_body_frame(self):
        """:421-427 (inline in the reference's post_physics_step): ``base_lin_vel``, ``base_ang_vel``, ``projected_gravity``
        and the heading command ``commands[:, 2]``."""
        self._stages(_lib.ANYMAL_STAGE_FRAME)

    def check_termination(self):
        """:452-466: ``timeout_buf``, ``has_fallen``, ``reset_buf`` (rebound to the bool tensor, as :465 does)."""
        self._stages(_lib.ANYMAL_STAGE_TERMINATION)
        self.reset_buf = self._reset_bool

    def calculate_metrics(self):
        """:468-533: ``rew_buf`` and the ten episode sums, from the attributes ``compute_body_frame`` left."""
        self._stages(_lib.ANYMAL_STAGE_METRICS)

    def get_observations(self):
        """:535-552: ``obs_buf`` (no noise: that is :438-439) with the height scan of the views' current state."""
        self.refresh_dof_state_tensors()
        self.refresh_body_state_tensors()
        self._stages(_lib.ANYMAL_STAGE_OBS)
        return {self._anymals.name: {"obs_buf": self.obs_buf}}

    def is_done(self):
        pass
