"""FrankaCabinetTask -- host-side mirror of /root/reference/omniisaacgymenvs/tasks/franka_cabinet.py:48-601, whose arithmetic
runs in liboige_hip.so (csrc/franka_cabinet.hip).  Same constructor, attributes and methods: ``get_observations /
calculate_metrics / is_done / pre_physics_step / reset_idx / post_reset / compute_grasp_transforms / compute_franka_reward``
plus the fused ``post_physics_step``.  No torch/CPU fallback.

Kept from the reference on purpose:
  * ``franka_rfinger_pos`` is read from the LEFT finger view (:337), so ``around_handle_reward`` and
    ``finger_dist_reward`` are identically zero; set ``task.use_right_finger_view = True`` for the evidently intended binding.
  * ``post_reset`` reads ``self._frankas.gripper_indices`` (:424), which this fork's FrankaView does not define (it has
    ``finger_indices``); the synthetic view provides both, equal to the two finger joints (7, 8).
``init_data`` (:213-281) derives ``franka_local_grasp_pos/rot`` from USD prim poses; without a stage the synthetic view
supplies panda_link7 / finger poses and the same ``tf_inverse`` / ``tf_combine`` chain runs on the host in torch.
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .base.rl_task import RLTask

# stand-ins for values the reference reads from USD assets that are not in the tree (Franka joint limits, link poses)
FRANKA_DOF_LOWER = [-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973, 0.0, 0.0]
FRANKA_DOF_UPPER = [2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973, 0.04, 0.04]


def _quat_mul(a, b):
    w1, x1, y1, z1 = a
    w2, x2, y2, z2 = b
    return torch.stack([w1 * w2 - x1 * x2 - y1 * y2 - z1 * z2, w1 * x2 + x1 * w2 + y1 * z2 - z1 * y2,
                        w1 * y2 - x1 * z2 + y1 * w2 + z1 * x2, w1 * z2 + x1 * y2 - y1 * x2 + z1 * w2])


def _quat_apply(q, v):
    xyz = q[1:]
    t = torch.linalg.cross(xyz, v) * 2
    return v + q[0] * t + torch.linalg.cross(xyz, t)


class SyntheticFrankaView(SyntheticArticulationView):
    """FrankaView surface used by the task (views/franka_view.py:43-100); with ``num_dof=8`` and the finger-view names of
    views/tomodj_view.py:43-100 / tomospc_view.py it also stands in for TomoDJView / TomoSPCView."""

    def __init__(self, name: str, count: int, device, num_dof: int = 9, finger_view_names=("lfingers_view", "rfingers_view")) -> None:
        lo, hi = FRANKA_DOF_LOWER[9 - num_dof:], FRANKA_DOF_UPPER[9 - num_dof:]
        limits = torch.stack([torch.tensor(lo), torch.tensor(hi)], dim=-1)
        names = {f"panda_joint{i + 1}": i for i in range(num_dof - 2)}
        names.update(panda_finger_joint1=num_dof - 2, panda_finger_joint2=num_dof - 1)
        super().__init__(name=name, count=count, num_dof=num_dof, device=device, dof_limits=limits, dof_names=names)
        self._hands = SyntheticRigidPrimView("hands_view", count, device)
        self._lfingers = SyntheticRigidPrimView(finger_view_names[0], count, device)
        self._rfingers = SyntheticRigidPrimView(finger_view_names[1], count, device)
        self._fingertip_views = {finger_view_names[0]: self._lfingers, finger_view_names[1]: self._rfingers}
        # env-local set-up poses of panda_link7 and the two fingers (pos, wxyz), what init_data reads from the stage
        self.link7_pose = torch.tensor([0.0880, 0.0, 0.9260, 0.0, 1.0, 0.0, 0.0])
        self.lfinger_pose = torch.tensor([0.0880, 0.0200, 0.7606, 0.0, 1.0, 0.0, 0.0])
        self.rfinger_pose = torch.tensor([0.0880, -0.0200, 0.7606, 0.0, 1.0, 0.0, 0.0])

    @property
    def finger_indices(self):
        return [self.get_dof_index("panda_finger_joint1"), self.get_dof_index("panda_finger_joint2")]

    gripper_indices = finger_indices
    hands = property(lambda self: self._hands)
    fingertip_views = property(lambda self: self._fingertip_views)


class SyntheticCabinetView(SyntheticArticulationView):
    def __init__(self, name: str, count: int, device) -> None:
        super().__init__(name=name, count=count, num_dof=4, device=device)
        self._drawers = SyntheticRigidPrimView("drawers_view", count, device)

    drawers = property(lambda self: self._drawers)


class FrankaCabinetTask(RLTask):
    # what the TomoDJ / TomoSPC clones change (tasks/tomodj_cabinet.py, tasks/tomospc_cabinet.py)
    NUM_DOFS = 9
    REWARD_VARIANT = _lib.CABINET_REWARD_FRANKA
    OPEN_BONUS_THRESHOLD = 0.2     # :592
    OPEN_DONE_THRESHOLD = 0.39     # :593, :470
    DEFAULT_DOF_POS = [1.157, -1.066, -0.155, -2.239, -1.841, 1.003, 0.469, 0.035, 0.035]  # :277-279
    ROBOT_VIEW_NAME = "franka_view"
    FINGER_VIEW_NAMES = ("lfingers_view", "rfingers_view")
    RIGHT_FINGER_FROM_RIGHT_VIEW = False  # :337 reads the LEFT finger view twice

    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self.distX_offset = 0.04
        self.dt = 1 / 60.0
        self._num_observations = 2 * self.NUM_DOFS + 5
        self._num_actions = self.NUM_DOFS
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
        self.use_right_finger_view = self.RIGHT_FINGER_FROM_RIGHT_VIEW
        self._step_count = 0
        RLTask.__init__(self, name, env)

    def update_config(self, sim_config):
        self._sim_config = sim_config
        self._cfg = sim_config.config
        self._task_cfg = sim_config.task_config
        e = self._task_cfg["env"]
        self._num_envs = e["numEnvs"]
        self._env_spacing = e["envSpacing"]
        self._max_episode_length = e["episodeLength"]
        self.action_scale = e["actionScale"]
        self.start_position_noise = e["startPositionNoise"]
        self.start_rotation_noise = e["startRotationNoise"]
        self.num_props = e["numProps"]
        self.dof_vel_scale = e["dofVelocityScale"]
        self.dist_reward_scale = e["distRewardScale"]
        self.rot_reward_scale = e["rotRewardScale"]
        self.around_handle_reward_scale = e["aroundHandleRewardScale"]
        self.open_reward_scale = e["openRewardScale"]
        self.finger_dist_reward_scale = e["fingerDistRewardScale"]
        self.action_penalty_scale = e["actionPenaltyScale"]
        self.finger_close_reward_scale = e["fingerCloseRewardScale"]

    def set_up_scene(self, scene=None) -> None:
        RLTask.set_up_scene(self, scene)
        n, dev = self._num_envs, self.device
        self._frankas = SyntheticFrankaView(self.ROBOT_VIEW_NAME, n, dev, num_dof=self.NUM_DOFS, finger_view_names=self.FINGER_VIEW_NAMES)
        self._cabinets = SyntheticCabinetView("cabinet_view", n, dev)
        if self.num_props > 0:
            self._props = SyntheticRigidPrimView("prop_view", n * self.num_props, dev)
            self._props.positions[:] = (self._env_pos.repeat_interleave(self.num_props, 0)
                                        + torch.tensor([0.0515, 0.0, 0.7172], device=dev))  # drawer_pos (:173)
        self.init_data()

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def init_data(self) -> None:  # :213-281
        dev, n = self._device, self._num_envs
        v = self._frankas
        hand_pose, lfinger_pose, rfinger_pose = v.link7_pose, v.lfinger_pose, v.rfinger_pose
        finger_pose = torch.zeros(7)
        finger_pose[0:3] = (lfinger_pose[0:3] + rfinger_pose[0:3]) / 2.0
        finger_pose[3:7] = lfinger_pose[3:7]
        inv_rot = hand_pose[3:7] * torch.tensor([1.0, -1.0, -1.0, -1.0])           # tf_inverse
        inv_pos = -_quat_apply(inv_rot, hand_pose[0:3])
        local_rot = _quat_mul(inv_rot, finger_pose[3:7])                            # tf_combine
        local_pos = _quat_apply(inv_rot, finger_pose[0:3]) + inv_pos
        local_pos = local_pos + torch.tensor([0, 0.04, 0])
        self.franka_local_grasp_pos = local_pos.to(dev).repeat((n, 1))
        self.franka_local_grasp_rot = local_rot.to(dev).repeat((n, 1))
        drawer_local_grasp_pose = torch.tensor([0.3, 0.01, 0.0, 1.0, 0.0, 0.0, 0.0], device=dev)
        self.drawer_local_grasp_pos = drawer_local_grasp_pose[0:3].repeat((n, 1))
        self.drawer_local_grasp_rot = drawer_local_grasp_pose[3:7].repeat((n, 1))
        rep = lambda x: torch.tensor(x, device=dev, dtype=torch.float).repeat((n, 1))
        self.gripper_forward_axis, self.drawer_inward_axis = rep([0, 0, 1]), rep([-1, 0, 0])
        self.gripper_up_axis, self.drawer_up_axis = rep([0, 1, 0]), rep([0, 0, 1])
        self.franka_default_dof_pos = torch.tensor(self.DEFAULT_DOF_POS, dtype=torch.float32, device=dev)
        self.actions = torch.zeros((n, self.num_actions), device=dev)
        self._rows_host = None  # the row of every repeated tensor, read back once (no per-step sync)

    def post_reset(self):
        n, dev = self._num_envs, self._device
        f = dict(dtype=torch.float32, device=dev)
        self.num_franka_dofs = self._frankas.num_dof
        self.franka_dof_pos = torch.zeros((n, self.num_franka_dofs), device=dev)
        dof_limits = self._frankas.get_dof_limits()
        self.franka_dof_lower_limits = dof_limits[0, :, 0].to(device=dev)
        self.franka_dof_upper_limits = dof_limits[0, :, 1].to(device=dev)
        self.franka_dof_speed_scales = torch.ones_like(self.franka_dof_lower_limits)
        self.franka_dof_speed_scales[self._frankas.gripper_indices] = 0.1
        self.franka_dof_targets = torch.zeros((n, self.num_franka_dofs), **f)
        self.franka_grasp_rot, self.franka_grasp_pos = torch.zeros((n, 4), **f), torch.zeros((n, 3), **f)
        self.drawer_grasp_rot, self.drawer_grasp_pos = torch.zeros((n, 4), **f), torch.zeros((n, 3), **f)
        P = self.num_props
        if P > 0:
            self.default_prop_pos, self.default_prop_rot = self._props.get_world_poses()
            self.prop_indices = torch.arange(n * P, device=dev).view(n, P)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "dof_pos": torch.zeros((n, self.NUM_DOFS), **f), "dof_vel": torch.zeros((n, self.NUM_DOFS), **f),
            "cabinet_dof_pos": torch.zeros((n, 4), **f), "cabinet_dof_vel": torch.zeros((n, 4), **f),
            "prop_ids": torch.zeros(max(n * P, 1), **i64), "prop_count": torch.zeros(1, **i64),
            "prop_pos": torch.zeros((max(n * P, 1), 3), **f), "prop_rot": torch.zeros((max(n * P, 1), 4), **f),
        }
        self._counts_key = None
        self._rows_host = {k: getattr(self, k)[0].tolist() for k in (
            "franka_local_grasp_pos", "franka_local_grasp_rot", "drawer_local_grasp_pos", "drawer_local_grasp_rot", "gripper_forward_axis",
            "drawer_inward_axis", "gripper_up_axis", "drawer_up_axis")}
        self._rows_host.update(lower=self.franka_dof_lower_limits.tolist(), upper=self.franka_dof_upper_limits.tolist(),
                               speed=self.franka_dof_speed_scales.tolist(), default=self.franka_default_dof_pos.tolist())
        # randomize all envs
        indices = torch.arange(n, dtype=torch.int64, device=dev)
        self.reset_idx(indices)

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    _PARAM_SCALARS = ("action_scale", "dof_vel_scale", "dt", "clip_actions", "dist_reward_scale", "rot_reward_scale",
                      "around_handle_reward_scale", "open_reward_scale", "finger_dist_reward_scale", "action_penalty_scale",
                      "finger_close_reward_scale", "NUM_DOFS", "REWARD_VARIANT", "OPEN_BONUS_THRESHOLD", "OPEN_DONE_THRESHOLD",
                      "_max_episode_length", "num_props")

    def _params(self) -> _lib.FrankaCabinetParams:
        d = self.__dict__
        sc = [d[k] if k in d else getattr(self, k) for k in self._PARAM_SCALARS]
        return self._block("params", (self._rows_host,), sc, self._build_params)

    def _build_params(self) -> _lib.FrankaCabinetParams:
        p, h = _lib.FrankaCabinetParams(), self._rows_host
        p.action_scale, p.dof_vel_scale, p.dt = self.action_scale, self.dof_vel_scale, self.dt
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale", "finger_dist_reward_scale",
                  "action_penalty_scale", "finger_close_reward_scale"):
            setattr(p, k, getattr(self, k))
        for k in ("franka_local_grasp_pos", "franka_local_grasp_rot", "drawer_local_grasp_pos", "drawer_local_grasp_rot",
                  "gripper_forward_axis", "drawer_inward_axis", "gripper_up_axis", "drawer_up_axis"):
            arr = getattr(p, k)
            for i, x in enumerate(h[k]):
                arr[i] = x
        for i in range(self.NUM_DOFS):
            p.dof_lower[i], p.dof_upper[i] = h["lower"][i], h["upper"][i]
            p.dof_speed_scales[i], p.default_dof_pos[i] = h["speed"][i], h["default"][i]
        p.num_dofs, p.reward_variant = self.NUM_DOFS, self.REWARD_VARIANT
        p.open_bonus_threshold, p.open_done_threshold = self.OPEN_BONUS_THRESHOLD, self.OPEN_DONE_THRESHOLD
        p.max_episode_length = int(self._max_episode_length)
        p.num_props = int(self.num_props)
        return p

    def _buffers(self) -> _lib.FrankaCabinetBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.actions, self.franka_dof_targets,
              self.franka_grasp_rot, self.franka_grasp_pos, self.drawer_grasp_rot, self.drawer_grasp_pos, self._reset_workspace,
              self._tile_counts_u8)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.FrankaCabinetBuffers:
        n, rq = self._num_envs, _lib.require_device_tensor
        D = self.NUM_DOFS
        rq(self.obs_buf, torch.float32, (n, 2 * D + 5), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.actions, torch.float32, (n, D), "actions")
        rq(self.franka_dof_targets, torch.float32, (n, D), "franka_dof_targets")
        b = _lib.FrankaCabinetBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.actions, b.franka_dof_targets = self.actions.data_ptr(), self.franka_dof_targets.data_ptr()
        b.franka_grasp_rot, b.franka_grasp_pos = self.franka_grasp_rot.data_ptr(), self.franka_grasp_pos.data_ptr()
        b.drawer_grasp_rot, b.drawer_grasp_pos = self.drawer_grasp_rot.data_ptr(), self.drawer_grasp_pos.data_ptr()
        b.reset_workspace = self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        return b

    def _state(self) -> _lib.FrankaCabinetState:
        hand_pos, hand_rot = self._frankas._hands.get_world_poses(clone=False)
        drawer_pos, drawer_rot = self._cabinets._drawers.get_world_poses(clone=False)
        self.franka_dof_pos = self._frankas.get_joint_positions(clone=False)
        franka_dof_vel = self._frankas.get_joint_velocities(clone=False)
        self.cabinet_dof_pos = self._cabinets.get_joint_positions(clone=False)
        self.cabinet_dof_vel = self._cabinets.get_joint_velocities(clone=False)
        self.franka_lfinger_pos, self.franka_lfinger_rot = self._frankas._lfingers.get_world_poses(clone=False)
        rview = self._frankas._rfingers if self.use_right_finger_view else self._frankas._lfingers  # :337
        self.franka_rfinger_pos, self.franka_rfinger_rot = rview.get_world_poses(clone=False)
        live = (hand_pos, hand_rot, drawer_pos, drawer_rot, franka_dof_vel)
        ts = (*live, self.franka_dof_pos, self.cabinet_dof_pos, self.cabinet_dof_vel, self.franka_lfinger_pos, self.franka_rfinger_pos)
        return self._block("state", ts, (), lambda: self._build_state(*live))

    def _build_state(self, hand_pos, hand_rot, drawer_pos, drawer_rot, franka_dof_vel) -> _lib.FrankaCabinetState:
        n, rq = self._num_envs, _lib.require_device_tensor
        for t, w, name in ((hand_pos, 3, "hand_pos"), (hand_rot, 4, "hand_rot"), (drawer_pos, 3, "drawer_pos"), (drawer_rot, 4, "drawer_rot"),
                           (self.franka_dof_pos, self.NUM_DOFS, "franka_dof_pos"), (franka_dof_vel, self.NUM_DOFS, "franka_dof_vel"),
                           (self.cabinet_dof_pos, 4, "cabinet_dof_pos"), (self.cabinet_dof_vel, 4, "cabinet_dof_vel"),
                           (self.franka_lfinger_pos, 3, "franka_lfinger_pos"), (self.franka_rfinger_pos, 3, "franka_rfinger_pos")):
            rq(t, torch.float32, (n, w), name)
        s = _lib.FrankaCabinetState()
        s.hand_pos, s.hand_rot, s.drawer_pos, s.drawer_rot = hand_pos.data_ptr(), hand_rot.data_ptr(), drawer_pos.data_ptr(), drawer_rot.data_ptr()
        s.franka_dof_pos, s.franka_dof_vel = self.franka_dof_pos.data_ptr(), franka_dof_vel.data_ptr()
        s.cabinet_dof_pos, s.cabinet_dof_vel = self.cabinet_dof_pos.data_ptr(), self.cabinet_dof_vel.data_ptr()
        s.lfinger_pos, s.rfinger_pos = self.franka_lfinger_pos.data_ptr(), self.franka_rfinger_pos.data_ptr()
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_franka_cabinet_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                          _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_franka_cabinet_post_physics")
        if stages & _lib.STAGE_DONE:
            self._mark_counts(self.reset_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self) -> dict:
        self._post(_lib.STAGE_OBS)
        return {self._frankas.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self) -> None:
        self._post(_lib.STAGE_METRICS)

    def is_done(self) -> None:
        self._post(_lib.STAGE_DONE)

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        self._post(_lib.STAGE_ALL if self._env.world.is_playing() else _lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def compute_grasp_transforms(self, *args):
        raise NotImplementedError("compute_grasp_transforms runs inside oige_franka_cabinet_post_physics (obs stage); "
                                  "read task.franka_grasp_rot/pos, task.drawer_grasp_rot/pos after get_observations()")

    def compute_franka_reward(self, *args):
        raise NotImplementedError("compute_franka_reward runs inside oige_franka_cabinet_post_physics (metrics stage)")

    def _reset_io(self, u_dof=None) -> _lib.FrankaCabinetResetIO:
        props = (self.default_prop_pos, self.default_prop_rot) if self.num_props > 0 else ()
        ts = (u_dof, *props, *self._reset_out.values(), self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed, self.num_props), lambda: self._build_reset_io(u_dof))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u_dof) -> _lib.FrankaCabinetResetIO:
        r, o = _lib.FrankaCabinetResetIO(), self._reset_out
        if u_dof is not None:
            r.u_dof = u_dof.data_ptr()
        if self.num_props > 0:
            r.default_prop_pos, r.default_prop_rot = self.default_prop_pos.data_ptr(), self.default_prop_rot.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step_ptr = self._graph_step_ptr()
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        r.cabinet_dof_pos_out, r.cabinet_dof_vel_out = o["cabinet_dof_pos"].data_ptr(), o["cabinet_dof_vel"].data_ptr()
        r.prop_ids, r.prop_count = o["prop_ids"].data_ptr(), o["prop_count"].data_ptr()
        r.prop_pos_out, r.prop_rot_out = o["prop_pos"].data_ptr(), o["prop_rot"].data_ptr()
        return r

    def _apply_reset_writes(self, n=None) -> None:
        o, P = self._reset_out, self.num_props
        if n is None:
            kw, sl = dict(indices=o["env_ids"], count=o["count"]), slice(None)
            pkw, psl = dict(indices=o["prop_ids"], count=o["prop_count"]), slice(None)
        else:
            kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
            pkw, psl = dict(indices=o["prop_ids"][: n * P].to(torch.int32)), slice(0, n * P)
        self._cabinets.set_joint_positions(o["cabinet_dof_pos"][sl], **kw)    # :393-398
        self._cabinets.set_joint_velocities(o["cabinet_dof_vel"][sl], **kw)
        if P > 0:
            self._props.set_world_poses(o["prop_pos"][psl], o["prop_rot"][psl], **pkw)  # :401-406
        self._frankas.set_joint_position_targets(o["dof_pos"][sl], **kw)    # :408 (franka_dof_targets[env_ids] == pos)
        self._frankas.set_joint_positions(o["dof_pos"][sl], **kw)
        self._frankas.set_joint_velocities(o["dof_vel"][sl], **kw)

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, self.NUM_DOFS), "actions")
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
        rc = _lib.load().oige_franka_cabinet_pre_physics(C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), n, flags,
                                                         _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_franka_cabinet_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        self._frankas.set_joint_position_targets(self.franka_dof_targets)  # :371-373, all envs

    def reset_idx(self, env_ids):
        num = len(env_ids)
        if num == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        u = torch.rand((num, self.num_franka_dofs), device=dev)  # the reference's only draw (:382)
        o = self._reset_out
        o["env_ids"][:num] = env_ids
        o["count"].fill_(num)
        p, b, r = self._params(), self._buffers(), self._reset_io(u)
        rc = _lib.load().oige_franka_cabinet_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num,
                                                       _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_franka_cabinet_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num)
