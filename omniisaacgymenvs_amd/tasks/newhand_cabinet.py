"""TomoDJnewhandCabinetTask / TomoDJnewhandTendonCabinetTask / TomoSPCnewhandCabinetTask -- host-side mirrors of the fork's
five-finger cabinet tasks (/root/reference/omniisaacgymenvs/tasks/tomodj_newhand_cabinet.py, tomodj_newhandTendon_cabinet.py,
tomospc_newhand_cabinet.py), whose arithmetic runs in liboige_hip.so (csrc/newhand_cabinet.hip).  Same constructor, attributes
and methods (``get_observations / calculate_metrics / is_done / pre_physics_step / reset_idx / post_reset``) plus the fused
``post_physics_step``; the reference's ``tomodj_*`` / ``tomospc_*`` attribute names alias one shared storage (``robot_*``).
No torch/CPU fallback.

Stand-ins for what the reference reads from USD assets that are not in the tree: joint limits, and the DOF order of the
Tendon hand (arm 0-6, thumb 7-9, then four fingers of four joints each whose first three are actuated).
"""
from __future__ import annotations

import ctypes as C

import torch

from .. import _lib
from ..sim.synthetic import SyntheticArticulationView, SyntheticRigidPrimView
from .base.rl_task import RLTask

FINGER_VIEWS = ("_thumbs", "_forefingers", "_middle_fingers", "_ring_fingers", "_little_fingers")


class SyntheticNewhandView(SyntheticArticulationView):
    """TomoDJView / TomoSPCView (newhand) surface used by the tasks (views/tomodj_newhand_view.py, tomodj_newhandTendon_view.py)."""

    def __init__(self, name: str, count: int, device, num_dof: int, num_arm: int, actuated=None) -> None:
        g = torch.Generator().manual_seed(num_dof)
        lo, hi = -(torch.rand(num_dof, generator=g) + 0.3), torch.rand(num_dof, generator=g) + 0.3
        super().__init__(name=name, count=count, num_dof=num_dof, device=device, dof_limits=torch.stack([lo, hi], dim=-1))
        self._hands = SyntheticRigidPrimView("hands_view", count, device)
        for v in FINGER_VIEWS:
            setattr(self, v, SyntheticRigidPrimView(v[1:] + "_view", count, device))
        self._arm_indices = list(range(num_arm))
        if actuated is None:
            self._finger_indices = list(range(num_arm, num_dof))
        else:
            self._finger_indices = [j for j in actuated if j >= num_arm]
        self._actuated_dof_indices = self._arm_indices + self._finger_indices

    finger_indices = property(lambda self: self._finger_indices)
    arm_indices = property(lambda self: self._arm_indices)
    actuated_dof_indices = property(lambda self: self._actuated_dof_indices)


class _NewhandCabinetTask(RLTask):
    _PREFIX = "tomodj"
    NUM_DOFS, NUM_ARM = 26, 7
    ACTUATED = None                      # Tendon: the 22 actuated joints
    ROT_REWARD_THRESHOLDS = False
    TERMINATE_ON_LIMITS = False
    AROUND_UPPER = 0.3149
    KEYPOINT_WIDTH, KEYPOINT_Z0, KEYPOINT_Z1 = 0.1, 0.0165, 0.0135
    LEFT_HAND_FORWARD_AXIS, LEFT_HAND_UP_AXIS = [1, 0, 0], [0, 0, 1]
    DEFAULT_DOF_POS = [0.52, 0.52, 0.26, -0.26, -0.25, 0.17, -0.52] + [0.0] * 19

    # ------------------------------------------------------------------ attribute aliases (tomodj_* / tomospc_* -> robot_*)
    def _alias(self, name: str):
        p = type(self)._PREFIX
        if name.startswith(p + "_"):
            return "robot_" + name[len(p) + 1:]
        if name == f"_{p}s":
            return "_robots"
        if name == f"num_{p}_dofs":
            return "num_robot_dofs"
        return None

    def __getattr__(self, name):
        alias = self._alias(name)
        if alias is None:
            raise AttributeError(f"{type(self).__name__!s} object has no attribute {name!r}")
        return getattr(self, alias)

    def __setattr__(self, name, value):
        object.__setattr__(self, self._alias(name) or name, value)

    def __init__(self, name, sim_config, env, offset=None) -> None:
        self.update_config(sim_config)
        self.distX_offset = 0.04
        self.dt = 1 / 60.0
        self._num_observations = 2 * self.NUM_DOFS + 5
        self._num_actions = self.NUM_DOFS if self.ACTUATED is None else len(self.ACTUATED)
        self.fused = True
        self.sync_free = True
        self.supports_action_clamp = True
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
        from .franka_cabinet import SyntheticCabinetView

        RLTask.set_up_scene(self, scene)
        n, dev = self._num_envs, self.device
        self._robots = SyntheticNewhandView(f"{self._PREFIX}_view", n, dev, self.NUM_DOFS, self.NUM_ARM, self.ACTUATED)
        self._cabinets = SyntheticCabinetView("cabinet_view", n, dev)
        if self.num_props > 0:
            self._props = SyntheticRigidPrimView("prop_view", n * self.num_props, dev)
            self._props.positions[:] = self._env_pos.repeat_interleave(self.num_props, 0) + torch.tensor([0.0515, 0.0, 0.7172], device=dev)
        self.init_data()

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    def init_data(self) -> None:
        dev, n = self._device, self._num_envs
        rep = lambda x: torch.tensor(x, device=dev, dtype=torch.float).repeat((n, 1))
        self.robot_local_thumb_pose_pos = rep([0.038, 0.008, 0.0])
        self.robot_local_another_finger_pose_pos = rep([0.024, -0.002, 0.0])
        self.robot_default_dof_pos = torch.tensor(self.DEFAULT_DOF_POS, dtype=torch.float32, device=dev)
        self.left_hand_forward_axis, self.drawer_inward_axis = rep(self.LEFT_HAND_FORWARD_AXIS), rep([-1, 0, 0])
        self.left_hand_up_axis, self.drawer_up_axis = rep(self.LEFT_HAND_UP_AXIS), rep([0, 0, 1])
        self.actions = torch.zeros((n, self.num_actions), device=dev)

    def post_reset(self):
        n, dev, D = self._num_envs, self._device, self.NUM_DOFS
        f = dict(dtype=torch.float32, device=dev)
        if self.ACTUATED is not None:
            self.actuated_dof_indices = self._robots.actuated_dof_indices
            self._actuated_index = torch.tensor(self.actuated_dof_indices, dtype=torch.int64, device=dev)
        self.num_robot_dofs = self._robots.num_dof
        self.robot_dof_pos = torch.zeros((n, D), device=dev)
        dof_limits = self._robots.get_dof_limits()
        self.robot_dof_lower_limits = dof_limits[0, :, 0].to(device=dev)
        self.robot_dof_upper_limits = dof_limits[0, :, 1].to(device=dev)
        self.robot_dof_speed_scales = torch.ones_like(self.robot_dof_lower_limits)
        self.robot_dof_speed_scales[self._robots._finger_indices] = 5
        self.robot_dof_targets = torch.zeros((n, D), **f)
        self.keypoints_finger = torch.zeros((n, 5, 3), **f)
        self.keypoints_drawer = torch.zeros_like(self.keypoints_finger, device=dev)
        self.identity_quat = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev).unsqueeze(0).repeat(n, 1)
        self.drawer_local_keypoint_offsets = torch.zeros((5, 3), device=dev)
        self.drawer_local_keypoint_offsets[1:, 1] = (torch.linspace(0.0, 1.0, 4, device=dev) - 0.5) * self.KEYPOINT_WIDTH - 0.01
        self.drawer_local_keypoint_offsets[0, 1] = self.drawer_local_keypoint_offsets[1, 1]
        for i in range(5):
            self.drawer_local_keypoint_offsets[i, 0] = 0.305
            self.drawer_local_keypoint_offsets[i, 2] = self.KEYPOINT_Z0 if i == 0 else self.KEYPOINT_Z1
        P = self.num_props
        if P > 0:
            self.default_prop_pos, self.default_prop_rot = self._props.get_world_poses()
            self.prop_indices = torch.arange(n * P, device=dev).view(n, P)
        self.terminate_buf = torch.zeros_like(self.reset_buf)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._tile_counts_u8 = _lib.new_tile_counts_u8(n, dev)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "dof_pos": torch.zeros((n, D), **f), "dof_vel": torch.zeros((n, D), **f),
            "cabinet_dof_pos": torch.zeros((n, 4), **f), "cabinet_dof_vel": torch.zeros((n, 4), **f),
            "prop_ids": torch.zeros(max(n * P, 1), **i64), "prop_count": torch.zeros(1, **i64),
            "prop_pos": torch.zeros((max(n * P, 1), 3), **f), "prop_rot": torch.zeros((max(n * P, 1), 4), **f),
        }
        self._counts_key = None
        self._rows_host = {k: getattr(self, k)[0].tolist() for k in (
            "robot_local_thumb_pose_pos", "robot_local_another_finger_pose_pos", "left_hand_forward_axis", "drawer_inward_axis",
            "left_hand_up_axis", "drawer_up_axis")}
        self._rows_host.update(lower=self.robot_dof_lower_limits.tolist(), upper=self.robot_dof_upper_limits.tolist(),
                               speed=self.robot_dof_speed_scales.tolist(), default=self.robot_default_dof_pos.tolist(),
                               offsets=self.drawer_local_keypoint_offsets.tolist())
        # randomize all envs
        self.reset_idx(torch.arange(n, dtype=torch.int64, device=dev))
        self.terminate_buf = torch.zeros_like(self.reset_buf)

    # ------------------------------------------------------------------ C-ABI argument blocks (cached: RLTask._block keys them
    # on the identity of the tensors they point into and on the value of their scalars)
    _PARAM_SCALARS = ("action_scale", "dof_vel_scale", "dt", "clip_actions", "dist_reward_scale", "rot_reward_scale",
                      "around_handle_reward_scale", "open_reward_scale", "action_penalty_scale", "AROUND_UPPER", "NUM_DOFS",
                      "num_actions", "ROT_REWARD_THRESHOLDS", "TERMINATE_ON_LIMITS", "_max_episode_length", "num_props")

    def _params(self) -> _lib.NewhandCabinetParams:
        d = self.__dict__
        sc = [d[k] if k in d else getattr(self, k) for k in self._PARAM_SCALARS]
        act = None if self.ACTUATED is None else self.actuated_dof_indices
        return self._block("params", (self._rows_host, act), sc, self._build_params)

    def _build_params(self) -> _lib.NewhandCabinetParams:
        p, h, D = _lib.NewhandCabinetParams(), self._rows_host, self.NUM_DOFS
        p.action_scale, p.dof_vel_scale, p.dt = self.action_scale, self.dof_vel_scale, self.dt
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        for k in ("dist_reward_scale", "rot_reward_scale", "around_handle_reward_scale", "open_reward_scale", "action_penalty_scale"):
            setattr(p, k, getattr(self, k))
        p.around_lower, p.around_upper = 0.2732, self.AROUND_UPPER
        p.num_dofs, p.num_actions = D, self.num_actions
        p.rot_reward_thresholds, p.terminate_on_limits = int(self.ROT_REWARD_THRESHOLDS), int(self.TERMINATE_ON_LIMITS)
        p.max_episode_length, p.num_props = int(self._max_episode_length), int(self.num_props)
        for k in range(5):
            for c in range(3):
                p.drawer_local_keypoint_offsets[k][c] = h["offsets"][k][c]
        for dst, src in (("local_thumb_pose_pos", "robot_local_thumb_pose_pos"), ("local_another_finger_pose_pos", "robot_local_another_finger_pose_pos"),
                         ("left_hand_forward_axis",) * 2, ("drawer_inward_axis",) * 2, ("left_hand_up_axis",) * 2, ("drawer_up_axis",) * 2):
            arr = getattr(p, dst)
            for i, x in enumerate(h[src]):
                arr[i] = x
        for i in range(D):
            p.dof_lower[i], p.dof_upper[i] = h["lower"][i], h["upper"][i]
            p.dof_speed_scales[i], p.default_dof_pos[i] = h["speed"][i], h["default"][i]
        act = list(range(D)) if self.ACTUATED is None else list(self.actuated_dof_indices)
        for i, j in enumerate(act):
            p.actuated_dof_indices[i] = int(j)
        return p

    def _buffers(self) -> _lib.NewhandCabinetBuffers:
        ts = (self.obs_buf, self.rew_buf, self.reset_buf, self.progress_buf, self.actions, self.robot_dof_targets,
              self.keypoints_finger, self.keypoints_drawer, self.terminate_buf, self._reset_workspace, self._tile_counts_u8,
              getattr(self, "_actuated_targets", None))
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.NewhandCabinetBuffers:
        n, D, rq = self._num_envs, self.NUM_DOFS, _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, 2 * D + 5), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        rq(self.reset_buf, torch.int64, (n,), "reset_buf")
        rq(self.progress_buf, torch.int64, (n,), "progress_buf")
        rq(self.actions, torch.float32, (n, self.num_actions), "actions")
        rq(self.robot_dof_targets, torch.float32, (n, D), "dof_targets")
        rq(self.keypoints_finger, torch.float32, (n, 5, 3), "keypoints_finger")
        rq(self.keypoints_drawer, torch.float32, (n, 5, 3), "keypoints_drawer")
        rq(self.terminate_buf, torch.int64, (n,), "terminate_buf")
        b = _lib.NewhandCabinetBuffers()
        b.obs_buf, b.rew_buf = self.obs_buf.data_ptr(), self.rew_buf.data_ptr()
        b.reset_buf, b.progress_buf = self.reset_buf.data_ptr(), self.progress_buf.data_ptr()
        b.actions, b.dof_targets = self.actions.data_ptr(), self.robot_dof_targets.data_ptr()
        b.keypoints_finger, b.keypoints_drawer = self.keypoints_finger.data_ptr(), self.keypoints_drawer.data_ptr()
        b.terminate_buf, b.reset_workspace = self.terminate_buf.data_ptr(), self._reset_workspace.data_ptr()
        b.tile_counts_u8 = self._tile_counts_u8.data_ptr()
        if self.ACTUATED is not None:  # Tendon: robot_dof_targets[:, actuated_dof_indices] comes out of the pre kernel
            if getattr(self, "_actuated_targets", None) is None:
                self._actuated_targets = torch.zeros(n, self.num_actions, dtype=torch.float32, device=self._device)
            b.actuated_targets_out = self._actuated_targets.data_ptr()
        return b

    def _state(self) -> _lib.NewhandCabinetState:
        _, self.left_hand_rot = self._robots._hands.get_world_poses(clone=False)
        fingers = [getattr(self._robots, v).get_world_poses(clone=False) for v in FINGER_VIEWS]
        self.fingers_rot = [r for _, r in fingers]
        self.drawer_pos, self.drawer_rot = self._cabinets._drawers.get_world_poses(clone=False)
        self.robot_dof_pos = self._robots.get_joint_positions(clone=False)
        dof_vel = self._robots.get_joint_velocities(clone=False)
        self.cabinet_dof_pos = self._cabinets.get_joint_positions(clone=False)
        self.cabinet_dof_vel = self._cabinets.get_joint_velocities(clone=False)
        ts = (self.left_hand_rot, *(t for pr in fingers for t in pr), self.drawer_pos, self.drawer_rot, self.robot_dof_pos, dof_vel,
              self.cabinet_dof_pos, self.cabinet_dof_vel)
        return self._block("state", ts, (), lambda: self._build_state(fingers, dof_vel))

    def _build_state(self, fingers, dof_vel) -> _lib.NewhandCabinetState:
        n, D, rq = self._num_envs, self.NUM_DOFS, _lib.require_device_tensor
        s = _lib.NewhandCabinetState()
        rq(self.left_hand_rot, torch.float32, (n, 4), "left_hand_rot")
        s.hand_rot = self.left_hand_rot.data_ptr()
        for k, (pos, rot) in enumerate(fingers):
            rq(pos, torch.float32, (n, 3), f"finger {k} pos")
            rq(rot, torch.float32, (n, 4), f"finger {k} rot")
            s.finger_pos[k], s.finger_rot[k] = pos.data_ptr(), rot.data_ptr()
        for t, w, name in ((self.drawer_pos, 3, "drawer_pos"), (self.drawer_rot, 4, "drawer_rot"), (self.robot_dof_pos, D, "dof_pos"),
                           (dof_vel, D, "dof_vel"), (self.cabinet_dof_pos, 4, "cabinet_dof_pos"), (self.cabinet_dof_vel, 4, "cabinet_dof_vel")):
            rq(t, torch.float32, (n, w), name)
        s.drawer_pos, s.drawer_rot = self.drawer_pos.data_ptr(), self.drawer_rot.data_ptr()
        s.dof_pos, s.dof_vel = self.robot_dof_pos.data_ptr(), dof_vel.data_ptr()
        s.cabinet_dof_pos, s.cabinet_dof_vel = self.cabinet_dof_pos.data_ptr(), self.cabinet_dof_vel.data_ptr()
        return s

    def _post(self, stages: int) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        rc = _lib.load().oige_newhand_cabinet_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                           _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_newhand_cabinet_post_physics")
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

    def post_physics_step(self):
        if not self.fused:
            return RLTask.post_physics_step(self)
        self._post(_lib.STAGE_ALL if self._env.world.is_playing() else _lib.STAGE_PROGRESS)
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

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
        self._cabinets.set_joint_positions(o["cabinet_dof_pos"][sl], **kw)
        self._cabinets.set_joint_velocities(o["cabinet_dof_vel"][sl], **kw)
        if P > 0:
            self._props.set_world_poses(o["prop_pos"][psl], o["prop_rot"][psl], **pkw)
        self._robots.set_joint_position_targets(o["dof_pos"][sl], **kw)
        self._robots.set_joint_positions(o["dof_pos"][sl], **kw)
        self._robots.set_joint_velocities(o["dof_vel"][sl], **kw)

    def _checked_actions(self, actions):
        actions = actions.to(self._device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, self.num_actions), "actions")
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
        rc = _lib.load().oige_newhand_cabinet_pre_physics(C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), n, flags,
                                                          _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_newhand_cabinet_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_reset_writes(None)
        if self.ACTUATED is None:
            self._robots.set_joint_position_targets(self.robot_dof_targets)
        else:  # Tendon: only the actuated joints' targets are handed over (joint_indices=...)
            # (a Python list as index would be turned into a device tensor on every call: keep one)
            self._robots.set_joint_position_targets(self._actuated_targets, joint_indices=self.actuated_dof_indices)

    def reset_idx(self, env_ids):
        num = len(env_ids)
        if num == 0:
            return
        dev = self._device
        env_ids = env_ids.to(device=dev, dtype=torch.int64).contiguous()
        u = torch.rand((num, self.num_robot_dofs), device=dev)  # the reference's only draw
        o = self._reset_out
        o["env_ids"][:num] = env_ids
        o["count"].fill_(num)
        p, b, r = self._params(), self._buffers(), self._reset_io(u)
        rc = _lib.load().oige_newhand_cabinet_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, num,
                                                        _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_newhand_cabinet_reset_idx")
        self._counts_key = None
        self._apply_reset_writes(num)


class TomoDJnewhandCabinetTask(_NewhandCabinetTask):
    pass


class TomoDJnewhandTendonCabinetTask(_NewhandCabinetTask):
    ACTUATED = list(range(7)) + [7, 8, 9, 10, 11, 12, 14, 15, 16, 18, 19, 20, 22, 23, 24]
    TERMINATE_ON_LIMITS = True


class TomoSPCnewhandCabinetTask(_NewhandCabinetTask):
    _PREFIX = "tomospc"
    NUM_DOFS, NUM_ARM = 25, 6
    ROT_REWARD_THRESHOLDS = True
    AROUND_UPPER = 0.3154
    KEYPOINT_WIDTH, KEYPOINT_Z0, KEYPOINT_Z1 = 0.075, 0.015, 0.005
    LEFT_HAND_FORWARD_AXIS, LEFT_HAND_UP_AXIS = [0, 1, 0], [1, 0, 0]
    DEFAULT_DOF_POS = [-0.52, 0.26, 0.79, 0.35, 1.05, 0.0] + [0.0] * 19
