"""InHandManipulationTask -- host-side mirror of the reference class
(/root/reference/omniisaacgymenvs/tasks/shared/in_hand_manipulation.py:43-441) whose arithmetic runs
in liboige_hip.so (csrc/hand.hip); base of ShadowHandTask and AllegroHandTask.  Same method names and buffer contract:
``calculate_metrics / pre_physics_step / is_done / reset_target_pose / reset_idx / post_reset``
plus a fused ``post_physics_step``.  No torch/CPU fallback.

RNG: ``reset_idx`` / ``reset_target_pose`` draw with torch exactly like the reference
(``torch_rand_float(-1, 1, (n, 2*24+5))`` then the nested ``(n, 4)`` draw; :380,:399) and pass the
draws to the kernel; the sync-free ``pre_physics_step`` uses the in-kernel Philox instead.
The PhysX-side ``dr.physics_view.step_randomization`` (:364-372) is delegated and not reproduced.
"""
from __future__ import annotations

import ctypes as C
from abc import abstractmethod

import torch

from ... import _lib
from ..base.rl_task import RLTask

_OBS_TYPES = {"openai": _lib.HAND_OBS_OPENAI, "full": _lib.HAND_OBS_FULL}
_ALLEGRO_OBS_TYPES = {"full_no_vel": _lib.HAND_OBS_ALLEGRO_FULL_NO_VEL, "full": _lib.HAND_OBS_ALLEGRO_FULL}


class InHandManipulationTask(RLTask):
    def __init__(self, name, env, offset=None) -> None:
        InHandManipulationTask.update_config(self)
        self.fused = True
        self.sync_free = True
        self._step_count = 0
        self.supports_action_clamp = True
        RLTask.__init__(self, name, env)

        n, dev = self.num_envs, self.device
        self.x_unit_tensor = torch.tensor([1, 0, 0], dtype=torch.float, device=dev).repeat((n, 1))
        self.y_unit_tensor = torch.tensor([0, 1, 0], dtype=torch.float, device=dev).repeat((n, 1))
        self.z_unit_tensor = torch.tensor([0, 0, 1], dtype=torch.float, device=dev).repeat((n, 1))

        self.reset_goal_buf = self.reset_buf.clone()
        self.successes = torch.zeros(n, dtype=torch.float, device=dev)
        self.consecutive_successes = torch.zeros(1, dtype=torch.float, device=dev)
        self.randomization_buf = torch.zeros(n, dtype=torch.long, device=dev)

        self.av_factor = torch.tensor(self.av_factor, dtype=torch.float, device=dev)
        self._av_factor_host = float(self.av_factor)  # fp32 round trip, as the jit call sees it (:59,:293)
        self.total_successes = 0
        self.total_resets = 0

    def update_config(self):
        e = self._task_cfg["env"]
        self._num_envs = e["numEnvs"]
        self._env_spacing = e["envSpacing"]
        self.dist_reward_scale = e["distRewardScale"]
        self.rot_reward_scale = e["rotRewardScale"]
        self.action_penalty_scale = e["actionPenaltyScale"]
        self.success_tolerance = e["successTolerance"]
        self.reach_goal_bonus = e["reachGoalBonus"]
        self.fall_dist = e["fallDistance"]
        self.fall_penalty = e["fallPenalty"]
        self.rot_eps = e["rotEps"]
        self.vel_obs_scale = e["velObsScale"]
        self.reset_position_noise = e["resetPositionNoise"]
        self.reset_rotation_noise = e["resetRotationNoise"]
        self.reset_dof_pos_noise = e["resetDofPosRandomInterval"]
        self.reset_dof_vel_noise = e["resetDofVelRandomInterval"]
        self.hand_dof_speed_scale = e["dofSpeedScale"]
        self.use_relative_control = e["useRelativeControl"]
        self.act_moving_average = e["actionsMovingAverage"]
        self.max_episode_length = e["episodeLength"]
        self.reset_time = e.get("resetTime", -1.0)
        self.print_success_stat = e["printNumSuccesses"]
        self.max_consecutive_successes = e["maxConsecutiveSuccesses"]
        self.av_factor = e.get("averFactor", 0.1)
        self.dt = 1.0 / 60
        control_freq_inv = e.get("controlFrequencyInv", 1)
        if self.reset_time > 0.0:
            self.max_episode_length = int(round(self.reset_time / (control_freq_inv * self.dt)))
            print("Reset time: ", self.reset_time)
            print("New episode length: ", self.max_episode_length)

    @abstractmethod
    def get_hand_view(self, scene):
        pass

    @abstractmethod
    def get_starting_positions(self):
        pass

    def set_up_scene(self, scene=None) -> None:
        from ...sim.synthetic import SyntheticRigidPrimView

        self.get_starting_positions()
        dev = self.device
        self.object_start_translation = self.hand_start_translation.clone()
        self.object_start_translation[1] += self.pose_dy
        self.object_start_translation[2] += self.pose_dz
        self.object_start_orientation = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev)
        self.goal_displacement_tensor = torch.tensor([-0.2, -0.06, 0.12], device=dev)
        self.goal_start_translation = self.object_start_translation + self.goal_displacement_tensor
        self.goal_start_translation[2] -= 0.04
        self.goal_start_orientation = torch.tensor([1.0, 0.0, 0.0, 0.0], device=dev)

        RLTask.set_up_scene(self, scene)
        self._hands = self.get_hand_view(scene)
        self._objects = SyntheticRigidPrimView("object_view", self._num_envs, dev)
        self._objects.positions[:] = self._env_pos + self.object_start_translation
        self._goals = SyntheticRigidPrimView("goal_view", self._num_envs, dev)
        self._goals.positions[:] = self._env_pos + self.goal_start_translation

    def initialize_views(self, scene=None):
        self.set_up_scene(scene)

    # ------------------------------------------------------------------ C-ABI argument blocks
    _PARAM_SCALARS = ("obs_type", "asymmetric_obs", "max_consecutive_successes", "use_relative_control", "max_episode_length",
                      "dist_reward_scale", "rot_reward_scale", "rot_eps", "action_penalty_scale", "success_tolerance",
                      "reach_goal_bonus", "fall_dist", "fall_penalty", "_av_factor_host", "vel_obs_scale", "force_torque_obs_scale",
                      "hand_dof_speed_scale", "dt", "act_moving_average", "reset_position_noise", "reset_dof_pos_noise",
                      "reset_dof_vel_noise", "clip_actions")

    def _params(self) -> _lib.HandParams:
        """Cached (RLTask._block): rebuilt when one of the public scalars or the host copy of the tables changed."""
        d = self.__dict__
        return self._block("params", (self._host_consts,), [d.get(k) for k in self._PARAM_SCALARS], self._build_params)

    def _build_params(self) -> _lib.HandParams:
        obs_types = _ALLEGRO_OBS_TYPES if self.num_fingertips == 0 else _OBS_TYPES  # AllegroHand has no fingertip bodies
        if self.obs_type not in obs_types:
            raise _lib.OigeError(
                f"observationType {self.obs_type!r}: full_no_vel raises a shape error in the reference "
                "(shadow_hand.py:190) and non-asymmetric full_state is dead code there (shadow_hand.py:263-306)")
        p = _lib.HandParams()
        p.obs_type = obs_types[self.obs_type]
        p.asymmetric_obs = int(bool(self.asymmetric_obs))
        p.max_consecutive_successes = int(self.max_consecutive_successes)
        p.use_relative_control = int(bool(self.use_relative_control))
        p.max_episode_length = float(self.max_episode_length)
        p.dist_reward_scale = self.dist_reward_scale
        p.rot_reward_scale = self.rot_reward_scale
        p.rot_eps = self.rot_eps
        p.action_penalty_scale = self.action_penalty_scale
        p.success_tolerance = self.success_tolerance
        p.reach_goal_bonus = float(self.reach_goal_bonus)
        p.fall_dist = self.fall_dist
        p.fall_penalty = self.fall_penalty
        p.av_factor = self._av_factor_host
        p.vel_obs_scale = self.vel_obs_scale
        p.force_torque_obs_scale = self.force_torque_obs_scale
        p.hand_dof_speed_scale = self.hand_dof_speed_scale
        p.dt = self.dt
        p.act_moving_average = self.act_moving_average
        p.reset_position_noise = self.reset_position_noise
        p.reset_dof_pos_noise = self.reset_dof_pos_noise
        p.reset_dof_vel_noise = self.reset_dof_vel_noise
        p.clip_actions = float(min(self.clip_actions, 3.0e38))
        p.one_minus_act_moving_average = 1.0 - self.act_moving_average
        p.dof_speed_scale_dt = self.hand_dof_speed_scale * self.dt
        h = self._host_consts
        p.goal_displacement[:] = h["goal_displacement"]
        for name in ("dof_lower", "dof_upper", "dof_default_pos", "dof_default_vel"):
            arr = getattr(p, name)
            for j, v in enumerate(h[name]):
                arr[j] = v
        for j, v in enumerate(h["actuated"]):
            p.actuated_dof_indices[j] = v
        return p

    def _refresh_host_consts(self) -> None:
        self._host_consts = {
            "goal_displacement": self.goal_displacement_tensor.tolist(),
            "dof_lower": self.hand_dof_lower_limits.tolist(),
            "dof_upper": self.hand_dof_upper_limits.tolist(),
            "dof_default_pos": self.hand_dof_default_pos.tolist(),
            "dof_default_vel": self.hand_dof_default_vel.tolist(),
            "actuated": [int(i) for i in self.actuated_dof_indices],
        }

    def _buffers(self) -> _lib.HandBuffers:
        ts = (self.obs_buf, self.states_buf if self._num_states > 0 else None, self.rew_buf, self.reset_buf, self.reset_goal_buf,
              self.progress_buf, self.successes, self.consecutive_successes, self.goal_pos, self.goal_rot, self.actions,
              self.prev_targets, self.cur_targets, self.randomization_buf if self._dr_randomizer.randomize else None,
              self._reset_workspace, self._goal_workspace, self._reduce_workspace)
        return self._block("buffers", ts, (), self._build_buffers)

    def _build_buffers(self) -> _lib.HandBuffers:
        n = self._num_envs
        rq = _lib.require_device_tensor
        rq(self.obs_buf, torch.float32, (n, self._num_observations), "obs_buf")
        rq(self.rew_buf, torch.float32, (n,), "rew_buf")
        for name in ("reset_buf", "reset_goal_buf", "progress_buf"):
            rq(getattr(self, name), torch.int64, (n,), name)
        rq(self.successes, torch.float32, (n,), "successes")
        rq(self.consecutive_successes, torch.float32, (1,), "consecutive_successes")
        rq(self.goal_pos, torch.float32, (n, 3), "goal_pos")
        rq(self.goal_rot, torch.float32, (n, 4), "goal_rot")
        rq(self.actions, torch.float32, (n, self._num_actions), "actions")
        rq(self.prev_targets, torch.float32, (n, self.num_hand_dofs), "prev_targets")
        rq(self.cur_targets, torch.float32, (n, self.num_hand_dofs), "cur_targets")
        b = _lib.HandBuffers()
        b.obs_buf = self.obs_buf.data_ptr()
        if self._num_states > 0:
            rq(self.states_buf, torch.float32, (n, self._num_states), "states_buf")
            b.states_buf = self.states_buf.data_ptr()
        b.rew_buf, b.reset_buf = self.rew_buf.data_ptr(), self.reset_buf.data_ptr()
        b.reset_goal_buf, b.progress_buf = self.reset_goal_buf.data_ptr(), self.progress_buf.data_ptr()
        b.successes, b.consecutive_successes = self.successes.data_ptr(), self.consecutive_successes.data_ptr()
        b.goal_pos, b.goal_rot = self.goal_pos.data_ptr(), self.goal_rot.data_ptr()
        b.actions = self.actions.data_ptr()
        b.prev_targets, b.cur_targets = self.prev_targets.data_ptr(), self.cur_targets.data_ptr()
        if self._dr_randomizer.randomize:
            b.randomization_buf = self.randomization_buf.data_ptr()  # only feeds the PhysX-side DR (:364-372)
        b.reset_workspace = self._reset_workspace.data_ptr()
        b.goal_workspace = self._goal_workspace.data_ptr()
        b.reduce_workspace = self._reduce_workspace.data_ptr()
        return b

    def _state(self) -> _lib.HandState:
        """The view getters of get_object_goal_observations (:256-261) and shadow_hand.py:123-135."""
        nf = self.num_fingertips
        object_pos, object_rot = self._objects.get_world_poses(clone=False)
        object_velocities = self._objects.get_velocities(clone=False)
        hand_dof_pos = self._hands.get_joint_positions(clone=False)
        hand_dof_vel = self._hands.get_joint_velocities(clone=False)
        fingers = sensors = (None,)
        if nf > 0:  # shadow_hand.py:123-127
            fingers = (*self._hands._fingers.get_world_poses(clone=False), self._hands._fingers.get_velocities(clone=False))
        want_sensor = self.obs_type == "full_state" or self.asymmetric_obs
        if want_sensor:
            sensors = (self._hands.get_measured_joint_forces(joint_indices=self._hands._sensor_indices),)
        live = (object_pos, object_rot, object_velocities, hand_dof_pos, hand_dof_vel, self._env_pos, *fingers, *sensors)
        return self._block("state", live, (nf, want_sensor), lambda: self._build_state(live, nf, want_sensor))

    def _build_state(self, live, nf, want_sensor) -> _lib.HandState:
        n, nd = self._num_envs, self.num_hand_dofs
        object_pos, object_rot, object_velocities, hand_dof_pos, hand_dof_vel = live[:5]
        rq = _lib.require_device_tensor
        rq(object_pos, torch.float32, (n, 3), "object_pos")
        rq(object_rot, torch.float32, (n, 4), "object_rot")
        rq(object_velocities, torch.float32, (n, 6), "object_velocities")
        rq(hand_dof_pos, torch.float32, (n, nd), "hand_dof_pos")
        rq(hand_dof_vel, torch.float32, (n, nd), "hand_dof_vel")
        rq(self._env_pos, torch.float32, (n, 3), "_env_pos")
        s = _lib.HandState()
        s.object_pos, s.object_rot, s.object_velocities = object_pos.data_ptr(), object_rot.data_ptr(), object_velocities.data_ptr()
        s.hand_dof_pos, s.hand_dof_vel = hand_dof_pos.data_ptr(), hand_dof_vel.data_ptr()
        if nf > 0:  # shadow_hand.py:123-127
            fingertip_pos, fingertip_rot, fingertip_velocities = live[6:9]
            rq(fingertip_pos, torch.float32, (n * nf, 3), "fingertip_pos")
            rq(fingertip_rot, torch.float32, (n * nf, 4), "fingertip_rot")
            rq(fingertip_velocities, torch.float32, (n * nf, 6), "fingertip_velocities")
            s.fingertip_pos, s.fingertip_rot = fingertip_pos.data_ptr(), fingertip_rot.data_ptr()
            s.fingertip_velocities = fingertip_velocities.data_ptr()
        if want_sensor:
            vec_sensor = live[-1]
            rq(vec_sensor, torch.float32, (n, nf, 6), "vec_sensor_tensor")
            s.vec_sensor = vec_sensor.data_ptr()
        s.env_pos = self._env_pos.data_ptr()
        return s

    def _post(self, stages: int, obs_noise=None) -> None:
        p, s, b = self._params(), self._state(), self._buffers()
        if obs_noise is not None:
            rc = _lib.load().oige_hand_post_physics_dr(C.byref(p), C.byref(s), C.byref(b), C.byref(obs_noise), self._num_envs,
                                                       stages, _lib.current_stream(self.obs_buf.device))
            _lib.check(rc, "oige_hand_post_physics_dr")
        else:
            rc = _lib.load().oige_hand_post_physics(C.byref(p), C.byref(s), C.byref(b), self._num_envs, stages,
                                                    _lib.current_stream(self.obs_buf.device))
            _lib.check(rc, "oige_hand_post_physics")
        if stages & _lib.STAGE_METRICS:
            self._mark_counts(self.reset_buf, self.reset_goal_buf)

    # ------------------------------------------------------------------ RLTask methods
    def get_observations(self):
        self._post(_lib.STAGE_OBS)
        return {self._hands.name: {"obs_buf": self.obs_buf}}

    def calculate_metrics(self):
        self._post(_lib.STAGE_METRICS)
        self._after_metrics()

    def _after_metrics(self):
        # :284 `self.consecutive_successes.mean()`: the mean of the (1,) tensor is that element; a 0-dim view of it instead of a
        # 5 us reduction launch per step (the value is rewritten in place by the next step's finalize kernel, like the
        # episode extras of the other tasks)
        cs = self.consecutive_successes
        self.extras["consecutive_successes"] = cs.reshape(()) if cs.numel() == 1 else cs.mean()
        if not self._dr_randomizer.randomize:
            pass  # randomization_buf only feeds the delegated PhysX DR; advanced in-kernel when DR is on
        if self.print_success_stat:
            self.total_resets = self.total_resets + self.reset_buf.sum()
            direct_average_successes = self.total_successes + self.successes.sum()
            self.total_successes = self.total_successes + (self.successes * self.reset_buf).sum()
            print("Direct average consecutive successes = {:.1f}".format(
                direct_average_successes / (self.total_resets + self.num_envs)))
            if self.total_resets > 0:
                print("Post-Reset average consecutive successes = {:.1f}".format(self.total_successes / self.total_resets))

    def is_done(self):
        pass

    @property
    def supports_fused_dr(self) -> bool:
        """True when post/pre_physics_step can apply the wrapper's observation / action noise inside their own kernels."""
        return self.fused and self.sync_free and self.num_fingertips > 0

    def post_physics_step(self, obs_noise=None):
        """``obs_noise``: a NoiseArgs block (Randomizer.fused_observations_args) -> the observation noise and the clamped
        copy of vec_env_rlgames.py:94-97 + :40 are produced while the obs tile streams out of the kernel."""
        if not self.fused:
            if obs_noise is not None:
                raise _lib.OigeError("post_physics_step(obs_noise=...) needs task.fused")
            return RLTask.post_physics_step(self)
        if self._env.world.is_playing():
            self._post(_lib.STAGE_PROGRESS | _lib.STAGE_OBS | _lib.STAGE_METRICS, obs_noise)
            self._after_metrics()
        else:
            self.progress_buf[:] += 1
        return self.obs_buf, self.rew_buf, self.reset_buf, self.extras

    def _reset_io(self, u_goal=None, u_reset=None, u_reset_goal=None) -> _lib.HandResetIO:
        ts = (self.object_init_pos, self.goal_init_pos, self._env_pos, u_goal, u_reset, u_reset_goal, *self._reset_out.values(),
              self._step_dev if self._capturing else None)
        r = self._block("reset_io", ts, (self._rand_seed,), lambda: self._build_reset_io(u_goal, u_reset, u_reset_goal))
        r.step = self._step_count  # the one field that changes from step to step
        return r

    def _build_reset_io(self, u_goal, u_reset, u_reset_goal) -> _lib.HandResetIO:
        r = _lib.HandResetIO()
        r.object_init_pos, r.goal_init_pos = self.object_init_pos.data_ptr(), self.goal_init_pos.data_ptr()
        r.env_pos = self._env_pos.data_ptr()
        if u_goal is not None:
            r.u_goal = u_goal.data_ptr()
        if u_reset is not None:
            r.u_reset, r.u_reset_goal = u_reset.data_ptr(), u_reset_goal.data_ptr()
        r.seed = int(self._rand_seed) & 0xFFFFFFFFFFFFFFFF
        r.step = self._step_count
        r.step_ptr = self._graph_step_ptr()
        o = self._reset_out
        r.goal_ids, r.goal_count = o["goal_ids"].data_ptr(), o["goal_count"].data_ptr()
        r.goal_world_pos_out, r.goal_rot_out = o["goal_world_pos"].data_ptr(), o["goal_rot"].data_ptr()
        r.env_ids, r.count = o["env_ids"].data_ptr(), o["count"].data_ptr()
        r.r_goal_world_pos_out, r.r_goal_rot_out = o["r_goal_world_pos"].data_ptr(), o["r_goal_rot"].data_ptr()
        r.object_pos_out, r.object_rot_out, r.object_vel_out = (
            o["object_pos"].data_ptr(), o["object_rot"].data_ptr(), o["object_vel"].data_ptr())
        r.dof_pos_out, r.dof_vel_out = o["dof_pos"].data_ptr(), o["dof_vel"].data_ptr()
        return r

    def _apply_goal_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:
            self._goals.set_world_poses(o["goal_world_pos"], o["goal_rot"], indices=o["goal_ids"], count=o["goal_count"])
        else:
            self._goals.set_world_poses(o["goal_world_pos"][:n], o["goal_rot"][:n], o["goal_ids"][:n].to(torch.int32))

    def _apply_reset_writes(self, n=None) -> None:
        o = self._reset_out
        if n is None:  # whole-capacity buffers + the device-side count: nothing to slice
            ids, cnt = o["env_ids"], o["count"]
            self._goals.set_world_poses(o["r_goal_world_pos"], o["r_goal_rot"], indices=ids, count=cnt)
            self._objects.set_velocities(o["object_vel"], indices=ids, count=cnt)
            self._objects.set_world_poses(o["object_pos"], o["object_rot"], indices=ids, count=cnt)
            self._hands.set_joint_position_targets(o["dof_pos"], indices=ids, count=cnt)
            self._hands.set_joint_positions(o["dof_pos"], indices=ids, count=cnt)
            self._hands.set_joint_velocities(o["dof_vel"], indices=ids, count=cnt)
            return
        kw, sl = dict(indices=o["env_ids"][:n].to(torch.int32)), slice(0, n)
        self._goals.set_world_poses(o["r_goal_world_pos"][sl], o["r_goal_rot"][sl], **kw)
        self._objects.set_velocities(o["object_vel"][sl], **kw)
        self._objects.set_world_poses(o["object_pos"][sl], o["object_rot"][sl], **kw)
        self._hands.set_joint_position_targets(o["dof_pos"][sl], **kw)
        self._hands.set_joint_positions(o["dof_pos"][sl], **kw)
        self._hands.set_joint_velocities(o["dof_vel"][sl], **kw)

    def pre_physics_step(self, actions, action_noise=None):
        """``action_noise``: a NoiseArgs block over ``actions`` (Randomizer.fused_actions_args) -> the action noise of
        vec_env_rlgames.py:72-75 is applied while the kernel reads the actions (sync-free flow only)."""
        if not self.world.is_playing():
            return
        if action_noise is not None and not self.sync_free:
            raise _lib.OigeError("pre_physics_step(action_noise=...) needs task.sync_free")
        n = self._num_envs
        actions = self._block("actions_in", (actions,), (), lambda: self._checked_actions(actions))
        self._advance_step_counter()
        flags = 0
        o = self._reset_out
        if not self.sync_free:
            # the reference's flow: host-visible id lists, torch-drawn noise (:318-329)
            _lib.compact(self.reset_buf, self._reset_workspace, o["env_ids"], o["count"])
            _lib.compact(self.reset_goal_buf, self._goal_workspace, o["goal_ids"], o["goal_count"])
            ne, ng = int(o["count"].item()), int(o["goal_count"].item())
            env_ids = o["env_ids"][:ne].clone()
            if ng > 0:
                self.reset_target_pose(o["goal_ids"][:ng].clone())
            if ne > 0:
                self.reset_idx(env_ids)
        else:
            if self._counts_valid(self.reset_buf, self.reset_goal_buf):  # (and the group scans: the post launch's last kernel)
                flags |= _lib.PRE_COUNTS_VALID | _lib.PRE_OFFSETS_VALID
        if self.clamp_actions_in_kernel:
            flags |= _lib.PRE_CLAMP_ACTIONS  # the wrapper's clamp (vec_env_rlgames.py:77), folded into the kernel
        p, b, r = self._params(), self._buffers(), self._reset_io()
        if action_noise is not None:
            rc = _lib.load().oige_hand_pre_physics_dr(
                C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), C.byref(action_noise), self.actions.data_ptr(),
                self._targets_out.data_ptr(), n, flags, _lib.current_stream(self.obs_buf.device))
            _lib.check(rc, "oige_hand_pre_physics_dr")
        else:
            rc = _lib.load().oige_hand_pre_physics(
                C.byref(p), actions.data_ptr(), C.byref(b), C.byref(r), self.actions.data_ptr(), self._targets_out.data_ptr(),
                n, flags, _lib.current_stream(self.obs_buf.device))
            _lib.check(rc, "oige_hand_pre_physics")
        self._counts_key = None
        if self.sync_free:
            self._apply_goal_writes(None)
            self._apply_reset_writes(None)
        self._hands.set_joint_position_targets(self._targets_out, indices=None, joint_indices=self.actuated_dof_indices)

    def _checked_actions(self, actions):
        actions = actions.to(self.device)
        _lib.require_device_tensor(actions, torch.float32, (self._num_envs, self._num_actions), "actions")
        return actions

    def reset_target_pose(self, env_ids):
        n = len(env_ids)
        if n == 0:
            return
        env_ids = env_ids.to(device=self.device, dtype=torch.int64).contiguous()
        rand_floats = (2.0 * torch.rand(n, 4, device=self.device) + (-1.0)).contiguous()  # torch_rand_float(-1, 1, (n, 4))
        o = self._reset_out
        o["goal_ids"][:n] = env_ids
        o["goal_count"].fill_(n)
        p, b, r = self._params(), self._buffers(), self._reset_io(u_goal=rand_floats)
        rc = _lib.load().oige_hand_reset_target_pose(C.byref(p), C.byref(b), C.byref(r), self._num_envs, n,
                                                     _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_hand_reset_target_pose")
        self._counts_key = None
        self._apply_goal_writes(n)

    def reset_idx(self, env_ids):
        n = len(env_ids)
        if n == 0:
            return
        env_ids = env_ids.to(device=self.device, dtype=torch.int64).contiguous()
        nd = self.num_hand_dofs
        # same draws, same order as the reference: (n, 2*nd+5) first, then reset_target_pose's (n, 4)
        rand_floats = (2.0 * torch.rand(n, nd * 2 + 5, device=self.device) + (-1.0)).contiguous()
        goal_rand = (2.0 * torch.rand(n, 4, device=self.device) + (-1.0)).contiguous()
        o = self._reset_out
        o["env_ids"][:n] = env_ids
        o["count"].fill_(n)
        p, b, r = self._params(), self._buffers(), self._reset_io(u_reset=rand_floats, u_reset_goal=goal_rand)
        rc = _lib.load().oige_hand_reset_idx(C.byref(p), C.byref(b), C.byref(r), self._num_envs, n,
                                             _lib.current_stream(self.obs_buf.device))
        _lib.check(rc, "oige_hand_reset_idx")
        self._counts_key = None
        self.hand_dof_targets[env_ids, :] = o["dof_pos"][:n]
        self._apply_reset_writes(n)

    def post_reset(self):
        dev, n = self.device, self.num_envs
        self.num_hand_dofs = self._hands.num_dof
        self.actuated_dof_indices = self._hands.actuated_dof_indices
        f = dict(dtype=torch.float, device=dev)
        self.hand_dof_targets = torch.zeros((n, self.num_hand_dofs), **f)
        self.prev_targets = torch.zeros((n, self.num_hand_dofs), **f)
        self.cur_targets = torch.zeros((n, self.num_hand_dofs), **f)

        dof_limits = self._hands.get_dof_limits()
        self.hand_dof_lower_limits, self.hand_dof_upper_limits = torch.t(dof_limits[0].to(dev))
        self.hand_dof_default_pos = torch.zeros(self.num_hand_dofs, **f)
        self.hand_dof_default_vel = torch.zeros(self.num_hand_dofs, **f)

        self.object_init_pos, self.object_init_rot = self._objects.get_world_poses()
        self.object_init_pos -= self._env_pos
        self.object_init_velocities = torch.zeros_like(self._objects.get_velocities(), **f)

        self.goal_pos = self.object_init_pos.clone()
        self.goal_pos[:, 2] -= 0.04
        self.goal_rot = self.object_init_rot.clone()
        self.goal_init_pos = self.goal_pos.clone()
        self.goal_init_rot = self.goal_rot.clone()
        self.actions = torch.zeros((n, self._num_actions), **f)

        # device-side scratch of the HIP path
        self._targets_out = torch.zeros((n, self._num_actions), **f)
        self._reset_workspace = _lib.new_workspace(n, dev)
        self._goal_workspace = _lib.new_workspace(n, dev)
        self._reduce_workspace = torch.zeros(2 * max((n + 63) // 64, 1), **f)
        i64 = dict(dtype=torch.int64, device=dev)
        self._reset_out = {
            "goal_ids": torch.zeros(n, **i64), "goal_count": torch.zeros(1, **i64),
            "goal_world_pos": torch.zeros((n, 3), **f), "goal_rot": torch.zeros((n, 4), **f),
            "env_ids": torch.zeros(n, **i64), "count": torch.zeros(1, **i64),
            "r_goal_world_pos": torch.zeros((n, 3), **f), "r_goal_rot": torch.zeros((n, 4), **f),
            "object_pos": torch.zeros((n, 3), **f), "object_rot": torch.zeros((n, 4), **f),
            "object_vel": torch.zeros((n, 6), **f),
            "dof_pos": torch.zeros((n, self.num_hand_dofs), **f), "dof_vel": torch.zeros((n, self.num_hand_dofs), **f),
        }
        self._counts_key = None
        self._refresh_host_consts()

        # randomize all envs
        indices = torch.arange(self._num_envs, dtype=torch.int64, device=self._device)
        self.reset_idx(indices)

        if self._dr_randomizer.randomize:
            self._dr_randomizer.set_up_domain_randomization(self)
