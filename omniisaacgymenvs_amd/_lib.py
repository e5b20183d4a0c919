"""ctypes binding of liboige_hip.so (include/oige_hip.h).

There is no CPU fallback: if the shared library is missing or a call fails, the task
methods raise.  The library is built in-tree by ``python -m omniisaacgymenvs_amd.build``.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("OIGE_LIB") or os.path.join(_HERE, "liboige_hip.so")  # OIGE_LIB: kernel-variant experiments

OIGE_MAX_DOF = 32
TILE_ENVS = 64

STAGE_PROGRESS, STAGE_OBS, STAGE_METRICS, STAGE_DONE, STAGE_ALL = 1, 2, 4, 8, 15
PRE_CLAMP_ACTIONS, PRE_COUNTS_VALID, PRE_OFFSETS_VALID = 1, 2, 4

_f = C.c_float
_fp = C.c_void_p  # device pointers travel as integers


class LocomotionParams(C.Structure):
    _fields_ = [
        ("num_dof", C.c_int32), ("num_sensors", C.c_int32), ("limit_cost_kind", C.c_int32), ("reserved0", C.c_int32),
        ("dt", _f), ("dof_vel_scale", _f), ("contact_force_scale", _f), ("angular_velocity_scale", _f),
        ("heading_weight", _f), ("up_weight", _f), ("actions_cost_scale", _f), ("energy_cost_scale", _f),
        ("joints_at_limit_cost_scale", _f), ("death_cost", _f), ("termination_height", _f), ("alive_reward_scale", _f),
        ("max_episode_length", _f), ("power_scale", _f), ("clip_actions", _f), ("reserved1", _f),
        ("target", _f * 3), ("inv_start_rot", _f * 4), ("basis_vec0", _f * 3), ("basis_vec1", _f * 3),
        ("reserved2", _f * 3),
        ("dof_limits_lower", _f * OIGE_MAX_DOF), ("dof_limits_upper", _f * OIGE_MAX_DOF),
        ("joint_gears", _f * OIGE_MAX_DOF), ("motor_effort_ratio", _f * OIGE_MAX_DOF),
    ]


class LocomotionState(C.Structure):
    _fields_ = [(n, _fp) for n in ("torso_position", "torso_rotation", "velocities", "dof_pos", "dof_vel", "sensor_force_torques")]


class LocomotionBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "obs_buf", "rew_buf", "reset_buf", "progress_buf", "potentials", "prev_potentials", "up_vec", "heading_vec",
        "actions", "reset_workspace", "tile_counts_u8", "stats_slots")]


class LocomotionResetIO(C.Structure):
    _fields_ = [
        ("initial_root_pos", _fp), ("initial_root_rot", _fp), ("initial_dof_pos", _fp), ("u_pos", _fp), ("u_vel", _fp),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("env_ids", _fp), ("count", _fp), ("dof_pos_out", _fp), ("dof_vel_out", _fp),
        ("root_pos_out", _fp), ("root_rot_out", _fp), ("root_vel_out", _fp), ("step_ptr", _fp),
    ]


class CartpoleParams(C.Structure):
    _fields_ = [
        ("cart_dof_idx", C.c_int32), ("pole_dof_idx", C.c_int32), ("num_dof", C.c_int32), ("reserved0", C.c_int32),
        ("reset_dist", _f), ("max_push_effort", _f), ("max_episode_length", _f), ("reserved1", _f),
    ]


class CartpoleBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "reset_workspace")]


class CartpoleResetIO(C.Structure):
    _fields_ = ([("u", _fp), ("seed", C.c_uint64), ("step", C.c_uint64), ("env_ids", _fp), ("count", _fp),
                ("dof_pos_out", _fp), ("dof_vel_out", _fp)]) + [("step_ptr", _fp)]


class HandParams(C.Structure):
    _fields_ = [
        ("obs_type", C.c_int32), ("asymmetric_obs", C.c_int32), ("max_consecutive_successes", C.c_int32),
        ("use_relative_control", C.c_int32),
        ("max_episode_length", _f), ("dist_reward_scale", _f), ("rot_reward_scale", _f), ("rot_eps", _f),
        ("action_penalty_scale", _f), ("success_tolerance", _f), ("reach_goal_bonus", _f), ("fall_dist", _f),
        ("fall_penalty", _f), ("av_factor", _f), ("vel_obs_scale", _f), ("force_torque_obs_scale", _f),
        ("hand_dof_speed_scale", _f), ("dt", _f), ("act_moving_average", _f), ("reset_position_noise", _f),
        ("reset_dof_pos_noise", _f), ("reset_dof_vel_noise", _f), ("clip_actions", _f),
        ("one_minus_act_moving_average", _f), ("dof_speed_scale_dt", _f), ("reserved0", _f * 3),
        ("goal_displacement", _f * 3), ("reserved1", _f),
        ("dof_lower", _f * OIGE_MAX_DOF), ("dof_upper", _f * OIGE_MAX_DOF),
        ("dof_default_pos", _f * OIGE_MAX_DOF), ("dof_default_vel", _f * OIGE_MAX_DOF),
        ("actuated_dof_indices", C.c_int32 * OIGE_MAX_DOF),
    ]


class HandState(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "object_pos", "object_rot", "object_velocities", "fingertip_pos", "fingertip_rot", "fingertip_velocities",
        "hand_dof_pos", "hand_dof_vel", "vec_sensor", "env_pos")]


class HandBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "obs_buf", "states_buf", "rew_buf", "reset_buf", "reset_goal_buf", "progress_buf", "successes",
        "consecutive_successes", "goal_pos", "goal_rot", "actions", "prev_targets", "cur_targets",
        "randomization_buf", "reset_workspace", "goal_workspace", "reduce_workspace")]


class HandResetIO(C.Structure):
    _fields_ = ([
        ("object_init_pos", _fp), ("goal_init_pos", _fp), ("env_pos", _fp),
        ("u_goal", _fp), ("u_reset", _fp), ("u_reset_goal", _fp),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("goal_ids", _fp), ("goal_count", _fp), ("goal_world_pos_out", _fp), ("goal_rot_out", _fp),
        ("env_ids", _fp), ("count", _fp), ("r_goal_world_pos_out", _fp), ("r_goal_rot_out", _fp),
        ("object_pos_out", _fp), ("object_rot_out", _fp), ("object_vel_out", _fp),
        ("dof_pos_out", _fp), ("dof_vel_out", _fp),
    ]) + [("step_ptr", _fp)]


HAND_OBS_OPENAI, HAND_OBS_FULL = 0, 2
HAND_OBS_ALLEGRO_FULL_NO_VEL, HAND_OBS_ALLEGRO_FULL = 3, 4


AR_TERMINATION, AR_LIN_VEL_XY, AR_LIN_VEL_Z, AR_ANG_VEL_Z, AR_ANG_VEL_XY, AR_ORIENT, AR_TORQUE, AR_JOINT_ACC, \
    AR_BASE_HEIGHT, AR_ACTION_RATE, AR_HIP, AR_FALLEN_OVER, AR_COUNT = range(13)


class AnymalParams(C.Structure):
    _fields_ = [
        ("lin_vel_scale", _f), ("ang_vel_scale", _f), ("dof_pos_scale", _f), ("dof_vel_scale", _f),
        ("height_meas_scale", _f), ("action_scale", _f), ("Kp", _f), ("Kd", _f), ("max_episode_length_s", _f),
        ("max_episode_length", C.c_int32), ("curriculum", C.c_int32), ("init_done", C.c_int32), ("add_noise", C.c_int32),
        ("env_rows", C.c_int32), ("env_cols", C.c_int32), ("hs_rows", C.c_int32), ("hs_cols", C.c_int32),
        ("horizontal_scale", _f), ("vertical_scale", _f), ("border_size", _f), ("half_env_length", _f),
        ("rew_scales", _f * 12), ("cmd_lo", _f * 3), ("cmd_span", _f * 3), ("base_init_state", _f * 13),
        ("default_dof_pos", _f * 12), ("noise_scales", _f * 6), ("reserved", _f * 3),
    ]


class AnymalState(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "base_pos", "base_quat", "base_velocities", "dof_pos", "dof_vel", "knee_contact", "base_contact",
        "height_samples", "terrain_origins", "height_min")]


class AnymalBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in (
        "obs_buf", "rew_buf", "reset_buf", "has_fallen", "timeout_buf", "progress_buf", "commands", "actions",
        "last_actions", "last_dof_vel", "torques", "feet_air_time", "episode_sums", "terrain_levels", "terrain_types",
        "env_origins", "extras", "reset_workspace", "reduce_workspace", "tile_counts_u8")]


class AnymalResetIO(C.Structure):
    _fields_ = ([
        ("u_dof_offset", _fp), ("u_dof_vel", _fp), ("u_xy", _fp), ("u_cmd", _fp), ("noise_u", _fp), ("reset_mask", _fp),
        ("seed", C.c_uint64), ("step", C.c_uint64),
        ("env_ids", _fp), ("count", _fp), ("base_pos_out", _fp), ("base_quat_out", _fp), ("base_vel_out", _fp),
        ("dof_pos_out", _fp), ("dof_vel_out", _fp),
    ]) + [("step_ptr", _fp), ("push_velocities", _fp), ("push_count", _fp), ("push_interval", C.c_int64), ("push_offset", C.c_int64)]


class AnymalFrame(C.Structure):
    _fields_ = [("base_lin_vel", _fp), ("base_ang_vel", _fp), ("projected_gravity", _fp)]


(ANYMAL_STAGE_PROGRESS, ANYMAL_STAGE_FRAME, ANYMAL_STAGE_TERMINATION, ANYMAL_STAGE_METRICS, ANYMAL_STAGE_RESET, ANYMAL_STAGE_OBS,
 ANYMAL_STAGE_NOISE, ANYMAL_STAGE_LAST, ANYMAL_STAGE_ALL) = 1, 2, 4, 8, 16, 32, 64, 128, 255


class AnymalFlatParams(C.Structure):
    _fields_ = [
        ("lin_vel_scale", _f), ("ang_vel_scale", _f), ("dof_pos_scale", _f), ("dof_vel_scale", _f), ("action_scale", _f),
        ("dt", _f), ("base_threshold", _f), ("clip_actions", _f), ("max_episode_length", C.c_int32), ("reserved0", C.c_int32),
        ("rew_scales", _f * 6), ("cmd_lo", _f * 3), ("cmd_span", _f * 3), ("default_dof_pos", _f * 12), ("dof_lower", _f * 12),
        ("dof_upper", _f * 12),
    ]


class AnymalFlatState(C.Structure):
    _fields_ = [(n, _fp) for n in ("base_pos", "base_rot", "velocities", "dof_pos", "dof_vel")]


class AnymalFlatBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "commands", "actions", "last_actions",
                                   "last_dof_vel", "current_targets", "reset_workspace", "tile_counts_u8")]


class AnymalFlatResetIO(C.Structure):
    _fields_ = ([("initial_root_pos", _fp), ("initial_root_rot", _fp), ("u_vel", _fp), ("u_cmd", _fp), ("seed", C.c_uint64),
                ("step", C.c_uint64), ("env_ids", _fp), ("count", _fp), ("dof_pos_out", _fp), ("dof_vel_out", _fp),
                ("root_pos_out", _fp), ("root_rot_out", _fp), ("root_vel_out", _fp)]) + [("step_ptr", _fp)]


class QuadcopterParams(C.Structure):
    _fields_ = [("dof_rate", _f), ("thrust_rate", _f), ("max_thrust", _f), ("clip_actions", _f),
                ("max_episode_length", C.c_int32), ("reserved0", C.c_int32 * 3), ("dof_lower", _f * 8), ("dof_upper", _f * 8)]


class QuadcopterState(C.Structure):
    _fields_ = [(n, _fp) for n in ("root_pos", "root_rot", "root_velocities", "dof_pos", "env_pos")]


class QuadcopterBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "target_positions",
                                   "dof_position_targets", "thrusts", "forces", "reset_workspace", "tile_counts_u8")]


class QuadcopterResetIO(C.Structure):
    _fields_ = ([("initial_root_pos", _fp), ("initial_root_rot", _fp), ("u_dof", _fp), ("u_xyz", _fp), ("seed", C.c_uint64),
                ("step", C.c_uint64), ("env_ids", _fp), ("count", _fp), ("dof_pos_out", _fp), ("dof_vel_out", _fp),
                ("root_pos_out", _fp), ("root_rot_out", _fp), ("root_vel_out", _fp)]) + [("step_ptr", _fp)]


class BallBalanceParams(C.Structure):
    _fields_ = [("rate", _f), ("ball_low", _f), ("clip_actions", _f), ("max_episode_length", C.c_int32), ("num_dof", C.c_int32),
                ("actuated_dof_indices", C.c_int32 * 3), ("dof_lower", _f * 8), ("dof_upper", _f * 8)]


class BallBalanceState(C.Structure):
    _fields_ = [(n, _fp) for n in ("ball_pos", "ball_velocities", "dof_pos", "dof_vel", "sensors", "env_pos")]


class BallBalanceBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "dof_position_targets", "reset_workspace", "tile_counts_u8")]


class BallBalanceResetIO(C.Structure):
    _fields_ = ([(n, _fp) for n in ("initial_ball_pos", "initial_bot_pos", "initial_bot_rot", "initial_bot_velocities",
                                   "initial_dof_positions", "u")] + [("seed", C.c_uint64), ("step", C.c_uint64)] + [
        (n, _fp) for n in ("env_ids", "count", "ball_pos_out", "ball_rot_out", "ball_vel_out", "bot_pos_out", "bot_rot_out",
                           "bot_vel_out", "dof_pos_out")]) + [("step_ptr", _fp)]


class IngenuityParams(C.Structure):
    _fields_ = [("dt", _f), ("thrust_limit", _f), ("thrust_lateral_component", _f), ("clip_actions", _f),
                ("max_episode_length", C.c_int32), ("reserved0", C.c_int32 * 3)]


class IngenuityState(C.Structure):
    _fields_ = [(n, _fp) for n in ("root_pos", "root_rot", "root_velocities", "env_pos")]


class IngenuityBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "target_positions", "thrusts", "dof_pos",
                                   "dof_vel", "reset_workspace", "target_workspace", "reset_counts_u8", "target_counts_u8")]


class IngenuityResetIO(C.Structure):
    _fields_ = ([(n, _fp) for n in ("initial_root_pos", "initial_root_rot", "initial_ball_rot", "env_pos", "u_reset", "u_target")] + [
        ("seed", C.c_uint64), ("step", C.c_uint64)] + [
        (n, _fp) for n in ("env_ids", "count", "dof_pos_out", "dof_vel_out", "root_pos_out", "root_rot_out", "root_vel_out",
                           "target_ids", "target_count", "ball_pos_out", "ball_rot_out")]) + [("step_ptr", _fp)]


class FrankaCabinetParams(C.Structure):
    _fields_ = [(n, _f) for n in ("action_scale", "dof_vel_scale", "dt", "clip_actions", "dist_reward_scale", "rot_reward_scale",
                                  "around_handle_reward_scale", "open_reward_scale", "finger_dist_reward_scale",
                                  "action_penalty_scale", "finger_close_reward_scale", "open_bonus_threshold",
                                  "open_done_threshold")] + [
        ("num_dofs", C.c_int32), ("reward_variant", C.c_int32), ("reserved0", C.c_int32), ("franka_local_grasp_pos", _f * 3), ("franka_local_grasp_rot", _f * 4), ("drawer_local_grasp_pos", _f * 3),
        ("drawer_local_grasp_rot", _f * 4), ("gripper_forward_axis", _f * 3), ("drawer_inward_axis", _f * 3),
        ("gripper_up_axis", _f * 3), ("drawer_up_axis", _f * 3), ("dof_lower", _f * 9), ("dof_upper", _f * 9),
        ("dof_speed_scales", _f * 9), ("default_dof_pos", _f * 9), ("max_episode_length", C.c_int32), ("num_props", C.c_int32)]


class FrankaCabinetState(C.Structure):
    _fields_ = [(n, _fp) for n in ("hand_pos", "hand_rot", "drawer_pos", "drawer_rot", "franka_dof_pos", "franka_dof_vel",
                                   "cabinet_dof_pos", "cabinet_dof_vel", "lfinger_pos", "rfinger_pos")]


class FrankaCabinetBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "actions", "franka_dof_targets",
                                   "franka_grasp_rot", "franka_grasp_pos", "drawer_grasp_rot", "drawer_grasp_pos", "reset_workspace", "tile_counts_u8")]


class FrankaCabinetResetIO(C.Structure):
    _fields_ = ([("u_dof", _fp), ("default_prop_pos", _fp), ("default_prop_rot", _fp), ("seed", C.c_uint64), ("step", C.c_uint64)] + [
        (n, _fp) for n in ("env_ids", "count", "dof_pos_out", "dof_vel_out", "cabinet_dof_pos_out", "cabinet_dof_vel_out", "prop_ids",
                           "prop_count", "prop_pos_out", "prop_rot_out")]) + [("step_ptr", _fp)]


class NewhandCabinetParams(C.Structure):
    _fields_ = [(n, _f) for n in ("action_scale", "dof_vel_scale", "dt", "clip_actions", "dist_reward_scale", "rot_reward_scale",
                                  "around_handle_reward_scale", "open_reward_scale", "action_penalty_scale", "around_lower",
                                  "around_upper", "reserved0")] + [
        (n, C.c_int32) for n in ("num_dofs", "num_actions", "rot_reward_thresholds", "terminate_on_limits", "max_episode_length",
                                 "num_props")] + [
        ("drawer_local_keypoint_offsets", (_f * 3) * 5), ("local_thumb_pose_pos", _f * 3), ("local_another_finger_pose_pos", _f * 3),
        ("left_hand_forward_axis", _f * 3), ("drawer_inward_axis", _f * 3), ("left_hand_up_axis", _f * 3), ("drawer_up_axis", _f * 3),
        ("dof_lower", _f * 26), ("dof_upper", _f * 26), ("dof_speed_scales", _f * 26), ("default_dof_pos", _f * 26),
        ("actuated_dof_indices", C.c_int32 * 26)]


class NewhandCabinetState(C.Structure):
    _fields_ = [("hand_rot", _fp), ("finger_pos", _fp * 5), ("finger_rot", _fp * 5)] + [
        (n, _fp) for n in ("drawer_pos", "drawer_rot", "dof_pos", "dof_vel", "cabinet_dof_pos", "cabinet_dof_vel")]


class NewhandCabinetBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "actions", "dof_targets", "keypoints_finger",
                                   "keypoints_drawer", "terminate_buf", "reset_workspace", "actuated_targets_out", "tile_counts_u8")]


OBSERVER_MAX_KEYS = 64
CABINET_REWARD_FRANKA, CABINET_REWARD_TOMO = 0, 1


class ObserverBatch(C.Structure):
    _fields_ = [("values", _fp * OBSERVER_MAX_KEYS), ("lengths", C.c_int32 * OBSERVER_MAX_KEYS), ("num_keys", C.c_int32),
                ("reserved0", C.c_int32)]


class EpisodeStatsIo(C.Structure):
    _fields_ = [("rew_buf", _fp), ("reset_buf", _fp), ("progress_buf", _fp), ("successes", _fp), ("slots", _fp),
                ("reset_elem_bytes", C.c_int32), ("reserved0", C.c_int32)]


EPISODE_STATS_MAX_BLOCKS = 1024


class CrazyflieParams(C.Structure):
    _fields_ = [("motor_tau_up", _f), ("motor_tau_down", _f), ("prop_max_rot", _f), ("clip_actions", _f), ("thrust_max", _f * 4),
                ("max_episode_length", C.c_int32), ("reserved0", C.c_int32 * 3)]


class CrazyflieState(C.Structure):
    _fields_ = [(n, _fp) for n in ("root_pos", "root_rot", "root_velocities", "env_pos")]


class CrazyflieBuffers(C.Structure):
    _fields_ = [(n, _fp) for n in ("obs_buf", "rew_buf", "reset_buf", "progress_buf", "target_positions", "actions", "thrusts",
                                   "thrust_cmds_damp", "thrust_rot_damp", "dof_pos", "dof_vel", "episode_sums", "extras",
                                   "reset_workspace", "target_workspace", "reduce_workspace")]


class CrazyflieResetIO(C.Structure):
    _fields_ = [(n, _fp) for n in ("initial_root_pos", "initial_root_rot", "initial_ball_rot", "env_pos", "thrust_noise4", "env_ids",
                                   "count", "dof_pos_out", "dof_vel_out", "root_pos_out", "root_rot_out", "root_vel_out",
                                   "target_ids", "target_count", "ball_pos_out", "ball_rot_out")]


NOISE_ADDITIVE, NOISE_SCALING = 0, 1
NOISE_GAUSSIAN, NOISE_UNIFORM, NOISE_LOGUNIFORM = 0, 1, 2


class NoiseTerm(C.Structure):
    _fields_ = [("enabled", C.c_int32), ("operation", C.c_int32), ("distribution", C.c_int32),
                ("frequency_interval", C.c_int32), ("p0", _f), ("p1", _f)]


class NoiseArgs(C.Structure):
    _fields_ = ([(n, _fp) for n in (
        "buffer", "reset_buf", "counter", "correlated_noise", "clamped_out", "reset_samples", "interval_samples",
        "reset_workspace", "interval_workspace")] + [
        ("seed", C.c_uint64), ("step", C.c_uint64), ("stream_id", C.c_int32), ("width", C.c_int32), ("clip", _f),
        ("reserved0", _f), ("on_reset", NoiseTerm), ("on_interval", NoiseTerm)]) + [("step_ptr", _fp)]


# name -> (restype, argtypes); every symbol include/oige_hip.h declares
_P = C.c_void_p
SIGNATURES = {
    "oige_last_error": (C.c_char_p, []),
    "oige_version": (C.c_int, []),
    "oige_num_tiles": (C.c_int64, [C.c_int64]),
    "oige_workspace_ints": (C.c_int64, [C.c_int64]),
    "oige_tile_counts_u8_bytes": (C.c_int64, [C.c_int64]),
    "oige_count_flags_i64": (C.c_int, [_P, C.c_int64, _P, _P]),
    "oige_compact_i64": (C.c_int, [_P, C.c_int64, _P, _P, _P, _P]),
    "oige_compact_static_max_envs": (C.c_int64, []),
    "oige_compact_i64_static": (C.c_int, [_P, C.c_int64, _P, _P, _P]),
    "oige_locomotion_post_physics": (C.c_int, [C.POINTER(LocomotionParams), C.POINTER(LocomotionState),
                                               C.POINTER(LocomotionBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_locomotion_dof_at_limit_cost": (C.c_int, [C.POINTER(LocomotionParams), _P, _P, C.c_int64, _P]),
    "oige_locomotion_pre_physics": (C.c_int, [C.POINTER(LocomotionParams), _P, C.POINTER(LocomotionBuffers),
                                              C.POINTER(LocomotionResetIO), _P, _P, C.c_int64, C.c_uint32, _P]),
    "oige_locomotion_reset_idx": (C.c_int, [C.POINTER(LocomotionParams), C.POINTER(LocomotionBuffers),
                                            C.POINTER(LocomotionResetIO), C.c_int64, C.c_int64, _P]),
    "oige_cartpole_post_physics": (C.c_int, [C.POINTER(CartpoleParams), _P, _P, C.POINTER(CartpoleBuffers), C.c_int64,
                                             C.c_uint32, _P]),
    "oige_cartpole_pre_physics": (C.c_int, [C.POINTER(CartpoleParams), _P, C.POINTER(CartpoleBuffers),
                                            C.POINTER(CartpoleResetIO), _P, C.c_int64, C.c_uint32, _P]),
    "oige_cartpole_reset_idx": (C.c_int, [C.POINTER(CartpoleParams), C.POINTER(CartpoleBuffers),
                                          C.POINTER(CartpoleResetIO), C.c_int64, C.c_int64, _P]),
    "oige_hand_post_physics": (C.c_int, [C.POINTER(HandParams), C.POINTER(HandState), C.POINTER(HandBuffers), C.c_int64,
                                         C.c_uint32, _P]),
    "oige_hand_pre_physics": (C.c_int, [C.POINTER(HandParams), _P, C.POINTER(HandBuffers), C.POINTER(HandResetIO), _P, _P,
                                        C.c_int64, C.c_uint32, _P]),
    "oige_hand_post_physics_dr": (C.c_int, [C.POINTER(HandParams), C.POINTER(HandState), C.POINTER(HandBuffers),
                                            C.POINTER(NoiseArgs), C.c_int64, C.c_uint32, _P]),
    "oige_hand_pre_physics_dr": (C.c_int, [C.POINTER(HandParams), _P, C.POINTER(HandBuffers), C.POINTER(HandResetIO),
                                           C.POINTER(NoiseArgs), _P, _P, C.c_int64, C.c_uint32, _P]),
    "oige_hand_reset_idx": (C.c_int, [C.POINTER(HandParams), C.POINTER(HandBuffers), C.POINTER(HandResetIO), C.c_int64,
                                      C.c_int64, _P]),
    "oige_hand_reset_target_pose": (C.c_int, [C.POINTER(HandParams), C.POINTER(HandBuffers), C.POINTER(HandResetIO),
                                              C.c_int64, C.c_int64, _P]),
    "oige_anymal_flat_post_physics": (C.c_int, [C.POINTER(AnymalFlatParams), C.POINTER(AnymalFlatState),
                                                C.POINTER(AnymalFlatBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_anymal_flat_pre_physics": (C.c_int, [C.POINTER(AnymalFlatParams), _P, C.POINTER(AnymalFlatBuffers),
                                               C.POINTER(AnymalFlatResetIO), _P, C.c_int64, C.c_uint32, _P]),
    "oige_anymal_flat_reset_idx": (C.c_int, [C.POINTER(AnymalFlatParams), C.POINTER(AnymalFlatBuffers),
                                             C.POINTER(AnymalFlatResetIO), C.c_int64, C.c_int64, _P]),
    "oige_quadcopter_post_physics": (C.c_int, [C.POINTER(QuadcopterParams), C.POINTER(QuadcopterState),
                                               C.POINTER(QuadcopterBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_quadcopter_pre_physics": (C.c_int, [C.POINTER(QuadcopterParams), _P, C.POINTER(QuadcopterBuffers),
                                              C.POINTER(QuadcopterResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_quadcopter_reset_idx": (C.c_int, [C.POINTER(QuadcopterParams), C.POINTER(QuadcopterBuffers),
                                            C.POINTER(QuadcopterResetIO), C.c_int64, C.c_int64, _P]),
    "oige_ball_balance_post_physics": (C.c_int, [C.POINTER(BallBalanceParams), C.POINTER(BallBalanceState),
                                                 C.POINTER(BallBalanceBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_ball_balance_pre_physics": (C.c_int, [C.POINTER(BallBalanceParams), _P, C.POINTER(BallBalanceBuffers),
                                                C.POINTER(BallBalanceResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_ball_balance_reset_idx": (C.c_int, [C.POINTER(BallBalanceParams), C.POINTER(BallBalanceBuffers),
                                              C.POINTER(BallBalanceResetIO), C.c_int64, C.c_int64, _P]),
    "oige_ingenuity_post_physics": (C.c_int, [C.POINTER(IngenuityParams), C.POINTER(IngenuityState), C.POINTER(IngenuityBuffers),
                                              C.c_int64, C.c_uint32, _P]),
    "oige_ingenuity_pre_physics": (C.c_int, [C.POINTER(IngenuityParams), _P, C.POINTER(IngenuityBuffers),
                                             C.POINTER(IngenuityResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_ingenuity_reset_idx": (C.c_int, [C.POINTER(IngenuityParams), C.POINTER(IngenuityBuffers), C.POINTER(IngenuityResetIO),
                                           C.c_int64, C.c_int64, _P]),
    "oige_ingenuity_set_targets": (C.c_int, [C.POINTER(IngenuityParams), C.POINTER(IngenuityBuffers), C.POINTER(IngenuityResetIO),
                                             C.c_int64, C.c_int64, _P]),
    "oige_crazyflie_post_physics": (C.c_int, [C.POINTER(CrazyflieParams), C.POINTER(CrazyflieState), C.POINTER(CrazyflieBuffers),
                                              C.c_int64, C.c_uint32, _P]),
    "oige_crazyflie_pre_physics": (C.c_int, [C.POINTER(CrazyflieParams), _P, _P, C.POINTER(CrazyflieBuffers),
                                             C.POINTER(CrazyflieResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_crazyflie_reset_idx": (C.c_int, [C.POINTER(CrazyflieParams), C.POINTER(CrazyflieBuffers), C.POINTER(CrazyflieResetIO),
                                           C.c_int64, C.c_int64, _P]),
    "oige_crazyflie_set_targets": (C.c_int, [C.POINTER(CrazyflieParams), C.POINTER(CrazyflieBuffers), C.POINTER(CrazyflieResetIO),
                                             C.c_int64, C.c_int64, _P]),
    "oige_franka_cabinet_post_physics": (C.c_int, [C.POINTER(FrankaCabinetParams), C.POINTER(FrankaCabinetState),
                                                   C.POINTER(FrankaCabinetBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_franka_cabinet_pre_physics": (C.c_int, [C.POINTER(FrankaCabinetParams), _P, C.POINTER(FrankaCabinetBuffers),
                                                  C.POINTER(FrankaCabinetResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_franka_cabinet_reset_idx": (C.c_int, [C.POINTER(FrankaCabinetParams), C.POINTER(FrankaCabinetBuffers),
                                                C.POINTER(FrankaCabinetResetIO), C.c_int64, C.c_int64, _P]),
    "oige_newhand_cabinet_post_physics": (C.c_int, [C.POINTER(NewhandCabinetParams), C.POINTER(NewhandCabinetState),
                                                    C.POINTER(NewhandCabinetBuffers), C.c_int64, C.c_uint32, _P]),
    "oige_newhand_cabinet_pre_physics": (C.c_int, [C.POINTER(NewhandCabinetParams), _P, C.POINTER(NewhandCabinetBuffers),
                                                   C.POINTER(FrankaCabinetResetIO), C.c_int64, C.c_uint32, _P]),
    "oige_newhand_cabinet_reset_idx": (C.c_int, [C.POINTER(NewhandCabinetParams), C.POINTER(NewhandCabinetBuffers),
                                                 C.POINTER(FrankaCabinetResetIO), C.c_int64, C.c_int64, _P]),
    "oige_episode_stats_blocks": (C.c_int64, [C.c_int64]),
    "oige_episode_stats_accumulate": (C.c_int, [C.POINTER(EpisodeStatsIo), C.c_int64, _P]),
    "oige_episode_stats_fold": (C.c_int, [_P, C.c_int64, _P, C.c_double, _P, _P]),
    "oige_episode_stats_fold_scratch": (C.c_int64, [C.c_int64]),
    "oige_observer_accumulate": (C.c_int, [C.POINTER(ObserverBatch), _P, _P, _P]),
    "oige_observer_means": (C.c_int, [_P, _P, _P, C.c_int32, C.c_int32, _P]),
    "oige_stream_copy": (C.c_int, [_P, _P, C.c_int64, C.c_int32, _P]),
    "oige_dr_noise_count": (C.c_int, [C.POINTER(NoiseArgs), C.c_int64, _P]),
    "oige_dr_noise_apply": (C.c_int, [C.POINTER(NoiseArgs), C.c_int64, _P]),
    "oige_anymal_height_min": (C.c_int, [_P, C.c_int32, C.c_int32, _P, _P]),
    "oige_anymal_torques": (C.c_int, [C.POINTER(AnymalParams), _P, _P, _P, _P, C.c_int64, _P]),
    "oige_anymal_torques_clone": (C.c_int, [C.POINTER(AnymalParams), _P, _P, _P, _P, _P, C.c_int64, _P]),
    "oige_anymal_post_physics": (C.c_int, [C.POINTER(AnymalParams), C.POINTER(AnymalState), C.POINTER(AnymalBuffers),
                                           C.POINTER(AnymalResetIO), C.c_int64, _P]),
    "oige_anymal_reset_idx": (C.c_int, [C.POINTER(AnymalParams), C.POINTER(AnymalState), C.POINTER(AnymalBuffers),
                                        C.POINTER(AnymalResetIO), C.c_int64, _P]),
    "oige_anymal_post_physics_stages": (C.c_int, [C.POINTER(AnymalParams), C.POINTER(AnymalState), C.POINTER(AnymalBuffers),
                                                  C.POINTER(AnymalResetIO), C.POINTER(AnymalFrame), C.c_int64, C.c_uint32, _P]),
    "oige_anymal_update_terrain_level": (C.c_int, [C.POINTER(AnymalParams), _P, _P, _P, _P, _P, _P, _P, C.c_int64, C.c_int64, _P]),
}

_lib: Optional[C.CDLL] = None


class OigeError(RuntimeError):
    pass


def load() -> C.CDLL:
    """Load liboige_hip.so; raises loudly when it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OigeError(
            f"{LIB_PATH} is missing: build it with `python -m omniisaacgymenvs_amd.build` "
            "(hipcc --offload-arch=gfx950).  There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the symbol is not exported
        fn.restype = res
        fn.argtypes = args
    if lib.oige_version() != 1:
        raise OigeError(f"liboige_hip.so ABI version {lib.oige_version()} != 1; rebuild")
    _lib = lib
    return lib


def check(rc: int, what: str) -> None:
    if rc != 0:
        msg = load().oige_last_error()
        raise OigeError(f"{what} failed (code {rc}): {msg.decode() if msg else ''}")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    """data_ptr of a contiguous tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_contiguous():
        raise OigeError("liboige_hip needs contiguous row-major tensors")
    return t.data_ptr()


def require_device_tensor(t: torch.Tensor, dtype: torch.dtype, shape, name: str) -> None:
    if t.dtype != dtype or tuple(t.shape) != tuple(shape) or not t.is_contiguous():
        raise OigeError(
            f"{name}: expected contiguous {dtype} tensor of shape {tuple(shape)}, got {t.dtype} {tuple(t.shape)} "
            f"contiguous={t.is_contiguous()}"
        )
    if t.device.type != "cuda":
        raise OigeError(f"{name}: liboige_hip kernels need a ROCm device tensor, got device {t.device} (no CPU fallback)")


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)


def current_stream(device: torch.device) -> int:
    """Raw handle of torch's current stream on ``device`` (what the kernels are launched on).  The raw getter costs 0.3 us
    against 4 us for ``torch.cuda.current_stream(device).cuda_stream`` (two calls per step: a quarter of a 4096-env step)."""
    if _raw_stream is not None:
        idx = device.index
        return _raw_stream(torch.cuda.current_device() if idx is None else idx)
    return torch.cuda.current_stream(device).cuda_stream


def workspace_ints(n: int) -> int:
    return int(load().oige_workspace_ints(n))


def new_workspace(n: int, device) -> torch.Tensor:
    """Zero-initialised int32 reset workspace (tile counts + group offsets)."""
    return torch.zeros(max(workspace_ints(n), 4), dtype=torch.int32, device=device)


def new_tile_counts_u8(n: int, device) -> torch.Tensor:
    """Zero-initialised byte twin of the tile counts (``*_buffers.tile_counts_u8``)."""
    return torch.zeros(max(int(load().oige_tile_counts_u8_bytes(n)), 16), dtype=torch.uint8, device=device)


def compact_static(flags: torch.Tensor, ids_out: torch.Tensor, count_out: torch.Tensor) -> None:
    """``compact`` without a workspace argument: the library's own per-GPU static scratch (at most
    ``oige_compact_static_max_envs()`` flags; allocates nothing; calls overlapping on one GPU race)."""
    check(load().oige_compact_i64_static(ptr(flags), flags.shape[0], ptr(ids_out), ptr(count_out),
                                         current_stream(flags.device)), "oige_compact_i64_static")


def compact(flags: torch.Tensor, workspace: torch.Tensor, ids_out: torch.Tensor, count_out: torch.Tensor) -> None:
    """ids_out[:count] = flags.nonzero().squeeze(-1) without a host sync."""
    n = flags.shape[0]
    check(load().oige_compact_i64(ptr(flags), n, ptr(workspace), ptr(ids_out), ptr(count_out),
                                  current_stream(flags.device)), "oige_compact_i64")
