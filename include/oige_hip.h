/* liboige_hip.so -- C ABI of the MI355X (gfx950) per-step task math.
 *
 * The reference (OmniIsaacGymEnvs, 100 % Python) has no FFI; the "operator API" this
 * library sits behind is the RLTask method contract that VecEnvRLGames.step drives
 * (/root/reference/omniisaacgymenvs/envs/vec_env_rlgames.py:79,92 and
 * tasks/base/rl_task.py:278-298).  Each entry point below names the reference method(s)
 * whose arithmetic it replaces.  omniisaacgymenvs_amd/_lib.py is the ctypes binding the
 * Python task classes call through.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to a contiguous row-major buffer owned by the caller
 *    (torch tensor .data_ptr()); the library never allocates or frees tensor memory;
 *  - floats are fp32; flags and counters are int64 (rl_task.py:165-166); quaternions wxyz;
 *  - calls are asynchronous launches on `stream` (a hipStream_t; NULL = default stream) and
 *    perform no host synchronisation;
 *  - return 0 on success, a negative OIGE_E* code otherwise; oige_last_error() gives the
 *    message for the calling thread.  Nothing throws across the ABI.
 */
#ifndef OIGE_HIP_H
#define OIGE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OIGE_ABI_VERSION 1
#define OIGE_MAX_DOF 32
#define OIGE_TILE_ENVS 64 /* one wavefront = one tile of envs; granularity of tile_reset_counts */

#define OIGE_OK 0
#define OIGE_EINVAL (-1)  /* bad argument (null pointer, size, alignment, unsupported D/S) */
#define OIGE_ELAUNCH (-2) /* HIP launch error */

typedef void* oige_stream_t; /* hipStream_t */

const char* oige_last_error(void);
int oige_version(void);

/* Measurement aid, no counterpart in the reference: a 16-byte-per-lane streaming copy of `bytes` from src to dst.
 * `streams` (1..64; bytes % (16*streams) == 0) splits both buffers into that many sub-arrays walked in lockstep, the way
 * a task kernel walks its separate (N,k) tensors.  bench.py times it to report the chip's practical read+write stream
 * rate beside the 8 TB/s spec peak.                                                                    */
int oige_stream_copy(const void* src, void* dst, int64_t bytes, int32_t streams, oige_stream_t stream);

/* ------------------------------------------------------------------ compaction
 * reset_env_ids = reset_buf.nonzero().squeeze(-1): ascending int64 ids, without the host
 * sync of len() (tasks/shared/locomotion.py:121-122, cartpole.py:112-113,
 * in_hand_manipulation.py:318-319, anymal_terrain.py:433, randomize.py:284,315).
 *
 * workspace: oige_workspace_ints(n) int32 entries, 16-byte aligned, holding
 *   [ per-tile counts of non-zero flags (64 envs per tile), padded to whole 256-env groups |
 *     exclusive prefix per 256-env group, then the total ].
 * ids_out has capacity n; *count_out (device int64) receives the number of ids.
 */
int64_t oige_num_tiles(int64_t n);
int64_t oige_workspace_ints(int64_t n);
int64_t oige_tile_counts_u8_bytes(int64_t n); /* size of oige_locomotion_buffers.tile_counts_u8 */
int oige_count_flags_i64(const int64_t* flags, int64_t n, int32_t* workspace, oige_stream_t stream);
int oige_compact_i64(const int64_t* flags, int64_t n, int32_t* workspace, int64_t* ids_out,
                     int64_t* count_out, oige_stream_t stream);

/* No counterpart in the reference: oige_compact_i64 for callers that cannot allocate (code under HIP graph capture,
 * tight serving loops, harnesses that forbid hipMalloc) and so have no workspace to pass.  Same kernels, same results;
 * the workspace is a 320 KiB array inside the library's device image (every GPU the library runs on holds its own copy
 * until the process ends).  The call allocates and frees nothing.  n <= oige_compact_static_max_envs() (4 194 304);
 * larger n is rejected with OIGE_EINVAL.  All calls on one GPU share that GPU's array: calls that overlap on one GPU
 * (two streams, two threads) race -- serialise them, or use oige_compact_i64 with separate workspaces.
 * `stream` (when not NULL) must belong to the CURRENT device -- the scratch address is per device and is looked up for the
 * current one; a stream of another GPU is rejected with OIGE_EINVAL.  The first call on a device performs that look-up
 * (hipGetSymbolAddress): make it outside stream capture (any call will do, n = 0 included); later calls are capturable
 * and replayable (tests/test_zz_no_alloc.py).                                                                       */
int64_t oige_compact_static_max_envs(void);
int oige_compact_i64_static(const int64_t* flags, int64_t n, int64_t* ids_out, int64_t* count_out, oige_stream_t stream);

/* ------------------------------------------------------------------ locomotion (Ant, Humanoid)
 * Scalars: LocomotionTask.update_config (tasks/shared/locomotion.py:50-65) and
 * post_reset (:162-184); tables: tasks/ant.py:89-93, tasks/humanoid.py:91-121.
 * The per-env-constant (N,k) tensors of the reference (targets, inv_start_rot,
 * basis_vec0/1, locomotion.py:168-177) are passed once as scalars.                      */
typedef struct oige_locomotion_params {
  int32_t num_dof;         /* D */
  int32_t num_sensors;     /* S */
  int32_t limit_cost_kind; /* 0: Ant, ant.py:107-109; 1: Humanoid, humanoid.py:135-141 */
  int32_t reserved0;
  float dt;                /* 1/60, locomotion.py:178 */
  float dof_vel_scale;
  float contact_force_scale;
  float angular_velocity_scale;
  float heading_weight;
  float up_weight;
  float actions_cost_scale;
  float energy_cost_scale;
  float joints_at_limit_cost_scale;
  float death_cost;
  float termination_height;
  float alive_reward_scale;
  float max_episode_length; /* compared as float: is_done(..., max_episode_length: float) */
  float power_scale;
  float clip_actions;       /* used only when OIGE_PRE_CLAMP_ACTIONS is set */
  float reserved1;
  float target[3];
  float inv_start_rot[4];
  float basis_vec0[3];
  float basis_vec1[3];
  float reserved2[3];
  float dof_limits_lower[OIGE_MAX_DOF];
  float dof_limits_upper[OIGE_MAX_DOF];
  float joint_gears[OIGE_MAX_DOF];
  float motor_effort_ratio[OIGE_MAX_DOF];
} oige_locomotion_params;

/* What the ArticulationView getters return (locomotion.py:76-84); read-only. */
typedef struct oige_locomotion_state {
  const float* torso_position;       /* (N,3) world */
  const float* torso_rotation;       /* (N,4) wxyz  */
  const float* velocities;           /* (N,6) lin, ang */
  const float* dof_pos;              /* (N,D) */
  const float* dof_vel;              /* (N,D) */
  const float* sensor_force_torques; /* (N,S,6) */
} oige_locomotion_state;

/* RLTask buffers (rl_task.py:162-166) + LocomotionTask state (locomotion.py:169-184). */
typedef struct oige_locomotion_buffers {
  float* obs_buf;          /* (N, 12+3D+6S) */
  float* rew_buf;          /* (N,) */
  int64_t* reset_buf;      /* (N,) */
  int64_t* progress_buf;   /* (N,) */
  float* potentials;       /* (N,) */
  float* prev_potentials;  /* (N,) */
  float* up_vec;           /* (N,3) or NULL (written back as locomotion.py:90) */
  float* heading_vec;      /* (N,3) or NULL (locomotion.py:91) */
  const float* actions;    /* (N,D) the task's self.actions */
  int32_t* reset_workspace; /* oige_workspace_ints(N) int32, zero-initialised once by the caller,
                               16-byte aligned.  The DONE stage leaves the per-tile counts of
                               non-zero reset flags here for the next pre_physics (may be NULL for
                               post_physics; required by pre_physics) */
  uint8_t* tile_counts_u8;  /* optional (NULL = not used): oige_tile_counts_u8_bytes(N) bytes, 16-byte aligned.  The same
                               per-tile counts as one byte each (a count is at most 64), written wherever the int32 ones
                               are.  With it pre_physics derives each group's offset in the id list from the counts in
                               front of it inside its own launch up to 4096 groups (1 M envs) instead of 2048: no scan
                               launch between post_physics and pre_physics.  Pass the same buffer to post_physics and pre_physics (with
                               OIGE_PRE_COUNTS_VALID the pre call trusts what the post call left in it).  No counterpart in the
                               reference. */
  double* stats_slots;      /* optional (NULL = not kept): 4 doubles per 64-env tile, zeroed once by the caller.  A fused
                               post_physics call (OBS + METRICS + DONE) adds the tile's {sum reward, done envs, sum of their
                               episode lengths} to its slot: the rank's episode statistics for oige_episode_stats_fold
                               without a launch or a byte of extra traffic per step (multi-GPU runs). */
} oige_locomotion_buffers;

/* stages of RLTask.post_physics_step (rl_task.py:289-295), any subset, executed in this order */
#define OIGE_STAGE_PROGRESS 1u /* progress_buf[:] += 1                      rl_task.py:289 */
#define OIGE_STAGE_OBS 2u      /* get_observations      locomotion.py:75-115, jit :225-284 */
#define OIGE_STAGE_METRICS 4u  /* calculate_metrics     locomotion.py:190-206, jit :296-345,
                                  + get_dof_at_limit_cost ant.py:107 / humanoid.py:135      */
#define OIGE_STAGE_DONE 8u     /* is_done               locomotion.py:208-211, jit :288-292 */
#define OIGE_STAGE_ALL 15u

int oige_locomotion_post_physics(const oige_locomotion_params* params, const oige_locomotion_state* state,
                                 const oige_locomotion_buffers* buffers, int64_t num_envs, uint32_t stages,
                                 oige_stream_t stream);

/* reset_idx inputs/outputs (locomotion.py:133-158).  Outputs are the compacted rows the view
 * setters take (set_joint_positions/velocities, set_world_poses, set_velocities with
 * indices=env_ids), in ascending env order; capacity N rows each.                          */
typedef struct oige_locomotion_reset_io {
  const float* initial_root_pos; /* (N,3) */
  const float* initial_root_rot; /* (N,4) */
  const float* initial_dof_pos;  /* (N,D) */
  const float* u_pos;            /* NULL, or (n,D) pre-drawn U[0,1): 1st draw, locomotion.py:137 */
  const float* u_vel;            /* NULL, or (n,D) pre-drawn U[0,1): 2nd draw, locomotion.py:139 */
  uint64_t seed;                 /* in-kernel Philox4x32-10 key when u_pos/u_vel are NULL */
  uint64_t step;                 /* mixed into the Philox counter */
  int64_t* env_ids;              /* out (N,)  ascending ids == reset_buf.nonzero().squeeze(-1) */
  int64_t* count;                /* out device scalar: number of ids */
  float* dof_pos_out;            /* out (n,D) */
  float* dof_vel_out;            /* out (n,D) */
  float* root_pos_out;           /* out (n,3) */
  float* root_rot_out;           /* out (n,4) */
  float* root_vel_out;           /* out (n,6) zeros */
  const uint64_t* step_ptr;      /* NULL, or a device scalar read INSTEAD of `step`: a step captured into a HIP graph bakes
                                    its by-value arguments, so the replaying caller advances this counter on the stream */
} oige_locomotion_reset_io;

#define OIGE_PRE_CLAMP_ACTIONS 1u /* fold VecEnvRLGames' torch.clamp(actions, +-clip) (vec_env_rlgames.py:77) */
#define OIGE_PRE_COUNTS_VALID 2u  /* the workspaces' tile counts already match the flag buffers (skip the recount): true right
                                   * after a *_post_physics call with OIGE_STAGE_DONE on the same buffers and nothing written
                                   * to them since.  Ingenuity / Crazyflie: the DONE stage also leaves target_workspace's
                                   * counts (reset | progress % 500 == 0), so progress_buf must be untouched as well.
                                   * "The same buffers" includes the workspaces themselves: that post call must have been given
                                   * the SAME non-NULL reset_workspace (and target_workspace) and the SAME tile_counts_u8
                                   * table(s) this pre call gets -- the post kernels leave counts only in the tables they
                                   * are handed (a NULL there skips them), and a table first supplied to the pre call holds
                                   * whatever it held before: stale ranks, wrong ids, no diagnostic.  When in doubt leave the
                                   * flag off: the recount is one 5 us launch.                                          */
#define OIGE_PRE_OFFSETS_VALID 4u /* in-hand tasks: the group offsets are valid too (oige_hand_post_physics[_dr] with
                                   * OIGE_STAGE_METRICS scans both workspaces in its last launch): implies COUNTS_VALID,
                                   * same condition; pre_physics then starts at its kernel */

/* get_dof_at_limit_cost() as the reference's public method returns it (tasks/ant.py:102-109: jit sum(obs[:, 12:12+D] > 0.99)
 * -> (N,) int64; tasks/humanoid.py:130-141: scaled, motor-effort-weighted excess over 0.98 -> (N,) fp32), from obs_buf.
 * cost_out: int64[N] when params->limit_cost_kind == 0, float[N] when 1.  (Inside oige_locomotion_post_physics the METRICS
 * stage evaluates the same function in registers; this entry point exists for callers of the method itself.)            */
int oige_locomotion_dof_at_limit_cost(const oige_locomotion_params* params, const float* obs_buf, void* cost_out,
                                      int64_t num_envs, oige_stream_t stream);

/* LocomotionTask.pre_physics_step (locomotion.py:117-131): compaction of reset_buf, reset_idx of
 * the flagged envs, actions clone, forces = actions * joint_gears * power_scale.
 * actions_out may be NULL (no clone) or alias actions_in.                                   */
int oige_locomotion_pre_physics(const oige_locomotion_params* params, const float* actions_in,
                                const oige_locomotion_buffers* buffers, const oige_locomotion_reset_io* reset_io,
                                float* actions_out, float* forces_out, int64_t num_envs, uint32_t flags,
                                oige_stream_t stream);

/* LocomotionTask.reset_idx(env_ids) with caller-provided ids (locomotion.py:133-158; used by
 * post_reset :187-188).  reset_io->env_ids / ->count are INPUTS here (count: device scalar).     */
int oige_locomotion_reset_idx(const oige_locomotion_params* params, const oige_locomotion_buffers* buffers,
                              const oige_locomotion_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                              oige_stream_t stream);

/* ------------------------------------------------------------------ ShadowHand / AllegroHand in-hand manipulation
 * tasks/shared/in_hand_manipulation.py (reward, resets, action scaling), tasks/shadow_hand.py and
 * tasks/allegro_hand.py (observation packers).  Scalars: update_config (:63-97), ShadowHand.yaml,
 * AllegroHand.yaml.  obs_type also selects the hand: the ALLEGRO values mean 16 DOFs, all actuated
 * (views/allegro_hand_view.py:53), no fingertip / sensor inputs, no states_buf; every (N,24) / (N,20)
 * array below is then (N,16) and the u_reset table is (ne, 37).                                       */
#define OIGE_HAND_OBS_OPENAI 0 /* 42 cols, shadow_hand.py:162-165 */
#define OIGE_HAND_OBS_FULL 2   /* 157 cols, shadow_hand.py:198-217 */
#define OIGE_HAND_OBS_ALLEGRO_FULL_NO_VEL 3 /* 50 cols, allegro_hand.py:118-128 */
#define OIGE_HAND_OBS_ALLEGRO_FULL 4        /* 72 cols, allegro_hand.py:130-144 */
#define OIGE_ALLEGRO_DOFS 16
#define OIGE_HAND_DOFS 24
#define OIGE_HAND_ACTIONS 20
#define OIGE_HAND_FINGERTIPS 5
#define OIGE_HAND_STATES 187   /* asymmetric full_state states_buf, shadow_hand.py:220-262 */

typedef struct oige_hand_params {
  int32_t obs_type;                   /* OIGE_HAND_OBS_* */
  int32_t asymmetric_obs;             /* also fill states_buf (N,187) */
  int32_t max_consecutive_successes;
  int32_t use_relative_control;
  float max_episode_length;           /* float, as compute_hand_reward's signature (:464) */
  float dist_reward_scale;
  float rot_reward_scale;
  float rot_eps;
  float action_penalty_scale;
  float success_tolerance;
  float reach_goal_bonus;
  float fall_dist;
  float fall_penalty;
  float av_factor;
  float vel_obs_scale;
  float force_torque_obs_scale;       /* 10.0, shadow_hand.py:84 */
  float hand_dof_speed_scale;
  float dt;                           /* 1/60, in_hand_manipulation.py:92 */
  float act_moving_average;
  float reset_position_noise;
  float reset_dof_pos_noise;
  float reset_dof_vel_noise;
  float clip_actions;                 /* with OIGE_PRE_CLAMP_ACTIONS */
  float one_minus_act_moving_average; /* (float)(1.0 - actionsMovingAverage), folded in double like Python (:350) */
  float dof_speed_scale_dt;           /* (float)(dofSpeedScale * dt), folded in double like Python (:335) */
  float reserved0[3];
  float goal_displacement[3];         /* (-0.2,-0.06,0.12), :111 */
  float reserved1;
  float dof_lower[OIGE_MAX_DOF];
  float dof_upper[OIGE_MAX_DOF];
  float dof_default_pos[OIGE_MAX_DOF];
  float dof_default_vel[OIGE_MAX_DOF];
  int32_t actuated_dof_indices[OIGE_MAX_DOF]; /* 20 sorted indices, views/shadow_hand_view.py:59-84 */
} oige_hand_params;

/* View getters of shadow_hand.py:121-135 (world frame, read-only: the reference's in-place
 * `-= env_pos` on these buffers is NOT reproduced, the subtraction happens in registers). */
typedef struct oige_hand_state {
  const float* object_pos;           /* (N,3) */
  const float* object_rot;           /* (N,4) */
  const float* object_velocities;    /* (N,6) */
  const float* fingertip_pos;        /* (5N,3) */
  const float* fingertip_rot;        /* (5N,4) */
  const float* fingertip_velocities; /* (5N,6) */
  const float* hand_dof_pos;         /* (N,24) */
  const float* hand_dof_vel;         /* (N,24) */
  const float* vec_sensor;           /* (N,5,6); may be NULL unless asymmetric_obs */
  const float* env_pos;              /* (N,3) RLTask._env_pos */
} oige_hand_state;

typedef struct oige_hand_buffers {
  float* obs_buf;                /* (N,42|157) */
  float* states_buf;             /* (N,187) or NULL */
  float* rew_buf;                /* (N,) */
  int64_t* reset_buf;            /* (N,) */
  int64_t* reset_goal_buf;       /* (N,) */
  int64_t* progress_buf;         /* (N,) */
  float* successes;              /* (N,) float, accumulates int64 goal_resets (:500) */
  float* consecutive_successes;  /* (1,) */
  float* goal_pos;               /* (N,3) env-local */
  float* goal_rot;               /* (N,4) */
  const float* actions;          /* (N,20) the task's self.actions */
  float* prev_targets;           /* (N,24) */
  float* cur_targets;            /* (N,24) */
  int64_t* randomization_buf;    /* (N,) or NULL: += 1 per calculate_metrics (:297) */
  int32_t* reset_workspace;      /* oige_workspace_ints(N): tile counts of reset_buf */
  int32_t* goal_workspace;       /* oige_workspace_ints(N): tile counts of reset_goal_buf */
  float* reduce_workspace;       /* 2*oige_num_tiles(N) floats: per-tile (num_resets, sum successes*resets) */
} oige_hand_buffers;

/* post_physics_step: progress += 1 (rl_task.py:289), get_observations (shadow_hand.py:120-152),
 * calculate_metrics = compute_hand_reward (in_hand_manipulation.py:263-297, jit :457-531) including
 * the two global sums that update consecutive_successes (deterministic two-stage reduction).
 * is_done is a no-op in the reference (:374): OIGE_STAGE_DONE is accepted and ignored.        */
int oige_hand_post_physics(const oige_hand_params* params, const oige_hand_state* state,
                           const oige_hand_buffers* buffers, int64_t num_envs, uint32_t stages,
                           oige_stream_t stream);

/* Draw layout for the exact-parity path (all U(-1,1), i.e. torch_rand_float(-1, 1, ...) outputs):
 *   u_goal  (ng,4)  reset_target_pose(goal_env_ids)                      (:380)
 *   u_reset (ne,53) reset_idx(env_ids)                                   (:399)
 *   u_reset_goal (ne,4) the nested reset_target_pose(env_ids)            (:401 -> :380)
 * NULL -> in-kernel Philox keyed by (seed, step, env id, stream).                             */
typedef struct oige_hand_reset_io {
  const float* object_init_pos;  /* (N,3) env-local (:236-237) */
  const float* goal_init_pos;    /* (N,3) (:246) */
  const float* env_pos;          /* (N,3) */
  const float* u_goal;
  const float* u_reset;
  const float* u_reset_goal;
  uint64_t seed;
  uint64_t step;
  /* goal-only resets (reset_target_pose of goal_env_ids): rows for _goals.set_world_poses */
  int64_t* goal_ids;             /* out (N,) ascending == reset_goal_buf.nonzero() */
  int64_t* goal_count;           /* out device scalar */
  float* goal_world_pos_out;     /* out (ng,3) goal_pos + displacement + env_pos */
  float* goal_rot_out;           /* out (ng,4) */
  /* full resets (reset_idx of env_ids) */
  int64_t* env_ids;              /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                /* out device scalar */
  float* r_goal_world_pos_out;   /* out (ne,3) */
  float* r_goal_rot_out;         /* out (ne,4) */
  float* object_pos_out;         /* out (ne,3) world */
  float* object_rot_out;         /* out (ne,4) */
  float* object_vel_out;         /* out (ne,6) zeros */
  float* dof_pos_out;            /* out (ne,24) also the position targets (:433-436) */
  float* dof_vel_out;            /* out (ne,24) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_hand_reset_io;

/* pre_physics_step (:314-362): both compactions, reset_target_pose of goal ids, reset_idx of env
 * ids, actions clone, scale/EMA/clamp (or relative control) into cur_targets/prev_targets;
 * targets_out (N,20) = cur_targets[:, actuated] for set_joint_position_targets (:360-362).     */
int oige_hand_pre_physics(const oige_hand_params* params, const float* actions_in,
                          const oige_hand_buffers* buffers, const oige_hand_reset_io* reset_io,
                          float* actions_out, float* targets_out, int64_t num_envs, uint32_t flags,
                          oige_stream_t stream);

/* reset_idx(env_ids) / reset_target_pose(env_ids) with caller-provided ids (device count scalar in
 * reset_io->count / ->goal_count, ids in ->env_ids / ->goal_ids).                                */
int oige_hand_reset_idx(const oige_hand_params* params, const oige_hand_buffers* buffers,
                        const oige_hand_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                        oige_stream_t stream);
int oige_hand_reset_target_pose(const oige_hand_params* params, const oige_hand_buffers* buffers,
                                const oige_hand_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                oige_stream_t stream);

/* ------------------------------------------------------------------ AnymalTerrain
 * tasks/anymal_terrain.py.  Scalars: update_config (:85-155), cfg/task/AnymalTerrain.yaml;
 * terrain constants: tasks/utils/anymal_terrain_generator.py:38-60.                         */
#define OIGE_ANYMAL_DOFS 12
#define OIGE_ANYMAL_OBS 188
#define OIGE_ANYMAL_SCAN_POINTS 140  /* 14 x 10 grid, init_height_points :173-187 */
#define OIGE_ANYMAL_EPISODE_SUMS 13  /* episode_sums dict order, :68-82 */

/* indices into rew_scales (each already multiplied by dt in double, :140-141) */
enum {
  OIGE_AR_TERMINATION = 0, OIGE_AR_LIN_VEL_XY, OIGE_AR_LIN_VEL_Z, OIGE_AR_ANG_VEL_Z, OIGE_AR_ANG_VEL_XY,
  OIGE_AR_ORIENT, OIGE_AR_TORQUE, OIGE_AR_JOINT_ACC, OIGE_AR_BASE_HEIGHT, OIGE_AR_ACTION_RATE, OIGE_AR_HIP,
  OIGE_AR_FALLEN_OVER, OIGE_AR_COUNT
};

typedef struct oige_anymal_params {
  float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale, height_meas_scale;
  float action_scale, Kp, Kd;
  float max_episode_length_s;
  int32_t max_episode_length;   /* int(20 / dt + 0.5) = 1000 (:132) */
  int32_t curriculum;           /* update_terrain_level active (:370) */
  int32_t init_done;            /* (:370) */
  int32_t add_noise;            /* (:438) */
  int32_t env_rows, env_cols;   /* terrain levels x types (10 x 20) */
  int32_t hs_rows, hs_cols;     /* height_samples shape (1200 x 2000) */
  float horizontal_scale, vertical_scale, border_size, half_env_length;
  float rew_scales[OIGE_AR_COUNT];
  float cmd_lo[3], cmd_span[3]; /* x, y, heading: lower and (upper - lower) folded in double (:341-349) */
  float base_init_state[13];    /* pos, rot wxyz, lin vel, ang vel (:118-123) */
  float default_dof_pos[OIGE_ANYMAL_DOFS];
  float noise_scales[6];        /* noise_scale_vec values for cols 0:3, 3:6, 6:9, 12:24, 24:36, 36:176 (:157-171) */
  float reserved[3];
} oige_anymal_params;

/* refresh_*_state_tensors (:382-389) + contact forces (:459,:462) + terrain tables; read-only */
typedef struct oige_anymal_state {
  const float* base_pos;         /* (N,3) */
  const float* base_quat;        /* (N,4) */
  const float* base_velocities;  /* (N,6) */
  const float* dof_pos;          /* (N,12) */
  const float* dof_vel;          /* (N,12) */
  const float* knee_contact;     /* (4N,3) _knees.get_net_contact_forces */
  const float* base_contact;     /* (N,3)  _base.get_net_contact_forces */
  const int16_t* height_samples; /* (hs_rows, hs_cols) int16 (:196-198) */
  const float* terrain_origins;  /* (env_rows, env_cols, 3) (:241) */
  const int16_t* height_min;     /* NULL, or (hs_rows, hs_cols) from oige_anymal_height_min(height_samples): one gather per
                                    scan point instead of the two of :598-600 */
} oige_anymal_state;

typedef struct oige_anymal_buffers {
  float* obs_buf;            /* (N,188) */
  float* rew_buf;            /* (N,) */
  uint8_t* reset_buf;        /* (N,) BOOL: the reference rebinds reset_buf to a bool tensor (:465) */
  uint8_t* has_fallen;       /* (N,) bool (:462) */
  int64_t* timeout_buf;      /* (N,) (:453) */
  int64_t* progress_buf;     /* (N,) */
  float* commands;           /* (N,4) x vel, y vel, yaw vel, heading (:276) */
  const float* actions;      /* (N,12) */
  float* last_actions;       /* (N,12) */
  float* last_dof_vel;       /* (N,12) */
  const float* torques;      /* (N,12) last sub-step's torques (:405) */
  float* feet_air_time;      /* (N,4) */
  float* episode_sums;       /* (13,N) row k = episode_sums[key k] */
  int64_t* terrain_levels;   /* (N,) */
  const int64_t* terrain_types; /* (N,) */
  float* env_origins;        /* (N,3) */
  float* extras;             /* (14,) device: 13 x rew_<key> means / max_episode_length_s, terrain_level mean (:361-367) */
  int32_t* reset_workspace;  /* oige_workspace_ints(N) */
  float* reduce_workspace;   /* (16 + 512) * oige_num_tiles(N) + 4 floats, 16-byte aligned: per-tile partial sums, then
                                the 8-float pose record per env that the head kernel hands to the scan kernel */
  uint8_t* tile_counts_u8;   /* optional: oige_tile_counts_u8_bytes(N) bytes, 16-byte aligned (as oige_locomotion_buffers.
                                tile_counts_u8).  With it, up to 4096 groups, the step runs without its scan + norm launch: the
                                head kernel's waves derive their offset in the id list and the command norm (:376) themselves */
} oige_anymal_buffers;

/* reset rows for the view setters (:334-339), compacted in ascending env order */
typedef struct oige_anymal_reset_io {
  const float* u_dof_offset; /* NULL or (n,12) U[0,1): 1st draw (:321) */
  const float* u_dof_vel;    /* (n,12) 2nd draw (:322) */
  const float* u_xy;         /* (n,2) 3rd draw (:330) */
  const float* u_cmd;        /* (n,3) 4th-6th draws: x, y, heading (:341-349) */
  const float* noise_u;      /* NULL or (N,188) U[0,1) replacing torch.rand_like (:439) */
  const uint8_t* reset_mask; /* oige_anymal_reset_idx only: (N,) 1 where env is in env_ids */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;          /* out (N,) == reset_buf.nonzero().flatten() (:433) */
  int64_t* count;            /* out device scalar */
  float* base_pos_out;       /* out (n,3) */
  float* base_quat_out;      /* out (n,4) */
  float* base_vel_out;       /* out (n,6) */
  float* dof_pos_out;        /* out (n,12) */
  float* dof_vel_out;        /* out (n,12) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
  /* push_robots (:446-450) on the device, so that no step differs from another on the host (HIP-graph capture):
   * oige_anymal_post_physics, before anything reads the velocities, sets push_velocities[:, 0:2] = U(-1, 1) for every env
   * when (step + push_offset) % push_interval == 0 -- `step` as above (by value or *step_ptr), push_offset =
   * common_step_counter - step on the host (:418-420 counts post_physics_step calls) -- and writes *push_count = N on such
   * a step, 0 otherwise: the (values, count) hand-over for `set_velocities`.  Draws: Philox stream 7, two per env.
   * push_velocities NULL or push_interval <= 0: no push (the caller does it, the reference's way).               */
  float* push_velocities;   /* NULL or (N,6): the view's base velocities, updated in place like the reference does */
  int64_t* push_count;      /* out device scalar (required with push_velocities) */
  int64_t push_interval;    /* :131-133 int(pushInterval_s / dt + 0.5) */
  int64_t push_offset;
} oige_anymal_reset_io;

/* get_heights (:598-600) reads height_samples[px, py] and height_samples[px + 1, py + 1] and keeps the smaller: that
 * minimum depends on the map alone, so it is tabulated once per map: out[r, c] = min(h[r, c], h[r + 1, c + 1]) for
 * r <= rows - 2, c <= cols - 2 (the only cells the clipped indices can address), h[r, c] elsewhere.  Integer min, then
 * the same `* vertical_scale`: bit-identical heights, half the gathers (the scan kernel's address unit was 80 % busy). */
int oige_anymal_height_min(const int16_t* height_samples, int32_t hs_rows, int32_t hs_cols, int16_t* out, oige_stream_t stream);

/* One decimation sub-step of pre_physics_step (:398-403):
 * torques = clip(Kp*(action_scale*actions + default_dof_pos - dof_pos) - Kd*dof_vel, -80, 80)   */
int oige_anymal_torques(const oige_anymal_params* params, const float* actions, const float* dof_pos,
                        const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream);
/* The same with `self.actions = actions.clone()` (:390) folded in: actions_out (N,12) receives the copy the rest of the
 * step reads (the first decimation sub-step's launch; the following ones use oige_anymal_torques on the copy). */
int oige_anymal_torques_clone(const oige_anymal_params* params, const float* actions, float* actions_out, const float* dof_pos,
                              const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream);

/* post_physics_step (:409-444) in the reference's order: progress += 1; body-frame velocities,
 * projected gravity, heading command (:421-427); check_termination (:452-466); calculate_metrics
 * (:468-533); the in-step reset_idx of the flagged envs (:433-435, :318-380); get_observations with
 * the height scan (:535-604); uniform observation noise (:438-439); last_* copies (:441-442).   */
int oige_anymal_post_physics(const oige_anymal_params* params, const oige_anymal_state* state,
                             const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io,
                             int64_t num_envs, oige_stream_t stream);

/* The same step, stage by stage: the reference's post_physics_step (:409-444) calls check_termination (:452-466),
 * calculate_metrics (:468-533) and get_observations (:535-552) as methods of their own, after computing the body-frame
 * quantities inline (:421-427) and keeping them as attributes.  `stages` is any subset of
 *   PROGRESS     progress_buf += 1                                                          (:410)
 *   FRAME        base_lin_vel / base_ang_vel / projected_gravity -> `frame`, commands[:, 2] (:421-427)
 *   TERMINATION  timeout_buf, has_fallen, reset_buf (bool), the per-tile reset counts       (:452-466)
 *   METRICS      rew_buf, episode_sums; reads `frame`, reset_buf, has_fallen, timeout_buf   (:468-533)
 *   OBS          obs_buf WITHOUT noise (heights from the current view state); reads `frame` (:535-552)
 * executed in that order; OIGE_ANYMAL_STAGE_ALL is oige_anymal_post_physics (which also runs the in-step reset_idx, the
 * observation noise and the last_* copies: RESET / NOISE / LAST exist only inside the fused step; on their own they are
 * oige_anymal_reset_idx and the reference's own two elementwise lines, :438-442).  `frame` is required for FRAME / METRICS /
 * OBS and optional for ALL (NULL: the three tensors are not materialised).  Split == fused bit for bit
 * (tests/test_anymal_gpu.py::test_split_stage_methods_match_fused).                                               */
#define OIGE_ANYMAL_STAGE_PROGRESS 1u
#define OIGE_ANYMAL_STAGE_FRAME 2u
#define OIGE_ANYMAL_STAGE_TERMINATION 4u
#define OIGE_ANYMAL_STAGE_METRICS 8u
#define OIGE_ANYMAL_STAGE_RESET 16u
#define OIGE_ANYMAL_STAGE_OBS 32u
#define OIGE_ANYMAL_STAGE_NOISE 64u
#define OIGE_ANYMAL_STAGE_LAST 128u
#define OIGE_ANYMAL_STAGE_ALL 255u

typedef struct oige_anymal_frame { /* (N,3) each: self.base_lin_vel, self.base_ang_vel, self.projected_gravity (:422-424) */
  float* base_lin_vel;
  float* base_ang_vel;
  float* projected_gravity;
} oige_anymal_frame;

int oige_anymal_post_physics_stages(const oige_anymal_params* params, const oige_anymal_state* state,
                                    const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io,
                                    const oige_anymal_frame* frame, int64_t num_envs, uint32_t stages, oige_stream_t stream);

/* update_terrain_level(env_ids) on its own (:369-380; inside a step reset_idx runs it in the head kernel):
 * distance of each id's root from its env origin against the whole-matrix norm of ALL the ids' xy commands (:376),
 * level -= / += / clip / % env_rows, env_origins[ids] = terrain_origins[level, type].  No-op unless init_done and curriculum. */
int oige_anymal_update_terrain_level(const oige_anymal_params* params, const float* root_pos /* (N,3) */,
                                     const float* terrain_origins, const float* commands /* (N,4) */,
                                     int64_t* terrain_levels, const int64_t* terrain_types, float* env_origins,
                                     const int64_t* env_ids, int64_t num_ids, int64_t num_envs, oige_stream_t stream);

/* reset_idx(env_ids) on its own (:318-367; post_reset :314-315 calls it with every env and
 * init_done = 0): env_ids are given as the byte mask reset_io->reset_mask; same outputs.          */
int oige_anymal_reset_idx(const oige_anymal_params* params, const oige_anymal_state* state,
                          const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io,
                          int64_t num_envs, oige_stream_t stream);

/* ------------------------------------------------------------------ Anymal (flat ground)
 * tasks/anymal.py.  Scalars: update_config (:51-101), cfg/task/Anymal.yaml.          */
#define OIGE_ANYMAL_FLAT_OBS 48

typedef struct oige_anymal_flat_params {
  float lin_vel_scale, ang_vel_scale, dof_pos_scale, dof_vel_scale;
  float action_scale;
  float dt;                    /* 1/60 (:90) */
  float base_threshold;        /* 0.51, is_base_below_threshold(threshold=0.51, ground_heights=0.0) (:307) */
  float clip_actions;          /* with OIGE_PRE_CLAMP_ACTIONS */
  int32_t max_episode_length;  /* int(episodeLength_s / dt + 0.5) (:92) */
  int32_t reserved0;
  float rew_scales[6];         /* lin_vel_xy, ang_vel_z, lin_vel_z, joint_acc, action_rate, cosmetic, each x dt (:96-97) */
  float cmd_lo[3], cmd_span[3];/* x, y, yaw: lower and (upper - lower) folded in double (:217-225) */
  float default_dof_pos[OIGE_ANYMAL_DOFS];
  float dof_lower[OIGE_ANYMAL_DOFS];
  float dof_upper[OIGE_ANYMAL_DOFS];
} oige_anymal_flat_params;

typedef struct oige_anymal_flat_state { /* view getters of :144-147 / :275-278; read-only */
  const float* base_pos;    /* (N,3) */
  const float* base_rot;    /* (N,4) */
  const float* velocities;  /* (N,6) */
  const float* dof_pos;     /* (N,12) */
  const float* dof_vel;     /* (N,12) */
} oige_anymal_flat_state;

typedef struct oige_anymal_flat_buffers {
  float* obs_buf;           /* (N,48) */
  float* rew_buf;           /* (N,) */
  int64_t* reset_buf;       /* (N,) */
  int64_t* progress_buf;    /* (N,) */
  float* commands;          /* (N,3) x, y, yaw (:250-253) */
  const float* actions;     /* (N,12) the task's self.actions */
  float* last_actions;      /* (N,12) */
  float* last_dof_vel;      /* (N,12) */
  float* current_targets;   /* (N,12); pre_physics / reset_idx only */
  int32_t* reset_workspace; /* oige_workspace_ints(N): tile counts of reset_buf */
  uint8_t* tile_counts_u8;  /* optional: as oige_locomotion_buffers.tile_counts_u8 */
} oige_anymal_flat_buffers;

typedef struct oige_anymal_flat_reset_io {
  const float* initial_root_pos; /* (N,3) */
  const float* initial_root_rot; /* (N,4) */
  const float* u_vel;            /* NULL, or (n,12) pre-drawn U[0,1): 1st draw (:199) */
  const float* u_cmd;            /* NULL, or (n,3): 2nd-4th draws x, y, yaw (:217-225) */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;              /* out (N,) == reset_buf.nonzero().squeeze(-1) (in for oige_anymal_flat_reset_idx) */
  int64_t* count;                /* out device scalar (in for oige_anymal_flat_reset_idx) */
  float* dof_pos_out;            /* out (n,12) */
  float* dof_vel_out;            /* out (n,12) */
  float* root_pos_out;           /* out (n,3) */
  float* root_rot_out;           /* out (n,4) */
  float* root_vel_out;           /* out (n,6) zeros */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_anymal_flat_reset_io;

/* RLTask.post_physics_step (rl_task.py:289-295) for AnymalTask: progress += 1, get_observations (:143-178),
 * calculate_metrics (:274-309, refreshes last_actions / last_dof_vel), is_done (:311-314).  `stages` as for locomotion. */
int oige_anymal_flat_post_physics(const oige_anymal_flat_params* params, const oige_anymal_flat_state* state,
                                  const oige_anymal_flat_buffers* buffers, int64_t num_envs, uint32_t stages,
                                  oige_stream_t stream);
/* pre_physics_step (:180-194): compaction of reset_buf, reset_idx of the flagged envs, actions clone, position targets */
int oige_anymal_flat_pre_physics(const oige_anymal_flat_params* params, const float* actions_in,
                                 const oige_anymal_flat_buffers* buffers, const oige_anymal_flat_reset_io* reset_io,
                                 float* actions_out, int64_t num_envs, uint32_t flags, oige_stream_t stream);
/* reset_idx(env_ids) (:196-231) with caller-provided ids and, optionally, torch-drawn noise */
int oige_anymal_flat_reset_idx(const oige_anymal_flat_params* params, const oige_anymal_flat_buffers* buffers,
                               const oige_anymal_flat_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                               oige_stream_t stream);

/* ------------------------------------------------------------------ observation / action noise
 * utils/domain_randomization/randomize.py:283-410 (the part of Randomizer that is task math).      */
#define OIGE_NOISE_ADDITIVE 0
#define OIGE_NOISE_SCALING 1
#define OIGE_NOISE_GAUSSIAN 0   /* "gaussian" / "normal": torch.normal(mean=p[0], std=p[1]) (:345-351) */
#define OIGE_NOISE_UNIFORM 1    /* (p[1] - p[0]) * rand + p[0]                              (:352-355) */
#define OIGE_NOISE_LOGUNIFORM 2 /* exp((log p[1] - log p[0]) * rand + log p[0])            (:356-361) */

typedef struct oige_noise_term {
  int32_t enabled;            /* the "on_reset" / "on_interval" key is present (:288,:298) */
  int32_t operation;          /* OIGE_NOISE_ADDITIVE | OIGE_NOISE_SCALING */
  int32_t distribution;       /* OIGE_NOISE_* */
  int32_t frequency_interval; /* on_interval only (:300) */
  float p0;                   /* gaussian: mean; uniform: p[0]; loguniform: log(p[0]) */
  float p1;                   /* gaussian: std;  uniform: p[1] - p[0]; loguniform: log(p[1]) - log(p[0]) (folded in double, like Python) */
} oige_noise_term;

typedef struct oige_noise_args {
  float* buffer;                 /* (N,W) observations or actions, noised IN PLACE (:289,:305) */
  const int64_t* reset_buf;      /* (N,) the task's post-is_done reset_buf (vec_env_rlgames.py:72-75,94-97) */
  int32_t* counter;              /* (N,) int32 _observations/_actions_counter_buffer (:242-244) */
  float* correlated_noise;       /* (N,W) _observations/_actions_correlated_noise (:245-247); NULL if no on_reset */
  float* clamped_out;            /* NULL, or (N,W): clamp(buffer, -clip, clip), the copy VecEnvRLGames hands the policy (:40) */
  const float* reset_samples;    /* NULL -> in-kernel Philox; else (k,W) rows in reset_ids order (:381-400) */
  const float* interval_samples; /* NULL -> in-kernel Philox; else (m,W) rows in randomize_ids order (:346-361) */
  int32_t* reset_workspace;      /* oige_workspace_ints(N) each; needed only with pre-drawn samples: */
  int32_t* interval_workspace;   /*   oige_dr_noise_count leaves counts, group offsets and the totals k, m there */
  uint64_t seed;
  uint64_t step;
  int32_t stream_id;             /* 0 observations, 1 actions: separates the Philox streams */
  int32_t width;                 /* W */
  float clip;
  float reserved0;
  oige_noise_term on_reset;
  oige_noise_term on_interval;
  const uint64_t* step_ptr;      /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay of the wrapper's step) */
} oige_noise_args;

/* First half of the exact-replay flow: decides reset ids (:284) and randomize ids (:299-303) WITHOUT
 * touching the counter and leaves their compaction ranks in the two workspaces; the totals k and m are
 * workspace[oige_workspace_ints(N) - 1] (the caller reads them to size its torch draws, as the
 * reference's len() does).                                                                       */
int oige_dr_noise_count(const oige_noise_args* args, int64_t num_envs, oige_stream_t stream);

/* apply_observations_randomization / apply_actions_randomization (:283-342): counter update, correlated
 * noise refresh of the reset envs + application to all envs, uncorrelated noise of the randomize ids,
 * and optionally the clamped copy.  One launch.                                                   */
int oige_dr_noise_apply(const oige_noise_args* args, int64_t num_envs, oige_stream_t stream);

/* The in-hand kernels can apply the noise themselves (the wrapper's DR passes folded into the task
 * kernels), in-kernel Philox samples only, values bit-identical to the separate oige_dr_noise_apply launch:
 *   _post_physics_dr: vec_env_rlgames.py:94-97 -- observation noise with the reset_buf this very step produced, while the
 *     obs tile streams out; obs_noise->buffer must be buffers->obs_buf (left noised, unclamped, as the reference leaves
 *     it), ->clamped_out receives what _process_data (:40) hands the policy.  ShadowHand only (AllegroHand.yaml has no DR).
 *   _pre_physics_dr: vec_env_rlgames.py:72-77 -- action noise with the reset_buf as it stands on entry, while the actions
 *     are read; action_noise->buffer must be actions_inout (left noised, unclamped); with OIGE_PRE_CLAMP_ACTIONS
 *     actions_out is the clamped clone the task keeps.  ->clamped_out must be NULL.                                      */
int oige_hand_post_physics_dr(const oige_hand_params* params, const oige_hand_state* state,
                              const oige_hand_buffers* buffers, const oige_noise_args* obs_noise,
                              int64_t num_envs, uint32_t stages, oige_stream_t stream);
int oige_hand_pre_physics_dr(const oige_hand_params* params, float* actions_inout,
                             const oige_hand_buffers* buffers, const oige_hand_reset_io* reset_io,
                             const oige_noise_args* action_noise, float* actions_out, float* targets_out,
                             int64_t num_envs, uint32_t flags, oige_stream_t stream);

/* ------------------------------------------------------------------ Quadcopter
 * tasks/quadcopter.py.  Scalars: cfg/task/Quadcopter.yaml:7-17, quadcopter.py:46-55,153-161.                  */
#define OIGE_QUAD_DOFS 8
#define OIGE_QUAD_OBS 21
#define OIGE_QUAD_ACTIONS 12

typedef struct oige_quadcopter_params {
  float dof_rate;             /* (float)(dt * 8 pi): self.dt * dof_action_speed_scale, folded in double as Python does (:153-154) */
  float thrust_rate;          /* (float)(dt * 100) (:159-160) */
  float max_thrust;           /* 2.0 (:53-55) */
  float clip_actions;         /* with OIGE_PRE_CLAMP_ACTIONS */
  int32_t max_episode_length; /* 500 */
  int32_t reserved0[3];
  float dof_lower[OIGE_QUAD_DOFS];
  float dof_upper[OIGE_QUAD_DOFS];
} oige_quadcopter_params;

typedef struct oige_quadcopter_state {  /* view getters of get_observations (:125-127), world frame, read-only */
  const float* root_pos;        /* (N,3) */
  const float* root_rot;        /* (N,4) wxyz */
  const float* root_velocities; /* (N,6) */
  const float* dof_pos;         /* (N,8) */
  const float* env_pos;         /* (N,3) RLTask._env_pos */
} oige_quadcopter_state;

typedef struct oige_quadcopter_buffers {
  float* obs_buf;                /* (N,21) */
  float* rew_buf;                /* (N,) */
  int64_t* reset_buf;            /* (N,) */
  int64_t* progress_buf;         /* (N,) */
  const float* target_positions; /* (N,3) (:190-191) */
  float* dof_position_targets;   /* (N,8) */
  float* thrusts;                /* (N,4) */
  float* forces;                 /* (N,4,3): only the z components are written, all three zeroed on reset (:163-170) */
  int32_t* reset_workspace;      /* as oige_locomotion_buffers.reset_workspace */
  uint8_t* tile_counts_u8;  /* optional: as oige_locomotion_buffers.tile_counts_u8 */
} oige_quadcopter_buffers;

/* post_physics_step: progress += 1, get_observations (:124-141), calculate_metrics (:229-251), is_done (:253-261, NOT
 * sticky).  `stages` as for locomotion.                                                                         */
int oige_quadcopter_post_physics(const oige_quadcopter_params* params, const oige_quadcopter_state* state,
                                 const oige_quadcopter_buffers* buffers, int64_t num_envs, uint32_t stages,
                                 oige_stream_t stream);

typedef struct oige_quadcopter_reset_io {
  const float* initial_root_pos; /* (N,3) (:197) */
  const float* initial_root_rot; /* (N,4) */
  const float* u_dof;            /* NULL, or (n,8) pre-drawn U[0,1): the draw of :206 */
  const float* u_xyz;            /* NULL, or (n,3): the three (n,1) draws of :210-212, one per column */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;              /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                /* out device scalar */
  float* dof_pos_out;            /* out (n,8) */
  float* dof_vel_out;            /* out (n,8) zeros */
  float* root_pos_out;           /* out (n,3) */
  float* root_rot_out;           /* out (n,4) */
  float* root_vel_out;           /* out (n,6) zeros */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_quadcopter_reset_io;

/* pre_physics_step (:143-175): compaction, reset_idx of the flagged envs, DOF-target and thrust integration + clamps, z
 * forces; reset envs end with thrusts = forces = 0 and targets = their re-drawn DOF positions.  The reference keeps no
 * clone of the actions on the task, so there is no actions_out.                                                   */
int oige_quadcopter_pre_physics(const oige_quadcopter_params* params, const float* actions_in,
                                const oige_quadcopter_buffers* buffers, const oige_quadcopter_reset_io* reset_io,
                                int64_t num_envs, uint32_t flags, oige_stream_t stream);

/* reset_idx(env_ids) with caller-provided ids (:203-227).                                                         */
int oige_quadcopter_reset_idx(const oige_quadcopter_params* params, const oige_quadcopter_buffers* buffers,
                              const oige_quadcopter_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                              oige_stream_t stream);

/* ------------------------------------------------------------------ BallBalance
 * tasks/ball_balance.py.  Scalars: cfg/task/BallBalance.yaml:7-13, ball_balance.py:58-71.                      */
#define OIGE_BBOT_OBS 24
#define OIGE_BBOT_MAX_DOFS 8

typedef struct oige_ball_balance_params {
  float rate;                   /* (float)(dt * actionSpeedScale), folded in double as Python does (:196-198) */
  float ball_low;               /* (float)(ball_radius * 1.5) (:305) */
  float clip_actions;           /* with OIGE_PRE_CLAMP_ACTIONS */
  int32_t max_episode_length;   /* 600 */
  int32_t num_dof;              /* row width of dof_pos / dof_vel / dof_position_targets (3..8) */
  int32_t actuated_dof_indices[3]; /* lower_leg0..2 (:277-280) */
  float dof_lower[OIGE_BBOT_MAX_DOFS];
  float dof_upper[OIGE_BBOT_MAX_DOFS];
} oige_ball_balance_params;

typedef struct oige_ball_balance_state {  /* view getters of get_observations (:160-169), world frame, read-only */
  const float* ball_pos;        /* (N,3) */
  const float* ball_velocities; /* (N,6) */
  const float* dof_pos;         /* (N,num_dof) */
  const float* dof_vel;         /* (N,num_dof) */
  const float* sensors;         /* (N,3,6) measured joint forces of upper_leg0..2 */
  const float* env_pos;         /* (N,3) */
} oige_ball_balance_state;

typedef struct oige_ball_balance_buffers {
  float* obs_buf;               /* (N,24) */
  float* rew_buf;               /* (N,) */
  int64_t* reset_buf;           /* (N,) */
  int64_t* progress_buf;        /* (N,) */
  float* dof_position_targets;  /* (N,num_dof) */
  int32_t* reset_workspace;     /* as oige_locomotion_buffers.reset_workspace */
  uint8_t* tile_counts_u8;  /* optional: as oige_locomotion_buffers.tile_counts_u8 */
} oige_ball_balance_buffers;

/* post_physics_step: progress += 1, get_observations (:159-185), calculate_metrics (:286-299), is_done (:301-308, sticky). */
int oige_ball_balance_post_physics(const oige_ball_balance_params* params, const oige_ball_balance_state* state,
                                   const oige_ball_balance_buffers* buffers, int64_t num_envs, uint32_t stages,
                                   oige_stream_t stream);

typedef struct oige_ball_balance_reset_io {
  const float* initial_ball_pos;       /* (N,3) post_reset constants (:266-272) */
  const float* initial_bot_pos;        /* (N,3) */
  const float* initial_bot_rot;        /* (N,4) */
  const float* initial_bot_velocities; /* (N,6) */
  const float* initial_dof_positions;  /* (N,num_dof) */
  const float* u;                      /* NULL, or (n,5) pre-drawn U[0,1): dists, dirs, hspeeds, vspeeds, height (:220-232) */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;                    /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                      /* out device scalar */
  float* ball_pos_out;                 /* out (n,3) */
  float* ball_rot_out;                 /* out (n,4) identity */
  float* ball_vel_out;                 /* out (n,6) */
  float* bot_pos_out;                  /* out (n,3) */
  float* bot_rot_out;                  /* out (n,4) */
  float* bot_vel_out;                  /* out (n,6) */
  float* dof_pos_out;                  /* out (n,num_dof) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_ball_balance_reset_io;

/* pre_physics_step (:187-206): compaction, reset_idx of the flagged envs, integrate + clamp the position targets, zero
 * the targets of the reset envs.                                                                                  */
int oige_ball_balance_pre_physics(const oige_ball_balance_params* params, const float* actions_in,
                                  const oige_ball_balance_buffers* buffers, const oige_ball_balance_reset_io* reset_io,
                                  int64_t num_envs, uint32_t flags, oige_stream_t stream);

/* reset_idx(env_ids) with caller-provided ids (:208-260).                                                         */
int oige_ball_balance_reset_idx(const oige_ball_balance_params* params, const oige_ball_balance_buffers* buffers,
                                const oige_ball_balance_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                oige_stream_t stream);

/* ------------------------------------------------------------------ Ingenuity
 * tasks/ingenuity.py.  Scalars: cfg/task/Ingenuity.yaml:5-13, ingenuity.py:56-57.                                */
#define OIGE_INGENUITY_OBS 13

typedef struct oige_ingenuity_params {
  float dt;
  float thrust_limit;             /* 2000 */
  float thrust_lateral_component; /* 0.2 */
  float clip_actions;             /* with OIGE_PRE_CLAMP_ACTIONS */
  int32_t max_episode_length;     /* 2000 */
  int32_t reserved0[3];
} oige_ingenuity_params;

typedef struct oige_ingenuity_state {  /* view getters of get_observations (:129-130), world frame, read-only */
  const float* root_pos;        /* (N,3) */
  const float* root_rot;        /* (N,4) */
  const float* root_velocities; /* (N,6) */
  const float* env_pos;         /* (N,3) */
} oige_ingenuity_state;

typedef struct oige_ingenuity_buffers {
  float* obs_buf;            /* (N,13) */
  float* rew_buf;            /* (N,) */
  int64_t* reset_buf;        /* (N,) */
  int64_t* progress_buf;     /* (N,) */
  float* target_positions;   /* (N,3) rewritten by set_targets */
  float* thrusts;            /* (N,2,3) */
  float* dof_pos;            /* (N,4) the task's own copy (:201); columns 1, 3 re-drawn on reset */
  float* dof_vel;            /* (N,4) the task's own copy; zeroed on reset, columns 1 / 3 = +-50 every step (:178-179) */
  int32_t* reset_workspace;  /* oige_workspace_ints(N): reset ids */
  int32_t* target_workspace; /* oige_workspace_ints(N): set_targets ids */
  uint8_t* reset_counts_u8;  /* optional, both or neither: the two tables' tile counts as bytes, */
  uint8_t* target_counts_u8; /*   as oige_locomotion_buffers.tile_counts_u8                      */
} oige_ingenuity_buffers;

/* post_physics_step: progress += 1, get_observations (:128-147), calculate_metrics (:244-268), is_done (:270-281). */
int oige_ingenuity_post_physics(const oige_ingenuity_params* params, const oige_ingenuity_state* state,
                                const oige_ingenuity_buffers* buffers, int64_t num_envs, uint32_t stages,
                                oige_stream_t stream);

typedef struct oige_ingenuity_reset_io {
  const float* initial_root_pos; /* (N,3) */
  const float* initial_root_rot; /* (N,4) */
  const float* initial_ball_rot; /* (N,4) */
  const float* env_pos;          /* (N,3) */
  const float* u_reset;          /* NULL, or (n,5) U[0,1): dof_pos[:,1], dof_pos[:,3], x, y, z (:228-236) */
  const float* u_target;         /* NULL, or (m,3) U[0,1): the (m,2) draw then the (m,) draw of :217-218 */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;              /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                /* out device scalar */
  float* dof_pos_out;            /* out (n,4) */
  float* dof_vel_out;            /* out (n,4) zeros */
  float* root_pos_out;           /* out (n,3) */
  float* root_rot_out;           /* out (n,4) */
  float* root_vel_out;           /* out (n,6) zeros */
  int64_t* target_ids;           /* out (N,) ascending == (progress_buf % 500 == 0).nonzero() after the resets */
  int64_t* target_count;         /* out device scalar */
  float* ball_pos_out;           /* out (m,3) target + env_pos, z + 0.4 */
  float* ball_rot_out;           /* out (m,4) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_ingenuity_reset_io;

/* pre_physics_step (:149-190): both compactions, reset_idx, set_targets, thrust vectors, rotor speeds.             */
int oige_ingenuity_pre_physics(const oige_ingenuity_params* params, const float* actions_in,
                               const oige_ingenuity_buffers* buffers, const oige_ingenuity_reset_io* reset_io,
                               int64_t num_envs, uint32_t flags, oige_stream_t stream);

/* reset_idx(env_ids) (:225-249) / set_targets(env_ids) (:213-223) with caller-provided ids (device counts in
 * reset_io->count / ->target_count).                                                                              */
int oige_ingenuity_reset_idx(const oige_ingenuity_params* params, const oige_ingenuity_buffers* buffers,
                             const oige_ingenuity_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                             oige_stream_t stream);
int oige_ingenuity_set_targets(const oige_ingenuity_params* params, const oige_ingenuity_buffers* buffers,
                               const oige_ingenuity_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                               oige_stream_t stream);

/* ------------------------------------------------------------------ Crazyflie
 * tasks/crazyflie.py.  Scalars: cfg/task/Crazyflie.yaml:5-13, crazyflie.py:57-90,262-268.                        */
#define OIGE_CRAZYFLIE_OBS 18
#define OIGE_CRAZYFLIE_EPISODE_SUMS 8  /* rew_pos, rew_orient, rew_effort, rew_spin, raw_dist, raw_orient, raw_effort, raw_spin */

typedef struct oige_crazyflie_params {
  float motor_tau_up;         /* (float)(4 dt / (motor_damp_time_up + 1e-6)) (:77) */
  float motor_tau_down;       /* (:78) */
  float prop_max_rot;         /* 433.3 (:268) */
  float clip_actions;         /* with OIGE_PRE_CLAMP_ACTIONS */
  float thrust_max[4];        /* grav_z * mass * thrust_to_weight * motor_assymetry / 4 (:262) */
  int32_t max_episode_length; /* 700 */
  int32_t reserved0[3];
} oige_crazyflie_params;

typedef struct oige_crazyflie_state {  /* view getters of get_observations (:141-142), world frame, read-only */
  const float* root_pos;        /* (N,3) */
  const float* root_rot;        /* (N,4) */
  const float* root_velocities; /* (N,6) */
  const float* env_pos;         /* (N,3) */
} oige_crazyflie_state;

typedef struct oige_crazyflie_buffers {
  float* obs_buf;            /* (N,18) */
  float* rew_buf;            /* (N,) */
  int64_t* reset_buf;        /* (N,) */
  int64_t* progress_buf;     /* (N,) */
  float* target_positions;   /* (N,3) */
  float* actions;            /* (N,4) the task's self.actions (:180), read by the effort reward */
  float* thrusts;            /* (N,4,3) */
  float* thrust_cmds_damp;   /* (N,4) */
  float* thrust_rot_damp;    /* (N,4) */
  float* dof_pos;            /* (N,4) the task's own copy */
  float* dof_vel;            /* (N,4) the task's own copy: rotor speeds (:247-251) */
  float* episode_sums;       /* (8,N): the eight episode_sums tensors as rows of one block */
  float* extras;             /* (8,) extras["episode"][key], written when at least one env resets */
  int32_t* reset_workspace;  /* oige_workspace_ints(N) */
  int32_t* target_workspace; /* oige_workspace_ints(N) */
  float* reduce_workspace;   /* 8 * ceil(N / 256) floats */
} oige_crazyflie_buffers;

/* post_physics_step: progress += 1, get_observations (:140-164), calculate_metrics (:337-378), is_done (:380-395). */
int oige_crazyflie_post_physics(const oige_crazyflie_params* params, const oige_crazyflie_state* state,
                                const oige_crazyflie_buffers* buffers, int64_t num_envs, uint32_t stages,
                                oige_stream_t stream);

typedef struct oige_crazyflie_reset_io {
  const float* initial_root_pos; /* (N,3) */
  const float* initial_root_rot; /* (N,4) */
  const float* initial_ball_rot; /* (N,4) */
  const float* env_pos;          /* (N,3) */
  const float* thrust_noise4;    /* (4,) the step's 0.01 * torch.randn(4) (:197), drawn by the caller */
  int64_t* env_ids;              /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                /* out device scalar */
  float* dof_pos_out;            /* out (n,4) zeros */
  float* dof_vel_out;            /* out (n,4) zeros */
  float* root_pos_out;           /* out (n,3) */
  float* root_rot_out;           /* out (n,4) */
  float* root_vel_out;           /* out (n,6) zeros */
  int64_t* target_ids;           /* out (N,) ascending == (progress_buf % 500 == 0).nonzero() after the resets */
  int64_t* target_count;         /* out device scalar */
  float* ball_pos_out;           /* out (m,3) */
  float* ball_rot_out;           /* out (m,4) */
} oige_crazyflie_reset_io;

/* pre_physics_step (:166-258): both compactions, reset_idx incl. the episode-sum means (extras), set_targets, motor
 * filter, thrust along the body z axis (root_rot: the view's current orientations), rotor speeds.                 */
int oige_crazyflie_pre_physics(const oige_crazyflie_params* params, const float* actions_in, const float* root_rot,
                               const oige_crazyflie_buffers* buffers, const oige_crazyflie_reset_io* reset_io,
                               int64_t num_envs, uint32_t flags, oige_stream_t stream);

/* reset_idx(env_ids) rows, flags and filter states (:313-329; the episode-sum means of :331-335 stay two torch lines
 * with the caller in this flow) / set_targets(env_ids) (:300-311), caller-provided ids.                           */
int oige_crazyflie_reset_idx(const oige_crazyflie_params* params, const oige_crazyflie_buffers* buffers,
                             const oige_crazyflie_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                             oige_stream_t stream);
int oige_crazyflie_set_targets(const oige_crazyflie_params* params, const oige_crazyflie_buffers* buffers,
                               const oige_crazyflie_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                               oige_stream_t stream);

/* ------------------------------------------------------------------ Cartpole
 * tasks/cartpole.py:91-164.                                                                 */
typedef struct oige_cartpole_params {
  int32_t cart_dof_idx;   /* cartpole.py:147 */
  int32_t pole_dof_idx;   /* cartpole.py:148 */
  int32_t num_dof;        /* row width of dof_pos/dof_vel (2) */
  int32_t reserved0;
  float reset_dist;       /* Cartpole.yaml:10 */
  float max_push_effort;  /* Cartpole.yaml:11 */
  float max_episode_length; /* 500, cartpole.py:44 */
  float reserved1;
} oige_cartpole_params;

typedef struct oige_cartpole_buffers {
  float* obs_buf;         /* (N,4) */
  float* rew_buf;         /* (N,) */
  int64_t* reset_buf;     /* (N,) */
  int64_t* progress_buf;  /* (N,) */
  int32_t* reset_workspace; /* as oige_locomotion_buffers.reset_workspace */
} oige_cartpole_buffers;

/* post_physics_step: progress += 1, get_observations (:91-106), calculate_metrics (:153-158),
 * is_done (:160-164, NOT sticky, >= max_episode_length).  `stages` as for locomotion.       */
int oige_cartpole_post_physics(const oige_cartpole_params* params, const float* dof_pos, const float* dof_vel,
                               const oige_cartpole_buffers* buffers, int64_t num_envs, uint32_t stages,
                               oige_stream_t stream);

typedef struct oige_cartpole_reset_io {
  const float* u;      /* NULL, or (4,n) pre-drawn U[0,1): rows = the four torch.rand(n) draws in
                          cartpole.py:129,130,134,135 order */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;    /* out (N,) */
  int64_t* count;      /* out device scalar */
  float* dof_pos_out;  /* out (n,num_dof) */
  float* dof_vel_out;  /* out (n,num_dof) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_cartpole_reset_io;

/* pre_physics_step (:108-122): compaction, reset_idx (:124-144), forces[:, cart] = max_effort * a[:,0] */
int oige_cartpole_pre_physics(const oige_cartpole_params* params, const float* actions /* (N,1) */,
                              const oige_cartpole_buffers* buffers, const oige_cartpole_reset_io* reset_io,
                              float* forces_out /* (N,num_dof) */, int64_t num_envs, uint32_t flags,
                              oige_stream_t stream);
int oige_cartpole_reset_idx(const oige_cartpole_params* params, const oige_cartpole_buffers* buffers,
                            const oige_cartpole_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                            oige_stream_t stream);

/* ------------------------------------------------------------------ FrankaCabinet, TomoDJCabinet, TomoSPCCabinet
 *   tasks/franka_cabinet.py; tasks/tomodj_cabinet.py:373-712 and tasks/tomospc_cabinet.py are the same
 * skeleton with 9 / 8 DOFs and their own reward variant.  Scalars: cfg/task/FrankaCabinet.yaml:8-32 (TomoDJCabinet.yaml,
 * TomoSPCCabinet.yaml), franka_cabinet.py:52-53,257-279.
 * The per-env tensors the reference builds with .repeat((num_envs, 1)) (:257-275: local grasp poses and the four axes) have
 * identical rows; they cross the ABI as the row.                                                                 */
#define OIGE_FRANKA_DOFS 9 /* array capacity; num_dofs = 9 (Franka, TomoDJ) or 8 (TomoSPC); obs width = 2 * num_dofs + 5 */
#define OIGE_FRANKA_OBS 23
#define OIGE_CABINET_DOFS 4
#define OIGE_CABINET_REWARD_FRANKA 0 /* franka_cabinet.py:526-601 */
#define OIGE_CABINET_REWARD_TOMO 1   /* tomodj_cabinet.py:605-684 == tomospc_cabinet.py: dist_reward = -d, rot_reward zeroed
                                        below 0.6 and doubled from 0.8, finger-height tests (<=, >=) */

typedef struct oige_franka_cabinet_params {
  float action_scale;                /* 7.5 */
  float dof_vel_scale;               /* 0.1 */
  float dt;                          /* 1/60 (:53) */
  float clip_actions;                /* with OIGE_PRE_CLAMP_ACTIONS */
  float dist_reward_scale, rot_reward_scale, around_handle_reward_scale, open_reward_scale;
  float finger_dist_reward_scale, action_penalty_scale, finger_close_reward_scale;
  float open_bonus_threshold;        /* second drawer bonus: 0.2 (franka_cabinet.py:592), 0.1 (tomodj_cabinet.py:675) */
  float open_done_threshold;         /* third bonus and is_done: 0.39 (:593, :470), 0.19 (tomodj_cabinet.py:676, :569) */
  int32_t num_dofs;                  /* 9 or 8; the finger-close term reads joints num_dofs-2, num_dofs-1 (:571) */
  int32_t reward_variant;            /* OIGE_CABINET_REWARD_* */
  int32_t reserved0;
  float franka_local_grasp_pos[3];   /* (:253-257) */
  float franka_local_grasp_rot[4];   /* wxyz (:258) */
  float drawer_local_grasp_pos[3];   /* (0.3, 0.01, 0) (:260-261) */
  float drawer_local_grasp_rot[4];   /* (1, 0, 0, 0) (:262) */
  float gripper_forward_axis[3];     /* (0, 0, 1) (:264) */
  float drawer_inward_axis[3];       /* (-1, 0, 0) (:267) */
  float gripper_up_axis[3];          /* (0, 1, 0) (:270) */
  float drawer_up_axis[3];           /* (0, 0, 1) (:273) */
  float dof_lower[OIGE_FRANKA_DOFS]; /* (:421-422) */
  float dof_upper[OIGE_FRANKA_DOFS];
  float dof_speed_scales[OIGE_FRANKA_DOFS]; /* 1, gripper joints 0.1 (:423-424) */
  float default_dof_pos[OIGE_FRANKA_DOFS];  /* (:277-279) */
  int32_t max_episode_length;        /* 500 */
  int32_t num_props;                 /* 4; 0 = no prop rows */
} oige_franka_cabinet_params;

typedef struct oige_franka_cabinet_state {  /* view getters of get_observations (:284-289, :336-337), read-only */
  const float* hand_pos;        /* (N,3) world */
  const float* hand_rot;        /* (N,4) wxyz */
  const float* drawer_pos;      /* (N,3) */
  const float* drawer_rot;      /* (N,4) */
  const float* franka_dof_pos;  /* (N,9) */
  const float* franka_dof_vel;  /* (N,9) */
  const float* cabinet_dof_pos; /* (N,4): column 3 = drawer_top_joint */
  const float* cabinet_dof_vel; /* (N,4) */
  const float* lfinger_pos;     /* (N,3) */
  const float* rfinger_pos;     /* (N,3): the reference passes the LEFT finger view here too (:337) */
} oige_franka_cabinet_state;

typedef struct oige_franka_cabinet_buffers {
  float* obs_buf;             /* (N,23) */
  float* rew_buf;             /* (N,) */
  int64_t* reset_buf;         /* (N,) sticky (:470-473) */
  int64_t* progress_buf;      /* (N,) */
  float* actions;             /* (N,9): written by pre_physics (:368), read by the reward */
  float* franka_dof_targets;  /* (N,9) */
  float* franka_grasp_rot;    /* (N,4) or NULL: the task attributes of :293-296, written by the obs stage */
  float* franka_grasp_pos;    /* (N,3) or NULL */
  float* drawer_grasp_rot;    /* (N,4) or NULL */
  float* drawer_grasp_pos;    /* (N,3) or NULL */
  int32_t* reset_workspace;   /* as oige_locomotion_buffers.reset_workspace */
  uint8_t* tile_counts_u8;    /* optional: as oige_locomotion_buffers.tile_counts_u8 */
} oige_franka_cabinet_buffers;

/* post_physics_step: progress += 1, get_observations (:283-358, incl. compute_grasp_transforms :475-494),
 * calculate_metrics (:439-466 -> compute_franka_reward :496-601), is_done (:468-473).  `stages` as for locomotion.      */
int oige_franka_cabinet_post_physics(const oige_franka_cabinet_params* params, const oige_franka_cabinet_state* state,
                                     const oige_franka_cabinet_buffers* buffers, int64_t num_envs, uint32_t stages,
                                     oige_stream_t stream);

typedef struct oige_franka_cabinet_reset_io {
  const float* u_dof;            /* NULL, or (n,9) pre-drawn U[0,1): the draw of :382 */
  const float* default_prop_pos; /* (N*num_props,3) (:430), may be NULL when num_props == 0 */
  const float* default_prop_rot; /* (N*num_props,4) */
  uint64_t seed;
  uint64_t step;
  int64_t* env_ids;              /* out (N,) ascending == reset_buf.nonzero() */
  int64_t* count;                /* out device scalar */
  float* dof_pos_out;            /* out (n,9): also the rows of set_joint_position_targets(..., indices) (:408) */
  float* dof_vel_out;            /* out (n,9) zeros */
  float* cabinet_dof_pos_out;    /* out (n,4) zeros (:393-398) */
  float* cabinet_dof_vel_out;    /* out (n,4) zeros */
  int64_t* prop_ids;             /* out (n*num_props,) = prop_indices[env_ids].flatten() (:403-405) */
  int64_t* prop_count;           /* out device scalar = count * num_props */
  float* prop_pos_out;           /* out (n*num_props,3) */
  float* prop_rot_out;           /* out (n*num_props,4) */
  const uint64_t* step_ptr; /* NULL, or a device scalar read INSTEAD of `step` (HIP-graph replay), as
                               oige_locomotion_reset_io.step_ptr */
} oige_franka_cabinet_reset_io;

/* pre_physics_step (:360-373): compaction, reset_idx of the flagged envs, actions clone, then for ALL envs (the reset ones
 * start from their new positions) targets = clamp(targets + speed_scales * dt * actions * action_scale, lower, upper).      */
int oige_franka_cabinet_pre_physics(const oige_franka_cabinet_params* params, const float* actions_in,
                                    const oige_franka_cabinet_buffers* buffers,
                                    const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, uint32_t flags,
                                    oige_stream_t stream);

/* reset_idx(env_ids) with caller-provided ids (:375-414).                                                          */
int oige_franka_cabinet_reset_idx(const oige_franka_cabinet_params* params, const oige_franka_cabinet_buffers* buffers,
                                  const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                  oige_stream_t stream);

/* ------------------------------------------------------------------ TomoDJnewhandCabinet, TomoDJnewhandTendonCabinet,
 * TomoSPCnewhandCabinet: the fork's five-finger cabinet tasks, tasks/tomodj_newhand_cabinet.py (26 DOFs),
 * tasks/tomodj_newhandTendon_cabinet.py (26 DOFs, 22 actuated) and tasks/tomospc_newhand_cabinet.py (25 DOFs).  Line numbers
 * below are tomodj_newhand_cabinet.py's.  Per-env tensors with identical rows cross the ABI as the row.              */
#define OIGE_NEWHAND_MAX_DOFS 26
#define OIGE_NEWHAND_KEYPOINTS 5

typedef struct oige_newhand_cabinet_params {
  float action_scale, dof_vel_scale, dt, clip_actions;
  float dist_reward_scale, rot_reward_scale, around_handle_reward_scale, open_reward_scale, action_penalty_scale;
  float around_lower;                 /* 0.2732: every fingertip keypoint's x offset from the drawer must exceed it ... */
  float around_upper;                 /* ... and stay below 0.3149 (TomoSPC: 0.3154) */
  float reserved0;
  int32_t num_dofs;                   /* 26 or 25; obs width = 2 * num_dofs + 5 */
  int32_t num_actions;                /* 26, 25, or 22 (Tendon: the actuated joints) */
  int32_t rot_reward_thresholds;      /* TomoSPC: rot_reward zeroed below 0.6, doubled from 0.8 */
  int32_t terminate_on_limits;        /* Tendon: terminate_buf = any scaled joint position outside +-1.05; -100 reward */
  int32_t max_episode_length;         /* 500 */
  int32_t num_props;
  float drawer_local_keypoint_offsets[OIGE_NEWHAND_KEYPOINTS][3]; /* post_reset */
  float local_thumb_pose_pos[3];              /* (0.038, 0.008, 0) */
  float local_another_finger_pose_pos[3];     /* (0.024, -0.002, 0) */
  float left_hand_forward_axis[3], drawer_inward_axis[3], left_hand_up_axis[3], drawer_up_axis[3];
  float dof_lower[OIGE_NEWHAND_MAX_DOFS], dof_upper[OIGE_NEWHAND_MAX_DOFS];
  float dof_speed_scales[OIGE_NEWHAND_MAX_DOFS];  /* 1, finger joints 5 */
  float default_dof_pos[OIGE_NEWHAND_MAX_DOFS];
  int32_t actuated_dof_indices[OIGE_NEWHAND_MAX_DOFS]; /* first num_actions entries: action column -> joint */
} oige_newhand_cabinet_params;

typedef struct oige_newhand_cabinet_state {  /* view getters of get_observations, read-only */
  const float* hand_rot;                           /* (N,4) left hand */
  const float* finger_pos[OIGE_NEWHAND_KEYPOINTS]; /* (N,3) each: thumb, fore, middle, ring, little */
  const float* finger_rot[OIGE_NEWHAND_KEYPOINTS]; /* (N,4) each */
  const float* drawer_pos;                         /* (N,3) */
  const float* drawer_rot;                         /* (N,4) */
  const float* dof_pos;                            /* (N,D) */
  const float* dof_vel;                            /* (N,D) */
  const float* cabinet_dof_pos;                    /* (N,4): column 3 = drawer_top_joint */
  const float* cabinet_dof_vel;                    /* (N,4) */
} oige_newhand_cabinet_state;

typedef struct oige_newhand_cabinet_buffers {
  float* obs_buf;            /* (N, 2D+5) */
  float* rew_buf;            /* (N,) */
  int64_t* reset_buf;        /* (N,) sticky */
  int64_t* progress_buf;     /* (N,) */
  float* actions;            /* (N,A) */
  float* dof_targets;        /* (N,D) */
  float* keypoints_finger;   /* (N,5,3) or NULL: task attributes written by the obs stage */
  float* keypoints_drawer;   /* (N,5,3) or NULL */
  int64_t* terminate_buf;    /* (N,) with terminate_on_limits: read by the reward (the PREVIOUS is_done's flags), rewritten by is_done */
  int32_t* reset_workspace;  /* as oige_locomotion_buffers.reset_workspace */
  float* actuated_targets_out; /* (N,A) or NULL: pre_physics also writes dof_targets[:, actuated_dof_indices], the block the Tendon
                                * task hands to set_joint_position_targets(joint_indices=...) (tomodj_newhandTendon_cabinet.py) */
  uint8_t* tile_counts_u8;       /* optional: as oige_locomotion_buffers.tile_counts_u8 */
} oige_newhand_cabinet_buffers;

/* post_physics_step: progress += 1, get_observations, calculate_metrics (compute_*_reward), is_done.  `stages` as for locomotion. */
int oige_newhand_cabinet_post_physics(const oige_newhand_cabinet_params* params, const oige_newhand_cabinet_state* state,
                                      const oige_newhand_cabinet_buffers* buffers, int64_t num_envs, uint32_t stages,
                                      oige_stream_t stream);

/* reset rows: same layout as oige_franka_cabinet_reset_io with D = num_dofs columns; u_dof is (n,D).               */
int oige_newhand_cabinet_pre_physics(const oige_newhand_cabinet_params* params, const float* actions_in,
                                     const oige_newhand_cabinet_buffers* buffers,
                                     const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, uint32_t flags,
                                     oige_stream_t stream);
int oige_newhand_cabinet_reset_idx(const oige_newhand_cabinet_params* params, const oige_newhand_cabinet_buffers* buffers,
                                   const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                   oige_stream_t stream);

/* ------------------------------------------------------------------ Episode-stat observer
 * utils/rlgames/rlgames_utils.py:52-85 (RLGPUAlgoObserver.process_infos / after_print_stats).                     */
#define OIGE_OBSERVER_MAX_KEYS 64

typedef struct oige_observer_batch {  /* one step's infos["episode"]: per key a device pointer to `length` floats */
  const float* values[OIGE_OBSERVER_MAX_KEYS];
  int32_t lengths[OIGE_OBSERVER_MAX_KEYS]; /* 1 for a scalar / 0-dim tensor (:80-81), n for a 1-D tensor */
  int32_t num_keys;
  int32_t reserved0;
} oige_observer_batch;

/* process_infos (:52-56): sums[k] += sum(values[k][0:length]) in f64, counts[k] += length.  One launch per step.     */
int oige_observer_accumulate(const oige_observer_batch* batch, double* sums /* (num_keys,) */,
                             int64_t* counts /* (num_keys,) */, oige_stream_t stream);

/* after_print_stats (:73-85): means_out[k] = (float)(sums[k] / counts[k]) (nan when nothing was appended, as torch.mean of
 * an empty tensor); clear != 0 also zeroes sums and counts (ep_infos.clear(), :85).                                */
int oige_observer_means(double* sums, int64_t* counts, float* means_out /* (num_keys,) */, int32_t num_keys,
                        int32_t clear, oige_stream_t stream);

/* ---- per-step episode statistics of one rank (the multi-GPU path's only exchange) ----------------------------------
 * The reference runs one process per GPU with its own envs and seed + rank (scripts/rlgames_train.py:114-120,171); what is
 * combined across ranks is the episode-statistics means the trainer logs once per epoch (utils/rlgames/rlgames_utils.py:
 * 52-56 collects per step, :72-96 reduces per epoch).  This rank's running sums live on the device as SLOTS of four f64
 *     slot[0] += sum(rew_buf)   slot[1] += count(reset_buf != 0)   slot[2] += sum(progress_buf where reset_buf != 0)
 *     slot[3] += sum(successes) (0 when NULL)
 * each slot owned by one workgroup of oige_episode_stats_accumulate (one launch per step) -- or by one 64-env tile of a post
 * kernel that keeps them itself (oige_locomotion_buffers.stats_slots: no launch at all).  oige_episode_stats_fold adds the
 * slots up into acc[0..3] and sets acc[4] = env_steps when a collective is due; the caller all-reduces `acc` (40 bytes) over
 * RCCL.  Every sum has a fixed order: results do not depend on scheduling.  No allocation, no synchronisation.         */
#define OIGE_EPISODE_STATS_MAX_BLOCKS 1024

typedef struct oige_episode_stats_io {
  const float* rew_buf;        /* (N,) */
  const void* reset_buf;       /* (N,) int64, or bool / uint8 (AnymalTerrain, anymal_terrain.py:465) */
  const int64_t* progress_buf; /* (N,) */
  const float* successes;      /* (N,) or NULL */
  double* slots;               /* 4 * oige_episode_stats_blocks(N) doubles, zeroed once by the caller, read-modify-write */
  int32_t reset_elem_bytes;    /* 8 or 1 */
  int32_t reserved0;
} oige_episode_stats_io;

int64_t oige_episode_stats_blocks(int64_t num_envs);
int oige_episode_stats_accumulate(const oige_episode_stats_io* io, int64_t num_envs, oige_stream_t stream);
int oige_episode_stats_fold(const double* slots, int64_t num_slots, double* acc /* (5,) */, double env_steps,
                            double* scratch /* oige_episode_stats_fold_scratch(num_slots) doubles; NULL when that is 0 */,
                            oige_stream_t stream);
int64_t oige_episode_stats_fold_scratch(int64_t num_slots);

#ifdef __cplusplus
}
#endif
#endif /* OIGE_HIP_H */
