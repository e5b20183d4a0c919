// AnymalTerrain per-step task math for gfx950 (MI355X).
//
// Replaces /root/reference/omniisaacgymenvs/tasks/anymal_terrain.py:
//   pre_physics_step PD torques :391-407, post_physics_step :409-444 (frame quantities :421-427,
//   check_termination :452-466, calculate_metrics :468-533, in-step reset_idx :318-367 with
//   update_terrain_level :369-380, get_observations + get_heights :535-604, obs noise :438-439,
//   last_* copies :441-442), jit quat_apply_yaw :608-612, wrap_to_pi :615-619.
//
// post_physics = four launches on one stream, no host sync:
//   flags   : progress += 1, timeout / has_fallen / reset flags, per-tile reset counts and the
//             per-tile sum of squared xy-commands of the reset envs            (~90 B/env)
//   scan    : group offsets of the reset compaction + the reference's whole-matrix command norm
//             (anymal_terrain.py:376 takes torch.norm over ALL reset envs' commands)
//   main    : one 256-thread workgroup per 64-env tile: reward, episode sums, the in-step reset,
//             140-point yaw-rotated height scan (2 int16 gathers per point from the L2-resident
//             heightfield), obs tile assembled in LDS and streamed out with the uniform noise
//   extras  : means of the episode sums over the reset envs, mean terrain level
// HBM-bound: 1985 algorithmic bytes per env-step; the gathers hit L2/Infinity Cache.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

constexpr int AD = OIGE_ANYMAL_DOFS;          // 12
constexpr int AO = OIGE_ANYMAL_OBS;           // 188
constexpr int AP = OIGE_ANYMAL_SCAN_POINTS;   // 140
constexpr int AE = OIGE_ANYMAL_EPISODE_SUMS;  // 13
constexpr int ANT = 256;
constexpr int RW = 16;  // reduce_workspace floats per tile

struct AnymalPost {
  oige_anymal_params p;
  oige_anymal_state s;
  oige_anymal_buffers b;
  oige_anymal_reset_io r;
  int64_t n;
  int mode;  // 0: post_physics_step, 1: standalone reset_idx(env_ids) driven by r.reset_mask
};

// wrap_to_pi AS TORCHSCRIPT RUNS IT (:615-619): `angles %= 2*pi` of a scripted function is aten::fmod_
// (C fmod, sign of the dividend), then `-= 2*pi * (angles > pi)`.
__device__ __forceinline__ float wrap_to_pi(float x) {
  const float m = __builtin_fmodf(x, OIGE_TWO_PI_F);
  return m - OIGE_TWO_PI_F * (m > OIGE_PI_F ? 1.0f : 0.0f);
}

__device__ __forceinline__ float wave_sum_ff(float v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// ------------------------------------------------------------------ PD torques (:398-403)
__global__ __launch_bounds__(256) void anymal_torques_kernel(const oige_anymal_params p, const float* __restrict__ actions,
                                                             const float* __restrict__ dof_pos,
                                                             const float* __restrict__ dof_vel,
                                                             float* __restrict__ torques, int64_t total) {
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = 4 * f;  // total = N*12 is a multiple of 4 and a float4 never straddles a row
  if (e >= total) return;
  const float4 a = *reinterpret_cast<const float4*>(actions + e);
  const float4 q = *reinterpret_cast<const float4*>(dof_pos + e);
  const float4 v = *reinterpret_cast<const float4*>(dof_vel + e);
  const int c0 = (int)(e % AD);
  const float av[4] = {a.x, a.y, a.z, a.w}, qv[4] = {q.x, q.y, q.z, q.w}, vv[4] = {v.x, v.y, v.z, v.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = p.Kp * ((p.action_scale * av[k] + p.default_dof_pos[c0 + k]) - qv[k]) - p.Kd * vv[k];
    o[k] = (t != t) ? t : (t < -80.0f ? -80.0f : (t > 80.0f ? 80.0f : t));
  }
  *reinterpret_cast<float4*>(torques + e) = make_float4(o[0], o[1], o[2], o[3]);
}

// ------------------------------------------------------------------ flags (:410, :452-466)
__global__ __launch_bounds__(GROUP) void anymal_flags_kernel(const AnymalPost a) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool live = env < a.n;
  bool reset = false;
  float cmd2 = 0.0f;
  if (live && a.mode == 0) {
    const int64_t progress = a.b.progress_buf[env] + 1;  // :410
    a.b.progress_buf[env] = progress;
    const bool timeout = progress >= (int64_t)a.p.max_episode_length - 1;  // :453-457
    const float4* kc = reinterpret_cast<const float4*>(a.s.knee_contact + env * 12);
    const float4 k0 = kc[0], k1 = kc[1], k2 = kc[2];
    int knees = 0;  // :458-461
    knees += norm3(f3{k0.x, k0.y, k0.z}) > 1.0f ? 1 : 0;
    knees += norm3(f3{k0.w, k1.x, k1.y}) > 1.0f ? 1 : 0;
    knees += norm3(f3{k1.z, k1.w, k2.x}) > 1.0f ? 1 : 0;
    knees += norm3(f3{k2.y, k2.z, k2.w}) > 1.0f ? 1 : 0;
    const float* bc = a.s.base_contact + env * 3;
    const bool has_fallen = (norm3(f3{bc[0], bc[1], bc[2]}) > 1.0f) | (knees > 1);  // :462-464
    reset = has_fallen | timeout;                                                     // :465-466
    a.b.timeout_buf[env] = timeout ? 1 : 0;
    a.b.has_fallen[env] = has_fallen ? 1 : 0;
    a.b.reset_buf[env] = reset ? 1 : 0;
  } else if (live) {
    reset = a.r.reset_mask[env] != 0;  // reset_idx(env_ids): the caller marked env_ids
  }
  if (reset) {
    const float2 c = *reinterpret_cast<const float2*>(a.b.commands + env * 4);
    cmd2 = c.x * c.x + c.y * c.y;
  }
  const unsigned long long m = __ballot(reset);
  const float s = wave_sum_ff(cmd2);
  if (lane_id() == 0) {
    const int64_t t = env / TILE;
    a.b.reset_workspace[t] = __popcll(m);
    if (t < num_tiles(a.n)) a.b.reduce_workspace[t * RW + 0] = s;
  }
}

// whole-matrix norm of the reset envs' xy commands (:376): sqrt of the sum over tiles, tile order
__global__ __launch_bounds__(1024) void anymal_command_norm_kernel(float* __restrict__ ws, int64_t tiles) {
  __shared__ double part[1024 / WAVE];
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < tiles; i += 1024) acc += (double)ws[i * RW];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 1024 / WAVE; ++i) t += part[i];
    ws[tiles * RW + 0] = __builtin_sqrtf((float)t);
  }
}

// ------------------------------------------------------------------ main
template <bool FULLTILE>
__device__ __forceinline__ void anymal_post_tile(const AnymalPost& a, float* __restrict__ lds, int64_t t, int nv) {
  float* obs_t = lds;                       // 64 x 188
  float* s_pose = lds + TILE * AO;           // 64 x float4: normalised yaw quat (w, z), base pos (x, y)
  float* s_pz = s_pose + TILE * 4;           // 64: base z - 0.5
  float* s_hx = s_pz + TILE;                 // 140 + 140: the scan grid (:173-187)
  float* s_hy = s_hx + AP;
  float* s_ns = s_hy + AP;                   // 188: noise_scale_vec (:157-171)
  const oige_anymal_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t env0 = t * TILE;
  const int64_t T = num_tiles(a.n);
  constexpr int C_DPOS = 12, C_DVEL = 24, C_ACT = 176, C_LA = 36, C_LDV = 48, C_TQ = 60;  // scratch in the scan columns

  const bool step = a.mode == 0;
  // ---- A/B: the six 12-wide arrays, lane-linear, into tile columns
  if (step) {
    WgChunk<AD, ANT> c0, c1, c2, c3, c4, c5;
    wg_load_chunk<AD, ANT, FULLTILE>(c0, a.s.dof_pos + env0 * AD, nv * AD);
    wg_load_chunk<AD, ANT, FULLTILE>(c1, a.s.dof_vel + env0 * AD, nv * AD);
    wg_load_chunk<AD, ANT, FULLTILE>(c2, a.b.actions + env0 * AD, nv * AD);
    wg_load_chunk<AD, ANT, FULLTILE>(c3, a.b.last_actions + env0 * AD, nv * AD);
    wg_load_chunk<AD, ANT, FULLTILE>(c4, a.b.last_dof_vel + env0 * AD, nv * AD);
    wg_load_chunk<AD, ANT, FULLTILE>(c5, a.b.torques + env0 * AD, nv * AD);
    wg_for_each<AD, ANT>(c0, [&](int e, int c, float v) { obs_t[e * AO + C_DPOS + c] = v; });
    wg_for_each<AD, ANT>(c1, [&](int e, int c, float v) { obs_t[e * AO + C_DVEL + c] = v; });
    wg_for_each<AD, ANT>(c2, [&](int e, int c, float v) { obs_t[e * AO + C_ACT + c] = v; });
    wg_for_each<AD, ANT>(c3, [&](int e, int c, float v) { obs_t[e * AO + C_LA + c] = v; });
    wg_for_each<AD, ANT>(c4, [&](int e, int c, float v) { obs_t[e * AO + C_LDV + c] = v; });
    wg_for_each<AD, ANT>(c5, [&](int e, int c, float v) { obs_t[e * AO + C_TQ + c] = v; });
  }
  if (step) {
    if (tid < AP) {  // meshgrid(x, y, indexing='ij') flattened: k = ix * 10 + iy
      const int ix = tid / 10, iy = tid % 10;
      s_hx[tid] = 0.1f * (float)(ix < 7 ? ix - 8 : ix - 5);  // :178-180
      s_hy[tid] = 0.1f * (float)(iy < 5 ? iy - 5 : iy - 4);  // :175-177
    }
    if (tid < AO) {
      const int col = tid;
      s_ns[tid] = col < 3 ? p.noise_scales[0] : col < 6 ? p.noise_scales[1] : col < 9 ? p.noise_scales[2]
                : col < 12 ? 0.0f : col < 24 ? p.noise_scales[3] : col < 36 ? p.noise_scales[4]
                : col < 176 ? p.noise_scales[5] : 0.0f;
    }
  }
  __syncthreads();

  // ---- C: per-env math, one env per lane of wave 0
  if (tid < TILE) {
    const bool live = FULLTILE || tid < nv;
    const int64_t env = live ? env0 + tid : env0;  // dead lanes shadow env0, never store
    float* row = obs_t + tid * AO;
    float dp[AD], dv[AD], act[AD], la[AD], ldv[AD], tq[AD];
#pragma unroll
    for (int j = 0; j < AD; ++j) {
      dp[j] = row[C_DPOS + j]; dv[j] = row[C_DVEL + j]; act[j] = row[C_ACT + j];
      la[j] = row[C_LA + j]; ldv[j] = row[C_LDV + j]; tq[j] = row[C_TQ + j];
    }
    const float* bp3 = a.s.base_pos + env * 3;
    f3 bpos{bp3[0], bp3[1], bp3[2]};
    const float4 q4v = *reinterpret_cast<const float4*>(a.s.base_quat + env * 4);
    q4 bq{q4v.x, q4v.y, q4v.z, q4v.w};
    float2 v0 = make_float2(0.f, 0.f), v1 = v0, v2 = v0;
    if (step) {  // reset_idx(env_ids) on its own reads no velocities
      const float2* vp = reinterpret_cast<const float2*>(a.s.base_velocities + env * 6);
      v0 = vp[0]; v1 = vp[1]; v2 = vp[2];
    }
    const float4 cm = *reinterpret_cast<const float4*>(a.b.commands + env * 4);
    float cmd[4] = {cm.x, cm.y, cm.z, cm.w};
    const bool reset = live && (step ? a.b.reset_buf[env] : a.r.reset_mask[env]) != 0;
    const bool has_fallen = step && a.b.has_fallen[env] != 0;
    const int64_t timeout = step ? a.b.timeout_buf[env] : 0;

    if (!step) {
#pragma unroll
      for (int j = 0; j < AD; ++j) { dp[j] = 0.f; dv[j] = 0.f; act[j] = 0.f; la[j] = 0.f; ldv[j] = 0.f; tq[j] = 0.f; }
    }
    // frame quantities (:422-427)
    const f3 lin = quat_rotate_inverse(bq, f3{v0.x, v0.y, v1.x});
    const f3 ang = quat_rotate_inverse(bq, f3{v1.y, v2.x, v2.y});
    const f3 grav = quat_rotate_inverse(bq, f3{0.0f, 0.0f, -1.0f});
    const f3 fwd = quat_apply(bq, f3{1.0f, 0.0f, 0.0f});
    const float heading = atan2f(fwd.y, fwd.x);
    {
      const float w = 0.5f * wrap_to_pi(cmd[3] - heading);
      if (step) cmd[2] = (w != w) ? w : (w < -1.0f ? -1.0f : (w > 1.0f ? 1.0f : w));
    }

    // calculate_metrics (:468-533)
    const float* rs = p.rew_scales;
    const float ex = cmd[0] - lin.x, ey = cmd[1] - lin.y;
    const float lin_vel_error = ex * ex + ey * ey;
    const float ez = cmd[2] - ang.z;
    const float ang_vel_error = ez * ez;
    const float rew_lin_vel_xy = expf((-lin_vel_error) / 0.25f) * rs[OIGE_AR_LIN_VEL_XY];
    const float rew_ang_vel_z = expf((-ang_vel_error) / 0.25f) * rs[OIGE_AR_ANG_VEL_Z];
    const float rew_lin_vel_z = (lin.z * lin.z) * rs[OIGE_AR_LIN_VEL_Z];
    const float rew_ang_vel_xy = (ang.x * ang.x + ang.y * ang.y) * rs[OIGE_AR_ANG_VEL_XY];
    const float rew_orient = (grav.x * grav.x + grav.y * grav.y) * rs[OIGE_AR_ORIENT];
    const float bh = bpos.z - 0.52f;
    const float rew_base_height = (bh * bh) * rs[OIGE_AR_BASE_HEIGHT];
    float s_tq = 0.f, s_acc = 0.f, s_rate = 0.f, s_hip = 0.f;
#pragma unroll
    for (int j = 0; j < AD; ++j) {
      const float t2 = tq[j] * tq[j];
      const float d1 = ldv[j] - dv[j], d2 = la[j] - act[j];
      s_tq = j == 0 ? t2 : s_tq + t2;
      s_acc = j == 0 ? d1 * d1 : s_acc + d1 * d1;
      s_rate = j == 0 ? d2 * d2 : s_rate + d2 * d2;
      if (j < 4) {
        const float h = __builtin_fabsf(dp[j] - p.default_dof_pos[j]);
        s_hip = j == 0 ? h : s_hip + h;
      }
    }
    const float rew_torque = s_tq * rs[OIGE_AR_TORQUE];
    const float rew_joint_acc = s_acc * rs[OIGE_AR_JOINT_ACC];
    const float rew_action_rate = s_rate * rs[OIGE_AR_ACTION_RATE];
    const float rew_hip = s_hip * rs[OIGE_AR_HIP];
    const float rew_fallen_over = (has_fallen ? 1.0f : 0.0f) * rs[OIGE_AR_FALLEN_OVER];
    float rew = rew_lin_vel_xy + rew_ang_vel_z;
    rew += rew_lin_vel_z; rew += rew_ang_vel_xy; rew += rew_orient; rew += rew_base_height;
    rew += rew_torque; rew += rew_joint_acc; rew += rew_action_rate; rew += rew_hip; rew += rew_fallen_over;
    rew = (rew != rew) ? rew : (rew < 0.0f ? 0.0f : rew);  // torch.clip(min=0)
    // `termination * reset_buf * ~timeout_buf`: bitwise NOT of an int64 0/1 (:521) -> -1 / -2
    rew += (rs[OIGE_AR_TERMINATION] * (reset ? 1.0f : 0.0f)) * (float)(~timeout);

    // episode sums (:524-533); dict order :68-82 -> row index
    float es[AE];
    if (live) {
#pragma unroll
      for (int k = 0; k < AE; ++k) es[k] = a.b.episode_sums[(int64_t)k * a.n + env];
      if (step) {
      es[0] += rew_lin_vel_xy; es[2] += rew_ang_vel_z; es[1] += rew_lin_vel_z; es[3] += rew_ang_vel_xy;
      es[4] += rew_orient; es[5] += rew_torque; es[6] += rew_joint_acc; es[11] += rew_action_rate;
      es[7] += rew_base_height; es[12] += rew_hip;
      }
    } else {
#pragma unroll
      for (int k = 0; k < AE; ++k) es[k] = 0.f;
    }

    // in-step reset (:433-435 -> reset_idx :318-367)
    const unsigned long long rmask = __ballot(reset);
    int64_t level = live ? a.b.terrain_levels[env] : 0;
    q4 hq = bq;       // pose the height scan sees
    f3 hpos = bpos;
    if (reset) {
      const int64_t g = t / GROUP_TILES;
      const int32_t* tile_counts = a.b.reset_workspace;
      const int32_t* group_offsets = a.b.reset_workspace + padded_tiles(a.n);
      int64_t k = group_offsets[g];
      for (int64_t tt = g * GROUP_TILES; tt < t; ++tt) k += tile_counts[tt];
      k += rank_below(rmask);
      float u[29];
      if (a.r.u_dof_offset) {
#pragma unroll
        for (int j = 0; j < AD; ++j) {
          u[j] = a.r.u_dof_offset[k * AD + j];
          u[AD + j] = a.r.u_dof_vel[k * AD + j];
        }
        u[24] = a.r.u_xy[k * 2]; u[25] = a.r.u_xy[k * 2 + 1];
        u[26] = a.r.u_cmd[k * 3]; u[27] = a.r.u_cmd[k * 3 + 1]; u[28] = a.r.u_cmd[k * 3 + 2];
      } else {
#pragma unroll
        for (int blk = 0; blk < 8; ++blk) {
          const u4 c{(uint32_t)env, (uint32_t)blk | (1u << 16), (uint32_t)a.r.step, (uint32_t)(a.r.step >> 32)};
          const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
          const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
          for (int qd = 0; qd < 4; ++qd)
            if (4 * blk + qd < 29) u[4 * blk + qd] = u01(w[qd]);
        }
      }
#pragma unroll
      for (int j = 0; j < AD; ++j) {
        dp[j] = p.default_dof_pos[j] * (1.0f * u[j] + 0.5f);  // :321,:324  U(0.5,1.5)
        dv[j] = 0.2f * u[AD + j] + (-0.1f);                    // :322,:325  U(-0.1,0.1)
      }
      const int64_t type = a.b.terrain_types[env];
      f3 origin{a.b.env_origins[env * 3], a.b.env_origins[env * 3 + 1], a.b.env_origins[env * 3 + 2]};
      if (p.init_done && p.curriculum) {  // update_terrain_level :369-380
        const float gnorm = a.b.reduce_workspace[T * RW + 0];
        const float dist = norm3(f3{bpos.x - origin.x, bpos.y - origin.y, 0.0f});
        level -= dist < (gnorm * p.max_episode_length_s) * 0.25f ? 1 : 0;
        level += dist > p.half_env_length ? 1 : 0;
        level = (level < 0 ? 0 : level) % p.env_rows;
        const float* to = a.s.terrain_origins + (level * p.env_cols + type) * 3;
        origin = f3{to[0], to[1], to[2]};
        a.b.terrain_levels[env] = level;
        a.b.env_origins[env * 3] = origin.x; a.b.env_origins[env * 3 + 1] = origin.y; a.b.env_origins[env * 3 + 2] = origin.z;
      }
      hpos = f3{(p.base_init_state[0] + origin.x) + (1.0f * u[24] + (-0.5f)),
                (p.base_init_state[1] + origin.y) + (1.0f * u[25] + (-0.5f)), p.base_init_state[2] + origin.z};  // :328-330
      hq = q4{p.base_init_state[3], p.base_init_state[4], p.base_init_state[5], p.base_init_state[6]};           // :331
      a.r.base_pos_out[k * 3] = hpos.x; a.r.base_pos_out[k * 3 + 1] = hpos.y; a.r.base_pos_out[k * 3 + 2] = hpos.z;
      *reinterpret_cast<float4*>(a.r.base_quat_out + k * 4) = make_float4(hq.w, hq.x, hq.y, hq.z);
#pragma unroll
      for (int c = 0; c < 6; ++c) a.r.base_vel_out[k * 6 + c] = p.base_init_state[7 + c];  // :332
      float4* dpo = reinterpret_cast<float4*>(a.r.dof_pos_out + k * AD);
      float4* dvo = reinterpret_cast<float4*>(a.r.dof_vel_out + k * AD);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        dpo[c] = make_float4(dp[4 * c], dp[4 * c + 1], dp[4 * c + 2], dp[4 * c + 3]);
        dvo[c] = make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]);
      }
      // commands (:341-352)
      cmd[0] = p.cmd_span[0] * u[26] + p.cmd_lo[0];
      cmd[1] = p.cmd_span[1] * u[27] + p.cmd_lo[1];
      cmd[3] = p.cmd_span[2] * u[28] + p.cmd_lo[2];
      const float keep = norm3(f3{cmd[0], cmd[1], 0.0f}) > 0.25f ? 1.0f : 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) cmd[c] *= keep;
      *reinterpret_cast<float4*>(a.b.feet_air_time + env * 4) = make_float4(0.f, 0.f, 0.f, 0.f);  // :356
      a.b.progress_buf[env] = 0;                                                                    // :357
      a.b.reset_buf[env] = 1;  // :358 (already 1 inside post_physics_step)
      a.r.env_ids[k] = env;
    }
    // per-tile sums for extras (:361-367): episode sums of the reset envs BEFORE they are zeroed
#pragma unroll
    for (int k = 0; k < AE; ++k) {
      const float part = wave_sum_ff(reset ? es[k] : 0.f);
      if (tid == 0) a.b.reduce_workspace[t * RW + 1 + k] = part;
      if (reset) es[k] = 0.0f;  // :366
    }
    {
      const float lv = wave_sum_ff(live ? (float)level : 0.f);
      if (tid == 0) a.b.reduce_workspace[t * RW + 14] = lv;
    }
    if (live) {
#pragma unroll
      for (int k = 0; k < AE; ++k) a.b.episode_sums[(int64_t)k * a.n + env] = es[k];
      if (step) a.b.rew_buf[env] = rew;
      if (step || reset) *reinterpret_cast<float4*>(a.b.commands + env * 4) = make_float4(cmd[0], cmd[1], cmd[2], cmd[3]);
      // last_actions[:] = actions[:]; last_dof_vel[:] = dof_vel[:] -- the view row already holds the reset
      // velocities, so reset envs end the step with the NEW dof_vel, not 0 (:355 overwritten by :442)
      if (step || reset) {  // standalone reset_idx: last_actions = last_dof_vel = 0 for env_ids (:354-355)
        float4* lao = reinterpret_cast<float4*>(a.b.last_actions + env * AD);
        float4* ldo = reinterpret_cast<float4*>(a.b.last_dof_vel + env * AD);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          lao[c] = step ? make_float4(act[4 * c], act[4 * c + 1], act[4 * c + 2], act[4 * c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
          ldo[c] = step ? make_float4(dv[4 * c], dv[4 * c + 1], dv[4 * c + 2], dv[4 * c + 3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        }
      }
    }
    // observation head (:540-550)
    row[0] = lin.x * p.lin_vel_scale; row[1] = lin.y * p.lin_vel_scale; row[2] = lin.z * p.lin_vel_scale;
    row[3] = ang.x * p.ang_vel_scale; row[4] = ang.y * p.ang_vel_scale; row[5] = ang.z * p.ang_vel_scale;
    row[6] = grav.x; row[7] = grav.y; row[8] = grav.z;
    row[9] = cmd[0] * p.lin_vel_scale; row[10] = cmd[1] * p.lin_vel_scale; row[11] = cmd[2] * p.ang_vel_scale;
#pragma unroll
    for (int j = 0; j < AD; ++j) {
      row[C_DPOS + j] = dp[j] * p.dof_pos_scale;
      row[C_DVEL + j] = dv[j] * p.dof_vel_scale;
    }
    // yaw-only quaternion (:609-611): zero x,y then normalize (4-wide row: unfused sum)
    const float yn = norm4(q4{hq.w, 0.0f, 0.0f, hq.z});
    const float yd = yn < 1e-9f ? 1e-9f : yn;
    *reinterpret_cast<float4*>(s_pose + tid * 4) = make_float4(hq.w / yd, hq.z / yd, hpos.x, hpos.y);
    s_pz[tid] = hpos.z - 0.5f;  // (:538) base z - 0.5, the same subtraction for each of the env's 140 points
  }
  if (!step) return;  // uniform across the workgroup
  __syncthreads();

  // ---- height scan (:582-604, :537-539): 64 x 140 points over 256 threads
#ifndef OIGE_EXP_NO_SCAN
  // quat_apply(q_yaw, (hx, hy, 0)) with q_yaw = (w, 0, 0, z) (:608-612), written out with the terms that are exactly
  // zero removed (the surviving products, sums and their order are oige_math.hpp's quat_apply: t = 2 (qv x v);
  // v + w t + qv x t), so every finite result is bit-identical:
  //   t = (-(2 z hy), 2 z hx, 0)      qv x t = (-(z t.y), z t.x, 0)
  const int hs_cols = p.hs_cols, max_r = p.hs_rows - 2, max_c = p.hs_cols - 2;
#pragma unroll 2
  for (int pidx = tid; pidx < TILE * AP; pidx += ANT) {
    const int e = pidx / AP, k = pidx - e * AP;
    const float hx = s_hx[k], hy = s_hy[k];
    const float4 ps = *reinterpret_cast<const float4*>(s_pose + e * 4);  // w, z, px, py
    const float tx = -((ps.y * hy) * 2.0f), ty = (ps.y * hx) * 2.0f;
    const float rx = (hx + ps.x * tx) + (-(ps.y * ty));
    const float ry = (hy + ps.x * ty) + (ps.y * tx);
    const float wx = ((rx + ps.z) + p.border_size) / p.horizontal_scale;  // :586-593
    const float wy = ((ry + ps.w) + p.border_size) / p.horizontal_scale;
    // .long() truncates toward zero; v_cvt_i32_f32 does too and saturates, and the clip that follows (:596-597) maps
    // every saturated value where int64 truncation + clip would (out-of-range low -> 0, high -> rows - 2)
    int px = (int)wx, py = (int)wy;
    px = px < 0 ? 0 : (px > max_r ? max_r : px);
    py = py < 0 ? 0 : (py > max_c ? max_c : py);
    const int idx = px * hs_cols + py;
    const int16_t h1 = a.s.height_samples[idx];
    const int16_t h2 = a.s.height_samples[idx + hs_cols + 1];
    const float mh = (float)(h1 < h2 ? h1 : h2) * p.vertical_scale;
    float hv = s_pz[e] - mh;
    hv = (hv != hv) ? hv : (hv < -1.0f ? -1.0f : (hv > 1.0f ? 1.0f : hv));
    obs_t[e * AO + 36 + k] = hv * p.height_meas_scale;
  }
#endif
  __syncthreads();

  // ---- D: obs tile -> HBM with the uniform noise (:438-439)
  constexpr int NF4 = TILE * AO / 4;
#pragma unroll
  for (int it = 0; it < (NF4 + ANT - 1) / ANT; ++it) {
    const int f = it * ANT + tid;
    if (f >= NF4) continue;
    const int e = 4 * f;
    if (!FULLTILE && e >= nv * AO) continue;  // AO % 4 == 0: a float4 is wholly valid or wholly not
    float4 v = *reinterpret_cast<const float4*>(obs_t + e);
#ifndef OIGE_EXP_NO_NOISE
    if (p.add_noise) {
      const int c = e % AO;  // 4 consecutive columns of one row; region boundaries are multiples of 4 except 3/6/9
      float uu[4];
      const int64_t gf = (env0 * AO + e) / 4;
      if (a.r.noise_u) {
        const float4 t4 = *reinterpret_cast<const float4*>(a.r.noise_u + env0 * AO + e);
        uu[0] = t4.x; uu[1] = t4.y; uu[2] = t4.z; uu[3] = t4.w;
      } else {
        const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | (3u << 16), (uint32_t)a.r.step, (uint32_t)(a.r.step >> 32)};
        const u4 x = philox4x32_10(cc, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
        uu[0] = u01(x.x); uu[1] = u01(x.y); uu[2] = u01(x.z); uu[3] = u01(x.w);
      }
      const float4 sc = *reinterpret_cast<const float4*>(s_ns + c);
      float vv[4] = {v.x, v.y, v.z, v.w};
      const float scv[4] = {sc.x, sc.y, sc.z, sc.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) vv[k] = vv[k] + (2.0f * uu[k] - 1.0f) * scv[k];
      v = make_float4(vv[0], vv[1], vv[2], vv[3]);
    }
#endif
    *reinterpret_cast<float4*>(a.b.obs_buf + env0 * AO + e) = v;
  }
}

__global__ __launch_bounds__(ANT) void anymal_post_kernel(const AnymalPost a) {
  extern __shared__ __align__(16) float lds[];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    anymal_post_tile<true>(a, lds, t, TILE);
  else
    anymal_post_tile<false>(a, lds, t, (int)left);
}

// extras (:361-367): mean over the reset envs of each episode sum / max_episode_length_s, mean terrain level
__global__ __launch_bounds__(1024) void anymal_extras_kernel(const AnymalPost a) {
  __shared__ double part[1024 / WAVE][AE + 1];
  const int64_t T = num_tiles(a.n);
  double acc[AE + 1];
#pragma unroll
  for (int k = 0; k <= AE; ++k) acc[k] = 0.0;
  for (int64_t i = threadIdx.x; i < T; i += 1024) {
#pragma unroll
    for (int k = 0; k <= AE; ++k) acc[k] += (double)a.b.reduce_workspace[i * RW + 1 + k];
  }
#pragma unroll
  for (int k = 0; k <= AE; ++k) {
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) acc[k] += __shfl_xor(acc[k], off, WAVE);
    if (lane_id() == 0) part[threadIdx.x >> 6][k] = acc[k];
  }
  __syncthreads();
  if (threadIdx.x <= AE) {
    const int k = threadIdx.x;
    double tot = 0.0;
    for (int i = 0; i < 1024 / WAVE; ++i) tot += part[i][k];
    const int64_t G = num_groups(a.n);
    const int32_t count = a.b.reset_workspace[padded_tiles(a.n) + G];
    if (k < AE) {
      if (count > 0) a.b.extras[k] = ((float)tot / (float)count) / a.p.max_episode_length_s;
    } else if (count > 0) {
      a.b.extras[AE] = (float)tot / (float)a.n;
    }
    if (k == 0) *a.r.count = (int64_t)count;
  }
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_anymal_torques(const oige_anymal_params* params, const float* actions, const float* dof_pos,
                        const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream) {
  OIGE_REQUIRE(params && actions && dof_pos && dof_vel && torques_out, "anymal_torques: null pointer");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 256, "anymal_torques: num_envs %lld out of range", (long long)num_envs);
  OIGE_REQUIRE(aligned16(actions) && aligned16(dof_pos) && aligned16(dof_vel) && aligned16(torques_out),
               "anymal_torques: buffers must be 16-byte aligned");
  const int64_t total = num_envs * AD;
  if (total == 0) return OIGE_OK;
  const int64_t f4 = total / 4;
  hipLaunchKernelGGL(anymal_torques_kernel, dim3((unsigned)((f4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *params,
                     actions, dof_pos, dof_vel, torques_out, total);
  return check_launch("anymal_torques_kernel");
}

static int anymal_launch(const oige_anymal_params* params, const oige_anymal_state* state,
                         const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io, int64_t num_envs,
                         int mode, oige_stream_t stream) {
  OIGE_REQUIRE(params && state && buffers && reset_io, "anymal: null struct pointer");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 256, "anymal: num_envs %lld out of range", (long long)num_envs);
  const oige_anymal_state& s = *state;
  const oige_anymal_buffers& b = *buffers;
  const oige_anymal_reset_io& r = *reset_io;
  OIGE_REQUIRE(s.base_pos && s.base_quat && s.terrain_origins, "anymal: null base pose / terrain_origins pointer");
  if (mode == 0)
    OIGE_REQUIRE(s.base_velocities && s.dof_pos && s.dof_vel && s.knee_contact && s.base_contact && s.height_samples &&
                     b.obs_buf && b.rew_buf && b.has_fallen && b.timeout_buf && b.actions && b.torques,
                 "anymal_post_physics: null state/buffer pointer");
  else
    OIGE_REQUIRE(r.reset_mask, "anymal_reset_idx: null reset_mask");
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.commands && b.last_actions && b.last_dof_vel && b.feet_air_time &&
                   b.episode_sums && b.terrain_levels && b.terrain_types && b.env_origins && b.extras && b.reset_workspace &&
                   b.reduce_workspace,
               "anymal: null buffer pointer");
  OIGE_REQUIRE(r.env_ids && r.count && r.base_pos_out && r.base_quat_out && r.base_vel_out && r.dof_pos_out && r.dof_vel_out,
               "anymal: null reset output pointer");
  const bool drawn = r.u_dof_offset != nullptr;
  OIGE_REQUIRE(drawn == (r.u_dof_vel != nullptr) && drawn == (r.u_xy != nullptr) && drawn == (r.u_cmd != nullptr),
               "anymal: pass all four pre-drawn reset tables or none");
  OIGE_REQUIRE(aligned16(s.base_quat) && aligned16(s.dof_pos) && aligned16(s.dof_vel) && aligned16(s.knee_contact) &&
                   aligned16(b.obs_buf) && aligned16(b.commands) && aligned16(b.actions) && aligned16(b.last_actions) &&
                   aligned16(b.last_dof_vel) && aligned16(b.torques) && aligned16(b.feet_air_time) && aligned16(r.base_quat_out) &&
                   aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out) && aligned16(b.reset_workspace) &&
                   aligned16(r.noise_u) && (reinterpret_cast<uintptr_t>(s.base_velocities) & 7u) == 0,
               "anymal: row buffers must be 16-byte aligned (base_velocities 8)");
  OIGE_REQUIRE(params->hs_rows >= 2 && params->hs_cols >= 2 && params->env_rows >= 1 && params->env_cols >= 1,
               "anymal: bad terrain dimensions");
  if (num_envs == 0) return OIGE_OK;
  AnymalPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.r = r;
  a.n = num_envs;
  a.mode = mode;
  hipStream_t st = (hipStream_t)stream;
  const int64_t G = num_groups(num_envs), T = num_tiles(num_envs);
  hipLaunchKernelGGL(anymal_flags_kernel, dim3((unsigned)G), dim3(GROUP), 0, st, a);
  if (int rc = check_launch("anymal_flags_kernel")) return rc;
  if (int rc = launch_scan_groups(num_envs, b.reset_workspace, st)) return rc;
  hipLaunchKernelGGL(anymal_command_norm_kernel, dim3(1), dim3(1024), 0, st, b.reduce_workspace, T);
  if (int rc = check_launch("anymal_command_norm_kernel")) return rc;
  const size_t lds = sizeof(float) * (TILE * AO + TILE * 4 + TILE + 2 * AP + AO);
  hipLaunchKernelGGL(anymal_post_kernel, dim3((unsigned)T), dim3(ANT), lds, st, a);
  if (int rc = check_launch("anymal_post_kernel")) return rc;
  hipLaunchKernelGGL(anymal_extras_kernel, dim3(1), dim3(1024), 0, st, a);
  return check_launch("anymal_extras_kernel");
}

int oige_anymal_post_physics(const oige_anymal_params* params, const oige_anymal_state* state,
                             const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io, int64_t num_envs,
                             oige_stream_t stream) {
  return anymal_launch(params, state, buffers, reset_io, num_envs, 0, stream);
}

int oige_anymal_reset_idx(const oige_anymal_params* params, const oige_anymal_state* state,
                          const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io, int64_t num_envs,
                          oige_stream_t stream) {
  return anymal_launch(params, state, buffers, reset_io, num_envs, 1, stream);
}

}  // extern "C"
