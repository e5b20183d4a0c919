// Ant / Humanoid per-step task math for gfx950 (MI355X): hand-written HIP, one wavefront per
// tile of 64 envs, HBM-bound (no MFMA: ~0.3 flop/B).
//
// Replaces, behind the unchanged LocomotionTask method contract, the reference's
//   post_physics_step  tasks/base/rl_task.py:278-298          (progress += 1, then ...)
//   get_observations   tasks/shared/locomotion.py:75-115, jit :225-284
//   calculate_metrics  locomotion.py:190-206, jit :296-345; ant.py:107-109; humanoid.py:135-141
//   is_done            locomotion.py:208-211, jit :288-292
//   pre_physics_step   locomotion.py:117-131
//   reset_idx          locomotion.py:133-158
//
// post_physics kernel, per tile (algorithmic HBM bytes per env: Ant 532 (+24 with up/heading
// vectors), Humanoid 748 (+24)):
//   A. issue every global load of the tile up front: per-env scalars/rows (pos, rot, vel,
//      potentials, reset, progress) one env per lane, and the four wide arrays (dof_pos,
//      dof_vel, sensors, actions) as lane-linear float4 runs (1 KiB per wave-instruction);
//   B. scale the pass-through runs and scatter them into a (64 x num_obs) LDS obs tile;
//   C. one env per lane: quaternion/frame math -> obs[0:12] into the tile; unscale dof_pos in
//      place in the tile; reward and done from the tile row and registers; coalesced scalar
//      stores of rew/reset/progress/potentials;
//   D. stream the finished obs tile to HBM as one contiguous run of float4 stores.
// The obs row never makes an HBM round trip between obs, reward and done.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

struct LocoPost {
  oige_locomotion_params p;
  oige_locomotion_state s;
  oige_locomotion_buffers b;
  int64_t n;
  uint32_t stages;
  float max_len_minus_1;  // (float)((double)max_episode_length - 1.0): `progress >= max_len - 1`
};

template <int D>
__device__ __forceinline__ float limit_cost_ant(const float (&o)[D]) {
  int cnt = 0;  // ant.py:109: sum(obs > 0.99) -- no abs, int64 count, promoted to fp32 at the subtraction
#pragma unroll
  for (int j = 0; j < D; ++j) cnt += o[j] > 0.99f ? 1 : 0;
  return (float)cnt;
}

template <int D>
__device__ __forceinline__ float limit_cost_humanoid(const float (&o)[D], const oige_locomotion_params& p) {
  float acc = 0.0f;  // humanoid.py:137-140
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float a = __builtin_fabsf(o[j]);
    const float scaled = (p.joints_at_limit_cost_scale * (a - 0.98f)) / 0.02f;
    const float term = ((a > 0.98f ? 1.0f : 0.0f) * scaled) * p.motor_effort_ratio[j];
    acc = j == 0 ? term : acc + term;
  }
  return acc;
}

// reward + done for one env from its obs row pieces (locomotion.py:314-345, :290-291)
// The per-DOF sums of the reward (locomotion.py:322-333; get_dof_at_limit_cost): actions_cost, electricity_cost, the limit cost
template <int D, int LIMIT>
__device__ __forceinline__ void dof_costs(const oige_locomotion_params& p, const float (&dps)[D], const float (&dvs)[D],
                                          const float (&act)[D], float& actions_cost, float& electricity_cost, float& lim) {
  actions_cost = 0.0f;
  electricity_cost = 0.0f;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float a2 = act[j] * act[j];
    const float el = __builtin_fabsf(act[j] * dvs[j]) * p.motor_effort_ratio[j];
    actions_cost = j == 0 ? a2 : actions_cost + a2;
    electricity_cost = j == 0 ? el : electricity_cost + el;
  }
  lim = LIMIT == 0 ? limit_cost_ant<D>(dps) : limit_cost_humanoid<D>(dps, p);
}

// ... and the rest of it, from the head quantities and those three sums
__device__ __forceinline__ float reward_total(const oige_locomotion_params& p, float obs0, float up_proj, float heading_proj, float pot,
                                              float prev_pot, float actions_cost, float electricity_cost, float lim) {
  const float hw = p.heading_weight;
  const float heading_reward = heading_proj > 0.8f ? hw : (hw * heading_proj) / 0.8f;
  const float up_reward = up_proj > 0.93f ? (0.0f + p.up_weight) : 0.0f;
  const float progress_reward = pot - prev_pot;
  float total = progress_reward + p.alive_reward_scale;
  total = total + up_reward;
  total = total + heading_reward;
  total = total - p.actions_cost_scale * actions_cost;
  total = total - p.energy_cost_scale * electricity_cost;
  total = total - lim;
  return obs0 < p.termination_height ? p.death_cost : total;
}

template <int D, int LIMIT>
__device__ __forceinline__ float reward_of(const oige_locomotion_params& p, float obs0, float up_proj,
                                           float heading_proj, const float (&dps)[D], const float (&dvs)[D],
                                           const float (&act)[D], float pot, float prev_pot) {
  float actions_cost, electricity_cost, lim;
  dof_costs<D, LIMIT>(p, dps, dvs, act, actions_cost, electricity_cost, lim);
  return reward_total(p, obs0, up_proj, heading_proj, pot, prev_pot, actions_cost, electricity_cost, lim);
}

// Threads per 64-env tile in the post kernel.  One wave per tile while ten tiles fit a CU's LDS (Ant: 15.4 KB); two waves
// share a bigger tile (Humanoid: 22.3 KB, seven per CU), which keeps as many waves resident: 76.6 -> 72.9 us at 512 K envs,
// Ant unchanged either way, four waves slower on both.  -DOIGE_POST_NT=<n>: A/B builds.
template <int D, int S>
constexpr int post_threads() {
#ifdef OIGE_POST_NT
  return OIGE_POST_NT;
#else
  return (12 + 3 * D + 6 * S) * TILE * 4 > 16 * 1024 ? 2 * WAVE : WAVE;
#endif
}

template <int D, int S, int LIMIT, bool FULL>
__device__ __forceinline__ void post_tile(const LocoPost& a, float* __restrict__ tile, float* __restrict__ s_sums, int64_t t, int nv) {
  constexpr int OBS = 12 + 3 * D + 6 * S;
  constexpr int PNT = post_threads<D, S>();
  constexpr int C_DPOS = 12, C_DVEL = 12 + D, C_SENS = 12 + 2 * D, C_ACT = 12 + 2 * D + 6 * S;
  const oige_locomotion_params& p = a.p;
  const int lane = lane_id();
  const bool env_wave = PNT == WAVE || threadIdx.x < WAVE;  // the wave that does the one-env-per-lane part
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = env_wave && (FULL || lane < nv);
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS;
  const bool do_done = a.stages & OIGE_STAGE_DONE;

  // ---- A: all loads in flight
  Chunk<D, PNT> c_dpos, c_dvel, c_act;
  Chunk<6 * S, PNT> c_sens;
  load_chunk<D, FULL, PNT>(c_dpos, a.s.dof_pos + env0 * D, nv * D);
  load_chunk<D, FULL, PNT>(c_dvel, a.s.dof_vel + env0 * D, nv * D);
  load_chunk<6 * S, FULL, PNT>(c_sens, a.s.sensor_force_torques + env0 * (6 * S), nv * 6 * S);
  load_chunk<D, FULL, PNT>(c_act, a.b.actions + env0 * D, nv * D);

  f3 pos{0.f, 0.f, 1.f}, lin{0.f, 0.f, 0.f}, ang{0.f, 0.f, 0.f};
  q4 rot{1.f, 0.f, 0.f, 0.f};
  float pot_in = 0.f;
  int64_t reset_in = 0, progress = 0;
  if (live) {
    const float* pp = a.s.torso_position + env * 3;
    pos = f3{pp[0], pp[1], pp[2]};
    const float4 r4 = *reinterpret_cast<const float4*>(a.s.torso_rotation + env * 4);
    rot = q4{r4.x, r4.y, r4.z, r4.w};
    const float2* vp = reinterpret_cast<const float2*>(a.s.velocities + env * 6);
    const float2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
    lin = f3{v0.x, v0.y, v1.x};
    ang = f3{v1.y, v2.x, v2.y};
    pot_in = a.b.potentials[env];
    if (do_done) reset_in = a.b.reset_buf[env];
    if (do_progress || do_done) progress = a.b.progress_buf[env];
  }
  // the tile's episode-statistics slot (oige_hip.h: stats_slots), fetched with the other loads by the lane that owns it
  const bool keep_stats = a.b.stats_slots != nullptr && do_metrics && do_done;
  double st_rew = 0.0, st_done = 0.0, st_len = 0.0;
  if (keep_stats && env_wave && lane == 0) {
    const double* sl = a.b.stats_slots + 4 * t;
    const double2 lo = *reinterpret_cast<const double2*>(sl);
    st_rew = lo.x; st_done = lo.y; st_len = sl[2];
  }

  // ---- B: pass-through columns into the obs tile
  scatter_chunk<D, OBS, C_DPOS, false, PNT>(c_dpos, tile, 1.0f);  // raw; unscaled per env in C
  scatter_chunk<D, OBS, C_DVEL, true, PNT>(c_dvel, tile, p.dof_vel_scale);
  scatter_chunk<6 * S, OBS, C_SENS, true, PNT>(c_sens, tile, p.contact_force_scale);
  scatter_chunk<D, OBS, C_ACT, false, PNT>(c_act, tile, 1.0f);

  // ---- C: per-env math (locomotion.py:249-263), one env per lane.  One-wave tiles (Ant): the wave does all of it.  Two-wave
  // tiles (Humanoid): the first wave takes the head -- quaternion products, Euler angles, potentials -- while the second,
  // which used to idle through this phase, takes the 3 x D pass over the env's DOF columns (unscale, the two cost sums, the
  // limit cost) and hands the three sums over through LDS; the phase then lasts as long as the longer half, not their sum
  // (r03: the tile's LDS footprint holds a CU to seven tiles, so the kernel's rate is tiles x bytes / tile lifetime).  The
  // operations and their order are those of reward_of: bit-identical.
  constexpr bool SPLIT = PNT != WAVE;
  if (SPLIT) __syncthreads();  // B's cross-wave LDS writes visible
  if (SPLIT && !env_wave) {
    float* row = tile + lane * OBS;
    float dps[D], dvs[D], act[D];
#pragma unroll
    for (int j = 0; j < D; ++j) {
      dps[j] = unscale(row[C_DPOS + j], p.dof_limits_lower[j], p.dof_limits_upper[j]);  // :263
      dvs[j] = row[C_DVEL + j];
      act[j] = row[C_ACT + j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) row[C_DPOS + j] = dps[j];
    if (do_metrics) {
      float ac, ec, lim;
      dof_costs<D, LIMIT>(p, dps, dvs, act, ac, ec, lim);
      s_sums[lane] = ac;
      s_sums[TILE + lane] = ec;
      s_sums[2 * TILE + lane] = lim;
    }
  }
  float rew = 0.0f, up_proj = 0.f, heading_proj = 0.f, pot = 0.f, prev_pot = 0.f, obs0 = 0.f;
  float dps[SPLIT ? 1 : D], dvs[SPLIT ? 1 : D], act[SPLIT ? 1 : D];
  if (env_wave) {
  if (do_progress) progress += 1;  // rl_task.py:289, before is_done reads it

  const f3 target{p.target[0], p.target[1], p.target[2]};
  f3 to_target{target.x - pos.x, target.y - pos.y, 0.0f};  // :249-250
  prev_pot = pot_in;                                       // :252
  pot = (-norm3(to_target)) / p.dt;                        // :253

  const q4 inv_start{p.inv_start_rot[0], p.inv_start_rot[1], p.inv_start_rot[2], p.inv_start_rot[3]};
  const f3 target_dir = normalize3(to_target);
  const q4 tq = quat_mul(rot, inv_start);
  const f3 up_vec = quat_rotate(tq, f3{p.basis_vec1[0], p.basis_vec1[1], p.basis_vec1[2]});
  const f3 heading_vec = quat_rotate(tq, f3{p.basis_vec0[0], p.basis_vec0[1], p.basis_vec0[2]});
  up_proj = up_vec.z;
  heading_proj = dot3(heading_vec, target_dir);
  const f3 vel_loc = quat_rotate_inverse(tq, lin);
  const f3 angvel_loc = quat_rotate_inverse(tq, ang);
  float roll, pitch, yaw;
  get_euler_xyz(tq, roll, pitch, yaw);
  (void)pitch;
  const float walk_target_angle = atan2f(target.z - pos.z, target.x - pos.x);  // z, not y (warp/shared/locomotion.py:347)
  const float angle_to_target = walk_target_angle - yaw;

  float head[12];
  head[0] = pos.z;
  head[1] = vel_loc.x; head[2] = vel_loc.y; head[3] = vel_loc.z;
  head[4] = angvel_loc.x * p.angular_velocity_scale;
  head[5] = angvel_loc.y * p.angular_velocity_scale;
  head[6] = angvel_loc.z * p.angular_velocity_scale;
  head[7] = normalize_angle(yaw);
  head[8] = normalize_angle(roll);
  head[9] = normalize_angle(angle_to_target);
  head[10] = up_proj;
  head[11] = heading_proj;
  obs0 = head[0];

  if (!SPLIT) __syncthreads();  // B's cross-lane LDS writes visible (single-wave workgroup: a wave-local fence)

  float* row = tile + lane * OBS;
  if (OBS % 4 == 0) {
#pragma unroll
    for (int k = 0; k < 3; ++k)
      *reinterpret_cast<float4*>(row + 4 * k) = make_float4(head[4 * k], head[4 * k + 1], head[4 * k + 2], head[4 * k + 3]);
  } else {
#pragma unroll
    for (int k = 0; k < 12; ++k) row[k] = head[k];
  }
  if (!SPLIT) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      dps[j] = unscale(row[C_DPOS + j], p.dof_limits_lower[j], p.dof_limits_upper[j]);  // :263
      dvs[j] = row[C_DVEL + j];
      act[j] = row[C_ACT + j];
    }
#pragma unroll
    for (int j = 0; j < D; ++j) row[C_DPOS + j] = dps[j];
  }

  if (live) {
    a.b.potentials[env] = pot;
    a.b.prev_potentials[env] = prev_pot;
    if (a.b.up_vec) {
      float* u = a.b.up_vec + env * 3;
      u[0] = up_vec.x; u[1] = up_vec.y; u[2] = up_vec.z;
    }
    if (a.b.heading_vec) {
      float* h = a.b.heading_vec + env * 3;
      h[0] = heading_vec.x; h[1] = heading_vec.y; h[2] = heading_vec.z;
    }
    if (do_progress) a.b.progress_buf[env] = progress;
  }
  }  // env_wave: head
  if (SPLIT && do_metrics) __syncthreads();  // the second wave's three sums (do_metrics is uniform)
  if (env_wave) {
  if (do_metrics) {
    if (SPLIT) {
      rew = reward_total(p, obs0, up_proj, heading_proj, pot, prev_pot, s_sums[lane], s_sums[TILE + lane], s_sums[2 * TILE + lane]);
    } else {
      float ac, ec, lim;
      dof_costs<D, LIMIT>(p, reinterpret_cast<const float (&)[D]>(dps), reinterpret_cast<const float (&)[D]>(dvs),
                          reinterpret_cast<const float (&)[D]>(act), ac, ec, lim);
      rew = reward_total(p, obs0, up_proj, heading_proj, pot, prev_pot, ac, ec, lim);
    }
    if (live) a.b.rew_buf[env] = rew;
  }
  if (do_done) {
    // is_done (:290-291): sticky; progress compared in fp32 against float(max_len - 1)
    int64_t r = obs0 < p.termination_height ? 1 : reset_in;
    r = (float)progress >= a.max_len_minus_1 ? 1 : r;
    if (live) a.b.reset_buf[env] = r;
    const unsigned long long m = __ballot(live && r != 0);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
    if (keep_stats) {  // this step's share of the rank's episode statistics: lane order, then += into the tile's own slot
      double rs = live ? (double)rew : 0.0;
      int ls = (live && r != 0) ? (int)progress : 0;
#pragma unroll
      for (int off = WAVE / 2; off > 0; off >>= 1) {
        rs += __shfl_xor(rs, off, WAVE);
        ls += __shfl_xor(ls, off, WAVE);
      }
      if (lane == 0) {
        double* sl = a.b.stats_slots + 4 * t;
        *reinterpret_cast<double2*>(sl) = make_double2(st_rew + rs, st_done + (double)__popcll(m));
        sl[2] = st_len + (double)ls;
      }
    }
  }
  }  // env_wave

  __syncthreads();  // C's row writes visible to the lane-linear readout

  // ---- D: finished obs tile -> HBM, one contiguous run
  store_tile<OBS, FULL, PNT>(tile, a.b.obs_buf + env0 * OBS, nv * OBS);
}

template <int D, int S, int LIMIT>
__global__ __launch_bounds__((post_threads<D, S>())) void locomotion_post_kernel(const LocoPost a) {
  constexpr int OBS = 12 + 3 * D + 6 * S;
  __shared__ __align__(16) float tile[TILE * OBS];
  __shared__ float s_sums[post_threads<D, S>() != WAVE ? 3 * TILE : 1];  // two-wave tiles: the second wave's three sums per env
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    post_tile<D, S, LIMIT, true>(a, tile, s_sums, t, TILE);
  else
    post_tile<D, S, LIMIT, false>(a, tile, s_sums, t, (int)left);
}

// ---------------------------------------------------------------- split-stage kernels
// Used when the host calls calculate_metrics() / is_done() / progress on their own (the
// RLTask methods stay individually callable); they read obs_buf rows back from memory the
// way the reference's separate jit functions do.  Not the hot path.
template <int D, int S, int LIMIT>
__global__ __launch_bounds__(256) void locomotion_metrics_kernel(const LocoPost a) {
  constexpr int OBS = 12 + 3 * D + 6 * S;
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= a.n) return;
  const float* row = a.b.obs_buf + env * OBS;
  float dps[D], dvs[D], act[D];
#pragma unroll
  for (int j = 0; j < D; ++j) {
    dps[j] = row[12 + j];
    dvs[j] = row[12 + D + j];
    act[j] = a.b.actions[env * D + j];
  }
  a.b.rew_buf[env] = reward_of<D, LIMIT>(a.p, row[0], row[10], row[11], dps, dvs, act, a.b.potentials[env],
                                         a.b.prev_potentials[env]);
}

// get_dof_at_limit_cost() as a tensor of its own (ant.py:102-109 -> (N,) int64 count, humanoid.py:130-141 -> (N,) fp32):
// the METRICS stage evaluates the same functions in registers; callers of the public method get them from obs_buf.
template <int D, int S, int LIMIT>
__global__ __launch_bounds__(256) void locomotion_limit_cost_kernel(const oige_locomotion_params p, const float* __restrict__ obs_buf,
                                                                    void* __restrict__ out, int64_t n) {
  constexpr int OBS = 12 + 3 * D + 6 * S;
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env >= n) return;
  float dps[D];
#pragma unroll
  for (int j = 0; j < D; ++j) dps[j] = obs_buf[env * OBS + 12 + j];
  if (LIMIT == 0)
    static_cast<int64_t*>(out)[env] = (int64_t)limit_cost_ant<D>(dps);  // an exact small integer in fp32
  else
    static_cast<float*>(out)[env] = limit_cost_humanoid<D>(dps, p);
}

__global__ __launch_bounds__(256) void locomotion_done_kernel(const LocoPost a, int obs_width) {
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = env < a.n;
  int64_t r = 0;
  if (live) {
    r = a.b.obs_buf[env * obs_width] < a.p.termination_height ? 1 : a.b.reset_buf[env];
    r = (float)a.b.progress_buf[env] >= a.max_len_minus_1 ? 1 : r;
    a.b.reset_buf[env] = r;
  }
  const unsigned long long m = __ballot(live && r != 0);
  if (a.b.reset_workspace && lane_id() == 0) a.b.reset_workspace[env / TILE] = __popcll(m);
  if (a.b.tile_counts_u8 && lane_id() == 0) a.b.tile_counts_u8[env / TILE] = (uint8_t)__popcll(m);
}

__global__ __launch_bounds__(256) void progress_increment_kernel(int64_t* __restrict__ progress, int64_t n) {
  const int64_t env = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (env < n) progress[env] += 1;
}

// ---------------------------------------------------------------- pre_physics / reset
struct LocoPre {
  oige_locomotion_params p;
  oige_locomotion_buffers b;
  oige_locomotion_reset_io r;
  const float* actions_in;
  float* actions_out;
  float* forces_out;
  int64_t n;
  uint32_t flags;
};

// the rows reset_idx reads for env i (locomotion.py:133-158), fetched apart from the arithmetic so that a caller can issue them
// together
template <int D>
struct ResetRow {
  float dof0[D];
  float ip[3];
  float4 iq;
};

template <int D>
__device__ __forceinline__ ResetRow<D> reset_load(const oige_locomotion_reset_io& r, int64_t i) {
  ResetRow<D> w;
#pragma unroll
  for (int j = 0; j < D; ++j) w.dof0[j] = r.initial_dof_pos[i * D + j];
#pragma unroll
  for (int j = 0; j < 3; ++j) w.ip[j] = r.initial_root_pos[i * 3 + j];
  w.iq = *reinterpret_cast<const float4*>(r.initial_root_rot + i * 4);
  return w;
}

// the 2*D uniform draws of reset_idx for env i / compacted row k: [0,D) -> dof_pos (1st torch.rand, :137), [D,2D) -> dof_vel (2nd, :139)
template <int D>
__device__ __forceinline__ void reset_draws(const oige_locomotion_reset_io& r, int64_t i, int64_t k, float (&u)[2 * D]) {
  if (r.u_pos != nullptr) {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      u[j] = r.u_pos[k * D + j];
      u[D + j] = r.u_vel[k * D + j];
    }
  } else {
    const uint64_t step = step_of(r);  // device counter: graph replay (include/oige_hip.h)
#pragma unroll
    for (int blk = 0; blk < (2 * D + 3) / 4; ++blk) {
      const u4 c{(uint32_t)i, (uint32_t)blk, (uint32_t)step, (uint32_t)(step >> 32)};
      const u4 x = philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
      const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (4 * blk + q < 2 * D) u[4 * blk + q] = u01(w[q]);
    }
  }
}

// the DOF rows of reset_idx (:137-139) from the draws
template <int D>
__device__ __forceinline__ void reset_dof_rows(const oige_locomotion_params& p, const float (&dof0)[D], const float (&u)[2 * D],
                                               float (&qp)[D], float (&qv)[D]) {
#pragma unroll
  for (int j = 0; j < D; ++j) {
    // torch_rand_float(lo, hi): (hi - lo) * u + lo, with hi - lo folded in double first (:137,:139)
    const float dp = 0.4f * u[j] + (-0.2f);
    qp[j] = tensor_clamp(dof0[j] + dp, p.dof_limits_lower[j], p.dof_limits_upper[j]);  // :138
    qv[j] = 0.2f * u[D + j] + (-0.1f);
  }
}

// prev_potentials[i] = potentials[i] of a reset env (:151-154)
__device__ __forceinline__ float reset_potential(const oige_locomotion_params& p, const float (&ip)[3]) {
  const f3 tt{p.target[0] - ip[0], p.target[1] - ip[1], 0.0f};  // :151-152
  return (-norm3(tt)) / p.dt;                                   // :153
}

// reset_idx for env `i` landing in compacted row `k` (locomotion.py:133-158): everything, one env per thread (the reference-flow
// entry point oige_locomotion_reset_idx; the pre kernel below splits the same work over natural and dense lanes)
template <int D>
__device__ __forceinline__ void reset_env(const oige_locomotion_params& p, const oige_locomotion_buffers& b,
                                          const oige_locomotion_reset_io& r, const ResetRow<D>& w, int64_t i, int64_t k) {
  float u[2 * D], qp[D], qv[D];
  reset_draws<D>(r, i, k, u);
  reset_dof_rows<D>(p, w.dof0, u, qp, qv);
  if (D % 4 == 0 && aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out)) {  // whole rows as 16-byte stores
#pragma unroll
    for (int c = 0; c < D / 4; ++c) {
      reinterpret_cast<float4*>(r.dof_pos_out + k * D)[c] = make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]);
      reinterpret_cast<float4*>(r.dof_vel_out + k * D)[c] = make_float4(qv[4 * c], qv[4 * c + 1], qv[4 * c + 2], qv[4 * c + 3]);
    }
  } else {
#pragma unroll
    for (int j = 0; j < D; ++j) {
      r.dof_pos_out[k * D + j] = qp[j];
      r.dof_vel_out[k * D + j] = qv[j];
    }
  }
  const float* ip = w.ip;
  r.root_pos_out[k * 3 + 0] = ip[0];
  r.root_pos_out[k * 3 + 1] = ip[1];
  r.root_pos_out[k * 3 + 2] = ip[2];
  *reinterpret_cast<float4*>(r.root_rot_out + k * 4) = w.iq;
#pragma unroll
  for (int c = 0; c < 6; ++c) r.root_vel_out[k * 6 + c] = 0.0f;  // :142
  const float pot = reset_potential(p, w.ip);
  b.prev_potentials[i] = pot;
  b.potentials[i] = pot;  // :154
  b.reset_buf[i] = 0;     // :157
  b.progress_buf[i] = 0;  // :158
  r.env_ids[k] = i;
}

// pre_physics_step of one 256-env group.  A workgroup's life is a chain of memory round trips (microseconds each under
// load), and the kernel's rate is resident workgroups x bytes per workgroup / that lifetime, so the chain is kept to TWO:
//   trip 1  the reset flags, the tile counts in front of the group and the action quads, issued together, nothing
//           conditional (idle lanes re-read a valid element);
//   then    rank of the flagged envs; forces / clone stores and the flags' own zeroing (natural lanes, coalesced);
//           groups without a flagged env end here;
//   trip 2  the reset rows of the group's flagged envs on DENSE lanes (thread j takes the j-th flagged env: a wave with a
//           few flagged lanes would run the Philox draws for all 64), overlapping the acknowledgement of the stores above;
//   then    the compacted output rows, staged in LDS and written as contiguous runs.
// SCANNED: the group offsets are in the workspace (scan_groups_kernel ran in front).  Otherwise the workgroup derives its
// offset from the tile counts itself: from the int32 counts (group_rank_unscanned, up to LOCO_UNSCANNED_MAX_GROUPS groups)
// or from their byte twin (up to UNSCANNED_U8_MAX_GROUPS).
constexpr int64_t LOCO_UNSCANNED_MAX_GROUPS = 2048;
// Forces and clone are written once: streaming stores.  The actions are read with a PLAIN load: the policy has just written
// them, and a 32 MB tensor that recent is still in the memory-side cache, which a non-temporal load does not use (Ant, 1 M
// envs: pre kernel 37.3 -> 34.1 us; profiles/r03/pre_variants.txt).  -DOIGE_LOCO_PRE_NT_LD / _PLAIN_{FORCES,CLONE}: A/B builds
__device__ __forceinline__ float4 pre_ld4(const float* p) {
#ifndef OIGE_LOCO_PRE_NT_LD
  return *reinterpret_cast<const float4*>(p);
#else
  return ld4_stream(p);
#endif
}
__device__ __forceinline__ void pre_st4_forces(float* p, const float4 t) {
#ifdef OIGE_LOCO_PRE_PLAIN_FORCES
  *reinterpret_cast<float4*>(p) = t;
#else
  st4_stream(p, t);
#endif
}
__device__ __forceinline__ void pre_st4_clone(float* p, const float4 t) {
#ifdef OIGE_LOCO_PRE_PLAIN_CLONE
  *reinterpret_cast<float4*>(p) = t;
#else
  st4_stream(p, t);
#endif
}
template <int D, int RANK, int PART = 0>  // PART (A/B builds): 0 the whole step, 1 forces / clone only, 2 compaction + resets only
__global__ __launch_bounds__(GROUP) void locomotion_pre_kernel(const LocoPre a) {
  constexpr int NF4 = GROUP * D / 4;  // float4 pieces of a full group's action run
  constexpr int ITS = (NF4 + GROUP - 1) / GROUP;
#ifdef OIGE_LOCO_PRE_STAGE_ALL  // A/B builds (profiles/r03/pre_variants.txt)
  constexpr bool STAGE = true;
#else
  constexpr bool STAGE = (D % 4) != 0;  // rows that are whole float4s are contiguous across dense lanes as they are
#endif
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP];
  __shared__ __align__(16) float s_gear[(D + 3) / 4 * 4];
  __shared__ __align__(16) float s_rows[STAGE ? GROUP * D : 4];
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t G = num_groups(a.n);
  const int64_t env = g * GROUP + tid;
  const int64_t base = g * GROUP * D;  // the group's contiguous run of 256*D floats; a multiple of D and of 4
  const int64_t total = a.n * D;
  const bool full = (g + 1) * GROUP <= a.n;

  // ---- trip 1
  const int64_t flag_raw = PART == 1 ? 0 : a.b.reset_buf[env < a.n ? env : a.n - 1];
  U8Prefix pre{};
  if (PART != 1) pre = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  float v[ITS][4];
  if (PART == 2) {
  } else if (full) {
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int f = it * GROUP + tid;
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (NF4 % GROUP == 0 || f < NF4) t = pre_ld4(a.actions_in + base + 4 * (int64_t)f);
      v[it][0] = t.x; v[it][1] = t.y; v[it][2] = t.z; v[it][3] = t.w;
    }
  } else {
#pragma unroll
    for (int it = 0; it < ITS; ++it) {
      const int64_t e = base + 4 * (int64_t)(it * GROUP + tid);
#pragma unroll
      for (int k = 0; k < 4; ++k) v[it][k] = e + k < total ? a.actions_in[e + k] : 0.f;
    }
  }
  {  // joint_gears -> LDS: compile-time indices keep the argument reads scalar (a run-time index into a kernel-argument
     // array is a VECTOR load from the argument block)
    float gj = 0.f;
#pragma unroll
    for (int j = 0; j < D; ++j) gj = tid == j ? a.p.joint_gears[j] : gj;
    if (tid < (D + 3) / 4 * 4) s_gear[tid] = gj;
  }

  // ---- compaction rank of the flagged envs (:121-123)
  const bool flagged = env < a.n && flag_raw != 0;
  int k0 = 0, cnt = 0;
  int64_t k = -1;
  if (PART == 1) {
    __syncthreads();
  } else {
    k = rank_finish<RANK>(pre, flagged, a.b.reset_workspace, a.b.tile_counts_u8, G, g, lds8, k0, cnt);
  }
  if (flagged) s_list[(int)(k - k0)] = tid;
  if (PART != 1 && g == G - 1 && tid == 0) *a.r.count = (int64_t)k0 + cnt;
  __syncthreads();  // s_list (and s_gear, for a workgroup that took the scanned rank)

  // ---- trip 2, issued IN FRONT of the stores below (loads and stores share one in-order counter: behind them, the wait for
  // these rows would also be the wait for the stores' acknowledgement).  Wave w takes the group's dense rows [64 w, 64 w + 64):
  // thread j of it the (64 w + j)-th flagged env, so a wave with a few flagged lanes does not run the Philox draws for all 64.
  // Nothing conditional: idle lanes re-read the group's first env (one request for all of them).
  const int w = tid >> 6, lane = lane_id();
  const int row0 = WAVE * w;
  const int nw = cnt - row0 < 0 ? 0 : (cnt - row0 > WAVE ? WAVE : cnt - row0);  // dense rows of this wave
  const bool mine = lane < nw;
  const int64_t di = g * GROUP + (mine ? s_list[row0 + lane] : 0);
  const int64_t kk = (int64_t)k0 + row0 + lane;
  ResetRow<D> row;
  if (PART != 1) row = reset_load<D>(a.r, di);
  float u[2 * D];
  if (PART != 1 && a.r.u_pos != nullptr) {  // the reference-flow tables: (count, D), so idle lanes may not touch them
#pragma unroll
    for (int j = 0; j < D; ++j) {
      u[j] = mine ? a.r.u_pos[kk * D + j] : 0.f;
      u[D + j] = mine ? a.r.u_vel[kk * D + j] : 0.f;
    }
  }

  // ---- forces = actions * joint_gears * power_scale (:125-126), lane-linear over the group's run; optional clamp
  // (vec_env_rlgames.py:77) and clone (:125)
#pragma unroll
  for (int it = 0; it < (PART == 2 ? 0 : ITS); ++it) {
    const int f = it * GROUP + tid;
    const int64_t e = base + 4 * (int64_t)f;
    if (NF4 % GROUP != 0 && f >= NF4) break;
    float fo[4];
    int col = (4 * f) % D;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (a.flags & OIGE_PRE_CLAMP_ACTIONS) v[it][q] = clamp_sym(v[it][q], a.p.clip_actions);
      fo[q] = (v[it][q] * s_gear[col]) * a.p.power_scale;
      if (++col == D) col = 0;
    }
    if (full) {
      if (a.actions_out) pre_st4_clone(a.actions_out + e, make_float4(v[it][0], v[it][1], v[it][2], v[it][3]));
      pre_st4_forces(a.forces_out + e, make_float4(fo[0], fo[1], fo[2], fo[3]));
    } else {
#pragma unroll
      for (int q = 0; q < 4; ++q)
        if (e + q < total) {
          if (a.actions_out) a.actions_out[e + q] = v[it][q];
          a.forces_out[e + q] = fo[q];
        }
    }
  }
  if (flagged) {  // natural lanes: consecutive lanes, consecutive flags
    a.b.reset_buf[env] = 0;     // :157
    a.b.progress_buf[env] = 0;  // :158
  }
  // A wave without a dense row ends here: its slots go to the next workgroup while the dense wave(s) finish (no barrier below).
  if (nw == 0) return;
#ifdef OIGE_LOCO_PRE_NO_RESET  // timing-only build
  return;
#endif

  // ---- reset_idx of this wave's dense rows (:133-158)
  if (a.r.u_pos == nullptr) reset_draws<D>(a.r, di, kk, u);
  float qp[D], qv[D];
  reset_dof_rows<D>(a.p, row.dof0, u, qp, qv);
  const float pot = reset_potential(a.p, row.ip);
  if (mine) {
    a.b.prev_potentials[di] = pot;
    a.b.potentials[di] = pot;  // :154
    a.r.env_ids[kk] = di;
    *reinterpret_cast<float4*>(a.r.root_rot_out + kk * 4) = row.iq;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.root_pos_out[kk * 3 + c] = row.ip[c];
    if (!STAGE) {
#pragma unroll
      for (int c = 0; c < D / 4; ++c) {
        reinterpret_cast<float4*>(a.r.dof_pos_out + kk * D)[c] = make_float4(qp[4 * c], qp[4 * c + 1], qp[4 * c + 2], qp[4 * c + 3]);
        reinterpret_cast<float4*>(a.r.dof_vel_out + kk * D)[c] = make_float4(qv[4 * c], qv[4 * c + 1], qv[4 * c + 2], qv[4 * c + 3]);
      }
    }
  }
  {  // root_vel rows are zeros (:142): the wave's rows are one contiguous run of nw * 6 floats, 8-byte aligned
    float2* z = reinterpret_cast<float2*>(a.r.root_vel_out + ((int64_t)k0 + row0) * 6);
    for (int e = lane; e < nw * 3; e += WAVE) z[e] = make_float2(0.f, 0.f);
  }
  if (STAGE) {
    // Rows that are not whole float4s (Humanoid: 21 floats) would leave each lane walking its own 84-byte row with scalar
    // stores; the wave's compacted rows are ONE contiguous run of nw * D floats, so they are staged in the wave's own part of
    // s_rows and written lane-linearly, dof_pos then dof_vel.  Wave-local: no workgroup barrier.
    float* mine_rows = s_rows + row0 * D;
    const int run = nw * D;
#pragma unroll
    for (int j = 0; j < D; ++j) mine_rows[lane * D + j] = qp[j];
    wave_lds_fence();
    float* dst = a.r.dof_pos_out + ((int64_t)k0 + row0) * D;
    for (int e = lane; e < run; e += WAVE) dst[e] = mine_rows[e];
    wave_lds_fence();
#pragma unroll
    for (int j = 0; j < D; ++j) mine_rows[lane * D + j] = qv[j];
    wave_lds_fence();
    dst = a.r.dof_vel_out + ((int64_t)k0 + row0) * D;
    for (int e = lane; e < run; e += WAVE) dst[e] = mine_rows[e];
  }
}

template <int D>
__global__ __launch_bounds__(256) void locomotion_reset_idx_kernel(const LocoPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  reset_env<D>(a.p, a.b, a.r, reset_load<D>(a.r, i), i, k);
}

// ---------------------------------------------------------------- host dispatch
static int check_params(const oige_locomotion_params* p) {
  OIGE_REQUIRE(p != nullptr, "locomotion: null params");
  const bool ant = p->num_dof == 8 && p->num_sensors == 4;
  const bool humanoid = p->num_dof == 21 && p->num_sensors == 2;
  OIGE_REQUIRE(ant || humanoid, "locomotion: unsupported (num_dof=%d, num_sensors=%d); built for Ant (8,4) and Humanoid (21,2)",
               p->num_dof, p->num_sensors);
  OIGE_REQUIRE(p->limit_cost_kind == 0 || p->limit_cost_kind == 1, "locomotion: limit_cost_kind must be 0 or 1");
  return OIGE_OK;
}

template <int D, int S>
static int launch_post(const LocoPost& a, hipStream_t s) {
  const int64_t T = num_tiles(a.n);
  if (T == 0) return OIGE_OK;
  if (a.stages & OIGE_STAGE_OBS) {
    if (a.p.limit_cost_kind == 0)
      hipLaunchKernelGGL((locomotion_post_kernel<D, S, 0>), dim3((unsigned)T), dim3(post_threads<D, S>()), 0, s, a);
    else
      hipLaunchKernelGGL((locomotion_post_kernel<D, S, 1>), dim3((unsigned)T), dim3(post_threads<D, S>()), 0, s, a);
    return check_launch("locomotion_post_kernel");
  }
  const unsigned blocks = (unsigned)((a.n + 255) / 256);
  if (a.stages & OIGE_STAGE_PROGRESS) {
    hipLaunchKernelGGL(progress_increment_kernel, dim3(blocks), dim3(256), 0, s, a.b.progress_buf, a.n);
    if (int rc = check_launch("progress_increment_kernel")) return rc;
  }
  if (a.stages & OIGE_STAGE_METRICS) {
    if (a.p.limit_cost_kind == 0)
      hipLaunchKernelGGL((locomotion_metrics_kernel<D, S, 0>), dim3(blocks), dim3(256), 0, s, a);
    else
      hipLaunchKernelGGL((locomotion_metrics_kernel<D, S, 1>), dim3(blocks), dim3(256), 0, s, a);
    if (int rc = check_launch("locomotion_metrics_kernel")) return rc;
  }
  if (a.stages & OIGE_STAGE_DONE) {
    hipLaunchKernelGGL(locomotion_done_kernel, dim3(blocks), dim3(256), 0, s, a, 12 + 3 * D + 6 * S);
    if (int rc = check_launch("locomotion_done_kernel")) return rc;
  }
  return OIGE_OK;
}

template <int D>
static int launch_pre(const LocoPre& a, hipStream_t s) {
  const int64_t G = num_groups(a.n);
  if (G == 0) return OIGE_OK;
  if (!(a.flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(a.b.reset_buf, a.n, a.b.reset_workspace, s, a.b.tile_counts_u8)) return rc;
  const int rank = pick_rank(G, a.b.tile_counts_u8, LOCO_UNSCANNED_MAX_GROUPS);
  if (rank == RANK_U8) {
#ifdef OIGE_LOCO_PRE_SPLIT
    hipLaunchKernelGGL((locomotion_pre_kernel<D, RANK_U8, 1>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
    hipLaunchKernelGGL((locomotion_pre_kernel<D, RANK_U8, 2>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
#else
    hipLaunchKernelGGL((locomotion_pre_kernel<D, RANK_U8>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
#endif
  } else if (rank == RANK_INT32) {
    hipLaunchKernelGGL((locomotion_pre_kernel<D, RANK_INT32>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (int rc = launch_scan_groups(a.n, a.b.reset_workspace, s)) return rc;
    hipLaunchKernelGGL((locomotion_pre_kernel<D, RANK_SCANNED>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("locomotion_pre_kernel");
}

static int check_reset_io(const oige_locomotion_reset_io* r) {
  OIGE_REQUIRE(r != nullptr, "locomotion: null reset_io");
  OIGE_REQUIRE(r->initial_root_pos && r->initial_root_rot && r->initial_dof_pos, "locomotion: null initial_* pointer");
  OIGE_REQUIRE((r->u_pos == nullptr) == (r->u_vel == nullptr), "locomotion: pass both u_pos and u_vel or neither");
  OIGE_REQUIRE(r->env_ids && r->count && r->dof_pos_out && r->dof_vel_out && r->root_pos_out && r->root_rot_out &&
                   r->root_vel_out,
               "locomotion: null reset output pointer");
  OIGE_REQUIRE(aligned16(r->initial_root_rot) && aligned16(r->root_rot_out), "locomotion: (N,4) buffers must be 16-byte aligned");
  OIGE_REQUIRE((reinterpret_cast<uintptr_t>(r->root_vel_out) & 7u) == 0, "locomotion: root_vel_out must be 8-byte aligned");
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_locomotion_post_physics(const oige_locomotion_params* params, const oige_locomotion_state* state,
                                 const oige_locomotion_buffers* buffers, int64_t num_envs, uint32_t stages,
                                 oige_stream_t stream) {
  if (int rc = check_params(params)) return rc;
  OIGE_REQUIRE(buffers != nullptr, "locomotion_post_physics: null buffers");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 128, "locomotion_post_physics: num_envs %lld out of range",
               (long long)num_envs);
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "locomotion_post_physics: bad stages mask 0x%x", stages);
  const oige_locomotion_buffers& b = *buffers;
  OIGE_REQUIRE(b.obs_buf && b.progress_buf, "locomotion_post_physics: null obs_buf/progress_buf");
  if (stages & (OIGE_STAGE_OBS | OIGE_STAGE_METRICS))
    OIGE_REQUIRE(b.potentials && b.prev_potentials && b.actions, "locomotion_post_physics: null potentials/actions");
  if (stages & OIGE_STAGE_METRICS) OIGE_REQUIRE(b.rew_buf, "locomotion_post_physics: null rew_buf");
  if (stages & OIGE_STAGE_DONE) OIGE_REQUIRE(b.reset_buf, "locomotion_post_physics: null reset_buf");
  OIGE_REQUIRE(aligned16(b.stats_slots), "locomotion_post_physics: stats_slots must be 16-byte aligned");
  LocoPost a;
  a.p = *params;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  a.max_len_minus_1 = (float)((double)params->max_episode_length - 1.0);
  if (stages & OIGE_STAGE_OBS) {
    OIGE_REQUIRE(state != nullptr, "locomotion_post_physics: null state");
    a.s = *state;
    OIGE_REQUIRE(a.s.torso_position && a.s.torso_rotation && a.s.velocities && a.s.dof_pos && a.s.dof_vel &&
                     a.s.sensor_force_torques,
                 "locomotion_post_physics: null state pointer");
    OIGE_REQUIRE(aligned16(a.s.torso_rotation) && aligned16(a.s.dof_pos) && aligned16(a.s.dof_vel) &&
                     aligned16(a.s.sensor_force_torques) && aligned16(b.actions) && aligned16(b.obs_buf) &&
                     (reinterpret_cast<uintptr_t>(a.s.velocities) & 7u) == 0,
                 "locomotion_post_physics: state/obs/actions buffers must be 16-byte aligned (velocities 8)");
  } else {
    a.s = oige_locomotion_state{};
  }
  hipStream_t s = (hipStream_t)stream;
  return params->num_dof == 8 ? launch_post<8, 4>(a, s) : launch_post<21, 2>(a, s);
}

int oige_locomotion_dof_at_limit_cost(const oige_locomotion_params* params, const float* obs_buf, void* cost_out,
                                      int64_t num_envs, oige_stream_t stream) {
  if (int rc = check_params(params)) return rc;
  OIGE_REQUIRE(obs_buf && cost_out, "locomotion_dof_at_limit_cost: null pointer");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 128, "locomotion_dof_at_limit_cost: num_envs %lld out of range",
               (long long)num_envs);
  OIGE_REQUIRE((reinterpret_cast<uintptr_t>(cost_out) & 7u) == 0, "locomotion_dof_at_limit_cost: cost_out must be 8-byte aligned");
  if (num_envs == 0) return OIGE_OK;
  const unsigned blocks = (unsigned)((num_envs + 255) / 256);
  hipStream_t s = (hipStream_t)stream;
  if (params->num_dof == 8) {
    if (params->limit_cost_kind == 0)
      hipLaunchKernelGGL((locomotion_limit_cost_kernel<8, 4, 0>), dim3(blocks), dim3(256), 0, s, *params, obs_buf, cost_out, num_envs);
    else
      hipLaunchKernelGGL((locomotion_limit_cost_kernel<8, 4, 1>), dim3(blocks), dim3(256), 0, s, *params, obs_buf, cost_out, num_envs);
  } else {
    if (params->limit_cost_kind == 0)
      hipLaunchKernelGGL((locomotion_limit_cost_kernel<21, 2, 0>), dim3(blocks), dim3(256), 0, s, *params, obs_buf, cost_out, num_envs);
    else
      hipLaunchKernelGGL((locomotion_limit_cost_kernel<21, 2, 1>), dim3(blocks), dim3(256), 0, s, *params, obs_buf, cost_out, num_envs);
  }
  return check_launch("locomotion_limit_cost_kernel");
}

int oige_locomotion_pre_physics(const oige_locomotion_params* params, const float* actions_in,
                                const oige_locomotion_buffers* buffers, const oige_locomotion_reset_io* reset_io,
                                float* actions_out, float* forces_out, int64_t num_envs, uint32_t flags,
                                oige_stream_t stream) {
  if (int rc = check_params(params)) return rc;
  if (int rc = check_reset_io(reset_io)) return rc;
  OIGE_REQUIRE(buffers && actions_in && forces_out, "locomotion_pre_physics: null pointer");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 128, "locomotion_pre_physics: num_envs %lld out of range",
               (long long)num_envs);
  const oige_locomotion_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.potentials && b.prev_potentials && b.reset_workspace,
               "locomotion_pre_physics: null reset/progress/potentials/workspace");
  OIGE_REQUIRE(!b.tile_counts_u8 || aligned16(b.tile_counts_u8), "locomotion_pre_physics: tile_counts_u8 must be 16-byte aligned");
  OIGE_REQUIRE(aligned16(actions_in) && aligned16(forces_out) && aligned16(actions_out) && aligned16(b.reset_workspace),
               "locomotion_pre_physics: actions/forces/workspace must be 16-byte aligned");
  OIGE_REQUIRE(params->num_dof % 4 != 0 || (aligned16(reset_io->dof_pos_out) && aligned16(reset_io->dof_vel_out)),
               "locomotion_pre_physics: dof_pos_out/dof_vel_out must be 16-byte aligned");
  LocoPre a;
  a.p = *params;
  a.b = b;
  a.r = *reset_io;
  a.actions_in = actions_in;
  a.actions_out = actions_out;
  a.forces_out = forces_out;
  a.n = num_envs;
  a.flags = flags;
  hipStream_t s = (hipStream_t)stream;
  return params->num_dof == 8 ? launch_pre<8>(a, s) : launch_pre<21>(a, s);
}

int oige_locomotion_reset_idx(const oige_locomotion_params* params, const oige_locomotion_buffers* buffers,
                              const oige_locomotion_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                              oige_stream_t stream) {
  if (int rc = check_params(params)) return rc;
  if (int rc = check_reset_io(reset_io)) return rc;
  OIGE_REQUIRE(buffers != nullptr, "locomotion_reset_idx: null buffers");
  OIGE_REQUIRE(num_envs >= 0 && max_ids >= 0 && max_ids <= num_envs, "locomotion_reset_idx: bad num_envs/max_ids");
  const oige_locomotion_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.potentials && b.prev_potentials,
               "locomotion_reset_idx: null reset/progress/potentials");
  if (max_ids == 0) return OIGE_OK;
  LocoPre a;
  a.p = *params;
  a.b = b;
  a.r = *reset_io;
  a.actions_in = nullptr;
  a.actions_out = nullptr;
  a.forces_out = nullptr;
  a.n = num_envs;
  a.flags = 0;
  hipStream_t s = (hipStream_t)stream;
  const unsigned blocks = (unsigned)((max_ids + 255) / 256);
  if (params->num_dof == 8)
    hipLaunchKernelGGL((locomotion_reset_idx_kernel<8>), dim3(blocks), dim3(256), 0, s, a, max_ids);
  else
    hipLaunchKernelGGL((locomotion_reset_idx_kernel<21>), dim3(blocks), dim3(256), 0, s, a, max_ids);
  return check_launch("locomotion_reset_idx_kernel");
}

}  // extern "C"
