// Crazyflie per-step task math for gfx950 (MI355X) -- SURVEY.md 8f rank 3.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/crazyflie.py behind the unchanged CrazyflieTask method contract:
//   get_observations :140-164, calculate_metrics :337-378, is_done :380-395, pre_physics_step :166-258,
//   set_targets :300-311, reset_idx :313-335.
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, 256 algorithmic bytes per env-step (R root pos 12 + rot 16 +
//   vel 24 + target 12 + env_pos 12 + actions 16 + progress 8 + 8 episode sums 32; W obs 72 + rew 4 + reset 8 + progress 8 +
//   episode sums 32); the (64 x 18) observation tile goes through LDS and streams out as one run.
// pre_physics: two ordered compactions (reset ids; set_targets ids = progress % 500 == 0 after the resets), the first-order
//   motor filter, the step's shared 4-vector of thrust noise (drawn by the host with torch.randn, as the reference does),
//   thrust along the body z axis, rotor speeds; the means of the episode sums over the reset envs (extras) are reduced
//   per 256-env group and folded by a single-workgroup kernel in group order.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

// lane-linear / per-env whole-line float4 streams outside the tile helpers: non-temporal like those (oige_tile.hpp);
// measured per task at 1 M envs, kept where the whole step gained (Crazyflie step 0.183 -> 0.178 ms; the same change lost 2-3 % on
// Quadcopter, Ingenuity and FrankaCabinet, whose pre kernels keep plain accesses)
#define LD4S(p) ld4_stream(p)
#define ST4S(p, v) st4_stream((p), (v))

namespace oige {

constexpr int CO = OIGE_CRAZYFLIE_OBS;   // 18
constexpr int CROW = CO;                 // LDS rows are compact, so the tile IS the output run (stride 18: 2-way bank conflicts)
constexpr int CE = OIGE_CRAZYFLIE_EPISODE_SUMS;  // 8

struct CfPost {
  oige_crazyflie_params p;
  oige_crazyflie_state s;
  oige_crazyflie_buffers b;
  int64_t n;
  uint32_t stages;
};

template <bool FULL>
__device__ __forceinline__ void cf_post_tile(const CfPost& a, float* __restrict__ tile, int64_t t, int nv) {
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;
  f3 pos{0.f, 0.f, 1.f}, epos{0.f, 0.f, 0.f}, tgt{0.f, 0.f, 1.f}, lin{0.f, 0.f, 0.f}, ang{0.f, 0.f, 0.f};
  q4 rot{1.f, 0.f, 0.f, 0.f};
  float4 act = make_float4(0.f, 0.f, 0.f, 0.f);
  int64_t progress = 0;
  if (live) {
    const float* p3 = a.s.root_pos + env * 3;
    pos = f3{p3[0], p3[1], p3[2]};
    const float* e3 = a.s.env_pos + env * 3;
    epos = f3{e3[0], e3[1], e3[2]};
    const float* t3 = a.b.target_positions + env * 3;
    tgt = f3{t3[0], t3[1], t3[2]};
    const float4 r4 = *reinterpret_cast<const float4*>(a.s.root_rot + env * 4);
    rot = q4{r4.x, r4.y, r4.z, r4.w};
    const float2* vp = reinterpret_cast<const float2*>(a.s.root_velocities + env * 6);
    const float2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
    lin = f3{v0.x, v0.y, v1.x};
    ang = f3{v1.y, v2.x, v2.y};
    if (do_metrics) act = *reinterpret_cast<const float4*>(a.b.actions + env * 4);
    if (do_progress || do_done) progress = a.b.progress_buf[env];
  }
  // the env's eight episode sums, fetched with everything else (r03: written as `sums[k] += add[k]` eight times over they were
  // eight serial load -> store -> load round trips, each load waiting for the previous store's acknowledgement; 32 resident
  // waves per CU hid them all the same: 44.3 us either way)
  float es[CE];
#pragma unroll
  for (int k = 0; k < CE; ++k) es[k] = 0.f;
  if (live && do_metrics) {
#pragma unroll
    for (int k = 0; k < CE; ++k) es[k] = a.b.episode_sums[(int64_t)k * a.n + env];
  }
  if (do_progress) progress += 1;  // rl_task.py:289
  const f3 rp{pos.x - epos.x, pos.y - epos.y, pos.z - epos.z};
  const f3 d{tgt.x - rp.x, tgt.y - rp.y, tgt.z - rp.z};
  const float target_dist = __builtin_sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);
  const f3 rz = quat_rotate(rot, f3{0.0f, 0.0f, 1.0f});  // quat_axis(q, 2)
  if (do_obs) {
    const f3 rx = quat_rotate(rot, f3{1.0f, 0.0f, 0.0f}), ry = quat_rotate(rot, f3{0.0f, 1.0f, 0.0f});
    float* row = tile + lane * CROW;
    row[0] = d.x; row[1] = d.y; row[2] = d.z;
    row[3] = rx.x; row[4] = rx.y; row[5] = rx.z;
    row[6] = ry.x; row[7] = ry.y; row[8] = ry.z;
    row[9] = rz.x; row[10] = rz.y; row[11] = rz.z;
    row[12] = lin.x; row[13] = lin.y; row[14] = lin.z;
    row[15] = ang.x; row[16] = ang.y; row[17] = ang.z;
  }
  if (do_metrics) {  // :337-378
    const float pos_reward = 1.0f / (1.0f + target_dist);
    const float up_reward = (rz.z != rz.z) ? rz.z : (rz.z < 0.0f ? 0.0f : (rz.z > 1.0f ? 1.0f : rz.z));
    const float effort = ((act.x * act.x + act.y * act.y) + act.z * act.z) + act.w * act.w;
    const float effort_reward = 0.05f * expf(-0.5f * effort);
    const float spin = (ang.x * ang.x + ang.y * ang.y) + ang.z * ang.z;
    const float spin_reward = 0.01f * expf(-1.0f * spin);
    if (live) {
      a.b.rew_buf[env] = (pos_reward + pos_reward * (up_reward + spin_reward)) - effort_reward;
      const float add[CE] = {pos_reward, up_reward, effort_reward, spin_reward, target_dist, rz.z, effort, spin};
#pragma unroll
      for (int k = 0; k < CE; ++k) a.b.episode_sums[(int64_t)k * a.n + env] = es[k] + add[k];
    }
  }
  if (do_done) {  // :380-395, not sticky
    const bool die = (target_dist > 5.0f) | (rp.z < 0.5f) | (rp.z > 5.0f) | (rz.z < 0.0f);
    const bool r = (progress >= (int64_t)a.p.max_episode_length - 1) | die;
    if (live) a.b.reset_buf[env] = r ? 1 : 0;
    // the tile's counts of the two lists the next pre_physics_step compacts (reset_buf; reset_buf | progress % 500 == 0,
    // :183-187), so that it can skip its counting launch (OIGE_PRE_COUNTS_VALID)
    if (a.b.reset_workspace && a.b.target_workspace) {
      const unsigned long long mr = __ballot(live && r), mt = __ballot(live && (r || progress % 500 == 0));
      if (lane == 0) {
        a.b.reset_workspace[t] = __popcll(mr);
        a.b.target_workspace[t] = __popcll(mt);
      }
    }
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) store_tile<CO, FULL>(tile, a.b.obs_buf + env0 * CO, nv * CO);  // rows are compact: one contiguous run
}

__global__ __launch_bounds__(WAVE) void crazyflie_post_kernel(const CfPost a) {
  __shared__ __align__(16) float tile[TILE * CROW];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    cf_post_tile<true>(a, tile, t, TILE);
  else
    cf_post_tile<false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct CfPre {
  oige_crazyflie_params p;
  oige_crazyflie_buffers b;
  oige_crazyflie_reset_io r;
  const float* actions_in;
  const float* root_rot;
  int64_t n;
  uint32_t flags;
};

__global__ __launch_bounds__(GROUP) void crazyflie_count_kernel(const CfPre a) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool live = env < a.n;
  const int64_t le = live ? env : 0;
  const int64_t fr_ld = a.b.reset_buf[le], pg_ld = a.b.progress_buf[le];  // side by side, not conditional (crazyflie_pre_kernel)
  const bool fr = live & (fr_ld != 0);
  const bool ft = live & ((fr_ld != 0) | (pg_ld % 500 == 0));
  const unsigned long long mr = __ballot(fr), mt = __ballot(ft);
  if (lane_id() == 0) {
    const int64_t t = env / TILE;
    a.b.reset_workspace[t] = __popcll(mr);
    a.b.target_workspace[t] = __popcll(mt);
  }
}

// reset_idx (:313-335) rows and flags for env i -> compacted row k (the episode sums are handled by the caller), from
// the rows the caller fetched (ip = initial_root_pos[i], iq = initial_root_rot[i]): the kernels keep every load in front of
// their first store (a load issued behind a store waits for that store's round trip)
__device__ __forceinline__ void cf_reset_rows(const CfPre& a, int64_t i, int64_t k, const float (&ip)[3], const float4 iq) {
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(a.b.dof_pos + i * 4) = z4;  // torch_rand_float(-0.0, 0.0, ...) == 0 (:316)
  *reinterpret_cast<float4*>(a.r.dof_pos_out + k * 4) = z4;
  *reinterpret_cast<float4*>(a.r.dof_vel_out + k * 4) = z4;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.r.root_pos_out[k * 3 + c] = ip[c] + 0.0f;
  *reinterpret_cast<float4*>(a.r.root_rot_out + k * 4) = iq;
#pragma unroll
  for (int c = 0; c < 6; ++c) a.r.root_vel_out[k * 6 + c] = 0.0f;
  a.b.reset_buf[i] = 0;
  a.b.progress_buf[i] = 0;
  a.r.env_ids[k] = i;
}

__device__ __forceinline__ void cf_set_target(const CfPre& a, int64_t i, int64_t k, const float (&ep)[3], const float4 bq) {  // :300-311
  float* tp = a.b.target_positions + i * 3;
  tp[0] = 0.0f; tp[1] = 0.0f; tp[2] = 2.0f;
  a.r.ball_pos_out[k * 3 + 0] = 0.0f + ep[0];
  a.r.ball_pos_out[k * 3 + 1] = 0.0f + ep[1];
  a.r.ball_pos_out[k * 3 + 2] = (2.0f + ep[2]) + 0.0f;
  *reinterpret_cast<float4*>(a.r.ball_rot_out + k * 4) = bq;
  a.r.target_ids[k] = i;
}

__device__ __forceinline__ float wave_sum_cf(float v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

__global__ __launch_bounds__(GROUP) void crazyflie_pre_kernel(const CfPre a) {
  __shared__ int lds_a[GROUP_TILES], lds_b[GROUP_TILES];
  __shared__ float s_part[GROUP_TILES][CE];
  const oige_crazyflie_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const int32_t* off_r = a.b.reset_workspace + G * GROUP_TILES;
  const int32_t* off_t = a.b.target_workspace + G * GROUP_TILES;
  const bool live = env < a.n;
  // ---- every load of the thread first.  The two flags through a clamped index, side by side (r03: written as `live &&
  // flag[env]` each was a branch around a load with a full wait behind it -- two serial round trips in front of the rest)
  const int64_t le = live ? env : 0;  // dead lanes shadow env 0, never store
  const int64_t fr_ld = a.b.reset_buf[le];
  const int64_t pg_ld = a.b.progress_buf[le];
  const float4 av4 = *reinterpret_cast<const float4*>(a.actions_in + le * 4);
  // (reset envs read their filter states too and ignore them: reset_idx zeroes both, :328-329)
  const float4 cd_in = *reinterpret_cast<const float4*>(a.b.thrust_cmds_damp + le * 4);
  const float4 rd_in = LD4S(a.b.thrust_rot_damp + le * 4);
  const float4 r4 = *reinterpret_cast<const float4*>(a.root_rot + le * 4);
  const float tn[4] = {a.r.thrust_noise4[0], a.r.thrust_noise4[1], a.r.thrust_noise4[2], a.r.thrust_noise4[3]};  // (a device tensor)
  const int32_t total_r = off_r[G], total_t = off_t[G];
  const bool fr = live && fr_ld != 0;
  const bool ft = live & ((fr_ld != 0) | (pg_ld % 500 == 0));  // (bitwise: behind `||` the compiler sinks the second load into a branch)
  float ip[3] = {0.f, 0.f, 0.f}, ep[3] = {0.f, 0.f, 0.f};
  float4 iq = make_float4(0.f, 0.f, 0.f, 0.f), bq = iq;
  float es[CE];
#pragma unroll
  for (int k = 0; k < CE; ++k) es[k] = 0.0f;
  if (fr) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ip[c] = a.r.initial_root_pos[env * 3 + c];
    iq = *reinterpret_cast<const float4*>(a.r.initial_root_rot + env * 4);
#pragma unroll
    for (int k = 0; k < CE; ++k) es[k] = a.b.episode_sums[(int64_t)k * a.n + env];
  }
  if (ft) {
#pragma unroll
    for (int c = 0; c < 3; ++c) ep[c] = a.r.env_pos[env * 3 + c];
    bq = *reinterpret_cast<const float4*>(a.r.initial_ball_rot + env * 4);
  }
  const int64_t kr = group_rank(fr, off_r, g, lds_a);
  const int64_t kt = group_rank(ft, off_t, g, lds_b);
  // episode sums of the reset envs: per-group partial sums for extras (:331-335); consumed before the first store so
  // that the wait for these (conditional) loads is not also a wait for stores
#pragma unroll
  for (int k = 0; k < CE; ++k) {
    const float s = wave_sum_cf(es[k]);
    if (lane_id() == 0) s_part[tid >> 6][k] = s;
  }
  __syncthreads();
  // ---- stores from here on
  if (tid < CE) {
    float s = 0.0f;
#pragma unroll
    for (int w = 0; w < GROUP_TILES; ++w) s += s_part[w][tid];
    a.b.reduce_workspace[g * CE + tid] = s;
  }
  // these rows are a handful of scalars per env: the flagged lanes write them themselves (no dense-lane pass needed)
  if (fr) {
    cf_reset_rows(a, env, kr, ip, iq);
#pragma unroll
    for (int k = 0; k < CE; ++k) a.b.episode_sums[(int64_t)k * a.n + env] = 0.0f;  // then zero (:335)
  }
  if (ft) cf_set_target(a, env, kt, ep, bq);
  if (live) {  // :174-251
    float av[4] = {av4.x, av4.y, av4.z, av4.w};
    if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
#pragma unroll
      for (int c = 0; c < 4; ++c) av[c] = clamp_sym(av[c], p.clip_actions);
    }
    ST4S(a.b.actions + env * 4, make_float4(av[0], av[1], av[2], av[3]));  // self.actions = actions (:180)
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const float4 cd4 = fr ? z4 : cd_in, rd4 = fr ? z4 : rd_in;  // reset_idx zeroed both filter states of the reset envs
    const float cd[4] = {cd4.x, cd4.y, cd4.z, cd4.w}, rd[4] = {rd4.x, rd4.y, rd4.z, rd4.w};
    const q4 rot{r4.x, r4.y, r4.z, r4.w};
    const f3 rx = quat_rotate(rot, f3{1.0f, 0.0f, 0.0f}), ry = quat_rotate(rot, f3{0.0f, 1.0f, 0.0f});
    const f3 rz = quat_rotate(rot, f3{0.0f, 0.0f, 1.0f});
    float ncd[4], nrd[4], th[12];
    // tau[tau > 1] = 1 applied to the two constants before the per-motor choice: picked straight from the argument block,
    // the choice becomes a run-time-indexed VECTOR load of it, issued behind the stores above
    const float tau_down = p.motor_tau_down > 1.0f ? 1.0f : p.motor_tau_down;
    const float tau_up = p.motor_tau_up > 1.0f ? 1.0f : p.motor_tau_up;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cl = clamp_sym(av[j], 1.0f);                 // torch.clamp(actions, min=-1.0, max=1.0)
      const float cmd = (cl + 1.0f) / 2.0f;
      const float tau = cmd < cd[j] ? tau_down : tau_up;
      const float trot = __builtin_sqrtf(cmd);                 // thrust_cmds ** 0.5
      nrd[j] = tau * (trot - rd[j]) + rd[j];
      float c2 = nrd[j] * nrd[j];                              // thrust_rot_damp ** 2
      c2 = c2 + cmd * tn[j];
      ncd[j] = (c2 != c2) ? c2 : (c2 < 0.0f ? 0.0f : (c2 > 1.0f ? 1.0f : c2));
      const float thr = p.thrust_max[j] * ncd[j];
      // matmul(rot_matrix, (0, 0, thr)): rows of the matrix are the body axes, so the result is their z components * thr
      th[3 * j + 0] = fr ? 0.0f : rx.z * thr;
      th[3 * j + 1] = fr ? 0.0f : ry.z * thr;
      th[3 * j + 2] = fr ? 0.0f : rz.z * thr;
    }
    *reinterpret_cast<float4*>(a.b.thrust_cmds_damp + env * 4) = make_float4(ncd[0], ncd[1], ncd[2], ncd[3]);
    ST4S(a.b.thrust_rot_damp + env * 4, make_float4(nrd[0], nrd[1], nrd[2], nrd[3]));
    float4* to = reinterpret_cast<float4*>(a.b.thrusts + env * 12);
    to[0] = make_float4(th[0], th[1], th[2], th[3]);
    to[1] = make_float4(th[4], th[5], th[6], th[7]);
    to[2] = make_float4(th[8], th[9], th[10], th[11]);
    *reinterpret_cast<float4*>(a.b.dof_vel + env * 4) =
        make_float4(ncd[0] * p.prop_max_rot, -1.0f * (ncd[1] * p.prop_max_rot), ncd[2] * p.prop_max_rot, -1.0f * (ncd[3] * p.prop_max_rot));
  }
  if (g == 0 && tid == 0) {
    *a.r.count = (int64_t)total_r;
    *a.r.target_count = (int64_t)total_t;
  }
}

// extras["episode"][key] = mean(episode_sums[key][env_ids]) / max_episode_length (:331-334): groups folded in order
constexpr int CF_EX_PER = 1024;  // threads per episode-sum key: one workgroup per key
__global__ __launch_bounds__(CF_EX_PER) void crazyflie_extras_kernel(const CfPre a) {
  __shared__ double part[CF_EX_PER / WAVE];
  const int k = blockIdx.x, j = threadIdx.x, lane = lane_id();
  const int64_t G = num_groups(a.n);
  double acc = 0.0;
  // f64 partial sums; at 1 M envs every thread has four independent loads (with one wave per key this launch was 64
  // dependent load-add rounds: 16 us of pure latency)
#pragma unroll 4
  for (int64_t i = j; i < G; i += CF_EX_PER) acc += (double)a.b.reduce_workspace[i * CE + k];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane == 0) part[j / WAVE] = acc;
  __syncthreads();
  if (j == 0) {
    double tot = 0.0;
#pragma unroll
    for (int w = 0; w < CF_EX_PER / WAVE; ++w) tot += part[w];
    const int32_t count = a.b.reset_workspace[G * GROUP_TILES + G];
    if (count > 0) a.b.extras[k] = ((float)tot / (float)count) / (float)a.p.max_episode_length;
  }
}

__global__ __launch_bounds__(256) void crazyflie_reset_idx_kernel(const CfPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  const float ip[3] = {a.r.initial_root_pos[i * 3], a.r.initial_root_pos[i * 3 + 1], a.r.initial_root_pos[i * 3 + 2]};
  cf_reset_rows(a, i, k, ip, *reinterpret_cast<const float4*>(a.r.initial_root_rot + i * 4));
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(a.b.thrust_cmds_damp + i * 4) = z4;
  *reinterpret_cast<float4*>(a.b.thrust_rot_damp + i * 4) = z4;
  // the episode sums and extras of this flow stay with the caller (two torch lines, as in the reference)
}

__global__ __launch_bounds__(256) void crazyflie_set_targets_kernel(const CfPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.target_count) return;
  const int64_t i = a.r.target_ids[k];
  if (i < 0 || i >= a.n) return;
  const float ep[3] = {a.r.env_pos[i * 3], a.r.env_pos[i * 3 + 1], a.r.env_pos[i * 3 + 2]};
  cf_set_target(a, i, k, ep, *reinterpret_cast<const float4*>(a.r.initial_ball_rot + i * 4));
}

static int check_cf(const oige_crazyflie_params* p, const oige_crazyflie_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_crazyflie_post_physics(const oige_crazyflie_params* params, const oige_crazyflie_state* state,
                                const oige_crazyflie_buffers* buffers, int64_t num_envs, uint32_t stages, oige_stream_t stream) {
  if (int rc = check_cf(params, buffers, num_envs, "crazyflie_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "crazyflie_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "crazyflie_post_physics: bad stages mask 0x%x", stages);
  const oige_crazyflie_state& s = *state;
  const oige_crazyflie_buffers& b = *buffers;
  OIGE_REQUIRE(s.root_pos && s.root_rot && s.root_velocities && s.env_pos, "crazyflie_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.target_positions && b.actions && b.episode_sums,
               "crazyflie_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.root_rot) && aligned16(b.obs_buf) && aligned16(b.actions) &&
                   (reinterpret_cast<uintptr_t>(s.root_velocities) & 7u) == 0,
               "crazyflie_post_physics: root_rot / obs_buf / actions must be 16-byte aligned (velocities 8)");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  CfPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipLaunchKernelGGL(crazyflie_post_kernel, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("crazyflie_post_kernel");
}

static int fill_cf_pre(CfPre& a, const oige_crazyflie_params* params, const oige_crazyflie_buffers* buffers,
                       const oige_crazyflie_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_cf(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  a.p = *params;
  a.b = *buffers;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  a.root_rot = nullptr;
  return OIGE_OK;
}

static int check_cf_reset(const CfPre& a, const char* who) {
  const oige_crazyflie_reset_io& r = a.r;
  OIGE_REQUIRE(a.b.reset_buf && a.b.progress_buf && a.b.dof_pos && a.b.thrust_cmds_damp && a.b.thrust_rot_damp &&
                   aligned16(a.b.dof_pos) && aligned16(a.b.thrust_cmds_damp) && aligned16(a.b.thrust_rot_damp),
               "%s: null/unaligned reset_buf / progress_buf / dof_pos / filter states", who);
  OIGE_REQUIRE(r.initial_root_pos && r.initial_root_rot && r.env_ids && r.count && r.dof_pos_out && r.dof_vel_out && r.root_pos_out &&
                   r.root_rot_out && r.root_vel_out,
               "%s: null reset pointer", who);
  OIGE_REQUIRE(aligned16(r.initial_root_rot) && aligned16(r.root_rot_out) && aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out),
               "%s: row buffers must be 16-byte aligned", who);
  return OIGE_OK;
}

static int check_cf_targets(const CfPre& a, const char* who) {
  const oige_crazyflie_reset_io& r = a.r;
  OIGE_REQUIRE(a.b.target_positions && r.env_pos && r.initial_ball_rot && r.target_ids && r.target_count && r.ball_pos_out && r.ball_rot_out,
               "%s: null set_targets pointer", who);
  OIGE_REQUIRE(aligned16(r.initial_ball_rot) && aligned16(r.ball_rot_out), "%s: (n,4) buffers must be 16-byte aligned", who);
  return OIGE_OK;
}

int oige_crazyflie_pre_physics(const oige_crazyflie_params* params, const float* actions_in, const float* root_rot,
                               const oige_crazyflie_buffers* buffers, const oige_crazyflie_reset_io* reset_io, int64_t num_envs,
                               uint32_t flags, oige_stream_t stream) {
  CfPre a;
  if (int rc = fill_cf_pre(a, params, buffers, reset_io, num_envs, "crazyflie_pre_physics")) return rc;
  if (int rc = check_cf_reset(a, "crazyflie_pre_physics")) return rc;
  if (int rc = check_cf_targets(a, "crazyflie_pre_physics")) return rc;
  OIGE_REQUIRE(actions_in && root_rot && aligned16(actions_in) && aligned16(root_rot), "crazyflie_pre_physics: null/unaligned actions / root_rot");
  OIGE_REQUIRE(a.b.actions && a.b.thrusts && a.b.dof_vel && a.b.episode_sums && a.b.extras && a.b.reduce_workspace && a.r.thrust_noise4 &&
                   aligned16(a.b.actions) && aligned16(a.b.thrusts) && aligned16(a.b.dof_vel),
               "crazyflie_pre_physics: null/unaligned actions / thrusts / dof_vel / episode_sums / extras / reduce_workspace / thrust_noise4");
  OIGE_REQUIRE(a.b.reset_workspace && a.b.target_workspace && aligned16(a.b.reset_workspace) && aligned16(a.b.target_workspace),
               "crazyflie_pre_physics: null/unaligned workspaces");
  a.actions_in = actions_in;
  a.root_rot = root_rot;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID)) {  // the post kernel's DONE stage leaves the same tile counts
    hipLaunchKernelGGL(crazyflie_count_kernel, dim3((unsigned)G), dim3(GROUP), 0, s, a);
    if (int rc = check_launch("crazyflie_count_kernel")) return rc;
  }
  if (int rc = launch_scan_groups2(num_envs, a.b.reset_workspace, a.b.target_workspace, s)) return rc;
  hipLaunchKernelGGL(crazyflie_pre_kernel, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  if (int rc = check_launch("crazyflie_pre_kernel")) return rc;
  hipLaunchKernelGGL(crazyflie_extras_kernel, dim3(CE), dim3(CF_EX_PER), 0, s, a);
  return check_launch("crazyflie_extras_kernel");
}

int oige_crazyflie_reset_idx(const oige_crazyflie_params* params, const oige_crazyflie_buffers* buffers,
                             const oige_crazyflie_reset_io* reset_io, int64_t num_envs, int64_t max_ids, oige_stream_t stream) {
  CfPre a;
  if (int rc = fill_cf_pre(a, params, buffers, reset_io, num_envs, "crazyflie_reset_idx")) return rc;
  if (int rc = check_cf_reset(a, "crazyflie_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "crazyflie_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(crazyflie_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("crazyflie_reset_idx_kernel");
}

int oige_crazyflie_set_targets(const oige_crazyflie_params* params, const oige_crazyflie_buffers* buffers,
                               const oige_crazyflie_reset_io* reset_io, int64_t num_envs, int64_t max_ids, oige_stream_t stream) {
  CfPre a;
  if (int rc = fill_cf_pre(a, params, buffers, reset_io, num_envs, "crazyflie_set_targets")) return rc;
  if (int rc = check_cf_targets(a, "crazyflie_set_targets")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "crazyflie_set_targets: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(crazyflie_set_targets_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("crazyflie_set_targets_kernel");
}

}  // extern "C"
