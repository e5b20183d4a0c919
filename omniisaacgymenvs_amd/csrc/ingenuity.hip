// Ingenuity per-step task math for gfx950 (MI355X) -- SURVEY.md 8f rank 3.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/ingenuity.py behind the unchanged IngenuityTask method contract:
//   get_observations :128-147, calculate_metrics :244-268, is_done :270-281, pre_physics_step :149-190,
//   set_targets :213-223, reset_idx :225-249.
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, 168 algorithmic bytes per env-step (R root pos 12 + rot 16 +
//   vel 24 + target 12 + env_pos 12 + progress 8; W obs 52 + rew 4 + reset 8 + progress 8 ... ), obs tile through LDS.
// pre_physics: TWO ordered compactions per step -- the reset ids, and the set_targets ids (progress % 500 == 0 AFTER the
//   resets, i.e. including every env just reset) -- then one env per thread for the thrust vectors and rotor speeds.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

constexpr int IO_ = OIGE_INGENUITY_OBS;  // 13
constexpr int IROW = IO_;                // LDS rows are compact (odd stride: bank-conflict free), so the tile IS the output run

struct IngPost {
  oige_ingenuity_params p;
  oige_ingenuity_state s;
  oige_ingenuity_buffers b;
  int64_t n;
  uint32_t stages;
};

template <bool FULL>
__device__ __forceinline__ void ing_post_tile(const IngPost& a, float* __restrict__ tile, int64_t t, int nv) {
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;
  f3 pos{0.f, 0.f, 1.f}, epos{0.f, 0.f, 0.f}, tgt{0.f, 0.f, 1.f}, lin{0.f, 0.f, 0.f}, ang{0.f, 0.f, 0.f};
  q4 rot{1.f, 0.f, 0.f, 0.f};
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
    if (do_progress || do_done) progress = a.b.progress_buf[env];
  }
  if (do_progress) progress += 1;  // rl_task.py:289
  const f3 rp{pos.x - epos.x, pos.y - epos.y, pos.z - epos.z};
  const f3 d{tgt.x - rp.x, tgt.y - rp.y, tgt.z - rp.z};
  const float target_dist = __builtin_sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);  // :251
  if (do_obs) {
    float* row = tile + lane * IROW;
    row[0] = d.x / 3.0f; row[1] = d.y / 3.0f; row[2] = d.z / 3.0f;
    row[3] = rot.w; row[4] = rot.x; row[5] = rot.y; row[6] = rot.z;
    row[7] = lin.x / 2.0f; row[8] = lin.y / 2.0f; row[9] = lin.z / 2.0f;
    row[10] = ang.x / OIGE_PI_F; row[11] = ang.y / OIGE_PI_F; row[12] = ang.z / OIGE_PI_F;
  }
  if (do_metrics) {  // :244-268
    const float pos_reward = 1.0f / (1.0f + (2.5f * target_dist) * target_dist);
    const f3 ups = quat_rotate(rot, f3{0.0f, 0.0f, 1.0f});  // quat_axis(q, 2)
    const float tiltage = __builtin_fabsf(1.0f - ups.z);
    const float up_reward = 1.0f / (1.0f + (30.0f * tiltage) * tiltage);
    const float spinnage = __builtin_fabsf(ang.z);
    const float spinnage_reward = 1.0f / (1.0f + (10.0f * spinnage) * spinnage);
    if (live) a.b.rew_buf[env] = pos_reward + pos_reward * (up_reward + spinnage_reward);
  }
  if (do_done) {  // :270-281, not sticky
    const bool die = (target_dist > 20.0f) | (rp.z < 0.5f);
    const bool r = (progress >= (int64_t)a.p.max_episode_length - 1) | die;
    if (live) a.b.reset_buf[env] = r ? 1 : 0;
    const unsigned long long m = __ballot(live && r);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.reset_counts_u8 && lane == 0) a.b.reset_counts_u8[t] = (uint8_t)__popcll(m);
    // and of the second list the next pre_physics_step compacts (reset or progress % 500 == 0, :154-160), so that it can
    // skip its counting launch (OIGE_PRE_COUNTS_VALID)
    const unsigned long long mt = __ballot(live && (r || progress % 500 == 0));
    if (a.b.target_workspace && lane == 0) a.b.target_workspace[t] = __popcll(mt);
    if (a.b.target_counts_u8 && lane == 0) a.b.target_counts_u8[t] = (uint8_t)__popcll(mt);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) store_tile<IO_, FULL>(tile, a.b.obs_buf + env0 * IO_, nv * IO_);  // rows are compact: one contiguous run
}

__global__ __launch_bounds__(WAVE) void ingenuity_post_kernel(const IngPost a) {
  __shared__ __align__(16) float tile[TILE * IROW];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    ing_post_tile<true>(a, tile, t, TILE);
  else
    ing_post_tile<false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct IngPre {
  oige_ingenuity_params p;
  oige_ingenuity_buffers b;
  oige_ingenuity_reset_io r;
  const float* actions_in;
  int64_t n;
  uint32_t flags;
};

// tile counts of both flag sets: reset_buf != 0, and "gets a new target": reset or progress % 500 == 0 (:154-160)
__global__ __launch_bounds__(GROUP) void ingenuity_count_kernel(const IngPre a) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool live = env < a.n;
  const int64_t le = live ? env : 0;
  const int64_t fr_ld = a.b.reset_buf[le], pg_ld = a.b.progress_buf[le];  // side by side, not conditional (ingenuity_pre_kernel)
  const bool fr = live & (fr_ld != 0);
  const bool ft = live & ((fr_ld != 0) | (pg_ld % 500 == 0));
  const unsigned long long mr = __ballot(fr), mt = __ballot(ft);
  if (lane_id() == 0) {
    const int64_t t = env / TILE;
    a.b.reset_workspace[t] = __popcll(mr);
    a.b.target_workspace[t] = __popcll(mt);
    if (a.b.reset_counts_u8) a.b.reset_counts_u8[t] = (uint8_t)__popcll(mr);
    if (a.b.target_counts_u8) a.b.target_counts_u8[t] = (uint8_t)__popcll(mt);
  }
}

__device__ __forceinline__ u4 ing_block(const IngPre& a, int64_t i, uint32_t blk) {
  const u4 c{(uint32_t)i, blk, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
  return philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
}

// the rows reset_idx / set_targets read for env i, fetched by the caller in front of its first store
struct IngResetRows {
  float4 dp, iq;
  float ip[3];
};
struct IngTargetRows {
  float4 bq;
  float ep[3];
};
__device__ __forceinline__ IngResetRows ing_fetch_reset(const IngPre& a, int64_t i) {
  IngResetRows w;
  w.dp = *reinterpret_cast<const float4*>(a.b.dof_pos + i * 4);
  w.iq = *reinterpret_cast<const float4*>(a.r.initial_root_rot + i * 4);
#pragma unroll
  for (int c = 0; c < 3; ++c) w.ip[c] = a.r.initial_root_pos[i * 3 + c];
  return w;
}
__device__ __forceinline__ IngTargetRows ing_fetch_target(const IngPre& a, int64_t i) {
  IngTargetRows w;
  w.bq = *reinterpret_cast<const float4*>(a.r.initial_ball_rot + i * 4);
#pragma unroll
  for (int c = 0; c < 3; ++c) w.ep[c] = a.r.env_pos[i * 3 + c];
  return w;
}

// reset_idx (:225-249) for env i -> compacted row k; u = dof1, dof3, x, y, z draws
__device__ __forceinline__ void ing_reset_env(const IngPre& a, int64_t i, int64_t k, const float (&u)[5], const IngResetRows& w) {
  float4 dp = w.dp;
  dp.y = 0.4f * u[0] + (-0.2f);
  dp.w = 0.4f * u[1] + (-0.2f);
  *reinterpret_cast<float4*>(a.b.dof_pos + i * 4) = dp;
  *reinterpret_cast<float4*>(a.r.dof_pos_out + k * 4) = dp;
  *reinterpret_cast<float4*>(a.r.dof_vel_out + k * 4) = make_float4(0.f, 0.f, 0.f, 0.f);  // dof_vel[env_ids, :] = 0 (:230)
#pragma unroll
  for (int c = 0; c < 3; ++c) a.r.root_pos_out[k * 3 + c] = w.ip[c] + (1.0f * u[2 + c] + (-0.5f));
  *reinterpret_cast<float4*>(a.r.root_rot_out + k * 4) = w.iq;
#pragma unroll
  for (int c = 0; c < 6; ++c) a.r.root_vel_out[k * 6 + c] = 0.0f;
  a.b.reset_buf[i] = 0;
  a.b.progress_buf[i] = 0;
  a.r.env_ids[k] = i;
}

// set_targets (:213-223) for env i -> compacted row k; u = x, y, z draws
__device__ __forceinline__ void ing_set_target(const IngPre& a, int64_t i, int64_t k, const float (&u)[3], const IngTargetRows& w) {
  const float tx = u[0] * 2.0f - 1.0f, ty = u[1] * 2.0f - 1.0f, tz = u[2] + 1.0f;
  float* tp = a.b.target_positions + i * 3;
  tp[0] = tx; tp[1] = ty; tp[2] = tz;
  a.r.ball_pos_out[k * 3 + 0] = tx + w.ep[0];
  a.r.ball_pos_out[k * 3 + 1] = ty + w.ep[1];
  a.r.ball_pos_out[k * 3 + 2] = (tz + w.ep[2]) + 0.4f;
  *reinterpret_cast<float4*>(a.r.ball_rot_out + k * 4) = w.bq;
  a.r.target_ids[k] = i;
}

// RANK: where the two lists' group offsets come from (oige_common.hpp: the scan launch, the int32 tile counts, their byte twins)
// Two round trips of loads, neither conditional, both in front of the first store (r03; quadcopter.hip has the reasoning):
// trip 1 the two flags side by side (clamped index), both byte-table prefixes, the env's own actions and DOF velocities; trip 2,
// behind the ranks, the rows of the dense lanes' two jobs.  The jobs themselves (Philox, scattered stores) run last; a wave
// without a dense lane ends before.
template <int RANK>
__global__ __launch_bounds__(GROUP) void ingenuity_pre_kernel(const IngPre a) {
  __shared__ int lds_a[2 * GROUP_TILES], lds_b[2 * GROUP_TILES];
  __shared__ int s_list_r[GROUP], s_list_t[GROUP];
  const oige_ingenuity_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const bool live = env < a.n;
  const int64_t le = live ? env : g * GROUP;
  // ---- trip 1 (reset envs read their actions too and ignore them)
  const int64_t fr_ld = a.b.reset_buf[le], pg_ld = a.b.progress_buf[le];
  const U8Prefix rpr = rank_issue<RANK>(a.b.reset_counts_u8, g, tid), rpt = rank_issue<RANK>(a.b.target_counts_u8, g, tid);
  const float2* ar = reinterpret_cast<const float2*>(a.actions_in + le * 6);
  const float2 a0 = ar[0], a1 = ar[1], a2 = ar[2];
  const float4 dv_in = *reinterpret_cast<const float4*>(a.b.dof_vel + le * 4);
  const bool fr = live & (fr_ld != 0);
  const bool ft = live & ((fr_ld != 0) | (pg_ld % 500 == 0));  // (bitwise: behind `||` the compiler sinks the second load into a branch)
  int k0r, cntr, k0t, cntt;
  const int64_t kr = rank_finish<RANK>(rpr, fr, a.b.reset_workspace, a.b.reset_counts_u8, G, g, lds_a, k0r, cntr);
  const int64_t kt = rank_finish<RANK>(rpt, ft, a.b.target_workspace, a.b.target_counts_u8, G, g, lds_b, k0t, cntt);
  if (fr) s_list_r[(int)(kr - k0r)] = tid;
  if (ft) s_list_t[(int)(kt - k0t)] = tid;
  __syncthreads();
  // ---- trip 2: dense lanes (thread j takes the j-th env of each list; idle lanes re-read the group's first env)
  const bool dense_r = tid < cntr, dense_t = tid < cntt;
  const int64_t ir = g * GROUP + (dense_r ? s_list_r[tid] : 0), it = g * GROUP + (dense_t ? s_list_t[tid] : 0);
  const IngResetRows rr = ing_fetch_reset(a, ir);
  const IngTargetRows tr = ing_fetch_target(a, it);
  // ---- stores
  if (live) {  // :162-185
    float th[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    if (!fr) {
      float av[6];
      av[0] = a0.x; av[1] = a0.y; av[2] = a1.x; av[3] = a1.y; av[4] = a2.x; av[5] = a2.y;
      if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
#pragma unroll
        for (int c = 0; c < 6; ++c) av[c] = clamp_sym(av[c], p.clip_actions);
      }
#pragma unroll
      for (int r = 0; r < 2; ++r) {
        const float vert = clamp_sym(av[3 * r + 2] * p.thrust_limit, p.thrust_limit);
        const float lx = clamp_sym(av[3 * r + 0] * p.thrust_lateral_component, p.thrust_lateral_component);
        const float ly = clamp_sym(av[3 * r + 1] * p.thrust_lateral_component, p.thrust_lateral_component);
        const float tz = p.dt * vert;
        th[3 * r + 2] = tz;
        th[3 * r + 0] = tz * lx;
        th[3 * r + 1] = tz * ly;
      }
    }
    float2* to = reinterpret_cast<float2*>(a.b.thrusts + env * 6);
    to[0] = make_float2(th[0], th[1]); to[1] = make_float2(th[2], th[3]); to[2] = make_float2(th[4], th[5]);
    float4 dv = dv_in;
    if (fr) { dv.x = 0.f; dv.z = 0.f; }  // :230, then :178-179 for every env
    dv.y = 50.0f;
    dv.w = -50.0f;
    *reinterpret_cast<float4*>(a.b.dof_vel + env * 4) = dv;
  }
  if (g == G - 1 && tid == 0) {
    *a.r.count = (int64_t)k0r + cntr;
    *a.r.target_count = (int64_t)k0t + cntt;
  }
  if (!(dense_r | dense_t)) return;
  if (dense_r) {  // reset_idx
    const u4 x = ing_block(a, ir, 0u), y = ing_block(a, ir, 1u);
    const float u[5] = {u01(x.x), u01(x.y), u01(x.z), u01(x.w), u01(y.x)};
    ing_reset_env(a, ir, (int64_t)k0r + tid, u, rr);
  }
  if (dense_t) {  // set_targets
    const u4 x = ing_block(a, it, 2u);
    const float u[3] = {u01(x.x), u01(x.y), u01(x.z)};
    ing_set_target(a, it, (int64_t)k0t + tid, u, tr);
  }
}

// reset_idx(env_ids) / set_targets(env_ids) with caller-provided ids and (optionally) pre-drawn U[0,1)
__global__ __launch_bounds__(256) void ingenuity_reset_idx_kernel(const IngPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[5];
  if (a.r.u_reset) {
#pragma unroll
    for (int c = 0; c < 5; ++c) u[c] = a.r.u_reset[k * 5 + c];
  } else {
    const u4 x = ing_block(a, i, 0u), y = ing_block(a, i, 1u);
    u[0] = u01(x.x); u[1] = u01(x.y); u[2] = u01(x.z); u[3] = u01(x.w); u[4] = u01(y.x);
  }
  ing_reset_env(a, i, k, u, ing_fetch_reset(a, i));
}

__global__ __launch_bounds__(256) void ingenuity_set_targets_kernel(const IngPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.target_count) return;
  const int64_t i = a.r.target_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[3];
  if (a.r.u_target) {
#pragma unroll
    for (int c = 0; c < 3; ++c) u[c] = a.r.u_target[k * 3 + c];
  } else {
    const u4 x = ing_block(a, i, 2u);
    u[0] = u01(x.x); u[1] = u01(x.y); u[2] = u01(x.z);
  }
  ing_set_target(a, i, k, u, ing_fetch_target(a, i));
}

static int check_ing(const oige_ingenuity_params* p, const oige_ingenuity_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_ingenuity_post_physics(const oige_ingenuity_params* params, const oige_ingenuity_state* state,
                                const oige_ingenuity_buffers* buffers, int64_t num_envs, uint32_t stages, oige_stream_t stream) {
  if (int rc = check_ing(params, buffers, num_envs, "ingenuity_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "ingenuity_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "ingenuity_post_physics: bad stages mask 0x%x", stages);
  const oige_ingenuity_state& s = *state;
  const oige_ingenuity_buffers& b = *buffers;
  OIGE_REQUIRE(s.root_pos && s.root_rot && s.root_velocities && s.env_pos, "ingenuity_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.target_positions, "ingenuity_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.root_rot) && aligned16(b.obs_buf) && (reinterpret_cast<uintptr_t>(s.root_velocities) & 7u) == 0,
               "ingenuity_post_physics: root_rot / obs_buf must be 16-byte aligned (velocities 8)");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  IngPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipLaunchKernelGGL(ingenuity_post_kernel, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("ingenuity_post_kernel");
}

static int fill_ing_pre(IngPre& a, const oige_ingenuity_params* params, const oige_ingenuity_buffers* buffers,
                        const oige_ingenuity_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_ing(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  a.p = *params;
  a.b = *buffers;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  return OIGE_OK;
}

static int check_ing_reset(const IngPre& a, const char* who) {
  const oige_ingenuity_reset_io& r = a.r;
  OIGE_REQUIRE(a.b.reset_buf && a.b.progress_buf && a.b.dof_pos && aligned16(a.b.dof_pos), "%s: null/unaligned reset_buf / progress_buf / dof_pos", who);
  OIGE_REQUIRE(r.initial_root_pos && r.initial_root_rot && r.env_ids && r.count && r.dof_pos_out && r.dof_vel_out && r.root_pos_out &&
                   r.root_rot_out && r.root_vel_out,
               "%s: null reset pointer", who);
  OIGE_REQUIRE(aligned16(r.initial_root_rot) && aligned16(r.root_rot_out) && aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out),
               "%s: row buffers must be 16-byte aligned", who);
  return OIGE_OK;
}

static int check_ing_targets(const IngPre& a, const char* who) {
  const oige_ingenuity_reset_io& r = a.r;
  OIGE_REQUIRE(a.b.target_positions && r.env_pos && r.initial_ball_rot && r.target_ids && r.target_count && r.ball_pos_out && r.ball_rot_out,
               "%s: null set_targets pointer", who);
  OIGE_REQUIRE(aligned16(r.initial_ball_rot) && aligned16(r.ball_rot_out), "%s: (n,4) buffers must be 16-byte aligned", who);
  return OIGE_OK;
}

int oige_ingenuity_pre_physics(const oige_ingenuity_params* params, const float* actions_in, const oige_ingenuity_buffers* buffers,
                               const oige_ingenuity_reset_io* reset_io, int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  IngPre a;
  if (int rc = fill_ing_pre(a, params, buffers, reset_io, num_envs, "ingenuity_pre_physics")) return rc;
  if (int rc = check_ing_reset(a, "ingenuity_pre_physics")) return rc;
  if (int rc = check_ing_targets(a, "ingenuity_pre_physics")) return rc;
  OIGE_REQUIRE(actions_in && (reinterpret_cast<uintptr_t>(actions_in) & 7u) == 0, "ingenuity_pre_physics: null/unaligned actions");
  OIGE_REQUIRE(a.b.thrusts && a.b.dof_vel && aligned16(a.b.dof_vel) && (reinterpret_cast<uintptr_t>(a.b.thrusts) & 7u) == 0,
               "ingenuity_pre_physics: null/unaligned thrusts / dof_vel");
  OIGE_REQUIRE((!a.b.reset_counts_u8 || aligned16(a.b.reset_counts_u8)) && (!a.b.target_counts_u8 || aligned16(a.b.target_counts_u8)),
               "ingenuity_pre_physics: the byte count tables must be 16-byte aligned");
  OIGE_REQUIRE(a.b.reset_workspace && a.b.target_workspace && aligned16(a.b.reset_workspace) && aligned16(a.b.target_workspace),
               "ingenuity_pre_physics: null/unaligned workspaces");
  OIGE_REQUIRE(!a.r.u_reset && !a.r.u_target, "ingenuity_pre_physics: pre-drawn noise goes through oige_ingenuity_reset_idx / _set_targets");
  a.actions_in = actions_in;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID)) {  // the post kernel's DONE stage leaves the same tile counts
    hipLaunchKernelGGL(ingenuity_count_kernel, dim3((unsigned)G), dim3(GROUP), 0, s, a);
    if (int rc = check_launch("ingenuity_count_kernel")) return rc;
  }
  const int rank = pick_rank(G, a.b.reset_counts_u8 && a.b.target_counts_u8 ? a.b.reset_counts_u8 : nullptr);
  if (rank == RANK_U8) {
    hipLaunchKernelGGL(ingenuity_pre_kernel<RANK_U8>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else if (rank == RANK_INT32) {
    hipLaunchKernelGGL(ingenuity_pre_kernel<RANK_INT32>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (int rc = launch_scan_groups2(num_envs, a.b.reset_workspace, a.b.target_workspace, s)) return rc;
    hipLaunchKernelGGL(ingenuity_pre_kernel<RANK_SCANNED>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("ingenuity_pre_kernel");
}

int oige_ingenuity_reset_idx(const oige_ingenuity_params* params, const oige_ingenuity_buffers* buffers,
                             const oige_ingenuity_reset_io* reset_io, int64_t num_envs, int64_t max_ids, oige_stream_t stream) {
  IngPre a;
  if (int rc = fill_ing_pre(a, params, buffers, reset_io, num_envs, "ingenuity_reset_idx")) return rc;
  if (int rc = check_ing_reset(a, "ingenuity_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "ingenuity_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(ingenuity_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("ingenuity_reset_idx_kernel");
}

int oige_ingenuity_set_targets(const oige_ingenuity_params* params, const oige_ingenuity_buffers* buffers,
                               const oige_ingenuity_reset_io* reset_io, int64_t num_envs, int64_t max_ids, oige_stream_t stream) {
  IngPre a;
  if (int rc = fill_ing_pre(a, params, buffers, reset_io, num_envs, "ingenuity_set_targets")) return rc;
  if (int rc = check_ing_targets(a, "ingenuity_set_targets")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "ingenuity_set_targets: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(ingenuity_set_targets_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("ingenuity_set_targets_kernel");
}

}  // extern "C"
