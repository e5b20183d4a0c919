// BallBalance per-step task math for gfx950 (MI355X) -- SURVEY.md 8f rank 3.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/ball_balance.py behind the unchanged BallBalanceTask method contract:
//   get_observations :159-185, calculate_metrics :286-299, is_done :301-308 (sticky), pre_physics_step :187-206,
//   reset_idx :208-260 (torch_random_dir_2 restated: angle = U(-pi, pi), (cos, sin)).
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, ~300 algorithmic bytes per env-step with 6 DOFs
//   (R ball pos 12 + vel 24 + dof pos/vel 48 + sensors 72 + env_pos 12 + progress 8 + reset 8; W obs 96 + rew 4 + reset 8 +
//   progress 8).  The (N,3,6) sensor rows arrive as a lane-linear float4 run through LDS; the (64 x 24) observation tile
//   is assembled in LDS and streamed out as one contiguous run.
// pre_physics: one workgroup per 256 envs, one env per thread (rows of at most 8 floats), flagged envs on dense lanes.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

constexpr int BO = OIGE_BBOT_OBS;        // 24
constexpr int BS = 18;                   // 3 sensors x 6 channels
constexpr int BMAXD = OIGE_BBOT_MAX_DOFS;

struct BallPost {
  oige_ball_balance_params p;
  oige_ball_balance_state s;
  oige_ball_balance_buffers b;
  int64_t n;
  uint32_t stages;
};

template <bool FULL>
__device__ __forceinline__ void ball_post_tile(const BallPost& a, float* __restrict__ obs_t, float* __restrict__ sen_t, int64_t t,
                                               int nv) {
  const oige_ball_balance_params& p = a.p;
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;

  Chunk<BS> c_sen;
  if (do_obs) load_chunk<BS, FULL>(c_sen, a.s.sensors + env0 * BS, nv * BS);
  f3 bpos{0.f, 0.f, 1.f}, epos{0.f, 0.f, 0.f}, lin{0.f, 0.f, 0.f};
  float q3[3] = {0.f, 0.f, 0.f}, qd3[3] = {0.f, 0.f, 0.f};
  int64_t progress = 0, reset_in = 0;
  if (live) {
    const float* p3 = a.s.ball_pos + env * 3;
    bpos = f3{p3[0], p3[1], p3[2]};
    const float* e3 = a.s.env_pos + env * 3;
    epos = f3{e3[0], e3[1], e3[2]};
    const float2* vp = reinterpret_cast<const float2*>(a.s.ball_velocities + env * 6);
    const float2 v0 = vp[0], v1 = vp[1];
    lin = f3{v0.x, v0.y, v1.x};
    if (do_obs) {
#pragma unroll
      for (int k = 0; k < 3; ++k) {
        q3[k] = a.s.dof_pos[env * p.num_dof + p.actuated_dof_indices[k]];
        qd3[k] = a.s.dof_vel[env * p.num_dof + p.actuated_dof_indices[k]];
      }
    }
    if (do_progress || do_done) progress = a.b.progress_buf[env];
    if (do_done) reset_in = a.b.reset_buf[env];
  }
  if (do_progress) progress += 1;  // rl_task.py:289
  const f3 bp{bpos.x - epos.x, bpos.y - epos.y, bpos.z - epos.z};  // ball_positions (:161)
  if (do_obs) {
    scatter_chunk<BS, BS, 0, false>(c_sen, sen_t, 1.0f);
    float* row = obs_t + lane * BO;
    row[0] = q3[0]; row[1] = q3[1]; row[2] = q3[2];
    row[3] = qd3[0]; row[4] = qd3[1]; row[5] = qd3[2];
    row[6] = bp.x; row[7] = bp.y; row[8] = bp.z;
    row[9] = lin.x; row[10] = lin.y; row[11] = lin.z;
  }
  if (do_metrics) {  // :286-299
    const float dz = bp.z - 0.7f;
    const float ball_dist = __builtin_sqrtf((bp.x * bp.x + dz * dz) + bp.y * bp.y);
    const float ball_speed = __builtin_sqrtf((lin.x * lin.x + lin.y * lin.y) + lin.z * lin.z);
    const float pos_reward = 1.0f / (1.0f + ball_dist);
    const float speed_reward = 1.0f / (1.0f + ball_speed);
    if (live) a.b.rew_buf[env] = pos_reward * speed_reward;
  }
  if (do_done) {  // :301-308, sticky
    int64_t r = progress >= (int64_t)p.max_episode_length - 1 ? 1 : reset_in;
    r = bp.z < p.ball_low ? 1 : r;
    if (live) a.b.reset_buf[env] = r;
    const unsigned long long m = __ballot(live && r != 0);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) {
    {  // sensor channels 0, 3, 4, 5 of the three sensors, / 20 (:171-174): obs[12 + 3 c' + s] = sensors[s][c]
      const float* sr = sen_t + lane * BS;
      float* row = obs_t + lane * BO;
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        row[12 + s] = sr[s * 6 + 0] / 20.0f;
        row[15 + s] = sr[s * 6 + 3] / 20.0f;
        row[18 + s] = sr[s * 6 + 4] / 20.0f;
        row[21 + s] = sr[s * 6 + 5] / 20.0f;
      }
    }
    __syncthreads();
    store_tile<BO, FULL>(obs_t, a.b.obs_buf + env0 * BO, nv * BO);
  }
}

__global__ __launch_bounds__(WAVE) void ball_balance_post_kernel(const BallPost a) {
  __shared__ __align__(16) float obs_t[TILE * BO];
  __shared__ __align__(16) float sen_t[TILE * BS];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    ball_post_tile<true>(a, obs_t, sen_t, t, TILE);
  else
    ball_post_tile<false>(a, obs_t, sen_t, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct BallPre {
  oige_ball_balance_params p;
  oige_ball_balance_buffers b;
  oige_ball_balance_reset_io r;
  const float* actions_in;
  int64_t n;
  uint32_t flags;
};

// the initial rows of env i that reset_idx copies out (:228-246), fetched in one go in front of the caller's first store
struct BallRows {
  float bp[3], bot_pos[3], bot_vel[6], dof[BMAXD];
  float4 bot_rot;
};
__device__ __forceinline__ BallRows ball_fetch_rows(const BallPre& a, int64_t i) {
  const oige_ball_balance_reset_io& r = a.r;
  const int D = a.p.num_dof;
  BallRows w;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    w.bp[c] = r.initial_ball_pos[i * 3 + c];
    w.bot_pos[c] = r.initial_bot_pos[i * 3 + c];
  }
  w.bot_rot = *reinterpret_cast<const float4*>(r.initial_bot_rot + i * 4);
#pragma unroll
  for (int c = 0; c < 6; ++c) w.bot_vel[c] = r.initial_bot_velocities[i * 6 + c];
#pragma unroll
  for (int j = 0; j < BMAXD; ++j) w.dof[j] = r.initial_dof_positions[i * D + (j < D ? j : 0)];  // (no conditional load)
  return w;
}

// reset_idx (:208-260) for env i landing in compacted row k; u = the five U[0,1) draws (dists, dirs, hspeeds, vspeeds, height)
__device__ __forceinline__ void ball_reset_env(const BallPre& a, int64_t i, int64_t k, const float (&u)[5], const BallRows& w) {
  const oige_ball_balance_reset_io& r = a.r;
  const int D = a.p.num_dof;
  const float dists = 0.399f * u[0] + 0.001f;                      // torch_rand_float(min_d, max_d)
  const float angle = OIGE_TWO_PI_F * u[1] + (-OIGE_PI_F);        // torch_random_dir_2: torch_rand_float(-pi, pi)
  const float dx = cosf(angle), dy = sinf(angle);
  const float speedscales = (dists - 0.001f) / 0.399f;
  const float hspeeds = 2.0f * u[2] + 0.0f;
  const float vspeeds = -(0.0f * u[3] + 5.0f);
  r.ball_pos_out[k * 3 + 0] = w.bp[0] + dists * dx;
  r.ball_pos_out[k * 3 + 1] = w.bp[1] + dists * dy;
  r.ball_pos_out[k * 3 + 2] = w.bp[2] + (1.0f * u[4] + 1.0f);
  *reinterpret_cast<float4*>(r.ball_rot_out + k * 4) = make_float4(1.f, 0.f, 0.f, 0.f);
  float* bv = r.ball_vel_out + k * 6;
  bv[0] = ((-speedscales) * hspeeds) * dx;
  bv[1] = ((-speedscales) * hspeeds) * dy;
  bv[2] = vspeeds;
  bv[3] = bv[4] = bv[5] = 0.0f;
#pragma unroll
  for (int c = 0; c < 3; ++c) r.bot_pos_out[k * 3 + c] = w.bot_pos[c];
  *reinterpret_cast<float4*>(r.bot_rot_out + k * 4) = w.bot_rot;
#pragma unroll
  for (int c = 0; c < 6; ++c) r.bot_vel_out[k * 6 + c] = w.bot_vel[c];
#pragma unroll
  for (int j = 0; j < BMAXD; ++j)
    if (j < D) r.dof_pos_out[k * D + j] = w.dof[j];
  a.b.reset_buf[i] = 0;
  a.b.progress_buf[i] = 0;
  r.env_ids[k] = i;
}

__device__ __forceinline__ void ball_draw(const BallPre& a, int64_t i, float (&u)[5]) {
  const u4 c0{(uint32_t)i, 0u, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
  const u4 c1{(uint32_t)i, 1u, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
  const u4 x = philox4x32_10(c0, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
  const u4 y = philox4x32_10(c1, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
  u[0] = u01(x.x); u[1] = u01(x.y); u[2] = u01(x.z); u[3] = u01(x.w); u[4] = u01(y.x);
}

// Two round trips of loads, neither conditional, both in front of the first store (r03; quadcopter.hip has the reasoning):
// trip 1 the flag (clamped index), the env's own target row and actions, the byte-table prefix; trip 2, behind the ranks, the
// dense lanes' initial rows (r02 fetched them one by one between the stores of reset_idx: thirteen serial round trips for every
// workgroup with a flagged env).  The dense lanes' rows leave last; a wave without a dense lane ends before.
template <int RANK>
__global__ __launch_bounds__(GROUP) void ball_balance_pre_kernel(const BallPre a) {
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP];
  const oige_ball_balance_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const bool live = env < a.n;
  const int D = p.num_dof;
  const int64_t le = live ? env : g * GROUP;
  // ---- trip 1
  const int64_t flag_ld = a.b.reset_buf[le];  // :191
  const U8Prefix rp = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  float tgv[BMAXD], ain[3];
#pragma unroll
  for (int j = 0; j < BMAXD; ++j) tgv[j] = a.b.dof_position_targets[le * D + (j < D ? j : 0)];
#pragma unroll
  for (int c = 0; c < 3; ++c) ain[c] = a.actions_in[le * 3 + c];
  const bool flagged = live && flag_ld != 0;
  int k0, cnt;  // offset of the group in the id list, flagged envs in the group
  const int64_t k = rank_finish<RANK>(rp, flagged, a.b.reset_workspace, a.b.tile_counts_u8, G, g, lds8, k0, cnt);
  if (flagged) s_list[(int)(k - k0)] = tid;
  __syncthreads();
  // ---- trip 2: dense lanes, thread j takes the group's j-th flagged env (idle lanes re-read the group's first env)
  const bool dense = tid < cnt;
  const int64_t di = g * GROUP + (dense ? s_list[tid] : 0);
  const BallRows rows = ball_fetch_rows(a, di);
  // ---- stores
  if (live) {  // :196-204
    float* tg = a.b.dof_position_targets + env * D;
    if (flagged) {
#pragma unroll
      for (int j = 0; j < BMAXD; ++j)
        if (j < D) tg[j] = 0.0f;
    } else {
      float add[BMAXD];
#pragma unroll
      for (int j = 0; j < BMAXD; ++j) add[j] = 0.0f;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float av = ain[c];
        if (a.flags & OIGE_PRE_CLAMP_ACTIONS) av = clamp_sym(av, p.clip_actions);
        const int j = p.actuated_dof_indices[c];
#pragma unroll
        for (int q = 0; q < BMAXD; ++q) add[q] = q == j ? p.rate * av : add[q];
      }
#pragma unroll
      for (int j = 0; j < BMAXD; ++j) {
        if (j < D) {
          const bool act = j == p.actuated_dof_indices[0] || j == p.actuated_dof_indices[1] || j == p.actuated_dof_indices[2];
          const float v = act ? tgv[j] + add[j] : tgv[j];
          tg[j] = tensor_clamp(v, p.dof_lower[j], p.dof_upper[j]);
        }
      }
    }
  }
  if (g == G - 1 && tid == 0) *a.r.count = (int64_t)k0 + cnt;
  if (!dense) return;
  float u[5];
  ball_draw(a, di, u);
  ball_reset_env(a, di, (int64_t)k0 + tid, u, rows);
}

__global__ __launch_bounds__(256) void ball_balance_reset_idx_kernel(const BallPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[5];
  if (a.r.u) {
#pragma unroll
    for (int c = 0; c < 5; ++c) u[c] = a.r.u[k * 5 + c];
  } else {
    ball_draw(a, i, u);
  }
  const BallRows rows = ball_fetch_rows(a, i);
  ball_reset_env(a, i, k, u, rows);
}

static int check_ball(const oige_ball_balance_params* p, const oige_ball_balance_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  OIGE_REQUIRE(p->num_dof >= 3 && p->num_dof <= OIGE_BBOT_MAX_DOFS, "%s: num_dof %d outside 3..%d", who, p->num_dof, OIGE_BBOT_MAX_DOFS);
  for (int k = 0; k < 3; ++k)
    OIGE_REQUIRE(p->actuated_dof_indices[k] >= 0 && p->actuated_dof_indices[k] < p->num_dof, "%s: bad actuated_dof_indices", who);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_ball_balance_post_physics(const oige_ball_balance_params* params, const oige_ball_balance_state* state,
                                   const oige_ball_balance_buffers* buffers, int64_t num_envs, uint32_t stages,
                                   oige_stream_t stream) {
  if (int rc = check_ball(params, buffers, num_envs, "ball_balance_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "ball_balance_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "ball_balance_post_physics: bad stages mask 0x%x", stages);
  const oige_ball_balance_state& s = *state;
  const oige_ball_balance_buffers& b = *buffers;
  OIGE_REQUIRE(s.ball_pos && s.ball_velocities && s.dof_pos && s.dof_vel && s.sensors && s.env_pos,
               "ball_balance_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf, "ball_balance_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.sensors) && aligned16(b.obs_buf) && (reinterpret_cast<uintptr_t>(s.ball_velocities) & 7u) == 0,
               "ball_balance_post_physics: sensors / obs_buf must be 16-byte aligned (velocities 8)");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  BallPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipLaunchKernelGGL(ball_balance_post_kernel, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("ball_balance_post_kernel");
}

static int fill_ball_pre(BallPre& a, const oige_ball_balance_params* params, const oige_ball_balance_buffers* buffers,
                         const oige_ball_balance_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_ball(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  const oige_ball_balance_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf, "%s: null buffer pointer", who);
  OIGE_REQUIRE(r->initial_ball_pos && r->initial_bot_pos && r->initial_bot_rot && r->initial_bot_velocities &&
                   r->initial_dof_positions && r->env_ids && r->count && r->ball_pos_out && r->ball_rot_out && r->ball_vel_out &&
                   r->bot_pos_out && r->bot_rot_out && r->bot_vel_out && r->dof_pos_out,
               "%s: null reset pointer", who);
  OIGE_REQUIRE(aligned16(r->initial_bot_rot) && aligned16(r->ball_rot_out) && aligned16(r->bot_rot_out),
               "%s: (n,4) buffers must be 16-byte aligned", who);
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  return OIGE_OK;
}

int oige_ball_balance_pre_physics(const oige_ball_balance_params* params, const float* actions_in,
                                  const oige_ball_balance_buffers* buffers, const oige_ball_balance_reset_io* reset_io,
                                  int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  BallPre a;
  if (int rc = fill_ball_pre(a, params, buffers, reset_io, num_envs, "ball_balance_pre_physics")) return rc;
  OIGE_REQUIRE(actions_in && buffers->dof_position_targets, "ball_balance_pre_physics: null actions / dof_position_targets");
  OIGE_REQUIRE(buffers->reset_workspace && aligned16(buffers->reset_workspace), "ball_balance_pre_physics: null/unaligned workspace");
  OIGE_REQUIRE(!reset_io->u, "ball_balance_pre_physics: pre-drawn noise goes through oige_ball_balance_reset_idx");
  a.actions_in = actions_in;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(a.b.reset_buf, num_envs, a.b.reset_workspace, s, a.b.tile_counts_u8)) return rc;
  const int rank = pick_rank(G, a.b.tile_counts_u8);
  if (rank == RANK_U8) {
    hipLaunchKernelGGL(ball_balance_pre_kernel<RANK_U8>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else if (rank == RANK_INT32) {
    hipLaunchKernelGGL(ball_balance_pre_kernel<RANK_INT32>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (int rc = launch_scan_groups(num_envs, a.b.reset_workspace, s)) return rc;
    hipLaunchKernelGGL(ball_balance_pre_kernel<RANK_SCANNED>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("ball_balance_pre_kernel");
}

int oige_ball_balance_reset_idx(const oige_ball_balance_params* params, const oige_ball_balance_buffers* buffers,
                                const oige_ball_balance_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                oige_stream_t stream) {
  BallPre a;
  if (int rc = fill_ball_pre(a, params, buffers, reset_io, num_envs, "ball_balance_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "ball_balance_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(ball_balance_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     max_ids);
  return check_launch("ball_balance_reset_idx_kernel");
}

}  // extern "C"
