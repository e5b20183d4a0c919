// Anymal (flat ground) per-step task math for gfx950 (MI355X).
//
// Replaces /root/reference/omniisaacgymenvs/tasks/anymal.py behind the unchanged AnymalTask method contract:
//   get_observations :143-178, calculate_metrics :274-309, is_done :311-314, pre_physics_step :180-194,
//   reset_idx :196-231, and AnymalView.is_base_below_threshold (robots/articulations/views/anymal_view.py:76-80).
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, 620 algorithmic bytes per env-step
//   (R pos 12 + rot 16 + vel 24 + commands 12 + progress 8 + 5 x 48 (dof_pos, dof_vel, actions, last_actions,
//    last_dof_vel); W obs 192 + rew 4 + reset 8 + progress 8 + last_actions 48 + last_dof_vel 48).
//   The five (N,12) arrays arrive as lane-linear float4 runs and are transposed through a (64 x 72) LDS tile whose
//   first 48 columns become the observation row; last_actions / last_dof_vel are refreshed straight from the
//   registers that hold actions / dof_vel (:304-305).
// pre_physics: one workgroup per 256 envs; every (N,12) array is walked lane-linear once (actions clone, position
//   targets, zeroed last_* rows of the reset envs, compacted dof rows), the few per-env reset scalars run on dense lanes.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

// lane-linear / per-env whole-line float4 streams outside the tile helpers: non-temporal like those (oige_tile.hpp);
// measured per task at 1 M envs, kept where the whole step gained (Anymal step 0.265 -> 0.253 ms; the same change lost 2-3 % on
// Quadcopter, Ingenuity and FrankaCabinet, whose pre kernels keep plain accesses)
#define LD4S(p) ld4_stream(p)
#define ST4S(p, v) st4_stream((p), (v))

namespace oige {

constexpr int FD = OIGE_ANYMAL_DOFS;        // 12
constexpr int FO = OIGE_ANYMAL_FLAT_OBS;    // 48
constexpr int FROW = FO + 2 * FD;           // LDS row: obs columns + last_actions + last_dof_vel

struct AnymalFlatPost {
  oige_anymal_flat_params p;
  oige_anymal_flat_state s;
  oige_anymal_flat_buffers b;
  int64_t n;
  uint32_t stages;
};

template <bool FULL>
__device__ __forceinline__ void anymal_flat_tile(const AnymalFlatPost& a, float* __restrict__ tile, int64_t t, int nv) {
  constexpr int C_DPOS = 12, C_DVEL = 24, C_ACT = 36, C_LA = 48, C_LDV = 60;
  const oige_anymal_flat_params& p = a.p;
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;

  // ---- A: loads in flight
  Chunk<FD> c_dpos, c_dvel, c_act, c_la, c_ldv;
  load_chunk<FD, FULL>(c_dpos, a.s.dof_pos + env0 * FD, nv * FD);
  load_chunk<FD, FULL>(c_dvel, a.s.dof_vel + env0 * FD, nv * FD);
  load_chunk<FD, FULL>(c_act, a.b.actions + env0 * FD, nv * FD);
  if (do_metrics) {
    load_chunk<FD, FULL>(c_la, a.b.last_actions + env0 * FD, nv * FD);
    load_chunk<FD, FULL>(c_ldv, a.b.last_dof_vel + env0 * FD, nv * FD);
  }
  float pz = 1.0f;
  q4 rot{1.f, 0.f, 0.f, 0.f};
  f3 lin{0.f, 0.f, 0.f}, ang{0.f, 0.f, 0.f}, cmd{0.f, 0.f, 0.f};
  int64_t progress = 0;
  if (live) {
    pz = a.s.base_pos[env * 3 + 2];
    const float4 r4 = *reinterpret_cast<const float4*>(a.s.base_rot + env * 4);
    rot = q4{r4.x, r4.y, r4.z, r4.w};
    const float2* vp = reinterpret_cast<const float2*>(a.s.velocities + env * 6);
    const float2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
    lin = f3{v0.x, v0.y, v1.x};
    ang = f3{v1.y, v2.x, v2.y};
    const float* c3 = a.b.commands + env * 3;
    cmd = f3{c3[0], c3[1], c3[2]};
    if (do_progress || do_done) progress = a.b.progress_buf[env];
  }

  // ---- B: rows into the tile; last_* refreshed from the registers that hold their new values (:304-305)
  scatter_chunk<FD, FROW, C_DPOS, false>(c_dpos, tile, 1.0f);
  scatter_chunk<FD, FROW, C_DVEL, false>(c_dvel, tile, 1.0f);
  scatter_chunk<FD, FROW, C_ACT, false>(c_act, tile, 1.0f);
  if (do_metrics) {
    scatter_chunk<FD, FROW, C_LA, false>(c_la, tile, 1.0f);
    scatter_chunk<FD, FROW, C_LDV, false>(c_ldv, tile, 1.0f);
    const int valid = nv * FD;
#pragma unroll
    for (int it = 0; it < Chunk<FD>::ITERS; ++it) {
      const int e = 4 * (it * WAVE + lane);
      if (FULL || e + 3 < valid) {  // FD % 4 == 0: a float4 is wholly valid or wholly not
        ST4S(a.b.last_actions + env0 * FD + e, c_act.v[it]);
        ST4S(a.b.last_dof_vel + env0 * FD + e, c_dvel.v[it]);
      }
    }
  }
  if (do_progress) progress += 1;  // rl_task.py:289
  __syncthreads();

  // ---- C: one env per lane
  float* row = tile + lane * FROW;
  const f3 blv = quat_rotate_inverse(rot, lin);  // :152 / :283
  const f3 bav = quat_rotate_inverse(rot, ang);  // :153 / :284
  const f3 grav = quat_rotate(rot, f3{0.0f, 0.0f, -1.0f});  // :154 (quat_rotate, not its inverse)
  float dp[FD], dv[FD], act[FD];
#pragma unroll
  for (int j = 0; j < FD; ++j) { dp[j] = row[C_DPOS + j]; dv[j] = row[C_DVEL + j]; act[j] = row[C_ACT + j]; }
  const bool fallen = pz - 0.0f < p.base_threshold;  // anymal_view.py:76-80 with ground_heights = 0.0
  if (do_metrics) {
    const float ex = cmd.x - blv.x, ey = cmd.y - blv.y, ez = cmd.z - bav.z;
    const float lin_vel_error = ex * ex + ey * ey;
    const float ang_vel_error = ez * ez;
    const float rew_lin_vel_xy = expf((-lin_vel_error) / 0.25f) * p.rew_scales[0];
    const float rew_ang_vel_z = expf((-ang_vel_error) / 0.25f) * p.rew_scales[1];
    const float rew_lin_vel_z = (blv.z * blv.z) * p.rew_scales[2];
    float s_acc = 0.f, s_rate = 0.f, s_cos = 0.f;
#pragma unroll
    for (int j = 0; j < FD; ++j) {
      const float d1 = row[C_LDV + j] - dv[j], d2 = row[C_LA + j] - act[j];
      s_acc = j == 0 ? d1 * d1 : s_acc + d1 * d1;
      s_rate = j == 0 ? d2 * d2 : s_rate + d2 * d2;
      if (j < 4) {
        const float h = __builtin_fabsf(dp[j] - p.default_dof_pos[j]);
        s_cos = j == 0 ? h : s_cos + h;
      }
    }
    float total = rew_lin_vel_xy + rew_ang_vel_z;
    total += s_acc * p.rew_scales[3];
    total += s_rate * p.rew_scales[4];
    total += s_cos * p.rew_scales[5];
    total += rew_lin_vel_z;  // :301
    total = (total != total) ? total : (total < 0.0f ? 0.0f : total);  // torch.clip(total, 0.0, None)
    if (fallen) total = -1.0f;                                         // :308
    if (live) a.b.rew_buf[env] = total;
  }
  if (do_done) {  // :313-314, not sticky
    const bool time_out = progress >= (int64_t)p.max_episode_length - 1;
    const bool r = time_out | fallen;
    if (live) a.b.reset_buf[env] = r ? 1 : 0;
    const unsigned long long m = __ballot(live && r);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  if (do_obs) {
    row[0] = blv.x * p.lin_vel_scale; row[1] = blv.y * p.lin_vel_scale; row[2] = blv.z * p.lin_vel_scale;
    row[3] = bav.x * p.ang_vel_scale; row[4] = bav.y * p.ang_vel_scale; row[5] = bav.z * p.ang_vel_scale;
    row[6] = grav.x; row[7] = grav.y; row[8] = grav.z;
    row[9] = cmd.x * p.lin_vel_scale; row[10] = cmd.y * p.lin_vel_scale; row[11] = cmd.z * p.ang_vel_scale;  // :157-161
#pragma unroll
    for (int j = 0; j < FD; ++j) {
      row[C_DPOS + j] = (dp[j] - p.default_dof_pos[j]) * p.dof_pos_scale;  // :155
      row[C_DVEL + j] = dv[j] * p.dof_vel_scale;                            // :170
    }
  }
  __syncthreads();

  // ---- D: the first 48 columns of every row -> obs_buf, one contiguous run (48 % 4 == 0: no float4 straddles rows)
  if (do_obs) {
    constexpr int NF4 = TILE * FO / 4;
    const int valid = nv * FO;
#pragma unroll
    for (int it = 0; it < NF4 / WAVE; ++it) {
      const int e = 4 * (it * WAVE + lane);
      if (FULL || e + 3 < valid) {
        const int r = e / FO, c = e - r * FO;
        *reinterpret_cast<float4*>(a.b.obs_buf + env0 * FO + e) = *reinterpret_cast<const float4*>(tile + r * FROW + c);
      }
    }
  }
}

__global__ __launch_bounds__(WAVE) void anymal_flat_post_kernel(const AnymalFlatPost a) {
  __shared__ __align__(16) float tile[TILE * FROW];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    anymal_flat_tile<true>(a, tile, t, TILE);
  else
    anymal_flat_tile<false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct AnymalFlatPre {
  oige_anymal_flat_params p;
  oige_anymal_flat_buffers b;
  oige_anymal_flat_reset_io r;
  const float* actions_in;
  float* actions_out;
  int64_t n;
  uint32_t flags;
};

// the per-env scalars of reset_idx (:196-231) for env i landing in compacted row k; u3 = U[0,1) draws for x, y, yaw;
// ip / iq = initial_root_pos[i] / initial_root_rot[i], fetched by the caller in front of its first store
__device__ __forceinline__ void anymal_flat_reset_head(const AnymalFlatPre& a, int64_t i, int64_t k, const float (&u3)[3],
                                                       const float (&ip)[3], const float4 iq) {
  const oige_anymal_flat_params& p = a.p;
  a.r.root_pos_out[k * 3 + 0] = ip[0]; a.r.root_pos_out[k * 3 + 1] = ip[1]; a.r.root_pos_out[k * 3 + 2] = ip[2];
  *reinterpret_cast<float4*>(a.r.root_rot_out + k * 4) = iq;
#pragma unroll
  for (int c = 0; c < 6; ++c) a.r.root_vel_out[k * 6 + c] = 0.0f;  // :205
#pragma unroll
  for (int c = 0; c < 3; ++c) a.b.commands[i * 3 + c] = p.cmd_span[c] * u3[c] + p.cmd_lo[c];  // :217-225
  a.b.reset_buf[i] = 0;     // :228
  a.b.progress_buf[i] = 0;  // :229
  a.r.env_ids[k] = i;
}

// Order (r03; quadcopter.hip has the reasoning): two round trips of loads, neither conditional, both in front of the first
// store -- trip 1 the flag (clamped index), the byte-table prefix, the group's action and target quads; trip 2, behind the
// ranks, the dense lanes' initial rows.  The dense lanes' root rows leave last (a wave without a dense lane ends before).
// Per-DOF rows of the params reach LDS through compile-time indices (indexed by tid they are vector loads of the argument block).
template <int RANK>
__global__ __launch_bounds__(GROUP) void anymal_flat_pre_kernel(const AnymalFlatPre a) {
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP], s_rank[GROUP];
  __shared__ float s_lo[FD], s_hi[FD], s_def[FD];
  const oige_anymal_flat_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const int64_t envs_here = (a.n - g * GROUP) < GROUP ? (a.n - g * GROUP) : GROUP;
  const int64_t base = g * GROUP * FD;
  const int nf4 = (int)(envs_here * FD / 4);
  // ---- loads
  const int64_t flag_ld = a.b.reset_buf[env < a.n ? env : g * GROUP];  // :184
  const U8Prefix rp = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  float4 av[FD / 4], ct4[FD / 4];
#pragma unroll
  for (int it = 0; it < FD / 4; ++it) {
    const int f = it * GROUP + tid;
    const int e = 4 * (f < nf4 ? f : 0);  // idle lanes re-read the group's first quad
    av[it] = *reinterpret_cast<const float4*>(a.actions_in + base + e);
    ct4[it] = LD4S(a.b.current_targets + base + e);  // (reset rows read theirs too and ignore it)
  }
  {
    float lo = 0.f, hi = 0.f, de = 0.f;
#pragma unroll
    for (int j = 0; j < FD; ++j) {
      lo = tid == j ? p.dof_lower[j] : lo;
      hi = tid == j ? p.dof_upper[j] : hi;
      de = tid == j ? p.default_dof_pos[j] : de;
    }
    if (tid < FD) {
      s_lo[tid] = lo;
      s_hi[tid] = hi;
      s_def[tid] = de;
    }
  }
  const bool flagged = env < a.n && flag_ld != 0;
  int k0, cnt;  // offset of the group in the id list, flagged envs in the group
  const int64_t k = rank_finish<RANK>(rp, flagged, a.b.reset_workspace, a.b.tile_counts_u8, G, g, lds8, k0, cnt);
  if (flagged) s_list[(int)(k - k0)] = tid;
  s_rank[tid] = flagged ? (int)(k - k0) : -1;
  __syncthreads();
  // ---- trip 2
  const bool dense = tid < cnt;  // dense lanes: thread j takes the group's j-th flagged env
  const int64_t di = g * GROUP + (dense ? s_list[tid] : 0);
  float ip[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) ip[c] = a.r.initial_root_pos[di * 3 + c];
  const float4 iq = *reinterpret_cast<const float4*>(a.r.initial_root_rot + di * 4);

  // ---- stores

  // lane-linear over the group's 256 x 12 floats
#pragma unroll
  for (int it = 0; it < FD / 4; ++it) {
    const int f = it * GROUP + tid;
    if (f >= nf4) continue;
    const int e = 4 * f;
    const int il = e / FD, j0 = e - il * FD;
    const int lr = s_rank[il];
    float4 v = av[it];
    if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
      v.x = clamp_sym(v.x, p.clip_actions); v.y = clamp_sym(v.y, p.clip_actions);
      v.z = clamp_sym(v.z, p.clip_actions); v.w = clamp_sym(v.w, p.clip_actions);
    }
    *reinterpret_cast<float4*>(a.actions_out + base + e) = v;  // self.actions[:] = actions (:189)
    float ct[4] = {ct4[it].x, ct4[it].y, ct4[it].z, ct4[it].w};
    if (lr >= 0) {  // reset row (:199-203, :230-231)
      const int64_t i = g * GROUP + il;
      const u4 c{(uint32_t)i, (uint32_t)(j0 >> 2), (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};  // draws j0 .. j0+3
      const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
      const int64_t row = ((int64_t)k0 + lr) * FD + j0;
      *reinterpret_cast<float4*>(a.r.dof_vel_out + row) =
          make_float4(0.2f * u01(x.x) + (-0.1f), 0.2f * u01(x.y) + (-0.1f), 0.2f * u01(x.z) + (-0.1f), 0.2f * u01(x.w) + (-0.1f));
#pragma unroll
      for (int q = 0; q < 4; ++q) ct[q] = s_def[j0 + q];
      *reinterpret_cast<float4*>(a.r.dof_pos_out + row) = make_float4(ct[0], ct[1], ct[2], ct[3]);
      ST4S(a.b.last_actions + base + e, make_float4(0.f, 0.f, 0.f, 0.f));
      ST4S(a.b.last_dof_vel + base + e, make_float4(0.f, 0.f, 0.f, 0.f));
    }
    const float aa[4] = {v.x, v.y, v.z, v.w};
    float o[4];
#pragma unroll
    for (int q = 0; q < 4; ++q)  // :190-193
      o[q] = tensor_clamp(ct[q] + (p.action_scale * aa[q]) * p.dt, s_lo[j0 + q], s_hi[j0 + q]);
    ST4S(a.b.current_targets + base + e, make_float4(o[0], o[1], o[2], o[3]));
  }
  if (g == G - 1 && tid == 0) *a.r.count = (int64_t)k0 + cnt;
  if (!dense) return;
  const u4 c{(uint32_t)di, 3u, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};  // draws 12, 13, 14: x, y, yaw
  const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
  const float u3[3] = {u01(x.x), u01(x.y), u01(x.z)};
  anymal_flat_reset_head(a, di, (int64_t)k0 + tid, u3, ip, iq);
}


// reset_idx(env_ids) with caller-provided ids and (optionally) pre-drawn noise: one thread per id
__global__ __launch_bounds__(256) void anymal_flat_reset_idx_kernel(const AnymalFlatPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[16];
  if (a.r.u_vel) {
#pragma unroll
    for (int j = 0; j < FD; ++j) u[j] = a.r.u_vel[k * FD + j];
#pragma unroll
    for (int c = 0; c < 3; ++c) u[FD + c] = a.r.u_cmd[k * 3 + c];
  } else {
#pragma unroll
    for (int blk = 0; blk < 4; ++blk) {
      const u4 c{(uint32_t)i, (uint32_t)blk, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
      const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
      u[4 * blk] = u01(x.x); u[4 * blk + 1] = u01(x.y); u[4 * blk + 2] = u01(x.z); u[4 * blk + 3] = u01(x.w);
    }
  }
#pragma unroll
  for (int j = 0; j < FD; ++j) {
    const float d = a.p.default_dof_pos[j];
    a.r.dof_pos_out[k * FD + j] = d;
    a.r.dof_vel_out[k * FD + j] = 0.2f * u[j] + (-0.1f);
    a.b.current_targets[i * FD + j] = d;
    a.b.last_actions[i * FD + j] = 0.0f;
    a.b.last_dof_vel[i * FD + j] = 0.0f;
  }
  const float u3[3] = {u[FD], u[FD + 1], u[FD + 2]};
  const float ip[3] = {a.r.initial_root_pos[i * 3], a.r.initial_root_pos[i * 3 + 1], a.r.initial_root_pos[i * 3 + 2]};
  anymal_flat_reset_head(a, i, k, u3, ip, *reinterpret_cast<const float4*>(a.r.initial_root_rot + i * 4));
}

static int check_flat(const oige_anymal_flat_params* p, const oige_anymal_flat_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_anymal_flat_post_physics(const oige_anymal_flat_params* params, const oige_anymal_flat_state* state,
                                  const oige_anymal_flat_buffers* buffers, int64_t num_envs, uint32_t stages,
                                  oige_stream_t stream) {
  if (int rc = check_flat(params, buffers, num_envs, "anymal_flat_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "anymal_flat_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "anymal_flat_post_physics: bad stages mask 0x%x", stages);
  const oige_anymal_flat_state& s = *state;
  const oige_anymal_flat_buffers& b = *buffers;
  OIGE_REQUIRE(s.base_pos && s.base_rot && s.velocities && s.dof_pos && s.dof_vel, "anymal_flat_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.commands && b.actions && b.last_actions &&
                   b.last_dof_vel,
               "anymal_flat_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.base_rot) && aligned16(s.dof_pos) && aligned16(s.dof_vel) && aligned16(b.obs_buf) &&
                   aligned16(b.actions) && aligned16(b.last_actions) && aligned16(b.last_dof_vel) &&
                   (reinterpret_cast<uintptr_t>(s.velocities) & 7u) == 0,
               "anymal_flat_post_physics: row buffers must be 16-byte aligned (velocities 8)");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  AnymalFlatPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipLaunchKernelGGL(anymal_flat_post_kernel, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("anymal_flat_post_kernel");
}

static int fill_flat_pre(AnymalFlatPre& a, const oige_anymal_flat_params* params, const oige_anymal_flat_buffers* buffers,
                         const oige_anymal_flat_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_flat(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  const oige_anymal_flat_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.commands && b.last_actions && b.last_dof_vel && b.current_targets,
               "%s: null buffer pointer", who);
  OIGE_REQUIRE(r->initial_root_pos && r->initial_root_rot && r->env_ids && r->count && r->dof_pos_out && r->dof_vel_out &&
                   r->root_pos_out && r->root_rot_out && r->root_vel_out,
               "%s: null reset pointer", who);
  OIGE_REQUIRE((r->u_vel == nullptr) == (r->u_cmd == nullptr), "%s: pass both u_vel and u_cmd or neither", who);
  OIGE_REQUIRE(aligned16(r->initial_root_rot) && aligned16(r->root_rot_out) && aligned16(r->dof_pos_out) &&
                   aligned16(r->dof_vel_out) && aligned16(b.last_actions) && aligned16(b.last_dof_vel) &&
                   aligned16(b.current_targets),
               "%s: row buffers must be 16-byte aligned", who);
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  a.actions_out = nullptr;
  return OIGE_OK;
}

int oige_anymal_flat_pre_physics(const oige_anymal_flat_params* params, const float* actions_in,
                                 const oige_anymal_flat_buffers* buffers, const oige_anymal_flat_reset_io* reset_io,
                                 float* actions_out, int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  AnymalFlatPre a;
  if (int rc = fill_flat_pre(a, params, buffers, reset_io, num_envs, "anymal_flat_pre_physics")) return rc;
  OIGE_REQUIRE(actions_in && actions_out && aligned16(actions_in) && aligned16(actions_out),
               "anymal_flat_pre_physics: null/unaligned actions");
  OIGE_REQUIRE(buffers->reset_workspace && aligned16(buffers->reset_workspace), "anymal_flat_pre_physics: null/unaligned workspace");
  OIGE_REQUIRE(!reset_io->u_vel, "anymal_flat_pre_physics: pre-drawn noise goes through oige_anymal_flat_reset_idx");
  a.actions_in = actions_in;
  a.actions_out = actions_out;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(a.b.reset_buf, num_envs, a.b.reset_workspace, s, a.b.tile_counts_u8)) return rc;
  const int rank = pick_rank(G, a.b.tile_counts_u8);
  if (rank == RANK_U8) {
    hipLaunchKernelGGL(anymal_flat_pre_kernel<RANK_U8>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else if (rank == RANK_INT32) {
    hipLaunchKernelGGL(anymal_flat_pre_kernel<RANK_INT32>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (int rc = launch_scan_groups(num_envs, a.b.reset_workspace, s)) return rc;
    hipLaunchKernelGGL(anymal_flat_pre_kernel<RANK_SCANNED>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("anymal_flat_pre_kernel");
}

int oige_anymal_flat_reset_idx(const oige_anymal_flat_params* params, const oige_anymal_flat_buffers* buffers,
                               const oige_anymal_flat_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                               oige_stream_t stream) {
  AnymalFlatPre a;
  if (int rc = fill_flat_pre(a, params, buffers, reset_io, num_envs, "anymal_flat_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "anymal_flat_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(anymal_flat_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream,
                     a, max_ids);
  return check_launch("anymal_flat_reset_idx_kernel");
}

}  // extern "C"
