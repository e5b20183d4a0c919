// Quadcopter per-step task math for gfx950 (MI355X) -- SURVEY.md 8f rank 3.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/quadcopter.py behind the unchanged QuadcopterTask method contract:
//   get_observations :124-141, calculate_metrics :229-251, is_done :253-261, pre_physics_step :143-175, reset_idx :203-227.
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, 216 algorithmic bytes per env-step
//   (R root pos 12 + rot 16 + vel 24 + dof_pos 32 + target 12 + env_pos 12 + progress 8; W obs 84 + rew 4 + reset 8 +
//    progress 8).  dof_pos arrives as a lane-linear float4 run; the (64 x 21) observation tile is assembled in LDS and
//   streamed out as one contiguous run.
// pre_physics: one workgroup per 256 envs; the (N,8) targets and the (N,12) actions are walked lane-linear, the flagged
//   envs' root rows run on dense lanes.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

constexpr int QD = OIGE_QUAD_DOFS;     // 8
constexpr int QO = OIGE_QUAD_OBS;      // 21
constexpr int QA = OIGE_QUAD_ACTIONS;  // 12
constexpr int QROW = QO;               // LDS rows are compact (odd stride: bank-conflict free), so the tile IS the output run

struct QuadPost {
  oige_quadcopter_params p;
  oige_quadcopter_state s;
  oige_quadcopter_buffers b;
  int64_t n;
  uint32_t stages;
};

template <bool FULL>
__device__ __forceinline__ void quad_post_tile(const QuadPost& a, float* __restrict__ tile, int64_t t, int nv) {
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;

  Chunk<QD> c_dof;
  if (do_obs) load_chunk<QD, FULL>(c_dof, a.s.dof_pos + env0 * QD, nv * QD);
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

  const f3 rp{pos.x - epos.x, pos.y - epos.y, pos.z - epos.z};  // root_positions (:129, :230)
  const f3 d{tgt.x - rp.x, tgt.y - rp.y, tgt.z - rp.z};
  const float target_dist = __builtin_sqrtf((d.x * d.x + d.y * d.y) + d.z * d.z);  // :235
  if (do_obs) {
    scatter_chunk<QD, QROW, 13, false>(c_dof, tile, 1.0f);  // :138
    float* row = tile + lane * QROW;
    row[0] = d.x / 3.0f; row[1] = d.y / 3.0f; row[2] = d.z / 3.0f;              // :134
    row[3] = rot.w; row[4] = rot.x; row[5] = rot.y; row[6] = rot.z;            // :135
    row[7] = lin.x / 2.0f; row[8] = lin.y / 2.0f; row[9] = lin.z / 2.0f;       // :136
    row[10] = ang.x / OIGE_PI_F; row[11] = ang.y / OIGE_PI_F; row[12] = ang.z / OIGE_PI_F;  // :137
  }
  if (do_metrics) {  // :229-251
    const float pos_reward = 1.0f / (1.0f + (3.0f * target_dist) * target_dist);
    const f3 ups = quat_rotate(rot, f3{0.0f, 0.0f, 1.0f});  // quat_axis(q, 2)
    const float tiltage = __builtin_fabsf(1.0f - ups.z);
    const float up_reward = 1.0f / (1.0f + (10.0f * tiltage) * tiltage);
    const float spinnage = __builtin_fabsf(ang.z);
    const float spinnage_reward = 1.0f / (1.0f + (0.001f * spinnage) * spinnage);
    float rew = pos_reward + pos_reward * ((up_reward + spinnage_reward) + (spinnage * spinnage) * (-0.0025f));
    rew = (rew != rew) ? rew : (rew < 0.0f ? 0.0f : rew);  // torch.clip(rew, 0.0, None)
    if (live) a.b.rew_buf[env] = rew;
  }
  if (do_done) {  // :253-261, not sticky
    const bool die = (target_dist > 3.0f) | (rp.z < 0.3f);
    const bool r = (progress >= (int64_t)a.p.max_episode_length - 1) | die;
    if (live) a.b.reset_buf[env] = r ? 1 : 0;
    const unsigned long long m = __ballot(live && r);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) store_tile<QO, FULL>(tile, a.b.obs_buf + env0 * QO, nv * QO);  // rows are compact: one contiguous run
}

__global__ __launch_bounds__(WAVE) void quadcopter_post_kernel(const QuadPost a) {
  __shared__ __align__(16) float tile[TILE * QROW];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    quad_post_tile<true>(a, tile, t, TILE);
  else
    quad_post_tile<false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct QuadPre {
  oige_quadcopter_params p;
  oige_quadcopter_buffers b;
  oige_quadcopter_reset_io r;
  const float* actions_in;
  int64_t n;
  uint32_t flags;
};

// U[0,1) draws of env i (in-kernel path): block 0, 1 -> the 8 DOF draws, block 2 -> x, y, z
__device__ __forceinline__ u4 quad_block(const QuadPre& a, int64_t i, uint32_t blk) {
  const u4 c{(uint32_t)i, blk, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
  return philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
}

// the root rows and flags of reset_idx (:209-227) for env i landing in compacted row k; ip / iq = initial_root_pos[i] /
// initial_root_rot[i], fetched by the caller in front of its first store
__device__ __forceinline__ void quad_reset_head(const QuadPre& a, int64_t i, int64_t k, const float (&u3)[3], const float (&ip)[3],
                                                const float4 iq) {
  a.r.root_pos_out[k * 3 + 0] = ip[0] + (3.0f * u3[0] + (-1.5f));   // torch_rand_float(-1.5, 1.5)
  a.r.root_pos_out[k * 3 + 1] = ip[1] + (3.0f * u3[1] + (-1.5f));
  a.r.root_pos_out[k * 3 + 2] = ip[2] + (1.7f * u3[2] + (-0.2f));   // torch_rand_float(-0.2, 1.5)
  *reinterpret_cast<float4*>(a.r.root_rot_out + k * 4) = iq;
#pragma unroll
  for (int c = 0; c < 6; ++c) a.r.root_vel_out[k * 6 + c] = 0.0f;
  a.b.reset_buf[i] = 0;
  a.b.progress_buf[i] = 0;
  a.r.env_ids[k] = i;
}

__device__ __forceinline__ float4 quad_dof_draw(const u4 x) {  // torch_rand_float(-0.2, 0.2): (0.2 - -0.2) * u + -0.2
  return make_float4(0.4f * u01(x.x) + (-0.2f), 0.4f * u01(x.y) + (-0.2f), 0.4f * u01(x.z) + (-0.2f), 0.4f * u01(x.w) + (-0.2f));
}

// Order (r03): two round trips of loads, neither conditional, both in front of the first store.  Trip 1: the flag (through a
// clamped index: written as `live && flag[env]` it was a branch around a load with a full wait behind it, a serial round trip
// in front of everything else), the target and action quads of the lane-linear walk, the env's thrusts and thrust actions.
// Trip 2, behind the ranks: the dense lanes' initial rows (idle lanes re-read the group's first env).  Then the elementwise
// stores; the dense lanes' reset rows last, where no other wave waits for their Philox draw (a wave without a dense lane ends
// before).  The DOF limit rows reach LDS through compile-time indices (indexed by tid they are VECTOR loads of the argument
// block).  A load issued behind a store waits for that store's round trip.
template <int RANK>
__global__ __launch_bounds__(GROUP) void quadcopter_pre_kernel(const QuadPre a) {
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP], s_rank[GROUP];
  __shared__ float s_lo[QD], s_hi[QD];
  const oige_quadcopter_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const bool live = env < a.n;
  const int64_t envs_here = (a.n - g * GROUP) < GROUP ? (a.n - g * GROUP) : GROUP;
  const int nf4 = (int)(envs_here * QD / 4);
  // ---- trip 1
  const int64_t le = live ? env : g * GROUP;
  const int64_t flag_ld = a.b.reset_buf[le];  // :147
  const U8Prefix rp = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  float4 tq[QD / 4], aq[QD / 4];
#pragma unroll
  for (int it = 0; it < QD / 4; ++it) {
    const int f = it * GROUP + tid;
    const int e = 4 * (f < nf4 ? f : 0);  // idle lanes re-read the group's first quad
    const int il = e / QD, j0 = e - il * QD;
    const int64_t i = g * GROUP + il;
    tq[it] = *reinterpret_cast<const float4*>(a.b.dof_position_targets + i * QD + j0);  // (reset rows read theirs too and ignore it)
    aq[it] = *reinterpret_cast<const float4*>(a.actions_in + i * QA + j0);
  }
  const float4 th_in = *reinterpret_cast<const float4*>(a.b.thrusts + le * 4);
  float4 ath = *reinterpret_cast<const float4*>(a.actions_in + le * QA + 8);
  {
    float lo = 0.f, hi = 0.f;
#pragma unroll
    for (int j = 0; j < QD; ++j) {
      lo = tid == j ? p.dof_lower[j] : lo;
      hi = tid == j ? p.dof_upper[j] : hi;
    }
    if (tid < QD) {
      s_lo[tid] = lo;
      s_hi[tid] = hi;
    }
  }
  const bool flagged = live && flag_ld != 0;
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
  const float thrust_rate = p.thrust_rate;
  // DOF position targets: lane-linear over the group's 256 x 8 floats; the matching actions are columns 0:8 of a 12-wide row
#pragma unroll
  for (int it = 0; it < QD / 4; ++it) {
    const int f = it * GROUP + tid;
    if (f >= nf4) continue;
    const int e = 4 * f;
    const int il = e / QD, j0 = e - il * QD;
    const int64_t i = g * GROUP + il;
    const int lr = s_rank[il];
    float4 out;
    if (lr >= 0) {  // reset env: targets = its re-drawn DOF positions (:171), also the compacted dof rows (:206-207)
      out = quad_dof_draw(quad_block(a, i, (uint32_t)(j0 >> 2)));
      const int64_t row = ((int64_t)k0 + lr) * QD + j0;
      *reinterpret_cast<float4*>(a.r.dof_pos_out + row) = out;
      *reinterpret_cast<float4*>(a.r.dof_vel_out + row) = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      const float4 t4 = tq[it];
      float4 av = aq[it];
      if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
        av.x = clamp_sym(av.x, p.clip_actions); av.y = clamp_sym(av.y, p.clip_actions);
        av.z = clamp_sym(av.z, p.clip_actions); av.w = clamp_sym(av.w, p.clip_actions);
      }
      out.x = tensor_clamp(t4.x + p.dof_rate * av.x, s_lo[j0 + 0], s_hi[j0 + 0]);  // :154-157
      out.y = tensor_clamp(t4.y + p.dof_rate * av.y, s_lo[j0 + 1], s_hi[j0 + 1]);
      out.z = tensor_clamp(t4.z + p.dof_rate * av.z, s_lo[j0 + 2], s_hi[j0 + 2]);
      out.w = tensor_clamp(t4.w + p.dof_rate * av.w, s_lo[j0 + 3], s_hi[j0 + 3]);
    }
    *reinterpret_cast<float4*>(a.b.dof_position_targets + i * QD + j0) = out;
  }
  // thrusts (N,4) and forces (N,4,3): one env per thread
  if (live) {
    float4 th = make_float4(0.f, 0.f, 0.f, 0.f);
    if (!flagged) {
      const float4 t4 = th_in;
      float4 av = ath;
      if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
        av.x = clamp_sym(av.x, p.clip_actions); av.y = clamp_sym(av.y, p.clip_actions);
        av.z = clamp_sym(av.z, p.clip_actions); av.w = clamp_sym(av.w, p.clip_actions);
      }
      th.x = tensor_clamp(t4.x + thrust_rate * av.x, -p.max_thrust, p.max_thrust);  // :160-161
      th.y = tensor_clamp(t4.y + thrust_rate * av.y, -p.max_thrust, p.max_thrust);
      th.z = tensor_clamp(t4.z + thrust_rate * av.z, -p.max_thrust, p.max_thrust);
      th.w = tensor_clamp(t4.w + thrust_rate * av.w, -p.max_thrust, p.max_thrust);
    }
    *reinterpret_cast<float4*>(a.b.thrusts + env * 4) = th;  // reset envs: 0 (:169)
    float4* fo = reinterpret_cast<float4*>(a.b.forces + env * 12);
    if (flagged) {  // :170
      fo[0] = fo[1] = fo[2] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {        // :163-166: only the z components are written, x and y keep their values
      float* fz = a.b.forces + env * 12;
      fz[2] = th.x; fz[5] = th.y; fz[8] = th.z; fz[11] = th.w;
    }
  }
  if (g == G - 1 && tid == 0) *a.r.count = (int64_t)k0 + cnt;
  // ---- the dense lanes' root rows and flags (:209-227)
  if (!dense) return;
  const u4 x = quad_block(a, di, 2u);
  const float u3[3] = {u01(x.x), u01(x.y), u01(x.z)};
  quad_reset_head(a, di, (int64_t)k0 + tid, u3, ip, iq);
}

// reset_idx(env_ids) with caller-provided ids and (optionally) pre-drawn U[0,1): one thread per id
__global__ __launch_bounds__(256) void quadcopter_reset_idx_kernel(const QuadPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float4 d0, d1;
  float u3[3];
  if (a.r.u_dof) {
    const float* u = a.r.u_dof + k * QD;
    d0 = make_float4(0.4f * u[0] + (-0.2f), 0.4f * u[1] + (-0.2f), 0.4f * u[2] + (-0.2f), 0.4f * u[3] + (-0.2f));
    d1 = make_float4(0.4f * u[4] + (-0.2f), 0.4f * u[5] + (-0.2f), 0.4f * u[6] + (-0.2f), 0.4f * u[7] + (-0.2f));
    u3[0] = a.r.u_xyz[k * 3]; u3[1] = a.r.u_xyz[k * 3 + 1]; u3[2] = a.r.u_xyz[k * 3 + 2];
  } else {
    d0 = quad_dof_draw(quad_block(a, i, 0u));
    d1 = quad_dof_draw(quad_block(a, i, 1u));
    const u4 x = quad_block(a, i, 2u);
    u3[0] = u01(x.x); u3[1] = u01(x.y); u3[2] = u01(x.z);
  }
  float4* dp = reinterpret_cast<float4*>(a.r.dof_pos_out + k * QD);
  float4* dv = reinterpret_cast<float4*>(a.r.dof_vel_out + k * QD);
  dp[0] = d0; dp[1] = d1;
  dv[0] = dv[1] = make_float4(0.f, 0.f, 0.f, 0.f);
  const float ip[3] = {a.r.initial_root_pos[i * 3], a.r.initial_root_pos[i * 3 + 1], a.r.initial_root_pos[i * 3 + 2]};
  quad_reset_head(a, i, k, u3, ip, *reinterpret_cast<const float4*>(a.r.initial_root_rot + i * 4));
}

static int check_quad(const oige_quadcopter_params* p, const oige_quadcopter_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_quadcopter_post_physics(const oige_quadcopter_params* params, const oige_quadcopter_state* state,
                                 const oige_quadcopter_buffers* buffers, int64_t num_envs, uint32_t stages,
                                 oige_stream_t stream) {
  if (int rc = check_quad(params, buffers, num_envs, "quadcopter_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "quadcopter_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "quadcopter_post_physics: bad stages mask 0x%x", stages);
  const oige_quadcopter_state& s = *state;
  const oige_quadcopter_buffers& b = *buffers;
  OIGE_REQUIRE(s.root_pos && s.root_rot && s.root_velocities && s.dof_pos && s.env_pos, "quadcopter_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.target_positions,
               "quadcopter_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.root_rot) && aligned16(s.dof_pos) && aligned16(b.obs_buf) &&
                   (reinterpret_cast<uintptr_t>(s.root_velocities) & 7u) == 0,
               "quadcopter_post_physics: row buffers must be 16-byte aligned (velocities 8)");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  QuadPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipLaunchKernelGGL(quadcopter_post_kernel, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("quadcopter_post_kernel");
}

static int fill_quad_pre(QuadPre& a, const oige_quadcopter_params* params, const oige_quadcopter_buffers* buffers,
                         const oige_quadcopter_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_quad(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  const oige_quadcopter_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf, "%s: null buffer pointer", who);
  OIGE_REQUIRE(r->initial_root_pos && r->initial_root_rot && r->env_ids && r->count && r->dof_pos_out && r->dof_vel_out &&
                   r->root_pos_out && r->root_rot_out && r->root_vel_out,
               "%s: null reset pointer", who);
  OIGE_REQUIRE((r->u_dof == nullptr) == (r->u_xyz == nullptr), "%s: pass both u_dof and u_xyz or neither", who);
  OIGE_REQUIRE(aligned16(r->initial_root_rot) && aligned16(r->root_rot_out) && aligned16(r->dof_pos_out) && aligned16(r->dof_vel_out),
               "%s: row buffers must be 16-byte aligned", who);
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  return OIGE_OK;
}

int oige_quadcopter_pre_physics(const oige_quadcopter_params* params, const float* actions_in,
                                const oige_quadcopter_buffers* buffers, const oige_quadcopter_reset_io* reset_io,
                                int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  QuadPre a;
  if (int rc = fill_quad_pre(a, params, buffers, reset_io, num_envs, "quadcopter_pre_physics")) return rc;
  const oige_quadcopter_buffers& b = *buffers;
  OIGE_REQUIRE(actions_in && aligned16(actions_in), "quadcopter_pre_physics: null/unaligned actions");
  OIGE_REQUIRE(b.dof_position_targets && b.thrusts && b.forces && aligned16(b.dof_position_targets) && aligned16(b.thrusts) &&
                   aligned16(b.forces),
               "quadcopter_pre_physics: null/unaligned dof_position_targets / thrusts / forces");
  OIGE_REQUIRE(b.reset_workspace && aligned16(b.reset_workspace), "quadcopter_pre_physics: null/unaligned workspace");
  OIGE_REQUIRE(!reset_io->u_dof, "quadcopter_pre_physics: pre-drawn noise goes through oige_quadcopter_reset_idx");
  a.actions_in = actions_in;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(a.b.reset_buf, num_envs, a.b.reset_workspace, s, a.b.tile_counts_u8)) return rc;
  const int rank = pick_rank(G, a.b.tile_counts_u8);
  if (rank == RANK_U8) {
    hipLaunchKernelGGL(quadcopter_pre_kernel<RANK_U8>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else if (rank == RANK_INT32) {
    hipLaunchKernelGGL(quadcopter_pre_kernel<RANK_INT32>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (int rc = launch_scan_groups(num_envs, a.b.reset_workspace, s)) return rc;
    hipLaunchKernelGGL(quadcopter_pre_kernel<RANK_SCANNED>, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("quadcopter_pre_kernel");
}

int oige_quadcopter_reset_idx(const oige_quadcopter_params* params, const oige_quadcopter_buffers* buffers,
                              const oige_quadcopter_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                              oige_stream_t stream) {
  QuadPre a;
  if (int rc = fill_quad_pre(a, params, buffers, reset_io, num_envs, "quadcopter_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "quadcopter_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(quadcopter_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a,
                     max_ids);
  return check_launch("quadcopter_reset_idx_kernel");
}

}  // extern "C"
