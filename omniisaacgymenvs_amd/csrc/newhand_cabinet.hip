// TomoDJnewhandCabinet / TomoDJnewhandTendonCabinet / TomoSPCnewhandCabinet per-step task math for gfx950 (MI355X) --
// SURVEY.md 8f rank 4.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/tomodj_newhand_cabinet.py (26 DOFs), tasks/tomodj_newhandTendon_cabinet.py
// (26 DOFs, 22 actuated, terminate_buf) and tasks/tomospc_newhand_cabinet.py (25 DOFs) behind their unchanged task method
// contract: get_observations (five drawer + five fingertip keypoints), calculate_metrics (compute_*_reward), is_done,
// pre_physics_step, reset_idx.
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound.  Algorithmic bytes per env-step at D DOFs, A actions:
//   R hand_rot 16 + 5 x (finger pos 12 + rot 16) + drawer 28 + dof_pos 4D + dof_vel 4D + cabinet rows 32 + actions 4A +
//   progress 8 + reset 8 (+ terminate 8);  W obs 4(2D+5) + rew 4 + reset 8 + progress 8 + keypoints 120 (+ terminate 8).
//   D = A = 26: 544 + 368 = 912.  The (N,D) / (N,A) inputs arrive as lane-linear float4 runs and are transposed through the
//   (64 x (2D+5)) observation tile in LDS (odd row stride), which then streams out as one contiguous run.
// pre_physics: one workgroup per 256 envs; flagged envs re-draw their joint positions on dense lanes and store their target
//   rows, then the group's 256 x A actions are walked lane-linear, each integrating its joint's target.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

// lane-linear / per-env whole-line float4 streams outside the tile helpers: non-temporal like those (oige_tile.hpp);
// measured per task at 1 M envs, kept where the whole step gained (TomoDJnewhandCabinet step 0.250 -> 0.245 ms; the same change lost 2-3 % on
// Quadcopter, Ingenuity and FrankaCabinet, whose pre kernels keep plain accesses)
#define LD4S(p) ld4_stream(p)
#define ST4S(p, v) st4_stream((p), (v))

namespace oige {

constexpr int NMAX = OIGE_NEWHAND_MAX_DOFS;  // 26
constexpr int NK = OIGE_NEWHAND_KEYPOINTS;   // 5

struct NhPost {
  oige_newhand_cabinet_params p;
  oige_newhand_cabinet_state s;
  oige_newhand_cabinet_buffers b;
  int64_t n;
  uint32_t stages;
};

OIGE_DEV f3 nh_f3(const float (&v)[3]) { return f3{v[0], v[1], v[2]}; }
OIGE_DEV float nh_sign(const float x) { return (x != x) ? x : (float)((x > 0.0f) - (x < 0.0f)); }  // torch.sign
OIGE_DEV f3 nh_keypoint(const q4 rot, const f3 pos, const f3 off) {  // tf_combine(rot, pos, identity, off)[1]
  const f3 r = quat_apply(rot, off);
  return f3{r.x + pos.x, r.y + pos.y, r.z + pos.z};
}

template <int D, int A, bool FULL>
__device__ __forceinline__ void nh_post_tile(const NhPost& a, float* __restrict__ tile, int64_t t, int nv) {
  constexpr int FO = 2 * D + 5;
  const oige_newhand_cabinet_params& p = a.p;
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;
  const bool term = p.terminate_on_limits != 0;
  const bool need_scaled = do_obs || (do_done && term);
  const bool need_points = do_obs || do_metrics;

  f3 dpos{0.f, 0.f, 0.f}, fpos[NK];
  q4 drot{1.f, 0.f, 0.f, 0.f}, hrot{1.f, 0.f, 0.f, 0.f}, frot[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) { fpos[k] = f3{0.f, 0.f, 0.f}; frot[k] = q4{1.f, 0.f, 0.f, 0.f}; }
  float cab3 = 0.f, cabv3 = 0.f;
  int64_t progress = 0, reset = 0, term_in = 0;
  if (live) {
    if (need_points) {
      const float* d3 = a.s.drawer_pos + env * 3;
      dpos = f3{d3[0], d3[1], d3[2]};
      const float4 dr = *reinterpret_cast<const float4*>(a.s.drawer_rot + env * 4);
      drot = q4{dr.x, dr.y, dr.z, dr.w};
#pragma unroll
      for (int k = 0; k < NK; ++k) {
        const float* f = a.s.finger_pos[k] + env * 3;
        fpos[k] = f3{f[0], f[1], f[2]};
        const float4 fr = *reinterpret_cast<const float4*>(a.s.finger_rot[k] + env * 4);
        frot[k] = q4{fr.x, fr.y, fr.z, fr.w};
      }
    }
    if (do_metrics) {
      const float4 hr = *reinterpret_cast<const float4*>(a.s.hand_rot + env * 4);
      hrot = q4{hr.x, hr.y, hr.z, hr.w};
      if (term) term_in = a.b.terminate_buf[env];
    }
    cab3 = a.s.cabinet_dof_pos[env * 4 + 3];
    if (do_obs) cabv3 = a.s.cabinet_dof_vel[env * 4 + 3];
    if (do_progress || do_done) progress = a.b.progress_buf[env];
    if (do_done) reset = a.b.reset_buf[env];
  }
  if (do_progress) progress += 1;  // rl_task.py:289

  f3 kd[NK], kf[NK];
#pragma unroll
  for (int k = 0; k < NK; ++k) {
    kd[k] = nh_keypoint(drot, dpos, nh_f3(p.drawer_local_keypoint_offsets[k]));
    kf[k] = nh_keypoint(frot[k], fpos[k], k == 0 ? nh_f3(p.local_thumb_pose_pos) : nh_f3(p.local_another_finger_pose_pos));
  }

  float* row = tile + lane * FO;
  float action_penalty = 0.0f;
  if (do_metrics) {  // torch.sum(actions ** 2, dim=-1): the env's A actions through the tile
    // (r03: fetching this chunk and the two below together with the scalars above -- one round trip instead of three -- takes
    // 185 VGPRs and is no faster: 93.0 against 92.1 us; ten resident one-wave workgroups per CU hide the chain)
    Chunk<A> c_act;
    load_chunk<A, FULL>(c_act, a.b.actions + env0 * A, nv * A);
    scatter_chunk<A, FO, 0, false>(c_act, tile, 1.0f);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < A; ++j) action_penalty += row[j] * row[j];
    __syncthreads();
  }
  bool terminate_new = false;
  if (need_scaled) {
    Chunk<D> c_pos;
    load_chunk<D, FULL>(c_pos, a.s.dof_pos + env0 * D, nv * D);
    scatter_chunk<D, FO, 0, false>(c_pos, tile, 1.0f);
    if (do_obs) {
      Chunk<D> c_vel;
      load_chunk<D, FULL>(c_vel, a.s.dof_vel + env0 * D, nv * D);
      scatter_chunk<D, FO, D, true>(c_vel, tile, p.dof_vel_scale);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float sc = (2.0f * (row[j] - p.dof_lower[j])) / (p.dof_upper[j] - p.dof_lower[j]) - 1.0f;
      row[j] = sc;
      terminate_new |= (sc < -1.05f) | (sc > 1.05f);
    }
    if (do_obs) {
      f3 tt{0.f, 0.f, 0.f};  // torch.sum(keypoints_drawer - keypoints_finger, dim=1)
#pragma unroll
      for (int k = 0; k < NK; ++k) { tt.x += kd[k].x - kf[k].x; tt.y += kd[k].y - kf[k].y; tt.z += kd[k].z - kf[k].z; }
      row[2 * D] = tt.x; row[2 * D + 1] = tt.y; row[2 * D + 2] = tt.z;
      row[2 * D + 3] = cab3; row[2 * D + 4] = cabv3;
      if (live) {
#pragma unroll
        for (int k = 0; k < NK; ++k) {
          if (a.b.keypoints_finger) { float* o = a.b.keypoints_finger + (env * NK + k) * 3; o[0] = kf[k].x; o[1] = kf[k].y; o[2] = kf[k].z; }
          if (a.b.keypoints_drawer) { float* o = a.b.keypoints_drawer + (env * NK + k) * 3; o[0] = kd[k].x; o[1] = kd[k].y; o[2] = kd[k].z; }
        }
      }
    }
  }
  if (do_metrics) {
    float dk[NK];
#pragma unroll
    for (int k = 0; k < NK; ++k) dk[k] = norm3(f3{kd[k].x - kf[k].x, kd[k].y - kf[k].y, kd[k].z - kf[k].z});
    const float d = ((dk[1] + dk[2]) + dk[3]) + dk[4];
    const float dist_reward = (-(d + dk[0])) / 2.0f;
    const f3 axis1 = quat_apply(hrot, nh_f3(p.left_hand_forward_axis)), axis2 = quat_apply(drot, nh_f3(p.drawer_inward_axis));
    const f3 axis3 = quat_apply(hrot, nh_f3(p.left_hand_up_axis)), axis4 = quat_apply(drot, nh_f3(p.drawer_up_axis));
    const float dot1 = dot3(axis1, axis2), dot2 = dot3(axis3, axis4);
    float rot_reward = 0.5f * (nh_sign(dot1) * (dot1 * dot1) + nh_sign(dot2) * (dot2 * dot2));
    if (p.rot_reward_thresholds) {
      rot_reward = (rot_reward < 0.6f) ? 0.0f : rot_reward;
      rot_reward = (rot_reward >= 0.8f) ? 2.0f * rot_reward : rot_reward;
    }
    bool inside = true;
#pragma unroll
    for (int k = 0; k < NK; ++k) {
      const float dx = kf[k].x - dpos.x;
      inside = inside && (dx > p.around_lower) && (dx < p.around_upper);
    }
    const float around = inside ? 0.5f : 0.0f;
    const float open_reward = cab3 * around + cab3;
    float rew = p.dist_reward_scale * dist_reward + p.rot_reward_scale * rot_reward;
    rew = rew + p.around_handle_reward_scale * around;
    rew = rew + p.open_reward_scale * open_reward;
    rew = rew - p.action_penalty_scale * action_penalty;
    rew = (cab3 > 0.01f) ? rew + 0.5f : rew;
    rew = (cab3 > 0.1f) ? rew + around : rew;
    rew = (cab3 > 0.19f) ? rew + (2.0f * around) : rew;
    if (term) rew = (term_in == 1) ? -100.0f : rew;
    if (live) a.b.rew_buf[env] = rew;
  }
  if (do_done) {  // sticky
    const bool now = (cab3 > 0.19f) | (progress >= (int64_t)p.max_episode_length - 1) | (term && terminate_new);
    const bool r = (reset != 0) | now;
    if (live) {
      a.b.reset_buf[env] = now ? 1 : reset;
      if (term) a.b.terminate_buf[env] = terminate_new ? 1 : 0;
    }
    const unsigned long long m = __ballot(live && r);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) store_tile<FO, FULL>(tile, a.b.obs_buf + env0 * FO, nv * FO);
}

template <int D, int A>
__global__ __launch_bounds__(WAVE) void newhand_cabinet_post_kernel(const NhPost a) {
  __shared__ __align__(16) float tile[TILE * (2 * D + 5)];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    nh_post_tile<D, A, true>(a, tile, t, TILE);
  else
    nh_post_tile<D, A, false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct NhPre {
  oige_newhand_cabinet_params p;
  oige_newhand_cabinet_buffers b;
  oige_franka_cabinet_reset_io r;
  const float* actions_in;
  int64_t n;
  uint32_t flags;
};

template <int D>
__device__ __forceinline__ void nh_draws(const NhPre& a, int64_t i, float (&u)[NMAX + 2]) {
#pragma unroll
  for (uint32_t blk = 0; blk < (D + 3) / 4; ++blk) {
    const u4 c{(uint32_t)i, blk, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
    const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
    u[blk * 4 + 0] = u01(x.x); u[blk * 4 + 1] = u01(x.y); u[blk * 4 + 2] = u01(x.z); u[blk * 4 + 3] = u01(x.w);
  }
}

// reset_idx for env i landing in compacted row k: view rows, target row, flags.  `stage` != nullptr (the pre kernel): the new
// joint positions go to that LDS row only; the caller writes dof_pos_out / dof_vel_out as lane-linear runs and its target walk
// takes the row from LDS (a lane walking its own D-float row with scalar stores, three times over, was the pre kernel's hot spot)
template <int D>
__device__ __forceinline__ void nh_reset_rows(const NhPre& a, int64_t i, int64_t k, const float (&u)[NMAX + 2],
                                              float* __restrict__ stage = nullptr) {
  const oige_newhand_cabinet_params& p = a.p;
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float pos = tensor_clamp(p.default_dof_pos[j] + 0.25f * (u[j] - 0.5f), p.dof_lower[j], p.dof_upper[j]);
    if (stage) {
      stage[j] = pos;
    } else {
      a.r.dof_pos_out[k * D + j] = pos;
      a.r.dof_vel_out[k * D + j] = 0.0f;
      a.b.dof_targets[i * D + j] = pos;
    }
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(a.r.cabinet_dof_pos_out + k * 4) = z4;
  *reinterpret_cast<float4*>(a.r.cabinet_dof_vel_out + k * 4) = z4;
  const int P = p.num_props;
  for (int j = 0; j < P; ++j) {
    const int64_t src = i * P + j, dst = k * P + j;
    a.r.prop_ids[dst] = src;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.prop_pos_out[dst * 3 + c] = a.r.default_prop_pos[src * 3 + c];
    *reinterpret_cast<float4*>(a.r.prop_rot_out + dst * 4) = *reinterpret_cast<const float4*>(a.r.default_prop_rot + src * 4);
  }
  a.b.reset_buf[i] = 0;
  a.b.progress_buf[i] = 0;
  a.r.env_ids[k] = i;
}

// new joint position j of reset env i on its own (the Philox block that holds draw j): for the reset rows past NH_RMAX
template <int D>
__device__ __forceinline__ float nh_reset_pos(const NhPre& a, int64_t i, int j, float dflt, float lo, float hi) {
  const u4 c{(uint32_t)i, (uint32_t)(j >> 2), (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
  const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
  const uint32_t w = (j & 3) == 0 ? x.x : (j & 3) == 1 ? x.y : (j & 3) == 2 ? x.z : x.w;
  return tensor_clamp(dflt + 0.25f * (u01(w) - 0.5f), lo, hi);
}

// reset envs of a group whose new joint rows are staged in LDS; the synthetic bench state resets 20-25 % of a group per
// step (51-64 +- 7), the reference one env in a thousand; a group with more (the first step after a full reset) handles the
// rest element by element
constexpr int NH_RMAX = 96;

// pre_physics_step for one group of 256 envs.  Order (r02): EVERY lane-linear load first (the group's actions, 256 x A
// floats, and targets, 256 x D floats, as float4 pieces), then the stores -- a load issued behind a store waits for that
// store's round trip, and the r01 kernel ran reset stores -> action loads -> action stores -> target loads, with one
// 4-byte gather of `actions_in` per target element on top (26 per thread).  The clamped actions are staged in LDS for the
// target walk (r01 tried that with the full 256-row reset staging beside it: 53 KB, three workgroups per CU; the reset
// staging is now capped at NH_RMAX rows: 39 KB, four).
template <int D, int A, int RANK>
__global__ __launch_bounds__(GROUP) void newhand_cabinet_pre_kernel(const NhPre a) {
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP], s_rank[GROUP];
  __shared__ __align__(16) float s_rows[NH_RMAX * D];  // new joint positions of the group's reset envs, compacted row order
  __shared__ __align__(16) float s_act[GROUP * A];      // the group's (clamped) actions
  __shared__ float s_lo[D], s_hi[D], s_rate[D], s_def[D];
  __shared__ int s_inv[D];  // joint -> action column, -1: not actuated
  const oige_newhand_cabinet_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const bool live = env < a.n;
  const int64_t envs_here = (a.n - g * GROUP) < GROUP ? (a.n - g * GROUP) : GROUP;
  const bool clampa = a.flags & OIGE_PRE_CLAMP_ACTIONS;
  const int atotal = (int)envs_here * A, ttotal = (int)envs_here * D;
  const int64_t abase = g * GROUP * A, tbase = g * GROUP * D;
  constexpr int AIT = (GROUP * A / 4 + GROUP - 1) / GROUP, TIT = (GROUP * D / 4 + GROUP - 1) / GROUP;

  // ---- 0: loads (the flag through a clamped index: written as `live && flag[env]` it was a branch around a load with a full
  // wait behind it, a serial round trip in front of everything else; r03)
  const int64_t flag_ld = a.b.reset_buf[live ? env : g * GROUP];
  const U8Prefix rp = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  float av[AIT][4], tv[TIT][4];
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
#pragma unroll
    for (int c = 0; c < 4; ++c) av[it][c] = 0.f;
    if (e + 3 < atotal) {
      const float4 v = *reinterpret_cast<const float4*>(a.actions_in + abase + e);
      av[it][0] = v.x; av[it][1] = v.y; av[it][2] = v.z; av[it][3] = v.w;
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < atotal) av[it][c] = a.actions_in[abase + e + c];
    }
  }
#pragma unroll
  for (int it = 0; it < TIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
#pragma unroll
    for (int c = 0; c < 4; ++c) tv[it][c] = 0.f;
    if (e + 3 < ttotal) {
      const float4 t4 = LD4S(a.b.dof_targets + tbase + e);
      tv[it][0] = t4.x; tv[it][1] = t4.y; tv[it][2] = t4.z; tv[it][3] = t4.w;
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < ttotal) tv[it][c] = a.b.dof_targets[tbase + e + c];
    }
  }

  // ---- 1: ranks, lists, per-joint tables
  const bool flagged = live && flag_ld != 0;
  int k0, cnt;  // offset of the group in the id list, flagged envs in the group
  const int64_t k = rank_finish<RANK>(rp, flagged, a.b.reset_workspace, a.b.tile_counts_u8, G, g, lds8, k0, cnt);
  if (flagged) s_list[(int)(k - k0)] = tid;
  s_rank[tid] = flagged ? (int)(k - k0) : -1;
#pragma unroll
  for (int j = 0; j < D; ++j)
    if (tid == j) {
      s_lo[j] = p.dof_lower[j];
      s_hi[j] = p.dof_upper[j];
      s_def[j] = p.default_dof_pos[j];
      s_rate[j] = p.dof_speed_scales[j] * p.dt;  // left to right: speed_scales * dt * actions * action_scale
    }
  if (tid >= 64 && tid < 64 + D) {
    int inv = -1;
#pragma unroll
    for (int j = 0; j < A; ++j)
      if (p.actuated_dof_indices[j] == tid - 64) inv = j;
    s_inv[tid - 64] = inv;
  }
  __syncthreads();

  // ---- 2: resets on dense lanes: thread j takes the group's j-th flagged env; its new joint row goes to LDS row j (past
  // NH_RMAX: straight to memory)
  wait_loads_done();
  // the kernel's second and last round trip of loads, in front of every store: prop row e of the group's flagged envs = prop
  // e % P of the (e / P)-th flagged env, thread e takes it (rows beyond the workgroup's width: the loop at the end).  Not
  // conditional: an idle thread re-reads a valid row, and without props the actions stand in for the tables.  (r02 fetched
  // these rows at the end of the kernel, behind all of its stores: a third round trip, and a fourth for the rows' own stores.)
  const int P = p.num_props;
  const int prow = cnt * P;
  const bool has_prop = tid < prow;
  const int pr = has_prop ? tid / (P > 0 ? P : 1) : 0;
  const int64_t psrc = P > 0 ? (g * GROUP + (cnt > 0 ? s_list[pr] : 0)) * P + (has_prop ? tid - pr * P : 0) : 0;
  const float* pp_tab = P > 0 ? a.r.default_prop_pos : a.actions_in;
  const float* pq_tab = P > 0 ? a.r.default_prop_rot : a.actions_in;
  float pp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pp[c] = pp_tab[psrc * 3 + c];
  const float4 pq = *reinterpret_cast<const float4*>(pq_tab + psrc * 4);
  if (tid < cnt) {
    const int64_t i = g * GROUP + s_list[tid], kk = (int64_t)k0 + tid;
    float u[NMAX + 2];
    nh_draws<D>(a, i, u);
#pragma unroll
    for (int j = 0; j < D; ++j) {
      const float pos = tensor_clamp(p.default_dof_pos[j] + 0.25f * (u[j] - 0.5f), p.dof_lower[j], p.dof_upper[j]);
      if (tid < NH_RMAX) {
        s_rows[tid * D + j] = pos;
      } else {
        a.r.dof_pos_out[kk * D + j] = pos;
        a.r.dof_vel_out[kk * D + j] = 0.0f;
      }
    }
  }
  // the prop rows leave in front of every other store: here the wait for their loads is a wait for loads alone (behind stores
  // it would also be the wait for those stores' acknowledgement: one in-order counter), and it overlaps the dense lanes' draws
  if (has_prop) {
    const int64_t dst = (int64_t)k0 * P + tid;
    a.r.prop_ids[dst] = psrc;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.prop_pos_out[dst * 3 + c] = pp[c];
    *reinterpret_cast<float4*>(a.r.prop_rot_out + dst * 4) = pq;
  }
  if (tid < cnt) {
    const int64_t i = g * GROUP + s_list[tid], kk = (int64_t)k0 + tid;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a.r.cabinet_dof_pos_out + kk * 4) = z4;
    *reinterpret_cast<float4*>(a.r.cabinet_dof_vel_out + kk * 4) = z4;
    a.b.reset_buf[i] = 0;
    a.b.progress_buf[i] = 0;
    a.r.env_ids[kk] = i;
  }

  // ---- 3: actions.clone() (+ the wrapper's clamp), and the LDS copy the target walk reads
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
    if (e >= atotal) continue;
    if (clampa) {
#pragma unroll
      for (int c = 0; c < 4; ++c) av[it][c] = clamp_sym(av[it][c], p.clip_actions);
    }
    if (e + 3 < atotal) {
      ST4S(a.b.actions + abase + e, make_float4(av[it][0], av[it][1], av[it][2], av[it][3]));
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < atotal) a.b.actions[abase + e + c] = av[it][c];
    }
    *reinterpret_cast<float4*>(s_act + e) = make_float4(av[it][0], av[it][1], av[it][2], av[it][3]);
  }
  __syncthreads();  // the reset envs' new rows and the group's actions are visible to the whole workgroup

  // ---- 4: targets: joint j integrates action column s_inv[j] (none: unchanged)
  const bool want_act = a.b.actuated_targets_out != nullptr;  // dof_targets[:, actuated_dof_indices] as a second output
#pragma unroll
  for (int it = 0; it < TIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
    if (e >= ttotal) continue;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ee = e + c;
      if (ee >= ttotal) continue;
      const int il = ee / D, jj = ee - il * D;
      const int rk = s_rank[il];
      if (rk >= 0)  // dof_targets[env_ids] = pos (reset_idx), then this step's integration
        tv[it][c] = rk < NH_RMAX ? s_rows[rk * D + jj] : nh_reset_pos<D>(a, g * GROUP + il, jj, s_def[jj], s_lo[jj], s_hi[jj]);
      const int ja = s_inv[jj];
      if (ja < 0) continue;
      tv[it][c] = tensor_clamp(tv[it][c] + (s_rate[jj] * s_act[il * A + ja]) * p.action_scale, s_lo[jj], s_hi[jj]);
      if (want_act) s_act[il * A + ja] = tv[it][c];  // the action slot becomes the target slot (this thread alone touches it)
    }
    if (e + 3 < ttotal) {
      ST4S(a.b.dof_targets + tbase + e, make_float4(tv[it][0], tv[it][1], tv[it][2], tv[it][3]));
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < ttotal) a.b.dof_targets[tbase + e + c] = tv[it][c];
    }
  }

  if (want_act) {  // ---- 4b: the (N,A) block of actuated targets, lane-linear out of the LDS tile
    __syncthreads();
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int e = 4 * (it * GROUP + tid);
      if (e >= atotal) continue;
      if (e + 3 < atotal) {
        ST4S(a.b.actuated_targets_out + abase + e, *reinterpret_cast<const float4*>(s_act + e));
      } else {
        for (int c = 0; c < 4; ++c)
          if (e + c < atotal) a.b.actuated_targets_out[abase + e + c] = s_act[e + c];
      }
    }
  }

  // ---- 5: dof_pos_out / dof_vel_out rows k0 .. k0 + min(cnt, NH_RMAX): one contiguous run each, written lane-linearly
  {
    const int run = (cnt < NH_RMAX ? cnt : NH_RMAX) * D;
    float* dp = a.r.dof_pos_out + (int64_t)k0 * D;
    float* dv = a.r.dof_vel_out + (int64_t)k0 * D;
    for (int e = tid; e < run; e += GROUP) {
      dp[e] = s_rows[e];
      dv[e] = 0.0f;
    }
  }
  if (g == G - 1 && tid == 0) {
    *a.r.count = (int64_t)k0 + cnt;
    *a.r.prop_count = ((int64_t)k0 + cnt) * p.num_props;
  }
  // ---- 6: prop rows beyond the workgroup's width (more than 256 / num_props flagged envs in the group): loads behind stores, rare
  for (int w = tid + GROUP; w < prow; w += GROUP) {
    const int r = w / P, j = w - r * P;
    const int64_t src = (g * GROUP + s_list[r]) * P + j, dst = ((int64_t)k0 + r) * P + j;
    const float px = a.r.default_prop_pos[src * 3], py = a.r.default_prop_pos[src * 3 + 1], pz = a.r.default_prop_pos[src * 3 + 2];
    const float4 prq = *reinterpret_cast<const float4*>(a.r.default_prop_rot + src * 4);
    a.r.prop_ids[dst] = src;
    a.r.prop_pos_out[dst * 3] = px; a.r.prop_pos_out[dst * 3 + 1] = py; a.r.prop_pos_out[dst * 3 + 2] = pz;
    *reinterpret_cast<float4*>(a.r.prop_rot_out + dst * 4) = prq;
  }
}

template <int D>
__global__ __launch_bounds__(256) void newhand_cabinet_reset_idx_kernel(const NhPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) *a.r.prop_count = *a.r.count * a.p.num_props;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[NMAX + 2];
  if (a.r.u_dof) {
#pragma unroll
    for (int j = 0; j < D; ++j) u[j] = a.r.u_dof[k * D + j];
  } else {
    nh_draws<D>(a, i, u);
  }
  nh_reset_rows<D>(a, i, k, u);
}

static int check_newhand(const oige_newhand_cabinet_params* p, const oige_newhand_cabinet_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  OIGE_REQUIRE(p->num_props >= 0 && p->num_props <= 64, "%s: num_props %d out of range [0, 64]", who, p->num_props);
  const bool ok = (p->num_dofs == 26 && (p->num_actions == 26 || p->num_actions == 22)) || (p->num_dofs == 25 && p->num_actions == 25);
  OIGE_REQUIRE(ok, "%s: (num_dofs, num_actions) = (%d, %d) not supported: (26, 26), (26, 22) or (25, 25)", who, p->num_dofs, p->num_actions);
  for (int j = 0; j < p->num_actions; ++j)
    OIGE_REQUIRE(p->actuated_dof_indices[j] >= 0 && p->actuated_dof_indices[j] < p->num_dofs, "%s: actuated_dof_indices[%d] = %d out of range",
                 who, j, p->actuated_dof_indices[j]);
  OIGE_REQUIRE(!p->terminate_on_limits || b->terminate_buf, "%s: terminate_on_limits needs terminate_buf", who);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_newhand_cabinet_post_physics(const oige_newhand_cabinet_params* params, const oige_newhand_cabinet_state* state,
                                      const oige_newhand_cabinet_buffers* buffers, int64_t num_envs, uint32_t stages,
                                      oige_stream_t stream) {
  if (int rc = check_newhand(params, buffers, num_envs, "newhand_cabinet_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "newhand_cabinet_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "newhand_cabinet_post_physics: bad stages mask 0x%x", stages);
  const oige_newhand_cabinet_state& s = *state;
  const oige_newhand_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(s.hand_rot && s.drawer_pos && s.drawer_rot && s.dof_pos && s.dof_vel && s.cabinet_dof_pos && s.cabinet_dof_vel,
               "newhand_cabinet_post_physics: null state pointer");
  for (int k = 0; k < NK; ++k)
    OIGE_REQUIRE(s.finger_pos[k] && s.finger_rot[k] && aligned16(s.finger_rot[k]), "newhand_cabinet_post_physics: finger %d: null/unaligned pose", k);
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.actions, "newhand_cabinet_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.hand_rot) && aligned16(s.drawer_rot) && aligned16(s.dof_pos) && aligned16(s.dof_vel) && aligned16(b.actions) &&
                   aligned16(b.obs_buf),
               "newhand_cabinet_post_physics: row buffers must be 16-byte aligned");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  NhPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  hipStream_t st = (hipStream_t)stream;
  if (params->num_dofs == 25)
    hipLaunchKernelGGL((newhand_cabinet_post_kernel<25, 25>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
  else if (params->num_actions == 22)
    hipLaunchKernelGGL((newhand_cabinet_post_kernel<26, 22>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
  else
    hipLaunchKernelGGL((newhand_cabinet_post_kernel<26, 26>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
  return check_launch("newhand_cabinet_post_kernel");
}

static int fill_newhand_pre(NhPre& a, const oige_newhand_cabinet_params* params, const oige_newhand_cabinet_buffers* buffers,
                            const oige_franka_cabinet_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_newhand(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  const oige_newhand_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.dof_targets, "%s: null buffer pointer", who);
  OIGE_REQUIRE(r->env_ids && r->count && r->dof_pos_out && r->dof_vel_out && r->cabinet_dof_pos_out && r->cabinet_dof_vel_out && r->prop_count,
               "%s: null reset pointer", who);
  OIGE_REQUIRE(params->num_props == 0 || (r->default_prop_pos && r->default_prop_rot && r->prop_ids && r->prop_pos_out && r->prop_rot_out),
               "%s: num_props > 0 needs default_prop_pos/rot, prop_ids, prop_pos_out, prop_rot_out", who);
  OIGE_REQUIRE(aligned16(r->cabinet_dof_pos_out) && aligned16(r->cabinet_dof_vel_out) && aligned16(r->default_prop_rot) &&
                   aligned16(r->prop_rot_out),
               "%s: row buffers must be 16-byte aligned", who);
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  return OIGE_OK;
}

int oige_newhand_cabinet_pre_physics(const oige_newhand_cabinet_params* params, const float* actions_in,
                                     const oige_newhand_cabinet_buffers* buffers,
                                     const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, uint32_t flags,
                                     oige_stream_t stream) {
  NhPre a;
  if (int rc = fill_newhand_pre(a, params, buffers, reset_io, num_envs, "newhand_cabinet_pre_physics")) return rc;
  const oige_newhand_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(actions_in && aligned16(actions_in) && b.actions && aligned16(b.actions), "newhand_cabinet_pre_physics: null/unaligned actions");
  OIGE_REQUIRE(b.reset_workspace && aligned16(b.reset_workspace), "newhand_cabinet_pre_physics: null/unaligned workspace");
  OIGE_REQUIRE(!reset_io->u_dof, "newhand_cabinet_pre_physics: pre-drawn noise goes through oige_newhand_cabinet_reset_idx");
  a.actions_in = actions_in;
  a.flags = flags;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(a.b.reset_buf, num_envs, a.b.reset_workspace, s, a.b.tile_counts_u8)) return rc;
  const int rank = pick_rank(G, a.b.tile_counts_u8);
  if (rank == RANK_SCANNED)
    if (int rc = launch_scan_groups(num_envs, a.b.reset_workspace, s)) return rc;
#define OIGE_NH_PRE(D_, A_)                                                                                                  \
  do {                                                                                                                       \
    if (rank == RANK_U8) hipLaunchKernelGGL((newhand_cabinet_pre_kernel<D_, A_, RANK_U8>), dim3((unsigned)G), dim3(GROUP), 0, s, a); \
    else if (rank == RANK_INT32) hipLaunchKernelGGL((newhand_cabinet_pre_kernel<D_, A_, RANK_INT32>), dim3((unsigned)G), dim3(GROUP), 0, s, a); \
    else hipLaunchKernelGGL((newhand_cabinet_pre_kernel<D_, A_, RANK_SCANNED>), dim3((unsigned)G), dim3(GROUP), 0, s, a);     \
  } while (0)
  if (params->num_dofs == 25) OIGE_NH_PRE(25, 25);
  else if (params->num_actions == 22) OIGE_NH_PRE(26, 22);
  else OIGE_NH_PRE(26, 26);
#undef OIGE_NH_PRE
  return check_launch("newhand_cabinet_pre_kernel");
}

int oige_newhand_cabinet_reset_idx(const oige_newhand_cabinet_params* params, const oige_newhand_cabinet_buffers* buffers,
                                   const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                   oige_stream_t stream) {
  NhPre a;
  if (int rc = fill_newhand_pre(a, params, buffers, reset_io, num_envs, "newhand_cabinet_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "newhand_cabinet_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  const dim3 grid((unsigned)((max_ids + 255) / 256));
  if (params->num_dofs == 25)
    hipLaunchKernelGGL(newhand_cabinet_reset_idx_kernel<25>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  else
    hipLaunchKernelGGL(newhand_cabinet_reset_idx_kernel<26>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("newhand_cabinet_reset_idx_kernel");
}

}  // extern "C"
