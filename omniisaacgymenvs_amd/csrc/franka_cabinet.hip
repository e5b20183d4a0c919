// FrankaCabinet / TomoDJCabinet / TomoSPCCabinet per-step task math for gfx950 (MI355X) -- SURVEY.md 8f rank 4.
//
// Replaces /root/reference/omniisaacgymenvs/tasks/franka_cabinet.py behind the unchanged FrankaCabinetTask method contract:
//   get_observations :283-358 (compute_grasp_transforms :475-494), calculate_metrics :439-466 (compute_franka_reward
//   :496-601), is_done :468-473, pre_physics_step :360-373, reset_idx :375-414.
// tasks/tomodj_cabinet.py:373-712 (9 DOFs) and tasks/tomospc_cabinet.py (8 DOFs) share the skeleton; their reward variant
// (OIGE_CABINET_REWARD_TOMO) and thresholds come in through the params.  Kernels are instantiated for D = 9 and D = 8.
//
// post_physics: one wavefront per tile of 64 envs, HBM-bound, 388 algorithmic bytes per env-step
//   (R hand 28 + drawer 28 + dof_pos 36 + dof_vel 36 + cabinet pos/vel rows 32 + finger z 2 x 4 + actions 36 + progress 8 +
//    reset 8 = 220; W obs 92 + rew 4 + reset 8 + progress 8 + grasp transforms 56 = 168.  The finger rows' x, y come along
//    with z (12-byte rows), and with the reference's binding both finger pointers are the same buffer).
//   The three (N,9) inputs arrive as lane-linear float4 runs and are transposed through the (64 x 23) observation tile in LDS
//   (odd row stride: bank-conflict free), which then streams out as one contiguous run.
// pre_physics: one workgroup per 256 envs; flagged envs re-draw their joint positions on dense lanes (rows for the view
//   setters, new positions parked in LDS), then the group's 256 x 9 targets / actions are walked lane-linear.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_tile.hpp"

namespace oige {

constexpr int FMAX = OIGE_FRANKA_DOFS;  // 9: capacity of the per-joint tables in the params
constexpr int CD = OIGE_CABINET_DOFS;   // 4

struct FrankaPost {
  oige_franka_cabinet_params p;
  oige_franka_cabinet_state s;
  oige_franka_cabinet_buffers b;
  int64_t n;
  uint32_t stages;
};

OIGE_DEV f3 f3_of(const float (&v)[3]) { return f3{v[0], v[1], v[2]}; }
OIGE_DEV q4 q4_of(const float (&v)[4]) { return q4{v[0], v[1], v[2], v[3]}; }
OIGE_DEV float sign_f(const float x) { return (x != x) ? x : (float)((x > 0.0f) - (x < 0.0f)); }  // torch.sign

template <int FD, bool FULL>
__device__ __forceinline__ void franka_post_tile(const FrankaPost& a, float* __restrict__ tile, int64_t t, int nv) {
  constexpr int FO = 2 * FD + 5;  // obs width: 23 (9 DOFs) / 21 (8 DOFs), odd -> bank-conflict free rows
  const oige_franka_cabinet_params& p = a.p;
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS, do_done = a.stages & OIGE_STAGE_DONE;

  Chunk<FD> c_pos, c_vel, c_act;
  if (do_obs || do_metrics) load_chunk<FD, FULL>(c_pos, a.s.franka_dof_pos + env0 * FD, nv * FD);
  if (do_obs) load_chunk<FD, FULL>(c_vel, a.s.franka_dof_vel + env0 * FD, nv * FD);
  if (do_metrics) load_chunk<FD, FULL>(c_act, a.b.actions + env0 * FD, nv * FD);

  f3 hpos{0.f, 0.f, 0.f}, dpos{0.f, 0.f, 0.f};
  q4 hrot{1.f, 0.f, 0.f, 0.f}, drot{1.f, 0.f, 0.f, 0.f};
  float cab3 = 0.f, cabv3 = 0.f, lfz = 0.f, rfz = 0.f;
  int64_t progress = 0, reset = 0;
  if (live) {
    if (do_obs || do_metrics) {
      const float* h3 = a.s.hand_pos + env * 3;
      hpos = f3{h3[0], h3[1], h3[2]};
      const float* d3 = a.s.drawer_pos + env * 3;
      dpos = f3{d3[0], d3[1], d3[2]};
      const float4 hr = *reinterpret_cast<const float4*>(a.s.hand_rot + env * 4);
      hrot = q4{hr.x, hr.y, hr.z, hr.w};
      const float4 dr = *reinterpret_cast<const float4*>(a.s.drawer_rot + env * 4);
      drot = q4{dr.x, dr.y, dr.z, dr.w};
    }
    cab3 = a.s.cabinet_dof_pos[env * CD + 3];
    if (do_obs) cabv3 = a.s.cabinet_dof_vel[env * CD + 3];
    if (do_metrics) {
      lfz = a.s.lfinger_pos[env * 3 + 2];
      rfz = a.s.rfinger_pos[env * 3 + 2];
    }
    if (do_progress || do_done) progress = a.b.progress_buf[env];
    if (do_done) reset = a.b.reset_buf[env];
  }
  if (do_progress) progress += 1;  // rl_task.py:289

  // compute_grasp_transforms (:475-494): tf_combine = (quat_mul(q1, q2), quat_apply(q1, t2) + t1)
  const q4 fg_rot = quat_mul(hrot, q4_of(p.franka_local_grasp_rot));
  const f3 fa = quat_apply(hrot, f3_of(p.franka_local_grasp_pos));
  const f3 fg_pos{fa.x + hpos.x, fa.y + hpos.y, fa.z + hpos.z};
  const q4 dg_rot = quat_mul(drot, q4_of(p.drawer_local_grasp_rot));
  const f3 da = quat_apply(drot, f3_of(p.drawer_local_grasp_pos));
  const f3 dg_pos{da.x + dpos.x, da.y + dpos.y, da.z + dpos.z};

  float action_penalty = 0.0f;
  float* row = tile + lane * FO;
  if (do_metrics) {  // torch.sum(actions ** 2, dim=-1) (:575): the env's 9 actions through the tile
    scatter_chunk<FD, FO, 0, false>(c_act, tile, 1.0f);
    __syncthreads();
#pragma unroll
    for (int j = 0; j < FD; ++j) action_penalty += row[j] * row[j];
    __syncthreads();
  }
  float jp7 = 0.f, jp8 = 0.f;
  if (do_obs || do_metrics) {
    scatter_chunk<FD, FO, 0, false>(c_pos, tile, 1.0f);
    if (do_obs) scatter_chunk<FD, FO, FD, true>(c_vel, tile, p.dof_vel_scale);  // :349
    __syncthreads();
    jp7 = row[FD - 2];  // the two finger joints: 7, 8 (franka_cabinet.py:571, tomodj) / 6, 7 (tomospc_cabinet.py)
    jp8 = row[FD - 1];
    if (do_obs) {
#pragma unroll
      for (int j = 0; j < FD; ++j)  // :339-344
        row[j] = (2.0f * (row[j] - p.dof_lower[j])) / (p.dof_upper[j] - p.dof_lower[j]) - 1.0f;
      row[2 * FD] = dg_pos.x - fg_pos.x; row[2 * FD + 1] = dg_pos.y - fg_pos.y; row[2 * FD + 2] = dg_pos.z - fg_pos.z;  // :345
      row[2 * FD + 3] = cab3; row[2 * FD + 4] = cabv3;
      if (live) {
        if (a.b.franka_grasp_rot) *reinterpret_cast<float4*>(a.b.franka_grasp_rot + env * 4) = make_float4(fg_rot.w, fg_rot.x, fg_rot.y, fg_rot.z);
        if (a.b.drawer_grasp_rot) *reinterpret_cast<float4*>(a.b.drawer_grasp_rot + env * 4) = make_float4(dg_rot.w, dg_rot.x, dg_rot.y, dg_rot.z);
        if (a.b.franka_grasp_pos) { float* o = a.b.franka_grasp_pos + env * 3; o[0] = fg_pos.x; o[1] = fg_pos.y; o[2] = fg_pos.z; }
        if (a.b.drawer_grasp_pos) { float* o = a.b.drawer_grasp_pos + env * 3; o[0] = dg_pos.x; o[1] = dg_pos.y; o[2] = dg_pos.z; }
      }
    }
  }
  if (do_metrics) {  // compute_franka_reward (:526-601)
    const float d = norm3(f3{fg_pos.x - dg_pos.x, fg_pos.y - dg_pos.y, fg_pos.z - dg_pos.z});
    const bool tomo = p.reward_variant == OIGE_CABINET_REWARD_TOMO;
    float dist_reward;
    if (tomo) {
      dist_reward = -d;  // tomodj_cabinet.py:607
    } else {
      dist_reward = 1.0f / (1.0f + d * d);
      dist_reward *= dist_reward;
      dist_reward = (d <= 0.02f) ? dist_reward * 2.0f : dist_reward;
    }
    const f3 axis1 = quat_apply(fg_rot, f3_of(p.gripper_forward_axis)), axis2 = quat_apply(dg_rot, f3_of(p.drawer_inward_axis));
    const f3 axis3 = quat_apply(fg_rot, f3_of(p.gripper_up_axis)), axis4 = quat_apply(dg_rot, f3_of(p.drawer_up_axis));
    const float dot1 = dot3(axis1, axis2), dot2 = dot3(axis3, axis4);
    float rot_reward = 0.5f * (sign_f(dot1) * (dot1 * dot1) + sign_f(dot2) * (dot2 * dot2));
    if (tomo) {  // tomodj_cabinet.py:622-623
      rot_reward = (rot_reward < 0.6f) ? 0.0f : rot_reward;
      rot_reward = (rot_reward >= 0.8f) ? 2.0f * rot_reward : rot_reward;
    }
    const bool around_b = tomo ? (lfz <= dg_pos.z) && (rfz >= dg_pos.z) : (lfz > dg_pos.z) && (rfz < dg_pos.z);
    const float around = around_b ? 0.5f : 0.0f;
    const float finger_dist_reward = around_b ? (0.04f - __builtin_fabsf(lfz - dg_pos.z)) + (0.04f - __builtin_fabsf(rfz - dg_pos.z)) : 0.0f;
    const float finger_close_reward = (d <= 0.03f) ? (0.04f - jp7) + (0.04f - jp8) : 0.0f;
    const float open_reward = cab3 * around + cab3;
    float rew = p.dist_reward_scale * dist_reward + p.rot_reward_scale * rot_reward;
    rew = rew + p.around_handle_reward_scale * around;
    rew = rew + p.open_reward_scale * open_reward;
    rew = rew + p.finger_dist_reward_scale * finger_dist_reward;
    rew = rew - p.action_penalty_scale * action_penalty;
    rew = rew + finger_close_reward * p.finger_close_reward_scale;
    rew = (cab3 > 0.01f) ? rew + 0.5f : rew;
    rew = (cab3 > p.open_bonus_threshold) ? rew + around : rew;
    rew = (cab3 > p.open_done_threshold) ? rew + (2.0f * around) : rew;
    if (live) a.b.rew_buf[env] = rew;
  }
  if (do_done) {  // :468-473, sticky
    const bool now = (cab3 > p.open_done_threshold) | (progress >= (int64_t)p.max_episode_length - 1);
    const bool r = (reset != 0) | now;
    const int64_t out = now ? 1 : reset;
    if (live) a.b.reset_buf[env] = out;
    const unsigned long long m = __ballot(live && r);
    if (a.b.reset_workspace && lane == 0) a.b.reset_workspace[t] = __popcll(m);
    if (a.b.tile_counts_u8 && lane == 0) a.b.tile_counts_u8[t] = (uint8_t)__popcll(m);
  }
  if (do_progress && live) a.b.progress_buf[env] = progress;
  __syncthreads();
  if (do_obs) store_tile<FO, FULL>(tile, a.b.obs_buf + env0 * FO, nv * FO);
}

template <int FD>
__global__ __launch_bounds__(WAVE) void franka_cabinet_post_kernel(const FrankaPost a) {
  __shared__ __align__(16) float tile[TILE * (2 * FD + 5)];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    franka_post_tile<FD, true>(a, tile, t, TILE);
  else
    franka_post_tile<FD, false>(a, tile, t, (int)left);
}

// ---------------------------------------------------------------- pre_physics / reset
struct FrankaPre {
  oige_franka_cabinet_params p;
  oige_franka_cabinet_buffers b;
  oige_franka_cabinet_reset_io r;
  const float* actions_in;
  int64_t n;
  uint32_t flags;
};

// the nine U[0,1) draws of env i (in-kernel path): Philox blocks 0..2 keyed (seed; env, block, step)
__device__ __forceinline__ void franka_draws(const FrankaPre& a, int64_t i, float (&u)[FMAX]) {
#pragma unroll
  for (uint32_t blk = 0; blk < 3; ++blk) {
    const u4 c{(uint32_t)i, blk, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
    const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
    u[blk * 4 + 0] = u01(x.x);
    if (blk < 2) { u[blk * 4 + 1] = u01(x.y); u[blk * 4 + 2] = u01(x.z); u[blk * 4 + 3] = u01(x.w); }
  }
}

// reset_idx (:375-414) for env i landing in compacted row k; pos[] = its new joint positions
// DOF_ROWS = false (the pre kernel): the caller writes the dof_pos_out / dof_vel_out rows of its group as one lane-linear run
template <int FD, bool DOF_ROWS = true>
__device__ __forceinline__ void franka_reset_rows(const FrankaPre& a, int64_t i, int64_t k, const float (&u)[FMAX], float (&pos)[FMAX]) {
  const oige_franka_cabinet_params& p = a.p;
#pragma unroll
  for (int j = 0; j < FD; ++j) {
    pos[j] = tensor_clamp(p.default_dof_pos[j] + 0.25f * (u[j] - 0.5f), p.dof_lower[j], p.dof_upper[j]);  // :380-385
    if (DOF_ROWS) {
      a.r.dof_pos_out[k * FD + j] = pos[j];
      a.r.dof_vel_out[k * FD + j] = 0.0f;
    }
  }
  const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
  *reinterpret_cast<float4*>(a.r.cabinet_dof_pos_out + k * CD) = z4;  // :393-398
  *reinterpret_cast<float4*>(a.r.cabinet_dof_vel_out + k * CD) = z4;
  const int P = p.num_props;
  for (int j = 0; j < P; ++j) {  // :401-406
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

// Two round trips of loads, both in front of the first store (r03; quadcopter.hip has the reasoning).  Trip 1: the flag
// (clamped index), the byte-table prefix, the walk's action and target quads.  Trip 2, behind the ranks: the default rows of
// the props of the group's flagged envs, one prop row per thread over the whole workgroup (r02 copied them prop by prop on
// the dense lane, a load behind a store each time: num_props serial round trips in every workgroup with a flagged env).
// The dense lanes' Philox draws stay in front of the walk: it starts the flagged envs' rows from their new positions.
template <int FD, int RANK>
__global__ __launch_bounds__(GROUP) void franka_cabinet_pre_kernel(const FrankaPre a) {
  __shared__ int lds8[2 * GROUP_TILES];
  __shared__ int s_list[GROUP], s_rank[GROUP];
  __shared__ float s_pos[GROUP * FD];
  __shared__ float s_lo[FD], s_hi[FD], s_rate[FD];
  const oige_franka_cabinet_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const bool live = env < a.n;
  // ---- trip 1
  const int64_t flag_ld = a.b.reset_buf[live ? env : g * GROUP];  // :364
  const U8Prefix rp = rank_issue<RANK>(a.b.tile_counts_u8, g, tid);
  const int64_t envs_here = (a.n - g * GROUP) < GROUP ? (a.n - g * GROUP) : GROUP;
  const int total = (int)envs_here * FD;
  const int64_t base = g * GROUP * FD;
  constexpr int WIT = (GROUP * FD / 4 + GROUP - 1) / GROUP;
  float avq[WIT][4], tvq[WIT][4];
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
#pragma unroll
    for (int c = 0; c < 4; ++c) avq[it][c] = tvq[it][c] = 0.f;
    if (e + 3 < total) {
      const float4 a4 = *reinterpret_cast<const float4*>(a.actions_in + base + e);
      const float4 t4 = *reinterpret_cast<const float4*>(a.b.franka_dof_targets + base + e);
      avq[it][0] = a4.x; avq[it][1] = a4.y; avq[it][2] = a4.z; avq[it][3] = a4.w;
      tvq[it][0] = t4.x; tvq[it][1] = t4.y; tvq[it][2] = t4.z; tvq[it][3] = t4.w;
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < total) { avq[it][c] = a.actions_in[base + e + c]; tvq[it][c] = a.b.franka_dof_targets[base + e + c]; }
    }
  }
  const bool flagged = live && flag_ld != 0;
  int k0, cnt;  // offset of the group in the id list, flagged envs in the group
  const int64_t k = rank_finish<RANK>(rp, flagged, a.b.reset_workspace, a.b.tile_counts_u8, G, g, lds8, k0, cnt);
  if (flagged) s_list[(int)(k - k0)] = tid;
  s_rank[tid] = flagged ? (int)(k - k0) : -1;
#pragma unroll
  for (int j = 0; j < FD; ++j)
    if (tid == j) {
      s_lo[j] = p.dof_lower[j];
      s_hi[j] = p.dof_upper[j];
      s_rate[j] = p.dof_speed_scales[j] * p.dt;  // left to right: speed_scales * dt * actions * action_scale (:369)
    }
  __syncthreads();

  // ---- trip 2: prop row e of the group's flagged envs (:401-406) = prop e % P of the (e / P)-th flagged env; thread e takes
  // it (rows beyond the workgroup's width: the loop at the end).  Not conditional: an idle thread re-reads a valid row, and
  // without props the actions stand in for the tables
  const int P = p.num_props;
  const int prow = cnt * P;  // prop rows of this group
  const bool has_prop = tid < prow;
  const int pr = has_prop ? tid / (P > 0 ? P : 1) : 0;
  const int64_t psrc = P > 0 ? (g * GROUP + (cnt > 0 ? s_list[pr] : 0)) * P + (has_prop ? tid - pr * P : 0) : 0;
  const float* pp_tab = P > 0 ? a.r.default_prop_pos : a.actions_in;
  const float* pq_tab = P > 0 ? a.r.default_prop_rot : a.actions_in;
  float pp[3];
#pragma unroll
  for (int c = 0; c < 3; ++c) pp[c] = pp_tab[psrc * 3 + c];
  const float4 pq = *reinterpret_cast<const float4*>(pq_tab + psrc * 4);

  const bool dense = tid < cnt;  // dense lanes: thread j takes the group's j-th flagged env
  const int dl = dense ? s_list[tid] : 0;
  const int64_t di = g * GROUP + dl;
  if (dense) {
    float u[FMAX], pos[FMAX];
    franka_draws(a, di, u);
#pragma unroll
    for (int j = 0; j < FD; ++j) {
      pos[j] = tensor_clamp(p.default_dof_pos[j] + 0.25f * (u[j] - 0.5f), p.dof_lower[j], p.dof_upper[j]);  // :380-385
      s_pos[dl * FD + j] = pos[j];  // franka_dof_targets[env_ids] = pos (:389)
    }
  }
  __syncthreads();

  // ---- stores
  if (has_prop) {
    const int64_t dst = (int64_t)k0 * P + tid;
    a.r.prop_ids[dst] = psrc;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.prop_pos_out[dst * 3 + c] = pp[c];
    *reinterpret_cast<float4*>(a.r.prop_rot_out + dst * 4) = pq;
  }
  if (dense) {
    const int64_t kk = (int64_t)k0 + tid;
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    *reinterpret_cast<float4*>(a.r.cabinet_dof_pos_out + kk * CD) = z4;  // :393-398
    *reinterpret_cast<float4*>(a.r.cabinet_dof_vel_out + kk * CD) = z4;
    a.b.reset_buf[di] = 0;
    a.b.progress_buf[di] = 0;
    a.r.env_ids[kk] = di;
  }
  {  // view rows k0 .. k0 + cnt of dof_pos_out / dof_vel_out: one contiguous run each, written lane-linearly from s_pos
    const int run = cnt * FD;
    float* dp = a.r.dof_pos_out + (int64_t)k0 * FD;
    float* dv = a.r.dof_vel_out + (int64_t)k0 * FD;
    for (int e = tid; e < run; e += GROUP) {
      const int r = e / FD, j = e - r * FD;
      dp[e] = s_pos[s_list[r] * FD + j];
      dv[e] = 0.0f;
    }
  }

  const bool clampa = a.flags & OIGE_PRE_CLAMP_ACTIONS;
#pragma unroll
  for (int it = 0; it < WIT; ++it) {
    const int e = 4 * (it * GROUP + tid);
    if (e >= total) continue;
    float (&av)[4] = avq[it];
    float (&tv)[4] = tvq[it];
    const bool vec = e + 3 < total;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
      const int ee = e + c;
      if (ee >= total) continue;
      const int il = ee / FD, j = ee - il * FD;
      if (clampa) av[c] = clamp_sym(av[c], p.clip_actions);
      const float start = s_rank[il] >= 0 ? s_pos[il * FD + j] : tv[c];
      tv[c] = tensor_clamp(start + (s_rate[j] * av[c]) * p.action_scale, s_lo[j], s_hi[j]);  // :369-370
    }
    if (vec) {
      *reinterpret_cast<float4*>(a.b.actions + base + e) = make_float4(av[0], av[1], av[2], av[3]);  // actions.clone() (:368)
      *reinterpret_cast<float4*>(a.b.franka_dof_targets + base + e) = make_float4(tv[0], tv[1], tv[2], tv[3]);
    } else {
      for (int c = 0; c < 4; ++c)
        if (e + c < total) { a.b.actions[base + e + c] = av[c]; a.b.franka_dof_targets[base + e + c] = tv[c]; }
    }
  }
  if (g == G - 1 && tid == 0) {
    *a.r.count = (int64_t)k0 + cnt;
    *a.r.prop_count = ((int64_t)k0 + cnt) * p.num_props;
  }
  // prop rows beyond the workgroup's width (more than 256 / num_props flagged envs in the group): loads behind stores, rare
  for (int e = tid + GROUP; e < prow; e += GROUP) {
    const int r = e / P, j = e - r * P;
    const int64_t src = (g * GROUP + s_list[r]) * P + j, dst = (int64_t)k0 * P + e;
    a.r.prop_ids[dst] = src;
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.prop_pos_out[dst * 3 + c] = a.r.default_prop_pos[src * 3 + c];
    *reinterpret_cast<float4*>(a.r.prop_rot_out + dst * 4) = *reinterpret_cast<const float4*>(a.r.default_prop_rot + src * 4);
  }
}

// reset_idx(env_ids) with caller-provided ids and (optionally) the pre-drawn (n,9) U[0,1): one thread per id
template <int FD>
__global__ __launch_bounds__(256) void franka_cabinet_reset_idx_kernel(const FrankaPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k == 0) *a.r.prop_count = *a.r.count * a.p.num_props;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  float u[FMAX], pos[FMAX];
  u[FMAX - 1] = 0.0f;
  if (a.r.u_dof) {
#pragma unroll
    for (int j = 0; j < FD; ++j) u[j] = a.r.u_dof[k * FD + j];
  } else {
    franka_draws(a, i, u);
  }
  franka_reset_rows<FD>(a, i, k, u, pos);
#pragma unroll
  for (int j = 0; j < FD; ++j) a.b.franka_dof_targets[i * FD + j] = pos[j];  // :389
}

static int check_franka(const oige_franka_cabinet_params* p, const oige_franka_cabinet_buffers* b, int64_t n, const char* who) {
  OIGE_REQUIRE(p && b, "%s: null params/buffers", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 128, "%s: num_envs %lld out of range", who, (long long)n);
  OIGE_REQUIRE(p->num_props >= 0 && p->num_props <= 64, "%s: num_props %d out of range [0, 64]", who, p->num_props);
  OIGE_REQUIRE(p->num_dofs == 9 || p->num_dofs == 8, "%s: num_dofs %d not supported (9: Franka, TomoDJ; 8: TomoSPC)", who, p->num_dofs);
  OIGE_REQUIRE(p->reward_variant == OIGE_CABINET_REWARD_FRANKA || p->reward_variant == OIGE_CABINET_REWARD_TOMO,
               "%s: unknown reward_variant %d", who, p->reward_variant);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_franka_cabinet_post_physics(const oige_franka_cabinet_params* params, const oige_franka_cabinet_state* state,
                                     const oige_franka_cabinet_buffers* buffers, int64_t num_envs, uint32_t stages,
                                     oige_stream_t stream) {
  if (int rc = check_franka(params, buffers, num_envs, "franka_cabinet_post_physics")) return rc;
  OIGE_REQUIRE(state != nullptr, "franka_cabinet_post_physics: null state");
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "franka_cabinet_post_physics: bad stages mask 0x%x", stages);
  const oige_franka_cabinet_state& s = *state;
  const oige_franka_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(s.hand_pos && s.hand_rot && s.drawer_pos && s.drawer_rot && s.franka_dof_pos && s.franka_dof_vel && s.cabinet_dof_pos &&
                   s.cabinet_dof_vel && s.lfinger_pos && s.rfinger_pos,
               "franka_cabinet_post_physics: null state pointer");
  OIGE_REQUIRE(b.obs_buf && b.rew_buf && b.reset_buf && b.progress_buf && b.actions, "franka_cabinet_post_physics: null buffer pointer");
  OIGE_REQUIRE(aligned16(s.hand_rot) && aligned16(s.drawer_rot) && aligned16(s.franka_dof_pos) && aligned16(s.franka_dof_vel) &&
                   aligned16(b.actions) && aligned16(b.obs_buf) && aligned16(b.franka_grasp_rot) && aligned16(b.drawer_grasp_rot),
               "franka_cabinet_post_physics: row buffers must be 16-byte aligned");
  const int64_t T = num_tiles(num_envs);
  if (T == 0) return OIGE_OK;
  FrankaPost a;
  a.p = *params;
  a.s = s;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  if (params->num_dofs == 9)
    hipLaunchKernelGGL(franka_cabinet_post_kernel<9>, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  else
    hipLaunchKernelGGL(franka_cabinet_post_kernel<8>, dim3((unsigned)T), dim3(WAVE), 0, (hipStream_t)stream, a);
  return check_launch("franka_cabinet_post_kernel");
}

static int fill_franka_pre(FrankaPre& a, const oige_franka_cabinet_params* params, const oige_franka_cabinet_buffers* buffers,
                           const oige_franka_cabinet_reset_io* r, int64_t n, const char* who) {
  if (int rc = check_franka(params, buffers, n, who)) return rc;
  OIGE_REQUIRE(r != nullptr, "%s: null reset_io", who);
  const oige_franka_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(b.reset_buf && b.progress_buf && b.franka_dof_targets, "%s: null buffer pointer", who);
  OIGE_REQUIRE(r->env_ids && r->count && r->dof_pos_out && r->dof_vel_out && r->cabinet_dof_pos_out && r->cabinet_dof_vel_out &&
                   r->prop_count,
               "%s: null reset pointer", who);
  OIGE_REQUIRE(params->num_props == 0 || (r->default_prop_pos && r->default_prop_rot && r->prop_ids && r->prop_pos_out && r->prop_rot_out),
               "%s: num_props > 0 needs default_prop_pos/rot, prop_ids, prop_pos_out, prop_rot_out", who);
  OIGE_REQUIRE(aligned16(r->cabinet_dof_pos_out) && aligned16(r->cabinet_dof_vel_out) && aligned16(r->default_prop_rot) &&
                   aligned16(r->prop_rot_out) && aligned16(b.franka_dof_targets),
               "%s: row buffers must be 16-byte aligned", who);
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.actions_in = nullptr;
  return OIGE_OK;
}

int oige_franka_cabinet_pre_physics(const oige_franka_cabinet_params* params, const float* actions_in,
                                    const oige_franka_cabinet_buffers* buffers,
                                    const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, uint32_t flags,
                                    oige_stream_t stream) {
  FrankaPre a;
  if (int rc = fill_franka_pre(a, params, buffers, reset_io, num_envs, "franka_cabinet_pre_physics")) return rc;
  const oige_franka_cabinet_buffers& b = *buffers;
  OIGE_REQUIRE(actions_in && aligned16(actions_in) && b.actions && aligned16(b.actions),
               "franka_cabinet_pre_physics: null/unaligned actions");
  OIGE_REQUIRE(b.reset_workspace && aligned16(b.reset_workspace), "franka_cabinet_pre_physics: null/unaligned workspace");
  OIGE_REQUIRE(!reset_io->u_dof, "franka_cabinet_pre_physics: pre-drawn noise goes through oige_franka_cabinet_reset_idx");
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
  if (params->num_dofs == 9) {
    if (rank == RANK_U8) hipLaunchKernelGGL((franka_cabinet_pre_kernel<9, RANK_U8>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
    else if (rank == RANK_INT32) hipLaunchKernelGGL((franka_cabinet_pre_kernel<9, RANK_INT32>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
    else hipLaunchKernelGGL((franka_cabinet_pre_kernel<9, RANK_SCANNED>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
  } else {
    if (rank == RANK_U8) hipLaunchKernelGGL((franka_cabinet_pre_kernel<8, RANK_U8>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
    else if (rank == RANK_INT32) hipLaunchKernelGGL((franka_cabinet_pre_kernel<8, RANK_INT32>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
    else hipLaunchKernelGGL((franka_cabinet_pre_kernel<8, RANK_SCANNED>), dim3((unsigned)G), dim3(GROUP), 0, s, a);
  }
  return check_launch("franka_cabinet_pre_kernel");
}

int oige_franka_cabinet_reset_idx(const oige_franka_cabinet_params* params, const oige_franka_cabinet_buffers* buffers,
                                  const oige_franka_cabinet_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                  oige_stream_t stream) {
  FrankaPre a;
  if (int rc = fill_franka_pre(a, params, buffers, reset_io, num_envs, "franka_cabinet_reset_idx")) return rc;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "franka_cabinet_reset_idx: bad max_ids");
  if (max_ids == 0) return OIGE_OK;
  const dim3 grid((unsigned)((max_ids + 255) / 256));
  if (params->num_dofs == 9)
    hipLaunchKernelGGL(franka_cabinet_reset_idx_kernel<9>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  else
    hipLaunchKernelGGL(franka_cabinet_reset_idx_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("franka_cabinet_reset_idx_kernel");
}

}  // extern "C"
