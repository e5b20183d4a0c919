// ShadowHand / AllegroHand in-hand manipulation per-step task math for gfx950 (MI355X).
//
// AllegroHand (tasks/allegro_hand.py:101-144: 16 DOFs, all actuated, no fingertip bodies, no states) shares the reward,
// reset and action-scaling code below (instantiated for 16 DOFs) and has its own one-wave-per-tile post kernel:
// 632 algorithmic bytes per env-step with the `full` 72-column observation.
//
// Replaces, behind the unchanged InHandManipulationTask / ShadowHandTask method contract:
//   get_observations    tasks/shadow_hand.py:120-262 (openai 42 / full 157 obs, asymmetric states 187)
//   calculate_metrics   tasks/shared/in_hand_manipulation.py:263-297, jit compute_hand_reward :457-531
//   pre_physics_step    in_hand_manipulation.py:314-362
//   reset_target_pose   :377-395, jit randomize_rotation :450-453
//   reset_idx           :397-441
//
// post_physics: one workgroup of 256 threads (4 waves) owns one tile of 64 envs.  Every input
// array's slice for the tile is one contiguous run, moved with lane-linear float4 loads and
// transposed through LDS into the obs / states tiles, which stream back to HBM as contiguous
// runs; wave 0 does the per-env quaternion / reward math in between.  HBM-bound: 1720 algorithmic
// bytes per env-step for openai obs + asymmetric states.
// The two global sums of compute_hand_reward (:522-523) are reduced deterministically: one
// (count, sum) pair per tile from a fixed shuffle tree, then a single-workgroup pass in tile order.
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_noise.hpp"
#include "oige_scan.hpp"
#include "oige_tile.hpp"

// lane-linear whole-line streams outside the tile helpers (noise epilogue / prologue, target walk)
// (non-temporal like those: ShadowHand + DR step 0.800 -> 0.763 ms, AllegroHand 0.304 -> 0.273 ms at 1 M envs)
#define LD4S(p) ld4_stream(p)
#define ST4S(p, v) st4_stream((p), (v))

namespace oige {

constexpr int HD = OIGE_HAND_DOFS;        // 24
constexpr int HA = OIGE_HAND_ACTIONS;     // 20
constexpr int HF = OIGE_HAND_FINGERTIPS;  // 5
constexpr int HS = OIGE_HAND_STATES;      // 187
#ifndef OIGE_HAND_NT
#define OIGE_HAND_NT 256
#endif
constexpr int HNT = OIGE_HAND_NT;         // threads per tile

struct HandPost {
  oige_hand_params p;
  oige_hand_state s;
  oige_hand_buffers b;
  int64_t n;
  uint32_t stages;
  float max_len_minus_1;   // (float)((double)max_episode_length - 1.0)
  float half_fall_penalty; // (float)(0.5 * (double)fall_penalty)
  // observation noise applied while the obs tile streams out (randomize.py:283-312 + the clamp of
  // vec_env_rlgames.py:40), instead of a separate pass over obs_buf; dr == 0: none
  oige_noise_args dn;
  int dr;
};

struct HandRewardOut {
  float reward, successes;
  int64_t resets, goal_resets, progress;
};

// compute_hand_reward for one env (:482-520)
__device__ __forceinline__ HandRewardOut hand_reward(const HandPost& a, const f3 object_pos, const q4 object_rot,
                                                     const f3 target_pos, const q4 target_rot, float action_penalty,
                                                     int64_t reset_in, int64_t reset_goal_in, int64_t progress,
                                                     float successes) {
  const oige_hand_params& p = a.p;
  const float goal_dist = norm3(f3{object_pos.x - target_pos.x, object_pos.y - target_pos.y, object_pos.z - target_pos.z});
  const q4 qd = quat_mul(object_rot, quat_conjugate(target_rot));
  float nv = norm3(f3{qd.x, qd.y, qd.z});
  nv = (nv != nv) ? nv : (nv > 1.0f ? 1.0f : nv);  // torch.clamp(max=1.0)
  const float rot_dist = 2.0f * asinf(nv);
  const float ard = __builtin_fabsf(rot_dist);
  const float dist_rew = goal_dist * p.dist_reward_scale;
  const float rot_rew = (1.0f / (ard + p.rot_eps)) * p.rot_reward_scale;
  float reward = (dist_rew + rot_rew) + action_penalty * p.action_penalty_scale;
  HandRewardOut o;
  o.goal_resets = ard <= p.success_tolerance ? 1 : reset_goal_in;
  o.successes = successes + (float)o.goal_resets;
  reward = o.goal_resets == 1 ? reward + p.reach_goal_bonus : reward;
  const bool fell = goal_dist >= p.fall_dist;
  reward = fell ? reward + p.fall_penalty : reward;
  int64_t resets = fell ? 1 : reset_in;
  if (p.max_consecutive_successes > 0) {
    progress = ard <= p.success_tolerance ? 0 : progress;
    resets = o.successes >= (float)p.max_consecutive_successes ? 1 : resets;
  }
  const bool timeout = (float)progress >= a.max_len_minus_1;
  resets = timeout ? 1 : resets;
  if (p.max_consecutive_successes > 0) reward = timeout ? reward + a.half_fall_penalty : reward;
  o.reward = reward;
  o.resets = resets;
  o.progress = progress;
  return o;
}

__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// ---- post_physics tile passes ------------------------------------------------------------------------------
// A 64-env tile is processed as two passes of HTE = 32 envs: the LDS a workgroup needs halves (~30 KB with the
// asymmetric states), so five workgroups (20 waves) stay resident per CU instead of two (the register budget is kept
// under 102 VGPRs for that: one pass's loads in flight at a time).  Per-tile results (reset / goal-reset counts, the two partial
// sums of compute_hand_reward) accumulate over the passes and are written once.
constexpr int HTE = 32;
// keeps the scheduler from hoisting every scatter's LDS address arithmetic above the first one (200 VGPRs -> spills)
#ifdef OIGE_NO_SCHED_FENCE
#define OIGE_SCHED_FENCE()
#else
#define OIGE_SCHED_FENCE() __builtin_amdgcn_sched_barrier(0)
#endif

#ifndef OIGE_HAND_TE
#define OIGE_HAND_TE HTE
#endif
constexpr int HPE = OIGE_HAND_TE;  // envs per pass (64 = the whole tile in one pass)

// in-kernel DR: the pass's slice of the correlated-noise buffer is fetched with the inputs where that costs <= 8 VGPRs
// (openai obs: 2 float4 per thread); the 157-column obs reads it inside the epilogue loop instead (5 float4 per thread)
template <int OW>
constexpr bool hand_prefetch_corr() { return WgChunk<OW, HNT, HPE>::ITERS <= 2; }

template <bool NEED_FULLCOLS, bool ASYM, int OW>
struct HandLoads {
  WgChunk<hand_prefetch_corr<OW>() ? OW : 4, HNT, HPE> c_corr;
  WgChunk<HD, HNT, HPE> c_dpos, c_dvel;
  WgChunk<3 * HF, HNT, HPE> c_ftp;
  WgChunk<4 * HF, HNT, HPE> c_ftr;
  WgChunk<6 * HF, HNT, HPE> c_ftv, c_sens;
  WgChunk<HA, HNT, HPE> c_act;
  f3 epos, opos, olin, oang, gpos;
  q4 orot, grot;
  int64_t reset_in, reset_goal_in, progress, rand_in;
  float successes;
  int cnt_in;  // in-kernel DR: the env's noise counter
};

struct HandTileAcc {
  int n_reset, n_goal;
  float nres, fin;
};

// OBS: 0 openai, 2 full.  ASYM: also states_buf.
template <int OBS, bool ASYM, bool FULL>
__device__ __forceinline__ void hand_post_load(const HandPost& a, int64_t env0, int nv,
                                               HandLoads<ASYM || OBS == OIGE_HAND_OBS_FULL, ASYM, OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157>& L,
                                               const int tid) {
  constexpr bool NEED_FULLCOLS = ASYM || OBS == OIGE_HAND_OBS_FULL;  // dof/ft_rot/ft_vel columns wanted
  constexpr int OW = OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS;
  if (a.stages & OIGE_STAGE_OBS) {
    wg_load_chunk<3 * HF, HNT, FULL, HPE>(L.c_ftp, a.s.fingertip_pos + env0 * 3 * HF, nv * 3 * HF, tid);
    wg_load_chunk<HA, HNT, FULL, HPE>(L.c_act, a.b.actions + env0 * HA, nv * HA, tid);
    if (NEED_FULLCOLS) {
      wg_load_chunk<HD, HNT, FULL, HPE>(L.c_dpos, a.s.hand_dof_pos + env0 * HD, nv * HD, tid);
      wg_load_chunk<HD, HNT, FULL, HPE>(L.c_dvel, a.s.hand_dof_vel + env0 * HD, nv * HD, tid);
      wg_load_chunk<4 * HF, HNT, FULL, HPE>(L.c_ftr, a.s.fingertip_rot + env0 * 4 * HF, nv * 4 * HF, tid);
      wg_load_chunk<6 * HF, HNT, FULL, HPE>(L.c_ftv, a.s.fingertip_velocities + env0 * 6 * HF, nv * 6 * HF, tid);
    }
    if (ASYM) wg_load_chunk<6 * HF, HNT, FULL, HPE>(L.c_sens, a.s.vec_sensor + env0 * 6 * HF, nv * 6 * HF, tid);
    if (hand_prefetch_corr<OW>() && a.dr && a.dn.on_reset.enabled)  // inside the epilogue loop it is a second exposed memory latency per pass
      wg_load_chunk<hand_prefetch_corr<OW>() ? OW : 4, HNT, FULL, HPE>(L.c_corr, a.dn.correlated_noise + env0 * OW, nv * OW, tid);
  }
  // per-env rows (first lanes of wave 0: one env per lane)
  const bool live = tid < HPE && (FULL || tid < nv);
  const int64_t env = env0 + tid;
  L.epos = L.opos = L.olin = L.oang = L.gpos = f3{0, 0, 0};
  L.orot = L.grot = q4{1, 0, 0, 0};
  L.reset_in = L.reset_goal_in = L.progress = L.rand_in = 0;
  L.successes = 0.f;
  L.cnt_in = 0;
  if (live) {
    const float* e3 = a.s.env_pos + env * 3;
    L.epos = f3{e3[0], e3[1], e3[2]};
    const float* o3 = a.s.object_pos + env * 3;
    L.opos = f3{o3[0], o3[1], o3[2]};
    const float4 r4 = *reinterpret_cast<const float4*>(a.s.object_rot + env * 4);
    L.orot = q4{r4.x, r4.y, r4.z, r4.w};
    if (a.stages & OIGE_STAGE_OBS) {
      const float2* vp = reinterpret_cast<const float2*>(a.s.object_velocities + env * 6);
      const float2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
      L.olin = f3{v0.x, v0.y, v1.x};
      L.oang = f3{v1.y, v2.x, v2.y};
    }
    const float* g3 = a.b.goal_pos + env * 3;
    L.gpos = f3{g3[0], g3[1], g3[2]};
    const float4 g4 = *reinterpret_cast<const float4*>(a.b.goal_rot + env * 4);
    L.grot = q4{g4.x, g4.y, g4.z, g4.w};
    if (do_progress || do_metrics) L.progress = a.b.progress_buf[env];
    if (do_metrics) {
      L.reset_in = a.b.reset_buf[env];
      L.reset_goal_in = a.b.reset_goal_buf[env];
      L.successes = a.b.successes[env];
      // read-modify-write state of the METRICS stage, fetched here with the inputs: loaded after the stage's stores,
      // each would wait for those stores' round trip to HBM (loads and stores share vmcnt)
      if (a.b.randomization_buf) L.rand_in = a.b.randomization_buf[env];
      if (a.dr) L.cnt_in = a.dn.counter[env];
    }
  }
}

template <int OBS, bool ASYM, bool FULL>
__device__ __forceinline__ void hand_post_process(const HandPost& a, float* __restrict__ lds, int64_t env0, int nv,
                                                  const HandLoads<ASYM || OBS == OIGE_HAND_OBS_FULL, ASYM, OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157>& L,
                                                  HandTileAcc& acc, const int tid) {
  constexpr int OW = OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157;
  constexpr bool NEED_FULLCOLS = ASYM || OBS == OIGE_HAND_OBS_FULL;
  // LDS carve: [obs tile | states tile | env_pos HPEx3 | lo 24 | hi 24 | noise decisions HPE | reset-env list HPE + count]
  float* obs_t = lds;
  float* st_t = lds + HPE * OW;
  float* s_env = st_t + (ASYM ? HPE * HS : 0);
  float* s_lo = s_env + HPE * 3;
  float* s_hi = s_lo + HD;
  int* s_dec = reinterpret_cast<int*>(s_hi + HD);  // HPE noise decisions (dr only)
  int* s_rl = s_dec + HPE;                         // envs of this pass that reset (compacted, ascending), then their count
  constexpr int AC0 = OBS == OIGE_HAND_OBS_OPENAI ? 22 : 137;  // first action column of the obs tile

  const oige_hand_params& p = a.p;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS;
  const bool envlane = tid < HPE;
  const bool live = envlane && (FULL || tid < nv);
  const int64_t env = env0 + tid;
  const f3 epos = L.epos, opos = L.opos, olin = L.olin, oang = L.oang, gpos = L.gpos;
  const q4 orot = L.orot, grot = L.grot;
  int64_t progress = L.progress;

  if (envlane) {
    s_env[tid * 3 + 0] = epos.x;
    s_env[tid * 3 + 1] = epos.y;
    s_env[tid * 3 + 2] = epos.z;
  }
  // The correlated-noise quads were fetched before env_pos, so they have arrived by now; pin that here.  Left to their
  // first use in the epilogue, the compiler waits for them with vmcnt(0) AFTER the previous quad's stores were issued,
  // i.e. every quad of noise waited for a store round trip to HBM.
  constexpr int CIT = WgChunk<OW, HNT, HPE>::ITERS;
  float4 cq[hand_prefetch_corr<OW>() ? CIT : 1];
  if (hand_prefetch_corr<OW>()) {
#pragma unroll
    for (int ci = 0; ci < CIT; ++ci) {
      cq[ci] = L.c_corr.v[hand_prefetch_corr<OW>() ? ci : 0];
      asm volatile("" : "+v"(cq[ci].x), "+v"(cq[ci].y), "+v"(cq[ci].z), "+v"(cq[ci].w));
    }
  }
  {
    const int j = HNT > TILE ? tid - TILE : tid;  // a second wave fills the limit tables where there is one
    if (j >= 0 && j < HD) {
      s_lo[j] = p.dof_lower[j];
      s_hi[j] = p.dof_upper[j];
    }
  }
  __syncthreads();

  // ---- B: pass-through columns
  if (a.stages & OIGE_STAGE_OBS) {
    wg_for_each<3 * HF, HNT, HPE>(L.c_ftp, [&](int e, int c, float v) {
      const float w = v - s_env[e * 3 + (c % 3)];  // fingertip_pos -= env_pos (shadow_hand.py:124-126)
      if (OBS == OIGE_HAND_OBS_OPENAI) obs_t[e * OW + c] = w;
      if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + 72 + c] = w;
      if (ASYM) st_t[e * HS + 72 + c] = w;
    }, tid);
    OIGE_SCHED_FENCE();
    wg_for_each<HA, HNT, HPE>(L.c_act, [&](int e, int c, float v) {
      if (OBS == OIGE_HAND_OBS_OPENAI) obs_t[e * OW + 22 + c] = v;
      if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + 137 + c] = v;
      if (ASYM) st_t[e * HS + 167 + c] = v;
    }, tid);
    OIGE_SCHED_FENCE();
    if (NEED_FULLCOLS) {
      wg_for_each<HD, HNT, HPE>(L.c_dpos, [&](int e, int c, float v) {
        const float w = unscale(v, s_lo[c], s_hi[c]);
        if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + c] = w;
        if (ASYM) st_t[e * HS + c] = w;
      }, tid);
      OIGE_SCHED_FENCE();
      wg_for_each<HD, HNT, HPE>(L.c_dvel, [&](int e, int c, float v) {
        const float w = p.vel_obs_scale * v;
        if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + HD + c] = w;
        if (ASYM) st_t[e * HS + HD + c] = w;
      }, tid);
      OIGE_SCHED_FENCE();
      wg_for_each<4 * HF, HNT, HPE>(L.c_ftr, [&](int e, int c, float v) {
        if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + 87 + c] = v;
        if (ASYM) st_t[e * HS + 87 + c] = v;
      }, tid);
      OIGE_SCHED_FENCE();
      wg_for_each<6 * HF, HNT, HPE>(L.c_ftv, [&](int e, int c, float v) {
        if (OBS == OIGE_HAND_OBS_FULL) obs_t[e * OW + 107 + c] = v;
        if (ASYM) st_t[e * HS + 107 + c] = v;
      }, tid);
      OIGE_SCHED_FENCE();
    }
    if (ASYM)
      wg_for_each<6 * HF, HNT, HPE>(L.c_sens, [&](int e, int c, float v) { st_t[e * HS + 137 + c] = p.force_torque_obs_scale * v; }, tid);
  }

  // ---- C: per-env math (wave 0)
  float action_penalty = 0.f;
  const bool tile_actions = (a.stages & OIGE_STAGE_OBS) && do_metrics;  // the action columns are in the tile: no 2nd HBM pass
  if (tile_actions) __syncthreads();
  if (tid < WAVE) {
    const f3 opl{opos.x - epos.x, opos.y - epos.y, opos.z - epos.z};  // object_pos -= env_pos (:258)
    if ((a.stages & OIGE_STAGE_OBS) && envlane) {
      const q4 rel = quat_mul(orot, quat_conjugate(grot));
      if (OBS == OIGE_HAND_OBS_OPENAI) {
        float* r = obs_t + tid * OW;
        r[15] = opl.x; r[16] = opl.y; r[17] = opl.z;
        r[18] = rel.w; r[19] = rel.x; r[20] = rel.y; r[21] = rel.z;
      }
      if (NEED_FULLCOLS) {
        const float blk[24] = {opl.x, opl.y, opl.z, orot.w, orot.x, orot.y, orot.z, olin.x, olin.y, olin.z,
                               p.vel_obs_scale * oang.x, p.vel_obs_scale * oang.y, p.vel_obs_scale * oang.z,
                               gpos.x, gpos.y, gpos.z, grot.w, grot.x, grot.y, grot.z, rel.w, rel.x, rel.y, rel.z};
        if (OBS == OIGE_HAND_OBS_FULL) {
#pragma unroll
          for (int k = 0; k < 24; ++k) obs_t[tid * OW + 48 + k] = blk[k];
        }
        if (ASYM) {
#pragma unroll
          for (int k = 0; k < 24; ++k) st_t[tid * HS + 48 + k] = blk[k];
        }
      }
    }
    if (do_progress) progress += 1;  // rl_task.py:289
    if (do_metrics) {
      // torch.sum(actions**2, dim=-1) (:493); the tile row is read straight from global here because
      // with OBS skipped the tile holds no actions
      const float4* ar = reinterpret_cast<const float4*>(a.b.actions + env * HA);
      if (tile_actions) {
        if (envlane) {
          const float* at = obs_t + tid * OW + AC0;  // same values, same summation order as the global walk below
#pragma unroll
          for (int k = 0; k < HA; ++k) {
            const float sq = at[k] * at[k];
            action_penalty = k == 0 ? sq : action_penalty + sq;
          }
        }
      } else if (live) {
#pragma unroll
        for (int k = 0; k < HA / 4; ++k) {
          const float4 v = ar[k];
          const float s0 = v.x * v.x, s1 = v.y * v.y, s2 = v.z * v.z, s3 = v.w * v.w;
          action_penalty = k == 0 ? s0 : action_penalty + s0;
          action_penalty += s1;
          action_penalty += s2;
          action_penalty += s3;
        }
      }
      const HandRewardOut o = hand_reward(a, opl, orot, gpos, grot, action_penalty, L.reset_in, L.reset_goal_in, progress, L.successes);
      if (live) {
        a.b.rew_buf[env] = o.reward;
        a.b.reset_buf[env] = o.resets;
        a.b.reset_goal_buf[env] = o.goal_resets;
        a.b.progress_buf[env] = o.progress;
        a.b.successes[env] = o.successes;
        if (a.b.randomization_buf) a.b.randomization_buf[env] = L.rand_in + 1;  // :297
      }
      if (a.dr) {  // the wrapper hands the noise the reset_buf this step just produced (vec_env_rlgames.py:94-97)
        NoiseDecision d{false, false, 0};
        if (envlane) {
          d = noise_decide_cnt(a.dn, L.cnt_in, live, live && o.resets != 0);
          if (live) a.dn.counter[env] = d.cnt;
          s_dec[tid] = (d.reset ? 1 : 0) | (d.interval ? 2 : 0);
        }
        // the pass's reset envs, compacted: their correlated noise is refreshed on dense lanes below
        const bool rs = envlane && d.reset && a.dn.on_reset.enabled;
        const unsigned long long mrs = __ballot(rs);
        if (rs) s_rl[rank_below(mrs)] = tid;
        if (tid == 0) s_rl[HPE] = __popcll(mrs);
      }
      const bool r1 = live && o.resets != 0, g1 = live && o.goal_resets != 0;
      const unsigned long long mr = __ballot(r1), mg = __ballot(g1);
      // Sum(resets) counts VALUES (a stale reset_buf could hold values other than 0/1)
      acc.n_reset += __popcll(mr);
      acc.n_goal += __popcll(mg);
      acc.nres += wave_sum_f(live ? (float)o.resets : 0.f);
      acc.fin += wave_sum_f(live ? o.successes * (float)o.resets : 0.f);  // :523
    } else if (do_progress && live) {
      a.b.progress_buf[env] = progress;
    }
  }
  __syncthreads();

  // ---- D: tiles -> HBM
  if (a.stages & OIGE_STAGE_OBS) {
    if (a.dr) {
      // obs tile -> noise -> obs_buf (noised, as the reference leaves it) and the clamped copy the policy gets
      constexpr int NF4 = (HPE * OW) / 4;
      const int64_t base = env0 * OW;  // a multiple of 4: env0 is a multiple of HPE and HPE * OW % 4 == 0
      const int total = nv * OW;
      const bool has_reset = a.dn.on_reset.enabled != 0;
      // D0: correlated_noise[reset_ids] = sample (randomize.py:379-400) on the COMPACTED reset rows: work item = (reset
      // env, one of the float4 groups its row touches), dense over the workgroup's lanes.  The fresh value goes to the
      // buffer in HBM and is folded into the obs tile here; D1 then skips the stored value for those elements.
      const int n_rs = has_reset ? s_rl[HPE] : 0;  // uniform over the workgroup (dr implies the METRICS stage)
      if (n_rs > 0) {
        constexpr int GPR = (OW + 3 + 3) / 4;  // float4 groups of the flat (N,OW) buffer one row can touch
        const bool scale_r = a.dn.on_reset.operation == OIGE_NOISE_SCALING;
        for (int it = tid; it < n_rs * GPR; it += HNT) {
          const int r = it / GPR, j = it - r * GPR;
          const int le = s_rl[r];
          const int64_t row0 = base + (int64_t)le * OW;  // flat index of the row's first element
          const int64_t gf = (row0 >> 2) + j;
          const int64_t el0 = gf << 2;
          if (el0 < row0 + OW) {
            float sr[4];
            noise_reset_samples(a.dn, gf, sr);
#pragma unroll
            for (int k = 0; k < 4; ++k) {
              const int col = (int)(el0 + k - row0);
              if (col >= 0 && col < OW) {
                a.dn.correlated_noise[el0 + k] = sr[k];
                float* o = obs_t + le * OW + col;
                *o = scale_r ? *o * sr[k] : *o + sr[k];
              }
            }
          }
        }
        __syncthreads();
      }
#pragma unroll
      for (int ci = 0; ci < CIT; ++ci) {
        const int f = ci * HNT + tid;
        if (f >= NF4) continue;
        const int e = 4 * f;
        if (!FULL && e >= total) continue;
        const int nvalid = FULL ? 4 : (total - e < 4 ? total - e : 4);
        const float4 t4 = *reinterpret_cast<const float4*>(obs_t + e);
        float v[4] = {t4.x, t4.y, t4.z, t4.w};
        float c[4] = {0.f, 0.f, 0.f, 0.f};
        if (hand_prefetch_corr<OW>()) {
          const float4 q = cq[hand_prefetch_corr<OW>() ? ci : 0];
          c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
        } else if (has_reset) {
          const float* cor = a.dn.correlated_noise + base + e;
          if (nvalid == 4) {
            const float4 q = LD4S(cor);
            c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
          } else {
            for (int k = 0; k < nvalid; ++k) c[k] = cor[k];
          }
        }
        int le = e / OW, col = e - le * OW, dec[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          dec[k] = s_dec[le < HPE ? le : HPE - 1];
          if (++col == OW) { col = 0; ++le; }
        }
        noise_apply4_refreshed(a.dn, dec, nvalid, (base + e) >> 2, v, c);
        float* ob = a.b.obs_buf + base + e;
        float* cl = a.dn.clamped_out ? a.dn.clamped_out + base + e : nullptr;
        if (nvalid == 4) {
          ST4S(ob, make_float4(v[0], v[1], v[2], v[3]));
          if (cl)
            ST4S(cl, make_float4(clamp_sym(v[0], a.dn.clip), clamp_sym(v[1], a.dn.clip), clamp_sym(v[2], a.dn.clip),
                                 clamp_sym(v[3], a.dn.clip)));
        } else {
          for (int k = 0; k < nvalid; ++k) {
            ob[k] = v[k];
            if (cl) cl[k] = clamp_sym(v[k], a.dn.clip);
          }
        }
      }
    } else {
      wg_store_tile<OW, HNT, FULL, HPE>(obs_t, a.b.obs_buf + env0 * OW, nv * OW, tid);
    }
    if (ASYM) wg_store_tile<HS, HNT, FULL, HPE>(st_t, a.b.states_buf + env0 * HS, nv * HS, tid);
  }
}

template <int OBS, bool ASYM, bool FULL>
__device__ __forceinline__ void hand_post_tile(const HandPost& a, float* __restrict__ lds, int64_t t, int left) {
  constexpr int PASSES = TILE / HPE;
  HandTileAcc acc{0, 0, 0.f, 0.f};
#pragma unroll 1
  for (int h = 0; h < PASSES; ++h) {
    const int rest = left - h * HPE;
    const int nv = FULL ? HPE : (rest < 0 ? 0 : (rest > HPE ? HPE : rest));
    if (FULL || nv > 0) {
      // the thread id is laundered once per pass: everything derived from it (every LDS scatter address) would
      // otherwise be loop-invariant, and hoisting ~100 addresses out of the pass loop costs the occupancy this
      // two-pass scheme is for
      int tid = threadIdx.x;
      asm volatile("" : "+v"(tid));
      HandLoads<ASYM || OBS == OIGE_HAND_OBS_FULL, ASYM, OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157> L;
      hand_post_load<OBS, ASYM, FULL>(a, t * TILE + h * HPE, nv, L, tid);
      hand_post_process<OBS, ASYM, FULL>(a, lds, t * TILE + h * HPE, nv, L, acc, tid);
    }
    if (h + 1 < PASSES) __syncthreads();  // the LDS tiles are reused by the next pass
  }
  if ((a.stages & OIGE_STAGE_METRICS) && threadIdx.x == 0) {
    a.b.reset_workspace[t] = acc.n_reset;
    a.b.goal_workspace[t] = acc.n_goal;
    a.b.reduce_workspace[2 * t + 0] = acc.nres;
    a.b.reduce_workspace[2 * t + 1] = acc.fin;
  }
}

// four waves per SIMD = four workgroups per CU: the register allocator otherwise drifts a few registers past 128 and
// loses a workgroup per CU (no spills at 128; at 96, for a fifth workgroup, it spills 100+)
#ifndef OIGE_HAND_WAVES_PER_EU
#define OIGE_HAND_WAVES_PER_EU 4
#endif
#define OIGE_HAND_OCC __attribute__((amdgpu_waves_per_eu(OIGE_HAND_WAVES_PER_EU)))
template <int OBS, bool ASYM>
__global__ __launch_bounds__(HNT) OIGE_HAND_OCC void hand_post_kernel(const HandPost a) {
  extern __shared__ __align__(16) float lds[];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    hand_post_tile<OBS, ASYM, true>(a, lds, t, TILE);
  else
    hand_post_tile<OBS, ASYM, false>(a, lds, t, (int)left);
}

// ---- AllegroHand post_physics (allegro_hand.py:101-144 + compute_hand_reward) --------------------------------
// One wavefront per tile of 64 envs.  dof_pos / dof_vel / actions arrive as lane-linear float4 runs; with 16 columns
// a lane's float4 always covers the same four DOFs (column = 4*lane mod 16), so its unscale limits are loop-invariant.
constexpr int AD = OIGE_ALLEGRO_DOFS;  // 16 DOFs = 16 actions

template <int OW, int COL0>
__device__ __forceinline__ void allegro_put4(float* __restrict__ tile, int envl, int col, const float4 v) {
  float* d = tile + envl * OW + COL0 + col;
  if ((OW % 4 == 0) && (COL0 % 4 == 0)) {
    *reinterpret_cast<float4*>(d) = v;
  } else {
    d[0] = v.x; d[1] = v.y; d[2] = v.z; d[3] = v.w;
  }
}

template <int OBS, bool FULL>
__device__ __forceinline__ void allegro_post_tile(const HandPost& a, float* __restrict__ tile, int64_t t, int nv) {
  constexpr bool VEL = OBS == OIGE_HAND_OBS_ALLEGRO_FULL;
  constexpr int OW = VEL ? 72 : 50;
  constexpr int C_OBJ = VEL ? 32 : 16, C_ACT = VEL ? 56 : 34;
  const oige_hand_params& p = a.p;
  const int lane = lane_id();
  const int64_t env0 = t * TILE;
  const int64_t env = env0 + lane;
  const bool live = FULL || lane < nv;
  const bool do_progress = a.stages & OIGE_STAGE_PROGRESS, do_obs = a.stages & OIGE_STAGE_OBS;
  const bool do_metrics = a.stages & OIGE_STAGE_METRICS;

  // ---- A: loads in flight
  Chunk<AD> c_pos, c_vel, c_act;
  if (do_obs) {
    load_chunk<AD, FULL>(c_pos, a.s.hand_dof_pos + env0 * AD, nv * AD);
    if (VEL) load_chunk<AD, FULL>(c_vel, a.s.hand_dof_vel + env0 * AD, nv * AD);
    load_chunk<AD, FULL>(c_act, a.b.actions + env0 * AD, nv * AD);
  }
  f3 epos{0, 0, 0}, opos{0, 0, 0}, olin{0, 0, 0}, oang{0, 0, 0}, gpos{0, 0, 0};
  q4 orot{1, 0, 0, 0}, grot{1, 0, 0, 0};
  int64_t reset_in = 0, reset_goal_in = 0, progress = 0;
  float successes = 0.f, action_penalty = 0.f;
  if (live) {
    const float* e3 = a.s.env_pos + env * 3;
    epos = f3{e3[0], e3[1], e3[2]};
    const float* o3 = a.s.object_pos + env * 3;
    opos = f3{o3[0], o3[1], o3[2]};
    const float4 r4 = *reinterpret_cast<const float4*>(a.s.object_rot + env * 4);
    orot = q4{r4.x, r4.y, r4.z, r4.w};
    if (do_obs && VEL) {
      const float2* vp = reinterpret_cast<const float2*>(a.s.object_velocities + env * 6);
      const float2 v0 = vp[0], v1 = vp[1], v2 = vp[2];
      olin = f3{v0.x, v0.y, v1.x};
      oang = f3{v1.y, v2.x, v2.y};
    }
    const float* g3 = a.b.goal_pos + env * 3;
    gpos = f3{g3[0], g3[1], g3[2]};
    const float4 g4 = *reinterpret_cast<const float4*>(a.b.goal_rot + env * 4);
    grot = q4{g4.x, g4.y, g4.z, g4.w};
    if (do_progress || do_metrics) progress = a.b.progress_buf[env];
    if (do_metrics) {
      reset_in = a.b.reset_buf[env];
      reset_goal_in = a.b.reset_goal_buf[env];
      successes = a.b.successes[env];
      if (!do_obs) {  // split stages: no tile to take the actions from
        const float4* ar = reinterpret_cast<const float4*>(a.b.actions + env * AD);  // torch.sum(actions**2, dim=-1) (:493)
#pragma unroll
        for (int k = 0; k < AD / 4; ++k) {
          const float4 v = ar[k];
          const float s0 = v.x * v.x, s1 = v.y * v.y, s2 = v.z * v.z, s3 = v.w * v.w;
          action_penalty = k == 0 ? s0 : action_penalty + s0;
          action_penalty += s1;
          action_penalty += s2;
          action_penalty += s3;
        }
      }
    }
  }

  // ---- B: DOF / action columns into the tile
  const f3 opl{opos.x - epos.x, opos.y - epos.y, opos.z - epos.z};  // object_pos -= env_pos (in_hand_manipulation.py:258)
  if (do_obs) {
    const int col = (4 * lane) & (AD - 1);
    float lo[4], hi[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {  // selects over the scalar kernel arguments (a lane-indexed read would spill the table)
      lo[q] = col == 0 ? p.dof_lower[q] : col == 4 ? p.dof_lower[4 + q] : col == 8 ? p.dof_lower[8 + q] : p.dof_lower[12 + q];
      hi[q] = col == 0 ? p.dof_upper[q] : col == 4 ? p.dof_upper[4 + q] : col == 8 ? p.dof_upper[8 + q] : p.dof_upper[12 + q];
    }
#pragma unroll
    for (int it = 0; it < Chunk<AD>::ITERS; ++it) {
      const int envl = (it * WAVE + lane) >> 2;  // 4 float4 per 16-float row
      const float4 q4v = c_pos.v[it];
      allegro_put4<OW, 0>(tile, envl, col,
                          make_float4(unscale(q4v.x, lo[0], hi[0]), unscale(q4v.y, lo[1], hi[1]), unscale(q4v.z, lo[2], hi[2]),
                                      unscale(q4v.w, lo[3], hi[3])));  // :119-121 / :130-132
      if (VEL) {
        const float4 d = c_vel.v[it];
        allegro_put4<OW, AD>(tile, envl, col, make_float4(p.vel_obs_scale * d.x, p.vel_obs_scale * d.y, p.vel_obs_scale * d.z,
                                                          p.vel_obs_scale * d.w));  // :133
      }
      allegro_put4<OW, C_ACT>(tile, envl, col, c_act.v[it]);  // :128 / :144
    }
    // ---- C: the per-env block, one env per lane
    const q4 rel = quat_mul(orot, quat_conjugate(grot));
    float* r = tile + lane * OW + C_OBJ;
    if (VEL) {  // :135-142
      const float blk[24] = {opl.x, opl.y, opl.z, orot.w, orot.x, orot.y, orot.z, olin.x, olin.y, olin.z,
                             p.vel_obs_scale * oang.x, p.vel_obs_scale * oang.y, p.vel_obs_scale * oang.z,
                             gpos.x, gpos.y, gpos.z, grot.w, grot.x, grot.y, grot.z, rel.w, rel.x, rel.y, rel.z};
#pragma unroll
      for (int k = 0; k < 24; ++k) r[k] = blk[k];
    } else {  // :123-127
      const float blk[18] = {opl.x, opl.y, opl.z, orot.w, orot.x, orot.y, orot.z,
                             gpos.x, gpos.y, gpos.z, grot.w, grot.x, grot.y, grot.z, rel.w, rel.x, rel.y, rel.z};
#pragma unroll
      for (int k = 0; k < 18; ++k) r[k] = blk[k];
    }
  }

  if (do_obs && do_metrics) {
    // torch.sum(actions**2, dim=-1) (:493) from the action columns the tile already holds, same left-to-right order: a second
    // pass over a.b.actions in HBM would re-read 64 B/env that the streaming (non-temporal) chunk load did not leave in cache
    __syncthreads();  // single-wave workgroup: B's cross-lane LDS writes visible
    const float* ar = tile + lane * OW + C_ACT;
#pragma unroll
    for (int k = 0; k < AD; ++k) {
      const float sq = ar[k] * ar[k];
      action_penalty = k == 0 ? sq : action_penalty + sq;
    }
  }

  if (do_progress) progress += 1;  // rl_task.py:289
  if (do_metrics) {
    const HandRewardOut o = hand_reward(a, opl, orot, gpos, grot, action_penalty, reset_in, reset_goal_in, progress, successes);
    if (live) {
      a.b.rew_buf[env] = o.reward;
      a.b.reset_buf[env] = o.resets;
      a.b.reset_goal_buf[env] = o.goal_resets;
      a.b.progress_buf[env] = o.progress;
      a.b.successes[env] = o.successes;
      if (a.b.randomization_buf) a.b.randomization_buf[env] += 1;  // :297
    }
    const unsigned long long mr = __ballot(live && o.resets != 0), mg = __ballot(live && o.goal_resets != 0);
    const float nres = wave_sum_f(live ? (float)o.resets : 0.f);
    const float fin = wave_sum_f(live ? o.successes * (float)o.resets : 0.f);  // :523
    if (lane == 0) {
      a.b.reset_workspace[t] = __popcll(mr);
      a.b.goal_workspace[t] = __popcll(mg);
      a.b.reduce_workspace[2 * t + 0] = nres;
      a.b.reduce_workspace[2 * t + 1] = fin;
    }
  } else if (do_progress && live) {
    a.b.progress_buf[env] = progress;
  }
  __syncthreads();

  // ---- D: tile -> HBM
  if (do_obs) store_tile<OW, FULL>(tile, a.b.obs_buf + env0 * OW, nv * OW);
}

template <int OBS>
__global__ __launch_bounds__(WAVE) void allegro_post_kernel(const HandPost a) {
  __shared__ __align__(16) float tile[TILE * (OBS == OIGE_HAND_OBS_ALLEGRO_FULL ? 72 : 50)];
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (left >= TILE)
    allegro_post_tile<OBS, true>(a, tile, t, TILE);
  else
    allegro_post_tile<OBS, false>(a, tile, t, (int)left);
}

// consecutive_successes update (:522-529): single workgroup, tile order.  Workgroups 1 and 2 of the same launch run the
// group scans of the two compaction workspaces the post kernel just filled (the next pre_physics_step then starts at its
// kernel: OIGE_PRE_OFFSETS_VALID): three one-workgroup, latency-bound passes side by side instead of two launches.
__global__ __launch_bounds__(1024) void hand_finalize_kernel(const float* __restrict__ partials, int64_t tiles,
                                                             float* __restrict__ cons, float av, float one_minus_av,
                                                             int32_t* __restrict__ ws_a, int32_t* __restrict__ ws_b,
                                                             int64_t n_groups, int64_t padded) {
  static_assert(SCAN_THREADS == 1024, "the scan workgroups share this launch's block size");
  if (blockIdx.x > 0) {
    __shared__ int wave_tot[SCAN_THREADS / WAVE];
    int32_t* ws = blockIdx.x == 1 ? ws_a : ws_b;
    scan_groups_body(ws, n_groups, tiles, ws + padded, wave_tot);
    return;
  }
  __shared__ double s_n[1024 / WAVE], s_f[1024 / WAVE];
  double n = 0.0, f = 0.0;
#pragma unroll 8
  for (int64_t i = threadIdx.x; i < tiles; i += 1024) {  // same order per thread; the unrolled loads overlap
    n += (double)partials[2 * i];
    f += (double)partials[2 * i + 1];
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) {
    n += __shfl_xor(n, off, WAVE);
    f += __shfl_xor(f, off, WAVE);
  }
  if (lane_id() == 0) {
    s_n[threadIdx.x >> 6] = n;
    s_f[threadIdx.x >> 6] = f;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    double tn = 0.0, tf = 0.0;
    for (int i = 0; i < 1024 / WAVE; ++i) {
      tn += s_n[i];
      tf += s_f[i];
    }
    const float num_resets = (float)tn;  // int64 sum promoted to fp32 at the division
    const float fin = (float)tf;
    if (tn > 0.0) *cons = (av * fin) / num_resets + one_minus_av * (*cons);
  }
}

// ---------------------------------------------------------------- pre_physics / resets
struct HandPre {
  oige_hand_params p;
  oige_hand_buffers b;
  oige_hand_reset_io r;
  const float* actions_in;
  float* actions_out;
  float* targets_out;
  int64_t n;
  uint32_t flags;
  // action noise applied while the actions are read (randomize.py:314-342), instead of a separate pass; dr == 0: none.
  // dn.buffer == actions_in: the caller's tensor receives the noised (unclamped) values, as in the reference.
  oige_noise_args dn;
  int dr;
};

// U(-1,1) from Philox: torch_rand_float(-1, 1): (1 - (-1)) * u + (-1)
__device__ __forceinline__ float um11(uint32_t w) { return 2.0f * u01(w) + (-1.0f); }

// draw `count` U(-1,1) values for env i, stream `stream` into out[]
template <int COUNT>
__device__ __forceinline__ void hand_draw(const oige_hand_reset_io& r, int64_t i, uint32_t stream, float (&out)[COUNT]) {
#pragma unroll
  for (int blk = 0; blk < (COUNT + 3) / 4; ++blk) {
    const u4 c{(uint32_t)i, (uint32_t)blk | (stream << 16), (uint32_t)step_of(r), (uint32_t)(step_of(r) >> 32)};
    const u4 x = philox4x32_10(c, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    const uint32_t w[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int q = 0; q < 4; ++q)
      if (4 * blk + q < COUNT) out[4 * blk + q] = um11(w[q]);
  }
}

// quat_from_angle_axis(angle, unit axis e_k): normalize([cos(a/2), e_k * sin(a/2)]) with normalize(axis) = axis
template <int AXIS>
__device__ __forceinline__ q4 quat_from_angle_unit_axis(float angle) {
  const float th = angle / 2.0f;
  const float s = sinf(th), c = cosf(th);
  // normalize(axis): axis / clamp(norm, 1e-9) = (1,0,0)/1 -> exact; xyz = axis * sin
  q4 q{c, AXIS == 0 ? 1.0f * s : 0.0f * s, AXIS == 1 ? 1.0f * s : 0.0f * s, 0.0f * s};
  const float n = norm4(q);
  const float d = n < 1e-9f ? 1e-9f : n;
  return q4{q.w / d, q.x / d, q.y / d, q.z / d};
}

// randomize_rotation (:450-453)
__device__ __forceinline__ q4 randomize_rotation(float rand0, float rand1) {
  return quat_mul(quat_from_angle_unit_axis<0>(rand0 * OIGE_PI_F), quat_from_angle_unit_axis<1>(rand1 * OIGE_PI_F));
}

// reset_target_pose (:377-395) for env i -> compacted row k of (pos_out, rot_out), from rows the caller fetched
// (gi = goal_init_pos[i], ep = env_pos[i]) and the two rotation draws.  state_rot = false leaves goal_rot[i] alone: a
// goal-only reset of an env that reset_idx also resets this step (its nested reset_target_pose comes second in the
// reference, :324-329, and wins; goal_pos and reset_goal_buf get the same value from both)
__device__ __forceinline__ void hand_goal_write(const HandPre& a, int64_t i, int64_t k, float u0, float u1, const float (&gi)[3],
                                                const float (&ep)[3], bool state_rot, float* pos_out, float* rot_out) {
  const q4 nr = randomize_rotation(u0, u1);
  a.b.goal_pos[i * 3 + 0] = gi[0];
  a.b.goal_pos[i * 3 + 1] = gi[1];
  a.b.goal_pos[i * 3 + 2] = gi[2];
  if (state_rot) *reinterpret_cast<float4*>(a.b.goal_rot + i * 4) = make_float4(nr.w, nr.x, nr.y, nr.z);
  pos_out[k * 3 + 0] = (gi[0] + a.p.goal_displacement[0]) + ep[0];
  pos_out[k * 3 + 1] = (gi[1] + a.p.goal_displacement[1]) + ep[1];
  pos_out[k * 3 + 2] = (gi[2] + a.p.goal_displacement[2]) + ep[2];
  *reinterpret_cast<float4*>(rot_out + k * 4) = make_float4(nr.w, nr.x, nr.y, nr.z);
  a.b.reset_goal_buf[i] = 0;
}

__device__ __forceinline__ void hand_reset_goal(const HandPre& a, int64_t i, int64_t k, const float* u4row,
                                                uint32_t stream, float* pos_out, float* rot_out) {
  float u[4];
  if (u4row) {
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = u4row[k * 4 + q];
  } else {
    hand_draw<4>(a.r, i, stream, u);
  }
  const float* gp = a.r.goal_init_pos + i * 3;
  const float* epp = a.r.env_pos + i * 3;
  const float gi[3] = {gp[0], gp[1], gp[2]}, ep[3] = {epp[0], epp[1], epp[2]};
  hand_goal_write(a, i, k, u[0], u[1], gi, ep, true, pos_out, rot_out);
}

// reset_idx (:397-441) for env i -> compacted row k
template <int D>
__device__ __forceinline__ void hand_reset_env(const HandPre& a, int64_t i, int64_t k) {
  const oige_hand_params& p = a.p;
  float u[2 * D + 5];
  if (a.r.u_reset) {
#pragma unroll
    for (int q = 0; q < 2 * D + 5; ++q) u[q] = a.r.u_reset[k * (2 * D + 5) + q];
  } else {
    hand_draw<2 * D + 5>(a.r, i, 1u, u);
  }
  hand_reset_goal(a, i, k, a.r.u_reset_goal, 2u, a.r.r_goal_world_pos_out, a.r.r_goal_rot_out);  // :401
  const float* oi = a.r.object_init_pos + i * 3;
  const float* ep = a.r.env_pos + i * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) a.r.object_pos_out[k * 3 + c] = (oi[c] + p.reset_position_noise * u[c]) + ep[c];  // :404-406
  const q4 orot = randomize_rotation(u[3], u[4]);                                                       // :408-410
  *reinterpret_cast<float4*>(a.r.object_rot_out + k * 4) = make_float4(orot.w, orot.x, orot.y, orot.z);
#pragma unroll
  for (int c = 0; c < 6; ++c) a.r.object_vel_out[k * 6 + c] = 0.0f;  // :412-413
#pragma unroll
  for (int j = 0; j < D; ++j) {
    const float delta_max = p.dof_upper[j] - p.dof_default_pos[j];
    const float delta_min = p.dof_lower[j] - p.dof_default_pos[j];
    const float rand_delta = delta_min + ((delta_max - delta_min) * 0.5f) * (u[5 + j] + 1.0f);  // :419
    const float pos = p.dof_default_pos[j] + p.reset_dof_pos_noise * rand_delta;                 // :421
    const float vel = p.dof_default_vel[j] + p.reset_dof_vel_noise * u[5 + D + j];             // :426-429
    a.r.dof_pos_out[k * D + j] = pos;
    a.r.dof_vel_out[k * D + j] = vel;
    a.b.prev_targets[i * D + j] = pos;  // :431-432
    a.b.cur_targets[i * D + j] = pos;
  }
  a.b.progress_buf[i] = 0;  // :439-441
  a.b.reset_buf[i] = 0;
  a.b.successes[i] = 0.0f;
  a.r.env_ids[k] = i;
}

// draws idx0 .. idx0+3 of env i's U(-1,1) table on `stream`, idx0 = 4m + 1: words 1..3 of block m, word 0 of block m+1
__device__ __forceinline__ void hand_draw4_at(const oige_hand_reset_io& r, int64_t i, uint32_t stream, int idx0, float (&out)[4]) {
  const uint32_t m = (uint32_t)(idx0 >> 2);
  const u4 c0{(uint32_t)i, m | (stream << 16), (uint32_t)step_of(r), (uint32_t)(step_of(r) >> 32)};
  const u4 c1{(uint32_t)i, (m + 1u) | (stream << 16), (uint32_t)step_of(r), (uint32_t)(step_of(r) >> 32)};
  const u4 x0 = philox4x32_10(c0, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
  const u4 x1 = philox4x32_10(c1, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
  out[0] = um11(x0.y); out[1] = um11(x0.z); out[2] = um11(x0.w); out[3] = um11(x1.x);
}

// per-DOF rows of the params, copied to LDS by the kernels that index them by a run-time column: indexed in place
// (the kernel-argument block) such an access is a VECTOR load, and behind a store it waits for that store's round trip
struct HandDofTab {
  const float* lower;
  const float* upper;
  const float* def_pos;
  const float* def_vel;
};

// the re-drawn DOF state of reset env i, columns j0 .. j0+3 (:415-432): pos (= the new prev and cur targets) and vel
__device__ __forceinline__ void hand_redraw_quad(const HandPre& a, const HandDofTab& t, int64_t i, int D, int j0, float (&pos)[4],
                                                 float (&vel)[4]) {
  const oige_hand_params& p = a.p;
  float up[4], uv[4] = {0.f, 0.f, 0.f, 0.f};
  hand_draw4_at(a.r, i, 1u, 5 + j0, up);
  if (p.reset_dof_vel_noise != 0.0f) hand_draw4_at(a.r, i, 1u, 5 + D + j0, uv);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const int j = j0 + q;
    const float delta_max = t.upper[j] - t.def_pos[j];
    const float delta_min = t.lower[j] - t.def_pos[j];
    const float rand_delta = delta_min + ((delta_max - delta_min) * 0.5f) * (up[q] + 1.0f);  // :419
    pos[q] = t.def_pos[j] + p.reset_dof_pos_noise * rand_delta;                               // :421
    vel[q] = t.def_vel[j] + p.reset_dof_vel_noise * uv[q];                                    // :426-429
  }
}

// reset envs of a group whose re-drawn rows go through LDS (computed on dense lanes); a group with more (the first step
// after a full reset) handles the rest per lane inside the elementwise passes
constexpr int HAND_RMAX = 64;
// threads per group of 256 envs in hand_pre_kernel (template parameter HPNT): the elementwise passes are lane-linear over
// the group's rows, so the thread count is free; more threads = fewer float4 held per thread between the load phase and
// the store phases.  512 pays for the ShadowHand kernel with the action noise inside (201 -> 192 us: 157 VGPRs at 256
// threads), 256 everywhere else (AllegroHand: 94 vs 102 us).
#ifndef OIGE_HAND_PRE_NT
#define OIGE_HAND_PRE_NT 512
#endif
#ifdef OIGE_HAND_PRE_WAVES
#define OIGE_HAND_PRE_OCC __attribute__((amdgpu_waves_per_eu(OIGE_HAND_PRE_WAVES, OIGE_HAND_PRE_WAVES)))
#else
#define OIGE_HAND_PRE_OCC
#endif

// pre_physics_step (:314-362) for one group of 256 envs.  Order: EVERY load first, then every store -- loads and stores
// share one in-order counter, so a load issued behind a store waits for that store's round trip to HBM, and a load
// inside a conditional block makes its join wait for everything in flight (the first version ran load -> store ->
// load ... eleven times per thread, each step a full round trip: 3.6 TB/s).
//   0. issue the group's lane-linear loads: actions (256x20 contiguous floats), their correlated noise, prev / cur
//      targets (256x24); flags, noise counters, compaction offsets;
//   1. compaction ranks of reset_goal_buf / reset_buf, the flagged envs packed into LDS lists;
//   2. the flagged envs run reset_target_pose / reset_idx (without its DOF rows) on DENSE lanes -- reset envs on the
//      first lanes of the group, goal-only envs from lane 64 on, so the two run side by side on different SIMDs -- and
//      the DOF rows / correlated action noise of the reset envs are re-drawn on dense lanes too (one float4 per lane:
//      as a per-lane branch of the elementwise passes every wave holding one reset env ran the Philox draws for all
//      64 lanes, six times over), written as compacted dof_pos_out / dof_vel_out rows and staged in LDS;
//   3. actions: noise -> clamp -> actions_out, LDS tile;
//   4. prev/cur targets: reset rows take their staged DOF positions, then every actuated element gets
//      scale -> EMA -> clamp (or the relative-control update); results land in the LDS action tile;
//   5. the LDS tile streams out as targets_out.
template <int D, int A, bool DR, int HPNT>
__global__ __launch_bounds__(HPNT) OIGE_HAND_PRE_OCC void hand_pre_kernel(const HandPre a) {
  __shared__ int lds_a[GROUP_TILES], lds_b[GROUP_TILES];
  __shared__ int s_list_r[GROUP], s_list_g[GROUP], s_rank[GROUP];
  __shared__ int s_inv[D];
  __shared__ int s_dec[DR ? GROUP : 1];
  __shared__ __align__(16) float s_act[GROUP * A];
  __shared__ __align__(16) float s_rpos[HAND_RMAX * D], s_rvel[HAND_RMAX * D];
  __shared__ __align__(16) float s_rcor[DR ? HAND_RMAX * A : 4];
  __shared__ __align__(16) float s_tab[4 * D];
  const HandDofTab tab{s_tab, s_tab + D, s_tab + 2 * D, s_tab + 3 * D};
  const oige_hand_params& p = a.p;
  const int tid = threadIdx.x;
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + tid;
  const int64_t G = num_groups(a.n);
  const int32_t* off_r = a.b.reset_workspace + G * GROUP_TILES;
  const int32_t* off_g = a.b.goal_workspace + G * GROUP_TILES;
  const bool live = tid < GROUP && env < a.n;
  const int64_t envs_here = (a.n - g * GROUP) < GROUP ? (a.n - g * GROUP) : GROUP;
  const int64_t base_a = g * GROUP * A, base_d = g * GROUP * D;
  const int nf4a = (int)(envs_here * A / 4), nf4d = (int)(envs_here * D / 4);  // A % 4 == 0, D % 4 == 0
  static_assert(A % 4 == 0 && D % 4 == 0 && D <= 32, "whole float4 per row; the actuated set is a 32-bit mask");
  const bool has_cor = DR && a.dn.on_reset.enabled != 0;

  // ---- 0: loads.  None is conditional: a lane past the group's end re-reads the group's first float4, a quad that does
  // not need its cur_targets (all four DOFs actuated) re-reads its prev_targets quad, a missing correlated-noise buffer
  // reads the actions again (same lines, already on their way: no HBM traffic)
  // (the flags too: written as `live ? flag[env] : 0` each became a branch around a load with a full wait behind it -- two
  // serial round trips in front of everything else; r03)
  const int64_t env_ld = live ? env : g * GROUP;
  const int64_t f_env_ld = a.b.reset_buf[env_ld];
  const int64_t f_goal_ld = a.b.reset_goal_buf[env_ld];
  int cnt_in = 0;
  if (DR) cnt_in = a.dn.counter[env_ld];
  uint32_t amask = 0;  // uniform: bit j set = DOF j actuated
#pragma unroll
  for (int t = 0; t < A; ++t) amask |= 1u << p.actuated_dof_indices[t];
  constexpr int AIT = (GROUP * (A / 4) + HPNT - 1) / HPNT, DIT = (GROUP * (D / 4) + HPNT - 1) / HPNT;
  float4 act[AIT], cor[DR ? AIT : 1], prv[DIT], cur[DIT];
  {
    const float* cor_src = has_cor ? a.dn.correlated_noise : a.actions_in;
#pragma unroll
    for (int it = 0; it < AIT; ++it) {
      const int f = it * HPNT + tid;
      const int fa = f < nf4a ? f : 0;
      act[it] = LD4S(a.actions_in + base_a + 4 * fa);
      if (DR) cor[it] = LD4S(cor_src + base_a + 4 * fa);
    }
#pragma unroll
    for (int it = 0; it < DIT; ++it) {
      const int f = it * HPNT + tid;
      const int fd = f < nf4d ? f : 0;
      const int e = 4 * fd;
      const int j0 = e % D;
      const bool all_act = ((amask >> j0) & 0xFu) == 0xFu;
      prv[it] = LD4S(a.b.prev_targets + base_d + e);
      cur[it] = *reinterpret_cast<const float4*>((all_act ? a.b.prev_targets : a.b.cur_targets) + base_d + e);
    }
  }
  const int32_t k0g = off_g[g], k0r = off_r[g];
  const int cntg = off_g[g + 1] - k0g, cntr = off_r[g + 1] - k0r;
  // the per-step action noise of this thread's quads, drawn while the loads above are in flight (it depends on none of them)
  float nsi[DR ? AIT : 1][4];
  const bool draw_ahead = DR && a.dn.on_interval.enabled != 0;
  if (draw_ahead) {
#pragma unroll
    for (int it = 0; it < AIT; ++it) noise_interval_samples(a.dn, (base_a >> 2) + it * HPNT + tid, nsi[it]);
  }

  // ---- 1: ranks and lists
  const bool f_env = live && f_env_ld != 0, f_goal = live && f_goal_ld != 0;
  {  // the per-DOF rows of the argument block through compile-time indices (scalar loads: indexed by tid they are VECTOR
     // loads of the argument block, with a full wait for everything in flight behind them)
    float lo = 0.f, hi = 0.f, dp = 0.f, dv = 0.f;
    int inv = -1;
#pragma unroll
    for (int j = 0; j < D; ++j) {
      lo = tid == j ? p.dof_lower[j] : lo;
      hi = tid == j ? p.dof_upper[j] : hi;
      dp = tid == j ? p.dof_default_pos[j] : dp;
      dv = tid == j ? p.dof_default_vel[j] : dv;
    }
#pragma unroll
    for (int t = 0; t < A; ++t) inv = tid == p.actuated_dof_indices[t] ? t : inv;
    if (tid < D) {
      s_inv[tid] = inv;
      s_tab[tid] = lo;
      s_tab[D + tid] = hi;
      s_tab[2 * D + tid] = dp;
      s_tab[3 * D + tid] = dv;
    }
  }
  int my_cnt = 0;
  if (DR) {  // the wrapper hands the noise the reset_buf as it stands before this step's resets (vec_env_rlgames.py:72-75)
    const NoiseDecision d = noise_decide_cnt(a.dn, cnt_in, live, f_env);
    my_cnt = d.cnt;
    if (tid < GROUP) s_dec[tid] = (d.reset ? 1 : 0) | (d.interval ? 2 : 0);
  }
  const int64_t kg = group_rank(f_goal, off_g, g, lds_b);  // (barrier inside)
  const int64_t kr = group_rank(f_env, off_r, g, lds_a);
  if (f_goal) s_list_g[(int)(kg - k0g)] = tid;
  if (f_env) s_list_r[(int)(kr - k0r)] = tid;
  if (tid < GROUP) s_rank[tid] = f_env ? (int)(kr - k0r) : -1;
  __syncthreads();
  // (no wait_loads_done() here: measured 192 -> 198 us; the prev / cur target quads are not needed before phase 4)

  // ---- 2: the flagged envs' jobs on DENSE lanes: reset_idx's object rows and flags on lanes 0.., goal-only envs on lanes
  // 64.., reset_idx's nested reset_target_pose on lanes 128.. (three different waves).  Only their ROWS are fetched here (the
  // kernel's second and last round trip of loads, issued in front of every store); the jobs themselves -- two Philox blocks,
  // four sincos, scattered stores -- run behind phase 6, where no other wave waits for them (r03: in front of the barrier
  // below they were 2-4 us of every workgroup's ~20 us life).  A lane's first role keeps its two rows in registers; a lane
  // with a second role (more than 64 flagged envs in the group: the step after a full reset) fetches that role's rows late.
  const int tg = (tid + HPNT - WAVE) & (HPNT - 1), trg = (tid + HPNT - 2 * WAVE) & (HPNT - 1);
  static_assert((HPNT & (HPNT - 1)) == 0 && HPNT >= 256, "lane rotations assume a power-of-two workgroup of at least four waves");
  const bool do_r = tid < cntr, do_g = tg < cntg, do_rg = trg < cntr;
  const int il_r = do_r ? s_list_r[tid] : 0, il_g = do_g ? s_list_g[tg] : 0, il_rg = do_rg ? s_list_r[trg] : 0;
  const int64_t ir = g * GROUP + il_r, ig = g * GROUP + il_g, irg = g * GROUP + il_rg;
  const bool g_keeps_rot = do_g && s_rank[il_g] < 0;
  const int role = do_r ? 0 : do_g ? 1 : do_rg ? 2 : -1;
  float row_a[3], row_b[3];  // role 0: env_pos, object_init_pos; roles 1, 2: goal_init_pos, env_pos
  {  // unconditional (a lane without a role re-reads the group's first env: one request for all of them): inside a
     // conditional block the block's join would wait for the rows right here
    const int64_t i1 = role == 0 ? ir : role == 1 ? ig : role == 2 ? irg : g * GROUP;
    const float* src_a = role == 0 ? a.r.env_pos : a.r.goal_init_pos;
    const float* src_b = role == 0 ? a.r.object_init_pos : a.r.env_pos;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      row_a[c] = src_a[i1 * 3 + c];
      row_b[c] = src_b[i1 * 3 + c];
    }
  }
  if (DR && live) a.dn.counter[env] = my_cnt;  // (behind every load of this thread)
  {
    // DOF rows of the first HAND_RMAX reset envs, one float4 per lane, into LDS (their compacted rows leave in phase 6:
    // stores issued here would sit in front of phase 3's first use of the loaded actions); the correlated-noise samples
    // start half a workgroup further on, so that the two kinds of draw run on different waves
    const int tc = (tid + HPNT / 2) & (HPNT - 1);
    const int nr = cntr < HAND_RMAX ? cntr : HAND_RMAX;
    for (int w = tid; w < nr * (D / 4); w += HPNT) {
      const int r = w / (D / 4), j0 = 4 * (w - r * (D / 4));
      float pos[4], vel[4];
      hand_redraw_quad(a, tab, g * GROUP + s_list_r[r], D, j0, pos, vel);
      *reinterpret_cast<float4*>(s_rpos + r * D + j0) = make_float4(pos[0], pos[1], pos[2], pos[3]);
      *reinterpret_cast<float4*>(s_rvel + r * D + j0) = make_float4(vel[0], vel[1], vel[2], vel[3]);
    }
    if (DR && has_cor) {  // correlated_noise_buffer[reset_ids] = sample (randomize.py:379-400), same Philox key as noise_apply4
      for (int w = tc; w < nr * (A / 4); w += HPNT) {
        const int r = w / (A / 4), q = w - r * (A / 4);
        const int64_t gf = (base_a >> 2) + (int64_t)s_list_r[r] * (A / 4) + q;
        float sr[4];
        noise_reset_samples(a.dn, gf, sr);
        *reinterpret_cast<float4*>(s_rcor + r * A + 4 * q) = make_float4(sr[0], sr[1], sr[2], sr[3]);
      }
    }
  }
  __syncthreads();

  // ---- 3: actions -> (noise) -> (clamp) -> actions_out, LDS
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int f = it * HPNT + tid;
    if (f < nf4a) {
      float4 v = act[it];
      if (DR) {  // A % 4 == 0: the four elements belong to one env
        const int il = (4 * f) / A;
        float vv[4] = {v.x, v.y, v.z, v.w}, c[4] = {0.f, 0.f, 0.f, 0.f};
        if (has_cor) { c[0] = cor[it].x; c[1] = cor[it].y; c[2] = cor[it].z; c[3] = cor[it].w; }
        int de = s_dec[il];
#ifdef OIGE_EXP_PRE_NO_NOISE
        de = 0;
#endif
        const int lr = s_rank[il];
        if (has_cor && lr >= 0 && lr < HAND_RMAX) {  // staged fresh sample: applied like any env's correlated noise
          const float4 q = *reinterpret_cast<const float4*>(s_rcor + lr * A + (4 * f - il * A));
          c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
          de &= ~1;
        }
        const int dec[4] = {de, de, de, de};
        const bool dirty = draw_ahead ? noise_apply4_drawn(a.dn, dec, 4, (base_a >> 2) + f, vv, c, nsi[it])
                                      : noise_apply4(a.dn, dec, 4, (base_a >> 2) + f, vv, c);
        v = make_float4(vv[0], vv[1], vv[2], vv[3]);
        ST4S(a.dn.buffer + base_a + 4 * f, v);
        if (dirty) *reinterpret_cast<float4*>(a.dn.correlated_noise + base_a + 4 * f) = make_float4(c[0], c[1], c[2], c[3]);
      }
      if (a.flags & OIGE_PRE_CLAMP_ACTIONS) {
        v.x = clamp_sym(v.x, p.clip_actions); v.y = clamp_sym(v.y, p.clip_actions);
        v.z = clamp_sym(v.z, p.clip_actions); v.w = clamp_sym(v.w, p.clip_actions);
      }
      if (a.actions_out) *reinterpret_cast<float4*>(a.actions_out + base_a + 4 * f) = v;
      *reinterpret_cast<float4*>(s_act + 4 * f) = v;
    }
  }
  __syncthreads();

  // ---- 4: prev/cur targets, lane-linear over the group's 256 x 24 floats
  const float one_minus = p.one_minus_act_moving_average;
  const float speed_dt = p.dof_speed_scale_dt;
#pragma unroll
  for (int it = 0; it < DIT; ++it) {
    const int f = it * HPNT + tid;
    if (f >= nf4d) continue;
    const int e = 4 * f;
    const int il = e / D, j0 = e - il * D;
    const int lr = s_rank[il];
    int kk[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) kk[q] = s_inv[j0 + q];
    float prev[4] = {prv[it].x, prv[it].y, prv[it].z, prv[it].w};
    float cu[4] = {cur[it].x, cur[it].y, cur[it].z, cur[it].w};  // an unactuated DOF keeps its current target
    if (lr >= 0) {  // reset row: re-drawn DOF state (:415-436)
      if (lr < HAND_RMAX) {
        const float4 q = *reinterpret_cast<const float4*>(s_rpos + lr * D + j0);
        prev[0] = q.x; prev[1] = q.y; prev[2] = q.z; prev[3] = q.w;
      } else {
        float vel[4];
        hand_redraw_quad(a, tab, g * GROUP + il, D, j0, prev, vel);
        const int64_t row = ((int64_t)k0r + lr) * D + j0;
        *reinterpret_cast<float4*>(a.r.dof_pos_out + row) = make_float4(prev[0], prev[1], prev[2], prev[3]);
        *reinterpret_cast<float4*>(a.r.dof_vel_out + row) = make_float4(vel[0], vel[1], vel[2], vel[3]);
      }
#pragma unroll
      for (int q = 0; q < 4; ++q) cu[q] = prev[q];  // :431-432
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      if (kk[q] < 0) continue;
      const int j = j0 + q;
      const float lo = tab.lower[j], hi = tab.upper[j];
      const float ac = s_act[il * A + kk[q]];
      float c;
      if (p.use_relative_control) {
        c = tensor_clamp(prev[q] + speed_dt * ac, lo, hi);  // :334-341
      } else {
        c = scale(ac, lo, hi);                                // :343-347
        c = p.act_moving_average * c + one_minus * prev[q];   // :348-351
        c = tensor_clamp(c, lo, hi);                           // :352-356
      }
      cu[q] = c;
      prev[q] = c;  // :358
      s_act[il * A + kk[q]] = c;  // the action slot becomes the target slot (same thread reads and writes it)
    }
    ST4S(a.b.cur_targets + base_d + e, make_float4(cu[0], cu[1], cu[2], cu[3]));
    ST4S(a.b.prev_targets + base_d + e, make_float4(prev[0], prev[1], prev[2], prev[3]));
  }
  __syncthreads();

  // ---- 5: targets_out = cur_targets[:, actuated] (:360-362)
#pragma unroll
  for (int it = 0; it < AIT; ++it) {
    const int f = it * HPNT + tid;
    if (f < nf4a) *reinterpret_cast<float4*>(a.targets_out + base_a + 4 * f) = *reinterpret_cast<const float4*>(s_act + 4 * f);
  }
  // ---- 6: the staged rows of the reset envs: compacted dof_pos_out / dof_vel_out rows (one contiguous run per group),
  // their refreshed correlated noise
  {
    const int nr = cntr < HAND_RMAX ? cntr : HAND_RMAX;
    for (int w = tid; w < nr * (D / 4); w += HPNT) {
      const int64_t row = (int64_t)k0r * D + 4 * w;
      *reinterpret_cast<float4*>(a.r.dof_pos_out + row) = *reinterpret_cast<const float4*>(s_rpos + 4 * w);
      *reinterpret_cast<float4*>(a.r.dof_vel_out + row) = *reinterpret_cast<const float4*>(s_rvel + 4 * w);
    }
    if (DR && has_cor) {
      for (int w = tid; w < nr * (A / 4); w += HPNT) {
        const int r = w / (A / 4), q = w - r * (A / 4);
        const int64_t gf = (base_a >> 2) + (int64_t)s_list_r[r] * (A / 4) + q;
        *reinterpret_cast<float4*>(a.dn.correlated_noise + 4 * gf) = *reinterpret_cast<const float4*>(s_rcor + r * A + 4 * q);
      }
    }
  }
  if (g == 0 && tid == 0) {
    *a.r.count = (int64_t)off_r[G];
    *a.r.goal_count = (int64_t)off_g[G];
  }
  // ---- 7: the dense jobs (rows fetched in phase 2).  A wave without one ends here.
  if (role < 0) return;
  if (do_r) {  // reset_idx (:397-441) without its DOF rows and its goal
    const int64_t k = (int64_t)k0r + tid;
    float u[8];
    hand_draw<8>(a.r, ir, 1u, u);  // draws 0..4 of the (n,53) table
#pragma unroll
    for (int c = 0; c < 3; ++c) a.r.object_pos_out[k * 3 + c] = (row_b[c] + p.reset_position_noise * u[c]) + row_a[c];  // :404-406
    const q4 orot = randomize_rotation(u[3], u[4]);                                                               // :408-410
    *reinterpret_cast<float4*>(a.r.object_rot_out + k * 4) = make_float4(orot.w, orot.x, orot.y, orot.z);
#pragma unroll
    for (int c = 0; c < 6; ++c) a.r.object_vel_out[k * 6 + c] = 0.0f;  // :412-413
    a.b.progress_buf[ir] = 0;  // :439-441
    a.b.reset_buf[ir] = 0;
    a.b.successes[ir] = 0.0f;
    a.r.env_ids[k] = ir;
  }
  if (do_g) {  // goal-only ids (:324-327); an env that is in both lists keeps reset_idx's rotation (:328-329)
    float gi[3], ep[3];
    if (role == 1) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { gi[c] = row_a[c]; ep[c] = row_b[c]; }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { gi[c] = a.r.goal_init_pos[ig * 3 + c]; ep[c] = a.r.env_pos[ig * 3 + c]; }
    }
    float u[4];
    hand_draw<4>(a.r, ig, 0u, u);
    hand_goal_write(a, ig, (int64_t)k0g + tg, u[0], u[1], gi, ep, g_keeps_rot, a.r.goal_world_pos_out, a.r.goal_rot_out);
    a.r.goal_ids[k0g + tg] = ig;
  }
  if (do_rg) {  // reset_idx's reset_target_pose (:401)
    float gi[3], ep[3];
    if (role == 2) {
#pragma unroll
      for (int c = 0; c < 3; ++c) { gi[c] = row_a[c]; ep[c] = row_b[c]; }
    } else {
#pragma unroll
      for (int c = 0; c < 3; ++c) { gi[c] = a.r.goal_init_pos[irg * 3 + c]; ep[c] = a.r.env_pos[irg * 3 + c]; }
    }
    float ug[4];
    hand_draw<4>(a.r, irg, 2u, ug);
    hand_goal_write(a, irg, (int64_t)k0r + trg, ug[0], ug[1], gi, ep, true, a.r.r_goal_world_pos_out, a.r.r_goal_rot_out);
  }
}

template <int D>
__global__ __launch_bounds__(256) void hand_reset_idx_kernel(const HandPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.count) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  hand_reset_env<D>(a, i, k);
}

__global__ __launch_bounds__(256) void hand_reset_goal_kernel(const HandPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= max_ids || k >= *a.r.goal_count) return;
  const int64_t i = a.r.goal_ids[k];
  if (i < 0 || i >= a.n) return;
  hand_reset_goal(a, i, k, a.r.u_goal, 0u, a.r.goal_world_pos_out, a.r.goal_rot_out);
}

// ---------------------------------------------------------------- host dispatch
static bool is_allegro(const oige_hand_params* p) {
  return p->obs_type == OIGE_HAND_OBS_ALLEGRO_FULL || p->obs_type == OIGE_HAND_OBS_ALLEGRO_FULL_NO_VEL;
}

static int check_hand(const oige_hand_params* p, const oige_hand_buffers* b, int64_t n) {
  OIGE_REQUIRE(p && b, "hand: null params/buffers");
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 256, "hand: num_envs %lld out of range", (long long)n);
  OIGE_REQUIRE(p->obs_type == OIGE_HAND_OBS_OPENAI || p->obs_type == OIGE_HAND_OBS_FULL || is_allegro(p),
               "hand: observationType must be openai (0) / full (2) for ShadowHand or allegro full_no_vel (3) / full (4); "
               "ShadowHand full_no_vel raises a shape error in the reference (shadow_hand.py:190) and non-asymmetric "
               "full_state is dead code there (shadow_hand.py:263-306)");
  if (is_allegro(p)) {
    OIGE_REQUIRE(!p->asymmetric_obs, "hand: AllegroHand has no states (allegro_hand.py:71)");
    for (int k = 0; k < OIGE_ALLEGRO_DOFS; ++k)
      OIGE_REQUIRE(p->actuated_dof_indices[k] == k, "hand: AllegroHand actuates all 16 DOFs in order (allegro_hand_view.py:53)");
    return OIGE_OK;
  }
  for (int k = 0; k < OIGE_HAND_ACTIONS; ++k)
    OIGE_REQUIRE(p->actuated_dof_indices[k] >= 0 && p->actuated_dof_indices[k] < OIGE_HAND_DOFS &&
                     (k == 0 || p->actuated_dof_indices[k] > p->actuated_dof_indices[k - 1]),
                 "hand: actuated_dof_indices must be 20 ascending indices below 24");
  return OIGE_OK;
}

template <int OBS, bool ASYM>
static int launch_hand_post(const HandPost& a, hipStream_t s) {
  constexpr int OW = OBS == OIGE_HAND_OBS_OPENAI ? 42 : 157;
  const size_t lds = sizeof(float) * (HPE * OW + (ASYM ? HPE * HS : 0) + HPE * 3 + 2 * HD + HPE + HPE + 4);
  static bool configured = false;
  if (!configured) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&hand_post_kernel<OBS, ASYM>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return fail(OIGE_ELAUNCH, "hand_post_kernel: cannot reserve %zu B of LDS: %s", lds, hipGetErrorString(e));
    configured = true;
  }
  hipLaunchKernelGGL((hand_post_kernel<OBS, ASYM>), dim3((unsigned)num_tiles(a.n)), dim3(HNT), lds, s, a);
  return check_launch("hand_post_kernel");
}

static int finalize_hand(const HandPost& a, hipStream_t s) {
  if (!(a.stages & OIGE_STAGE_METRICS)) return OIGE_OK;
  const float av = a.p.av_factor;  // reaches the jit function as float(fp32 tensor) (:59,:293)
  const bool scans = a.b.reset_workspace && a.b.goal_workspace;
  hipLaunchKernelGGL(hand_finalize_kernel, dim3(scans ? 3 : 1), dim3(1024), 0, s, a.b.reduce_workspace, num_tiles(a.n),
                     a.b.consecutive_successes, av, (float)(1.0 - (double)av), a.b.reset_workspace, a.b.goal_workspace,
                     num_groups(a.n), padded_tiles(a.n));
  return check_launch("hand_finalize_kernel");
}

}  // namespace oige

using namespace oige;

extern "C" {

static int hand_post_physics(const oige_hand_params* params, const oige_hand_state* state, const oige_hand_buffers* buffers,
                             const oige_noise_args* noise, int64_t num_envs, uint32_t stages, oige_stream_t stream) {
  if (int rc = check_hand(params, buffers, num_envs)) return rc;
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "hand_post_physics: bad stages mask 0x%x", stages);
  stages &= ~OIGE_STAGE_DONE;  // is_done is `pass` in the reference (:374-375)
  if (stages == 0 || num_envs == 0) return OIGE_OK;
  OIGE_REQUIRE(state != nullptr, "hand_post_physics: null state");
  const oige_hand_buffers& b = *buffers;
  const oige_hand_state& st = *state;
  const bool asym = params->asymmetric_obs != 0;
  OIGE_REQUIRE(st.object_pos && st.object_rot && st.env_pos && b.goal_pos && b.goal_rot && b.actions && b.progress_buf,
               "hand_post_physics: null object/goal/actions/progress pointer");
  OIGE_REQUIRE(aligned16(st.object_rot) && aligned16(b.goal_rot) && aligned16(b.actions),
               "hand_post_physics: object_rot/goal_rot/actions must be 16-byte aligned");
  if (stages & OIGE_STAGE_METRICS)
    OIGE_REQUIRE(b.rew_buf && b.reset_buf && b.reset_goal_buf && b.successes && b.consecutive_successes &&
                     b.reset_workspace && b.goal_workspace && b.reduce_workspace,
                 "hand_post_physics: null reward/reset/successes/workspace pointer");
  HandPost a;
  a.p = *params;
  a.s = st;
  a.b = b;
  a.n = num_envs;
  a.stages = stages;
  a.max_len_minus_1 = (float)((double)params->max_episode_length - 1.0);
  a.half_fall_penalty = (float)(0.5 * (double)params->fall_penalty);
  a.dr = 0;
  if (noise) {
    OIGE_REQUIRE(!is_allegro(params), "hand_post_physics_dr: the AllegroHand task configures no observation noise (AllegroHand.yaml)");
    OIGE_REQUIRE((stages & (OIGE_STAGE_OBS | OIGE_STAGE_METRICS)) == (OIGE_STAGE_OBS | OIGE_STAGE_METRICS),
                 "hand_post_physics_dr: the noise needs this step's observations and reset flags (stages OBS | METRICS)");
    if (int rc = noise_validate_fused(noise, b.obs_buf, params->obs_type == OIGE_HAND_OBS_OPENAI ? 42 : 157, "hand_post_physics_dr"))
      return rc;
    a.dn = *noise;
    a.dr = 1;
  }
  hipStream_t s = (hipStream_t)stream;
  int rc;
  if (is_allegro(params)) {
    if (stages & OIGE_STAGE_OBS) {
      OIGE_REQUIRE(b.obs_buf && aligned16(b.obs_buf), "hand_post_physics: obs_buf null or unaligned");
      OIGE_REQUIRE(st.hand_dof_pos && aligned16(st.hand_dof_pos), "hand_post_physics: null/unaligned hand_dof_pos");
      if (params->obs_type == OIGE_HAND_OBS_ALLEGRO_FULL)
        OIGE_REQUIRE(st.hand_dof_vel && aligned16(st.hand_dof_vel) && st.object_velocities &&
                         (reinterpret_cast<uintptr_t>(st.object_velocities) & 7u) == 0,
                     "hand_post_physics: null/unaligned hand_dof_vel or object_velocities");
    }
    const unsigned tiles = (unsigned)num_tiles(num_envs);
    if (params->obs_type == OIGE_HAND_OBS_ALLEGRO_FULL)
      hipLaunchKernelGGL((allegro_post_kernel<OIGE_HAND_OBS_ALLEGRO_FULL>), dim3(tiles), dim3(WAVE), 0, s, a);
    else
      hipLaunchKernelGGL((allegro_post_kernel<OIGE_HAND_OBS_ALLEGRO_FULL_NO_VEL>), dim3(tiles), dim3(WAVE), 0, s, a);
    rc = check_launch("allegro_post_kernel");
    if (rc) return rc;
    return finalize_hand(a, s);
  }
  if (stages & OIGE_STAGE_OBS) {
    OIGE_REQUIRE(b.obs_buf && aligned16(b.obs_buf), "hand_post_physics: obs_buf null or unaligned");
    OIGE_REQUIRE(st.object_velocities && st.fingertip_pos && aligned16(st.fingertip_pos) &&
                     (reinterpret_cast<uintptr_t>(st.object_velocities) & 7u) == 0,
                 "hand_post_physics: null/unaligned object_velocities or fingertip_pos");
    if (asym || params->obs_type == OIGE_HAND_OBS_FULL)
      OIGE_REQUIRE(st.fingertip_rot && st.fingertip_velocities && st.hand_dof_pos && st.hand_dof_vel &&
                       aligned16(st.fingertip_rot) && aligned16(st.fingertip_velocities) && aligned16(st.hand_dof_pos) &&
                       aligned16(st.hand_dof_vel),
                   "hand_post_physics: null/unaligned fingertip or hand dof state");
    if (asym)
      OIGE_REQUIRE(b.states_buf && st.vec_sensor && aligned16(b.states_buf) && aligned16(st.vec_sensor),
                   "hand_post_physics: asymmetric_obs needs states_buf and vec_sensor (16-byte aligned)");
  }
  if (params->obs_type == OIGE_HAND_OBS_OPENAI)
    rc = asym ? launch_hand_post<OIGE_HAND_OBS_OPENAI, true>(a, s) : launch_hand_post<OIGE_HAND_OBS_OPENAI, false>(a, s);
  else
    rc = asym ? launch_hand_post<OIGE_HAND_OBS_FULL, true>(a, s) : launch_hand_post<OIGE_HAND_OBS_FULL, false>(a, s);
  if (rc) return rc;
  return finalize_hand(a, s);
}

static int fill_pre(HandPre& a, const oige_hand_params* params, const oige_hand_buffers* buffers,
                    const oige_hand_reset_io* r, int64_t n) {
  if (int rc = check_hand(params, buffers, n)) return rc;
  OIGE_REQUIRE(r != nullptr, "hand: null reset_io");
  OIGE_REQUIRE(r->object_init_pos && r->goal_init_pos && r->env_pos, "hand: null init pointer");
  const oige_hand_buffers& b = *buffers;
  OIGE_REQUIRE(b.goal_pos && b.goal_rot && b.reset_buf && b.reset_goal_buf && b.progress_buf && b.successes &&
                   b.prev_targets && b.cur_targets,
               "hand: null goal/reset/progress/successes/targets pointer");
  OIGE_REQUIRE(aligned16(b.goal_rot), "hand: goal_rot must be 16-byte aligned");
  a.p = *params;
  a.b = b;
  a.r = *r;
  a.n = n;
  a.flags = 0;
  a.dr = 0;
  a.actions_in = nullptr;
  a.actions_out = nullptr;
  a.targets_out = nullptr;
  return OIGE_OK;
}

static int hand_pre_physics(const oige_hand_params* params, const float* actions_in, const oige_hand_buffers* buffers,
                            const oige_hand_reset_io* reset_io, const oige_noise_args* noise, float* actions_out,
                            float* targets_out, int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  HandPre a;
  if (int rc = fill_pre(a, params, buffers, reset_io, num_envs)) return rc;
  const oige_hand_reset_io& r = *reset_io;
  OIGE_REQUIRE(actions_in && targets_out && aligned16(actions_in) && aligned16(targets_out) && aligned16(actions_out),
               "hand_pre_physics: null/unaligned actions or targets");
  OIGE_REQUIRE(buffers->reset_workspace && buffers->goal_workspace && aligned16(buffers->reset_workspace) &&
                   aligned16(buffers->goal_workspace),
               "hand_pre_physics: null/unaligned workspaces");
  OIGE_REQUIRE(r.goal_ids && r.goal_count && r.goal_world_pos_out && r.goal_rot_out && r.env_ids && r.count &&
                   r.r_goal_world_pos_out && r.r_goal_rot_out && r.object_pos_out && r.object_rot_out && r.object_vel_out &&
                   r.dof_pos_out && r.dof_vel_out,
               "hand_pre_physics: null reset output pointer");
  OIGE_REQUIRE(aligned16(r.goal_rot_out) && aligned16(r.r_goal_rot_out) && aligned16(r.object_rot_out),
               "hand_pre_physics: (n,4) outputs must be 16-byte aligned");
  OIGE_REQUIRE(aligned16(a.b.prev_targets) && aligned16(a.b.cur_targets) && aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out),
               "hand_pre_physics: prev/cur targets and dof row outputs must be 16-byte aligned");
  OIGE_REQUIRE(!r.u_goal && !r.u_reset && !r.u_reset_goal,
               "hand_pre_physics: pre-drawn noise goes through oige_hand_reset_idx / oige_hand_reset_target_pose");
  a.actions_in = actions_in;
  a.actions_out = actions_out;
  a.targets_out = targets_out;
  a.flags = flags;
  if (noise) {
    if (int rc = noise_validate_fused(noise, actions_in, is_allegro(params) ? OIGE_ALLEGRO_DOFS : OIGE_HAND_ACTIONS,
                                      "hand_pre_physics_dr"))
      return rc;
    OIGE_REQUIRE(!noise->clamped_out, "hand_pre_physics_dr: the clamped actions are actions_out (OIGE_PRE_CLAMP_ACTIONS)");
    a.dn = *noise;
    a.dr = 1;
  }
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID)) {
    if (int rc = launch_count_flags(a.b.reset_buf, num_envs, a.b.reset_workspace, s)) return rc;
    if (int rc = launch_count_flags(a.b.reset_goal_buf, num_envs, a.b.goal_workspace, s)) return rc;
  }
  if ((flags & (OIGE_PRE_OFFSETS_VALID | OIGE_PRE_COUNTS_VALID)) != (OIGE_PRE_OFFSETS_VALID | OIGE_PRE_COUNTS_VALID))  // (both set: the post launch sequence already ran the scans, hand_finalize_kernel)
    if (int rc = launch_scan_groups2(num_envs, a.b.reset_workspace, a.b.goal_workspace, s)) return rc;
  const dim3 grid((unsigned)G);
  if (is_allegro(params)) {
    if (a.dr) hipLaunchKernelGGL((hand_pre_kernel<OIGE_ALLEGRO_DOFS, OIGE_ALLEGRO_DOFS, true, GROUP>), grid, dim3(GROUP), 0, s, a);
    else hipLaunchKernelGGL((hand_pre_kernel<OIGE_ALLEGRO_DOFS, OIGE_ALLEGRO_DOFS, false, GROUP>), grid, dim3(GROUP), 0, s, a);
  } else {
    if (a.dr) hipLaunchKernelGGL((hand_pre_kernel<OIGE_HAND_DOFS, OIGE_HAND_ACTIONS, true, OIGE_HAND_PRE_NT>), grid, dim3(OIGE_HAND_PRE_NT), 0, s, a);
    else hipLaunchKernelGGL((hand_pre_kernel<OIGE_HAND_DOFS, OIGE_HAND_ACTIONS, false, GROUP>), grid, dim3(GROUP), 0, s, a);
  }
  return check_launch("hand_pre_kernel");
}

int oige_hand_post_physics(const oige_hand_params* params, const oige_hand_state* state,
                           const oige_hand_buffers* buffers, int64_t num_envs, uint32_t stages, oige_stream_t stream) {
  return hand_post_physics(params, state, buffers, nullptr, num_envs, stages, stream);
}

int oige_hand_post_physics_dr(const oige_hand_params* params, const oige_hand_state* state,
                              const oige_hand_buffers* buffers, const oige_noise_args* obs_noise, int64_t num_envs,
                              uint32_t stages, oige_stream_t stream) {
  OIGE_REQUIRE(obs_noise, "hand_post_physics_dr: null noise args");
  return hand_post_physics(params, state, buffers, obs_noise, num_envs, stages, stream);
}

int oige_hand_pre_physics(const oige_hand_params* params, const float* actions_in, const oige_hand_buffers* buffers,
                          const oige_hand_reset_io* reset_io, float* actions_out, float* targets_out, int64_t num_envs,
                          uint32_t flags, oige_stream_t stream) {
  return hand_pre_physics(params, actions_in, buffers, reset_io, nullptr, actions_out, targets_out, num_envs, flags, stream);
}

int oige_hand_pre_physics_dr(const oige_hand_params* params, float* actions_inout, const oige_hand_buffers* buffers,
                             const oige_hand_reset_io* reset_io, const oige_noise_args* action_noise, float* actions_out,
                             float* targets_out, int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  OIGE_REQUIRE(action_noise, "hand_pre_physics_dr: null noise args");
  return hand_pre_physics(params, actions_inout, buffers, reset_io, action_noise, actions_out, targets_out, num_envs, flags,
                          stream);
}

int oige_hand_reset_idx(const oige_hand_params* params, const oige_hand_buffers* buffers,
                        const oige_hand_reset_io* reset_io, int64_t num_envs, int64_t max_ids, oige_stream_t stream) {
  HandPre a;
  if (int rc = fill_pre(a, params, buffers, reset_io, num_envs)) return rc;
  const oige_hand_reset_io& r = *reset_io;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "hand_reset_idx: bad max_ids");
  OIGE_REQUIRE(r.env_ids && r.count && r.r_goal_world_pos_out && r.r_goal_rot_out && r.object_pos_out && r.object_rot_out &&
                   r.object_vel_out && r.dof_pos_out && r.dof_vel_out,
               "hand_reset_idx: null reset pointer");
  OIGE_REQUIRE((r.u_reset == nullptr) == (r.u_reset_goal == nullptr), "hand_reset_idx: pass both u_reset and u_reset_goal or neither");
  OIGE_REQUIRE(aligned16(r.r_goal_rot_out) && aligned16(r.object_rot_out), "hand_reset_idx: (n,4) outputs must be 16-byte aligned");
  if (max_ids == 0) return OIGE_OK;
  const dim3 grid((unsigned)((max_ids + 255) / 256));
  if (is_allegro(params))
    hipLaunchKernelGGL(hand_reset_idx_kernel<OIGE_ALLEGRO_DOFS>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  else
    hipLaunchKernelGGL(hand_reset_idx_kernel<OIGE_HAND_DOFS>, grid, dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("hand_reset_idx_kernel");
}

int oige_hand_reset_target_pose(const oige_hand_params* params, const oige_hand_buffers* buffers,
                                const oige_hand_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                                oige_stream_t stream) {
  HandPre a;
  if (int rc = fill_pre(a, params, buffers, reset_io, num_envs)) return rc;
  const oige_hand_reset_io& r = *reset_io;
  OIGE_REQUIRE(max_ids >= 0 && max_ids <= num_envs, "hand_reset_target_pose: bad max_ids");
  OIGE_REQUIRE(r.goal_ids && r.goal_count && r.goal_world_pos_out && r.goal_rot_out && aligned16(r.goal_rot_out),
               "hand_reset_target_pose: null/unaligned goal pointer");
  if (max_ids == 0) return OIGE_OK;
  hipLaunchKernelGGL(hand_reset_goal_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0, (hipStream_t)stream, a, max_ids);
  return check_launch("hand_reset_goal_kernel");
}

}  // extern "C"
