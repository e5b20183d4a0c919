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
#include "oige_scan.hpp"

namespace oige {

constexpr int AD = OIGE_ANYMAL_DOFS;          // 12
constexpr int AO = OIGE_ANYMAL_OBS;           // 188
constexpr int AP = OIGE_ANYMAL_SCAN_POINTS;   // 140
constexpr int AE = OIGE_ANYMAL_EPISODE_SUMS;  // 13
constexpr int RW = 16;  // reduce_workspace floats per tile

struct AnymalPost {
  oige_anymal_params p;
  oige_anymal_state s;
  oige_anymal_buffers b;
  oige_anymal_reset_io r;
  int64_t n;
  int mode;  // 0: post_physics_step, 1: standalone reset_idx(env_ids) driven by r.reset_mask
  // 1: no scan / norm launch between the flags and the head kernel.  The flags kernel leaves the tile counts as bytes too
  // (b.tile_counts_u8) and one partial sum of the command norm per GROUP in the workspace's group-offset slots; a head wave
  // that holds a reset env sums both itself, and the last tile's wave leaves the total where the scan would have.
  int unscanned;
  // individually callable stages (oige_anymal_post_physics_stages): 0 = the fused step, else a mask of OIGE_ANYMAL_STAGE_*
  uint32_t stages;
  oige_anymal_frame x;  // base_lin_vel / base_ang_vel / projected_gravity as tensors of their own (optional in the fused step)
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

// ------------------------------------------------------------------ min(h[r, c], h[r + 1, c + 1]) per map cell
__global__ __launch_bounds__(256) void anymal_height_min_kernel(const int16_t* __restrict__ h, int rows, int cols,
                                                                int16_t* __restrict__ out) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (int64_t)rows * cols) return;
  const int r = (int)(i / cols), c = (int)(i - (int64_t)r * cols);
  const int16_t v = h[i];
  out[i] = (r + 1 < rows && c + 1 < cols) ? (v < h[i + cols + 1] ? v : h[i + cols + 1]) : v;
}

// ------------------------------------------------------------------ PD torques (:398-403)
// actions_out (optional): `self.actions = actions.clone()` (:390) folded into the first sub-step's launch
__global__ __launch_bounds__(256) void anymal_torques_kernel(const oige_anymal_params p, const float* __restrict__ actions,
                                                             const float* __restrict__ dof_pos,
                                                             const float* __restrict__ dof_vel,
                                                             float* __restrict__ torques, float* __restrict__ actions_out,
                                                             int64_t total) {
  // the twelve default positions through LDS: picked per lane straight from the argument block (`g3 == 0 ? p.x[k] : ...`) the
  // compiler folds the selects into ONE run-time-indexed load per element -- four vector loads of the argument block, each
  // with a full wait behind it, in front of the stores (r03: 12.4 -> 12.0 us)
  __shared__ __align__(16) float s_def[AD];
  {
    float dj = 0.f;
#pragma unroll
    for (int j = 0; j < AD; ++j) dj = (int)threadIdx.x == j ? p.default_dof_pos[j] : dj;
    if (threadIdx.x < AD) s_def[threadIdx.x] = dj;
  }
  const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t e = 4 * f;  // total = N*12 is a multiple of 4 and a float4 never straddles a row
  const bool live = e < total;
  const int64_t el = live ? e : 0;  // (idle lanes re-read the first quad; nothing conditional in front of the barrier)
  const float4 a = *reinterpret_cast<const float4*>(actions + el);
  const float4 q = *reinterpret_cast<const float4*>(dof_pos + el);
  const float4 v = *reinterpret_cast<const float4*>(dof_vel + el);
  __syncthreads();
  if (!live) return;
  const int g3 = (int)(f % 3);  // which third of the 12-wide row
  const float4 d4 = *reinterpret_cast<const float4*>(s_def + 4 * g3);
  const float av[4] = {a.x, a.y, a.z, a.w}, qv[4] = {q.x, q.y, q.z, q.w}, vv[4] = {v.x, v.y, v.z, v.w};
  const float dv[4] = {d4.x, d4.y, d4.z, d4.w};
  float o[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const float t = p.Kp * ((p.action_scale * av[k] + dv[k]) - qv[k]) - p.Kd * vv[k];
    o[k] = (t != t) ? t : (t < -80.0f ? -80.0f : (t > 80.0f ? 80.0f : t));
  }
  *reinterpret_cast<float4*>(torques + e) = make_float4(o[0], o[1], o[2], o[3]);
  if (actions_out) *reinterpret_cast<float4*>(actions_out + e) = a;
}

// ------------------------------------------------------------------ flags (:410, :452-466)
__global__ __launch_bounds__(GROUP) void anymal_flags_kernel(const AnymalPost a) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool live = env < a.n;
  bool reset = false;
  float cmd2 = 0.0f;
  if (a.mode == 0 && a.r.push_velocities && a.r.push_interval > 0) {  // push_robots (:419-420, :446-450)
    const uint64_t c = step_of(a.r) + (uint64_t)a.r.push_offset;       // common_step_counter
    const bool pushing = c % (uint64_t)a.r.push_interval == 0;          // uniform
    if (pushing && live) {
      const u4 cc{(uint32_t)env, 7u << 16, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
      const u4 x = philox4x32_10(cc, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
      // torch_rand_float(-1, 1, (N, 2)): (1 - (-1)) * u + (-1), columns 0:2 of the (N,6) velocities
      *reinterpret_cast<float2*>(a.r.push_velocities + env * 6) = make_float2(2.0f * u01(x.x) + (-1.0f), 2.0f * u01(x.y) + (-1.0f));
    }
    if (env == 0) *a.r.push_count = pushing ? a.n : 0;
  }
  const bool st_progress = a.stages == 0 || (a.stages & OIGE_ANYMAL_STAGE_PROGRESS);
  const bool st_term = a.stages == 0 || (a.stages & OIGE_ANYMAL_STAGE_TERMINATION);
  if (live && a.mode == 0 && !st_term) {  // progress += 1 on its own (:410)
    if (st_progress) a.b.progress_buf[env] += 1;
  } else if (live && a.mode == 0) {
    const int64_t progress = a.b.progress_buf[env] + (st_progress ? 1 : 0);  // :410
    if (st_progress) a.b.progress_buf[env] = progress;
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
  if (a.mode == 0 && !st_term) return;  // (uniform) nothing was flagged: the counts stay what they are
  const unsigned long long m = __ballot(reset);
  const float s = wave_sum_ff(cmd2);
  if (lane_id() == 0) {
    const int64_t t = env / TILE;
    a.b.reset_workspace[t] = __popcll(m);
    if (t < num_tiles(a.n)) a.b.reduce_workspace[t * RW + 0] = s;
  }
  if (a.unscanned) {  // (uniform)
    __shared__ float s_cmd2[GROUP_TILES];
    if (lane_id() == 0) {
      a.b.tile_counts_u8[env / TILE] = (uint8_t)__popcll(m);  // every tile slot of the padded table, zeros past the last env
      s_cmd2[threadIdx.x >> 6] = s;
    }
    __syncthreads();
    if (threadIdx.x == 0)
      reinterpret_cast<float*>(a.b.reset_workspace + padded_tiles(a.n))[blockIdx.x] = (s_cmd2[0] + s_cmd2[1]) + (s_cmd2[2] + s_cmd2[3]);
  }
}

// Two one-workgroup reductions over what the flags kernel left, in one launch: workgroup 0 scans the per-group reset counts
// (oige_scan.hpp), workgroup 1 takes the whole-matrix norm of the reset envs' xy commands (:376): sqrt of the sum over
// tiles, tile order.
__global__ __launch_bounds__(SCAN_THREADS) void anymal_scan_norm_kernel(int32_t* __restrict__ reset_ws, int64_t n_groups,
                                                                        int64_t padded, float* __restrict__ ws, int64_t tiles) {
  __shared__ int wave_tot[SCAN_THREADS / WAVE];
  __shared__ double part[SCAN_THREADS / WAVE];
  if (blockIdx.x == 0) {
    scan_groups_body(reset_ws, n_groups, tiles, reset_ws + padded, wave_tot);
    return;
  }
  double acc = 0.0;
  for (int64_t i = threadIdx.x; i < tiles; i += SCAN_THREADS) acc += (double)ws[i * RW];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < SCAN_THREADS / WAVE; ++i) t += part[i];
    ws[tiles * RW + 0] = __builtin_sqrtf((float)t);
  }
}

// ------------------------------------------------------------------ main
// LDS row of one env in the head kernel: 36 floats = 9 float4 (odd: per-lane 16-byte row accesses are bank-conflict
// free).  Columns 12:36 take dof_pos / dof_vel (raw, then scaled: observation columns 12:36); columns 0:12 first hold
// the env's actions (read by the action-rate sum and the last_actions copy) and are then overwritten by the 12 head
// columns of the observation, so the row ends up as observation columns 0:36.  The action columns of the observation
// (176:188) are re-read from `actions` in the epilogue (L2-hot).  r01 kept all 48 columns (+ 4 of padding): 13.5 KB of LDS
// per one-wave workgroup = 11 waves per CU, so the 4096 tiles of 256 K envs ran as 1.45 rounds; 9.4 KB fits all 16
// waves the 124 VGPRs allow, i.e. every tile is resident at once.
// last_actions, last_dof_vel and torques only feed three sums: each lane reads its own 48-byte rows of those straight
// from global.  The 140 height-scan columns never touch this tile: the scan kernel writes them.
constexpr int ARW = 36;
constexpr int L_ACT = 0, L_DPOS = 12, L_DVEL = 24;
constexpr int AP4 = AP / 4;  // 35 float4 of scan columns per env
// Pose record the head kernel leaves per env for the scan kernel, after the RW * tiles + 4 reduction floats of
// reduce_workspace: (yaw quat w, z, base x, y), (base z - 0.5, -, -, -).  (Routing the 48 head / action columns through
// the record too, so that the scan kernel writes whole 752-byte rows, was measured slower: 82 + 121 us against 79 + 99.)
constexpr int APOSE = 8;
__host__ __device__ inline int64_t pose_offset(int64_t tiles) { return tiles * RW + 4; }

// RN(x / h) from inv = RN(1 / h): q0 = x inv, then two fma residual corrections.  The second one starts from a
// faithful quotient and a correctly rounded reciprocal, which makes its result the correctly rounded quotient
// (Markstein), i.e. bit-identical to the IEEE division torch performs (:586-593) for every finite in-range x;
// tests/test_helpers_known_answers.py replays the sequence against true division.
__device__ __forceinline__ float div_by_const(float x, float h, float inv) {
  const float q0 = x * inv;
  const float q1 = __builtin_fmaf(__builtin_fmaf(-q0, h, x), inv, q0);
  return __builtin_fmaf(__builtin_fmaf(-q1, h, x), inv, q1);
}

__device__ __forceinline__ float4 lds4(const float* p) { return *reinterpret_cast<const float4*>(p); }
__device__ __forceinline__ void st4_plain(float* p, const float4 v) { *reinterpret_cast<float4*>(p) = v; }
#ifdef OIGE_ANYMAL_HEAD_NT
#define ANYMAL_HEAD_ST4(p, v) st4_stream((p), (v))
#else
#define ANYMAL_HEAD_ST4(p, v) st4_plain((p), (v))
#endif
#ifdef OIGE_ANYMAL_SCAN_PLAIN
#define ANYMAL_SCAN_ST4(p, v) st4_plain((p), (v))
#else
#define ANYMAL_SCAN_ST4(p, v) st4_stream((p), (v))
#endif

// uniform noise of four consecutive observation columns starting at flat float offset `off` (:438-439)
// NOISE_IN: the uniforms come from the caller's tensor (the parity tests' replay path).  A compile-time split: as a
// run-time test the load sat in a conditional block and its join waited for EVERY memory operation in flight, the
// previous pass's observation stores included -- one HBM round trip per float4 of noise, whichever way the test went.
// `gf` = off / 4, the flat float4 index in obs_buf: below 2^31 for every accepted num_envs (< 2^23 envs x 47 float4), so the
// counter's second word is the stream tag alone and the index arithmetic stays 32-bit (64-bit multiplies are quarter rate)
template <bool NOISE_IN>
__device__ __forceinline__ void anymal_noise4(const oige_anymal_reset_io& r, uint64_t step, uint32_t gf, float (&uu)[4]) {
  if (NOISE_IN) {
    const float4 t4 = *reinterpret_cast<const float4*>(r.noise_u + 4 * (int64_t)gf);
    uu[0] = t4.x; uu[1] = t4.y; uu[2] = t4.z; uu[3] = t4.w;
  } else {
    const u4 cc{gf, 3u << 16, (uint32_t)step, (uint32_t)(step >> 32)};
    const u4 x = philox4x32_10(cc, (uint32_t)r.seed, (uint32_t)(r.seed >> 32));
    uu[0] = u01(x.x); uu[1] = u01(x.y); uu[2] = u01(x.z); uu[3] = u01(x.w);
  }
}

// SPLIT: a.stages selects FRAME / METRICS / OBS (the individually callable methods of the reference; never the in-step reset,
// which is reset_idx(env_ids) = mode 1).  The fused step is its own instantiation: every test below folds away in it.
template <bool FULLTILE, bool NOISE_IN, bool SPLIT = false>
__device__ __forceinline__ void anymal_head_tile(const AnymalPost& a, float* __restrict__ lds, int64_t t, int nv) {
  float* rows = lds;                         // 64 x 52
  float* s_ns = lds + TILE * ARW;            // noise_scale_vec (:157-171) of observation columns 0:36 (48 slots kept)
  const oige_anymal_params& p = a.p;
  const int tid = threadIdx.x;               // one wavefront per tile
  const uint64_t noise_step = step_of(a.r);  // read once: by value, or the device counter of a replayed graph
  const int64_t env0 = t * TILE;
  const int64_t T = num_tiles(a.n);
  float* pose_out = a.b.reduce_workspace + pose_offset(T) + env0 * APOSE;

  const bool step = a.mode == 0;
  const bool do_frame = !SPLIT || (a.stages & OIGE_ANYMAL_STAGE_FRAME);
  const bool do_metrics = !SPLIT || (a.stages & OIGE_ANYMAL_STAGE_METRICS);
  const bool do_obs = !SPLIT || (a.stages & OIGE_ANYMAL_STAGE_OBS);
  constexpr bool do_reset = !SPLIT;
  // The tile's 16 waves per CU are all resident at once (4096 tiles at 256 K envs = the whole grid), so the kernel lasts as
  // long as ONE wave's chain of memory round trips.  r02's chain had nine links (chunks | scalars, rows, sums | two passes over
  // each of the two prefix tables | terrain type and origin, behind the reset path's first stores | the new origin, behind more
  // of them | the actions again for the epilogue, behind all of them | the epilogue's stores); r03's has four:
  //   trip 0 + 1  pass 0 of the prefix tables, the three 12-wide chunks and the env's own scalars, issued together, nothing
  //               conditional (dead lanes shadow env0); the tables are summed while the rest is in flight;
  //   trip 2      behind the frame arithmetic: the env's 12-wide rows, episode sums, terrain level / type / origin;
  //   trip 3      the new terrain origin of the reset envs (it depends on the level, which depends on the norm), in front of
  //               the Philox draws and of every store of the reset path;
  // every store behind the last load, except the three action-column stores, which leave in front of trip 2 (a load issued
  // behind a store waits for that store's acknowledgement too -- one in-order counter -- and these are acknowledged by then).
  const bool live = FULLTILE || tid < nv;
  const int64_t env = live ? env0 + tid : env0;  // dead lanes shadow env0, never store
  float* row = rows + tid * ARW;
  // ---- trip 0: pass 0 of the two tables the reset path sums (tile-count bytes of the groups in front of this one, per-group
  // partial sums of the command norm: what the scan / norm launch would have left; L2 hits, written by the flags kernel).
  // First of all, when no register is taken yet; summed while trip 1 is in flight.  Two 16-byte chunks of each per lane
  // cover 512 groups; the rest, if any, in the loop below.
  constexpr int PU = 2;
  const int64_t grp = t / GROUP_TILES, NG = num_groups(a.n);
  const bool own_prefix = do_reset && a.unscanned;  // (uniform)
  const int64_t nq_u = (grp + 3) / 4, nq_g = NG / 4;  // 16-byte chunks of the byte table in front of the group / of the partial sums
  const float* gp = reinterpret_cast<const float*>(a.b.reset_workspace + padded_tiles(a.n));  // one partial sum per group
  uint4 pu[PU];
  float4 pg[PU];
  float gtail = 0.f;
  if (own_prefix) {
    const uint4* c4 = reinterpret_cast<const uint4*>(a.b.tile_counts_u8);
    // (fewer than four groups: no whole chunk of sums; the idle slots then re-read the byte table, always 16 bytes at least)
    const float4* gp4 = nq_g > 0 ? reinterpret_cast<const float4*>(gp) : reinterpret_cast<const float4*>(a.b.tile_counts_u8);
#pragma unroll
    for (int u = 0; u < PU; ++u) {  // idle slots re-read a valid chunk
      const int64_t j = lane_id() + u * WAVE;
      pu[u] = c4[j < nq_u ? j : (nq_u > 0 ? nq_u - 1 : 0)];
      pg[u] = gp4[j < nq_g ? j : (nq_g > 0 ? nq_g - 1 : 0)];
    }
    const int64_t jt = 4 * nq_g + lane_id();
    gtail = gp[jt < NG ? jt : 0];
  }

  // ---- trip 1
  Chunk<AD> c0, c1, c2;
  if (step) {
    load_chunk<AD, FULLTILE>(c0, a.s.dof_pos + env0 * AD, nv * AD);
    load_chunk<AD, FULLTILE>(c1, a.s.dof_vel + env0 * AD, nv * AD);
    load_chunk<AD, FULLTILE>(c2, a.b.actions + env0 * AD, nv * AD);
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
  const int flag_ld = step ? (int)a.b.reset_buf[env] : (int)a.r.reset_mask[env];
  const int fallen_ld = step ? (int)a.b.has_fallen[env] : 0;
  const int64_t timeout = step ? a.b.timeout_buf[env] : 0;
  int64_t k_front = 0;   // flagged envs in front of this tile's group
  float gnorm_own = 0.f;  // whole-matrix norm of the reset envs' xy commands (:376)
  if (own_prefix) {  // (uniform) the additions in the order the scan-free path has always taken them: chunk by chunk per lane
    const int64_t g = grp;
    unsigned fr = 0;
    double acc = 0.0;
    auto add_u = [&](const uint4 c, int64_t j) {
      const int64_t w0 = 4 * j;  // first word of the chunk; words >= g belong to this group or later ones
      fr = __builtin_amdgcn_sad_u8(w0 < g ? c.x : 0u, 0u, fr);
      fr = __builtin_amdgcn_sad_u8(w0 + 1 < g ? c.y : 0u, 0u, fr);
      fr = __builtin_amdgcn_sad_u8(w0 + 2 < g ? c.z : 0u, 0u, fr);
      fr = __builtin_amdgcn_sad_u8(w0 + 3 < g ? c.w : 0u, 0u, fr);
    };
#pragma unroll
    for (int u = 0; u < PU; ++u) {
      const int64_t j = lane_id() + u * WAVE;
      add_u(pu[u], j);
      if (j < nq_g) acc += ((double)pg[u].x + (double)pg[u].y) + ((double)pg[u].z + (double)pg[u].w);
    }
    const uint4* c4 = reinterpret_cast<const uint4*>(a.b.tile_counts_u8);
    const float4* gp4 = reinterpret_cast<const float4*>(gp);
    const int64_t nq_max = nq_u > nq_g ? nq_u : nq_g;
    for (int64_t j0 = lane_id() + PU * WAVE; j0 < nq_max; j0 += PU * WAVE) {  // more than 512 groups
      uint4 cu[PU];
      float4 cg[PU];
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int64_t j = j0 + u * WAVE;
        cu[u] = c4[j < nq_u ? j : nq_u - 1];
        cg[u] = gp4[j < nq_g ? j : nq_g - 1];
      }
#pragma unroll
      for (int u = 0; u < PU; ++u) {
        const int64_t j = j0 + u * WAVE;
        if (j < nq_u) add_u(cu[u], j);
        if (j < nq_g) acc += ((double)cg[u].x + (double)cg[u].y) + ((double)cg[u].z + (double)cg[u].w);
      }
    }
    acc += 4 * nq_g + lane_id() < NG ? (double)gtail : 0.0;
    int front = (int)fr;
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) {
      front += __shfl_xor(front, off, WAVE);
      acc += __shfl_xor(acc, off, WAVE);
    }
    k_front = front;
    gnorm_own = __builtin_sqrtf((float)acc);
  }
  // ---- the three chunks -> row columns (12 % 4 == 0: a float4 never straddles rows)
  if (step) {
    scatter_chunk<AD, ARW, L_DPOS, false>(c0, rows, 1.0f);
    scatter_chunk<AD, ARW, L_DVEL, false>(c1, rows, 1.0f);
    scatter_chunk<AD, ARW, L_ACT, false>(c2, rows, 1.0f);
    if (tid < 48) {
      const int col = tid;
      s_ns[col] = col < 3 ? p.noise_scales[0] : col < 6 ? p.noise_scales[1] : col < 9 ? p.noise_scales[2]
                : col < 12 ? 0.0f : col < 24 ? p.noise_scales[3] : col < 36 ? p.noise_scales[4] : 0.0f;
    }
  }
  __syncthreads();

  // The action columns of the observation (176:188) are a copy of `actions` (their noise scale is zero, :170: obs +
  // (2u - 1) * 0 is obs, no draw): written from the chunk's registers right here (r02 read the actions again for this, behind
  // the reset path's stores).  Trip 2 is issued behind these three stores: they are acknowledged about when its loads return.
  if (step && do_obs) {
#pragma unroll
    for (int it = 0; it < Chunk<AD>::ITERS; ++it) {  // 12 % 4 == 0: a float4 of the run never straddles two envs
      const int f = it * WAVE + tid;
      const int e = f / 3, c = f - 3 * e;
      if (!FULLTILE && e >= nv) continue;
      ANYMAL_HEAD_ST4(a.b.obs_buf + (env0 + e) * AO + 176 + 4 * c, c2.v[it]);
    }
  }
  // ---- C: per-env math, one env per lane
  {
    const bool flagged = live && flag_ld != 0;  // reset_buf, as the reward reads it
    const bool reset = do_reset && flagged;     // ... and as the in-step reset does
    const bool has_fallen = step && fallen_ld != 0;

    // frame quantities (:422-427)
    f3 lin, ang, grav;
    if (do_frame) {
      lin = quat_rotate_inverse(bq, f3{v0.x, v0.y, v1.x});
      ang = quat_rotate_inverse(bq, f3{v1.y, v2.x, v2.y});
      grav = quat_rotate_inverse(bq, f3{0.0f, 0.0f, -1.0f});
      const f3 fwd = quat_apply(bq, f3{1.0f, 0.0f, 0.0f});
      const float heading = atan2f(fwd.y, fwd.x);
      const float w = 0.5f * wrap_to_pi(cmd[3] - heading);
      if (step) cmd[2] = (w != w) ? w : (w < -1.0f ? -1.0f : (w > 1.0f ? 1.0f : w));
    } else {  // a later stage on its own reads what the FRAME stage left (calculate_metrics / get_observations use the attributes)
      const float* i0 = a.x.base_lin_vel + env * 3;
      const float* i1 = a.x.base_ang_vel + env * 3;
      const float* i2 = a.x.projected_gravity + env * 3;
      lin = f3{i0[0], i0[1], i0[2]};
      ang = f3{i1[0], i1[1], i1[2]};
      grav = f3{i2[0], i2[1], i2[2]};
    }

    // ---- trip 2 (behind the frame arithmetic, which needs trip 1 alone: issued in front of it, the two trips' registers and
    // the quaternion products' temporaries together are 167 VGPRs, 2 waves per SIMD instead of 4): the env's own 12-wide rows
    // (they only feed three sums), its episode sums, terrain level / type / origin, and pass 0 of the two tables the reset path
    // sums (tile-count bytes of the groups in front of this one, per-group partial sums of the command norm: what the scan /
    // norm launch would have left; L2 hits).  Two 16-byte chunks of each per lane cover 512 groups; the rest, if any, is
    // fetched where pass 0 is consumed, behind the metrics.
    float4 tq3[AD / 4], ldv3[AD / 4], la3[AD / 4];
    if (step) {
      const float4* tq_row = reinterpret_cast<const float4*>(a.b.torques + env * AD);
      const float4* ldv_row = reinterpret_cast<const float4*>(a.b.last_dof_vel + env * AD);
      const float4* la_row = reinterpret_cast<const float4*>(a.b.last_actions + env * AD);
#pragma unroll
      for (int g = 0; g < AD / 4; ++g) { tq3[g] = tq_row[g]; ldv3[g] = ldv_row[g]; la3[g] = la_row[g]; }
    }
    float es[AE];
    if (do_metrics || do_reset) {
#pragma unroll
      for (int k = 0; k < AE; ++k) es[k] = a.b.episode_sums[(int64_t)k * a.n + env];
    } else {
#pragma unroll
      for (int k = 0; k < AE; ++k) es[k] = 0.f;
    }
    int64_t level = a.b.terrain_levels[env];
    int64_t ttype = 0;
    f3 origin{0.f, 0.f, 0.f};
    if (do_reset) {
      ttype = a.b.terrain_types[env];
      origin = f3{a.b.env_origins[env * 3], a.b.env_origins[env * 3 + 1], a.b.env_origins[env * 3 + 2]};
    }

    if (do_frame && a.x.base_lin_vel && live) {  // the reference keeps them as attributes (:422-424)
      float* o0 = a.x.base_lin_vel + env * 3;
      float* o1 = a.x.base_ang_vel + env * 3;
      float* o2 = a.x.projected_gravity + env * 3;
      o0[0] = lin.x; o0[1] = lin.y; o0[2] = lin.z;
      o1[0] = ang.x; o1[1] = ang.y; o1[2] = ang.z;
      o2[0] = grav.x; o2[1] = grav.y; o2[2] = grav.z;
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
    if (step) {  // the 12-wide sums, four columns at a time (sequential order j = 0..11)
#pragma unroll
      for (int g = 0; g < AD / 4; ++g) {
        const float4 tq = tq3[g], ldv = ldv3[g], dv = lds4(row + L_DVEL + 4 * g);
        const float4 la = la3[g], act = lds4(row + L_ACT + 4 * g);
        const float tqv[4] = {tq.x, tq.y, tq.z, tq.w};
        const float d1[4] = {ldv.x - dv.x, ldv.y - dv.y, ldv.z - dv.z, ldv.w - dv.w};
        const float d2[4] = {la.x - act.x, la.y - act.y, la.z - act.z, la.w - act.w};
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const bool first = g == 0 && q == 0;
          s_tq = first ? tqv[q] * tqv[q] : s_tq + tqv[q] * tqv[q];
          s_acc = first ? d1[q] * d1[q] : s_acc + d1[q] * d1[q];
          s_rate = first ? d2[q] * d2[q] : s_rate + d2[q] * d2[q];
        }
        if (g == 0) {
          const float4 dp = lds4(row + L_DPOS);
          const float dpv[4] = {dp.x, dp.y, dp.z, dp.w};
#pragma unroll
          for (int q = 0; q < 4; ++q) {
            const float h = __builtin_fabsf(dpv[q] - p.default_dof_pos[q]);
            s_hip = q == 0 ? h : s_hip + h;
          }
        }
      }
    } else {  // standalone reset_idx: all rows read as zero
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const float h = __builtin_fabsf(0.0f - p.default_dof_pos[q]);
        s_hip = q == 0 ? h : s_hip + h;
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
    rew += (rs[OIGE_AR_TERMINATION] * (flagged ? 1.0f : 0.0f)) * (float)(~timeout);

    // episode sums (:524-533); dict order :68-82 -> row index
    if (step && do_metrics) {
      es[0] += rew_lin_vel_xy; es[2] += rew_ang_vel_z; es[1] += rew_lin_vel_z; es[3] += rew_ang_vel_xy;
      es[4] += rew_orient; es[5] += rew_torque; es[6] += rew_joint_acc; es[11] += rew_action_rate;
      es[7] += rew_base_height; es[12] += rew_hip;
    }

    // in-step reset (:433-435 -> reset_idx :318-367)
    const unsigned long long rmask = __ballot(reset);
    if (!live) level = 0;
    q4 hq = bq;       // pose the height scan sees
    f3 hpos = bpos;
    const bool last_tile = t == T - 1;
    if (own_prefix && last_tile && lane_id() == 0) {  // the total, where the scan leaves it (anymal_extras_body reads it)
      int total = (int)k_front;
      for (int64_t tt = grp * GROUP_TILES; tt <= t; ++tt) total += tt < t ? a.b.reset_workspace[tt] : __popcll(rmask);
      a.b.reset_workspace[padded_tiles(a.n) + NG] = total;
    }
    if (reset) {
      // ---- trip 3: update_terrain_level (:369-380) and the origin of the new level, fetched in front of the Philox draws and
      // of every store of the reset path (r02 fetched it, and the type and the old origin, between those stores).  32-bit
      // arithmetic: a level is below env_rows (the 64-bit remainder is a hundred instructions and a dozen registers)
      if (p.init_done && p.curriculum) {
        const float gnorm = a.unscanned ? gnorm_own : a.b.reduce_workspace[T * RW + 0];
        const float dist = norm3(f3{bpos.x - origin.x, bpos.y - origin.y, 0.0f});
        int lv = (int)level;
        lv -= dist < (gnorm * p.max_episode_length_s) * 0.25f ? 1 : 0;
        lv += dist > p.half_env_length ? 1 : 0;
        lv = (lv < 0 ? 0 : lv) % p.env_rows;
        level = lv;
        const float* to = a.s.terrain_origins + (lv * p.env_cols + (int)ttype) * 3;
        origin = f3{to[0], to[1], to[2]};
      }
      const int64_t g = t / GROUP_TILES;
      const int32_t* tile_counts = a.b.reset_workspace;
      const int32_t* group_offsets = a.b.reset_workspace + padded_tiles(a.n);
      int64_t k = a.unscanned ? k_front : (int64_t)group_offsets[g];
      for (int64_t tt = g * GROUP_TILES; tt < t; ++tt) k += tile_counts[tt];
      k += rank_below(rmask);
      // draws 0..11 dof offset, 12..23 dof vel, 24..25 xy, 26..28 commands; four at a time
      auto draw4 = [&](int blk, float (&u)[4]) {
        if (a.r.u_dof_offset) {
          if (blk < 3) {
            const float4 v = *reinterpret_cast<const float4*>(a.r.u_dof_offset + k * AD + 4 * blk);
            u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
          } else if (blk < 6) {
            const float4 v = *reinterpret_cast<const float4*>(a.r.u_dof_vel + k * AD + 4 * (blk - 3));
            u[0] = v.x; u[1] = v.y; u[2] = v.z; u[3] = v.w;
          } else if (blk == 6) {
            u[0] = a.r.u_xy[k * 2]; u[1] = a.r.u_xy[k * 2 + 1]; u[2] = a.r.u_cmd[k * 3]; u[3] = a.r.u_cmd[k * 3 + 1];
          } else {
            u[0] = a.r.u_cmd[k * 3 + 2]; u[1] = u[2] = u[3] = 0.f;
          }
        } else {
          const u4 c{(uint32_t)env, (uint32_t)blk | (1u << 16), (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
          const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
          u[0] = u01(x.x); u[1] = u01(x.y); u[2] = u01(x.z); u[3] = u01(x.w);
        }
      };
      float4* dpo = reinterpret_cast<float4*>(a.r.dof_pos_out + k * AD);
      float4* dvo = reinterpret_cast<float4*>(a.r.dof_vel_out + k * AD);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float u[4], w[4];
        draw4(c, u);
        draw4(3 + c, w);
        float4 dp, dv;
        dp.x = p.default_dof_pos[4 * c + 0] * (1.0f * u[0] + 0.5f);  // :321,:324  U(0.5,1.5)
        dp.y = p.default_dof_pos[4 * c + 1] * (1.0f * u[1] + 0.5f);
        dp.z = p.default_dof_pos[4 * c + 2] * (1.0f * u[2] + 0.5f);
        dp.w = p.default_dof_pos[4 * c + 3] * (1.0f * u[3] + 0.5f);
        dv.x = 0.2f * w[0] + (-0.1f);  // :322,:325  U(-0.1,0.1)
        dv.y = 0.2f * w[1] + (-0.1f);
        dv.z = 0.2f * w[2] + (-0.1f);
        dv.w = 0.2f * w[3] + (-0.1f);
        dpo[c] = dp;
        dvo[c] = dv;
        *reinterpret_cast<float4*>(row + L_DPOS + 4 * c) = dp;  // the observation sees the reset DOF state
        *reinterpret_cast<float4*>(row + L_DVEL + 4 * c) = dv;
      }
      float u6[4], u7[4];
      draw4(6, u6);
      draw4(7, u7);
      if (p.init_done && p.curriculum) {  // update_terrain_level :369-380 (computed above)
        a.b.terrain_levels[env] = level;
        a.b.env_origins[env * 3] = origin.x; a.b.env_origins[env * 3 + 1] = origin.y; a.b.env_origins[env * 3 + 2] = origin.z;
      }
      hpos = f3{(p.base_init_state[0] + origin.x) + (1.0f * u6[0] + (-0.5f)),
                (p.base_init_state[1] + origin.y) + (1.0f * u6[1] + (-0.5f)), p.base_init_state[2] + origin.z};  // :328-330
      hq = q4{p.base_init_state[3], p.base_init_state[4], p.base_init_state[5], p.base_init_state[6]};           // :331
      a.r.base_pos_out[k * 3] = hpos.x; a.r.base_pos_out[k * 3 + 1] = hpos.y; a.r.base_pos_out[k * 3 + 2] = hpos.z;
      *reinterpret_cast<float4*>(a.r.base_quat_out + k * 4) = make_float4(hq.w, hq.x, hq.y, hq.z);
#pragma unroll
      for (int c = 0; c < 6; ++c) a.r.base_vel_out[k * 6 + c] = p.base_init_state[7 + c];  // :332
      // commands (:341-352)
      cmd[0] = p.cmd_span[0] * u6[2] + p.cmd_lo[0];
      cmd[1] = p.cmd_span[1] * u6[3] + p.cmd_lo[1];
      cmd[3] = p.cmd_span[2] * u7[0] + p.cmd_lo[2];
      const float keep = norm3(f3{cmd[0], cmd[1], 0.0f}) > 0.25f ? 1.0f : 0.0f;
#pragma unroll
      for (int c = 0; c < 4; ++c) cmd[c] *= keep;
      *reinterpret_cast<float4*>(a.b.feet_air_time + env * 4) = make_float4(0.f, 0.f, 0.f, 0.f);  // :356
      a.b.progress_buf[env] = 0;                                                                    // :357
      a.b.reset_buf[env] = 1;  // :358 (already 1 inside post_physics_step)
      a.r.env_ids[k] = env;
    }
    // per-tile sums for extras (:361-367): episode sums of the reset envs BEFORE they are zeroed
    if (do_reset) {
      // the 14 wave sums advance together, one butterfly step at a time: 14 independent cross-lane moves in flight per
      // step instead of 14 dependent chains of six (the same additions in the same order per sum)
      float part[AE + 1];
#pragma unroll
      for (int k = 0; k < AE; ++k) part[k] = reset ? es[k] : 0.f;
      part[AE] = live ? (float)level : 0.f;
#pragma unroll
      for (int off = WAVE / 2; off > 0; off >>= 1) {
        float other[AE + 1];
#pragma unroll
        for (int k = 0; k <= AE; ++k) other[k] = __shfl_xor(part[k], off, WAVE);
#pragma unroll
        for (int k = 0; k <= AE; ++k) part[k] += other[k];
      }
      if (tid == 0) {
#pragma unroll
        for (int k = 0; k <= AE; ++k) a.b.reduce_workspace[t * RW + 1 + k] = part[k];
      }
#pragma unroll
      for (int k = 0; k < AE; ++k)
        if (reset) es[k] = 0.0f;  // :366
    }
    if (live) {
      if (do_metrics || do_reset) {
#pragma unroll
        for (int k = 0; k < AE; ++k) a.b.episode_sums[(int64_t)k * a.n + env] = es[k];
      }
      if (step && do_metrics) a.b.rew_buf[env] = rew;
      if ((step && do_frame) || reset) *reinterpret_cast<float4*>(a.b.commands + env * 4) = make_float4(cmd[0], cmd[1], cmd[2], cmd[3]);
      // last_actions[:] = actions[:]; last_dof_vel[:] = dof_vel[:] -- the view row already holds the reset
      // velocities, so reset envs end the step with the NEW dof_vel, not 0 (:355 overwritten by :442)
      if (do_reset && (step || reset)) {  // standalone reset_idx: last_actions = last_dof_vel = 0 for env_ids (:354-355)
        float4* lao = reinterpret_cast<float4*>(a.b.last_actions + env * AD);
        float4* ldo = reinterpret_cast<float4*>(a.b.last_dof_vel + env * AD);
        const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          lao[c] = step ? lds4(row + L_ACT + 4 * c) : z4;
          ldo[c] = step ? lds4(row + L_DVEL + 4 * c) : z4;
        }
      }
    }
    if (step && do_obs) {
      // observation head (:540-550), in place over the row's first 36 columns
      *reinterpret_cast<float4*>(row + 0) = make_float4(lin.x * p.lin_vel_scale, lin.y * p.lin_vel_scale, lin.z * p.lin_vel_scale,
                                                        ang.x * p.ang_vel_scale);
      *reinterpret_cast<float4*>(row + 4) = make_float4(ang.y * p.ang_vel_scale, ang.z * p.ang_vel_scale, grav.x, grav.y);
      *reinterpret_cast<float4*>(row + 8) = make_float4(grav.z, cmd[0] * p.lin_vel_scale, cmd[1] * p.lin_vel_scale,
                                                        cmd[2] * p.ang_vel_scale);
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float4 dp = lds4(row + L_DPOS + 4 * c), dv = lds4(row + L_DVEL + 4 * c);
        *reinterpret_cast<float4*>(row + L_DPOS + 4 * c) =
            make_float4(dp.x * p.dof_pos_scale, dp.y * p.dof_pos_scale, dp.z * p.dof_pos_scale, dp.w * p.dof_pos_scale);
        *reinterpret_cast<float4*>(row + L_DVEL + 4 * c) =
            make_float4(dv.x * p.dof_vel_scale, dv.y * p.dof_vel_scale, dv.z * p.dof_vel_scale, dv.w * p.dof_vel_scale);
      }
      // yaw-only quaternion (:609-611): zero x,y then normalize (4-wide row: unfused sum)
      const float yn = norm4(q4{hq.w, 0.0f, 0.0f, hq.z});
      const float yd = yn < 1e-9f ? 1e-9f : yn;
      // the pose the scan kernel works from; (:538) base z - 0.5 is the same subtraction for each of the env's 140 points
      if (live) {
        *reinterpret_cast<float4*>(pose_out + tid * APOSE) = make_float4(hq.w / yd, hq.z / yd, hpos.x, hpos.y);
        *reinterpret_cast<float4*>(pose_out + tid * APOSE + 4) = make_float4(hpos.z - 0.5f, 0.f, 0.f, 0.f);
      }
    }
  }
  if (!step || !do_obs) return;  // uniform across the workgroup
  __syncthreads();

  // ---- D1: head (cols 0:36) of every row -> obs_buf with its noise
  {
    constexpr int HF4 = TILE * 9;  // 9 float4 per env
#pragma unroll
    for (int it = 0; it < HF4 / WAVE; ++it) {
      const int f = it * WAVE + tid;
      const int e = f / 9, j = f - 9 * e;
      if (!FULLTILE && e >= nv) continue;
      const int col = 4 * j;  // LDS column == observation column
      float4 v = lds4(rows + e * ARW + col);
      const uint32_t gf = __umul24((uint32_t)env0 + (uint32_t)e, (uint32_t)(AO / 4)) + (uint32_t)j;  // env < 2^23
      const int64_t off = 4 * (int64_t)gf;  // (env0 + e) * AO + col
#ifndef OIGE_EXP_NO_NOISE
      if (p.add_noise) {
        float uu[4];
        anymal_noise4<NOISE_IN>(a.r, noise_step, gf, uu);
        const float4 sc = lds4(s_ns + col);
        v.x = v.x + (2.0f * uu[0] - 1.0f) * sc.x;
        v.y = v.y + (2.0f * uu[1] - 1.0f) * sc.y;
        v.z = v.z + (2.0f * uu[2] - 1.0f) * sc.z;
        v.w = v.w + (2.0f * uu[3] - 1.0f) * sc.w;
      }
#endif
      ANYMAL_HEAD_ST4(a.b.obs_buf + off, v);
    }
  }
}

// One kernel per (NOISE_IN, SPLIT) pair, picked by the host: the registers of the step's own kernel (Philox noise, fused) are
// then not set by the replay / stage variants (as branches of one kernel they took it from 112 to 158 VGPRs).
// Four waves per SIMD = 16 one-wave workgroups per CU = every tile of 256 K envs resident at once (9.4 KB of LDS each).  Left to
// itself the fused kernel takes 151 VGPRs (three waves per SIMD: 56 us); held to 128 it spills ten dwords per lane and runs in
// 49.6 us (r02, with its nine-trip chain: 112 VGPRs, 51.6 us).  -DOIGE_ANYMAL_HEAD_FREE_VGPRS: the A/B build
#ifndef OIGE_ANYMAL_HEAD_FREE_VGPRS
#define HEAD_OCC __attribute__((amdgpu_waves_per_eu(4, 4)))
#else
#define HEAD_OCC
#endif
template <bool NOISE_IN, bool SPLIT>
__global__ __launch_bounds__(WAVE) HEAD_OCC void anymal_head_kernel(const AnymalPost a) {
  __shared__ __align__(16) float lds[TILE * ARW + 48];  // rows + noise scales by column
  const int64_t t = blockIdx.x;
  const int64_t left = a.n - t * TILE;
  if (SPLIT) {  // individually called stages: one generic instantiation, off the hot path
    anymal_head_tile<false, NOISE_IN, true>(a, lds, t, left >= TILE ? TILE : (int)left);
    return;
  }
  if (left >= TILE) anymal_head_tile<true, NOISE_IN>(a, lds, t, TILE);
  else anymal_head_tile<false, NOISE_IN>(a, lds, t, (int)left);
}

static void launch_anymal_head(const AnymalPost& a, int64_t T, hipStream_t st) {
  const bool noise_in = a.r.noise_u != nullptr;
  if (a.stages != 0) {
    if (noise_in) hipLaunchKernelGGL((anymal_head_kernel<true, true>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
    else hipLaunchKernelGGL((anymal_head_kernel<false, true>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
  } else {
    if (noise_in) hipLaunchKernelGGL((anymal_head_kernel<true, false>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
    else hipLaunchKernelGGL((anymal_head_kernel<false, false>), dim3((unsigned)T), dim3(WAVE), 0, st, a);
  }
}

// ------------------------------------------------------------------ height scan (:582-604, :537-539)
// From the pose records the head kernel left.  quat_apply(q_yaw, (hx, hy, 0)) with q_yaw = (w, 0, 0, z) (:608-612), with the
// terms that are exactly zero removed (the surviving products, sums and their order are oige_math.hpp's quat_apply:
// t = 2 (qv x v); v + w t + qv x t, so every finite result is bit-identical):  t = (tx, ty, 0) = (-(2 z hy), 2 z hx, 0),
// rx = (hx + w tx) + (-(z ty)),  ry = (hy + w ty) + (z tx).  tx depends on the grid column only and ty on the grid row only:
// their four products are formed once per env (24 values instead of 140 x 6 multiplies); x / horizontal_scale through
// div_by_const.
typedef float v2f __attribute__((ext_vector_type(2)));

// div_by_const on both halves of a float2 (the same five IEEE operations per half)
__device__ __forceinline__ v2f div_by_const2(v2f x, float h, float inv) {
  const v2f hh = {h, h}, ii = {inv, inv};
  const v2f q0 = x * ii;
  const v2f q1 = __builtin_elementwise_fma(__builtin_elementwise_fma(-q0, hh, x), ii, q0);
  return __builtin_elementwise_fma(__builtin_elementwise_fma(-q1, hh, x), ii, q1);
}

// max(min(i, hi), 0) for hi >= 0 as the one median instruction (the compiler keeps two: it cannot see that hi >= 0)
__device__ __forceinline__ int clip_index(int i, int hi) {
  int r;
  asm("v_med3_i32 %0, %1, 0, %2" : "=v"(r) : "v"(i), "s"(hi));
  return r;
}

__device__ __forceinline__ void cursor_advance(int& e, int& k, int de, int dk, int period) {
  e += de;
  k += dk;
  if (k >= period) { k -= period; e += 1; }
}


// ONE LANE PER (env, grid column iy), the 14 grid rows ix unrolled in the lane; 32 envs = 320 threads per workgroup.
// Everything that depends on (env, iy) -- hy, w tx, z tx, the base position -- is a lane constant in registers, everything that
// depends on ix is a compile-time constant (hx) or one of two 8-byte LDS reads per PAIR of rows (-(z ty), w ty: 28 floats per
// env, tabulated by the env's own ten lanes) at immediate offsets from one per-lane base.  The packed operations pair two grid
// ROWS of one coordinate instead of the two coordinates of one point, so no operand has to be assembled from two sources.
// The 14 heights of a lane are gathered together; the observations are collected in LDS in the order of obs_buf's scan
// columns (float4 i of the tile is float4 i of its part of the buffer) and leave as float4 with their noise.  The noise of a
// float4 depends on its position alone, so its draws (twenty dependent 32 x 32 -> 64-bit multiplies each, the longest chain
// of the kernel) fill the two waits of a workgroup's life: half of them behind the load of the pose records, half behind
// the gathers.
// The r01-r03 kernel (git: 49f661b and before) walked 256-point chunks with one point per lane: per point it read four table
// entries through computed LDS addresses, advanced two (env, k) cursors and shuffled the halves of its operand pairs -- a
// quarter of the 250 instructions of its loop -- and drew its noise behind the gathers' wait.  Same values, bit for bit:
// 70.0 -> 66.4 us.  What else was tried (one lane per point for the gathers, resident workgroups fed by a queue or a fixed
// stride, staggered starts) and what the ablations say: profiles/r03/anymal_scan_layout.txt, DESIGN.md 3.0.
#ifdef OIGE_EXP_SCAN_STILE
constexpr int STILE = OIGE_EXP_SCAN_STILE;
constexpr int SNT = (STILE * 10 + WAVE - 1) / WAVE * WAVE;
#else
constexpr int STILE = 32;        // envs per scan workgroup
constexpr int SNT = STILE * 10;  // 320 threads (5 waves): thread j = env j / 10, grid column j % 10
#endif
constexpr int SCD = 28;          // per-env LDS table: -(z ty)[14], (w ty)[14]
__host__ __device__ inline int64_t scan_tiles(int64_t n) { return (n + STILE - 1) / STILE; }
__host__ __device__ constexpr float scan_hx(int ix) { return 0.1f * (float)(ix < 7 ? ix - 8 : ix - 5); }  // :178-180

template <bool HMIN, bool NOISE_IN>
__device__ __forceinline__ void anymal_scan_tile(const AnymalPost& a, float* __restrict__ s_cd, float* __restrict__ s_out,
                                                 int64_t t, int nv) {
  const oige_anymal_params& p = a.p;
  const int tid = threadIdx.x;
  const uint64_t noise_step = step_of(a.r);
  const int64_t env0 = t * STILE;
  const float* pose_in = a.b.reduce_workspace + pose_offset(num_tiles(a.n)) + env0 * APOSE;
  // (a workgroup with spare lanes: they repeat the first columns of the tile's last env -- same values to the same places)
  const int e = tid / 10 < STILE ? tid / 10 : STILE - 1, iy = tid < STILE * 10 ? tid - 10 * e : tid - STILE * 10;
  const int e_ld = e < nv ? e : nv - 1;  // lanes past a ragged tile's end repeat its last env; nothing of theirs is stored
#ifdef OIGE_EXP_SCAN_YAW
  float4 q = *reinterpret_cast<const float4*>(pose_in + e_ld * APOSE);
  q.x = OIGE_EXP_SCAN_YAW == 0 ? 1.0f : 0.70710678f;
  q.y = OIGE_EXP_SCAN_YAW == 0 ? 0.0f : 0.70710678f;
#else
  const float4 q = *reinterpret_cast<const float4*>(pose_in + e_ld * APOSE);  // yaw quat w, z, base x, base y
#endif
  const float pz = pose_in[e_ld * APOSE + 4];                                  // base z - 0.5

  // the float4 this thread sends out at the end: i = tid + 320 ps of the tile = float4 fk = i % 35 of env i / 35
  constexpr int F4 = STILE * AP4;  // 1120
  constexpr int NPASS = (F4 + SNT - 1) / SNT;
#ifdef OIGE_EXP_SCAN_EARLY
  constexpr int EARLY = OIGE_EXP_SCAN_EARLY;
#else
  constexpr int EARLY = NPASS / 2;  // draws behind the pose load; the others behind the gathers
#endif
  static_assert(AO % 4 == 0 && AP % 4 == 0, "whole float4 per observation row");
  float4 nz[NPASS];
  uint32_t gfs[NPASS];
  const float nscale = p.noise_scales[5];
  int fe = tid / AP4, fk = tid - AP4 * fe;
  auto draw = [&](int ps) {
    gfs[ps] = __umul24((uint32_t)env0 + (uint32_t)fe, (uint32_t)(AO / 4)) + 9u + (uint32_t)fk;  // env < 2^23
    const bool live = fe < nv && fe < STILE && (ps * SNT + SNT <= F4 || ps * SNT + tid < F4);
    if (!live) gfs[ps] = 0xffffffffu;
    nz[ps] = make_float4(0.f, 0.f, 0.f, 0.f);
#ifndef OIGE_EXP_NO_NOISE
    if (p.add_noise && live) {
      float uu[4];
      anymal_noise4<NOISE_IN>(a.r, noise_step, gfs[ps], uu);
      nz[ps] = make_float4((2.0f * uu[0] - 1.0f) * nscale, (2.0f * uu[1] - 1.0f) * nscale, (2.0f * uu[2] - 1.0f) * nscale,
                           (2.0f * uu[3] - 1.0f) * nscale);
    }
#endif
    cursor_advance(fe, fk, SNT / AP4, SNT % AP4, AP4);  // + 320 float4 = 9 envs + 5
  };
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ps = 0; ps < EARLY; ++ps) draw(ps);
  __builtin_amdgcn_sched_barrier(0);

  // meshgrid(x, y, indexing='ij') flattened: k = ix * 10 + iy
#pragma unroll
  for (int h = 0; h < 2; ++h) {  // the env's lane iy tabulates rows ix = iy and ix = iy + 10
    const int ix = iy + 10 * h;
    const float hx = 0.1f * (float)(ix < 7 ? ix - 8 : ix - 5);
    const float ty = (q.y * hx) * 2.0f;
    if (h == 0 || iy < 4) {
      s_cd[e * SCD + ix] = -(q.y * ty);
      s_cd[e * SCD + 14 + ix] = q.x * ty;
    }
  }
  const float hy = 0.1f * (float)(iy < 5 ? iy - 5 : iy - 4);  // :175-177
  const float tx = -((q.y * hy) * 2.0f);
  const float wtx = q.x * tx, ztx = q.y * tx;
  __syncthreads();
#ifndef OIGE_EXP_NO_SCAN
  const int hs_cols = p.hs_cols, max_r = p.hs_rows - 2, max_c = p.hs_cols - 2;
  const float hscale = p.horizontal_scale, hinv = 1.0f / p.horizontal_scale;
  const v2f border = {p.border_size, p.border_size};
  const char* hmin = reinterpret_cast<const char*>(a.s.height_min);  // NULL (HMIN false) -> the reference's two gathers
  const float* cd = s_cd + e * SCD;
  constexpr int NR = 14;
  uint32_t o1[NR];
  // rx = (hx + w tx) + (-(z ty)), ry = (hy + w ty) + (z tx); then ((r + base) + border) / horizontal_scale (:586-593) as
  // div_by_const2's five operations; .long() truncates toward zero, and so does v_cvt_i32_f32, which also saturates: the
  // clip that follows (:596-597) maps every saturated value where int64 truncation + clip would
#pragma unroll
  for (int m = 0; m < NR / 2; ++m) {
    const v2f nzty = *reinterpret_cast<const v2f*>(cd + 2 * m);
    const v2f wty = *reinterpret_cast<const v2f*>(cd + 14 + 2 * m);
    v2f x = v2f{scan_hx(2 * m), scan_hx(2 * m + 1)} + v2f{wtx, wtx};
    v2f y = v2f{hy, hy} + wty;
    x = x + nzty;
    y = y + v2f{ztx, ztx};
    x = x + v2f{q.z, q.z};
    y = y + v2f{q.w, q.w};
    x = x + border;
    y = y + border;
    const v2f qx = div_by_const2(x, hscale, hinv), qy = div_by_const2(y, hscale, hinv);
    // byte offsets into the int16 map (< 2 GiB): both factors below 2^24, v_mad_i32_i24 is full rate; unsigned 32-bit, so the
    // gathers take the table's base from scalar registers and the offset as it is
    o1[2 * m] = 2u * (uint32_t)(__mul24(clip_index((int)qx.x, max_r), hs_cols) + clip_index((int)qy.x, max_c));
    o1[2 * m + 1] = 2u * (uint32_t)(__mul24(clip_index((int)qx.y, max_r), hs_cols) + clip_index((int)qy.y, max_c));
  }
  int32_t hv[NR];
  // every gather of the lane leaves before the first is consumed
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int g = 0; g < NR; ++g) {
#if defined(OIGE_EXP_SCAN_NOGATHER)
    hv[g] = (int16_t)o1[g];
#else
    if (HMIN) {  // min(h[px, py], h[px + 1, py + 1]) tabulated per map: one gather (oige_anymal_height_min)
      hv[g] = *reinterpret_cast<const int16_t*>(hmin + o1[g]);
    } else {
      const char* hb = reinterpret_cast<const char*>(a.s.height_samples);
      const int16_t h1 = *reinterpret_cast<const int16_t*>(hb + o1[g]);
      const int16_t h2 = *reinterpret_cast<const int16_t*>(hb + (o1[g] + 2u * (uint32_t)(hs_cols + 1)));
      hv[g] = h1 < h2 ? h1 : h2;
    }
#endif
  }
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int ps = EARLY; ps < NPASS; ++ps) draw(ps);
  __builtin_amdgcn_sched_barrier(0);
  // torch.clip keeps NaN where the median instruction returns a bound.  The heights are finite (int16 x vertical_scale),
  // so the difference is NaN only where base z is: one test per wave instead of two instructions per point
  const bool odd = (pz != pz) || !(__builtin_fabsf(p.vertical_scale) < __builtin_inff());
  float* out = s_out + e * AP + iy;
  if (__builtin_amdgcn_ballot_w64(odd) == 0) {
#pragma unroll
    for (int g = 0; g < NR; ++g) {
      const float hd = pz - (float)hv[g] * p.vertical_scale;
      out[10 * g] = __builtin_fminf(__builtin_fmaxf(hd, -1.0f), 1.0f) * p.height_meas_scale;
    }
  } else {
#pragma unroll
    for (int g = 0; g < NR; ++g) {
      const float hd = pz - (float)hv[g] * p.vertical_scale;
      const float cl = __builtin_fminf(__builtin_fmaxf(hd, -1.0f), 1.0f);
      out[10 * g] = ((hd != hd) ? hd : cl) * p.height_meas_scale;
    }
  }
  __syncthreads();
#pragma unroll
  for (int ps = 0; ps < NPASS; ++ps) {
    if (gfs[ps] == 0xffffffffu) continue;
    float4 v = lds4(s_out + 4 * (ps * SNT + tid));
    if (p.add_noise) {  // (x + 0 would turn a -0 into +0)
      v.x = v.x + nz[ps].x;
      v.y = v.y + nz[ps].y;
      v.z = v.z + nz[ps].z;
      v.w = v.w + nz[ps].w;
    }
#ifdef OIGE_EXP_SCAN_NOSTORE
    if (v.x == 123456.75f)
#endif
    ANYMAL_SCAN_ST4(a.b.obs_buf + 4 * (int64_t)gfs[ps], v);  // (env0 + fe) * AO + 36 + 4 * fk
  }
#endif
}

// extras (:361-367): mean over the reset envs of each episode sum / max_episode_length_s, mean terrain level.
// One workgroup of NT threads folds the per-tile partial sums in f64.
template <int NT>
__device__ __forceinline__ void anymal_extras_body(const AnymalPost& a, double (*part)[AE + 1]) {
  const int64_t T = num_tiles(a.n);
  double acc[AE + 1];
#pragma unroll
  for (int k = 0; k <= AE; ++k) acc[k] = 0.0;
#pragma unroll 4
  for (int64_t i = threadIdx.x; i < T; i += NT) {
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
    for (int i = 0; i < NT / WAVE; ++i) tot += part[i][k];
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

// Workgroup 0 folds the extras while the others scan (a one-workgroup, latency-bound reduction that would otherwise run
// alone for ~12 us after the scan): workgroup b > 0 owns scan tile b - 1 (STILE envs).
#ifdef OIGE_EXP_SCAN_OCC
#define SCAN_OCC __attribute__((amdgpu_waves_per_eu(OIGE_EXP_SCAN_OCC, OIGE_EXP_SCAN_OCC)))
#else
#define SCAN_OCC
#endif
__global__ __launch_bounds__(SNT) SCAN_OCC void anymal_scan_kernel(const AnymalPost a) {
  __shared__ __align__(16) float s_out[STILE * AP];
  __shared__ __align__(16) float s_cd[STILE * SCD];
  if (blockIdx.x == 0) {
    static_assert(sizeof(double) * (SNT / WAVE) * (AE + 1) <= sizeof(float) * STILE * AP, "extras partials fit in s_out");
    static_assert(SNT >= AE + 1, "one thread per extras entry");
    if (a.stages == 0) anymal_extras_body<SNT>(a, reinterpret_cast<double (*)[AE + 1]>(s_out));  // (get_observations alone resets nothing)
    return;
  }
  const int64_t t = (int64_t)blockIdx.x - 1;
  const int64_t left = a.n - t * STILE;
  const int nv = left >= STILE ? STILE : (int)left;
  // the product path (tabulated min, Philox noise) is one specialisation; the others share slower generic ones
  const bool hmin = a.s.height_min != nullptr, noise_in = a.r.noise_u != nullptr;
  if (hmin && !noise_in) anymal_scan_tile<true, false>(a, s_cd, s_out, t, nv);
  else if (hmin) anymal_scan_tile<true, true>(a, s_cd, s_out, t, nv);
  else if (!noise_in) anymal_scan_tile<false, false>(a, s_cd, s_out, t, nv);
  else anymal_scan_tile<false, true>(a, s_cd, s_out, t, nv);
}
__host__ inline void launch_anymal_scan(const AnymalPost& a, hipStream_t st) {
  hipLaunchKernelGGL(anymal_scan_kernel, dim3((unsigned)scan_tiles(a.n) + 1), dim3(SNT), 0, st, a);
}

// update_terrain_level(env_ids) on its own (:369-380; reset_idx runs it inside the head kernel).  One workgroup: the
// whole-matrix norm of the ids' xy commands first (torch.norm over ALL of them, :376; f64 partials in thread order), then
// every id's level and origin.  A few thousand ids at most per call in the reference's regime; latency-bound.
__global__ __launch_bounds__(1024) void anymal_terrain_level_kernel(const oige_anymal_params p, const float* __restrict__ root_pos,
                                                                    const float* __restrict__ terrain_origins,
                                                                    const float* __restrict__ commands,
                                                                    int64_t* __restrict__ terrain_levels,
                                                                    const int64_t* __restrict__ terrain_types,
                                                                    float* __restrict__ env_origins, const int64_t* __restrict__ env_ids,
                                                                    int64_t num_ids, int64_t n) {
  __shared__ double part[1024 / WAVE];
  __shared__ float s_norm;
  double acc = 0.0;
  for (int64_t k = threadIdx.x; k < num_ids; k += 1024) {
    const int64_t i = env_ids[k];
    if (i < 0 || i >= n) continue;
    const float2 c = *reinterpret_cast<const float2*>(commands + i * 4);
    acc += (double)(c.x * c.x) + (double)(c.y * c.y);
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_xor(acc, off, WAVE);
  if (lane_id() == 0) part[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int w = 0; w < 1024 / WAVE; ++w) t += part[w];
    s_norm = __builtin_sqrtf((float)t);
  }
  __syncthreads();
  const float gnorm = s_norm;
  for (int64_t k = threadIdx.x; k < num_ids; k += 1024) {
    const int64_t i = env_ids[k];
    if (i < 0 || i >= n) continue;
    const float dist = norm3(f3{root_pos[i * 3] - env_origins[i * 3], root_pos[i * 3 + 1] - env_origins[i * 3 + 1], 0.0f});  // :375
    int64_t level = terrain_levels[i];
    level -= dist < (gnorm * p.max_episode_length_s) * 0.25f ? 1 : 0;  // :376
    level += dist > p.half_env_length ? 1 : 0;                          // :377
    level = (level < 0 ? 0 : level) % p.env_rows;                       // :378
    terrain_levels[i] = level;
    const float* to = terrain_origins + (level * p.env_cols + terrain_types[i]) * 3;  // :379
    env_origins[i * 3] = to[0]; env_origins[i * 3 + 1] = to[1]; env_origins[i * 3 + 2] = to[2];
  }
}

// stand-alone extras for reset_idx(env_ids) (no scan there)
__global__ __launch_bounds__(1024) void anymal_extras_kernel(const AnymalPost a) {
  __shared__ double part[1024 / WAVE][AE + 1];
  anymal_extras_body<1024>(a, part);
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_anymal_height_min(const int16_t* height_samples, int32_t hs_rows, int32_t hs_cols, int16_t* out, oige_stream_t stream) {
  OIGE_REQUIRE(height_samples && out && height_samples != out, "anymal_height_min: null or aliased pointer");
  OIGE_REQUIRE(hs_rows >= 2 && hs_cols >= 2 && (int64_t)hs_rows * hs_cols < ((int64_t)1 << 30), "anymal_height_min: bad map dimensions");
  const int64_t total = (int64_t)hs_rows * hs_cols;
  hipLaunchKernelGGL(anymal_height_min_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, (hipStream_t)stream, height_samples,
                     (int)hs_rows, (int)hs_cols, out);
  return check_launch("anymal_height_min_kernel");
}

static int anymal_torques(const oige_anymal_params* params, const float* actions, float* actions_out, const float* dof_pos,
                          const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream) {
  OIGE_REQUIRE(params && actions && dof_pos && dof_vel && torques_out, "anymal_torques: null pointer");
  OIGE_REQUIRE(num_envs >= 0 && num_envs < ((int64_t)1 << 31) / 256, "anymal_torques: num_envs %lld out of range", (long long)num_envs);
  OIGE_REQUIRE(aligned16(actions) && aligned16(dof_pos) && aligned16(dof_vel) && aligned16(torques_out) && aligned16(actions_out),
               "anymal_torques: buffers must be 16-byte aligned");
  OIGE_REQUIRE(actions_out != actions, "anymal_torques: actions_out must not alias actions");
  const int64_t total = num_envs * AD;
  if (total == 0) return OIGE_OK;
  const int64_t f4 = total / 4;
  hipLaunchKernelGGL(anymal_torques_kernel, dim3((unsigned)((f4 + 255) / 256)), dim3(256), 0, (hipStream_t)stream, *params,
                     actions, dof_pos, dof_vel, torques_out, actions_out, total);
  return check_launch("anymal_torques_kernel");
}

int oige_anymal_torques(const oige_anymal_params* params, const float* actions, const float* dof_pos,
                        const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream) {
  return anymal_torques(params, actions, nullptr, dof_pos, dof_vel, torques_out, num_envs, stream);
}

int oige_anymal_torques_clone(const oige_anymal_params* params, const float* actions, float* actions_out, const float* dof_pos,
                              const float* dof_vel, float* torques_out, int64_t num_envs, oige_stream_t stream) {
  OIGE_REQUIRE(actions_out != nullptr, "anymal_torques_clone: null actions_out");
  return anymal_torques(params, actions, actions_out, dof_pos, dof_vel, torques_out, num_envs, stream);
}

static int anymal_launch(const oige_anymal_params* params, const oige_anymal_state* state,
                         const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io, int64_t num_envs,
                         int mode, oige_stream_t stream, uint32_t stages = 0, const oige_anymal_frame* frame = nullptr) {
  OIGE_REQUIRE(params && state && buffers && reset_io, "anymal: null struct pointer");
  OIGE_REQUIRE((stages & ~OIGE_ANYMAL_STAGE_ALL) == 0, "anymal_post_physics_stages: bad stages mask 0x%x", stages);
  if (stages == OIGE_ANYMAL_STAGE_ALL) stages = 0;  // every stage = the fused step
  OIGE_REQUIRE(stages == 0 || !(stages & (OIGE_ANYMAL_STAGE_RESET | OIGE_ANYMAL_STAGE_NOISE | OIGE_ANYMAL_STAGE_LAST)),
               "anymal_post_physics_stages: RESET / NOISE / LAST only run inside the fused step (all stages); on their own they are "
               "reset_idx(env_ids) = oige_anymal_reset_idx and two elementwise torch lines (anymal_terrain.py:438-442)");
  const bool head_split = stages & (OIGE_ANYMAL_STAGE_FRAME | OIGE_ANYMAL_STAGE_METRICS | OIGE_ANYMAL_STAGE_OBS);
  if (frame) {
    const bool any = frame->base_lin_vel || frame->base_ang_vel || frame->projected_gravity;
    OIGE_REQUIRE(!any || (frame->base_lin_vel && frame->base_ang_vel && frame->projected_gravity),
                 "anymal: pass all three frame tensors or none");
  }
  if (head_split)
    OIGE_REQUIRE(frame && frame->base_lin_vel, "anymal_post_physics_stages: FRAME / METRICS / OBS on their own need the frame tensors "
                                               "(base_lin_vel, base_ang_vel, projected_gravity)");
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
  OIGE_REQUIRE(!r.push_velocities || (r.push_count && (reinterpret_cast<uintptr_t>(r.push_velocities) & 7u) == 0),
               "anymal: push_velocities needs push_count and 8-byte alignment");
  const bool drawn = r.u_dof_offset != nullptr;
  OIGE_REQUIRE(drawn == (r.u_dof_vel != nullptr) && drawn == (r.u_xy != nullptr) && drawn == (r.u_cmd != nullptr),
               "anymal: pass all four pre-drawn reset tables or none");
  OIGE_REQUIRE(aligned16(s.base_quat) && aligned16(s.dof_pos) && aligned16(s.dof_vel) && aligned16(s.knee_contact) &&
                   aligned16(b.obs_buf) && aligned16(b.commands) && aligned16(b.actions) && aligned16(b.last_actions) &&
                   aligned16(b.last_dof_vel) && aligned16(b.torques) && aligned16(b.feet_air_time) && aligned16(r.base_quat_out) &&
                   aligned16(r.dof_pos_out) && aligned16(r.dof_vel_out) && aligned16(b.reset_workspace) &&
                   aligned16(r.noise_u) && aligned16(b.reduce_workspace) && (reinterpret_cast<uintptr_t>(s.base_velocities) & 7u) == 0,
               "anymal: row buffers must be 16-byte aligned (base_velocities 8)");
  OIGE_REQUIRE(!b.tile_counts_u8 || aligned16(b.tile_counts_u8), "anymal: tile_counts_u8 must be 16-byte aligned");
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
  a.stages = stages;
  a.x = frame ? *frame : oige_anymal_frame{nullptr, nullptr, nullptr};
  hipStream_t st = (hipStream_t)stream;
  const int64_t G = num_groups(num_envs), T = num_tiles(num_envs);
  a.unscanned = (b.tile_counts_u8 && G <= UNSCANNED_U8_MAX_GROUPS) ? 1 : 0;
  if (stages != 0) {  // individually callable stages, in the reference's order (anymal_terrain.py:410-437)
    if (stages & (OIGE_ANYMAL_STAGE_PROGRESS | OIGE_ANYMAL_STAGE_TERMINATION)) {
      hipLaunchKernelGGL(anymal_flags_kernel, dim3((unsigned)G), dim3(GROUP), 0, st, a);
      if (int rc = check_launch("anymal_flags_kernel")) return rc;
    }
    if (head_split) {
      if (stages & OIGE_ANYMAL_STAGE_OBS) a.p.add_noise = 0;  // get_observations adds none (:535-552); :438-439 is the caller's line
      launch_anymal_head(a, T, st);
      if (int rc = check_launch("anymal_head_kernel")) return rc;
      if (stages & OIGE_ANYMAL_STAGE_OBS) {
        launch_anymal_scan(a, st);
        return check_launch("anymal_scan_kernel");
      }
    }
    return OIGE_OK;
  }
  hipLaunchKernelGGL(anymal_flags_kernel, dim3((unsigned)G), dim3(GROUP), 0, st, a);
  if (int rc = check_launch("anymal_flags_kernel")) return rc;
  if (!a.unscanned) {
    hipLaunchKernelGGL(anymal_scan_norm_kernel, dim3(2), dim3(SCAN_THREADS), 0, st, b.reset_workspace, G, padded_tiles(num_envs),
                       b.reduce_workspace, T);
    if (int rc = check_launch("anymal_scan_norm_kernel")) return rc;
  }
  launch_anymal_head(a, T, st);
  if (int rc = check_launch("anymal_head_kernel")) return rc;
  if (mode == 0) {  // T scan workgroups + one that folds the extras
    launch_anymal_scan(a, st);
    return check_launch("anymal_scan_kernel");
  }
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

int oige_anymal_post_physics_stages(const oige_anymal_params* params, const oige_anymal_state* state,
                                    const oige_anymal_buffers* buffers, const oige_anymal_reset_io* reset_io,
                                    const oige_anymal_frame* frame, int64_t num_envs, uint32_t stages, oige_stream_t stream) {
  OIGE_REQUIRE(stages != 0, "anymal_post_physics_stages: empty stages mask");
  return anymal_launch(params, state, buffers, reset_io, num_envs, 0, stream, stages, frame);
}

int oige_anymal_update_terrain_level(const oige_anymal_params* params, const float* root_pos, const float* terrain_origins,
                                     const float* commands, int64_t* terrain_levels, const int64_t* terrain_types,
                                     float* env_origins, const int64_t* env_ids, int64_t num_ids, int64_t num_envs,
                                     oige_stream_t stream) {
  OIGE_REQUIRE(params && root_pos && terrain_origins && commands && terrain_levels && terrain_types && env_origins,
               "anymal_update_terrain_level: null pointer");
  OIGE_REQUIRE(num_ids >= 0 && num_envs >= 0 && num_ids <= num_envs && (num_ids == 0 || env_ids), "anymal_update_terrain_level: bad ids");
  OIGE_REQUIRE(params->env_rows >= 1 && params->env_cols >= 1, "anymal_update_terrain_level: bad terrain dimensions");
  if (num_ids == 0 || !params->init_done || !params->curriculum) return OIGE_OK;  // :370-372
  hipLaunchKernelGGL(anymal_terrain_level_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, *params, root_pos, terrain_origins, commands,
                     terrain_levels, terrain_types, env_origins, env_ids, num_ids, num_envs);
  return check_launch("anymal_terrain_level_kernel");
}

}  // extern "C"
