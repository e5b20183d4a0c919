// Cartpole per-step task math for gfx950: /root/reference/omniisaacgymenvs/tasks/cartpole.py:91-164.
// 60 algorithmic bytes per env-step; one env per lane, every access is a
// coalesced 8/16-byte-per-lane stream, so there is nothing to stage through LDS.
#include "oige_common.hpp"
#include "oige_math.hpp"

namespace oige {

struct CartPost {
  oige_cartpole_params p;
  oige_cartpole_buffers b;
  const float* dof_pos;
  const float* dof_vel;
  int64_t n;
  uint32_t stages;
};

__global__ __launch_bounds__(GROUP) void cartpole_post_kernel(const CartPost a) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool live = env < a.n;
  const oige_cartpole_params& p = a.p;
  int64_t r = 0;
  if (live) {
    float cart_pos, cart_vel, pole_pos, pole_vel;
    if (a.stages & OIGE_STAGE_OBS) {
      // get_observations (:91-103)
      const float* qp = a.dof_pos + env * p.num_dof;
      const float* qv = a.dof_vel + env * p.num_dof;
      cart_pos = qp[p.cart_dof_idx];
      cart_vel = qv[p.cart_dof_idx];
      pole_pos = qp[p.pole_dof_idx];
      pole_vel = qv[p.pole_dof_idx];
      *reinterpret_cast<float4*>(a.b.obs_buf + env * 4) = make_float4(cart_pos, cart_vel, pole_pos, pole_vel);
    } else {
      const float4 o = *reinterpret_cast<const float4*>(a.b.obs_buf + env * 4);
      cart_pos = o.x; cart_vel = o.y; pole_pos = o.z; pole_vel = o.w;
    }
    int64_t progress = 0;
    if (a.stages & (OIGE_STAGE_PROGRESS | OIGE_STAGE_DONE)) progress = a.b.progress_buf[env];
    if (a.stages & OIGE_STAGE_PROGRESS) {  // rl_task.py:289
      progress += 1;
      a.b.progress_buf[env] = progress;
    }
    const bool cart_out = __builtin_fabsf(cart_pos) > p.reset_dist;
    const bool pole_out = __builtin_fabsf(pole_pos) > OIGE_HALF_PI_F;
    if (a.stages & OIGE_STAGE_METRICS) {  // calculate_metrics (:153-158)
      float rew = ((1.0f - pole_pos * pole_pos) - 0.01f * __builtin_fabsf(cart_vel)) - 0.005f * __builtin_fabsf(pole_vel);
      rew = cart_out ? -2.0f : rew;
      rew = pole_out ? -2.0f : rew;
      a.b.rew_buf[env] = rew;
    }
    if (a.stages & OIGE_STAGE_DONE) {  // is_done (:160-164): not sticky, >= max length
      r = cart_out ? 1 : 0;
      r = pole_out ? 1 : r;
      r = progress >= (int64_t)p.max_episode_length ? 1 : r;
      a.b.reset_buf[env] = r;
    }
  }
  if (a.stages & OIGE_STAGE_DONE) {
    const unsigned long long m = __ballot(live && r != 0);
    if (a.b.reset_workspace && lane_id() == 0) a.b.reset_workspace[env / TILE] = __popcll(m);
  }
}

struct CartPre {
  oige_cartpole_params p;
  oige_cartpole_buffers b;
  oige_cartpole_reset_io r;
  const float* actions;
  float* forces_out;
  int64_t n;
};

// reset_idx (:124-144) for env i landing in compacted row k; n_rows = number of rows of `u`
__device__ __forceinline__ void cartpole_reset_env(const CartPre& a, int64_t i, int64_t k, int64_t n_rows) {
  const oige_cartpole_params& p = a.p;
  float u[4];
  if (a.r.u) {
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = a.r.u[q * n_rows + k];
  } else {
    const u4 c{(uint32_t)i, 0u, (uint32_t)step_of(a.r), (uint32_t)(step_of(a.r) >> 32)};
    const u4 x = philox4x32_10(c, (uint32_t)a.r.seed, (uint32_t)(a.r.seed >> 32));
    u[0] = u01(x.x); u[1] = u01(x.y); u[2] = u01(x.z); u[3] = u01(x.w);
  }
  const float cart_q = 1.0f * (1.0f - 2.0f * u[0]);                  // :129
  const float pole_q = 0.39269908169872414f * (1.0f - 2.0f * u[1]);  // :130  0.125*pi folded in double
  const float cart_v = 0.5f * (1.0f - 2.0f * u[2]);                  // :134
  const float pole_v = 0.7853981633974483f * (1.0f - 2.0f * u[3]);   // :135  0.25*pi
  for (int j = 0; j < p.num_dof; ++j) {
    a.r.dof_pos_out[k * p.num_dof + j] = j == p.cart_dof_idx ? cart_q : (j == p.pole_dof_idx ? pole_q : 0.0f);
    a.r.dof_vel_out[k * p.num_dof + j] = j == p.cart_dof_idx ? cart_v : (j == p.pole_dof_idx ? pole_v : 0.0f);
  }
  a.b.reset_buf[i] = 0;     // :143
  a.b.progress_buf[i] = 0;  // :144
  a.r.env_ids[k] = i;
}

__global__ __launch_bounds__(GROUP) void cartpole_pre_kernel(const CartPre a) {
  __shared__ int lds4[GROUP_TILES];
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + threadIdx.x;
  const int64_t G = num_groups(a.n);
  const int32_t* group_offsets = a.b.reset_workspace + G * GROUP_TILES;
  const bool live = env < a.n;
  const bool flagged = live && a.b.reset_buf[env] != 0;
  const int64_t k = group_rank(flagged, group_offsets, g, lds4);
  const int64_t total = (int64_t)group_offsets[G];
  if (flagged) cartpole_reset_env(a, env, k, total);
  if (live) {  // forces (:118-119)
    const float f = a.p.max_push_effort * a.actions[env];
    for (int j = 0; j < a.p.num_dof; ++j) a.forces_out[env * a.p.num_dof + j] = j == a.p.cart_dof_idx ? f : 0.0f;
  }
  if (g == 0 && threadIdx.x == 0) *a.r.count = total;
}

__global__ __launch_bounds__(256) void cartpole_reset_idx_kernel(const CartPre a, int64_t max_ids) {
  const int64_t k = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  const int64_t cnt = *a.r.count;
  if (k >= max_ids || k >= cnt) return;
  const int64_t i = a.r.env_ids[k];
  if (i < 0 || i >= a.n) return;
  cartpole_reset_env(a, i, k, cnt);
}

static int check_cartpole(const oige_cartpole_params* p, const oige_cartpole_buffers* b, int64_t n) {
  OIGE_REQUIRE(p && b, "cartpole: null params/buffers");
  OIGE_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "cartpole: num_envs %lld out of range", (long long)n);
  OIGE_REQUIRE(p->num_dof >= 1 && p->num_dof <= OIGE_MAX_DOF, "cartpole: num_dof %d out of range", p->num_dof);
  OIGE_REQUIRE(p->cart_dof_idx >= 0 && p->cart_dof_idx < p->num_dof && p->pole_dof_idx >= 0 && p->pole_dof_idx < p->num_dof,
               "cartpole: dof index out of range");
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_cartpole_post_physics(const oige_cartpole_params* params, const float* dof_pos, const float* dof_vel,
                               const oige_cartpole_buffers* buffers, int64_t num_envs, uint32_t stages,
                               oige_stream_t stream) {
  if (int rc = check_cartpole(params, buffers, num_envs)) return rc;
  OIGE_REQUIRE((stages & ~OIGE_STAGE_ALL) == 0 && stages != 0, "cartpole_post_physics: bad stages mask 0x%x", stages);
  OIGE_REQUIRE(buffers->obs_buf && aligned16(buffers->obs_buf), "cartpole_post_physics: obs_buf null or not 16-byte aligned");
  if (stages & OIGE_STAGE_OBS) OIGE_REQUIRE(dof_pos && dof_vel, "cartpole_post_physics: null dof state");
  if (stages & (OIGE_STAGE_PROGRESS | OIGE_STAGE_DONE)) OIGE_REQUIRE(buffers->progress_buf, "cartpole_post_physics: null progress_buf");
  if (stages & OIGE_STAGE_METRICS) OIGE_REQUIRE(buffers->rew_buf, "cartpole_post_physics: null rew_buf");
  if (stages & OIGE_STAGE_DONE) OIGE_REQUIRE(buffers->reset_buf, "cartpole_post_physics: null reset_buf");
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  CartPost a{*params, *buffers, dof_pos, dof_vel, num_envs, stages};
  hipLaunchKernelGGL(cartpole_post_kernel, dim3((unsigned)G), dim3(GROUP), 0, (hipStream_t)stream, a);
  return check_launch("cartpole_post_kernel");
}

int oige_cartpole_pre_physics(const oige_cartpole_params* params, const float* actions,
                              const oige_cartpole_buffers* buffers, const oige_cartpole_reset_io* reset_io,
                              float* forces_out, int64_t num_envs, uint32_t flags, oige_stream_t stream) {
  if (int rc = check_cartpole(params, buffers, num_envs)) return rc;
  OIGE_REQUIRE(actions && reset_io && forces_out, "cartpole_pre_physics: null pointer");
  OIGE_REQUIRE(buffers->reset_buf && buffers->progress_buf && buffers->reset_workspace && aligned16(buffers->reset_workspace),
               "cartpole_pre_physics: null reset/progress/workspace (workspace 16-byte aligned)");
  OIGE_REQUIRE(reset_io->env_ids && reset_io->count && reset_io->dof_pos_out && reset_io->dof_vel_out,
               "cartpole_pre_physics: null reset output");
  OIGE_REQUIRE(reset_io->u == nullptr, "cartpole_pre_physics: pre-drawn u needs the row count; use reset_idx for that path");
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipStream_t s = (hipStream_t)stream;
  if (!(flags & OIGE_PRE_COUNTS_VALID))
    if (int rc = launch_count_flags(buffers->reset_buf, num_envs, buffers->reset_workspace, s)) return rc;
  if (int rc = launch_scan_groups(num_envs, buffers->reset_workspace, s)) return rc;
  CartPre a{*params, *buffers, *reset_io, actions, forces_out, num_envs};
  hipLaunchKernelGGL(cartpole_pre_kernel, dim3((unsigned)G), dim3(GROUP), 0, s, a);
  return check_launch("cartpole_pre_kernel");
}

int oige_cartpole_reset_idx(const oige_cartpole_params* params, const oige_cartpole_buffers* buffers,
                            const oige_cartpole_reset_io* reset_io, int64_t num_envs, int64_t max_ids,
                            oige_stream_t stream) {
  if (int rc = check_cartpole(params, buffers, num_envs)) return rc;
  OIGE_REQUIRE(reset_io && max_ids >= 0 && max_ids <= num_envs, "cartpole_reset_idx: bad reset_io/max_ids");
  OIGE_REQUIRE(buffers->reset_buf && buffers->progress_buf, "cartpole_reset_idx: null reset/progress");
  OIGE_REQUIRE(reset_io->env_ids && reset_io->count && reset_io->dof_pos_out && reset_io->dof_vel_out,
               "cartpole_reset_idx: null reset pointer");
  if (max_ids == 0) return OIGE_OK;
  CartPre a{*params, *buffers, *reset_io, nullptr, nullptr, num_envs};
  hipLaunchKernelGGL(cartpole_reset_idx_kernel, dim3((unsigned)((max_ids + 255) / 256)), dim3(256), 0,
                     (hipStream_t)stream, a, max_ids);
  return check_launch("cartpole_reset_idx_kernel");
}

}  // extern "C"
