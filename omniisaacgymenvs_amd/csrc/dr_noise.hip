// Observation / action noise of the reference's domain randomisation, for gfx950 (MI355X).
//
// Replaces /root/reference/omniisaacgymenvs/utils/domain_randomization/randomize.py:
//   apply_observations_randomization :283-312, apply_actions_randomization :314-342,
//   _apply_uncorrelated_noise :344-371, _apply_correlated_noise :373-410,
// and, optionally, the clamp that follows it in VecEnvRLGames._process_data
// (envs/vec_env_rlgames.py:40).
//
// One launch, HBM-bound: per element R buffer 4 + R correlated 4 + W buffer 4 (+ W clamped 4), per env
// R reset flag 8 + RW counter 8.  One wavefront owns a tile of 64 consecutive envs: lane = env for the
// counter / flag logic (:284-286, :299-304), then the wave streams the tile's contiguous 64*W floats
// with lane-linear 16-byte accesses, looking the env's decision up in LDS.
//
// Samples: either pre-drawn by the caller in the reference's compacted order (exact replay of torch's
// RNG: rows = reset ids / randomize ids ascending; needs oige_dr_noise_count first for the ranks), or
// generated in-kernel from Philox4x32-10 keyed by (seed, step, flat element index, stream) with
// Box-Muller for the gaussian (no host sync, distribution parity only).
#include "oige_common.hpp"
#include "oige_math.hpp"
#include "oige_noise.hpp"

namespace oige {

__global__ __launch_bounds__(GROUP) void dr_noise_count_kernel(const oige_noise_args a, int64_t n) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const NoiseDecision d = noise_decide(a, env, env < n);
  const unsigned long long mr = __ballot(d.reset), mi = __ballot(d.interval);
  if (lane_id() == 0) {
    const int64_t t = (int64_t)blockIdx.x * GROUP_TILES + (threadIdx.x >> 6);
    a.reset_workspace[t] = __popcll(mr);
    a.interval_workspace[t] = __popcll(mi);
  }
}

__global__ __launch_bounds__(GROUP) void dr_noise_apply_kernel(const oige_noise_args a, int64_t n) {
  __shared__ int s_dec[GROUP_TILES][TILE];
  __shared__ int s_rr[GROUP_TILES][TILE];
  __shared__ int s_ir[GROUP_TILES][TILE];
  __shared__ int lds8[2 * GROUP_TILES];
  const int64_t g = blockIdx.x;
  const int w = threadIdx.x >> 6, lane = lane_id();
  const int64_t env = g * GROUP + threadIdx.x;
  const bool live = env < n;
  const NoiseDecision d = noise_decide(a, env, live);
  if (live) a.counter[env] = d.cnt;
  const int64_t G = num_groups(n);
  (void)G;
  int rr = 0, ir = 0;
  if (a.reset_samples) rr = (int)group_rank(d.reset, a.reset_workspace + padded_tiles(n), g, lds8);
  if (a.interval_samples) ir = (int)group_rank(d.interval, a.interval_workspace + padded_tiles(n), g, lds8 + GROUP_TILES);
  s_dec[w][lane] = (d.reset ? 1 : 0) | (d.interval ? 2 : 0);
  s_rr[w][lane] = rr;
  s_ir[w][lane] = ir;
  __syncthreads();

  const int64_t env0 = (g * GROUP_TILES + w) * TILE;
  if (env0 >= n) return;
  const int W = a.width;
  const int nv = (n - env0) < TILE ? (int)(n - env0) : TILE;
  const int total = nv * W;
  const int64_t base = env0 * W;  // multiple of 64 floats: float4 pieces never straddle tiles
  float* __restrict__ buf = a.buffer + base;
  float* __restrict__ cor = a.correlated_noise + base;
  const bool has_reset = a.on_reset.enabled != 0;
  const bool scale_r = a.on_reset.operation == OIGE_NOISE_SCALING, scale_i = a.on_interval.operation == OIGE_NOISE_SCALING;
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint64_t step = step_of(a);  // device counter: graph replay
  const uint32_t s_lo = (uint32_t)step, s_hi = (uint32_t)(step >> 32);

  for (int f = lane; 4 * f < total; f += WAVE) {
    const int e = 4 * f;
    const bool whole = e + 3 < total;
    float v[4] = {0.f, 0.f, 0.f, 0.f}, c[4] = {0.f, 0.f, 0.f, 0.f};
    if (whole) {
      const float4 t = *reinterpret_cast<const float4*>(buf + e);
      v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w;
      if (has_reset) {
        const float4 q = *reinterpret_cast<const float4*>(cor + e);
        c[0] = q.x; c[1] = q.y; c[2] = q.z; c[3] = q.w;
      }
    } else {
      for (int k = 0; e + k < total; ++k) {
        v[k] = buf[e + k];
        if (has_reset) c[k] = cor[e + k];
      }
    }
    int le = e / W, col = e - le * W;
    const int64_t gf = (base + e) >> 2;
    bool have_r = false, have_i = false, dirty = false;
    float sr[4], si[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      if (e + k < total) {
        const int dec = s_dec[w][le];
        if (has_reset) {
          if (dec & 1) {  // correlated_noise_buffer[reset_ids] = sample  (:379-400)
            if (a.reset_samples) {
              c[k] = a.reset_samples[(int64_t)s_rr[w][le] * W + col];
            } else {
              if (!have_r) {
                const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(4 + 2 * a.stream_id) << 16), s_lo, s_hi};
                noise_samples(a.on_reset, philox4x32_10(cc, k0, k1), sr);
                have_r = true;
              }
              c[k] = sr[k];
            }
            dirty = true;
          }
          v[k] = scale_r ? v[k] * c[k] : v[k] + c[k];  // buffer (+|*)= correlated_noise_buffer, all envs (:404-407)
        }
        if (dec & 2) {  // buffer[randomize_ids] (+|*)= noise  (:365-368)
          float s;
          if (a.interval_samples) {
            s = a.interval_samples[(int64_t)s_ir[w][le] * W + col];
          } else {
            if (!have_i) {
              const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(5 + 2 * a.stream_id) << 16), s_lo, s_hi};
              noise_samples(a.on_interval, philox4x32_10(cc, k0, k1), si);
              have_i = true;
            }
            s = si[k];
          }
          v[k] = scale_i ? v[k] * s : v[k] + s;
        }
      }
      if (++col == W) { col = 0; ++le; }
    }
    if (whole) {
      *reinterpret_cast<float4*>(buf + e) = make_float4(v[0], v[1], v[2], v[3]);
      if (dirty) *reinterpret_cast<float4*>(cor + e) = make_float4(c[0], c[1], c[2], c[3]);
      if (a.clamped_out)
        *reinterpret_cast<float4*>(a.clamped_out + base + e) =
            make_float4(clamp_sym(v[0], a.clip), clamp_sym(v[1], a.clip), clamp_sym(v[2], a.clip), clamp_sym(v[3], a.clip));
    } else {
      for (int k = 0; e + k < total; ++k) {
        buf[e + k] = v[k];
        if (dirty) cor[e + k] = c[k];
        if (a.clamped_out) a.clamped_out[base + e + k] = clamp_sym(v[k], a.clip);
      }
    }
  }
}

static int noise_validate(const oige_noise_args* a, int64_t n, const char* who) {
  OIGE_REQUIRE(a, "%s: null args", who);
  OIGE_REQUIRE(n >= 0 && n < ((int64_t)1 << 31) / 256, "%s: num_envs %lld out of range", who, (long long)n);
  OIGE_REQUIRE(a->width >= 1 && (int64_t)a->width * TILE < ((int64_t)1 << 30), "%s: bad width %d", who, a->width);
  OIGE_REQUIRE(a->buffer && a->reset_buf && a->counter, "%s: null buffer / reset_buf / counter", who);
  OIGE_REQUIRE(!a->on_reset.enabled || a->correlated_noise, "%s: on_reset needs the correlated noise buffer", who);
  OIGE_REQUIRE(aligned16(a->buffer) && aligned16(a->correlated_noise) && aligned16(a->clamped_out),
               "%s: buffers must be 16-byte aligned", who);
  const oige_noise_term* terms[2] = {&a->on_reset, &a->on_interval};
  for (const oige_noise_term* t : terms) {
    if (!t->enabled) continue;
    OIGE_REQUIRE(t->operation == OIGE_NOISE_ADDITIVE || t->operation == OIGE_NOISE_SCALING, "%s: unknown operation %d", who,
                 t->operation);
    OIGE_REQUIRE(t->distribution >= OIGE_NOISE_GAUSSIAN && t->distribution <= OIGE_NOISE_LOGUNIFORM,
                 "%s: unknown distribution %d", who, t->distribution);
  }
  OIGE_REQUIRE(!(a->reset_samples || a->interval_samples) ||
                   (a->reset_workspace && a->interval_workspace && aligned16(a->reset_workspace) && aligned16(a->interval_workspace)),
               "%s: pre-drawn samples need both rank workspaces (16-byte aligned), filled by oige_dr_noise_count", who);
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

int oige_dr_noise_count(const oige_noise_args* args, int64_t num_envs, oige_stream_t stream) {
  if (int rc = noise_validate(args, num_envs, "dr_noise_count")) return rc;
  OIGE_REQUIRE(args->reset_workspace && args->interval_workspace && aligned16(args->reset_workspace) &&
                   aligned16(args->interval_workspace),
               "dr_noise_count: null / unaligned workspace");
  hipStream_t st = (hipStream_t)stream;
  const int64_t G = num_groups(num_envs);
  if (G > 0) {
    hipLaunchKernelGGL(dr_noise_count_kernel, dim3((unsigned)G), dim3(GROUP), 0, st, *args, num_envs);
    if (int rc = check_launch("dr_noise_count_kernel")) return rc;
  }
  return launch_scan_groups2(num_envs, args->reset_workspace, args->interval_workspace, st);
}

int oige_dr_noise_apply(const oige_noise_args* args, int64_t num_envs, oige_stream_t stream) {
  if (int rc = noise_validate(args, num_envs, "dr_noise_apply")) return rc;
  const int64_t G = num_groups(num_envs);
  if (G == 0) return OIGE_OK;
  hipLaunchKernelGGL(dr_noise_apply_kernel, dim3((unsigned)G), dim3(GROUP), 0, (hipStream_t)stream, *args, num_envs);
  return check_launch("dr_noise_apply_kernel");
}

}  // extern "C"
