// Observation / action noise of the reference's domain randomisation (randomize.py:283-410): the pieces shared by the
// stand-alone noise kernel (dr_noise.hip) and the in-hand kernels that apply the noise in their own epilogue /
// prologue (hand.hip).  Both produce bit-identical values: same decisions, same Philox keys
// (seed, step, flat float4 index of the buffer, stream), same arithmetic.
#pragma once
#include "oige_common.hpp"
#include "oige_math.hpp"

namespace oige {

struct NoiseDecision {
  bool reset, interval;
  int cnt;
};

// counter logic of :284-286 and :299-304 for one env
// (`reset` given by the caller: the fused kernels hold the flag in a register)
// `cnt` = counter[env] as the caller fetched it (kernels that store before they get here fetch it with their inputs: a
// load issued after a store waits, through vmcnt, for that store's round trip to HBM)
__device__ __forceinline__ NoiseDecision noise_decide_cnt(const oige_noise_args& a, int cnt, bool live, bool reset) {
  NoiseDecision d{false, false, 0};
  if (!live) return d;
  d.reset = reset;
  if (d.reset) cnt = 0;
  cnt += 1;
  d.interval = a.on_interval.enabled && cnt >= a.on_interval.frequency_interval;
  if (d.interval) cnt = 0;
  d.cnt = cnt;
  return d;
}

__device__ __forceinline__ NoiseDecision noise_decide_flag(const oige_noise_args& a, int64_t env, bool live, bool reset) {
  NoiseDecision d{false, false, 0};
  if (!live) return d;
  d.reset = reset;
  int cnt = a.counter[env];
  if (d.reset) cnt = 0;
  cnt += 1;
  d.interval = a.on_interval.enabled && cnt >= a.on_interval.frequency_interval;
  if (d.interval) cnt = 0;
  d.cnt = cnt;
  return d;
}

__device__ __forceinline__ NoiseDecision noise_decide(const oige_noise_args& a, int64_t env, bool live) {
  return noise_decide_flag(a, env, live, live && a.reset_buf[env] != 0);
}

// four samples of one distribution from one Philox block
__device__ __forceinline__ void noise_samples(const oige_noise_term& t, const u4 x, float out[4]) {
  const float u[4] = {u01(x.x), u01(x.y), u01(x.z), u01(x.w)};
  if (t.distribution == OIGE_NOISE_GAUSSIAN) {  // Box-Muller, two pairs; 1 - u in (0, 1]
    // Hardware transcendentals (v_log_f32 = log2, v_sin/v_cos_f32 take REVOLUTIONS, so the angle 2*pi*u needs no
    // multiply or range reduction): the precise libm calls made this kernel VALU-bound at 1.6 TB/s.  These
    // samples are distribution-parity only (torch's RNG cannot be replayed in-kernel); they agree with the numpy
    // restatement (oracle/dr_noise_oracle.py) to ~1e-6 relative.
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const float r = __builtin_amdgcn_sqrtf((-2.0f * 0.69314718055994530942f) * __builtin_amdgcn_logf(1.0f - u[2 * h]));
      out[2 * h] = (r * __builtin_amdgcn_cosf(u[2 * h + 1])) * t.p1 + t.p0;
      out[2 * h + 1] = (r * __builtin_amdgcn_sinf(u[2 * h + 1])) * t.p1 + t.p0;
    }
  } else if (t.distribution == OIGE_NOISE_UNIFORM) {  // (hi - lo) * rand + lo  (:353-355)
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = t.p1 * u[k] + t.p0;
  } else {  // exp((log hi - log lo) * rand + log lo)  (:357-361)
#pragma unroll
    for (int k = 0; k < 4; ++k) out[k] = __expf(t.p1 * u[k] + t.p0);
  }
}

// Noise of four consecutive flat elements (in-kernel Philox path): v = the buffer values, c = their correlated-noise
// values, dec[k] = decision bits of element k's env (1 reset, 2 interval), gf = flat float4 index of the quad in the
// (N,W) buffer.  Returns true when c changed (a reset env refreshed its correlated noise, :379-400).
__device__ __forceinline__ bool noise_apply4(const oige_noise_args& a, const int (&dec)[4], int nvalid, int64_t gf, float (&v)[4],
                                             float (&c)[4]) {
  const bool has_reset = a.on_reset.enabled != 0;
  const bool scale_r = a.on_reset.operation == OIGE_NOISE_SCALING, scale_i = a.on_interval.operation == OIGE_NOISE_SCALING;
  const uint32_t k0 = (uint32_t)a.seed, k1 = (uint32_t)(a.seed >> 32);
  const uint64_t step = step_of(a);  // device counter: graph replay
  const uint32_t s_lo = (uint32_t)step, s_hi = (uint32_t)(step >> 32);
  bool have_r = false, have_i = false, dirty = false;
  float sr[4], si[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= nvalid) continue;
    if (has_reset) {
      if (dec[k] & 1) {  // correlated_noise_buffer[reset_ids] = sample  (:379-400)
        if (!have_r) {
          const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(4 + 2 * a.stream_id) << 16), s_lo, s_hi};
          noise_samples(a.on_reset, philox4x32_10(cc, k0, k1), sr);
          have_r = true;
        }
        c[k] = sr[k];
        dirty = true;
      }
      v[k] = scale_r ? v[k] * c[k] : v[k] + c[k];  // buffer (+|*)= correlated_noise_buffer, all envs (:404-407)
    }
    if (dec[k] & 2) {  // buffer[randomize_ids] (+|*)= noise  (:365-368)
      if (!have_i) {
        const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(5 + 2 * a.stream_id) << 16), s_lo, s_hi};
        noise_samples(a.on_interval, philox4x32_10(cc, k0, k1), si);
        have_i = true;
      }
      v[k] = scale_i ? v[k] * si[k] : v[k] + si[k];
    }
  }
  return dirty;
}

// The two halves of noise_apply4 for kernels that refresh the correlated noise of reset envs on COMPACTED rows first
// (a wave no longer runs the second Philox + Box-Muller because one of its lanes holds a reset env):
// noise_reset_samples = the four fresh correlated-noise samples of flat float4 `gf` (same key as noise_apply4);
// noise_apply4_refreshed = noise_apply4 for a quad whose reset elements (dec & 1) already carry their fresh sample in v.
__device__ __forceinline__ void noise_reset_samples(const oige_noise_args& a, int64_t gf, float (&sr)[4]) {
  const uint64_t step = step_of(a);
  const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(4 + 2 * a.stream_id) << 16), (uint32_t)step, (uint32_t)(step >> 32)};
  noise_samples(a.on_reset, philox4x32_10(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32)), sr);
}

__device__ __forceinline__ void noise_apply4_refreshed(const oige_noise_args& a, const int (&dec)[4], int nvalid, int64_t gf,
                                                       float (&v)[4], const float (&c)[4]) {
  const bool has_reset = a.on_reset.enabled != 0;
  const bool scale_r = a.on_reset.operation == OIGE_NOISE_SCALING, scale_i = a.on_interval.operation == OIGE_NOISE_SCALING;
  const uint64_t step = step_of(a);
  bool have_i = false;
  float si[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= nvalid) continue;
    if (has_reset && !(dec[k] & 1)) v[k] = scale_r ? v[k] * c[k] : v[k] + c[k];
    if (dec[k] & 2) {
      if (!have_i) {
        const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(5 + 2 * a.stream_id) << 16), (uint32_t)step, (uint32_t)(step >> 32)};
        noise_samples(a.on_interval, philox4x32_10(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32)), si);
        have_i = true;
      }
      v[k] = scale_i ? v[k] * si[k] : v[k] + si[k];
    }
  }
}

// noise_apply4 with the uncorrelated ("interval") samples of the quad drawn AHEAD by noise_interval_samples: they depend on
// nothing a kernel loads (key = seed, step, flat float4 index, stream), so a kernel can compute them between issuing its
// loads and its first wait, where the Philox + Box-Muller arithmetic costs no time.  Same key, same arithmetic, same bits.
__device__ __forceinline__ void noise_interval_samples(const oige_noise_args& a, int64_t gf, float (&si)[4]) {
  const uint64_t step = step_of(a);
  const u4 cc{(uint32_t)gf, (uint32_t)(gf >> 32) | ((uint32_t)(5 + 2 * a.stream_id) << 16), (uint32_t)step, (uint32_t)(step >> 32)};
  noise_samples(a.on_interval, philox4x32_10(cc, (uint32_t)a.seed, (uint32_t)(a.seed >> 32)), si);
}

__device__ __forceinline__ bool noise_apply4_drawn(const oige_noise_args& a, const int (&dec)[4], int nvalid, int64_t gf, float (&v)[4],
                                                   float (&c)[4], const float (&si)[4]) {
  const bool has_reset = a.on_reset.enabled != 0;
  const bool scale_r = a.on_reset.operation == OIGE_NOISE_SCALING, scale_i = a.on_interval.operation == OIGE_NOISE_SCALING;
  bool have_r = false, dirty = false;
  float sr[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    if (k >= nvalid) continue;
    if (has_reset) {
      if (dec[k] & 1) {  // correlated_noise_buffer[reset_ids] = sample  (:379-400)
        if (!have_r) {
          noise_reset_samples(a, gf, sr);
          have_r = true;
        }
        c[k] = sr[k];
        dirty = true;
      }
      v[k] = scale_r ? v[k] * c[k] : v[k] + c[k];  // buffer (+|*)= correlated_noise_buffer, all envs (:404-407)
    }
    if (dec[k] & 2) v[k] = scale_i ? v[k] * si[k] : v[k] + si[k];  // buffer[randomize_ids] (+|*)= noise  (:365-368)
  }
  return dirty;
}

// what a fused kernel may be handed: Philox samples only, the buffer and width it expects
inline int noise_validate_fused(const oige_noise_args* a, const void* buffer, int width, const char* who) {
  OIGE_REQUIRE(a->buffer == buffer && a->width == width, "%s: noise args must address the kernel's own (N,%d) buffer", who, width);
  OIGE_REQUIRE(a->counter, "%s: null noise counter", who);
  OIGE_REQUIRE(!a->on_reset.enabled || a->correlated_noise, "%s: on_reset needs the correlated noise buffer", who);
  OIGE_REQUIRE(!a->reset_samples && !a->interval_samples,
               "%s: pre-drawn samples go through oige_dr_noise_count / oige_dr_noise_apply", who);
  OIGE_REQUIRE(aligned16(a->correlated_noise) && aligned16(a->clamped_out), "%s: noise buffers must be 16-byte aligned", who);
  const oige_noise_term* terms[2] = {&a->on_reset, &a->on_interval};
  for (const oige_noise_term* t : terms) {
    if (!t->enabled) continue;
    OIGE_REQUIRE(t->operation == OIGE_NOISE_ADDITIVE || t->operation == OIGE_NOISE_SCALING, "%s: unknown operation %d", who,
                 t->operation);
    OIGE_REQUIRE(t->distribution >= OIGE_NOISE_GAUSSIAN && t->distribution <= OIGE_NOISE_LOGUNIFORM,
                 "%s: unknown distribution %d", who, t->distribution);
  }
  return OIGE_OK;
}

}  // namespace oige
