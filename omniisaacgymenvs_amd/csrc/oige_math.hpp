// Device-side helper math for the per-step task kernels (gfx950).
//
// These restate the omni.isaac.core.utils.torch.{maths,rotations} helpers the reference's
// hot path calls (call sites cited per function; the helper package itself is not in the
// reference tree).  Every function evaluates in the helper's own operation
// order so fp32 rounding follows the reference: the translation unit is compiled with
// -ffp-contract=off, and an fma appears only where torch's CPU kernel fuses
// (torch.cross, torch.norm over a short row).  Quaternions are wxyz.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace oige {

struct f3 { float x, y, z; };
struct q4 { float w, x, y, z; };

#define OIGE_DEV __device__ __forceinline__

// torch.cross: component = fma(a_i, b_j, -(a_j * b_i))  (ATen cross kernel contracts this way)
OIGE_DEV f3 cross(const f3 a, const f3 b) {
  f3 r;
  r.x = __builtin_fmaf(a.y, b.z, -(a.z * b.y));
  r.y = __builtin_fmaf(a.z, b.x, -(a.x * b.z));
  r.z = __builtin_fmaf(a.x, b.y, -(a.y * b.x));
  return r;
}

// (n,1,3)x(n,3,1) bmm as the helpers spell a row-wise dot: ((p0 + p1) + p2), unfused
OIGE_DEV float dot3(const f3 a, const f3 b) { return (a.x * b.x + a.y * b.y) + a.z * b.z; }

// torch.norm(p=2, dim=-1) over a 3-vector: acc = fma(d, d, acc) left to right, then sqrt
OIGE_DEV float norm3(const f3 a) {
  float acc = a.x * a.x;
  acc = __builtin_fmaf(a.y, a.y, acc);
  acc = __builtin_fmaf(a.z, a.z, acc);
  return __builtin_sqrtf(acc);
}
OIGE_DEV float norm4(const q4 a) {  // 4-wide rows take torch's unfused path
  return __builtin_sqrtf(((a.w * a.w + a.x * a.x) + a.y * a.y) + a.z * a.z);
}

// normalize(x, eps=1e-9): x / clamp(||x||, min=eps)   (anymal_terrain.py:611; compute_heading_and_up)
OIGE_DEV f3 normalize3(const f3 a) {
  float n = norm3(a);
  n = n < 1e-9f ? 1e-9f : n;
  return f3{a.x / n, a.y / n, a.z / n};
}

// quat_mul, Isaac's factored form (factory_control.py:216-221 fixes the convention)
OIGE_DEV q4 quat_mul(const q4 a, const q4 b) {
  const float ww = (a.z + a.x) * (b.x + b.y);
  const float yy = (a.w - a.y) * (b.w + b.z);
  const float zz = (a.w + a.y) * (b.w - b.z);
  const float xx = (ww + yy) + zz;
  const float qq = 0.5f * (xx + (a.z - a.x) * (b.x - b.y));
  q4 r;
  r.w = (qq - ww) + (a.z - a.y) * (b.y - b.z);
  r.x = (qq - xx) + (a.x + a.w) * (b.x + b.w);
  r.y = (qq - yy) + (a.w - a.x) * (b.y + b.z);
  r.z = (qq - zz) + (a.z + a.y) * (b.w - b.x);
  return r;
}

OIGE_DEV q4 quat_conjugate(const q4 a) { return q4{a.w, -a.x, -a.y, -a.z}; }  // locomotion.py:171

// quat_rotate / quat_rotate_inverse: a = v(2w^2-1); b = 2 w (qv x v); c = 2 qv (qv . v); a (+|-) b + c
template <bool INVERSE>
OIGE_DEV f3 quat_rotate_t(const q4 q, const f3 v) {
  const f3 qv{q.x, q.y, q.z};
  const float s = 2.0f * (q.w * q.w) - 1.0f;
  const f3 cr = cross(qv, v);
  const float d = dot3(qv, v);
  f3 r;
  const float ax = v.x * s, ay = v.y * s, az = v.z * s;
  const float bx = (cr.x * q.w) * 2.0f, by = (cr.y * q.w) * 2.0f, bz = (cr.z * q.w) * 2.0f;
  const float cx = (qv.x * d) * 2.0f, cy = (qv.y * d) * 2.0f, cz = (qv.z * d) * 2.0f;
  if (INVERSE) {
    r.x = (ax - bx) + cx; r.y = (ay - by) + cy; r.z = (az - bz) + cz;
  } else {
    r.x = (ax + bx) + cx; r.y = (ay + by) + cy; r.z = (az + bz) + cz;
  }
  return r;
}
OIGE_DEV f3 quat_rotate(const q4 q, const f3 v) { return quat_rotate_t<false>(q, v); }
OIGE_DEV f3 quat_rotate_inverse(const q4 q, const f3 v) { return quat_rotate_t<true>(q, v); }  // anymal_terrain.py:422-424

// quat_apply: t = 2 (qv x v); v + w t + qv x t   (anymal_terrain.py:425,612)
OIGE_DEV f3 quat_apply(const q4 q, const f3 v) {
  const f3 qv{q.x, q.y, q.z};
  f3 t = cross(qv, v);
  t.x *= 2.0f; t.y *= 2.0f; t.z *= 2.0f;
  const f3 c2 = cross(qv, t);
  return f3{(v.x + q.w * t.x) + c2.x, (v.y + q.w * t.y) + c2.y, (v.z + q.w * t.z) + c2.z};
}

// torch.remainder(a, b) for b > 0: fmod, then shift negatives up by b
OIGE_DEV float remainder_pos(const float a, const float b) {
  float m = __builtin_fmodf(a, b);
  if (m != 0.0f && m < 0.0f) m += b;
  return m;
}

#define OIGE_PI_F 3.14159265358979323846f
#define OIGE_TWO_PI_F 6.28318530717958647692f
#define OIGE_HALF_PI_F 1.57079632679489661923f

// get_euler_xyz (formulas: tasks/warp/shared/locomotion.py:253-276), each angle % 2*pi
OIGE_DEV void get_euler_xyz(const q4 q, float& roll, float& pitch, float& yaw) {
  const float sinr_cosp = 2.0f * (q.w * q.x + q.y * q.z);
  const float cosr_cosp = ((q.w * q.w - q.x * q.x) - q.y * q.y) + q.z * q.z;
  const float r = atan2f(sinr_cosp, cosr_cosp);
  const float sinp = 2.0f * (q.w * q.y - q.z * q.x);
  float p;
  if (__builtin_fabsf(sinp) >= 1.0f) {
    const float sg = sinp > 0.0f ? 1.0f : (sinp < 0.0f ? -1.0f : 0.0f);  // torch.sign
    p = OIGE_HALF_PI_F * sg;
  } else {
    p = asinf(sinp);
  }
  const float siny_cosp = 2.0f * (q.w * q.z + q.x * q.y);
  const float cosy_cosp = ((q.w * q.w + q.x * q.x) - q.y * q.y) - q.z * q.z;
  const float y = atan2f(siny_cosp, cosy_cosp);
  // `% 2 pi` of an atan2 / asin result: |m| <= pi < 2 pi, so fmod(m, 2 pi) is m itself, bit for bit, and remainder_pos
  // reduces to its shift of the negatives (the library fmodf was ~30 instructions per angle for the identity)
  roll = r < 0.0f ? r + OIGE_TWO_PI_F : r;
  pitch = p < 0.0f ? p + OIGE_TWO_PI_F : p;
  yaw = y < 0.0f ? y + OIGE_TWO_PI_F : y;
}

// jit normalize_angle (locomotion.py:220-221): atan2(sin x, cos x), i.e. x reduced to (-pi, pi].  Evaluated as that
// reduction, in double (x - 2 pi rint(x / 2 pi): five instructions), instead of through sinf, cosf and atan2f (~100 each
// time, three times per env: a third of the locomotion post kernel's per-env arithmetic).  Not the reference's operation
// order, on purpose: the three angles feed no threshold (reward and done read obs[0], obs[10], obs[11] only), the result
// is the exact value rounded once -- within 1 ulp (2.4e-7) of torch's atan2(sin, cos) over [-3 pi - 1, 2 pi + 1] and on every
// float within 40 ulps of a multiple of pi, the side of the +-pi seam included (x is a float, never exactly an odd multiple
// of pi, so sin x and the reduction agree on the sign; oracle/div_check-style sweep in tests/test_helpers_known_answers.py).
OIGE_DEV float normalize_angle(const float x) {
  const double xd = (double)x;
  const double k = __builtin_rint(xd * 0.15915494309189535);  // 1 / (2 pi)
  return (float)__builtin_fma(-k, 6.283185307179586, xd);
}

// unscale / scale / tensor_clamp (locomotion.py:263,138; in_hand_manipulation.py:343-361)
OIGE_DEV float unscale(const float x, const float lo, const float hi) { return ((2.0f * x - hi) - lo) / (hi - lo); }
OIGE_DEV float scale(const float x, const float lo, const float hi) { return (0.5f * (x + 1.0f)) * (hi - lo) + lo; }
OIGE_DEV float tensor_clamp(const float t, const float lo, const float hi) {
  // torch.max(torch.min(t, hi), lo); NaN propagates like torch (min/max return NaN if either is NaN)
  float m = (t != t) ? t : (t < hi ? t : hi);
  return (m != m) ? m : (m > lo ? m : lo);
}
// torch.clamp(x, -c, c)
OIGE_DEV float clamp_sym(const float x, const float c) {
  if (x != x) return x;
  return x < -c ? -c : (x > c ? c : x);
}

// ------------------------------------------------------------------ Philox4x32-10
// Counter-based RNG for the no-host-sync reset path: key = (seed lo, seed hi),
// counter = (env id, draw slot, step lo, step hi).  Distribution parity with torch.rand
// only (torch's Philox stream cannot be replayed: SURVEY.md section 7, "RNG").
struct u4 { uint32_t x, y, z, w; };
OIGE_DEV u4 philox4x32_10(u4 c, uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c.x;
    const uint64_t p1 = (uint64_t)0xCD9E8D57u * c.z;
    u4 n;
    n.x = (uint32_t)(p1 >> 32) ^ c.y ^ k0;
    n.y = (uint32_t)p1;
    n.z = (uint32_t)(p0 >> 32) ^ c.w ^ k1;
    n.w = (uint32_t)p0;
    c = n;
    k0 += 0x9E3779B9u;
    k1 += 0xBB67AE85u;
  }
  return c;
}
// A uniform, launch-invariant 64-bit device scalar (a step counter a graph node advanced BEFORE this kernel), read through
// the constant address space: the compiler may then use a scalar load, hoist it and read it once.  As a plain global load
// every use was a conditional vector load followed by `s_waitcnt vmcnt(0)` at the join -- which also waits for every
// store the wave has in flight: in the ShadowHand noise epilogue each float4 of noise waited for the previous float4's
// stores to reach HBM, whether or not the pointer was set.
typedef const uint64_t __attribute__((address_space(4))) * oige_const_u64_ptr;
OIGE_DEV uint64_t load_uniform_u64(const uint64_t* p) { return *(oige_const_u64_ptr)(uintptr_t)p; }

// top 24 bits -> [0,1) exactly representable in fp32 (same grid torch.rand draws from)
// the Philox step counter of a reset_io block: by value, or from the device scalar a replayed HIP graph advances
template <class R>
OIGE_DEV uint64_t step_of(const R& r) { return r.step_ptr ? load_uniform_u64(r.step_ptr) : r.step; }

OIGE_DEV float u01(const uint32_t x) { return (float)(x >> 8) * 5.9604644775390625e-08f; }

}  // namespace oige
