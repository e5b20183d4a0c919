// Wave-tile staging: one wavefront owns a tile of 64 consecutive envs.
//
// The views hand over row-major (N,K) arrays.  A tile's slice of such an array is ONE
// contiguous run of 64*K floats, so the wave moves it with lane-linear 16-byte accesses
// (64 lanes x 16 B = 1 KiB per instruction, fully coalesced) and transposes through LDS,
// instead of each lane walking its own K-float row at a K*4-byte stride.
#pragma once
#include "oige_common.hpp"

namespace oige {

// Streaming 16-byte global accesses: every tile run is read once or written once per step, so the loads and stores carry
// the non-temporal hint (`global_load/store_dwordx4 ... nt`).  Measured on the Ant post kernel at 1 M envs: 113 us plain,
// 101 us nt loads, 102 us nt stores, 92.6 us both (5.15 -> 6.3 TB/s).  -DOIGE_PLAIN_LOAD / -DOIGE_PLAIN_STORE: A/B builds.
typedef float v4f_native __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float4 ld4_stream(const float* p) {
#ifndef OIGE_PLAIN_LOAD
  const v4f_native v = __builtin_nontemporal_load(reinterpret_cast<const v4f_native*>(p));
  return make_float4(v.x, v.y, v.z, v.w);
#else
  return *reinterpret_cast<const float4*>(p);
#endif
}
__device__ __forceinline__ void st4_stream(float* p, const float4 t) {
#ifndef OIGE_PLAIN_STORE
  v4f_native v;
  v.x = t.x; v.y = t.y; v.z = t.z; v.w = t.w;
  __builtin_nontemporal_store(v, reinterpret_cast<v4f_native*>(p));
#else
  *reinterpret_cast<float4*>(p) = t;
#endif
}


// NT = threads that share the tile (64: one wave; 128 / 256: the waves of one workgroup split every run)
template <int K, int NT = WAVE>
struct Chunk {
  static constexpr int NF4 = (TILE * K) / 4;          // float4 pieces in a full tile (TILE*K % 4 == 0)
  static constexpr int ITERS = (NF4 + NT - 1) / NT;   // per-thread float4 count
  static constexpr bool EXACT = (NF4 % NT) == 0;      // no thread masking needed on a full tile
  float4 v[ITERS];
};

// Load the tile's contiguous run.  `src` points at the tile's first float and is 16-byte
// aligned (tile start = 64*K floats = multiple of 256 B past a 16-byte aligned base).
// `valid` = number of floats that exist (64*K on a full tile).
template <int K, bool FULL, int NT = WAVE>
__device__ __forceinline__ void load_chunk(Chunk<K, NT>& c, const float* __restrict__ src, int valid) {
  const int lane = NT == WAVE ? lane_id() : (int)threadIdx.x;
#pragma unroll
  for (int it = 0; it < Chunk<K, NT>::ITERS; ++it) {
    const int f = it * NT + lane;
    const int e = 4 * f;
    if (FULL) {
      if (Chunk<K, NT>::EXACT || f < Chunk<K, NT>::NF4)
        c.v[it] = ld4_stream(src + e);
      else
        c.v[it] = make_float4(0.f, 0.f, 0.f, 0.f);
    } else {
      float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
      if (e + 3 < valid) {
        t = ld4_stream(src + e);
      } else {
        if (e + 0 < valid) t.x = src[e + 0];
        if (e + 1 < valid) t.y = src[e + 1];
        if (e + 2 < valid) t.z = src[e + 2];
      }
      c.v[it] = t;
    }
  }
}

// Scatter a loaded run into the (64 x ROW) LDS tile at column COL0, multiplying by `s`
// (MUL=false: plain copy).  Element e of the run belongs to env e / K, column e % K.
template <int K, int ROW, int COL0, bool MUL, int NT = WAVE>
__device__ __forceinline__ void scatter_chunk(const Chunk<K, NT>& c, float* __restrict__ tile, float s) {
  const int lane = NT == WAVE ? lane_id() : (int)threadIdx.x;
#pragma unroll
  for (int it = 0; it < Chunk<K, NT>::ITERS; ++it) {
    const int f = it * NT + lane;
    if (!Chunk<K, NT>::EXACT && f >= Chunk<K, NT>::NF4) continue;
    const int e = 4 * f;
    float4 t = c.v[it];
    if (MUL) { t.x *= s; t.y *= s; t.z *= s; t.w *= s; }
    if ((K % 4 == 0) && (ROW % 4 == 0) && (COL0 % 4 == 0)) {
      const int env = e / K, col = e % K;  // a float4 never straddles rows when K % 4 == 0
      *reinterpret_cast<float4*>(tile + env * ROW + COL0 + col) = t;
    } else {
      const float tv[4] = {t.x, t.y, t.z, t.w};
      int row = e / K, col = e - row * K;  // one division per float4, then a cursor (Humanoid: K = 21, 1200 float4 per tile)
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        tile[row * ROW + COL0 + col] = tv[k];
        if (++col == K) { col = 0; ++row; }
      }
    }
  }
}

// Store the assembled (64 x ROW) LDS tile as one contiguous run of `valid` floats.
template <int ROW, bool FULL, int NT = WAVE>
__device__ __forceinline__ void store_tile(const float* __restrict__ tile, float* __restrict__ dst, int valid) {
  constexpr int NF4 = (TILE * ROW) / 4;
  constexpr int ITERS = (NF4 + NT - 1) / NT;
  const int lane = NT == WAVE ? lane_id() : (int)threadIdx.x;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int f = it * NT + lane;
    const int e = 4 * f;
    if (FULL) {
      if ((NF4 % NT) == 0 || f < NF4)
        st4_stream(dst + e, *reinterpret_cast<const float4*>(tile + e));
    } else if (f < NF4) {
      if (e + 3 < valid) {
        st4_stream(dst + e, *reinterpret_cast<const float4*>(tile + e));
      } else {
        if (e + 0 < valid) dst[e + 0] = tile[e + 0];
        if (e + 1 < valid) dst[e + 1] = tile[e + 1];
        if (e + 2 < valid) dst[e + 2] = tile[e + 2];
      }
    }
  }
}

}  // namespace oige

// ------------------------------------------------------------------------------------------
// Workgroup-cooperative variants: NT threads (NT/64 waves) share ONE 64-env tile.  Used by the
// wide-row tasks (ShadowHand 187-float state rows, AnymalTerrain 188-float obs rows) where a
// tile is ~48 KB of LDS: four waves per tile keep 8-12 waves per CU resident instead of 2-3.
namespace oige {

// TE = envs per pass (64 = a whole tile; 32 = half a tile, which halves the LDS a workgroup needs).
template <int K, int NT, int TE = TILE>
struct WgChunk {
  static constexpr int NF4 = (TE * K) / 4;
  static constexpr int ITERS = (NF4 + NT - 1) / NT;
  static_assert((TE * K) % 4 == 0, "a pass's run must be a whole number of float4");
  float4 v[ITERS];
};

template <int K, int NT, bool FULL, int TE = TILE>
__device__ __forceinline__ void wg_load_chunk(WgChunk<K, NT, TE>& c, const float* __restrict__ src, int valid,
                                              const int tid = threadIdx.x) {
#pragma unroll
  for (int it = 0; it < WgChunk<K, NT, TE>::ITERS; ++it) {
    const int f = it * NT + tid;
    const int e = 4 * f;
    float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
    if (f < WgChunk<K, NT, TE>::NF4) {
      if (FULL || e + 3 < valid) {
        t = ld4_stream(src + e);
      } else {
        if (e + 0 < valid) t.x = src[e + 0];
        if (e + 1 < valid) t.y = src[e + 1];
        if (e + 2 < valid) t.z = src[e + 2];
      }
    }
    c.v[it] = t;
  }
}

// fn(env_in_pass, column, value) for every element of the run this thread holds
template <int K, int NT, int TE = TILE, typename F>
__device__ __forceinline__ void wg_for_each(const WgChunk<K, NT, TE>& c, F&& fn, const int tid = threadIdx.x) {
#pragma unroll
  for (int it = 0; it < WgChunk<K, NT, TE>::ITERS; ++it) {
    const int f = it * NT + tid;
    if (f >= WgChunk<K, NT, TE>::NF4) continue;
    const int e = 4 * f;
    const float tv[4] = {c.v[it].x, c.v[it].y, c.v[it].z, c.v[it].w};
    // one division per float4, not per element: the scatter address arithmetic was a third of the wide-row kernels'
    // VALU instructions.  K % 4 == 0: a float4 never straddles two rows.
    int row = e / K, col = e - row * K;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      fn(row, col, tv[k]);
      if (K % 4 == 0) {
        ++col;
      } else if (++col == K) {
        col = 0;
        ++row;
      }
    }
  }
}

template <int ROW, int NT, bool FULL, int TE = TILE>
__device__ __forceinline__ void wg_store_tile(const float* __restrict__ tile, float* __restrict__ dst, int valid,
                                              const int tid = threadIdx.x) {
  constexpr int NF4 = (TE * ROW) / 4;
  static_assert((TE * ROW) % 4 == 0, "a pass's output run must be a whole number of float4");
  constexpr int ITERS = (NF4 + NT - 1) / NT;
#pragma unroll
  for (int it = 0; it < ITERS; ++it) {
    const int f = it * NT + tid;
    const int e = 4 * f;
    if (f >= NF4) continue;
    if (FULL || e + 3 < valid) {
      st4_stream(dst + e, *reinterpret_cast<const float4*>(tile + e));
    } else {
      if (e + 0 < valid) dst[e + 0] = tile[e + 0];
      if (e + 1 < valid) dst[e + 1] = tile[e + 1];
      if (e + 2 < valid) dst[e + 2] = tile[e + 2];
    }
  }
}

}  // namespace oige
