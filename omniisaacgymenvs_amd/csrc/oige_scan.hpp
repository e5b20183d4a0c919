// Exclusive scan of the per-group reset counts by ONE workgroup of SCAN_THREADS threads (thread i owns a contiguous run of
// groups): shared by scan_groups_kernel (oige_core.hip) and kernels that give the scan one of their workgroups.
#pragma once
#include "oige_common.hpp"

namespace oige {

constexpr int SCAN_THREADS = 1024;

// sum of a group's four tile counts; slots past the last real tile (the padding of the last group) count as zero whatever
// they hold, so the scan does not depend on anyone having zeroed them
__device__ __forceinline__ int group_count(const int4* __restrict__ c4, int64_t g, int64_t n_tiles) {
  const int4 c = c4[g];
  const int64_t t0 = g * GROUP_TILES;
  if (t0 + GROUP_TILES <= n_tiles) return (c.x + c.y) + (c.z + c.w);
  return ((t0 < n_tiles ? c.x : 0) + (t0 + 1 < n_tiles ? c.y : 0)) + ((t0 + 2 < n_tiles ? c.z : 0) + (t0 + 3 < n_tiles ? c.w : 0));
}

__device__ __forceinline__ void scan_groups_body(const int32_t* __restrict__ tile_counts, int64_t n_groups, int64_t n_tiles,
                                                 int32_t* __restrict__ group_offsets, int* wave_tot /* [SCAN_THREADS / WAVE] in LDS */) {
  const int tid = threadIdx.x;
  const int64_t per = (n_groups + SCAN_THREADS - 1) / SCAN_THREADS;
  const int64_t g0 = (int64_t)tid * per;
  const int64_t g1 = g0 + per < n_groups ? g0 + per : n_groups;
  const int4* c4 = reinterpret_cast<const int4*>(tile_counts);  // GROUP_TILES == 4 counts per group
  // the first KEEP groups of the run are fetched together (independent loads, one latency) and kept in registers for the
  // second pass; up to 4096 groups (1 M envs) that is the whole run
  constexpr int KEEP = 4;
  int first[KEEP];
#pragma unroll
  for (int j = 0; j < KEEP; ++j) first[j] = g0 + j < g1 ? group_count(c4, g0 + j, n_tiles) : 0;
  int mine = (first[0] + first[1]) + (first[2] + first[3]);
  for (int64_t g = g0 + KEEP; g < g1; ++g) mine += group_count(c4, g, n_tiles);
  // inclusive scan of `mine` across the wave, then across waves
  int incl = mine;
#pragma unroll
  for (int off = 1; off < WAVE; off <<= 1) {
    const int up = __shfl_up(incl, off, WAVE);
    if (lane_id() >= off) incl += up;
  }
  const int w = tid >> 6;
  if (lane_id() == WAVE - 1) wave_tot[w] = incl;
  __syncthreads();
  int base = 0;
  for (int i = 0; i < w; ++i) base += wave_tot[i];
  int run = base + incl - mine;  // exclusive prefix at g0
#pragma unroll
  for (int j = 0; j < KEEP; ++j) {
    if (g0 + j < g1) group_offsets[g0 + j] = run;
    run += first[j];
  }
  for (int64_t g = g0 + KEEP; g < g1; ++g) {
    group_offsets[g] = run;
    run += group_count(c4, g, n_tiles);
  }
  if (tid == SCAN_THREADS - 1) {
    int total = 0;
    for (int i = 0; i < SCAN_THREADS / WAVE; ++i) total += wave_tot[i];
    group_offsets[n_groups] = total;
  }
}

}  // namespace oige
