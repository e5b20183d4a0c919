// Exclusive scan of the per-group reset counts by ONE workgroup of SCAN_THREADS threads (thread i owns a contiguous run of
// groups): shared by scan_groups_kernel (oige_core.hip) and kernels that give the scan one of their workgroups.
#pragma once
#include "oige_common.hpp"

namespace oige {

constexpr int SCAN_THREADS = 1024;

__device__ __forceinline__ void scan_groups_body(const int32_t* __restrict__ tile_counts, int64_t n_groups,
                                                 int32_t* __restrict__ group_offsets, int* wave_tot /* [SCAN_THREADS / WAVE] in LDS */) {
  const int tid = threadIdx.x;
  const int64_t per = (n_groups + SCAN_THREADS - 1) / SCAN_THREADS;
  const int64_t g0 = (int64_t)tid * per;
  const int64_t g1 = g0 + per < n_groups ? g0 + per : n_groups;
  const int4* c4 = reinterpret_cast<const int4*>(tile_counts);  // GROUP_TILES == 4 counts per group
  int mine = 0;
  for (int64_t g = g0; g < g1; ++g) {
    const int4 c = c4[g];
    mine += (c.x + c.y) + (c.z + c.w);
  }
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
  for (int64_t g = g0; g < g1; ++g) {
    group_offsets[g] = run;
    const int4 c = c4[g];
    run += (c.x + c.y) + (c.z + c.w);
  }
  if (tid == SCAN_THREADS - 1) {
    int total = 0;
    for (int i = 0; i < SCAN_THREADS / WAVE; ++i) total += wave_tot[i];
    group_offsets[n_groups] = total;
  }
}

}  // namespace oige
