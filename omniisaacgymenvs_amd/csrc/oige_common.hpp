// Host/device plumbing shared by every translation unit of liboige_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <stdarg.h>
#include <stdint.h>
#include <stdio.h>

#include "../../include/oige_hip.h"

namespace oige {

constexpr int WAVE = 64;              // gfx950 wavefront
constexpr int TILE = OIGE_TILE_ENVS;  // envs per wave-tile
constexpr int GROUP_TILES = 4;        // tiles per 256-env group (one pre_physics workgroup)
constexpr int GROUP = TILE * GROUP_TILES;
static_assert(TILE == WAVE, "one wavefront owns one tile of envs");

// thread-local last-error message (defined in oige_core.hip)
char* error_buffer();
constexpr size_t ERROR_BUFFER_BYTES = 512;

inline int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(error_buffer(), ERROR_BUFFER_BYTES, fmt, ap);
  va_end(ap);
  return code;
}

__host__ __device__ inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15u) == 0; }

inline int check_launch(const char* what) {
  hipError_t e = hipGetLastError();
  if (e != hipSuccess) return fail(OIGE_ELAUNCH, "%s: %s", what, hipGetErrorString(e));
  return OIGE_OK;
}

#define OIGE_REQUIRE(cond, ...)                                 \
  do {                                                          \
    if (!(cond)) return ::oige::fail(OIGE_EINVAL, __VA_ARGS__); \
  } while (0)

__host__ __device__ inline int64_t num_tiles(int64_t n) { return (n + TILE - 1) / TILE; }
__host__ __device__ inline int64_t num_groups(int64_t n) { return (n + GROUP - 1) / GROUP; }

// Reset workspace layout (int32): [ tile counts, padded to a multiple of 4 | group offsets (G+1) ].
// tile_counts[t]    = number of non-zero reset flags in envs [64t, 64t+64)
// group_offsets[g]  = number of non-zero flags in envs [0, 256g);  group_offsets[G] = total
__host__ __device__ inline int64_t padded_tiles(int64_t n) { return num_groups(n) * GROUP_TILES; }
inline int64_t workspace_ints(int64_t n) { return padded_tiles(n) + num_groups(n) + 1; }

// launches the count (optional) and scan kernels on `stream` (oige_core.hip)
int launch_count_flags(const int64_t* flags, int64_t n, int32_t* ws, hipStream_t stream, uint8_t* counts_u8 = nullptr);
int launch_scan_groups(int64_t n, int32_t* ws, hipStream_t stream);
int launch_scan_groups2(int64_t n, int32_t* ws_a, int32_t* ws_b, hipStream_t stream);  // two workspaces, one launch

// ---------------------------------------------------------------- device helpers
__device__ __forceinline__ int lane_id() { return threadIdx.x & (WAVE - 1); }

// number of set bits of `mask` strictly below this lane
__device__ __forceinline__ int rank_below(unsigned long long mask) {
  return __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0));
}

__device__ __forceinline__ int wave_sum(int v) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_xor(v, off, WAVE);
  return v;
}

// Rank (position in the ascending compacted id list) of env `env` if `flagged`, for a
// 256-thread workgroup that owns group `g` (envs [256g, 256g+256)).  Needs 4 ints of LDS.
// Every thread of the workgroup must call it.  Returns -1 for unflagged lanes.
// Every vector-memory operation this wave has issued is complete (s_waitcnt vmcnt(0); the other counters left alone).
// Placed once behind a kernel's load phase, in front of its first store: the compiler's own wait-count tracking then knows
// that no load is pending.  Without it, a register filled by a load inside a conditional block keeps its "pending" mark along
// every path that skipped the block, and each later use gets a full `s_waitcnt vmcnt(0)` of its own -- which, behind a
// store, is a wait for that store's round trip to HBM (loads and stores share the counter).
__device__ __forceinline__ void wait_loads_done() { __builtin_amdgcn_s_waitcnt(0x0F70); }

// LDS written by some lanes of this wave is read by others of the SAME wave: the LDS executes a wave's operations in order, so
// all that is needed is that the compiler keeps them in order and the wave does not run ahead of its own writes.
__device__ __forceinline__ void wave_lds_fence() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

__device__ __forceinline__ int64_t group_rank(bool flagged, const int32_t* group_offsets, int64_t g, int* lds4) {
  const unsigned long long mask = __ballot(flagged);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0 && w < GROUP_TILES) lds4[w] = __popcll(mask);  // (a workgroup may carry extra waves that own no env)
  __syncthreads();
  int before = 0;
#pragma unroll
  for (int i = 0; i < GROUP_TILES; ++i) before += (i < w) ? lds4[i] : 0;
  const int64_t r = (int64_t)group_offsets[g] + before + rank_below(mask);
  return flagged ? r : -1;
}

// The same rank WITHOUT the group scan: the workgroup sums the tile counts of every group in front of its own (the counts
// are final when the kernel starts; no workgroup waits for another) -- g / 64 independent 16-byte loads per thread at most,
// from a table of 16 bytes per group that stays in L2.  Saves the scan launch between a post and a pre kernel; costs
// 8 g bytes of L2 reads per workgroup on average.  Needs 2 * GROUP_TILES ints of LDS; every thread must call it.
// Returns the rank (-1 for unflagged lanes); k0 = flagged envs in front of the group, cnt = flagged envs in it.
__device__ __forceinline__ int64_t group_rank_unscanned(bool flagged, const int32_t* tile_counts, int64_t g, int* lds8, int& k0,
                                                        int& cnt) {
  const int4* c4 = reinterpret_cast<const int4*>(tile_counts);  // groups in front of g are whole: no padded slot among them
  int part = 0;
  constexpr int U = 8;
  for (int64_t j0 = threadIdx.x; j0 < g; j0 += (int64_t)U * GROUP) {
    int4 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // idle slots re-read a valid entry: a load inside a conditional block costs a full wait
      const int64_t j = j0 + (int64_t)u * GROUP;
      c[u] = c4[j < g ? j : g - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) part += (j0 + (int64_t)u * GROUP < g) ? (c[u].x + c[u].y) + (c[u].z + c[u].w) : 0;
  }
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) part += __shfl_xor(part, off, WAVE);
  const unsigned long long mask = __ballot(flagged);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0 && w < GROUP_TILES) {
    lds8[w] = __popcll(mask);
    lds8[GROUP_TILES + w] = part;
  }
  __syncthreads();
  int before = 0, all = 0, front = 0;
#pragma unroll
  for (int i = 0; i < GROUP_TILES; ++i) {
    before += (i < w) ? lds8[i] : 0;
    all += lds8[i];
    front += lds8[GROUP_TILES + i];
  }
  k0 = front;
  cnt = all;
  return flagged ? (int64_t)front + before + rank_below(mask) : -1;
}

// The same from a BYTE per tile (a count is at most 64): word w of the table = the four tile counts of group w, so the
// workgroup reads 4 g bytes instead of 16 g.  The table holds tile_counts_u8_bytes(n) bytes (whole 16-byte chunks).
__host__ __device__ inline int64_t tile_counts_u8_bytes(int64_t n) { return 16 * ((num_groups(n) + 3) / 4); }

// This thread's share of the sum of the bytes of words [0, g) of the table, NT threads taking 16-byte chunks in turn.
template <int NT, int U = 4>
__device__ __forceinline__ unsigned u8_words_partial(const uint8_t* counts_u8, int64_t g, int tid) {
  const uint4* c4 = reinterpret_cast<const uint4*>(counts_u8);
  const int64_t nq = (g + 3) / 4;  // 16-byte chunks that hold words [0, g)
  unsigned part = 0;
  for (int64_t j0 = tid; j0 < nq; j0 += (int64_t)U * NT) {
    uint4 c[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {  // idle slots re-read a valid chunk (no load inside a conditional block)
      const int64_t j = j0 + (int64_t)u * NT;
      c[u] = c4[j < nq ? j : nq - 1];
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const int64_t w0 = 4 * (j0 + (int64_t)u * NT);  // first word of the chunk; words >= g belong to this group or later ones
      part = __builtin_amdgcn_sad_u8(w0 < g ? c[u].x : 0u, 0u, part);  // v_sad_u8: the four bytes of a word summed in one operation
      part = __builtin_amdgcn_sad_u8(w0 + 1 < g ? c[u].y : 0u, 0u, part);
      part = __builtin_amdgcn_sad_u8(w0 + 2 < g ? c[u].z : 0u, 0u, part);
      part = __builtin_amdgcn_sad_u8(w0 + 3 < g ? c[u].w : 0u, 0u, part);
    }
  }
  return part;
}

__device__ __forceinline__ int64_t group_rank_unscanned_u8(bool flagged, const uint8_t* counts_u8, int64_t g, int* lds8, int& k0,
                                                           int& cnt) {
  int sum = (int)u8_words_partial<GROUP>(counts_u8, g, (int)threadIdx.x);
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) sum += __shfl_xor(sum, off, WAVE);
  const unsigned long long mask = __ballot(flagged);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0 && w < GROUP_TILES) {
    lds8[w] = __popcll(mask);
    lds8[GROUP_TILES + w] = sum;
  }
  __syncthreads();
  int before = 0, all = 0, front = 0;
#pragma unroll
  for (int i = 0; i < GROUP_TILES; ++i) {
    before += (i < w) ? lds8[i] : 0;
    all += lds8[i];
    front += lds8[GROUP_TILES + i];
  }
  k0 = front;
  cnt = all;
  return flagged ? (int64_t)front + before + rank_below(mask) : -1;
}

// Up to this many groups (128 K envs) the pre kernels take the unscanned rank and their host code skips the scan launch:
// the redundant reads grow with the square of the env count, and a kernel with few resident workgroups cannot hide them
// (newhand_cabinet_pre_kernel at 2048 groups: 67 -> 74 us, against the 4 us scan launch it saved).  The locomotion pre kernel
// has its own, higher limit (locomotion.hip).
constexpr int64_t UNSCANNED_MAX_GROUPS = 512;

// group_rank / group_rank_unscanned behind one call.  `ws` = the reset workspace (tile counts, then the group offsets when
// SCANNED), `lds8` = 2 * GROUP_TILES ints.
template <bool SCANNED>
__device__ __forceinline__ int64_t group_rank_of(bool flagged, const int32_t* ws, int64_t G, int64_t g, int* lds8, int& k0, int& cnt) {
  if (SCANNED) {
    const int32_t* group_offsets = ws + G * GROUP_TILES;
    k0 = group_offsets[g];
    cnt = group_offsets[g + 1] - k0;
    return group_rank(flagged, group_offsets, g, lds8);
  }
  return group_rank_unscanned(flagged, ws, g, lds8, k0, cnt);
}

// The three sources of a pre kernel's group offsets as one template parameter, and the host's choice between them.
constexpr int RANK_SCANNED = 0, RANK_INT32 = 1, RANK_U8 = 2;
constexpr int64_t UNSCANNED_U8_MAX_GROUPS = 4096;  // with the byte table the prefix reads are a quarter (1 M envs)

template <int RANK>
__device__ __forceinline__ int64_t group_rank_mode(bool flagged, const int32_t* ws, const uint8_t* counts_u8, int64_t G, int64_t g,
                                                   int* lds8, int& k0, int& cnt) {
  if (RANK == RANK_U8) return group_rank_unscanned_u8(flagged, counts_u8, g, lds8, k0, cnt);
  return group_rank_of<RANK == RANK_SCANNED>(flagged, ws, G, g, lds8, k0, cnt);
}

// The byte-table prefix in ONE pass, split in two so that its loads leave together with a kernel's other loads and are summed
// when they have arrived (u8_words_partial's loop puts a full wait inside its body): up to UNSCANNED_U8_MAX_GROUPS groups the
// four 16-byte chunks a thread takes cover every word in front of any group.
static_assert(UNSCANNED_U8_MAX_GROUPS <= 4 * 4 * GROUP, "one pass of four chunks per thread must cover the table");
struct U8Prefix {
  uint4 c[4];
};
__device__ __forceinline__ U8Prefix u8_prefix_issue(const uint8_t* counts_u8, int64_t g, int tid) {
  const uint4* c4 = reinterpret_cast<const uint4*>(counts_u8);
  const int nq = (int)((g + 3) / 4);  // chunks that hold words [0, g)
  U8Prefix p;
#pragma unroll
  for (int u = 0; u < 4; ++u) {  // idle slots re-read a valid chunk: no load inside a conditional block
    const int j = tid + u * GROUP;
    p.c[u] = c4[j < nq ? j : (nq > 0 ? nq - 1 : 0)];
  }
  return p;
}
__device__ __forceinline__ int u8_prefix_sum(const U8Prefix& p, int64_t g, int tid) {
  unsigned part = 0;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t w0 = 4 * (int64_t)(tid + u * GROUP);  // first word of the chunk; words >= g belong to this group or later ones
    part = __builtin_amdgcn_sad_u8(w0 < g ? p.c[u].x : 0u, 0u, part);
    part = __builtin_amdgcn_sad_u8(w0 + 1 < g ? p.c[u].y : 0u, 0u, part);
    part = __builtin_amdgcn_sad_u8(w0 + 2 < g ? p.c[u].z : 0u, 0u, part);
    part = __builtin_amdgcn_sad_u8(w0 + 3 < g ? p.c[u].w : 0u, 0u, part);
  }
  return wave_sum((int)part);
}

// group_rank_mode in two halves for a 256-thread workgroup: rank_issue<RANK> next to the kernel's first loads (RANK_U8: the
// byte-table chunks; the other modes issue nothing), rank_finish<RANK> where the flags have arrived.  Same results as
// group_rank_mode; every thread must call both.
template <int RANK>
__device__ __forceinline__ U8Prefix rank_issue(const uint8_t* counts_u8, int64_t g, int tid) {
  if (RANK == RANK_U8) return u8_prefix_issue(counts_u8, g, tid);
  return U8Prefix{};
}
template <int RANK>
__device__ __forceinline__ int64_t rank_finish(const U8Prefix& pre, bool flagged, const int32_t* ws, const uint8_t* counts_u8, int64_t G,
                                               int64_t g, int* lds8, int& k0, int& cnt) {
  if (RANK != RANK_U8) return group_rank_mode<RANK>(flagged, ws, counts_u8, G, g, lds8, k0, cnt);
  const int tid = (int)threadIdx.x;
  const int front = u8_prefix_sum(pre, g, tid);
  const unsigned long long mask = __ballot(flagged);
  const int w = tid >> 6;
  if (lane_id() == 0) {
    lds8[w] = __popcll(mask);
    lds8[GROUP_TILES + w] = front;
  }
  __syncthreads();
  int before = 0;
  k0 = 0;
  cnt = 0;
#pragma unroll
  for (int i = 0; i < GROUP_TILES; ++i) {
    before += (i < w) ? lds8[i] : 0;
    cnt += lds8[i];
    k0 += lds8[GROUP_TILES + i];
  }
  return flagged ? (int64_t)k0 + before + rank_below(mask) : -1;
}

inline int pick_rank(int64_t G, const uint8_t* counts_u8, int64_t int32_max_groups = UNSCANNED_MAX_GROUPS) {
  if (counts_u8 && G <= UNSCANNED_U8_MAX_GROUPS) return RANK_U8;
  return G <= int32_max_groups ? RANK_INT32 : RANK_SCANNED;
}

}  // namespace oige
