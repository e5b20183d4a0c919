// liboige_hip.so core: error reporting, version, and the ordered stream compaction that
// replaces `reset_buf.nonzero(as_tuple=False).squeeze(-1)` + `len()` (a forced device->host
// sync in the reference: tasks/shared/locomotion.py:121-122, cartpole.py:112-113,
// in_hand_manipulation.py:318-319, anymal_terrain.py:433, randomize.py:284,315).
//
// Three small kernels, all HBM/L2-bound integer work, no host sync, count stays on device:
//   count : one ballot+popcount per wave            -> tile_counts[t]      (8 B/env read)
//   scan  : one workgroup, exclusive scan per 256-env group -> group_offsets[g], total
//   emit  : per wave, rank = group offset + earlier waves + mbcnt(ballot); ids[rank] = env
// The task kernels (locomotion.hip, cartpole.hip, ...) fuse `count` into their DONE stage and
// `emit` into their reset stage, so a steady-state step only adds the tiny `scan` launch.
#include <atomic>

#include "oige_common.hpp"
#include "oige_scan.hpp"
#include "oige_tile.hpp"

namespace oige {

char* error_buffer() {
  static thread_local char buf[ERROR_BUFFER_BYTES] = {0};
  return buf;
}

__global__ __launch_bounds__(GROUP) void count_flags_kernel(const int64_t* __restrict__ flags, int64_t n,
                                                          int32_t* __restrict__ tile_counts, uint8_t* __restrict__ counts_u8) {
  const int64_t env = (int64_t)blockIdx.x * GROUP + threadIdx.x;
  const bool f = env < n && flags[env] != 0;
  const unsigned long long mask = __ballot(f);
  if (lane_id() == 0) {
    const int64_t t = (int64_t)blockIdx.x * GROUP_TILES + (threadIdx.x >> 6);
    tile_counts[t] = __popcll(mask);
    if (counts_u8) counts_u8[t] = (uint8_t)__popcll(mask);  // (the byte twin of the table, locomotion_pre_kernel)
  }
}

// One workgroup of 1024 threads; thread i owns a contiguous run of groups (oige_scan.hpp).
__global__ __launch_bounds__(SCAN_THREADS) void scan_groups_kernel(const int32_t* __restrict__ tile_counts,
                                                                   int64_t n_groups, int64_t n_tiles,
                                                                   int32_t* __restrict__ group_offsets) {
  __shared__ int wave_tot[SCAN_THREADS / WAVE];
  scan_groups_body(tile_counts, n_groups, n_tiles, group_offsets, wave_tot);
}

// Two independent scans (a task's second compaction: goal resets, target resets, the noise interval) in one launch:
// workgroup b scans workspace b.  Each is a one-workgroup, latency-bound pass; side by side they cost one.
__global__ __launch_bounds__(SCAN_THREADS) void scan_groups2_kernel(int32_t* __restrict__ ws_a, int32_t* __restrict__ ws_b,
                                                                    int64_t n_groups, int64_t n_tiles, int64_t padded) {
  __shared__ int wave_tot[SCAN_THREADS / WAVE];
  int32_t* ws = blockIdx.x == 0 ? ws_a : ws_b;
  scan_groups_body(ws, n_groups, n_tiles, ws + padded, wave_tot);
}

__global__ __launch_bounds__(GROUP) void emit_ids_kernel(const int64_t* __restrict__ flags, int64_t n,
                                                       const int32_t* __restrict__ group_offsets, int64_t n_groups,
                                                       int64_t* __restrict__ ids, int64_t* __restrict__ count) {
  __shared__ int lds4[GROUP_TILES];
  const int64_t g = blockIdx.x;
  const int64_t env = g * GROUP + threadIdx.x;
  const bool f = env < n && flags[env] != 0;
  const int64_t r = group_rank(f, group_offsets, g, lds4);
  if (f) ids[r] = env;
  if (g == 0 && threadIdx.x == 0) *count = (int64_t)group_offsets[n_groups];
}

// Calibration only: a streaming copy (16 B per lane, fully coalesced, read n bytes + write n bytes), the practical
// ceiling of a 50/50 read/write HBM stream on this chip; bench.py reports it beside the 8 TB/s spec peak.
// `streams` splits the buffers into that many equal sub-arrays and every thread copies the same index of each, the
// way one task kernel walks a dozen separate (N,k) tensors at once.
__global__ __launch_bounds__(256) void stream_copy_kernel(const float4* __restrict__ src, float4* __restrict__ dst, int64_t n4,
                                                          int streams) {
  const int64_t per = n4 / streams;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= per) return;
  for (int s = 0; s < streams; ++s)
    st4_stream(reinterpret_cast<float*>(dst + s * per + i), ld4_stream(reinterpret_cast<const float*>(src + s * per + i)));
}

int launch_count_flags(const int64_t* flags, int64_t n, int32_t* ws, hipStream_t stream, uint8_t* counts_u8) {
  const int64_t G = num_groups(n);
  if (G == 0) return OIGE_OK;
  hipLaunchKernelGGL(count_flags_kernel, dim3((unsigned)G), dim3(GROUP), 0, stream, flags, n, ws, counts_u8);
  return check_launch("count_flags_kernel");
}

int launch_scan_groups(int64_t n, int32_t* ws, hipStream_t stream) {
  const int64_t G = num_groups(n);
  hipLaunchKernelGGL(scan_groups_kernel, dim3(1), dim3(SCAN_THREADS), 0, stream, ws, G, num_tiles(n), ws + padded_tiles(n));
  return check_launch("scan_groups_kernel");
}

int launch_scan_groups2(int64_t n, int32_t* ws_a, int32_t* ws_b, hipStream_t stream) {
  hipLaunchKernelGGL(scan_groups2_kernel, dim3(2), dim3(SCAN_THREADS), 0, stream, ws_a, ws_b, num_groups(n), num_tiles(n), padded_tiles(n));
  return check_launch("scan_groups2_kernel");
}

// ---------------------------------------------------------------- the entry point that allocates nothing
// oige_compact_i64_static: oige_compact_i64 with the reset workspace (tile counts | group offsets | total) in a
// file-scope device array instead of a caller-provided buffer.  Each GPU the library runs on holds its own copy of the
// array until the process ends; the array's NAME is only a host-side stand-in, the kernels get the per-device address
// that hipGetSymbolAddress returned for it.
constexpr int64_t STATIC_MAX_ENVS = (int64_t)1 << 22;  // 4 M flags per call
constexpr int64_t STATIC_SCRATCH_INTS = (STATIC_MAX_ENVS / TILE) + (STATIC_MAX_ENVS / GROUP) + 4;  // 81 924 ints = 320 KiB
static __device__ __align__(256) int32_t scratch[STATIC_SCRATCH_INTS];

constexpr int STATIC_MAX_DEVICES = 64;
static std::atomic<int32_t*> scratch_ptr[STATIC_MAX_DEVICES];  // per device: each GPU has its own copy at its own address

// The address of `scratch` on the CURRENT device (looked up once per device, with that device current).  `stream` must
// belong to that device: a stream of another GPU would run the kernels there, writing through THIS device's address.
static int static_scratch(int32_t** out, hipStream_t stream) {
  int dev = -1;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= STATIC_MAX_DEVICES)
    return fail(OIGE_ELAUNCH, "compact_static: no current HIP device (or device index >= %d)", STATIC_MAX_DEVICES);
  if (stream != nullptr) {
    int sdev = -1;
    if (hipStreamGetDevice(stream, &sdev) == hipSuccess && sdev != dev)
      return fail(OIGE_EINVAL, "compact_static: the stream belongs to device %d but device %d is current (each GPU has its own scratch: "
                  "make the stream's device current)", sdev, dev);
  }
  int32_t* p = scratch_ptr[dev].load(std::memory_order_acquire);
  if (!p) {
    void* sym = nullptr;
    const hipError_t e = hipGetSymbolAddress(&sym, HIP_SYMBOL(scratch));
    if (e != hipSuccess || !sym) return fail(OIGE_ELAUNCH, "compact_static: hipGetSymbolAddress(scratch): %s", hipGetErrorString(e));
    p = static_cast<int32_t*>(sym);
    scratch_ptr[dev].store(p, std::memory_order_release);  // two racing first calls look the same address up twice
  }
  *out = p;
  return OIGE_OK;
}

}  // namespace oige

using namespace oige;

extern "C" {

const char* oige_last_error(void) { return error_buffer(); }
int oige_version(void) { return OIGE_ABI_VERSION; }
int64_t oige_num_tiles(int64_t n) { return n < 0 ? 0 : num_tiles(n); }
int64_t oige_workspace_ints(int64_t n) { return n < 0 ? 0 : workspace_ints(n); }
int64_t oige_tile_counts_u8_bytes(int64_t n) { return n < 0 ? 0 : tile_counts_u8_bytes(n); }

int oige_stream_copy(const void* src, void* dst, int64_t bytes, int32_t streams, oige_stream_t stream) {
  OIGE_REQUIRE(src && dst && bytes >= 0 && aligned16(src) && aligned16(dst), "stream_copy: bad arguments");
  OIGE_REQUIRE(streams >= 1 && streams <= 64 && bytes % (16 * (int64_t)streams) == 0, "stream_copy: bytes must be a multiple of 16 * streams");
  const int64_t n4 = bytes / 16;
  if (n4 == 0) return OIGE_OK;
  const int64_t g = (n4 / streams + 255) / 256;
  OIGE_REQUIRE(g < ((int64_t)1 << 31), "stream_copy: too large");
  hipLaunchKernelGGL(stream_copy_kernel, dim3((unsigned)g), dim3(256), 0, (hipStream_t)stream,
                     reinterpret_cast<const float4*>(src), reinterpret_cast<float4*>(dst), n4, (int)streams);
  return check_launch("stream_copy_kernel");
}

int oige_count_flags_i64(const int64_t* flags, int64_t n, int32_t* workspace, oige_stream_t stream) {
  OIGE_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "count_flags: num_envs %lld out of range", (long long)n);
  OIGE_REQUIRE(flags && workspace, "count_flags: null pointer");
  OIGE_REQUIRE(aligned16(workspace), "count_flags: workspace must be 16-byte aligned");
  return launch_count_flags(flags, n, workspace, (hipStream_t)stream);
}

int oige_compact_i64(const int64_t* flags, int64_t n, int32_t* workspace, int64_t* ids_out, int64_t* count_out,
                     oige_stream_t stream) {
  OIGE_REQUIRE(n >= 0 && n < (int64_t)1 << 31, "compact: num_envs %lld out of range", (long long)n);
  OIGE_REQUIRE(flags && workspace && ids_out && count_out, "compact: null pointer");
  OIGE_REQUIRE(aligned16(workspace), "compact: workspace must be 16-byte aligned");
  hipStream_t s = (hipStream_t)stream;
  int rc = launch_count_flags(flags, n, workspace, s);
  if (rc) return rc;
  rc = launch_scan_groups(n, workspace, s);
  if (rc) return rc;
  const int64_t G = num_groups(n);
  if (G > 0) {
    hipLaunchKernelGGL(emit_ids_kernel, dim3((unsigned)G), dim3(GROUP), 0, s, flags, n, workspace + padded_tiles(n), G,
                       ids_out, count_out);
    return check_launch("emit_ids_kernel");
  }
  return hipMemsetAsync(count_out, 0, sizeof(int64_t), s) == hipSuccess ? OIGE_OK
                                                                        : fail(OIGE_ELAUNCH, "compact: memset failed");
}

int64_t oige_compact_static_max_envs(void) { return STATIC_MAX_ENVS; }

int oige_compact_i64_static(const int64_t* flags, int64_t n, int64_t* ids_out, int64_t* count_out, oige_stream_t stream) {
  OIGE_REQUIRE(flags && ids_out && count_out, "compact_static: null pointer");
  OIGE_REQUIRE(n >= 0, "compact_static: num_envs %lld out of range", (long long)n);
  OIGE_REQUIRE(n <= STATIC_MAX_ENVS, "compact_static: %lld flags exceed the static scratch (at most %lld per call); use "
               "oige_compact_i64 with a caller workspace", (long long)n, (long long)STATIC_MAX_ENVS);
  static_assert(STATIC_SCRATCH_INTS >= (STATIC_MAX_ENVS / GROUP) * GROUP_TILES + STATIC_MAX_ENVS / GROUP + 1, "scratch too small");
  int32_t* ws = nullptr;
  int rc = static_scratch(&ws, (hipStream_t)stream);
  if (rc) return rc;
  // Nothing in the scratch has to start at zero: count writes every tile count of the n envs' groups (padding tiles
  // included), scan every group offset and the total, and emit reads only those -- the last call's contents are dead.
  return oige_compact_i64(flags, n, ws, ids_out, count_out, stream);
}

}  // extern "C"
