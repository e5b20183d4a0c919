// Episode-stat observer for gfx950 (MI355X) -- SURVEY.md 8f rank 4.
//
// Replaces the tensor bookkeeping of /root/reference/omniisaacgymenvs/utils/rlgames/rlgames_utils.py:52-85
// (RLGPUAlgoObserver.process_infos / after_print_stats): the reference appends infos["episode"] (a dict of scalars) to a
// Python list every step and, once per epoch, grows a tensor by torch.cat per key and per step before taking its mean
// (K keys x T steps launches + allocations).  Here each step is ONE launch that folds the step's K values into running f64
// sums and counts; the epoch's means are one more launch.  Latency-bound: K <= 64 keys, a wavefront per key.
#include "oige_common.hpp"

namespace oige {

__global__ __launch_bounds__(WAVE) void observer_accumulate_kernel(const oige_observer_batch b, double* __restrict__ sums,
                                                                   int64_t* __restrict__ counts) {
  const int k = blockIdx.x;
  const int lane = lane_id();
  const float* v = b.values[k];
  const int len = b.lengths[k];
  double acc = 0.0;
  for (int i = lane; i < len; i += WAVE) acc += (double)v[i];
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) acc += __shfl_down(acc, off, WAVE);
  if (lane == 0) {
    sums[k] += acc;
    counts[k] += len;
  }
}

__global__ __launch_bounds__(WAVE) void observer_means_kernel(double* __restrict__ sums, int64_t* __restrict__ counts,
                                                              float* __restrict__ means, int num_keys, int clear) {
  const int k = threadIdx.x;
  if (k >= num_keys) return;
  const int64_t c = counts[k];
  means[k] = c > 0 ? (float)(sums[k] / (double)c) : __builtin_nanf("");  // torch.mean of an empty tensor is nan
  if (clear) {
    sums[k] = 0.0;
    counts[k] = 0;
  }
}

// ---------------------------------------------------------------- per-step episode statistics of one rank (multi-GPU)
// The only cross-rank exchange of the path is the statistics vector the trainer logs (rlgames_utils.py:72-96; one process per
// GPU, scripts/rlgames_train.py:114-120,171).  This rank's contribution is kept on the device as SLOTS of four running f64 sums
// {sum reward, done envs, sum of the done envs' episode lengths, sum successes}: one slot per workgroup of the accumulate
// kernel below, or one per 64-env tile when a post kernel keeps them itself (locomotion_post_kernel: no extra launch, no
// extra traffic).  A slot is only ever touched by "its" workgroup / wave, step after step, so every sum has a fixed order.
// oige_episode_stats_fold adds the slots up (fixed order again) when a collective is due -- once per trainer epoch, not per step.
constexpr int STATS_THREADS = 256;
constexpr int STATS_ENVS_PER_THREAD = 4;

template <int N>
__device__ __forceinline__ void block_sum4(double (&v)[N], double* lds /* N * STATS_THREADS / WAVE */) {
#pragma unroll
  for (int k = 0; k < N; ++k)
#pragma unroll
    for (int off = WAVE / 2; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off, WAVE);
  const int w = threadIdx.x >> 6;
  if (lane_id() == 0)
#pragma unroll
    for (int k = 0; k < N; ++k) lds[k * (STATS_THREADS / WAVE) + w] = v[k];
  __syncthreads();
#pragma unroll
  for (int k = 0; k < N; ++k) {
    double t = 0.0;
#pragma unroll
    for (int i = 0; i < STATS_THREADS / WAVE; ++i) t += lds[k * (STATS_THREADS / WAVE) + i];
    v[k] = t;
  }
}

// one step of this rank: slot[blockIdx] += the block's share.  Four consecutive envs per thread and pass (16-byte loads).
__global__ __launch_bounds__(STATS_THREADS) void episode_stats_accumulate_kernel(const oige_episode_stats_io io, int64_t n) {
  __shared__ double lds[4 * STATS_THREADS / WAVE];
  double v[4] = {0.0, 0.0, 0.0, 0.0};  // rew, done, len, succ
  const uint8_t* r8 = static_cast<const uint8_t*>(io.reset_buf);
  const int64_t* r64 = static_cast<const int64_t*>(io.reset_buf);
  const int64_t stride = (int64_t)gridDim.x * STATS_THREADS * STATS_ENVS_PER_THREAD;
  for (int64_t i0 = ((int64_t)blockIdx.x * STATS_THREADS + threadIdx.x) * STATS_ENVS_PER_THREAD; i0 < n; i0 += stride) {
    if (i0 + STATS_ENVS_PER_THREAD <= n) {
      const float4 rw = *reinterpret_cast<const float4*>(io.rew_buf + i0);
      const longlong2 p0 = *reinterpret_cast<const longlong2*>(io.progress_buf + i0);
      const longlong2 p1 = *reinterpret_cast<const longlong2*>(io.progress_buf + i0 + 2);
      bool d[4];
      if (io.reset_elem_bytes == 1) {
        const uint32_t m = *reinterpret_cast<const uint32_t*>(r8 + i0);
        d[0] = (m & 0xFFu) != 0; d[1] = (m & 0xFF00u) != 0; d[2] = (m & 0xFF0000u) != 0; d[3] = (m & 0xFF000000u) != 0;
      } else {
        const longlong2 q0 = *reinterpret_cast<const longlong2*>(r64 + i0);
        const longlong2 q1 = *reinterpret_cast<const longlong2*>(r64 + i0 + 2);
        d[0] = q0.x != 0; d[1] = q0.y != 0; d[2] = q1.x != 0; d[3] = q1.y != 0;
      }
      const float rr[4] = {rw.x, rw.y, rw.z, rw.w};
      const long long pp[4] = {p0.x, p0.y, p1.x, p1.y};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[0] += (double)rr[k];
        v[1] += d[k] ? 1.0 : 0.0;
        v[2] += d[k] ? (double)pp[k] : 0.0;
      }
      if (io.successes) {
        const float4 sc = *reinterpret_cast<const float4*>(io.successes + i0);
        v[3] += ((double)sc.x + (double)sc.y) + ((double)sc.z + (double)sc.w);
      }
    } else {
      for (int64_t i = i0; i < n; ++i) {
        const bool dd = io.reset_elem_bytes == 1 ? r8[i] != 0 : r64[i] != 0;
        v[0] += (double)io.rew_buf[i];
        v[1] += dd ? 1.0 : 0.0;
        v[2] += dd ? (double)io.progress_buf[i] : 0.0;
        if (io.successes) v[3] += (double)io.successes[i];
      }
    }
  }
  block_sum4<4>(v, lds);
  if (threadIdx.x < 4) io.slots[4 * (int64_t)blockIdx.x + threadIdx.x] += v[threadIdx.x];
}

// Workgroup b adds up slots [b * per_block, (b + 1) * per_block) in a fixed order -> out[4 b .. 4 b + 3]; the launch that
// produces the final sums (one workgroup) also sets out[4] = env_steps.
__global__ __launch_bounds__(STATS_THREADS) void episode_stats_fold_kernel(const double* __restrict__ slots, int64_t num_slots,
                                                                           int64_t per_block, double* __restrict__ out,
                                                                           double env_steps, int last) {
  __shared__ double lds[4 * STATS_THREADS / WAVE];
  double t[4] = {0.0, 0.0, 0.0, 0.0};
  const double2* s2 = reinterpret_cast<const double2*>(slots);
  const int64_t lo_slot = (int64_t)blockIdx.x * per_block;
  const int64_t hi_slot = lo_slot + per_block < num_slots ? lo_slot + per_block : num_slots;
  for (int64_t b = lo_slot + threadIdx.x; b < hi_slot; b += STATS_THREADS) {
    const double2 lo = s2[2 * b], hi = s2[2 * b + 1];
    t[0] += lo.x; t[1] += lo.y; t[2] += hi.x; t[3] += hi.y;
  }
  block_sum4<4>(t, lds);
  if (threadIdx.x < 4) out[4 * (int64_t)blockIdx.x + threadIdx.x] = t[threadIdx.x];
  if (last && threadIdx.x == 4) out[4] = env_steps;
}

}  // namespace oige

using namespace oige;

extern "C" {

int64_t oige_episode_stats_blocks(int64_t num_envs) {
  const int64_t per = (int64_t)STATS_THREADS * STATS_ENVS_PER_THREAD;  // one pass of a workgroup
  const int64_t b = (num_envs + per - 1) / per;
  return b < 1 ? 1 : (b > OIGE_EPISODE_STATS_MAX_BLOCKS ? OIGE_EPISODE_STATS_MAX_BLOCKS : b);
}

int oige_episode_stats_accumulate(const oige_episode_stats_io* io, int64_t num_envs, oige_stream_t stream) {
  OIGE_REQUIRE(io != nullptr, "episode_stats_accumulate: null io");
  OIGE_REQUIRE(io->rew_buf && io->reset_buf && io->progress_buf && io->slots, "episode_stats_accumulate: null pointer");
  OIGE_REQUIRE(io->reset_elem_bytes == 1 || io->reset_elem_bytes == 8, "episode_stats_accumulate: reset_elem_bytes must be 1 (bool) or 8 (int64), not %d",
               io->reset_elem_bytes);
  OIGE_REQUIRE(num_envs >= 0, "episode_stats_accumulate: num_envs %lld out of range", (long long)num_envs);
  OIGE_REQUIRE(aligned16(io->rew_buf) && aligned16(io->reset_buf) && aligned16(io->progress_buf) && aligned16(io->successes) &&
                   aligned16(io->slots),
               "episode_stats_accumulate: buffers must be 16-byte aligned");
  if (num_envs == 0) return OIGE_OK;
  const int blocks = (int)oige_episode_stats_blocks(num_envs);
  hipLaunchKernelGGL(episode_stats_accumulate_kernel, dim3((unsigned)blocks), dim3(STATS_THREADS), 0, (hipStream_t)stream, *io, num_envs);
  return check_launch("episode_stats_accumulate_kernel");
}

int oige_episode_stats_fold(const double* slots, int64_t num_slots, double* acc, double env_steps, double* scratch,
                            oige_stream_t stream) {
  OIGE_REQUIRE(slots && acc, "episode_stats_fold: null pointer");
  OIGE_REQUIRE(num_slots >= 0 && aligned16(slots) && aligned16(scratch), "episode_stats_fold: bad slot count / alignment");
  hipStream_t s = (hipStream_t)stream;
  // more than a workgroup's worth of slots (a slot per 64-env tile: 16384 at 1 M envs): two levels, 256 slots per workgroup
  if (num_slots > 4 * STATS_THREADS) {
    OIGE_REQUIRE(scratch != nullptr, "episode_stats_fold: %lld slots need the scratch buffer (oige_episode_stats_fold_scratch doubles)",
                 (long long)num_slots);
    const int64_t blocks = (num_slots + STATS_THREADS - 1) / STATS_THREADS;
    OIGE_REQUIRE(blocks <= 4 * STATS_THREADS * STATS_THREADS, "episode_stats_fold: too many slots");
    hipLaunchKernelGGL(episode_stats_fold_kernel, dim3((unsigned)blocks), dim3(STATS_THREADS), 0, s, slots, num_slots, (int64_t)STATS_THREADS,
                       scratch, 0.0, 0);
    if (int rc = check_launch("episode_stats_fold_kernel")) return rc;
    slots = scratch;
    num_slots = blocks;
  }
  hipLaunchKernelGGL(episode_stats_fold_kernel, dim3(1), dim3(STATS_THREADS), 0, s, slots, num_slots, num_slots > 0 ? num_slots : 1, acc,
                     env_steps, 1);
  return check_launch("episode_stats_fold_kernel");
}

int64_t oige_episode_stats_fold_scratch(int64_t num_slots) {
  return num_slots > 4 * STATS_THREADS ? 4 * ((num_slots + STATS_THREADS - 1) / STATS_THREADS) : 0;
}

int oige_observer_accumulate(const oige_observer_batch* batch, double* sums, int64_t* counts, oige_stream_t stream) {
  OIGE_REQUIRE(batch && sums && counts, "observer_accumulate: null argument");
  OIGE_REQUIRE(batch->num_keys >= 0 && batch->num_keys <= OIGE_OBSERVER_MAX_KEYS, "observer_accumulate: num_keys %d out of range [0, %d]",
               batch->num_keys, OIGE_OBSERVER_MAX_KEYS);
  for (int k = 0; k < batch->num_keys; ++k)
    OIGE_REQUIRE(batch->lengths[k] >= 0 && (batch->lengths[k] == 0 || batch->values[k]), "observer_accumulate: key %d: null values / bad length",
                 k);
  if (batch->num_keys == 0) return OIGE_OK;
  hipLaunchKernelGGL(observer_accumulate_kernel, dim3((unsigned)batch->num_keys), dim3(WAVE), 0, (hipStream_t)stream, *batch, sums, counts);
  return check_launch("observer_accumulate_kernel");
}

int oige_observer_means(double* sums, int64_t* counts, float* means_out, int32_t num_keys, int32_t clear, oige_stream_t stream) {
  OIGE_REQUIRE(sums && counts && means_out, "observer_means: null argument");
  OIGE_REQUIRE(num_keys >= 0 && num_keys <= OIGE_OBSERVER_MAX_KEYS, "observer_means: num_keys %d out of range [0, %d]", num_keys,
               OIGE_OBSERVER_MAX_KEYS);
  if (num_keys == 0) return OIGE_OK;
  hipLaunchKernelGGL(observer_means_kernel, dim3(1), dim3(WAVE), 0, (hipStream_t)stream, sums, counts, means_out, (int)num_keys, (int)clear);
  return check_launch("observer_means_kernel");
}

}  // extern "C"
