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
// GPU, scripts/rlgames_train.py:114-120,171).  This rank's contribution is folded on the device, once per step, from the
// buffers the post kernel has just left (20 bytes per env, still in cache), into five running f64 sums that the caller hands
// to ONE small all-reduce: [sum reward, done envs, sum of the done envs' episode lengths, sum successes, envs].
// Two launches (launch gaps are ~0 on this part): per-workgroup partials in a fixed thread order, then one workgroup that
// folds them in a fixed order -- the sums do not depend on scheduling.
constexpr int STATS_THREADS = 256;

template <class T>
__device__ __forceinline__ T block_sum(T v, T* lds /* STATS_THREADS / WAVE entries */) {
#pragma unroll
  for (int off = WAVE / 2; off > 0; off >>= 1) v += __shfl_down(v, off, WAVE);
  const int w = threadIdx.x >> 6;
  __syncthreads();  // the previous use of lds is over
  if (lane_id() == 0) lds[w] = v;
  __syncthreads();
  T t = 0;
#pragma unroll
  for (int i = 0; i < STATS_THREADS / WAVE; ++i) t += lds[i];
  return t;
}

__global__ __launch_bounds__(STATS_THREADS) void episode_stats_partial_kernel(const oige_episode_stats_io io, int64_t n) {
  __shared__ double lds[STATS_THREADS / WAVE];
  double rew = 0.0, done = 0.0, len = 0.0, succ = 0.0;
  const uint8_t* r8 = static_cast<const uint8_t*>(io.reset_buf);
  const int64_t* r64 = static_cast<const int64_t*>(io.reset_buf);
  for (int64_t i = (int64_t)blockIdx.x * STATS_THREADS + threadIdx.x; i < n; i += (int64_t)gridDim.x * STATS_THREADS) {
    const bool d = io.reset_elem_bytes == 1 ? r8[i] != 0 : r64[i] != 0;
    rew += (double)io.rew_buf[i];
    done += d ? 1.0 : 0.0;
    len += d ? (double)io.progress_buf[i] : 0.0;
    if (io.successes) succ += (double)io.successes[i];
  }
  rew = block_sum(rew, lds);
  done = block_sum(done, lds);
  len = block_sum(len, lds);
  succ = block_sum(succ, lds);
  if (threadIdx.x == 0) {
    double* p = io.partials + 4 * (int64_t)blockIdx.x;
    p[0] = rew; p[1] = done; p[2] = len; p[3] = succ;
  }
}

__global__ __launch_bounds__(STATS_THREADS) void episode_stats_fold_kernel(const oige_episode_stats_io io, int blocks, int64_t n) {
  __shared__ double lds[STATS_THREADS / WAVE];
  double t[4] = {0.0, 0.0, 0.0, 0.0};
  for (int b = threadIdx.x; b < blocks; b += STATS_THREADS)
#pragma unroll
    for (int k = 0; k < 4; ++k) t[k] += io.partials[4 * (int64_t)b + k];
#pragma unroll
  for (int k = 0; k < 4; ++k) t[k] = block_sum(t[k], lds);
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < 4; ++k) io.acc[k] += t[k];
    io.acc[4] += (double)n;
  }
}

}  // namespace oige

using namespace oige;

extern "C" {

int64_t oige_episode_stats_blocks(int64_t num_envs) {
  const int64_t b = (num_envs + 4 * STATS_THREADS - 1) / (4 * STATS_THREADS);  // four envs per thread at least
  return b < 1 ? 1 : (b > OIGE_EPISODE_STATS_MAX_BLOCKS ? OIGE_EPISODE_STATS_MAX_BLOCKS : b);
}

int oige_episode_stats_accumulate(const oige_episode_stats_io* io, int64_t num_envs, oige_stream_t stream) {
  OIGE_REQUIRE(io != nullptr, "episode_stats_accumulate: null io");
  OIGE_REQUIRE(io->rew_buf && io->reset_buf && io->progress_buf && io->partials && io->acc, "episode_stats_accumulate: null pointer");
  OIGE_REQUIRE(io->reset_elem_bytes == 1 || io->reset_elem_bytes == 8, "episode_stats_accumulate: reset_elem_bytes must be 1 (bool) or 8 (int64), not %d",
               io->reset_elem_bytes);
  OIGE_REQUIRE(num_envs >= 0, "episode_stats_accumulate: num_envs %lld out of range", (long long)num_envs);
  if (num_envs == 0) return OIGE_OK;
  const int blocks = (int)oige_episode_stats_blocks(num_envs);
  hipLaunchKernelGGL(episode_stats_partial_kernel, dim3((unsigned)blocks), dim3(STATS_THREADS), 0, (hipStream_t)stream, *io, num_envs);
  if (int rc = check_launch("episode_stats_partial_kernel")) return rc;
  hipLaunchKernelGGL(episode_stats_fold_kernel, dim3(1), dim3(STATS_THREADS), 0, (hipStream_t)stream, *io, blocks, num_envs);
  return check_launch("episode_stats_fold_kernel");
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
