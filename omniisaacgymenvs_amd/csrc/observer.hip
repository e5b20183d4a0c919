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

}  // namespace oige

using namespace oige;

extern "C" {

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
