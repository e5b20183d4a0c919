// What can a kernel of the locomotion pre kernel's SHAPE reach on this part?  Per env (D = 8): read 32 B of actions (+ an
// 8-byte flag), write 32 B of forces and 32 B of clone -- 1 M envs = 109 MB per launch, a 17 us kernel at 6.4 TB/s.
// Sweeps workgroup size, float4 per thread and the non-temporal hints; prints the mean of back-to-back launches (HIP
// events) per variant.  Run under `rocprofv3 --kernel-trace --stats` for the per-kernel durations without the launch gaps.
//   hipcc --offload-arch=gfx950 -O3 -o profiles/tools/pre_floor profiles/tools/pre_floor.hip
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4f ld(const v4f* p) {
  return NT ? __builtin_nontemporal_load(p) : *p;
}
template <bool NT>
__device__ __forceinline__ void st(v4f* p, v4f v) {
  if (NT)
    __builtin_nontemporal_store(v, p);
  else
    *p = v;
}

// NTH threads, F4 float4 per thread, one contiguous run per workgroup
template <int NTH, int F4, bool NTL, bool NTS, bool FLAG, bool TWO>
__global__ __launch_bounds__(NTH) void shape_kernel(const v4f* __restrict__ a, v4f* __restrict__ b, v4f* __restrict__ c,
                                                     const int64_t* __restrict__ flag, int64_t* __restrict__ sink, long n4) {
  const long base = (long)blockIdx.x * NTH * F4;
  long fl = 0;
  if (FLAG) {
    const long env = (long)blockIdx.x * (NTH * F4 / 2) + threadIdx.x;  // two float4 per env
    if (threadIdx.x < NTH * F4 / 2) fl = flag[env];
  }
  v4f v[F4];
#pragma unroll
  for (int i = 0; i < F4; ++i) v[i] = ld<NTL>(a + base + i * NTH + threadIdx.x);
#pragma unroll
  for (int i = 0; i < F4; ++i) {
    v4f f = v[i] * 15.0f * 0.5f;
    st<NTS>(b + base + i * NTH + threadIdx.x, f);
    if (TWO) st<NTS>(c + base + i * NTH + threadIdx.x, v[i]);
  }
  if (FLAG && fl != 0) sink[blockIdx.x] = fl;
}

#define CHECK(x)                                                                 \
  do {                                                                           \
    hipError_t e_ = (x);                                                         \
    if (e_ != hipSuccess) {                                                      \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                    \
      exit(1);                                                                   \
    }                                                                            \
  } while (0)

static v4f *A, *B, *C;
static int64_t *F, *S;
static const long N_ENV = 1 << 20;
static const long N4 = N_ENV * 2;

template <int NTH, int F4, bool NTL, bool NTS, bool FLAG, bool TWO>
static void run(const char* name) {
  const int grid = (int)(N4 / (NTH * F4));
  hipEvent_t e0, e1;
  CHECK(hipEventCreate(&e0));
  CHECK(hipEventCreate(&e1));
  for (int i = 0; i < 5; ++i) hipLaunchKernelGGL((shape_kernel<NTH, F4, NTL, NTS, FLAG, TWO>), dim3(grid), dim3(NTH), 0, 0, A, B, C, F, S, N4);
  CHECK(hipDeviceSynchronize());
  const int reps = 100;
  CHECK(hipEventRecord(e0, 0));
  for (int i = 0; i < reps; ++i) hipLaunchKernelGGL((shape_kernel<NTH, F4, NTL, NTS, FLAG, TWO>), dim3(grid), dim3(NTH), 0, 0, A, B, C, F, S, N4);
  CHECK(hipEventRecord(e1, 0));
  CHECK(hipEventSynchronize(e1));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, e0, e1));
  const double us = ms * 1e3 / reps;
  const double bytes = (double)N4 * 16 * (TWO ? 3 : 2) + (FLAG ? N_ENV * 8.0 : 0.0);
  printf("%-44s grid %6d  %7.2f us/launch (events, back to back)  %6.2f TB/s\n", name, grid, us, bytes / us * 1e-6);
}

int main() {
  CHECK(hipMalloc(&A, N4 * 16));
  CHECK(hipMalloc(&B, N4 * 16));
  CHECK(hipMalloc(&C, N4 * 16));
  CHECK(hipMalloc(&F, N_ENV * 8));
  CHECK(hipMalloc(&S, 1 << 20));
  CHECK(hipMemset(A, 0, N4 * 16));
  CHECK(hipMemset(F, 0, N_ENV * 8));
  //        NTH F4  NTL    NTS    FLAG   TWO
  run<256, 2, true, true, false, true>("t256 f2 ntL ntS           1R 2W");
  run<256, 2, true, true, true, true>("t256 f2 ntL ntS flag      1R 2W");
  run<256, 2, false, true, true, true>("t256 f2 plL ntS flag      1R 2W");
  run<256, 2, false, false, true, true>("t256 f2 plL plS flag      1R 2W");
  run<256, 2, true, false, true, true>("t256 f2 ntL plS flag      1R 2W");
  run<256, 1, true, true, false, true>("t256 f1 ntL ntS           1R 2W");
  run<256, 4, true, true, false, true>("t256 f4 ntL ntS           1R 2W");
  run<256, 8, true, true, false, true>("t256 f8 ntL ntS           1R 2W");
  run<64, 2, true, true, false, true>("t64  f2 ntL ntS           1R 2W");
  run<64, 4, true, true, false, true>("t64  f4 ntL ntS           1R 2W");
  run<64, 8, true, true, false, true>("t64  f8 ntL ntS           1R 2W");
  run<64, 16, true, true, false, true>("t64  f16 ntL ntS          1R 2W");
  run<128, 4, true, true, false, true>("t128 f4 ntL ntS           1R 2W");
  run<512, 2, true, true, false, true>("t512 f2 ntL ntS           1R 2W");
  run<1024, 2, true, true, false, true>("t1024 f2 ntL ntS          1R 2W");
  run<256, 2, true, true, false, false>("t256 f2 ntL ntS           1R 1W");
  run<256, 4, true, true, false, false>("t256 f4 ntL ntS           1R 1W");
  run<64, 8, true, true, false, false>("t64  f8 ntL ntS           1R 1W");
  return 0;
}
