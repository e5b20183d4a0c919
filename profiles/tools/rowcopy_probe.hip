// Probe (r02): can a "column per lane" row walk -- lane j of a 3-wave group owns column j of the 187-float ShadowHand
// states row, reads its source array at a per-lane stride with dword loads and writes the row with dword stores, no LDS,
// all address math loop-invariant -- stream at the rate of the LDS-transposed float4 tiles?  Standalone: hipcc -O3
// --offload-arch=gfx950 rowcopy_probe.hip -o rowcopy_probe && ./rowcopy_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

struct Args { const float *dpos, *dvel, *ftp, *ftr, *ftv, *sens, *act; float* states; long n; };

template <int U, bool NT>
__global__ __launch_bounds__(256) void rowcopy(Args a) {
  const int tid = threadIdx.x;
  const long env0 = (long)blockIdx.x * 64;
  if (tid >= 192) return;
  const int j = tid;
  const bool valid = j < 187;
  const float* src; int K, c; float mul = 1.f;
  if (j < 24) { src = a.dpos; K = 24; c = j; }
  else if (j < 48) { src = a.dvel; K = 24; c = j - 24; mul = 0.2f; }
  else if (j < 72) { src = a.dpos; K = 24; c = j - 48; }   // stands in for the per-env block (LDS in the real kernel)
  else if (j < 87) { src = a.ftp; K = 15; c = j - 72; }
  else if (j < 107) { src = a.ftr; K = 20; c = j - 87; }
  else if (j < 137) { src = a.ftv; K = 30; c = j - 107; }
  else if (j < 167) { src = a.sens; K = 30; c = j - 137; mul = 10.f; }
  else { src = a.act; K = 20; c = (j < 187 ? j : 186) - 167; }
  const float* p = src + env0 * K + c;
  float* out = a.states + env0 * 187 + j;
#pragma unroll 1
  for (int i = 0; i < 64; i += U) {
    float v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (long)(i + u) * K) : p[(long)(i + u) * K];
#pragma unroll
    for (int u = 0; u < U; ++u) if (valid) { if (NT) __builtin_nontemporal_store(v[u] * mul, out + (long)(i + u) * 187); else out[(long)(i + u) * 187] = v[u] * mul; }
  }
}

template <int U, bool NT>
float run(Args a, int iters) {
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  const unsigned g = (unsigned)(a.n / 64);
  for (int i = 0; i < 3; ++i) hipLaunchKernelGGL((rowcopy<U, NT>), dim3(g), dim3(256), 0, 0, a);
  CK(hipEventRecord(e0));
  for (int i = 0; i < iters; ++i) hipLaunchKernelGGL((rowcopy<U, NT>), dim3(g), dim3(256), 0, 0, a);
  CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
  float ms; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / iters;
}

int main() {
  const long n = 1 << 20;
  Args a; a.n = n;
  float** ins[7] = {(float**)&a.dpos, (float**)&a.dvel, (float**)&a.ftp, (float**)&a.ftr, (float**)&a.ftv, (float**)&a.sens, (float**)&a.act};
  const int Ks[7] = {24, 24, 15, 20, 30, 30, 20};
  for (int k = 0; k < 7; ++k) { CK(hipMalloc(ins[k], n * Ks[k] * 4)); CK(hipMemset(*ins[k], 1, n * Ks[k] * 4)); }
  CK(hipMalloc(&a.states, n * 187 * 4));
  const double bytes = (double)n * (163 + 24 + 187) * 4;  // the stand-in block re-reads dpos (cache-served)
  printf("U=4  plain %.1f us  %.2f TB/s\n", run<4, false>(a, 20) * 1e3, bytes / run<4, false>(a, 20) / 1e9);
  printf("U=8  plain %.1f us  %.2f TB/s\n", run<8, false>(a, 20) * 1e3, bytes / run<8, false>(a, 20) / 1e9);
  printf("U=16 plain %.1f us  %.2f TB/s\n", run<16, false>(a, 20) * 1e3, bytes / run<16, false>(a, 20) / 1e9);
  printf("U=8  nt    %.1f us  %.2f TB/s\n", run<8, true>(a, 20) * 1e3, bytes / run<8, true>(a, 20) / 1e9);
  printf("U=16 nt    %.1f us  %.2f TB/s\n", run<16, true>(a, 20) * 1e3, bytes / run<16, true>(a, 20) / 1e9);
  printf("U=32 nt    %.1f us  %.2f TB/s\n", run<32, true>(a, 20) * 1e3, bytes / run<32, true>(a, 20) / 1e9);
  return 0;
}
