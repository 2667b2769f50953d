// Dev probe (GPU box): the memory-only floor of the float64 headline kernel's access pattern: 18 rows of 8-byte values
// read, then written, one lane per chain, 64-thread blocks, 2^20 and 2^22 chains.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int RW>
__global__ void __launch_bounds__(64) k(double *rw, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  double x[RW];
#pragma unroll
  for (int r = 0; r < RW; ++r) x[r] = rw[r * n + c];
#pragma unroll
  for (int r = 0; r < RW; ++r) rw[r * n + c] = x[r] * 1.0000001 + 1e-30;
}
int main() {
  for (int lg : {17, 18, 19, 20, 21, 22}) {   // the ramp: time against chain count (fixed cost + slope)
    const long long n = 1ll << lg;
    double *rw; (void)hipMalloc(&rw, 8 * n * 18); (void)hipMemset(rw, 0, 8 * n * 18);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 20; ++i) hipLaunchKernelGGL(k<18>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, rw, n);
    (void)hipEventRecord(e0);
    for (int i = 0; i < 100; ++i) hipLaunchKernelGGL(k<18>, dim3((unsigned)(n / 64)), dim3(64), 0, 0, rw, n);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    printf("float64, 18 read-modify-write rows, 2^%d chains: %.1f us  %.0f GB/s\n", lg, ms * 10, 288.0 * n / (ms * 10) / 1e3);
    (void)hipFree(rw);
  }
  return 0;
}
