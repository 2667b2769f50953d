// Dev probe (GPU box): the memory pattern of k_measure at 16 real parameters (2^20 chains, float32) without arithmetic:
// 16 read rows (x), 16 + 32 read-modify-write rows (mean, observables), 136 RMW rows (covariance), 136 written rows
// (factor).  Component-major against tile-major for the two packed fields; and plain HBM streaming reads for scale.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 tools/dev/rows_probe2.hip -o tools/variants/rows_probe2
#include <hip/hip_runtime.h>
#include <cstdio>

template <bool TILE>
__global__ void __launch_bounds__(64) k_measure_pattern(const float *x, float *mean, float *obs, float *cov, float *fac, long long n) {
  const long long c = (long long)blockIdx.x * 64 + threadIdx.x;
  if (c >= n) return;
  float s = 0.f;
  {
    float a[16], m[16];
#pragma unroll
    for (int r = 0; r < 16; ++r) a[r] = x[r * n + c];
#pragma unroll
    for (int r = 0; r < 16; ++r) m[r] = mean[r * n + c];
#pragma unroll
    for (int r = 0; r < 16; ++r) { mean[r * n + c] = m[r] + a[r]; s += a[r]; }
  }
  {
    float o[32];
#pragma unroll
    for (int r = 0; r < 32; ++r) o[r] = obs[r * n + c];
#pragma unroll
    for (int r = 0; r < 32; ++r) obs[r * n + c] = o[r] + s;
  }
  float *pc = TILE ? cov + (c >> 6) * 136ll * 64 + (c & 63) : cov + c;
  float *pf = TILE ? fac + (c >> 6) * 136ll * 64 + (c & 63) : fac + c;
  const long long step = TILE ? 64 : n;
  float m[136];
#pragma unroll
  for (int k = 0; k < 136; ++k) m[k] = pc[k * step];
#pragma unroll
  for (int k = 0; k < 136; ++k) { m[k] = m[k] * 0.99f + s; pc[k * step] = m[k]; }
#pragma unroll
  for (int k = 0; k < 136; ++k) pf[k * step] = m[k] + 1.0f;
}

__global__ void __launch_bounds__(256) k_read(const float4 *p, long long n4, float *out) {
  float s = 0.f;
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
    const float4 v = p[i];
    s += v.x + v.y + v.z + v.w;
  }
  if (s == 123.456f) out[0] = s;
}
__global__ void __launch_bounds__(256) k_write(float4 *p, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) p[i] = float4{1.f, 2.f, 3.f, 4.f};
}
__global__ void __launch_bounds__(256) k_copy(const float4 *p, float4 *q, long long n4) {
  for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) q[i] = p[i];
}

template <class F>
float time_it(F &&launch) {
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  for (int i = 0; i < 3; ++i) launch();
  (void)hipEventRecord(e0);
  for (int i = 0; i < 20; ++i) launch();
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
  float ms; (void)hipEventElapsedTime(&ms, e0, e1); return ms / 20 * 1e3f;
}

int main() {
  const long long n = 1 << 20;
  float *x, *mean, *obs, *cov, *fac;
  (void)hipMalloc(&x, 4 * n * 16); (void)hipMalloc(&mean, 4 * n * 16); (void)hipMalloc(&obs, 4 * n * 32);
  (void)hipMalloc(&cov, 4 * n * 136); (void)hipMalloc(&fac, 4 * n * 136);
  (void)hipMemset(x, 0, 4 * n * 16); (void)hipMemset(mean, 0, 4 * n * 16); (void)hipMemset(obs, 0, 4 * n * 32);
  (void)hipMemset(cov, 0, 4 * n * 136); (void)hipMemset(fac, 0, 4 * n * 136);
  const dim3 grid((unsigned)(n / 64)), block(64);
  const double bytes = (double)n * 4 * (16 + 32 + 64 + 272 + 136);
  float us = time_it([&] { hipLaunchKernelGGL(k_measure_pattern<false>, grid, block, 0, 0, x, mean, obs, cov, fac, n); });
  printf("k_measure(16,0) pattern, component-major: %7.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
  us = time_it([&] { hipLaunchKernelGGL(k_measure_pattern<true>, grid, block, 0, 0, x, mean, obs, cov, fac, n); });
  printf("k_measure(16,0) pattern, packed fields tile-major: %7.1f us  %5.0f GB/s\n", us, bytes / us / 1e3);
  // plain streaming over the two 570 MB packed fields (beyond the 256 MiB Infinity Cache)
  const long long n4 = n * 136 / 4;
  for (int blocks : {1024, 2048, 4096, 16384}) {
    const float r = time_it([&] { hipLaunchKernelGGL(k_read, dim3(blocks), dim3(256), 0, 0, (const float4 *)cov, n4, x); });
    const float w = time_it([&] { hipLaunchKernelGGL(k_write, dim3(blocks), dim3(256), 0, 0, (float4 *)fac, n4); });
    const float c = time_it([&] { hipLaunchKernelGGL(k_copy, dim3(blocks), dim3(256), 0, 0, (const float4 *)cov, (float4 *)fac, n4); });
    printf("570 MB float4 streams, %5d blocks: read %6.1f us %5.0f GB/s | write %6.1f us %5.0f GB/s | copy %6.1f us %5.0f GB/s\n", blocks,
           r, n4 * 16.0 / r / 1e3, w, n4 * 16.0 / w / 1e3, c, 2 * n4 * 16.0 / c / 1e3);
  }
  return 0;
}
